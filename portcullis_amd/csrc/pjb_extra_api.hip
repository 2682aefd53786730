// pjb_extra_api.hip -- the part of the C ABI behind `junc --extra` (depth, flanking counts, name multiplicities: pjb_extra_finish), `bamfilt`
// (pjb_filter_*) and `filt`'s feature rows (pjb_filt_features); kernels in pjb_extra.hip.h.
#define PJB_KERNELS_EXTRA 1
#include "pjb_host.hip.h"

static int extra_contig_dense(pjb_ctx *c, int32_t tid, std::vector<DevBatch> &batches, int64_t n_reads, u64 n_spliced, u32 P, u32 J,
                              const u32 *sidx, const u32 *jid_sorted, const u32 *pair_g, size_t row_base, bool codes_in_table) {
    hipStream_t st = c->stream;
    const int32_t L = c->ref_len[(size_t)tid];
    const size_t N = (size_t)n_reads;
    ExtraContig X;
    X.dense = true;
    X.codes_in_table = codes_in_table;
    X.tid = tid;
    X.len = L;
    X.row_base = row_base;
    X.n_rows = J;
    X.n_pairs = P;
    int rc;
    if ((rc = ensure(c, c->b_xtotal, 8))) return rc;
    if ((rc = ensure(c, c->x_pos, N * 4 + 16))) return rc;
    if ((rc = ensure(c, c->x_endx, N * 4 + 16))) return rc;
    if ((rc = ensure(c, c->x_q, N + 16))) return rc;
    if ((rc = ensure(c, c->x_prefq, (N + 1) * 4))) return rc;
    if ((rc = ensure(c, c->x_ce, ((size_t)L + 2) * 4))) return rc;
    if ((rc = ensure(c, c->x_zlist, (size_t)X_ZCAP * 4))) return rc;
    if ((rc = ensure(c, c->x_cnt, sizeof(ExtraCounters)))) return rc;
    struct Guard { // frees what this contig allocated unless it is handed over to the context
        ExtraContig *x;
        ~Guard() {
            if (!x) return;
            if (x->cover) (void)hipFree(x->cover);
            if (x->xr) (void)hipFree(x->xr);
            if (x->pair_code) (void)hipFree(x->pair_code);
            if (x->pair_row) (void)hipFree(x->pair_row);
            if (x->spl_codes) (void)hipFree(x->spl_codes);
        }
    } guard{&X};
    if (hipMalloc((void **)&X.cover, ((size_t)L + 2) * 4) != hipSuccess) return fail(c, PJB_ERR_NOMEM, "extra: depth array of target %d", tid);
    if (hipMalloc((void **)&X.spl_codes, std::max<size_t>((size_t)n_spliced, 1) * 8) != hipSuccess)
        return fail(c, PJB_ERR_NOMEM, "extra: name codes of target %d", tid);
    HIP_TRY(c, hipMemsetAsync(X.cover, 0, ((size_t)L + 2) * 4, st));
    HIP_TRY(c, hipMemsetAsync(c->x_ce.p, 0, ((size_t)L + 2) * 4, st));
    HIP_TRY(c, hipMemsetAsync((uint8_t *)c->x_q.p + N, 0, 1, st));
    ExtraCounters hc;
    memset(&hc, 0, sizeof hc);
    hc.hot_first = 0xffffffffu;
    HIP_TRY(c, hipMemcpyAsync(c->x_cnt.p, &hc, sizeof hc, hipMemcpyHostToDevice, st));
    ExtraCounters *d_cnt = (ExtraCounters *)c->x_cnt.p;
    int32_t *x_pos = (int32_t *)c->x_pos.p, *x_endx = (int32_t *)c->x_endx.p;
    uint8_t *x_q = (uint8_t *)c->x_q.p;
    u32 *prefq = (u32 *)c->x_prefq.p, *ce = (u32 *)c->x_ce.p;
    for (auto &b : batches)
        LAUNCH(c, "kx_classify", kx_classify, dim3((unsigned)((b.n + 255) / 256)), dim3(256), b, L, x_pos, x_endx, x_q, ce,
               (int32_t *)X.cover, (u32 *)c->x_zlist.p, X_ZCAP, d_cnt);
    {   // the spliced records' name codes, through the tile lists the contig's first kernels left in its slot
        CtlSlot &S = c->sl[c->cur_slot];
        u32 n_tiles = 0;
        for (auto &b : batches) n_tiles = std::max<u32>(n_tiles, b.tile_base + (u32)((b.n + K1_TILE - 1) / K1_TILE));
        if ((rc = ensure(c, c->x_tileoff, (size_t)n_tiles * 4 + 16))) return rc;
        LAUNCH(c, "kx_spliced_offsets", kx_spliced_offsets, dim3(1), dim3(1024), (const TileStats *)S.tile_stats.p, n_tiles, (u32 *)c->x_tileoff.p, d_cnt);
        for (auto &b : batches)
            LAUNCH(c, "kx_spliced_codes", kx_spliced_codes, dim3((unsigned)((b.n + K1_TILE - 1) / K1_TILE)), dim3(256), b, (const TileStats *)S.tile_stats.p,
                   (const u32 *)S.splidx.p, (const u32 *)c->x_tileoff.p, X.spl_codes);
    }
    if ((rc = run_scan(c, "kx_ends", ArrU32Fn{ce}, ExclusiveU32Sink{ce}, (u64)L + 2, (u64 *)c->b_xtotal.p))) return rc;
    if ((rc = run_scan(c, "kx_unspl", ArrU8Fn{x_q}, ExclusiveU32Sink{prefq}, (u64)N + 1, (u64 *)c->b_xtotal.p))) return rc;
    LAUNCH(c, "kx_cap_bound", kx_cap_bound, dim3((unsigned)((N + 255) / 256)), dim3(256), (const int32_t *)x_pos, (const uint8_t *)x_q,
           (const u32 *)prefq, (const u32 *)ce, (u32)N, L, (u32 *)nullptr, d_cnt);
    u32 n_unspl = 0;
    HIP_TRY(c, hipMemcpyAsync(&hc, d_cnt, sizeof hc, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(&n_unspl, prefq + N, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    if (hc.n_zero > X_ZCAP)
        return fail(c, PJB_ERR_ARG, "extra: target %d has %u mapped records without a reference span (limit %u)", tid, hc.n_zero, X_ZCAP);
    if (hc.max_buffered + 2 > PLP_MAXCNT) { // the pileup's record cap may bite: replay it over the hot span
        if ((rc = ensure(c, c->x_bound, N * 4 + 16))) return rc;
        if ((rc = ensure(c, c->x_de, ((size_t)L + 2) * 4))) return rc;
        if ((rc = ensure(c, c->x_dropped, N + 16))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->x_de.p, 0, ((size_t)L + 2) * 4, st));
        HIP_TRY(c, hipMemsetAsync(c->x_dropped.p, 0, N + 16, st));
        LAUNCH(c, "kx_cap_bound", kx_cap_bound, dim3((unsigned)((N + 255) / 256)), dim3(256), (const int32_t *)x_pos,
               (const uint8_t *)x_q, (const u32 *)prefq, (const u32 *)ce, (u32)N, L, (u32 *)c->x_bound.p, d_cnt);
        LAUNCH(c, "kx_cap_replay", kx_cap_replay, dim3(1), dim3(64), (const int32_t *)x_pos, (const int32_t *)x_endx, (const uint8_t *)x_q,
               (const u32 *)c->x_bound.p, (u32)N, L, (u32 *)c->x_de.p, (uint8_t *)c->x_dropped.p, d_cnt);
        for (auto &b : batches)
            LAUNCH(c, "kx_undo_dropped", kx_undo_dropped, dim3((unsigned)((b.n + 255) / 256)), dim3(256), b, L,
                   (const uint8_t *)c->x_dropped.p, (int32_t *)X.cover);
    }
    if ((rc = run_scan(c, "kx_depth", ArrI32Fn{(const int32_t *)X.cover}, InclusiveU32Sink{X.cover}, (u64)L + 1, (u64 *)c->b_xtotal.p)))
        return rc;
    X.has_unspliced = n_unspl > 0;
    X.n_spl = hc.n_spliced;
    if (J > 0) {
        if (hipMalloc((void **)&X.xr, (size_t)J * sizeof(ExtraRow)) != hipSuccess) return fail(c, PJB_ERR_NOMEM, "extra: rows of target %d", tid);
        HIP_TRY(c, hipMemsetAsync(X.xr, 0, (size_t)J * sizeof(ExtraRow), st));
        LAUNCH(c, "kx_flank", kx_flank, dim3((J + 255) / 256), dim3(256), (const pjb_junction_row *)c->sl[c->cur_slot].rows.p, J, (const int32_t *)x_pos,
               (u32)N, (const u32 *)prefq, (const u32 *)ce, L, (const u32 *)c->x_zlist.p, (const ExtraCounters *)d_cnt, X_ZCAP, X.xr);
        if (hipMalloc((void **)&X.pair_code, (size_t)P * 8) != hipSuccess || hipMalloc((void **)&X.pair_row, (size_t)P * 4) != hipSuccess)
            return fail(c, PJB_ERR_NOMEM, "extra: pair codes of target %d", tid);
        LAUNCH(c, "kx_pair_codes", kx_pair_codes, dim3((P + 255) / 256), dim3(256), sidx, jid_sorted, pair_g,
               (const DevBatch *)c->sl[c->cur_slot].batches.p, (int)batches.size(), P, (u32)row_base, X.pair_code, X.pair_row);
    }
    HIP_TRY(c, hipStreamSynchronize(st));
    if (c->ktime) ev_collect(c, MISC_POOL);
    c->xc.push_back(X);
    guard.x = nullptr;
    return PJB_OK;
}

#define XTRACE(what)                                                                                                              \
    do {                                                                                                                          \
        if (xtrace) {                                                                                                             \
            (void)hipStreamSynchronize(st);                                                                                       \
            const auto now_ = std::chrono::steady_clock::now();                                                                   \
            fprintf(stderr, "[xtrace] %-28s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now_ - xt0).count());     \
            xt0 = now_;                                                                                                           \
        }                                                                                                                         \
    } while (0)
// room in the name table for `add` more codes (load <= 1/2): a larger table takes over the old one's names
static int name_table_reserve(pjb_ctx *c, size_t add) {
    hipStream_t st = c->stream;
    const size_t need = (c->x_tab_n + add) + (c->x_tab_n + add) / 2 + 1; // load <= 2/3
    if (c->x_tab_n == 0 && c->x_tab_slots >= need) { // first codes of a file: wipe
        if (add) HIP_TRY(c, hipMemsetAsync(c->x_tab.p, 0xff, c->x_tab_slots * sizeof(NameSlot), st));
        return PJB_OK;
    }
    if (c->x_tab_slots >= need) return PJB_OK;
    if (need > 0xfffffff0ull) return fail(c, PJB_ERR_ARG, "extra: more than 2^31 spliced records");
    const size_t slots = std::min<size_t>(std::max<size_t>(2 * need + 16, 1024), 0xfffffff0ull); // (twice what is needed now: a file's targets arrive one by one)
    Buf nb;
    int rc = ensure(c, nb, slots * sizeof(NameSlot));
    if (rc) return rc;
    HIP_TRY(c, hipMemsetAsync(nb.p, 0xff, slots * sizeof(NameSlot), st));
    if (c->x_tab_n)
        LAUNCH(c, "kx_name_rehash", kx_name_rehash, dim3((unsigned)((c->x_tab_slots + 255) / 256)), dim3(256), (const NameSlot *)c->x_tab.p,
               (u32)c->x_tab_slots, (NameSlot *)nb.p, (u32)slots);
    if (c->x_tab.p) {
        HIP_TRY(c, hipStreamSynchronize(st));
        release(c->x_tab);
    }
    c->x_tab = nb;
    c->x_tab_slots = slots;
    return PJB_OK;
}
static int name_table_insert(pjb_ctx *c, const u64 *codes, u32 n) {
    if (!n) return PJB_OK;
    int rc = name_table_reserve(c, n);
    if (rc) return rc;
    LAUNCH(c, "kx_name_insert", kx_name_insert4, dim3((n + 1023) / 1024), dim3(256), codes, n, (NameSlot *)c->x_tab.p, (u32)c->x_tab_slots);
    c->x_tab_n += n;
    return PJB_OK;
}

// The same per-target work without an array of the target's length (pjb_extra.hip.h, "the sparse path"), in two parts.
// extra_pre needs the records only: queued on the service stream when the target's chain is queued, it runs beside the
// chains.  extra_contig needs the chain's rows and sorted pairs: queued when the chain is collected, beside the chains of
// the targets queued after this one; one wait at its end.  A target where the pileup's cap may bite goes through
// extra_contig_dense instead.
int extra_pre(pjb_ctx *c, Flight &f) {
    if (f.x_pre || c->extra_dense_only || f.empty) return PJB_OK;
    hipStream_t st = c->stream;
    CtlSlot &S = c->sl[f.slot];
    const size_t N = (size_t)f.n_reads;
    int rc;
    f.x_gap_cap = (u32)std::min<size_t>(N / 16 + 1024, 0x7fffffffu);
    f.x_spos = (int32_t *)xarena_alloc(c, N * 4 + 16); // (compacted: the records with a span)
    f.x_send = (int32_t *)xarena_alloc(c, N * 4 + 16);
    f.x_gapoff = (u32 *)xarena_alloc(c, (N / 256 + 2) * 4);
    f.x_gaps = (Gap *)xarena_alloc(c, (size_t)f.x_gap_cap * sizeof(Gap));
    if (!f.x_spos || !f.x_send || !f.x_gapoff || !f.x_gaps)
        return fail(c, PJB_ERR_NOMEM, "extra: no device memory for what target %d keeps (%zu records)", f.tid, N);
    if ((rc = ensure(c, S.x_q, N + 16)) || (rc = ensure(c, S.x_spos, N * 4 + 16)) || (rc = ensure(c, S.x_send, N * 4 + 16)) ||
        (rc = ensure(c, S.x_gapoff, (N / 256 + 2) * 4)) || (rc = ensure(c, S.x_zlist, (size_t)X_ZCAP * 4)) ||
        (rc = ensure(c, S.x_scnt, sizeof(SparseCounters) + sizeof(ExtraCounters))))
        return rc;
    uint8_t *q = (uint8_t *)S.x_q.p;
    SparseCounters *d_cnt = (SparseCounters *)S.x_scnt.p;
    if (f.x_k1) HIP_TRY(c, hipStreamWaitEvent(st, S.ev_xk1, 0)); // (the chain's k1_count classified the records)
    else {
        HIP_TRY(c, hipMemsetAsync(q + N, 0, 1, st));
        HIP_TRY(c, hipMemsetAsync(d_cnt, 0, sizeof(SparseCounters) + sizeof(ExtraCounters), st));
        for (auto &b : f.batches)
            LAUNCH(c, "kx_classify_sparse", kx_classify_sparse, dim3((unsigned)((b.n + 255) / 256)), dim3(256), b, (int32_t *)S.x_spos.p, (int32_t *)S.x_send.p,
                   q, (u32 *)S.x_zlist.p, X_ZCAP, d_cnt);
    }
    if ((rc = run_scan(c, "kx_spans", SparseFn{q},
                       SparseSink{f.x_spos, f.x_send, (u32 *)S.x_gapoff.p, f.x_gapoff, (const int32_t *)S.x_spos.p, (const int32_t *)S.x_send.p, q}, (u64)N + 1,
                       &d_cnt->total)))
        return rc;
    for (auto &b : f.batches)
        if (b.n > 0)
        {
            const u32 nblk = (u32)((((u64)b.base + (u64)b.n + 255) >> 8) - (b.base >> 8));
            LAUNCH(c, "kx_gaps", kx_gaps, dim3(std::min<u32>(nblk, 2048)), dim3(256), b, (const uint8_t *)q, (u32)N, (const u32 *)S.x_gapoff.p, f.x_gaps,
                   f.x_gap_cap, d_cnt, nblk);
        }
    if (N >= PLP_MAXCNT)
        LAUNCH(c, "kx_cap_check", kx_cap_check, dim3((unsigned)((N + 255) / 256)), dim3(256), (const int32_t *)f.x_spos, d_cnt);
    f.x_pre = true;
    return PJB_OK;
}

int extra_contig(pjb_ctx *c, Flight &f, int32_t tid, u64 n_spliced, u32 P, u32 J, size_t row_base) {
    std::vector<DevBatch> &batches = f.batches;
    if (c->extra_dense_only) return extra_contig_dense(c, tid, batches, f.n_reads, n_spliced, P, J, f.sidx, f.jid_sorted, f.pr.g, row_base, false);
    int rc;
    if ((rc = extra_pre(c, f))) return rc;
    hipStream_t st = c->stream;
    const int32_t L = c->ref_len[(size_t)tid];
    CtlSlot &S = c->sl[f.slot];
    ExtraContig X;
    X.tid = tid;
    X.len = L;
    X.row_base = row_base;
    X.n_rows = J;
    X.n_pairs = P;
    X.xr = J ? (ExtraRow *)xarena_alloc(c, (size_t)J * sizeof(ExtraRow)) : nullptr;
    X.pair_code = J ? (u64 *)xarena_alloc(c, (size_t)P * 8 + 16) : nullptr;
    X.pair_row = J ? (u32 *)xarena_alloc(c, (size_t)P * 4 + 16) : nullptr;
    if (J && (!X.xr || !X.pair_code || !X.pair_row)) return fail(c, PJB_ERR_NOMEM, "extra: no device memory for the pairs of target %d", tid);
    if ((rc = ensure(c, S.x_codes, std::max<size_t>((size_t)n_spliced, 1) * 8))) return rc;
    const bool xtrace = getenv("PJB_XTRACE") != nullptr;
    auto xt0 = std::chrono::steady_clock::now();
    XTRACE("post: pre-part done");
    SparseCounters *d_cnt = (SparseCounters *)S.x_scnt.p;
    ExtraCounters *d_xcnt = (ExtraCounters *)(d_cnt + 1);
    {   // the spliced records' name codes, through the tile lists the target's first kernels left in its slot -> the name table
        u32 n_tiles = 0;
        for (auto &b : batches) n_tiles = std::max<u32>(n_tiles, b.tile_base + (u32)((b.n + K1_TILE - 1) / K1_TILE));
        if ((rc = ensure(c, c->x_tileoff, (size_t)n_tiles * 4 + 16))) return rc;
        LAUNCH(c, "kx_spliced_offsets", kx_spliced_offsets, dim3(1), dim3(1024), (const TileStats *)S.tile_stats.p, n_tiles, (u32 *)c->x_tileoff.p, d_xcnt);
        for (auto &b : batches)
            LAUNCH(c, "kx_spliced_codes", kx_spliced_codes, dim3((unsigned)((b.n + K1_TILE - 1) / K1_TILE)), dim3(256), b, (const TileStats *)S.tile_stats.p,
                   (const u32 *)S.splidx.p, (const u32 *)c->x_tileoff.p, (u64 *)S.x_codes.p);
        XTRACE("post: codes");
        if ((rc = name_table_insert(c, (const u64 *)S.x_codes.p, (u32)n_spliced))) return rc;
        X.codes_in_table = true;
        XTRACE("post: insert");
    }
    if (J > 0) {
        HIP_TRY(c, hipMemsetAsync(X.xr, 0, (size_t)J * sizeof(ExtraRow), st));
        LAUNCH(c, "kx_flank_sparse", kx_flank_sparse, dim3((J + 255) / 256), dim3(256), (const pjb_junction_row *)S.rows.p, J,
               (const int32_t *)f.x_spos, (const int32_t *)f.x_send, L, (const u32 *)S.x_zlist.p, (const SparseCounters *)d_cnt, X_ZCAP, X.xr);
        LAUNCH(c, "kx_pair_codes", kx_pair_codes, dim3((P + 255) / 256), dim3(256), f.sidx, f.jid_sorted, f.pr.g, (const DevBatch *)S.batches.p,
               (int)batches.size(), P, (u32)row_base, X.pair_code, X.pair_row);
    }
    // one wait: the counters decide whether the sparse answer stands
    SparseCounters &hc = *(SparseCounters *)(S.pub + PUB_XCNT_AT);
    ExtraCounters &hx = *(ExtraCounters *)(S.pub + PUB_XCNT_AT + sizeof(SparseCounters));
    XTRACE("post: flank + pair codes");
    HIP_TRY(c, hipMemcpyAsync(&hc, d_cnt, sizeof(SparseCounters) + sizeof(ExtraCounters), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    XTRACE("post: counters");
    if (c->ktime) ev_collect(c, MISC_POOL);
    if (hc.n_zero > X_ZCAP)
        return fail(c, PJB_ERR_ARG, "extra: target %d has %u mapped records without a reference span (limit %u)", tid, hc.n_zero, X_ZCAP);
    if (hx.n_spliced != (u32)n_spliced)
        return fail(c, PJB_ERR_STATE, "extra: target %d: %u spliced records in the tile lists, the chain counted %llu", tid, hx.n_spliced, (unsigned long long)n_spliced);
    if (hc.need_dense) // the pileup's cap may bite (or the gap list is too small): the depth vector, as in round 2
        return extra_contig_dense(c, tid, batches, f.n_reads, n_spliced, P, J, f.sidx, f.jid_sorted, f.pr.g, row_base, true);
    X.has_unspliced = (u32)hc.total > 0;
    X.n_spl = hx.n_spliced;
    X.sparse = SparseDepth{f.x_spos, f.x_send, f.x_gaps, f.x_gapoff, (u32)hc.total, (u32)(hc.total >> 32), hc.max_span, hc.max_gap};
    c->xc.push_back(X);
    return PJB_OK;
}

// The device work of one contig, queued in one go.  The host does not learn a single count while the kernels run:
// buffers and grids are sized from LIMITS (pair_limit, junc_limit, the key format kf), the kernels read the actual
// counts from the control block in device memory (ContigStats) and stand still when a limit is exceeded.  Nothing
// here waits for the device: the last kernels (rows stream) write rows and control block into page-locked host memory
// and pjb_finish_contig_end waits for their event -- by which time the next contig may be queued behind this one.
extern "C" {
int pjb_extra_finish(pjb_ctx *c, const pjb_extra_row **rows_out, int64_t *n_out) {
    if (!c || !rows_out || !n_out) return PJB_ERR_ARG;
    const bool xtrace = getenv("PJB_XTRACE") != nullptr;
    auto xt0 = std::chrono::steady_clock::now();
    if (!c->extra) return fail(c, PJB_ERR_STATE, "pjb_extra_finish: the context was not created with PJB_FLAG_EXTRA");
    if (!c->open.empty()) return fail(c, PJB_ERR_STATE, "pjb_extra_finish: target %d is still open", c->open.begin()->first);
    if (c->n_fl) return fail(c, PJB_ERR_STATE, "pjb_extra_finish: target %d is still queued", c->fl[0].tid);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    int rc;
    const size_t Jall = c->rows_n;
    *rows_out = c->xrows_pinned;
    *n_out = (int64_t)Jall;
    if (Jall == 0) return PJB_OK;
    if (Jall > c->xrows_pinned_cap) {
        if (c->xrows_pinned) (void)hipHostFree(c->xrows_pinned);
        c->xrows_pinned = nullptr;
        c->xrows_pinned_cap = 0;
        const size_t cap = Jall + Jall / 4 + 1024;
        HIP_TRY(c, hipHostMalloc((void **)&c->xrows_pinned, cap * sizeof(pjb_extra_row), hipHostMallocDefault));
        c->xrows_pinned_cap = cap;
    }
    *rows_out = c->xrows_pinned;
    // every target's flanking counts into one table, parallel to the row table in HBM (which the kernels below read)
    if ((rc = ensure(c, c->x_xrall, Jall * (sizeof(ExtraRow) + sizeof(pjb_extra_row))))) return rc;
    ExtraRow *xr = (ExtraRow *)c->x_xrall.p;
    pjb_extra_row *xout = (pjb_extra_row *)(xr + Jall);
    HIP_TRY(c, hipMemsetAsync(xr, 0, Jall * sizeof(ExtraRow), st));
    for (auto &x : c->xc)
        if (x.n_rows) HIP_TRY(c, hipMemcpyAsync(xr + x.row_base, x.xr, x.n_rows * sizeof(ExtraRow), hipMemcpyDeviceToDevice, st));
    XTRACE("finish: memset + copies");
    // ---- splicedAlignmentMap over every spliced record of the file (src/junction_builder.cc:168-176): the targets' codes went
    // into the table as the targets were collected (a target of the dense path: now)
    for (auto &x : c->xc)
        if (!x.codes_in_table) {
            if ((rc = name_table_insert(c, (const u64 *)x.spl_codes, x.n_spl))) return rc;
            x.codes_in_table = true;
        }
    if (c->x_tab_n)
        for (auto &x : c->xc)
            if (x.n_pairs && x.n_rows)
                LAUNCH(c, "kx_name_sum", kx_name_sum, dim3((x.n_pairs + 1023) / 1024), dim3(256), (const u64 *)x.pair_code, (const u32 *)x.pair_row,
                       x.n_pairs, (const NameSlot *)c->x_tab.p, (u32)c->x_tab_slots, xr);
    // ---- JunctionSystem::calcCoverage (lib/src/junction_system.cc:231-242).  DepthParser::loadNextBatch
    // (lib/src/depth_parser.cc:112-164) returns the vector of the target it started in, but by then `last`
    // names the target the pileup has moved on to, and getCurrentRefIndex() selects THAT target's junctions:
    // every batch is applied to the junctions of the next target that has unspliced records; only the final
    // batch (the pileup ended inside it) meets its own junctions, after they were first given the previous
    // target's.  Targets without unspliced records never appear.
    XTRACE("finish: name sums");
    std::vector<const ExtraContig *> T;
    for (auto &x : c->xc)
        if (x.has_unspliced) T.push_back(&x);
    std::sort(T.begin(), T.end(), [](const ExtraContig *a, const ExtraContig *b) { return a->tid < b->tid; });
    const pjb_junction_row *rows = c->rows_table;
    for (size_t k = 0; k < T.size(); k++) {
        const ExtraContig &x = *T[k];
        if (!x.n_rows) continue;
        const ExtraContig *src = (k + 1 == T.size()) ? &x : (k > 0 ? T[k - 1] : nullptr);
        if (!src) continue; // the first target's junctions are never visited (unless it is also the last)
        if (src->dense)
            LAUNCH(c, "kx_coverage", kx_coverage, dim3((unsigned)((x.n_rows + 255) / 256)), dim3(256), rows, (u32)x.row_base, (u32)x.n_rows,
                   (const u32 *)src->cover, src->len, xr);
        else
            LAUNCH(c, "kx_coverage_sparse", kx_coverage_sparse, dim3((unsigned)((x.n_rows + 255) / 256)), dim3(256), rows, (u32)x.row_base,
                   (u32)x.n_rows, src->sparse, src->len, xr);
    }
    XTRACE("finish: coverage");
    LAUNCH(c, "kx_rows_out", kx_rows_out, dim3((unsigned)((Jall + 255) / 256)), dim3(256), rows, (const ExtraRow *)xr, (u32)Jall, xout);
    XTRACE("finish: rows_out");
    HIP_TRY(c, hipMemcpyAsync(c->xrows_pinned, xout, Jall * sizeof(pjb_extra_row), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    XTRACE("finish: D2H");
    if (c->ktime) ev_collect(c, MISC_POOL);
    return PJB_OK;
}

int pjb_filter_set_junctions(pjb_ctx *c, int32_t tid, const uint64_t *sorted_keys, int64_t n_keys) {
    if (!c) return PJB_ERR_ARG;
    if (tid < 0 || n_keys < 0 || n_keys > 0xfffffff0ll || (n_keys > 0 && !sorted_keys))
        return fail(c, PJB_ERR_ARG, "pjb_filter_set_junctions: bad arguments (tid %d)", tid);
    for (int64_t i = 1; i < n_keys; i++)
        if (sorted_keys[i - 1] >= sorted_keys[i]) return fail(c, PJB_ERR_ARG, "pjb_filter_set_junctions: keys must be strictly ascending");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    auto it = c->filter_keys.find(tid);
    if (it != c->filter_keys.end()) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (it->second.first) (void)hipFree(it->second.first);
        c->filter_keys.erase(it);
    }
    u64 *d = nullptr;
    if (n_keys) {
        if (hipMalloc((void **)&d, (size_t)n_keys * 8) != hipSuccess) return fail(c, PJB_ERR_NOMEM, "pjb_filter_set_junctions: %lld keys", (long long)n_keys);
        hipError_t e = hipMemcpy(d, sorted_keys, (size_t)n_keys * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(d);
            return fail(c, PJB_ERR_HIP, "pjb_filter_set_junctions: %s", hipGetErrorString(e));
        }
    }
    c->filter_keys[tid] = std::make_pair(d, (u32)n_keys);
    return PJB_OK;
}

int pjb_filter_batch(pjb_ctx *c, int32_t tid, const pjb_batch *b, int32_t clip_mode, uint8_t *codes_out) {
    if (!c) return PJB_ERR_ARG;
    if (!b || b->n_reads < 0 || (b->n_reads > 0 && (!b->pos || !b->cig_off || !b->cigar || !codes_out)))
        return fail(c, PJB_ERR_ARG, "pjb_filter_batch: bad batch");
    if (clip_mode < PJB_CLIP_HARD || clip_mode > PJB_CLIP_COMPLETE) return fail(c, PJB_ERR_ARG, "pjb_filter_batch: bad clip mode %d", clip_mode);
    if (b->n_reads == 0) return PJB_OK;
    if (b->n_reads > 0xfffffff0ll) return fail(c, PJB_ERR_ARG, "pjb_filter_batch: batch too large");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    const size_t n = (size_t)b->n_reads, n_ops = b->cig_off[n];
    int rc;
    if ((rc = ensure(c, c->f_pos, n * 4))) return rc;
    if ((rc = ensure(c, c->f_cigoff, (n + 1) * 4))) return rc;
    if ((rc = ensure(c, c->f_cigar, n_ops * 4 + 16))) return rc;
    if ((rc = ensure(c, c->f_codes, n + 16))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->f_pos.p, b->pos, n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->f_cigoff.p, b->cig_off, (n + 1) * 4, hipMemcpyHostToDevice, st));
    if (n_ops) HIP_TRY(c, hipMemcpyAsync(c->f_cigar.p, b->cigar, n_ops * 4, hipMemcpyHostToDevice, st));
    const u64 *keys = nullptr;
    u32 n_keys = 0;
    auto it = c->filter_keys.find(tid);
    if (it != c->filter_keys.end()) {
        keys = it->second.first;
        n_keys = it->second.second;
    }
    LAUNCH(c, "kf_filter", kf_filter, dim3((unsigned)((n + 255) / 256)), dim3(256), (const int32_t *)c->f_pos.p, (const u32 *)c->f_cigoff.p,
           (const u32 *)c->f_cigar.p, (u32)n, keys, n_keys, (int)clip_mode, (uint8_t *)c->f_codes.p);
    HIP_TRY(c, hipMemcpyAsync(codes_out, c->f_codes.p, n, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    if (c->ktime) ev_collect(c, MISC_POOL);
    return PJB_OK;
}

int pjb_filt_features(pjb_ctx *c, const pjb_junction_row *rows, int64_t n_rows, double mean_read_length, uint32_t l95,
                      const pjb_markov_models *models, double *features_out) {
    if (!c) return PJB_ERR_ARG;
    if (n_rows < 0 || (n_rows > 0 && (!rows || !features_out)) || !models || n_rows > 0xfffffff0ll)
        return fail(c, PJB_ERR_ARG, "pjb_filt_features: bad arguments");
    if (n_rows == 0) return PJB_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    const size_t n = (size_t)n_rows;
    int rc;
    if ((rc = ensure(c, c->g_rows, n * sizeof(pjb_junction_row)))) return rc;
    if ((rc = ensure(c, c->g_models, ((size_t)6 * PJB_KMER_TABLE + 2 * PJB_PW_LEN * 5) * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->g_refs, std::max<size_t>(c->contigs.size(), 1) * sizeof(GenomeRef)))) return rc;
    if ((rc = ensure(c, c->g_out, n * PJB_N_FEATURES * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->g_bad, sizeof(int)))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->g_rows.p, rows, n * sizeof(pjb_junction_row), hipMemcpyHostToDevice, st));
    DevModels M;
    memset(&M, 0, sizeof M);
    double *dm = (double *)c->g_models.p;
    const double *src[8] = {models->exon, models->intron, models->donor_t, models->donor_f, models->acceptor_t, models->acceptor_f,
                            models->donor_pw, models->acceptor_pw};
    const double **dst[8] = {&M.exon, &M.intron, &M.don_t, &M.don_f, &M.acc_t, &M.acc_f, &M.don_pw, &M.acc_pw};
    size_t at = 0;
    for (int k = 0; k < 8; k++) {
        const size_t cnt = k < 6 ? (size_t)PJB_KMER_TABLE : (size_t)PJB_PW_LEN * 5;
        if (src[k]) {
            HIP_TRY(c, hipMemcpyAsync(dm + at, src[k], cnt * sizeof(double), hipMemcpyHostToDevice, st));
            *dst[k] = dm + at;
        }
        at += cnt;
    }
    M.exon_size = models->exon ? models->exon_size : 0;
    M.intron_size = models->intron ? models->intron_size : 0;
    M.don_pw_size = models->donor_pw ? models->donor_pw_size : 0;
    M.acc_pw_size = models->acceptor_pw ? models->acceptor_pw_size : 0;
    std::vector<GenomeRef> refs(std::max<size_t>(c->contigs.size(), 1));
    for (size_t t = 0; t < c->contigs.size(); t++) {
        refs[t].d = c->contigs[t].present ? c->contigs[t].d : nullptr;
        refs[t].len = (int32_t)c->contigs[t].len;
    }
    HIP_TRY(c, hipMemcpyAsync(c->g_refs.p, refs.data(), refs.size() * sizeof(GenomeRef), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemsetAsync(c->g_bad.p, 0, sizeof(int), st));
    LAUNCH(c, "kg_features", kg_features, dim3((unsigned)((n + 255) / 256)), dim3(256), (const pjb_junction_row *)c->g_rows.p, (u32)n,
           (const GenomeRef *)c->g_refs.p, (int)c->contigs.size(), M, mean_read_length, (u32)l95, (double *)c->g_out.p, (int *)c->g_bad.p);
    int bad = 0;
    HIP_TRY(c, hipMemcpyAsync(features_out, c->g_out.p, n * PJB_N_FEATURES * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(&bad, c->g_bad.p, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    if (c->ktime) ev_collect(c, MISC_POOL);
    if (bad) return fail(c, PJB_ERR_STATE, "pjb_filt_features: a junction lies on a target whose genome was not uploaded");
    return PJB_OK;
}

} // extern "C"
