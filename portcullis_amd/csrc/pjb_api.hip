// pjb_api.hip -- C ABI (include/portcullis_amd.h) over the HIP kernels: contexts, genomes, batches, the kernel chains, rows.
// One context = one HIP device + its streams + grow-only scratch.  (--extra / bamfilt / filt: pjb_extra_api.hip; BGZF and BAM: pjb_ingest_api.hip.)
#define PJB_KERNELS_CHAIN 1
#include "pjb_host.hip.h"

static int slot_init(pjb_ctx *c, int k);
static void aux_streams(pjb_ctx *c) { // the rows stream and the row mirror's (15 - 20 ms each to create)
    if (!c->stream3) (void)hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking);
    if (!c->stream4) (void)hipStreamCreateWithFlags(&c->stream4, hipStreamNonBlocking);
}

extern "C" {

// Page-locked host memory.  hipHostMalloc zeroes and pins on the calling thread at ~4.6 GB/s (143 ms for the 768 MB ring of
// the end-to-end program, as much again for its genome buffers); anonymous huge-page memory touched by four threads and then
// registered is the same memory to a DMA (56.8 GB/s either way) after 23 ms (tools/debug/register_probe.hip,
// profiles/r03ap2_register_probe.txt).  Blocks below 8 MB, and anything mmap or the registration refuses, take hipHostMalloc.
namespace {
std::mutex g_host_mu;
std::map<void *, size_t> g_host_registered; // blocks of pjb_host_alloc that are mmap + hipHostRegister (value: mapped bytes)
} // namespace
void *pjb_host_alloc(size_t bytes) {
    if (bytes >= ((size_t)8 << 20)) {
        const size_t huge = (size_t)2 << 20, len = (bytes + huge - 1) & ~(huge - 1);
        void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p != MAP_FAILED) {
            (void)madvise(p, len, MADV_HUGEPAGE);
            const size_t nt = std::min<size_t>(4, len >> 24);
            auto touch = [&](size_t t) {
                const size_t per = ((len / std::max<size_t>(nt, 1)) + huge - 1) & ~(huge - 1), a = std::min(len, per * t), b = std::min(len, a + per);
                for (size_t o = a; o < b; o += 4096) ((volatile uint8_t *)p)[o] = 0;
            };
            if (nt <= 1) touch(0);
            else {
                std::vector<std::thread> th;
                for (size_t t = 0; t < nt; t++) th.emplace_back(touch, t);
                for (auto &x : th) x.join();
            }
            if (hipHostRegister(p, len, hipHostRegisterDefault) == hipSuccess) {
                std::lock_guard<std::mutex> lk(g_host_mu);
                g_host_registered[p] = len;
                return p;
            }
            (void)hipGetLastError();
            munmap(p, len);
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void pjb_host_free(void *p) {
    if (!p) return;
    size_t len = 0;
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        auto it = g_host_registered.find(p);
        if (it != g_host_registered.end()) {
            len = it->second;
            g_host_registered.erase(it);
        }
    }
    if (len) {
        (void)hipHostUnregister(p);
        munmap(p, len);
    } else
        (void)hipHostFree(p);
}

int pjb_host_register(void *p, size_t bytes) {
    if (!p || !bytes) return PJB_ERR_ARG;
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();
        return PJB_ERR_HIP;
    }
    return PJB_OK;
}
int pjb_host_unregister(void *p) {
    if (!p) return PJB_ERR_ARG;
    if (hipHostUnregister(p) != hipSuccess) {
        (void)hipGetLastError();
        return PJB_ERR_HIP;
    }
    return PJB_OK;
}

int pjb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pjb_create(pjb_ctx **out, const pjb_config *cfg) {
    if (!out || !cfg) return fail(nullptr, PJB_ERR_ARG, "pjb_create: null argument");
    *out = nullptr;
    if (cfg->abi_version != PJB_ABI_VERSION && cfg->abi_version != 3) // (3: the same entry points; pjb_batch ends at name_hash)
        return fail(nullptr, PJB_ERR_ARG, "pjb_create: ABI version %d, library is %d", cfg->abi_version, PJB_ABI_VERSION);
    if (cfg->orientation < PJB_OR_SE || cfg->orientation > PJB_OR_UNKNOWN)
        return fail(nullptr, PJB_ERR_ARG, "pjb_create: bad orientation %d", cfg->orientation);
    const bool ctrace = getenv("PJB_CREATE_TRACE") != nullptr; // (stderr: what the context's start is made of)
    const auto ct0 = std::chrono::steady_clock::now();
    auto cmark = [&](const char *what) {
        if (ctrace) fprintf(stderr, "[pjb_create] %7.1f ms: %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ct0).count(), what);
    };
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    cmark("hipGetDeviceCount");
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, PJB_ERR_NO_DEVICE,
                    "no HIP device available (%s); the junc hot path has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= n)
        return fail(nullptr, PJB_ERR_ARG, "pjb_create: device %d out of range (have %d)", cfg->device, n);
    e = hipSetDevice(cfg->device);
    if (e != hipSuccess) return fail(nullptr, PJB_ERR_HIP, "hipSetDevice(%d): %s", cfg->device, hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, cfg->device);
    if (e != hipSuccess) return fail(nullptr, PJB_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    cmark("hipSetDevice, hipGetDeviceProperties");
    const int n_cu = prop.multiProcessorCount;
    if (prop.warpSize != 64)
        return fail(nullptr, PJB_ERR_NO_DEVICE, "device %d (%s) is not a wave64 CDNA device", cfg->device, prop.gcnArchName);
    pjb_ctx *c = new (std::nothrow) pjb_ctx();
    if (!c) return fail(nullptr, PJB_ERR_NOMEM, "out of host memory");
    // The kernels' attributes -- the first call loads the code object, 30 - 70 ms -- are set by a thread of their own while
    // this one creates the streams (PJB_CREATE_SERIAL=1: afterwards, on this thread).
    const int dev = cfg->device;
    auto attributes = [dev] {
        (void)hipSetDevice(dev);
        ingest_kernel_attributes();
        // 12-bit digits need more dynamic LDS than the 64 KB a kernel gets without asking
        (void)hipFuncSetAttribute((const void *)rs_scatter<0, u64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rs_scatter_lds_bytes(RS_MAX_BITS));
        (void)hipFuncSetAttribute((const void *)rs_scatter<0, u32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rs_scatter_lds_bytes(RS_MAX_BITS, 4));
    };
    std::thread attr_thread(attributes);
    struct JoinAttr {
        std::thread &t;
        ~JoinAttr() {
            if (t.joinable()) t.join();
        }
    } join_attr{attr_thread};
    c->cfg = *cfg;
    memset(&c->timing, 0, sizeof c->timing);
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(nullptr, PJB_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    if (!(cfg->flags & PJB_FLAG_NO_CHAINS)) aux_streams(c); // (else: with the first chain)
    (void)hipEventCreateWithFlags(&c->ev_front, hipEventDisableTiming);
    c->scan_tiles = &c->b_scan_tiles;
    // The first four control slots get their streams and events now; the others when they are first used (slot_init).  Creating
    // a stream is ~10 ms on an idle device -- and blocked for 1.9 s once when it happened in the middle of an end-to-end run
    // (the runtime creates a hardware queue behind whatever the device is doing): a caller that wants more than four chains in
    // flight on a busy device queues that deep once, early.
    // (60 ms for four slots' streams and events; from four threads at once it is 70 - 100 ms: the runtime serialises them)
    cmark("main streams");
    if (!(cfg->flags & PJB_FLAG_NO_CHAINS))
        for (int k = 0; k < 4 && k < PJB_MAX_QUEUED; k++) (void)slot_init(c, k);
    cmark("chain slots");
    c->inflate_lanes = std::max(1, n_cu) * (160 * 1024 / ingest_lds_bytes()) * 64;
    c->ktime = (cfg->flags & PJB_FLAG_KERNEL_TIMING) != 0;
    c->extra = (cfg->flags & PJB_FLAG_EXTRA) != 0;
    if (const char *s = getenv("PJB_K1_SERIAL")) c->k1_serial = atoi(s) != 0;
    if (const char *s = getenv("PJB_K1S_BLOCKS")) c->k1s_blocks_forced = std::max(0, std::min(atoi(s), (int)K1S_BLOCKS));
    if (const char *s = getenv("PJB_RADIX_BITS")) {
        int v = atoi(s);
        if (v >= 4 && v <= RS_MAX_BITS) c->radix_max_bits = v;
    }
    cmark("options");
    attr_thread.join();
    cmark("kernel attributes");
    *out = c;
    return PJB_OK;
}


void pjb_destroy(pjb_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    (void)hipDeviceSynchronize(); // (contigs may still be queued)
    (void)exhume(c);
#ifdef K1E_PROF
    {
        unsigned long long h[16] = {0};
        (void)hipDeviceSynchronize();
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(pjb::g_k1e_prof), sizeof h) == hipSuccess) {
            unsigned long long tot = 0;
            for (int i = 0; i < 12; i++) tot += h[i];
            static const char *nm[12] = {"block start", "prologue", "shapes (wait ops)", "windows issue", "next records issue", "stage wait", "next ops issue", "compare+emit", "lists", "cand flush", "span of bases", "bases issue"};
            for (int i = 0; i < 12; i++) fprintf(stderr, "[k1e_prof] %-20s %6.2f %%  %llu\n", nm[i], tot ? 100.0 * (double)h[i] / (double)tot : 0.0, h[i]);
        }
    }
#endif
#ifdef K1E_HIST
    { // compare rounds per wavefront and trip of k1_emit: what ran (the longest lane's) against what the lanes needed
        unsigned long long h[4][32] = {{0}};
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(pjb::g_k1e_hist), sizeof h) == hipSuccess) {
            unsigned long long trips = 0, run = 0, need = 0, lanes = 0;
            for (int r = 0; r < 32; r++) trips += h[0][r], run += h[0][r] * (unsigned long long)r, need += h[1][r], lanes += h[2][r];
            fprintf(stderr, "[k1e_hist] 2-bit rounds: %llu wavefront-trips, %.3f rounds run a trip, %.3f rounds a lane needs (max / mean %.3f)\n", trips,
                    trips ? (double)run / (double)trips : 0.0, lanes ? (double)need / (double)lanes : 0.0,
                    need ? ((double)run / (double)trips) / ((double)need / (double)lanes) : 0.0);
            for (int r = 0; r < 32; r++)
                if (h[0][r] || h[3][r])
                    fprintf(stderr, "[k1e_hist]   %2d rounds: %10llu trips (2-bit)  %10llu trips (4-bit)\n", r, h[0][r], h[3][r]);
        }
    }
#endif
    while (!c->open.empty()) close_contig(c, c->open.begin()->first);
    extra_clear(c);
    for (auto &kv : c->filter_keys)
        if (kv.second.first) (void)hipFree(kv.second.first);
    c->filter_keys.clear();
    for (auto &g : c->contigs) free_contig(g);
    for (auto &b : c->genome_pool) release(b);
    c->genome_pool.clear();
    for (auto &sl : c->slab_pool)
        if (sl.p) (void)hipFree(sl.p);
    if (c->rows_pinned) (void)hipHostFree(c->rows_pinned);
    if (c->rows_table) (void)hipFree(c->rows_table);
    bam_stage_clear(c);
    for (int k = 0; k < PJB_MAX_QUEUED; k++) {
        CtlSlot &S = c->sl[k];
        if (S.pub) (void)hipHostFree(S.pub);
        if (S.batches_pinned) (void)hipHostFree(S.batches_pinned);
        for (auto &ev : S.ev)
            if (ev) (void)hipEventDestroy(ev);
        if (S.ev_rows) (void)hipEventDestroy(S.ev_rows);
        if (S.ev_done) (void)hipEventDestroy(S.ev_done);
        Buf *sb[] = {&S.x_q, &S.x_spos, &S.x_send, &S.x_gapoff, &S.x_zlist, &S.x_scnt, &S.x_codes, &S.cstats, &S.err, &S.gencount, &S.batches, &S.rows, &S.tile_cnt, &S.tile_stats, &S.splidx, &S.splpoff, &S.splrec, &S.tile_soff, &S.chunk_tile, &S.scan_parts, &S.members, &S.okey, &S.g,
                     &S.rec, &S.jidbam, &S.jkey, &S.total, &S.bitmap, &S.wrank, &S.pagecnt, &S.pagerank, &S.ends, &S.firstid,
                     &S.key[0], &S.key[1], &S.idx[0], &S.idx[1], &S.hist, &S.hist_scan, &S.hist_part, &S.bintotal, &S.scan_tiles, &S.jid, &S.seg, &S.runfirst,
                     &S.runstart, &S.ent, &S.entsum, &S.frag, &S.fragj, &S.masks, &S.acc, &S.ancl, &S.ancr, &S.genlist};
        for (Buf *b : sb) release(*b);
        hipEvent_t evs[] = {S.ev_k1, S.ev_xk1, S.ev_fork, S.ev_join, S.ev_fork2, S.ev_join2};
        for (hipEvent_t e : evs)
            if (e) (void)hipEventDestroy(e);
        if (S.main) (void)hipStreamDestroy(S.main);
        if (S.side) (void)hipStreamDestroy(S.side);
    }
    if (c->mirror_hdr) (void)hipHostFree(c->mirror_hdr);
    for (int k = 0; k < 2; k++) {
        if (c->stage[k]) (void)hipHostFree(c->stage[k]);
        if (c->stage_ev[k]) (void)hipEventDestroy(c->stage_ev[k]);
    }
    Buf *all[] = {&c->b_cursor, &c->b_scan_tiles, &c->b_hasx, &c->b_xtotal, &c->b_fasta_raw, &c->b_inf_comp, &c->b_inf_out, &c->b_inf_blocks, &c->b_inf_status, &c->b_inf_scratch, &c->b_inf_bitmap,
                  &c->b_bam_seg, &c->b_bam_rec, &c->b_bam_ctl, &c->f_pos, &c->f_cigoff, &c->f_cigar, &c->f_codes, &c->g_rows, &c->g_models, &c->g_refs,
                  &c->g_out, &c->g_bad,
                  &c->x_pos, &c->x_endx, &c->x_q, &c->x_prefq, &c->x_ce, &c->x_bound, &c->x_de, &c->x_dropped, &c->x_zlist, &c->x_cnt,
                  &c->x_tabk, &c->x_tabc, &c->x_rs, &c->x_re, &c->x_rr, &c->x_tileoff, &c->x_xrall, &c->x_tab,
                  &c->b_dfl_in, &c->b_dfl_sym, &c->b_dfl_slots, &c->b_dfl_size, &c->b_dfl_off, &c->b_dfl_packed};
    for (Buf *b : all) release(*b);
    for (auto &ch : c->xarena.chunks) (void)hipFree(ch.p);
    if (c->xrows_pinned) (void)hipHostFree(c->xrows_pinned);
    for (auto &pool : c->pools)
        for (auto &ev : pool.ev) (void)hipEventDestroy(ev);
    if (c->ev_front) (void)hipEventDestroy(c->ev_front);
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    if (c->stream4) (void)hipStreamDestroy(c->stream4);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *pjb_last_error(const pjb_ctx *c) {
    if (!c) return g_create_error.c_str();
    return g_thread_error_ctx == c ? g_thread_error.c_str() : ""; // (this thread's last failure on THIS context; none: the empty string)
}

int pjb_set_refs(pjb_ctx *c, int32_t n_refs, const int32_t *ref_len) {
    if (!c) return PJB_ERR_ARG;
    if (n_refs < 0 || (n_refs > 0 && !ref_len)) return fail(c, PJB_ERR_ARG, "pjb_set_refs: bad arguments");
    if (!c->open.empty()) return fail(c, PJB_ERR_STATE, "pjb_set_refs: contig %d is still open", c->open.begin()->first);
    for (auto &g : c->contigs) free_contig(g);
    c->ref_len.assign(ref_len, ref_len + n_refs);
    c->contigs.assign((size_t)n_refs, Contig());
    return PJB_OK;
}

// fasta_flag: the upload came through k0_fasta, whose verdict -- word 3 of the flags, raised before this call -- is read with the others: a
// record that was not laid out as stated returns 1 and leaves the context as it was
static int upload_common(pjb_ctx *c, int32_t tid, uint8_t *d, int64_t len, bool owned, bool do_upper, size_t d_cap = 0, bool fasta_flag = false) {
    static const bool prof = getenv("PJB_PROFILE_HOST") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_up0 = now();
    double t_alloc = 0;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    // the kernels' three flags (an 'X' among the bases, a character outside the 16-letter alphabet, a character outside ACGT) come back in
    // ONE copy behind the last kernel: every copy to pageable memory makes this thread wait for the stream, and this is the thread that
    // serves every target (three waits a genome were 0.2 s of a human run)
    int rc = ensure(c, c->b_hasx, 4 * sizeof(int));
    if (rc) return rc;
    int *d_flags = (int *)c->b_hasx.p;
    HIP_TRY(c, hipMemsetAsync(d_flags, 0, 3 * sizeof(int), c->stream));
    if (len > 0) {
        const int64_t nthreads = (len + 15) / 16;
        const unsigned nblk = (unsigned)((nthreads + 255) / 256);
        hipLaunchKernelGGL(k0_upper, dim3(nblk), dim3(256), 0, c->stream, d, len, do_upper ? 1 : 0, d_flags + 0);
    }
    // 4-bit codes for the word-parallel compare in k4 (after upper-casing)
    u32 *codes = nullptr;
    size_t codes_cap = 0;
    const int64_t n_words = (len + 7) / 8;
    static const bool no_seq2 = getenv("PJB_NO_SEQ2") && atoi(getenv("PJB_NO_SEQ2")) != 0;
    const size_t c2_at = ((size_t)(n_words + 2) + 3) & ~(size_t)3;                       // (the 2-bit codes start on a 16-byte boundary)
    const size_t c2_words = len > 0 && !no_seq2 ? (size_t)codes2_alloc_words(len) : 0;
    if (len > 0) {
        const double ta = now();
        codes = (u32 *)genome_take(c->genome_pool, (c2_at + c2_words) * 4, codes_cap);
        t_alloc = now() - ta;
        if (!codes) return fail(c, PJB_ERR_NOMEM, "hipMalloc(genome codes, %zu bytes) failed", (c2_at + c2_words) * 4);
        (void)hipMemsetAsync(codes + n_words, 0, 8, c->stream);
        hipLaunchKernelGGL(k0_encode, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, c->stream, (const uint8_t *)d, len,
                           codes, n_words, d_flags + 1);
    }
    // 2-bit codes and their exception bitmap for k1_emit's compares (PJB_NO_SEQ2=1: not built -- A/B runs): behind the 4-bit codes, in
    // the same allocation (a hipMalloc is milliseconds on the thread that serves every target)
    u32 *codes2 = nullptr;
    if (codes && c2_words) {
        codes2 = codes + c2_at;
        const int64_t n2w = codes2_words(len), nxw = gexc_words(len);
        (void)hipMemsetAsync(codes2 + n2w, 0, (size_t)K0_CODES2_PAD * 4, c->stream);
        (void)hipMemsetAsync(codes2 + n2w + K0_CODES2_PAD + nxw, 0, (size_t)K0_GEXC_PAD * 4, c->stream);
        hipLaunchKernelGGL(k0_encode2, dim3((unsigned)(((len + 63) / 64 + 255) / 256)), dim3(256), 0, c->stream, (const uint8_t *)d, len, codes2,
                           codes2 + n2w + K0_CODES2_PAD, d_flags + 2);
    }
    int flags[4] = {0, 0, 1, 0};
    HIP_TRY(c, hipMemcpyAsync(flags, d_flags, (fasta_flag ? 4 : 3) * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    const double t_sync0 = now();
    hipError_t se = hipStreamSynchronize(c->stream);
    if (se != hipSuccess) {
        if (codes) (void)hipFree(codes);
        return fail(c, PJB_ERR_HIP, "upload: %s", hipGetErrorString(se));
    }
    if (prof)
        fprintf(stderr, "[host profile] genome tid %d (%lld bases): codes allocation %.4f, launches %.4f, wait for the stream %.4f s\n", tid, (long long)len, t_alloc,
                t_sync0 - t_up0 - t_alloc, now() - t_sync0);
    if (fasta_flag && flags[3]) { // (the bytes were not a FASTA record of that geometry: nothing of this upload is kept)
        if (codes) (void)hipFree(codes);
        return 1;
    }
    const int hx = flags[0], exotic = flags[1], any_exc = codes2 ? flags[2] : 1;
    if (exotic && codes) {
        (void)hipFree(codes);
        codes = nullptr;
        codes2 = nullptr;
    }
    Contig &g = c->contigs[(size_t)tid];
    free_contig(g, &c->genome_pool);
    g.d = d;
    g.len = len;
    g.owned = owned;
    g.has_x = hx != 0;
    g.present = true;
    g.codes = codes;
    g.codes2 = codes2;
    g.any_exc = any_exc != 0;
    g.d_cap = owned ? d_cap : 0;
    g.codes_cap = codes ? codes_cap : 0;
    return PJB_OK;
}

int pjb_upload_contig(pjb_ctx *c, int32_t tid, const uint8_t *bases, int64_t len) {
    if (!c) return PJB_ERR_ARG;
    if (tid < 0 || (size_t)tid >= c->contigs.size() || len < 0 || (len > 0 && !bases))
        return fail(c, PJB_ERR_ARG, "pjb_upload_contig: bad arguments (tid %d)", tid);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    size_t d_cap = 0;
    uint8_t *d = (uint8_t *)genome_take(c->genome_pool, (size_t)std::max<int64_t>(len, 16), d_cap);
    if (!d) return fail(c, PJB_ERR_NOMEM, "hipMalloc(genome %lld) failed", (long long)len);
    hipError_t e = hipSuccess;
    // through the page-locked staging buffers in 32 MiB pieces (a pageable hipMemcpy is several times slower)
    const size_t PIECE = (size_t)32 << 20;
    for (size_t off = 0; off < (size_t)len; off += PIECE) {
        const size_t n = std::min(PIECE, (size_t)len - off);
        const unsigned si = c->stage_next++ & 1u;
        if (c->stage_busy[si]) {
            (void)hipEventSynchronize(c->stage_ev[si]);
            c->stage_busy[si] = false;
        }
        if (c->stage_cap[si] < n) {
            if (c->stage[si]) (void)hipHostFree(c->stage[si]);
            c->stage[si] = nullptr;
            c->stage_cap[si] = 0;
            if (hipHostMalloc((void **)&c->stage[si], PIECE, hipHostMallocDefault) != hipSuccess) {
                (void)hipFree(d);
                return fail(c, PJB_ERR_NOMEM, "upload: cannot allocate page-locked staging memory");
            }
            c->stage_cap[si] = PIECE;
        }
        if (!c->stage_ev[si]) (void)hipEventCreateWithFlags(&c->stage_ev[si], hipEventDisableTiming);
        parallel_copy(c->stage[si], bases + off, n);
        e = hipMemcpyAsync(d + off, c->stage[si], n, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) {
            (void)hipFree(d);
            return fail(c, PJB_ERR_HIP, "hipMemcpy(genome): %s", hipGetErrorString(e));
        }
        (void)hipEventRecord(c->stage_ev[si], c->stream);
        c->stage_busy[si] = true;
    }
    int rc = upload_common(c, tid, d, len, true, true, d_cap);
    if (rc) (void)hipFree(d);
    return rc;
}

int pjb_upload_contig_fasta(pjb_ctx *c, int32_t tid, const uint8_t *raw, int64_t raw_bytes, int32_t line_blen, int32_t line_len, int64_t len,
                            int *well_formed) {
    if (!c) return PJB_ERR_ARG;
    if (tid < 0 || (size_t)tid >= c->contigs.size() || len < 0 || raw_bytes < 0 || (raw_bytes > 0 && !raw) || line_blen <= 0 ||
        line_len < line_blen || !well_formed)
        return fail(c, PJB_ERR_ARG, "pjb_upload_contig_fasta: bad arguments (tid %d)", tid);
    *well_formed = 0;
    if (line_len - line_blen > 64) return PJB_OK; // (line ends are a byte or two; the kernel's check of them is a loop per line)
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    int rc;
    if ((rc = ensure(c, c->b_fasta_raw, (size_t)std::max<int64_t>(raw_bytes, 16)))) return rc;
    if ((rc = ensure(c, c->b_hasx, 4 * sizeof(int)))) return rc;
    size_t d_cap = 0;
    const double t_a0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    uint8_t *d = (uint8_t *)genome_take(c->genome_pool, (size_t)std::max<int64_t>(len, 16), d_cap);
    if (!d) return fail(c, PJB_ERR_NOMEM, "hipMalloc(genome %lld) failed", (long long)len);
    if (getenv("PJB_PROFILE_HOST"))
        fprintf(stderr, "[host profile] genome tid %d: bases allocation %.4f s\n", tid, std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_a0);
    struct Guard {
        uint8_t *d;
        ~Guard() {
            if (d) (void)hipFree(d);
        }
    } guard{d};
    // page-locked input (pjb_host_alloc): one DMA; otherwise through the staging buffers
    hipPointerAttribute_t at;
    const bool pinned = raw_bytes > 0 && hipPointerGetAttributes(&at, raw) == hipSuccess && at.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError();
    if (pinned) HIP_TRY(c, hipMemcpyAsync(c->b_fasta_raw.p, raw, (size_t)raw_bytes, hipMemcpyHostToDevice, c->stream));
    else if (raw_bytes > 0 && (rc = upload_staged(c, c->b_fasta_raw.p, raw, (size_t)raw_bytes))) return rc;
    // (k0_fasta's verdict is read with the other kernels' flags, behind the last of them: one wait a genome)
    HIP_TRY(c, hipMemsetAsync((int *)c->b_hasx.p + 3, 0, sizeof(int), c->stream));
    if (len > 0) {
        const int64_t nthreads = (len + 15) / 16;
        hipLaunchKernelGGL(k0_fasta, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, c->stream, (const uint8_t *)c->b_fasta_raw.p, raw_bytes, len,
                           line_blen, line_len, d, (int *)c->b_hasx.p + 3);
    }
    rc = upload_common(c, tid, d, len, true, true, d_cap, true);
    if (rc == 1) return PJB_OK; // (*well_formed stays 0: nothing was uploaded)
    if (rc) return rc;
    guard.d = nullptr;
    *well_formed = 1;
    return PJB_OK;
}

int pjb_upload_contig_device(pjb_ctx *c, int32_t tid, const uint8_t *d_bases_upper, int64_t len) {
    if (!c) return PJB_ERR_ARG;
    if (tid < 0 || (size_t)tid >= c->contigs.size() || len < 0 || (len > 0 && !d_bases_upper))
        return fail(c, PJB_ERR_ARG, "pjb_upload_contig_device: bad arguments (tid %d)", tid);
    return upload_common(c, tid, const_cast<uint8_t *>(d_bases_upper), len, false, false);
}

int pjb_release_contig(pjb_ctx *c, int32_t tid) {
    if (!c) return PJB_ERR_ARG;
    if (tid < 0 || (size_t)tid >= c->contigs.size()) return fail(c, PJB_ERR_ARG, "pjb_release_contig: bad tid %d", tid);
    Contig &g = c->contigs[(size_t)tid];
    (void)hipStreamSynchronize(c->stream);
    free_contig(g, &c->genome_pool);
    return PJB_OK;
}

static int add_batch(pjb_ctx *c, int32_t tid, const pjb_batch *b, bool device) {
    if (!c) return PJB_ERR_ARG;
    if (!b || b->n_reads < 0) return fail(c, PJB_ERR_ARG, "submit: bad batch");
    if (tid < 0 || (size_t)tid >= c->ref_len.size()) return fail(c, PJB_ERR_ARG, "submit: bad tid %d", tid);
    c->cur_tid = tid;
    OpenContig &oc = c->open[tid];
    if (b->n_reads == 0) return PJB_OK;
    if (!b->pos || !b->flag || !b->mapq || !b->xs || !b->l_qseq || !b->mtid || !b->mpos || !b->cig_off || !b->cigar ||
        !b->seq_off)
        return fail(c, PJB_ERR_ARG, "submit: null array in batch");
    if (c->extra && !b->name_hash) return fail(c, PJB_ERR_ARG, "submit: a PJB_FLAG_EXTRA context needs pjb_batch.name_hash");
    uint64_t total = 0;
    for (auto &x : oc.batches) total += (uint64_t)x.n;
    if (total + (uint64_t)b->n_reads >= 0xffffff00ull)
        return fail(c, PJB_ERR_ARG, "submit: more than 2^32 alignments on one target are not supported");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    DevBatch d;
    memset(&d, 0, sizeof d);
    d.n = b->n_reads;
    d.base = (uint32_t)total;
    if (device) {
        d.pos = b->pos; d.flag = b->flag; d.mapq = b->mapq; d.xs = b->xs; d.l_qseq = b->l_qseq; d.mtid = b->mtid;
        d.mpos = b->mpos; d.cig_off = b->cig_off; d.cigar = b->cigar; d.seq_off = b->seq_off; d.seq4 = b->seq4;
        d.name_hash = c->extra ? (const u64 *)b->name_hash : nullptr;
        if (c->cfg.abi_version >= 4 && b->seq2 && b->seq_exc) {
            if ((uintptr_t)b->seq2 & 3u) return fail(c, PJB_ERR_ARG, "submit: pjb_batch.seq2 must start on a 4-byte boundary");
            d.seq2 = (const uint32_t *)b->seq2;
            d.seq_exc = b->seq_exc;
        }
    } else {
        const size_t n = (size_t)b->n_reads;
        const size_t n_ops = b->cig_off[n], n_words = b->seq_off[n];
        const void *src[14] = {b->pos, b->flag, b->mapq, b->xs, b->l_qseq, b->mtid, b->mpos, b->cig_off, b->cigar, b->seq_off, b->seq4,
                               c->extra ? b->name_hash : nullptr, nullptr, nullptr};
        const bool two = c->cfg.abi_version >= 4 && b->seq2 && b->seq_exc;
        if (two) src[12] = b->seq2, src[13] = b->seq_exc;
        const size_t bytes[14] = {n * 4, n * 2, n, n, n * 4, n * 4, n * 4, (n + 1) * 4, n_ops * 4, (n + 1) * 4, n_words * 4,
                                  c->extra ? n * 8 : 0, two ? n_words * 2 : 0, two ? ((n + 31) / 32) * 4 : 0};
        // pack into a staging buffer, one DMA to a device slab region with the same packing
        size_t offs[14], total_b = 0;
        for (int k = 0; k < 14; k++) {
            offs[k] = total_b;
            total_b += (std::max<size_t>(bytes[k], 16) + 255) & ~(size_t)255;
        }
        const unsigned si = c->stage_next++ & 1u;
        if (c->stage_busy[si]) {
            HIP_TRY(c, hipEventSynchronize(c->stage_ev[si]));
            c->stage_busy[si] = false;
        }
        if (c->stage_cap[si] < total_b) {
            if (c->stage[si]) (void)hipHostFree(c->stage[si]);
            c->stage[si] = nullptr;
            c->stage_cap[si] = 0;
            const size_t want = total_b + total_b / 8;
            if (hipHostMalloc((void **)&c->stage[si], want, hipHostMallocDefault) != hipSuccess)
                return fail(c, PJB_ERR_NOMEM, "submit: cannot allocate %zu bytes of page-locked staging memory", want);
            c->stage_cap[si] = want;
        }
        if (!c->stage_ev[si]) HIP_TRY(c, hipEventCreateWithFlags(&c->stage_ev[si], hipEventDisableTiming));
        uint8_t *dev = (uint8_t *)slab_alloc(c, oc, total_b);
        if (!dev) return fail(c, PJB_ERR_NOMEM, "submit: out of device memory for a batch of %zu bytes", total_b);
        void *ptrs[14];
        for (int k = 0; k < 14; k++) {
            if (bytes[k] && src[k]) parallel_copy(c->stage[si] + offs[k], src[k], bytes[k]);
            ptrs[k] = dev + offs[k];
        }
        HIP_TRY(c, hipMemcpyAsync(dev, c->stage[si], total_b, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipEventRecord(c->stage_ev[si], c->stream));
        c->stage_busy[si] = true;
        d.pos = (const int32_t *)ptrs[0]; d.flag = (const uint16_t *)ptrs[1]; d.mapq = (const uint8_t *)ptrs[2];
        d.xs = (const uint8_t *)ptrs[3]; d.l_qseq = (const int32_t *)ptrs[4]; d.mtid = (const int32_t *)ptrs[5];
        d.mpos = (const int32_t *)ptrs[6]; d.cig_off = (const uint32_t *)ptrs[7]; d.cigar = (const uint32_t *)ptrs[8];
        d.seq_off = (const uint32_t *)ptrs[9]; d.seq4 = (const uint8_t *)ptrs[10];
        d.name_hash = c->extra ? (const u64 *)ptrs[11] : nullptr;
        if (two) d.seq2 = (const uint32_t *)ptrs[12], d.seq_exc = (const uint32_t *)ptrs[13];
    }
    oc.batches.push_back(d);
    if (!device) oc.on_main_stream = true;
    oc.last_known.push_back(device ? 0 : 1);
    oc.last_pos.push_back(device ? INT32_MIN : b->pos[b->n_reads - 1]);
    return PJB_OK;
}

int pjb_submit_batch(pjb_ctx *c, int32_t tid, const pjb_batch *b) { return add_batch(c, tid, b, false); }
int pjb_submit_batch_device(pjb_ctx *c, int32_t tid, const pjb_batch *b) { return add_batch(c, tid, b, true); }

// a contig without junctions still reports its counters through the row mirror
static void mirror_fold(pjb_ctx *c, const pjb_region_result &R, size_t new_rows) {
    c->mirror_rows += new_rows;
    c->mirror_acc[0] += (int64_t)R.spliced;
    c->mirror_acc[1] += (int64_t)R.unspliced;
    c->mirror_acc[2] += (int64_t)R.sum_len;
    c->mirror_acc[3] = std::min<int64_t>(c->mirror_acc[3], R.min_len);
    c->mirror_acc[4] = std::max<int64_t>(c->mirror_acc[4], R.max_len);
    int64_t *h = c->mirror_hdr;
    h[0] = (int64_t)c->mirror_rows;
    for (int k = 0; k < 5; k++) h[1 + k] = c->mirror_acc[k];
    h[6] = h[7] = 0;
}
static void mirror_reset(pjb_ctx *c) {
    c->mirror_rows = 0;
    c->mirror_acc[0] = c->mirror_acc[1] = c->mirror_acc[2] = 0;
    c->mirror_acc[3] = INT32_MAX;
    c->mirror_acc[4] = 0;
}
static int mirror_header_only(pjb_ctx *c, const pjb_region_result &R) {
    if (!c->mirror) return PJB_OK;
    aux_streams(c);
    mirror_fold(c, R, 0);
    HIP_TRY(c, hipMemcpyAsync(c->mirror, c->mirror_hdr, PJB_MIRROR_HEADER_BYTES, hipMemcpyHostToDevice, c->stream4));
    HIP_TRY(c, hipStreamSynchronize(c->stream4));
    return PJB_OK;
}

// --extra, per contig (calcExtraMetrics' per-target work, src/junction_builder.cc:293-312): the unspliced records'
// per-base depth, the junctions' flanking alignment counts, and the name codes phase 2 needs.  Runs after the
// contig's rows exist (b_rows, sidx, jid are still this contig's).
static void wait_flight(pjb_ctx *c, Flight &f);


// the host's row table holds `need` rows (the first c->rows_n of them are kept)
static int rows_pinned_reserve(pjb_ctx *c, size_t need) {
    if (need <= c->rows_pinned_cap && c->rows_pinned) return PJB_OK;
    int rc = rows_sync(c); // (a DMA into the table that is about to move)
    if (rc) return rc;
    const size_t ncap = std::max<size_t>(need * 3 / 2, 1024);
    pjb_junction_row *np = nullptr;
    const hipError_t e = hipHostMalloc((void **)&np, ncap * sizeof(pjb_junction_row), hipHostMallocPortable);
    if (e != hipSuccess) return fail(c, PJB_ERR_NOMEM, "hipHostMalloc(rows): %s", hipGetErrorString(e));
    if (c->rows_pinned && c->rows_n) memcpy(np, c->rows_pinned, std::min(c->rows_n, c->rows_pinned_cap) * sizeof(pjb_junction_row));
    if (c->rows_pinned) (void)hipHostFree(c->rows_pinned);
    c->rows_pinned = np;
    c->rows_pinned_cap = ncap;
    return PJB_OK;
}

// rows of the contigs collected so far plus the most the queued ones can add
static size_t rows_upper_bound(const pjb_ctx *c) {
    size_t n = c->rows_n;
    for (int k = 0; k < c->n_fl; k++)
        if (c->fl[k].queued) n += c->fl[k].lim.junc_limit;
    return n;
}

constexpr unsigned K6_BLOCKS = 128;
static int queue_contig(pjb_ctx *c, Flight &f) {
    std::vector<DevBatch> &batches = f.batches;
    CtlSlot &S = c->sl[f.slot];
    // the contig's chain runs on its slot's streams, beside the chain of the contig in the other slot; everything the
    // LAUNCH macro and run_scan do follows c->stream / c->scan_tiles until this function returns
    const hipStream_t service = c->stream, st = c->side_stream ? S.main : c->stream; // ("overlap" off: everything on one stream)
    struct ChainScope {
        pjb_ctx *c;
        hipStream_t service;
        ~ChainScope() {
            c->stream = service;
            c->scan_tiles = &c->b_scan_tiles;
        }
    } chain_scope{c, service};
    c->stream = st;
    c->scan_tiles = &S.scan_tiles;
    const u32 n_tiles = f.n_tiles;
    const ContigLimits &lim = f.lim;
    const int n_members = (int)f.tids.size();
    const bool group = n_members > 1;
    // the sequence the chain works on: the target itself, or the group's virtual sequence (every member at its offset)
    const int32_t ref_len = group ? (int32_t)f.vlen : c->ref_len[(size_t)f.tid];
    GroupTab GT;
    memset(&GT, 0, sizeof GT);
    GT.n = n_members;
    bool all_codes = true, any_x = false;
    for (int m = 0; m < n_members; m++) {
        const Contig &g = c->contigs[(size_t)f.tids[(size_t)m]];
        GT.voff[m] = f.voff[(size_t)m];
        GT.len[m] = (int32_t)g.len;
        GT.tid[m] = f.tids[(size_t)m];
        GT.d[m] = g.d;
        GT.codes[m] = g.codes;
        GT.codes2[m] = g.codes2;
        if (g.any_exc) GT.exc_members |= 1u << m;
        all_codes = all_codes && g.codes != nullptr;
        any_x = any_x || g.has_x;
    }
    const KeyFmt kf = lim.kf;
    const u32 PL = lim.pair_limit, JL = lim.junc_limit;
    const u32 SL = lim.sort_limit && lim.sort_limit < JL ? lim.sort_limit : JL; // (ids the sort's digits cover)
    int rc;
    // the place of this contig's rows: known here if nothing is queued ahead of it, else it follows on the device
    bool ahead = false;
    for (int k = 0; k < c->n_fl; k++) ahead |= c->fl[k].queued && &c->fl[k] != &f;
    const int64_t row_base = ahead ? -1 : (int64_t)c->rows_n, mirror_base = ahead ? -1 : (int64_t)c->mirror_rows;
    struct PoolScope { // LAUNCH brackets of this chain belong to the contig's slot
        pjb_ctx *c;
        ~PoolScope() { c->cur_pool = MISC_POOL; }
    } pool_scope{c};
    c->cur_pool = f.slot;
    c->cur_slot = f.slot;
    ev_drop(c, f.slot);
    if ((rc = ensure(c, S.batches, batches.size() * sizeof(DevBatch)))) return rc;
    if (batches.size() + 2 > S.batches_pinned_cap) { // (+ 2 descriptors' worth of room for a group's tile ranges: 33 words)
        if (S.batches_pinned) (void)hipHostFree(S.batches_pinned);
        S.batches_pinned = nullptr;
        S.batches_pinned_cap = 0;
        const size_t cap = std::max<size_t>(batches.size() * 2 + 2, 16);
        HIP_TRY(c, hipHostMalloc((void **)&S.batches_pinned, cap * sizeof(DevBatch), hipHostMallocDefault));
        S.batches_pinned_cap = cap;
    }
    memcpy(S.batches_pinned, batches.data(), batches.size() * sizeof(DevBatch));
    if ((rc = ensure(c, S.cstats, sizeof(ContigStats)))) return rc;
    if (!S.scan_parts.p) {
        if ((rc = ensure(c, S.scan_parts, sizeof(ScanPart) * K1S_BLOCKS))) return rc;
        HIP_TRY(c, hipMemset(S.scan_parts.p, 0, sizeof(ScanPart) * K1S_BLOCKS)); // (once per slot)
        HIP_TRY(c, hipStreamSynchronize(nullptr));
    }
    if ((rc = ensure(c, S.err, 8))) return rc;
    if (!c->b_cursor.p) {
        if ((rc = ensure(c, c->b_cursor, sizeof(RowCursor)))) return rc;
        HIP_TRY(c, hipMemset(c->b_cursor.p, 0, sizeof(RowCursor))); // (blocks_done)
        HIP_TRY(c, hipStreamSynchronize(nullptr));
    }
    if ((rc = ensure(c, S.tile_cnt, (size_t)n_tiles * 4))) return rc;
    if ((rc = ensure(c, S.tile_stats, (size_t)n_tiles * sizeof(TileStats)))) return rc;
    if ((rc = ensure(c, S.tile_soff, ((size_t)n_tiles + 1) * 4))) return rc;
    if ((rc = ensure(c, S.chunk_tile, ((size_t)n_tiles * (K1_TILE / 256) + 4) * 4))) return rc;
    if ((rc = ensure(c, S.splidx, (size_t)n_tiles * K1_TILE * 4))) return rc;
    if ((rc = ensure(c, S.splpoff, (size_t)n_tiles * K1_TILE * 4))) return rc;
    if ((rc = ensure(c, S.splrec, (size_t)n_tiles * K1_TILE * 16))) return rc;
    if (!S.members.p) {
        if ((rc = ensure(c, S.members, GROUP_MAX * sizeof(MemberStats) + GROUP_MAX * 4 + (GROUP_MAX + 1) * 4))) return rc;
        HIP_TRY(c, hipMemset(S.members.p, 0, S.members.cap)); // (member_junc: k6_rows_out (publish_chain) leaves it zeroed for the next chain)
        HIP_TRY(c, hipStreamSynchronize(nullptr));
    }
    MemberStats *d_members = (MemberStats *)S.members.p;
    u32 *d_member_junc = (u32 *)(d_members + GROUP_MAX);
    u32 *d_tile_lo = d_member_junc + GROUP_MAX;
    if ((rc = ensure(c, S.total, 8))) return rc;
    // ---- pair-sized buffers (one sort tile of slack: rs_scatter loads whole tiles unguarded)
    if ((rc = ensure(c, S.okey, ((size_t)PL + RS_TILE) * 8))) return rc; // the pairs' keys as emitted (BAM order): kept, the sort works on copies
    if ((rc = ensure(c, S.key[0], ((size_t)PL + RS_TILE) * 8))) return rc;
    if ((rc = ensure(c, S.key[1], ((size_t)PL + RS_TILE) * 8))) return rc;
    if ((rc = ensure(c, S.idx[0], ((size_t)PL + RS_TILE) * 4))) return rc;
    if ((rc = ensure(c, S.idx[1], ((size_t)PL + RS_TILE) * 4))) return rc;
    if ((rc = ensure(c, S.rec, ((size_t)PL + 1) * sizeof(PairRec)))) return rc;
    if ((rc = ensure(c, S.jid, (size_t)PL * 4 + 16))) return rc;
    if ((rc = ensure(c, S.jidbam, ((size_t)PL + RS_TILE) * 4))) return rc; // (the sort's first pass loads whole tiles)
    if (c->extra && (rc = ensure(c, S.g, (size_t)PL * 4 + 16))) return rc;
    if ((rc = ensure(c, S.seg, ((size_t)PL + 1) * 4))) return rc;
    if ((rc = ensure(c, S.runfirst, ((size_t)PL + 1) * 4))) return rc;
    if ((rc = ensure(c, S.runstart, ((size_t)PL + 1) * 4))) return rc;
    if ((rc = ensure(c, S.ent, (size_t)PL * 8 + 16))) return rc;
    const u32 pair_blocks = std::max<u32>(1, (PL + 255) / 256);
    // entries per sub-list (list_cap_forced: the test hook pjb_set_option("list_cap", n) -- a first attempt with a room that overflows)
    const u32 gen_cap = lim.list_cap ? lim.list_cap : (c->list_cap_forced ? c->list_cap_forced : gen_list_cap(PL));
    const u32 pack_nn = (u64)f.n_reads < (1ull << 28) ? 1u : 0u; // (EmitLists::pack_nn)
    if ((rc = ensure(c, S.genlist, (size_t)gen_cap * GEN_SHARDS * 8 * 3))) return rc; // (three lists: EmitLists)
    if ((rc = ensure(c, S.gencount, GEN_SHARDS * GEN_CNT_STRIDE * 4))) return rc; // a line per sub-list: reads, pairs
    // ---- junction-sized buffers
    const u32 slots_lim = JL + (PL + 63) / 64 + 1;
    if ((rc = ensure(c, S.frag, (size_t)slots_lim * F_WORDS * 4))) return rc;
    if ((rc = ensure(c, S.fragj, (size_t)slots_lim * 4))) return rc;
    if ((rc = ensure(c, S.acc, (size_t)JL * F_WORDS * 4 + 16))) return rc;
    if ((rc = ensure(c, S.ancl, (size_t)JL * 4 + 16))) return rc;
    if ((rc = ensure(c, S.ancr, (size_t)JL * 4 + 16))) return rc;
    if ((rc = ensure(c, S.jkey, (size_t)JL * 8 + 16))) return rc;
    if ((rc = ensure(c, S.rows, (size_t)JL * sizeof(pjb_junction_row) + 16))) return rc;
    // The row table lives twice, both grow-only: in HBM, where the rows stream appends each contig's rows (k6_rows_out),
    // and in page-locked host memory, filled by a DMA per contig once its row count is known (pjb_finish_contig_end).
    // (The kernel used to write the host table itself: 2 MB of PCIe stores per contig that slowed whatever ran beside
    // them -- the next contig's k1_count by 40 %.)
    // The HBM table has room for the most the queued chains can bring (JL is a bound, 5 - 20 x what a chain brings); the host table
    // grows when a chain is collected, to what it brought (rows_pinned_reserve) -- page-locking costs ~0.1 s per GB, and sized by the
    // bound the host table of the first 1-Gb chain was 400 MB: 70 ms on the thread that queues the chains, with every copy of the
    // program's run waiting behind it, for 17 MB of rows.
    size_t old = rows_upper_bound(c);
    if (old + JL > c->rows_cap) {
        for (int k = 0; k < c->n_fl; k++) // (the table moves: nothing may be writing to it)
            if (c->fl[k].queued && &c->fl[k] != &f) wait_flight(c, c->fl[k]);
        if ((rc = rows_sync(c))) return rc;
        old = std::min(old, c->rows_cap);
        const size_t ncap = std::max<size_t>((old + JL) * 3 / 2, 1024);
        pjb_junction_row *nd = nullptr;
        hipError_t e = hipMalloc((void **)&nd, ncap * sizeof(pjb_junction_row));
        if (e != hipSuccess) return fail(c, PJB_ERR_NOMEM, "hipMalloc(row table): %s", hipGetErrorString(e));
        if (old) {
            // (a device-to-device hipMemcpy may return before it has run, and the chains' streams do not wait for the null stream: without
            // the wait the rows a chain appends below `old` -- `old` is a bound -- could be overwritten by the tail of this copy.  The
            // hipFree that used to follow waited for the whole device; this waits for the copy.)
            HIP_TRY(c, hipMemcpyAsync(nd, c->rows_table, old * sizeof(pjb_junction_row), hipMemcpyDeviceToDevice, c->stream4));
            HIP_TRY(c, hipStreamSynchronize(c->stream4));
        }
        if (c->rows_table) bury(c, c->rows_table); // (nothing reads it any more; hipFree would wait for the device)
        c->rows_table = nd;
        c->rows_cap = ncap;
    }
    if (!S.pub) {
        HIP_TRY(c, hipHostMalloc((void **)&S.pub, PUB_BYTES, hipHostMallocMapped | hipHostMallocPortable));
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, S.pub, 0) != hipSuccess || !dp) return fail(c, PJB_ERR_HIP, "hipHostGetDevicePointer(control block) failed");
        S.pub_dev = (uint8_t *)dp;
    }

    // (records still being produced on the service stream -- host copies, BAM ingest -- come first)
    const hipStream_t front = st;
    {
        bool on_service = false;
        for (int32_t t : f.tids) {
            auto oit = c->open.find(t);
            on_service = on_service || (oit != c->open.end() && oit->second.on_main_stream);
        }
        if (st != service && on_service) {
            HIP_TRY(c, hipEventRecord(c->ev_front, service));
            HIP_TRY(c, hipStreamWaitEvent(st, c->ev_front, 0));
        }
    }
    HIP_TRY(c, hipMemcpyAsync(S.batches.p, S.batches_pinned, batches.size() * sizeof(DevBatch), hipMemcpyHostToDevice, front));
    if (group) { // (the tile ranges of the members; lives behind the batch descriptors in the page-locked staging block)
        u32 *h_lo = (u32 *)(S.batches_pinned + batches.size());
        for (int m = 0; m <= n_members; m++) h_lo[m] = f.tile_lo[(size_t)m];
        HIP_TRY(c, hipMemcpyAsync(d_tile_lo, h_lo, (size_t)(n_members + 1) * 4, hipMemcpyHostToDevice, front));
    }
    if (!S.at_rest) {
        HIP_TRY(c, hipMemsetAsync(S.err.p, 0xff, 8, front));
        HIP_TRY(c, hipMemsetAsync(S.gencount.p, 0, GEN_SHARDS * GEN_CNT_STRIDE * 4, front));
        HIP_TRY(c, hipMemsetAsync(d_member_junc, 0, GROUP_MAX * 4, front));
    }
    S.at_rest = false; // until k6_rows_out (publish_chain) is queued
    u64 *d_err = (u64 *)S.err.p;
    ContigStats *d_cs = (ContigStats *)S.cstats.p;
    const u32 *d_P = &d_cs->P, *d_J = &d_cs->J, *d_slots = &d_cs->n_slots;
    // stage boundaries are timed only under full instrumentation: an event between two kernels costs a ~6 us bubble
    const bool stage_events = c->ktime && c->ktime_only.empty();
#define STAGE_EVENT(k)                                            \
    do {                                                          \
        if (stage_events) HIP_TRY(c, hipEventRecord(S.ev[k], st));  \
    } while (0)
    HIP_TRY(c, hipEventRecord(S.ev[0], front));
    // K2d's bitmap and end slots (all-clear at rest): k1_emit sets the bits
    const size_t n_words = ((size_t)std::max(ref_len, 1) + 63) / 64;
    if (lim.dense) {
        const void *was[2] = {S.bitmap.p, S.ends.p};
        if ((rc = ensure(c, S.bitmap, n_words * 8 + 16))) return rc;
        if ((rc = ensure(c, S.wrank, n_words * 4 + 16))) return rc;
        const void *was_pc = S.pagecnt.p;
        if ((rc = ensure(c, S.pagecnt, ((n_words >> KD_PAGE_SHIFT) + 1) * 4 + 16))) return rc;
        if ((rc = ensure(c, S.pagerank, ((n_words >> KD_PAGE_SHIFT) + 1) * 4 + 16))) return rc;
        if ((rc = ensure(c, S.ends, (size_t)JL * DENSE_ENDS * 4 + 32))) return rc;
        if ((rc = ensure(c, S.firstid, (size_t)JL * 4 + 16))) return rc;
        if (!S.dense_at_rest || was[0] != S.bitmap.p || was[1] != S.ends.p || was_pc != S.pagecnt.p) { // (first use, new memory, or a chain that broke off)
            HIP_TRY(c, hipMemsetAsync(S.bitmap.p, 0, S.bitmap.cap, front));
            HIP_TRY(c, hipMemsetAsync(S.pagecnt.p, 0, S.pagecnt.cap, front));
            HIP_TRY(c, hipMemsetAsync(S.ends.p, 0xff, S.ends.cap, front));
        }
        S.dense_at_rest = false; // until kd_reset is queued
    }
    Pairs pr;
    pr.key = (u64 *)S.okey.p;
    pr.rec = (PairRec *)S.rec.p;
    pr.g = c->extra ? (u32 *)S.g.p : (u32 *)nullptr;
    f.pr = pr;
    u32 *d_gen_cnt = (u32 *)S.gencount.p;
    const bool fast_codes = all_codes && !any_x; // (else: no read is "simple", every pair takes k4b_generic's byte-wise walks)
    {
        // ---- K1a: count (a group's members: a tile whose alignments leave the member's own sequence is flagged); with
        // PJB_FLAG_EXTRA the first time also what the records span (a chain that is queued again leaves that alone: the
        // service stream may be reading it)
        const bool xk1 = c->extra && !c->extra_dense_only && !group && !f.x_k1 && !f.x_pre;
        XOut xo = {nullptr, nullptr, nullptr, nullptr, 0, nullptr};
        if (xk1) {
            const size_t N = (size_t)f.n_reads;
            if ((rc = ensure(c, S.x_q, N + 16)) || (rc = ensure(c, S.x_spos, N * 4 + 16)) || (rc = ensure(c, S.x_send, N * 4 + 16)) ||
                (rc = ensure(c, S.x_zlist, (size_t)X_ZCAP * 4)) || (rc = ensure(c, S.x_scnt, sizeof(SparseCounters) + sizeof(ExtraCounters))))
                return rc;
            HIP_TRY(c, hipMemsetAsync((uint8_t *)S.x_q.p + N, 0, 1, c->stream));
            HIP_TRY(c, hipMemsetAsync(S.x_scnt.p, 0, sizeof(SparseCounters) + sizeof(ExtraCounters), c->stream));
            xo = XOut{(int32_t *)S.x_spos.p, (int32_t *)S.x_send.p, (uint8_t *)S.x_q.p, (u32 *)S.x_zlist.p, X_ZCAP, (SparseCounters *)S.x_scnt.p};
        }
        // K1 of one chain fills the chip: the chains' K1 stages follow each other (this chain's waits for the last queued chain's),
        // and what comes behind a chain's K1 -- many small kernels -- runs beside the NEXT chain's K1 instead of beside its own twin
        if (c->k1_serial && c->last_k1_ev) HIP_TRY(c, hipStreamWaitEvent(st, c->last_k1_ev, 0));
        // (one launch over the chain's tiles: a block finds its batch from the tile index)
        if (xk1)
            LAUNCH(c, "k1_count", k1_count<true>, dim3(n_tiles), dim3(K1C_T), (const DevBatch *)S.batches.p, (int)batches.size(), (u32 *)S.tile_cnt.p, (TileStats *)S.tile_stats.p,
                   (u32 *)S.splidx.p, (u32 *)S.splpoff.p, (uint4 *)S.splrec.p, d_err, GT, 0, xo);
        else
            LAUNCH(c, "k1_count", k1_count<false>, dim3(n_tiles), dim3(K1C_T), (const DevBatch *)S.batches.p, (int)batches.size(), (u32 *)S.tile_cnt.p, (TileStats *)S.tile_stats.p,
                   (u32 *)S.splidx.p, (u32 *)S.splpoff.p, (uint4 *)S.splrec.p, d_err, GT, group ? 1 : 0, xo);
        if (xk1) {
            HIP_TRY(c, hipEventRecord(S.ev_xk1, c->stream));
            f.x_k1 = true;
        }
        LAUNCH(c, "k1_scan_tiles", k1_scan_tiles, dim3(c->k1s_blocks_forced ? (u32)c->k1s_blocks_forced : k1s_blocks(n_tiles)), dim3(K1S_THREADS), (u32 *)S.tile_cnt.p, (const TileStats *)S.tile_stats.p,
               n_tiles, d_cs, PL, kf, group ? INT32_MAX - 1 : ref_len, (const u64 *)nullptr, (u32 *)S.tile_soff.p, (u32 *)S.chunk_tile.p,
               (ScanPart *)S.scan_parts.p, ++S.scan_epoch);
        // ---- K1b: emit (coordinates in the group's virtual sequence): keys, the pairs' records -- complete for reads of the
        // simple shape --, K2d's candidate keys (free until the first scatter; they are used up before it), the list of
        // reads for k4b_generic
        EmitLists el;
        el.cand = lim.dense ? (u64 *)S.key[1].p : (u64 *)nullptr;
        el.bitmap = lim.dense ? (u64 *)S.bitmap.p : (u64 *)nullptr;
        el.page_cnt = lim.dense ? (u32 *)S.pagecnt.p : (u32 *)nullptr;
        el.cand_anc = (u64 *)S.ent.p; // (a pair-sized scratch buffer nothing else uses at this point)
        el.gen_list = (u64 *)S.genlist.p;
        el.gen_cnt = d_gen_cnt;
        el.gen_cap = gen_cap;
        el.pack_nn = pack_nn;
        // (one launch per chain -- per K1E_MAXB batches --: the blocks stride over the batches' trips of 256 spliced reads; a tile holds
        // ~300 spliced reads = 1.2 trips: a grid of half the tiles keeps two or three trips per block)
        for (size_t b0 = 0; b0 < batches.size(); b0 += K1E_MAXB) {
            const size_t nb = std::min<size_t>(K1E_MAXB, batches.size() - b0);
            u64 reads = 0;
            for (size_t bi = b0; bi < b0 + nb; bi++) reads += (u64)batches[bi].n;
            const u32 nt = (u32)std::min<u64>((reads + K1_TILE - 1) / K1_TILE + nb, 0x7fffffffu);
            // (tiles of reads per block: 2 / 4 / 8 / 16 = 9.17 / 8.96 / 8.94 / 9.05 ms a step, profiles/r05q_*; with round 6's 2-bit compare
            // 4 / 8 / 12 / 16 / 32 / 64 = 7.25 - 7.37 / 7.15 - 7.17 / 7.32 / 7.32 / 7.37 / 8.02: fewer, longer blocks make the kernel itself
            // faster still -- 970 us a chain in the step at 16 against 1 085 at 8 -- but the step no shorter: profiles/r06_k1_experiments.txt)
            static const u32 tiles_per_block = getenv("PJB_K1E_TILES") ? (u32)std::max(1, atoi(getenv("PJB_K1E_TILES"))) : 8u;
            const u32 grid = std::max<u32>(1, std::min<u32>(nt, std::max<u32>(1024, nt / tiles_per_block)));
            LAUNCH(c, "k1_emit", k1_emit, dim3(grid), dim3(256), (const DevBatch *)S.batches.p + b0, (int)nb, n_tiles, (const u32 *)S.tile_cnt.p,
                   (const u32 *)S.tile_soff.p, (const u32 *)S.chunk_tile.p, (const u32 *)S.splidx.p, (const u32 *)S.splpoff.p, (const uint4 *)S.splrec.p, pr, el, kf,
                   GT, fast_codes ? 1 : 0, (int)c->cfg.orientation, d_err, d_cs);
        }
        // (the next chain's K1 may start here: k1_generic -- a few reads walked by a few wavefronts, waiting for their loads -- runs beside
        // its k1_count, which is a stream)
        if (c->k1_serial && !getenv("PJB_K1_SERIAL_LATE")) {
            HIP_TRY(c, hipEventRecord(S.ev_k1, st));
            c->last_k1_ev = S.ev_k1;
        }
        // (per-member counters of a group: off the K1 stage -- beside the next chain's k1_count -- and before k5_finalize counts the
        // members' junctions into them)
        if (group)
            LAUNCH(c, "kg_member_stats", kg_member_stats, dim3((unsigned)n_members), dim3(256), (const u32 *)S.tile_cnt.p, (const TileStats *)S.tile_stats.p,
                   (const u32 *)d_tile_lo, n_members, d_members, n_tiles, (const ContigStats *)d_cs);
        // the reads k1_emit left: one launch over the chain's third list (the blocks stride over it)
        LAUNCH(c, "k1_generic", k1_generic, dim3(std::min<u32>(std::max<u32>(1, (u32)(((u64)gen_cap * GEN_SHARDS + K1E_T - 1) / K1E_T)), 1536u /* six blocks a CU; 512 .. 3072 measured: no difference */)), dim3(K1E_T),
               (const DevBatch *)S.batches.p, (int)batches.size(), (const u32 *)S.splidx.p, (const uint4 *)S.splrec.p, pr, el, kf, GT, fast_codes ? 1 : 0,
               (int)c->cfg.orientation, d_err, d_cs);
    }
    if (c->k1_serial && getenv("PJB_K1_SERIAL_LATE")) { // (experiment: behind k1_generic)
        HIP_TRY(c, hipEventRecord(S.ev_k1, st));
        c->last_k1_ev = S.ev_k1;
    }
    STAGE_EVENT(1);
    // k4b_generic: the pairs that need the generic walks, in BAM order, as soon as junction ids and anchors exist -- beside
    // the sort, on the side stream
    const u32 gen_grid = std::min<u32>(2 * (u32)(((u64)gen_cap * GEN_SHARDS + 255) / 256), 2560u); // (the blocks stride over the lists' entries)
    auto launch_k4b = [&]() -> int {
        LAUNCH(c, "k4b_generic", k4b_generic, dim3(gen_grid), dim3(256), (const u64 *)S.genlist.p, (const u32 *)d_gen_cnt, gen_cap, (const u64 *)pr.key,
               pr.rec, (const u32 *)S.jidbam.p, kf, (const DevBatch *)S.batches.p, (int)batches.size(), (const int32_t *)S.ancl.p, (const int32_t *)S.ancr.p,
               GT, any_x ? 1 : 0, any_x ? 0 : 1, d_err, (const ContigStats *)d_cs, pack_nn);
        return PJB_OK;
    };
    auto fork_k4b = [&]() -> int { // (the main stream has just produced jid_bam and the anchors)
        if (!c->side_stream) return launch_k4b();
        HIP_TRY(c, hipEventRecord(S.ev_fork, st));
        HIP_TRY(c, hipStreamWaitEvent(S.side, S.ev_fork, 0));
        c->stream = S.side;
        f.forked = true;
        const int rc2 = launch_k4b();
        c->stream = st;
        if (rc2) return rc2;
        HIP_TRY(c, hipEventRecord(S.ev_join, S.side));
        return PJB_OK;
    };

    // ---- K2d: ordered dense junction ids (the sort then works on 15-19 bits instead of 46-48)
    int sort_bits = kf.total_bits;
    // the sort's passes: as few as the widest digit allows, bits spread evenly (planned before kd_assign, which counts the first digit)
    const u32 rs_tiles = std::max<u32>(1, (PL + RS_TILE - 1) / RS_TILE);
    int n_pass = 1;
    std::vector<int> pass_bits;
    bool first_hist_done = false;
    auto plan_passes = [&]() -> int {
        n_pass = (sort_bits + c->radix_max_bits - 1) / c->radix_max_bits;
        if (n_pass < 1) n_pass = 1;
        pass_bits.assign((size_t)n_pass, sort_bits / n_pass);
        for (int p = 0; p < sort_bits % n_pass; p++) pass_bits[(size_t)p]++;
        const int dbits = pass_bits[0];
        int rc2;
        if ((rc2 = ensure(c, S.hist, (size_t)rs_tiles * (1u << dbits) * 4))) return rc2;
        if ((rc2 = ensure(c, S.hist_scan, (size_t)rs_tiles * (1u << dbits) * 4))) return rc2;
        if ((rc2 = ensure(c, S.bintotal, (size_t)4 << dbits))) return rc2;
        if ((rc2 = ensure(c, S.hist_part, (size_t)((rs_tiles + RSP_TILES - 1) / RSP_TILES) * (1u << dbits) * 4))) return rc2;
        return PJB_OK;
    };
    if (lim.dense) {
        const u32 cand_blocks = std::min<u32>(pair_blocks, 1024u); // (a few candidates per junction: these kernels stride)
        const u64 *okey = (const u64 *)pr.key;
        u64 *cand = (u64 *)S.key[1].p; // (k1_emit left the candidate keys here, and their starts' bits in the bitmap)
        // (ranks of the bitmap's words: a prefix sum over the pages' counts of starts, then the words of the pages that hold one)
        const u32 n_pages = (u32)((n_words + ((size_t)1 << KD_PAGE_SHIFT) - 1) >> KD_PAGE_SHIFT);
        if ((rc = run_scan(c, "kd_rank", ArrU32Fn{(const u32 *)S.pagecnt.p}, ExclusiveU32Sink{(u32 *)S.pagerank.p}, (u64)n_pages, (u64 *)S.total.p)))
            return rc;
        LAUNCH(c, "kd_rank_pages", kd_rank_pages, dim3((n_pages + 3) / 4), dim3(256), (const u64 *)S.bitmap.p, (const u32 *)S.pagecnt.p, (const u32 *)S.pagerank.p,
               (u32 *)S.wrank.p, n_pages, (u32)n_words);
        u32 *cand_rank = (u32 *)S.idx[1].p; // (free until the first scatter as well)
        LAUNCH(c, "kd_ends", kd_ends, dim3(cand_blocks), dim3(256), (const u64 *)cand, kf, (const u64 *)S.bitmap.p, (const u32 *)S.wrank.p, JL,
               (u32 *)S.ends.p, cand_rank, d_cs);
        if ((rc = run_scan(c, "kd_first", EndsCountFn{(const u32 *)S.ends.p}, FirstIdSink{(u32 *)S.firstid.p, (int32_t *)S.ancl.p, (int32_t *)S.ancr.p},
                           (u64)JL, (u64 *)S.total.p)))
            return rc;
        LAUNCH(c, "kd_table", kd_table, dim3(cand_blocks), dim3(256), (const u64 *)cand, (const u64 *)S.ent.p, (const u32 *)cand_rank, kf, JL,
               (const u32 *)S.ends.p, (const u32 *)S.firstid.p, (const u64 *)S.total.p, (u64 *)S.jkey.p, (int32_t *)S.ancl.p, (int32_t *)S.ancr.p, d_cs,
               (const u32 *)d_gen_cnt, gen_cap, SL);
        // (a block per tile of the sort: it leaves the tile's counts of the ids' first digit -- the first pass below has no rs_hist)
        sort_bits = std::max(1, bits_of((uint64_t)SL));
        if ((rc = plan_passes())) return rc;
        LAUNCH(c, "kd_assign", kd_assign, dim3(rs_tiles), dim3(256), okey, d_P, kf, (const u64 *)S.bitmap.p, (const u32 *)S.wrank.p,
               (const u32 *)S.ends.p, (const u32 *)S.firstid.p, JL, (const u64 *)S.total.p, (u32 *)S.jidbam.p, (u32 *)S.acc.p, (const u64 *)S.jkey.p,
               (const int32_t *)S.ancl.p, (const int32_t *)S.ancr.p, d_err, d_cs, pass_bits[0], (u32 *)S.hist.p);
        first_hist_done = true;
        if ((rc = fork_k4b())) return rc;
        LAUNCH(c, "kd_reset", kd_reset, dim3(cand_blocks), dim3(256), (const u64 *)cand, (const u32 *)cand_rank, kf, JL, (const ContigStats *)d_cs,
               (u64 *)S.bitmap.p, (u32 *)S.ends.p, (u32 *)S.pagecnt.p);
        S.dense_at_rest = true;
    } else if ((rc = plan_passes()))
        return rc;
    // ---- K2: radix sort (key, pair index)
    int cur = 0, shift = 0;
    // digit passes.  Dense ids are 32-bit keys: pass 0 reads them where kd_assign left them (BAM order; k4b_generic reads that array
    // beside the sort, so no pass writes to it), the ping-pong buffers are the first halves of the 64-bit key buffers.
    auto sort_pass = [&](auto key_tag, const void *kin_v, void *kout_v, const u32 *vin, u32 *vout, int bits) -> int {
        using K = decltype(key_tag);
        const K *kin = (const K *)kin_v;
        K *kout = (K *)kout_v;
        if (!(shift == 0 && first_hist_done)) // (dense ids: kd_assign counted the first digit)
            LAUNCH(c, "rs_hist", rs_hist<K>, dim3(rs_tiles), dim3(256), kin, d_P, shift, bits, (u32 *)S.hist.p, rs_tiles);
        {
            const u32 nb = 1u << bits, n_panels = (rs_tiles + RSP_TILES - 1) / RSP_TILES;
            LAUNCH(c, "rs_panel_sums", rs_panel_sums, dim3(n_panels, (nb + 255) / 256), dim3(256), (const u32 *)S.hist.p, rs_tiles, nb, (u32 *)S.hist_part.p);
            LAUNCH(c, "rs_panel_scan", rs_panel_scan, dim3(n_panels, (nb + 255) / 256), dim3(256), (const u32 *)S.hist.p, (const u32 *)S.hist_part.p, rs_tiles, nb,
                   (u32 *)S.hist_scan.p, (u32 *)S.bintotal.p);
        }
#define RS_SCATTER(B)                                                                                                                          \
    LAUNCH_LDS(c, "rs_scatter", (rs_scatter<B, K>), dim3(rs_tiles), dim3(256), rs_scatter_lds_bytes(bits, sizeof(K)), kin, vin, kout, vout, d_P, \
               shift, bits, (const u32 *)S.hist_scan.p, (const u32 *)S.bintotal.p, rs_tiles)
        switch (bits) { // the usual digit widths get an unrolled match loop
        case 9: RS_SCATTER(9); break;
        case 10: RS_SCATTER(10); break;
        case 11: RS_SCATTER(11); break;
        default: RS_SCATTER(0); break;
        }
#undef RS_SCATTER
        return PJB_OK;
    };
    for (int p = 0; p < n_pass; p++) {
        const int bits = pass_bits[(size_t)p];
        if (bits <= 0) break;
        const u32 *vin = p == 0 ? nullptr : (const u32 *)S.idx[cur].p;
        u32 *vout = (u32 *)S.idx[cur ^ 1].p;
        if (lim.dense) rc = sort_pass((u32)0, p == 0 ? S.jidbam.p : S.key[cur].p, S.key[cur ^ 1].p, vin, vout, bits);
        else rc = sort_pass((u64)0, p == 0 ? S.okey.p : S.key[cur].p, S.key[cur ^ 1].p, vin, vout, bits);
        if (rc) return rc;
        cur ^= 1;
        shift += bits;
    }
    f.n_pass = n_pass;
    const u32 *sidx = (const u32 *)S.idx[cur].p;
    f.sidx = sidx;
    const u32 *jid_sorted = lim.dense ? (const u32 *)S.key[cur].p : (const u32 *)S.jid.p; // junction id of every sorted pair
    f.jid_sorted = jid_sorted;
    STAGE_EVENT(2);
    if ((rc = ensure(c, S.entsum, (size_t)JL * 8 + 16))) return rc;
    const u32 n_slices_lim = (PL + 63) / 64 + 1;
    const hipStream_t tl = st;
    auto entropy_kernels = [&]() -> int {
        LAUNCH(c, "k5_entropy_sum", k5_entropy_sum, dim3(std::max<u32>(1, (JL + 15) / 16)), dim3(256), (const u32 *)S.seg.p, (const u32 *)S.runfirst.p,
               (const u32 *)S.runstart.p, d_J, (double *)S.entsum.p);
        return PJB_OK;
    };
    bool entropy_forked = false;
    auto fork_entropy = [&]() -> int { // the entropy kernels beside what follows on the main stream (small and latency-bound, both)
        if (!c->side_stream) return entropy_kernels();
        HIP_TRY(c, hipEventRecord(S.ev_fork2, st));
        HIP_TRY(c, hipStreamWaitEvent(S.side, S.ev_fork2, 0));
        c->stream = S.side;
        entropy_forked = true;
        const int rc2 = entropy_kernels();
        c->stream = st;
        if (rc2) return rc2;
        HIP_TRY(c, hipEventRecord(S.ev_join2, S.side));
        return PJB_OK;
    };
    if (lim.dense) {
        // ---- the usual chain: the sorted ids are the junction ids; K4 gathers the pairs -- one 32-byte record each -- folds them
        // to fragments and leaves the junction / run boundaries as bit masks; two small kernels turn the masks into
        // seg_off / run_first / run_start (no second pass over the pairs), then the entropy beside the fragment reduce
        if ((rc = ensure(c, S.masks, (size_t)n_slices_lim * 8 * 2 + (size_t)n_slices_lim * 4))) return rc;
        u64 *head_mask = (u64 *)S.masks.p, *run_mask = head_mask + n_slices_lim;
        u32 *run_base = (u32 *)(run_mask + n_slices_lim);
        STAGE_EVENT(3);
        STAGE_EVENT(4);
        if (f.forked) HIP_TRY(c, hipStreamWaitEvent(tl, S.ev_join, 0)); // k4b_generic's results (side stream) are needed from here on
#ifdef PJB_DEBUG_LAUNCH
        fprintf(stderr, "[k4_pairs] tid %d PL %u JL %u slots_lim %u n_slices_lim %u frag cap %zu fragj cap %zu masks cap %zu rec cap %zu idx cap %zu key cap %zu jkey cap %zu cur %d n_pass %d\n",
                f.tid, PL, JL, slots_lim, n_slices_lim, S.frag.cap, S.fragj.cap, S.masks.cap, S.rec.cap, S.idx[cur].cap, S.key[cur].cap, S.jkey.cap, cur, n_pass);
        {
            ContigStats hcs;
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(&hcs, d_cs, sizeof hcs, hipMemcpyDeviceToHost);
            u64 htot = 0;
            (void)hipMemcpy(&htot, S.total.p, 8, hipMemcpyDeviceToHost);
            fprintf(stderr, "[k4_pairs] cs: P %u n_pairs %llu J %u n_junc %u n_slots %u n_slices %u n_cand %u overflow %u R %u total %llu slot %d attempt %d\n", hcs.P,
                    (unsigned long long)hcs.n_pairs, hcs.J, hcs.n_junc, hcs.n_slots, hcs.n_slices, hcs.n_cand, hcs.overflow, hcs.R, (unsigned long long)htot, f.slot, f.attempt);
            // is the sort's output a permutation with ascending ids?  is its input what kd_assign wrote?
            std::vector<u32> hs(hcs.P), hj(hcs.P), hb(hcs.P);
            (void)hipMemcpy(hs.data(), sidx, (size_t)hcs.P * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hj.data(), jid_sorted, (size_t)hcs.P * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hb.data(), S.jidbam.p, (size_t)hcs.P * 4, hipMemcpyDeviceToHost);
            std::vector<char> seen(hcs.P, 0);
            size_t bad_idx = 0, dup = 0, bad_order = 0, bad_in = 0, mism = 0;
            long first_bad = -1;
            for (size_t i = 0; i < hcs.P; i++) {
                if (hb[i] >= hcs.J) bad_in++;
                if (hs[i] >= hcs.P) {
                    bad_idx++;
                    if (first_bad < 0) first_bad = (long)i;
                    continue;
                }
                if (seen[hs[i]]) dup++;
                seen[hs[i]] = 1;
                if (hj[i] != hb[hs[i]]) mism++;
                if (i && hj[i] < hj[i - 1]) bad_order++;
            }
            fprintf(stderr, "[k4_pairs] sort check: %zu indices out of range (first at %ld), %zu duplicates, %zu order breaks, %zu key/index mismatches, %zu input ids >= J\n", bad_idx,
                    first_bad, dup, bad_order, mism, bad_in);
        }
#endif
        LAUNCH(c, "k4_pairs", k4_pairs, dim3(pair_blocks), dim3(256), sidx, jid_sorted, (const PairRec *)pr.rec, (const u64 *)S.jkey.p, kf, d_P,
               (u32 *)S.frag.p, (int32_t *)S.fragj.p, head_mask, run_mask, (const ContigStats *)d_cs, d_err);
        if ((rc = run_scan(c, "k2_runs", Popc64Fn{(const u64 *)run_mask}, ExclusiveU32Sink{run_base}, (u64)n_slices_lim, (u64 *)S.total.p, &d_cs->n_slices)))
            return rc;
        LAUNCH(c, "k2_expand", k2_expand, dim3((pair_blocks + K2E_PER - 1) / K2E_PER), dim3(256), jid_sorted, (const u64 *)head_mask, (const u64 *)run_mask, (const u32 *)run_base,
               (const u64 *)S.total.p, (u32 *)S.seg.p, (u32 *)S.runfirst.p, (u32 *)S.runstart.p, d_cs);
        STAGE_EVENT(5);
        if ((rc = fork_entropy())) return rc;
    } else {
        // ---- a chain that sorted the full keys: junction ids and position runs from a scan over the sorted pairs (it fetches
        // every pair's read position), the junctions' keys, anchors, ids in BAM order -- then the generic pairs and K4
        const u64 *skey = (const u64 *)S.key[cur].p;
        HeadFn hf{skey, sidx, (const PairRec *)pr.rec};
        HeadSink hs{(u32 *)S.jid.p, (u32 *)S.seg.p, (u32 *)S.runfirst.p, (u32 *)S.runstart.p, skey, (u64 *)S.jkey.p, JL};
        if ((rc = run_scan(c, "k2_heads", hf, hs, (u64)PL, (u64 *)S.total.p, d_P))) return rc;
        LAUNCH(c, "k2_close", k2_close, dim3(1), dim3(1), (u64 *)S.total.p, (u32 *)S.seg.p, (u32 *)S.runfirst.p,
               (u32 *)S.runstart.p, d_cs, JL, (const u32 *)d_gen_cnt, gen_cap);
        STAGE_EVENT(3);
        if ((rc = fork_entropy())) return rc;
        LAUNCH(c, "kf_init", kf_init, dim3(std::max<u32>(1, std::min<u32>((JL * F_WORDS + 255) / 256, 4096))), dim3(256), (u32 *)S.acc.p, d_J, (int32_t *)S.ancl.p,
               (int32_t *)S.ancr.p);
        LAUNCH(c, "kf_anchors", kf_anchors, dim3(pair_blocks), dim3(256), sidx, (const u32 *)S.jid.p, (const PairRec *)pr.rec, d_P, (u32 *)S.jidbam.p,
               (int32_t *)S.ancl.p, (int32_t *)S.ancr.p);
        if ((rc = launch_k4b())) return rc;
        STAGE_EVENT(4);
        LAUNCH(c, "k4_pairs", k4_pairs, dim3(pair_blocks), dim3(256), sidx, jid_sorted, (const PairRec *)pr.rec, (const u64 *)S.jkey.p, kf, d_P,
               (u32 *)S.frag.p, (int32_t *)S.fragj.p, (u64 *)nullptr, (u64 *)nullptr, (const ContigStats *)d_cs, d_err);
        STAGE_EVENT(5);
    }

    // ---- K5: fragments -> junctions -> rows
    LAUNCH(c, "k5_frag_reduce", k5_frag_reduce, dim3((slots_lim + 4 * FRAG_SLOTS_PER_WAVE - 1) / (4 * FRAG_SLOTS_PER_WAVE)), dim3(256),
           (const u32 *)S.frag.p, (const int32_t *)S.fragj.p, d_slots, (u32 *)S.acc.p);
    if (entropy_forked) HIP_TRY(c, hipStreamWaitEvent(tl, S.ev_join2, 0));
    LAUNCH(c, "k5_finalize", k5_finalize, dim3(std::max<u32>(1, (JL + 255) / 256)), dim3(256), (const u64 *)S.jkey.p, (const u32 *)S.seg.p,
           (const u32 *)S.runfirst.p, (const u32 *)S.runstart.p, (const u32 *)S.acc.p,
           (const int32_t *)S.ancl.p, (const int32_t *)S.ancr.p, kf, GT, d_J, (const double *)S.entsum.p, (pjb_junction_row *)S.rows.p, d_err,
           d_member_junc);
    STAGE_EVENT(6);

    // ---- rows to the host (and into the caller's exchange slot), control block last: on the rows stream, so that the
    // next contig's first kernels need not wait for the PCIe writes
    u64 *mirror_table = nullptr;
    u32 mirror_room = 0; // rows the caller's slot can take (the kernel leaves the slot alone if the contig does not fit)
    if (c->mirror && c->mirror_cap >= PJB_MIRROR_HEADER_BYTES) {
        mirror_table = (u64 *)(c->mirror + PJB_MIRROR_HEADER_BYTES);
        mirror_room = (u32)std::min<size_t>((c->mirror_cap - PJB_MIRROR_HEADER_BYTES) / sizeof(pjb_junction_row), 0xffffffffu);
    }
    const hipStream_t rows_stream = c->side_stream ? c->stream3 : tl;
    if (rows_stream != tl) {
        HIP_TRY(c, hipEventRecord(S.ev_rows, tl));
        HIP_TRY(c, hipStreamWaitEvent(rows_stream, S.ev_rows, 0));
    }
    {
        struct StreamScope {
            pjb_ctx *c;
            hipStream_t main;
            ~StreamScope() { c->stream = main; }
        } scope{c, c->stream};
        c->stream = rows_stream; // LAUNCH (and its event bracket) follow c->stream
        LAUNCH(c, "k6_rows_out", k6_rows_out, dim3(K6_BLOCKS), dim3(256), (const u64 *)S.rows.p, (const ContigStats *)d_cs,
               (u64 *)c->rows_table, row_base, mirror_base, (RowCursor *)c->b_cursor.p, mirror_table, mirror_room, d_err, d_gen_cnt, S.pub_dev,
               (const MemberStats *)d_members, d_member_junc, n_members);
    }
    S.at_rest = true;
    HIP_TRY(c, hipEventRecord(S.ev[7], rows_stream));
    HIP_TRY(c, hipEventRecord(S.ev_done, rows_stream));
#undef STAGE_EVENT
    f.queued = true;
    return PJB_OK;
}

// blocks until the contig's rows and control block are on the host
static void wait_flight(pjb_ctx *c, Flight &f) {
    if (f.queued) (void)hipEventSynchronize(c->sl[f.slot].ev_done);
}

// The contigs queued behind fl[0] are taken back (after an overflow or an error of fl[0] their rows are in the wrong
// place): their device work is allowed to finish and is thrown away; their own pjb_finish_contig_end queues them again.
static void unqueue_followers(pjb_ctx *c) {
    for (int k = 1; k < c->n_fl; k++) {
        Flight &g = c->fl[k];
        if (!g.queued) continue;
        wait_flight(c, g);
        if (g.forked) (void)hipStreamSynchronize(c->sl[g.slot].side);
        g.queued = false;
        g.forked = false;
        ev_drop(c, g.slot);
    }
}

static void pop_flight(pjb_ctx *c) {
    if (c->n_fl <= 0) return;
    c->slot_busy[c->fl[0].slot] = false;
    for (int k = 1; k < c->n_fl; k++) c->fl[k - 1] = c->fl[k];
    c->n_fl--;
    c->fl[c->n_fl] = Flight();
}

// The flight's batch list, virtual offsets, tile ranges and limits from the open targets named in f.tids.
static void prepare_flight(pjb_ctx *c, Flight &f) {
    f.batches.clear();
    f.batch_member.clear();
    f.voff.assign(f.tids.size(), 0);
    f.tile_lo.assign(f.tids.size() + 1, 0);
    f.m_reads.assign(f.tids.size(), 0);
    f.n_tiles = 0;
    f.n_reads = 0;
    int64_t at = 0;
    bool genomes_ok = true;
    for (size_t m = 0; m < f.tids.size(); m++) {
        const int32_t tid = f.tids[m];
        const int32_t ref_len = c->ref_len[(size_t)tid];
        f.voff[m] = (int32_t)at;
        at += (((int64_t)std::max(ref_len, 1) + GROUP_GAP) + 63) & ~(int64_t)63;
        f.tile_lo[m] = f.n_tiles;
        const Contig &G = c->contigs[(size_t)tid];
        auto open_it = c->open.find(tid);
        if (open_it == c->open.end() || open_it->second.batches.empty()) continue;
        // without the target's genome only the counting stage may run: any pair then shows up as an overflow
        if (!G.present || G.len != ref_len) genomes_ok = false;
        OpenContig &oc = open_it->second;
        int32_t prev_pos = INT32_MIN;
        const int32_t *prev_ptr = nullptr;
        for (size_t k = 0; k < oc.batches.size(); k++) {
            DevBatch b = oc.batches[k];
            b.base = (uint32_t)f.n_reads; // read ordinals and tile numbers run through the whole group
            b.tile_base = f.n_tiles;
            f.n_tiles += (u32)((b.n + K1_TILE - 1) / K1_TILE);
            b.prev_pos = prev_pos;
            b.prev_pos_ptr = prev_ptr; // sortedness across batches: the last position of the previous batch, wherever it is known
            f.n_reads += b.n;
            f.m_reads[m] += b.n;
            if (oc.last_known[k]) {
                prev_pos = oc.last_pos[k];
                prev_ptr = nullptr;
            } else
                prev_ptr = b.pos + (b.n - 1);
            b.member = (int32_t)m;
            f.batches.push_back(b);
            f.batch_member.push_back((int)m);
        }
    }
    f.tile_lo[f.tids.size()] = f.n_tiles;
    f.vlen = f.tids.size() > 1 ? at : c->ref_len[(size_t)f.tid];
    f.empty = f.batches.empty();
    // ---- limits.  Pairs: a share of the reads (a chain with more N operations than that is repeated once with the
    // exact count).  Junctions: a share of the pair limit, at least twice what a chain of this context has had.  Key:
    // coordinates of the (virtual) sequence and the longest intron seen by this context so far.
    ContigLimits &lim = f.lim;
    const u64 guess = std::min<u64>(0xffffff00ull, (u64)f.n_reads * 5 / 8 + 4096);
    lim.pair_limit = (u32)std::max<u64>(guess, 4096);
    lim.junc_limit = std::max<u32>(std::max<u32>(lim.pair_limit / 32, 4096), 2 * c->junc_seen * (u32)std::max<size_t>(1, f.tids.size() > 1 ? 2 : 1));
    if (!genomes_ok) lim.pair_limit = 0;
    lim.kf.raw = 0;
    lim.kf.lbits = std::max(1, c->lbits_seen);
    lim.kf.total_bits = lim.kf.lbits + std::max(1, bits_of((uint64_t)std::max<int64_t>(f.vlen, 1)));
    lim.dense = c->dense_ids;
    // (twice what a chain of this context's targets has had, per member; a chain with more is repeated with digits for junc_limit)
    // (junctions per read: chains of one file have about the same, whatever their targets' number and size)
    lim.sort_limit = c->junc_per_read > 0 ? std::min<u32>(lim.junc_limit, std::max<u32>(c->sort_floor, (u32)std::min<double>(4.0e9, 2.0 * c->junc_per_read * (double)f.n_reads + 64.0))) : 0u;
}

// streams and events of a control slot, at its first use (a context that never queues eight chains never pays for them)
static int slot_init(pjb_ctx *c, int k) {
    CtlSlot &S = c->sl[k];
    if (S.ev_done) return PJB_OK;
    for (auto &ev : S.ev) HIP_TRY(c, hipEventCreate(&ev));
    HIP_TRY(c, hipEventCreateWithFlags(&S.ev_rows, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&S.ev_done, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&S.ev_k1, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&S.ev_xk1, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&S.ev_fork, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&S.ev_join, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&S.ev_fork2, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&S.ev_join2, hipEventDisableTiming));
    // (main before side, slot after slot: the order in which streams are created decides which hardware queue they share, and
    // every other order tried -- side first, all mains first, one to three streams created before -- cost the configs[2] step
    // 5 - 12 %: profiles/r03z_stream_order.txt)
    HIP_TRY(c, hipStreamCreateWithFlags(&S.main, hipStreamNonBlocking));
    HIP_TRY(c, hipStreamCreateWithFlags(&S.side, hipStreamNonBlocking));
    return PJB_OK;
}

// the chain and, with PJB_FLAG_EXTRA, the part of the extra metrics that needs the records only (service stream, beside the chain)
static int queue_chain(pjb_ctx *c, Flight &f) {
    int rc = queue_contig(c, f);
    if (!rc && c->extra) rc = extra_pre(c, f);
    return rc;
}

static int begin_flight(pjb_ctx *c, const int32_t *tids, int32_t n, const char *who) {
    if (!c) return PJB_ERR_ARG;
    if (!tids || n < 1 || n > GROUP_MAX) return fail(c, PJB_ERR_ARG, "%s: 1 to %d targets", who, GROUP_MAX);
    aux_streams(c);
    for (int32_t k = 0; k < n; k++) {
        if (tids[k] < 0 || (size_t)tids[k] >= c->ref_len.size()) return fail(c, PJB_ERR_ARG, "%s: bad tid %d", who, tids[k]);
        for (int32_t q = 0; q < k; q++)
            if (tids[q] == tids[k]) return fail(c, PJB_ERR_ARG, "%s: target %d named twice", who, tids[k]);
    }
    if (c->n_fl >= PJB_MAX_QUEUED)
        return fail(c, PJB_ERR_STATE, "%s: %d chains are queued already (oldest: target %d); collect one first", who, c->n_fl, c->fl[0].tid);
    for (int k = 0; k < c->n_fl; k++)
        for (int32_t t : c->fl[k].tids)
            for (int32_t q = 0; q < n; q++)
                if (t == tids[q]) return fail(c, PJB_ERR_STATE, "%s: target %d is queued already", who, t);
    if (n > 1) {
        // what a group needs (a caller that gets PJB_ERR_ARG here finishes the targets one by one)
        if (c->extra) return fail(c, PJB_ERR_ARG, "%s: PJB_FLAG_EXTRA contexts finish one target at a time", who);
        int64_t vlen = 0;
        for (int32_t k = 0; k < n; k++) {
            const Contig &G = c->contigs[(size_t)tids[k]];
            vlen += (((int64_t)std::max(c->ref_len[(size_t)tids[k]], 1) + GROUP_GAP) + 63) & ~(int64_t)63;
            auto it = c->open.find(tids[k]);
            const bool has_reads = it != c->open.end() && !it->second.batches.empty();
            if (has_reads && (!G.present || G.len != c->ref_len[(size_t)tids[k]]))
                return fail(c, PJB_ERR_ARG, "%s: the genome of target %d has not been uploaded", who, tids[k]);
            if (has_reads && (!G.codes || G.has_x))
                return fail(c, PJB_ERR_ARG, "%s: target %d has characters outside the 16-letter alphabet: finish it alone", who, tids[k]);
        }
        if (vlen >= (int64_t)INT32_MAX - (1 << 20)) return fail(c, PJB_ERR_ARG, "%s: the group's targets add up to %lld bases (limit 2^31)", who, (long long)vlen);
    }
    c->cur_tid = tids[0];
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    Flight &f = c->fl[c->n_fl];
    int slot = 0;
    while (slot < PJB_MAX_QUEUED && c->slot_busy[slot]) slot++;
    if (slot >= PJB_MAX_QUEUED) return fail(c, PJB_ERR_STATE, "%s: no free control slot", who);
    {
        const int rc = slot_init(c, slot);
        if (rc) return rc;
    }
    f = Flight();
    f.slot = slot;
    f.tid = tids[0];
    f.tids.assign(tids, tids + n);
    c->slot_busy[slot] = true;
    prepare_flight(c, f);
    c->n_fl++;
    if (f.empty) return PJB_OK;
    // a chain that was taken back goes first (rows are in queue order): the end of the oldest one queues them all
    for (int k = 0; k + 1 < c->n_fl; k++)
        if (!c->fl[k].queued && !c->fl[k].empty) return PJB_OK;
    const int rc = queue_chain(c, f);
    if (rc) { // nothing of this chain stays behind
        (void)hipDeviceSynchronize();
        c->n_fl--;
        c->slot_busy[c->fl[c->n_fl].slot] = false;
        std::vector<int32_t> members = c->fl[c->n_fl].tids;
        c->fl[c->n_fl] = Flight();
        for (int32_t t : members) close_contig(c, t);
    }
    return rc;
}

int pjb_finish_contig_begin(pjb_ctx *c, int32_t tid) { return begin_flight(c, &tid, 1, "finish"); }
int pjb_finish_group_begin(pjb_ctx *c, const int32_t *tids, int32_t n_tids) { return begin_flight(c, tids, n_tids, "finish_group"); }

// Collects the chain in fl[0] -- waits for it, repeats it when a limit it was queued with turned out too small -- and
// fills one result per member.  A group whose members turn out to hold alignments outside their own sequence is taken
// apart: *redo_single is set and nothing is committed.
static int collect_flight(pjb_ctx *c, pjb_region_result *res, bool *redo_single) {
    Flight &f = c->fl[0];
    const int n_members = (int)f.tids.size();
    const bool group = n_members > 1;
    CtlSlot &S = c->sl[f.slot];
    ContigLimits &lim = f.lim;
    const int32_t tid = f.tid;
    const int32_t ref_len = group ? (int32_t)f.vlen : c->ref_len[(size_t)tid];
    int rc;
    ContigStats cs;
    u64 herr = ~0ull;
    int64_t repeats = 0, repeat_reasons = 0;
    for (;; f.attempt++) {
        if (!f.queued && (rc = queue_chain(c, f))) return rc;
        wait_flight(c, f);
        memcpy(&cs, S.pub, sizeof cs);
        memcpy(&herr, S.pub + PUB_ERR_AT, 8);
        if ((rc = check_device_error(c, herr))) return rc;
        for (int m = 0; m < n_members && cs.n_pairs > 0; m++) {
            const Contig &G = c->contigs[(size_t)f.tids[(size_t)m]];
            if (f.m_reads[(size_t)m] == 0) continue;
            if (!G.present) return fail(c, PJB_ERR_STATE, "finish: genome of target %d was not uploaded", f.tids[(size_t)m]);
            if (G.len != c->ref_len[(size_t)f.tids[(size_t)m]])
                return fail(c, PJB_ERR_ARG, "finish: genome of target %d has %lld bases, header says %d", f.tids[(size_t)m], (long long)G.len,
                            c->ref_len[(size_t)f.tids[(size_t)m]]);
        }
        if (!cs.overflow) break;
        // (a repeat for the read lists' room alone has a budget of its own: it must not use up the attempts the other limits may need)
        if (cs.overflow == OVF_LISTS && f.list_attempt < 3) {
            f.list_attempt++;
            f.attempt--;
        } else if (f.attempt >= 3)
            return fail(c, PJB_ERR_STATE, "finish: limits of target %d did not settle (overflow bits %u)", tid, cs.overflow);
        // a limit was too small: the control block says by how much; everything is queued again (and so is the chain
        // queued behind this one: its rows went where this one's belong)
        repeats++;
        repeat_reasons |= (int64_t)cs.overflow;
        unqueue_followers(c);
        if (f.forked) (void)hipStreamSynchronize(c->sl[f.slot].side);
        f.queued = f.forked = false;
        if (cs.overflow & OVF_PAIRS) {
            if (cs.n_pairs >= 0xfffffff0ull)
                return fail(c, PJB_ERR_ARG, group ? "finish_group: more than 2^32 spliced pairs in one group: finish its targets in smaller groups"
                                                  : "finish: more than 2^32 spliced pairs on one target are not supported");
            lim.pair_limit = (u32)cs.n_pairs + 64;
            lim.junc_limit = std::max<u32>(lim.junc_limit, lim.pair_limit / 8);
        }
        if (cs.overflow & OVF_KEYFMT) {
            const bool weird = cs.min_pos < 0 || cs.max_end > ref_len || cs.max_end < 0;
            if (weird && group) { // (k1_count reports INT32_MAX for a member whose alignments leave it)
                *redo_single = true;
                return PJB_OK;
            }
            if (weird) {
                lim.kf.raw = 1;
                lim.kf.lbits = 32;
                lim.kf.total_bits = 64;
                lim.dense = false; // the bitmap of intron starts needs coordinates inside the contig
            } else {
                lim.kf.lbits = std::max(1, bits_of((uint64_t)cs.max_nlen));
                lim.kf.total_bits = lim.kf.lbits + std::max(1, bits_of((uint64_t)std::max(ref_len, 1)));
            }
        }
        const bool sort_only = (cs.overflow & OVF_JUNC) && lim.sort_limit && cs.n_junc <= lim.junc_limit; // (the buffers were large enough)
        if (cs.overflow & OVF_JUNC) lim.sort_limit = 0;
        if ((cs.overflow & OVF_JUNC) && !sort_only) lim.junc_limit = std::max<u32>(cs.n_junc + 64, (cs.overflow & OVF_DENSE) || !lim.dense ? 0u : lim.junc_limit * 4);
        if (cs.overflow & OVF_DENSE) lim.dense = false; // a donor with more alternative acceptors than K2d keeps: sort the full keys
        if (cs.overflow & OVF_LISTS) lim.list_cap = std::max(gen_list_cap(lim.pair_limit), (cs.list_need + cs.list_need / 4 + 511u) & ~255u); // (k1_generic's entries depend on the appends' order: some slack)
    }
    if (!lim.kf.raw) c->lbits_seen = std::max(c->lbits_seen, std::max(1, bits_of((uint64_t)cs.max_nlen)));
    const u32 P = cs.P, J = cs.J;
    // ---- one result per member
    pjb_region_result Rsum;
    memset(&Rsum, 0, sizeof Rsum);
    Rsum.min_len = INT32_MAX;
    if (!group) {
        pjb_region_result &R = res[0];
        R.spliced = cs.spliced;
        R.unspliced = cs.unspliced;
        R.sum_len = cs.sum_len;
        R.min_len = cs.min_len;
        R.max_len = cs.max_len;
        R.n_reads = f.n_reads;
        R.n_pairs = (int64_t)cs.n_pairs;
        R.n_junctions = J;
        Rsum = R;
    } else {
        const MemberStats *ms = (const MemberStats *)(S.pub + PUB_MEMBERS_AT);
        u64 pairs = 0, juncs = 0;
        for (int m = 0; m < n_members; m++) {
            pjb_region_result &R = res[m];
            memset(&R, 0, sizeof R);
            R.min_len = INT32_MAX;
            if (f.m_reads[(size_t)m] == 0) continue;
            R.spliced = ms[m].spliced;
            R.unspliced = ms[m].unspliced;
            R.sum_len = ms[m].sum_len;
            R.min_len = ms[m].min_len;
            R.max_len = ms[m].max_len;
            R.n_reads = f.m_reads[(size_t)m];
            R.n_pairs = (int64_t)ms[m].n_pairs;
            R.n_junctions = ms[m].n_junc;
            pairs += ms[m].n_pairs;
            juncs += ms[m].n_junc;
        }
        if (pairs != cs.n_pairs || juncs != J)
            return fail(c, PJB_ERR_STATE, "finish_group: members hold %llu pairs / %llu junctions, the chain %llu / %u", (unsigned long long)pairs,
                        (unsigned long long)juncs, (unsigned long long)cs.n_pairs, J);
        Rsum.spliced = cs.spliced;
        Rsum.unspliced = cs.unspliced;
        Rsum.sum_len = cs.sum_len;
        Rsum.min_len = cs.min_len;
        Rsum.max_len = cs.max_len;
    }
    c->junc_seen = std::max(c->junc_seen, group ? J / (u32)n_members : J);
    if (f.n_reads > 0) c->junc_per_read = std::max(c->junc_per_read, (double)J / (double)f.n_reads);
    c->timing.sort_passes = f.n_pass;
    c->timing.repeats = repeats;
    c->timing.repeat_reasons = repeat_reasons;
    c->timing.generic_pairs = 0;
    c->timing.generic_reads = 0;
    c->timing.checked_reads = *(const u32 *)(S.pub + PUB_CHECKED_AT);
    for (u32 k = 0; k < GEN_SHARDS; k++) {
        c->timing.generic_pairs += ((const u32 *)(S.pub + PUB_GEN_AT))[k];
        c->timing.generic_reads += ((const u32 *)(S.pub + PUB_GREADS_AT))[k];
    }
    c->timing.position_runs = cs.R;
    c->timing.candidates = lim.dense ? cs.n_cand : 0;
    const size_t old = c->rows_n;
    {
        u32 at[2];
        memcpy(at, S.pub + PUB_BASE_AT, 8);
        if (at[0] != (u32)old || (c->mirror && at[1] != (u32)c->mirror_rows))
            return fail(c, PJB_ERR_STATE, "finish: rows of target %d went to %u (exchange slot %u), expected %zu (%zu)", tid, at[0], at[1], old, c->mirror_rows);
    }
    if (c->mirror) { // the rows are in the exchange slot already (k6_rows_out); the header follows, covered by a small wait
        const size_t need = PJB_MIRROR_HEADER_BYTES + (c->mirror_rows + J) * sizeof(pjb_junction_row);
        if (need > c->mirror_cap) return fail(c, PJB_ERR_ARG, "finish: %zu rows do not fit the row mirror (%zu bytes)", c->mirror_rows + J, c->mirror_cap);
        mirror_fold(c, Rsum, J);
        HIP_TRY(c, hipMemcpyAsync(c->mirror, c->mirror_hdr, PJB_MIRROR_HEADER_BYTES, hipMemcpyHostToDevice, c->stream4));
        HIP_TRY(c, hipStreamSynchronize(c->stream4));
    }
    if (J) { // the chain's rows: HBM table -> host table, by DMA, behind whatever the caller does next
        if ((rc = rows_pinned_reserve(c, old + J))) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->rows_pinned + old, c->rows_table + old, (size_t)J * sizeof(pjb_junction_row), hipMemcpyDeviceToHost, c->stream4));
        c->rows_copy_pending = true;
    }
    c->cur_slot = f.slot;
    if (c->extra && (rc = extra_contig(c, f, tid, cs.spliced, P, J, old))) return rc;
    c->rows_n = old + J;
    c->last_rows_n = J;
    c->last_slot = f.slot;
    if (c->ktime) ev_collect(c, f.slot);
    if (c->ktime && c->ktime_only.empty())
        for (int k = 0; k < 7; k++) (void)hipEventElapsedTime(&c->timing.stage_ms[k], S.ev[k], S.ev[k + 1]);
    (void)hipEventElapsedTime(&c->timing.total_ms, S.ev[0], S.ev[7]);
    return PJB_OK;
}

static int end_flight(pjb_ctx *c, const int32_t *tids, int32_t n, pjb_region_result *res, const char *who) {
    if (!c) return PJB_ERR_ARG;
    if (!tids || n < 1) return fail(c, PJB_ERR_ARG, "%s: no targets", who);
    bool same = c->n_fl > 0 && (int32_t)c->fl[0].tids.size() == n;
    for (int32_t k = 0; same && k < n; k++) same = c->fl[0].tids[(size_t)k] == tids[k];
    if (!same)
        return fail(c, PJB_ERR_STATE, "%s: target %d (and the %d named with it) is not the oldest queued chain (begin first; collect in the same order, "
                                      "with the same targets)", who, tids[0], n - 1);
    c->cur_tid = tids[0];
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    struct Closer { // also on every error path: the side stream (k4b_generic) may still read the contig's batches
        pjb_ctx *c;
        bool ok = false;
        ~Closer() {
            if (!ok) { // whatever is queued is in an unknown place now
                (void)hipDeviceSynchronize();
                unqueue_followers(c);
            }
            const std::vector<int32_t> members = c->fl[0].tids;
            pop_flight(c);
            for (int32_t t : members) close_contig(c, t);
        }
    } closer{c};
    std::vector<pjb_region_result> tmp((size_t)n);
    for (auto &R : tmp) {
        memset(&R, 0, sizeof R);
        R.min_len = INT32_MAX;
    }
    memset(&c->timing, 0, sizeof c->timing);
    c->last_rows_n = 0;
    if (res) memcpy(res, tmp.data(), tmp.size() * sizeof(pjb_region_result));
    int rc;
    if (c->fl[0].empty) {
        rc = mirror_header_only(c, tmp[0]);
        closer.ok = rc == PJB_OK;
        return rc;
    }
    bool redo_single = false;
    if ((rc = collect_flight(c, tmp.data(), &redo_single))) return rc;
    if (redo_single) {
        // a member holds alignments outside its own sequence: the members go through one by one, in this chain's slot
        const Flight whole = c->fl[0];
        size_t rows_total = 0;
        for (int32_t m = 0; m < n; m++) {
            Flight one = Flight();
            one.slot = whole.slot;
            one.tid = whole.tids[(size_t)m];
            one.tids.assign(1, one.tid);
            prepare_flight(c, one);
            c->fl[0] = one;
            if (one.empty) continue;
            bool again = false;
            rc = collect_flight(c, &tmp[(size_t)m], &again);
            if (rc) {
                c->fl[0] = whole; // (the closer releases every member)
                return rc;
            }
            rows_total += c->last_rows_n;
        }
        c->fl[0] = whole;
        c->fl[0].queued = c->fl[0].forked = false;
        c->last_rows_n = rows_total; // (pjb_collect_device covers the last member only: a caller of groups uses pjb_collect)
    }
    if (res) memcpy(res, tmp.data(), tmp.size() * sizeof(pjb_region_result));
    closer.ok = true;
    // followers that were taken back (or waited for this one) are queued now, in order, the first one's place known
    c->fl[0].queued = false; // (fl[0] is still this chain: its rows are collected, it is not "ahead" of anything)
    for (int k = 1; k < c->n_fl; k++) {
        Flight &g = c->fl[k];
        if (g.queued || g.empty) continue;
        // a follower that cannot be queued here is queued again -- and reports its error -- by its own _end; THIS
        // chain is collected and its rows are in the table
        if (queue_chain(c, g)) break;
    }
    return PJB_OK;
}

int pjb_finish_ready(pjb_ctx *c) {
    if (!c || c->n_fl <= 0) return 1;
    const Flight &f = c->fl[0];
    if (f.empty || !f.queued) return 1;
    const bool done = hipEventQuery(c->sl[f.slot].ev_done) == hipSuccess;
    (void)hipGetLastError(); // (hipErrorNotReady is not an error here)
    return done ? 1 : 0;
}

int pjb_finish_contig_end(pjb_ctx *c, int32_t tid, pjb_region_result *res) { return end_flight(c, &tid, 1, res, "finish"); }
int pjb_finish_group_end(pjb_ctx *c, const int32_t *tids, int32_t n_tids, pjb_region_result *results) {
    return end_flight(c, tids, n_tids, results, "finish_group");
}

int pjb_finish_contig(pjb_ctx *c, int32_t tid, pjb_region_result *res) {
    const int rc = pjb_finish_contig_begin(c, tid);
    if (rc) return rc;
    return pjb_finish_contig_end(c, tid, res);
}

int pjb_collect(pjb_ctx *c, const pjb_junction_row **rows, int64_t *n) {
    if (!c || !rows || !n) return PJB_ERR_ARG;
    int rc = rows_sync(c);
    if (rc) return rc;
    if (!c->rows_pinned && (rc = rows_pinned_reserve(c, 1))) return rc; // (a table, if an empty one: the pointer is never null)
    *rows = c->rows_pinned;
    *n = (int64_t)c->rows_n;
    return PJB_OK;
}

int pjb_collect_device(pjb_ctx *c, const pjb_junction_row **rows, int64_t *n) {
    if (!c || !rows || !n) return PJB_ERR_ARG;
    *rows = (const pjb_junction_row *)c->sl[c->last_slot].rows.p;
    *n = (int64_t)c->last_rows_n;
    return PJB_OK;
}

int pjb_set_row_mirror(pjb_ctx *c, void *device_buffer, int64_t cap_bytes) {
    if (!c) return PJB_ERR_ARG;
    if (device_buffer && cap_bytes < PJB_MIRROR_HEADER_BYTES) return fail(c, PJB_ERR_ARG, "set_row_mirror: buffer smaller than its header");
    if (c->n_fl) return fail(c, PJB_ERR_STATE, "set_row_mirror: target %d is still queued", c->fl[0].tid);
    c->mirror = (uint8_t *)device_buffer;
    c->mirror_cap = device_buffer ? (size_t)cap_bytes : 0;
    mirror_reset(c);
    if (c->mirror && !c->mirror_hdr) {
        HIP_TRY(c, hipSetDevice(c->cfg.device));
        HIP_TRY(c, hipHostMalloc((void **)&c->mirror_hdr, PJB_MIRROR_HEADER_BYTES, hipHostMallocDefault));
    }
    return PJB_OK;
}

int pjb_plan_groups(const int32_t *ref_len, const int32_t *tids, int32_t n_tids, int64_t max_bases, int32_t *group_of) {
    if (n_tids < 0 || (n_tids > 0 && (!ref_len || !tids || !group_of))) return PJB_ERR_ARG;
    if (max_bases <= 0) max_bases = (int64_t)1 << 30;
    auto span = [&](int32_t k) { return (((int64_t)std::max(ref_len[tids[k]], 1) + GROUP_GAP) + 63) & ~(int64_t)63; };
    auto plan = [&](int64_t cap) {
        int32_t g = 0, members = 0;
        int64_t tot = 0;
        for (int32_t k = 0; k < n_tids; k++) {
            const int64_t s = span(k);
            if (members && (tot + s > cap || members >= GROUP_MAX)) g++, members = 0, tot = 0;
            group_of[k] = g;
            members++;
            tot += s;
        }
        return n_tids ? g + 1 : 0;
    };
    for (int32_t k = 0; k < n_tids; k++)
        if (tids[k] < 0) return PJB_ERR_ARG;
    int32_t n = plan(max_bases);
    if (n == 1 && n_tids > 1) { // ONE chain of more than 0.6 Gb: two, so that the first one's tail has the second one's K1 stage beside it
        int64_t total = 0;
        for (int32_t k = 0; k < n_tids; k++) total += span(k);
        if (total > 600000000) n = plan((int64_t)((double)total * 0.55));
    }
    return n;
}

int pjb_merge_rows(const void *gathered, int32_t n_ranks, int64_t slot_stride_bytes, pjb_junction_row *rows_out, int64_t cap_rows, int64_t *n_rows,
                   pjb_region_result *totals) {
    if (n_rows) *n_rows = 0;
    if (!gathered || n_ranks < 1 || slot_stride_bytes < PJB_MIRROR_HEADER_BYTES || !n_rows || !totals || cap_rows < 0 || (cap_rows > 0 && !rows_out))
        return PJB_ERR_ARG;
    struct Run { // rows of one target in one rank's slot
        int32_t refid;
        const pjb_junction_row *first;
        int64_t n;
    };
    std::vector<Run> runs;
    pjb_region_result T;
    memset(&T, 0, sizeof T);
    T.min_len = INT32_MAX;
    int64_t total = 0;
    for (int32_t r = 0; r < n_ranks; r++) {
        const uint8_t *slot = (const uint8_t *)gathered + (size_t)r * (size_t)slot_stride_bytes;
        int64_t h[6];
        memcpy(h, slot, sizeof h); // n_rows, spliced, unspliced, sum_len, min_len, max_len
        if (h[0] < 0 || h[0] > (slot_stride_bytes - PJB_MIRROR_HEADER_BYTES) / (int64_t)sizeof(pjb_junction_row)) return PJB_ERR_ARG;
        T.spliced += (uint64_t)h[1];
        T.unspliced += (uint64_t)h[2];
        T.sum_len += (uint64_t)h[3];
        T.min_len = std::min<int32_t>(T.min_len, (int32_t)std::min<int64_t>(h[4], INT32_MAX));
        T.max_len = std::max<int32_t>(T.max_len, (int32_t)h[5]);
        const pjb_junction_row *rows = (const pjb_junction_row *)(slot + PJB_MIRROR_HEADER_BYTES);
        for (int64_t i = 0; i < h[0];) {
            int64_t j = i + 1;
            while (j < h[0] && rows[j].refid == rows[i].refid) j++;
            runs.push_back(Run{rows[i].refid, rows + i, j - i});
            i = j;
        }
        total += h[0];
    }
    T.n_reads = (int64_t)(T.spliced + T.unspliced);
    T.n_junctions = total;
    *totals = T;
    *n_rows = total;
    if (total > cap_rows) return PJB_ERR_ARG;
    std::stable_sort(runs.begin(), runs.end(), [](const Run &a, const Run &b) { return a.refid < b.refid; });
    pjb_junction_row *out = rows_out;
    for (const Run &u : runs) {
        memcpy(out, u.first, (size_t)u.n * sizeof(pjb_junction_row));
        out += u.n;
    }
    return PJB_OK;
}

int pjb_clear_rows(pjb_ctx *c) {
    if (!c) return PJB_ERR_ARG;
    if (c->n_fl) return fail(c, PJB_ERR_STATE, "clear_rows: target %d is still queued", c->fl[0].tid);
    (void)rows_sync(c);
    c->rows_n = 0;
    (void)exhume(c); // (nothing is queued: the buffers that were replaced while chains ran go back now)
    mirror_reset(c);
    if (c->extra) {
        (void)hipSetDevice(c->cfg.device);
        (void)hipStreamSynchronize(c->stream);
        extra_clear(c);
    }
    return PJB_OK;
}

int pjb_get_kernel_timing(const pjb_ctx *c, pjb_kernel_time *out, int32_t cap, int32_t *n) {
    if (!c || !n) return PJB_ERR_ARG;
    *n = (int32_t)c->knames.size();
    for (int32_t i = 0; out && i < cap && i < *n; i++) {
        memset(&out[i], 0, sizeof out[i]);
        strncpy(out[i].name, c->knames[(size_t)i].c_str(), sizeof(out[i].name) - 1);
        out[i].launches = c->kcount[(size_t)i];
        out[i].total_ms = c->kms[(size_t)i];
    }
    return PJB_OK;
}

int pjb_select_timed_kernels(pjb_ctx *c, const char *comma_separated_names) {
    if (!c) return PJB_ERR_ARG;
    c->ktime_only.clear();
    std::string s = comma_separated_names ? comma_separated_names : "";
    size_t a = 0;
    while (a < s.size()) {
        size_t b = s.find(',', a);
        if (b == std::string::npos) b = s.size();
        if (b > a) c->ktime_only.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    return PJB_OK;
}

int pjb_reset_kernel_timing(pjb_ctx *c) {
    if (!c) return PJB_ERR_ARG;
    std::fill(c->kcount.begin(), c->kcount.end(), 0);
    std::fill(c->kms.begin(), c->kms.end(), 0.0);
    return PJB_OK;
}

int pjb_set_option(pjb_ctx *c, const char *name, int64_t value) {
    if (!c || !name) return PJB_ERR_ARG;
    if (c->n_fl) return fail(c, PJB_ERR_STATE, "set_option: target %d is still queued", c->fl[0].tid);
    const std::string n = name;
    if (n == "overlap") c->side_stream = value != 0;
    else if (n == "dense_ids") c->dense_ids = value != 0;
    else if (n == "extra_dense") c->extra_dense_only = value != 0;
    else if (n == "sort_floor") c->sort_floor = (u32)std::max<int64_t>(1, std::min<int64_t>(value, 1 << 30));
    else if (n == "list_cap") c->list_cap_forced = (u32)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 30));
    else return fail(c, PJB_ERR_ARG, "set_option: unknown option '%s'", name);
    return PJB_OK;
}

int pjb_get_timing(const pjb_ctx *c, pjb_timing *out) {
    if (!c || !out) return PJB_ERR_ARG;
    if (c->cfg.abi_version >= 4) *out = c->timing;
    else memcpy(out, &c->timing, offsetof(pjb_timing, repeats)); // (ABI 3's pjb_timing ended at checked_reads)
    return PJB_OK;
}

} // extern "C"
