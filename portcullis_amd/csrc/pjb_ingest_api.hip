// pjb_ingest_api.hip -- the part of the C ABI that takes file bytes: BGZF inflate / deflate on the device, BAM record boundaries and transcoding
// (pjb_inflate_bgzf, pjb_deflate_bgzf, pjb_submit_bam, pjb_bam_*); kernels in pjb_ingest.hip.h and pjb_deflate.hip.h.
#include "pjb_host.hip.h"
#include "pjb_deflate.hip.h"
#include "pjb_ingest.hip.h"

void ingest_kernel_attributes() { (void)hipFuncSetAttribute((const void *)bgzf_decode, hipFuncAttributeMaxDynamicSharedMemorySize, I3_LDS_BYTES); }
int ingest_lds_bytes() { return I3_LDS_BYTES; }


// ---- device-side ingest ---------------------------------------------------------------------------
const char *inf_text(int code) {
    switch (code) {
    case INF_ERR_BTYPE: return "reserved DEFLATE block type";
    case INF_ERR_STORED: return "stored block length check failed";
    case INF_ERR_CODELENS: return "invalid code length set";
    case INF_ERR_CODE: return "invalid Huffman code";
    case INF_ERR_DIST: return "match distance before the start of the block";
    case INF_ERR_OVERRUN: return "block inflates or reads past its declared size";
    case INF_ERR_SIZE: return "block inflates to fewer bytes than its ISIZE";
    default: return "bad block";
    }
}

// hop over the BGZF block headers (bgzf.c:348-356 check_header, BSIZE from the BC extra subfield)
int scan_bgzf(pjb_ctx *c, const uint8_t *comp, int64_t n, std::vector<InfBlock> &blocks, int64_t &total_out) {
    int64_t off = 0;
    total_out = 0;
    while (off < n) {
        if (off + 18 > n) return fail(c, PJB_ERR_BGZF, "truncated BGZF block header at byte %lld", (long long)off);
        const uint8_t *h = comp + off;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4))
            return fail(c, PJB_ERR_BGZF, "not a BGZF block header at byte %lld", (long long)off);
        const uint32_t xlen = h[10] | (uint32_t)h[11] << 8;
        if (off + 12 + xlen > n) return fail(c, PJB_ERR_BGZF, "truncated BGZF extra field at byte %lld", (long long)off);
        int64_t bsize = -1;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t *f = h + 12 + x;
            const uint32_t slen = f[2] | (uint32_t)f[3] << 8;
            if (f[0] == 'B' && f[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = (int64_t)(f[4] | (uint32_t)f[5] << 8) + 1;
            x += 4 + slen;
        }
        if (bsize < 0) return fail(c, PJB_ERR_BGZF, "BGZF block at byte %lld has no BC field", (long long)off);
        if (bsize < (int64_t)xlen + 20 || off + bsize > n)
            return fail(c, PJB_ERR_BGZF, "BGZF block at byte %lld has an impossible size %lld", (long long)off, (long long)bsize);
        const uint8_t *foot = comp + off + bsize - 8;
        const uint32_t isize = foot[4] | (uint32_t)foot[5] << 8 | (uint32_t)foot[6] << 16 | (uint32_t)foot[7] << 24;
        if (isize > 65536u) return fail(c, PJB_ERR_BGZF, "BGZF block at byte %lld declares %u inflated bytes", (long long)off, isize);
        InfBlock b;
        b.in_off = (iu64)(off + 12 + xlen);
        b.in_len = (iu32)(bsize - xlen - 20);
        b.out_off = (iu64)total_out;
        b.out_len = isize;
        blocks.push_back(b);
        total_out += isize;
        off += bsize;
    }
    return PJB_OK;
}

// pageable host memory -> device through the two page-locked staging buffers: a few threads memcpy a
// piece into one buffer while the DMA engine drains the other
int upload_staged(pjb_ctx *c, void *dst, const uint8_t *src, size_t bytes) {
    const size_t PIECE = (size_t)64 << 20;
    for (size_t off = 0; off < bytes; off += PIECE) {
        const size_t nb = std::min(PIECE, bytes - off);
        const unsigned si = c->stage_next++ & 1u;
        if (c->stage_busy[si]) {
            HIP_TRY(c, hipEventSynchronize(c->stage_ev[si]));
            c->stage_busy[si] = false;
        }
        if (c->stage_cap[si] < nb) {
            if (c->stage[si]) (void)hipHostFree(c->stage[si]);
            c->stage[si] = nullptr;
            c->stage_cap[si] = 0;
            if (hipHostMalloc((void **)&c->stage[si], PIECE, hipHostMallocDefault) != hipSuccess)
                return fail(c, PJB_ERR_NOMEM, "cannot allocate %zu bytes of page-locked staging memory", PIECE);
            c->stage_cap[si] = PIECE;
        }
        if (!c->stage_ev[si]) HIP_TRY(c, hipEventCreateWithFlags(&c->stage_ev[si], hipEventDisableTiming));
        parallel_copy(c->stage[si], src + off, nb);
        HIP_TRY(c, hipMemcpyAsync((uint8_t *)dst + off, c->stage[si], nb, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipEventRecord(c->stage_ev[si], c->stream));
        c->stage_busy[si] = true;
    }
    return PJB_OK;
}


// comp already on the device (padded); blocks on the host
// the status words of a finished bgzf_inflate (d_status[nb] = "some block failed")
int inflate_status(pjb_ctx *c, const std::vector<InfBlock> &blocks, const int *d_status) {
    const size_t nb = blocks.size();
    int any = 0;
    HIP_TRY(c, hipMemcpy(&any, d_status + nb, 4, hipMemcpyDeviceToHost));
    if (!any) return PJB_OK;
    std::vector<int> status(nb);
    HIP_TRY(c, hipMemcpy(status.data(), d_status, nb * 4, hipMemcpyDeviceToHost));
    for (size_t b = 0; b < nb; b++)
        if (status[b])
            return fail(c, PJB_ERR_BGZF, "BGZF block %zu (payload at byte %llu): %s", b, (unsigned long long)blocks[b].in_off, inf_text(status[b]));
    return fail(c, PJB_ERR_BGZF, "BGZF inflate failed");
}

int inflate_on_device(pjb_ctx *c, const uint8_t *d_comp, const std::vector<InfBlock> &blocks, uint8_t *d_out) {
    int rc;
    const size_t nb = blocks.size();
    if (nb == 0) return PJB_OK;
    if ((rc = ensure(c, c->b_inf_blocks, nb * sizeof(InfBlock)))) return rc;
    if ((rc = ensure(c, c->b_inf_status, nb * 4 + 16))) return rc;
    // one launch: as many lanes as the chip holds at once (two 64-lane workgroups per CU: the tables' LDS), each taking
    // block after block from a counter
    size_t lanes = std::min<size_t>((nb + 63) / 64 * 64, (size_t)c->inflate_lanes);
    if (const char *e = getenv("PJB_INF_BLOCKS_PER_LAUNCH")) lanes = std::min<size_t>((nb + 63) / 64 * 64, (size_t)std::max(64, atoi(e)) / 64 * 64); // tests: few lanes, long lists
    if ((rc = ensure(c, c->b_inf_scratch, lanes * INF_SCRATCH_PER_LANE))) return rc;
    hipStream_t st = c->stream;
    HIP_TRY(c, hipMemcpyAsync(c->b_inf_blocks.p, blocks.data(), nb * sizeof(InfBlock), hipMemcpyHostToDevice, st));
    int *d_status = (int *)c->b_inf_status.p;
    int *d_any = d_status + nb;
    iu32 *d_next = (iu32 *)(d_any + 1);
    const iu32 ctl[2] = {0u, (iu32)lanes};
    HIP_TRY(c, hipMemcpyAsync(d_any, ctl, 8, hipMemcpyHostToDevice, st));
    {
        // decode (lane per block: literals in place, a token + a bitmap bit per match), then the copies (wave per block)
        if ((rc = ensure(c, c->b_inf_bitmap, nb * INF_BITMAP_WORDS * 8))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->b_inf_bitmap.p, 0, nb * INF_BITMAP_WORDS * 8, st));
        LAUNCH_LDS(c, "bgzf_decode", bgzf_decode, dim3((unsigned)(lanes / 64)), dim3(64), I3_LDS_BYTES, d_comp, (const InfBlock *)c->b_inf_blocks.p, (iu32)nb,
                   d_out, (uint8_t *)c->b_inf_scratch.p, d_status, d_any, d_next, (iu64 *)c->b_inf_bitmap.p, 8);
        LAUNCH(c, "bgzf_resolve", bgzf_resolve, dim3((unsigned)((nb + 3) / 4)), dim3(256), (const InfBlock *)c->b_inf_blocks.p, (iu32)nb, d_out,
               (const iu64 *)c->b_inf_bitmap.p, (const int *)d_status);
    }
    HIP_TRY(c, hipStreamSynchronize(st));
    if (c->ktime) ev_collect(c, MISC_POOL);
    return inflate_status(c, blocks, d_status);
}

extern "C" int pjb_inflate_bgzf(pjb_ctx *c, const uint8_t *comp, int64_t comp_bytes, uint8_t *out, int64_t out_cap,
                                int64_t *out_bytes) {
    if (!c || !out_bytes || comp_bytes < 0 || (comp_bytes && !comp)) return fail(c, PJB_ERR_ARG, "inflate_bgzf: bad arguments");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    std::vector<InfBlock> blocks;
    int64_t total = 0;
    int rc = scan_bgzf(c, comp, comp_bytes, blocks, total);
    if (rc) return rc;
    *out_bytes = total;
    if (total > out_cap) return fail(c, PJB_ERR_ARG, "inflate_bgzf: output needs %lld bytes, capacity is %lld", (long long)total, (long long)out_cap);
    if (total == 0) return PJB_OK;
    if (!out) return fail(c, PJB_ERR_ARG, "inflate_bgzf: no output buffer");
    if ((rc = ensure(c, c->b_inf_comp, (size_t)comp_bytes + INF_PAD))) return rc;
    if ((rc = ensure(c, c->b_inf_out, (size_t)total + 64))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->b_inf_comp.p, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync((uint8_t *)c->b_inf_comp.p + comp_bytes, 0, INF_PAD, c->stream));
    if ((rc = inflate_on_device(c, (const uint8_t *)c->b_inf_comp.p, blocks, (uint8_t *)c->b_inf_out.p))) return rc;
    HIP_TRY(c, hipMemcpy(out, c->b_inf_out.p, (size_t)total, hipMemcpyDeviceToHost));
    return PJB_OK;
}

// BGZF deflate on the device (pjb_deflate.hip.h): at most DFL_LAUNCH_BLOCKS blocks per launch (1 GB of symbol scratch)
constexpr int64_t DFL_LAUNCH_BLOCKS = 4096;
extern "C" int pjb_deflate_bgzf(pjb_ctx *c, const uint8_t *in, int64_t n_bytes, int32_t block_bytes, uint8_t *out, int64_t out_cap, int64_t *out_bytes,
                                uint32_t *member_size) {
    if (!c || !out_bytes || n_bytes < 0 || (n_bytes && (!in || !out))) return fail(c, PJB_ERR_ARG, "deflate_bgzf: bad arguments");
    if (block_bytes < 4 || block_bytes > (int32_t)DFL_IN_MAX || (block_bytes & 3))
        return fail(c, PJB_ERR_ARG, "deflate_bgzf: block_bytes must be a multiple of 4 between 4 and %u", DFL_IN_MAX);
    *out_bytes = 0;
    if (n_bytes == 0) return PJB_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    const int64_t n_blocks = (n_bytes + block_bytes - 1) / block_bytes;
    int rc;
    std::vector<u32> sizes;
    std::vector<iu64> offs;
    int64_t written = 0;
    for (int64_t b0 = 0; b0 < n_blocks; b0 += DFL_LAUNCH_BLOCKS) {
        const int64_t nb = std::min<int64_t>(DFL_LAUNCH_BLOCKS, n_blocks - b0);
        const int64_t in_off = b0 * block_bytes, in_len = std::min<int64_t>(n_bytes - in_off, nb * block_bytes);
        if ((rc = ensure(c, c->b_dfl_in, (size_t)in_len + 64)) || (rc = ensure(c, c->b_dfl_sym, (size_t)nb * DFL_SYM_STRIDE * 4)) ||
            (rc = ensure(c, c->b_dfl_slots, (size_t)nb * DFL_SLOT)) || (rc = ensure(c, c->b_dfl_size, (size_t)nb * 4)) ||
            (rc = ensure(c, c->b_dfl_off, (size_t)nb * 8)))
            return rc;
        HIP_TRY(c, hipMemcpyAsync(c->b_dfl_in.p, in + in_off, (size_t)in_len, hipMemcpyHostToDevice, st));
        LAUNCH(c, "bgzf_deflate", bgzf_deflate, dim3((unsigned)nb), dim3(64), (const uint8_t *)c->b_dfl_in.p, (iu64)in_len, (u32)block_bytes, (u32)nb,
               (u32 *)c->b_dfl_sym.p, (uint8_t *)c->b_dfl_slots.p, (u32 *)c->b_dfl_size.p);
        sizes.resize((size_t)nb);
        HIP_TRY(c, hipMemcpyAsync(sizes.data(), c->b_dfl_size.p, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        offs.resize((size_t)nb);
        iu64 total = 0;
        for (int64_t k = 0; k < nb; k++) {
            if (sizes[(size_t)k] < 26 || sizes[(size_t)k] > 65536) return fail(c, PJB_ERR_STATE, "deflate_bgzf: block %lld came out with %u bytes", (long long)(b0 + k), sizes[(size_t)k]);
            offs[(size_t)k] = total;
            total += sizes[(size_t)k];
            if (member_size) member_size[b0 + k] = sizes[(size_t)k];
        }
        if (written + (int64_t)total > out_cap)
            return fail(c, PJB_ERR_ARG, "deflate_bgzf: the output needs more than %lld bytes", (long long)out_cap);
        if ((rc = ensure(c, c->b_dfl_packed, (size_t)total + 64))) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->b_dfl_off.p, offs.data(), (size_t)nb * 8, hipMemcpyHostToDevice, st));
        LAUNCH(c, "bgzf_pack", bgzf_pack, dim3((unsigned)nb), dim3(256), (const uint8_t *)c->b_dfl_slots.p, (const u32 *)c->b_dfl_size.p,
               (const iu64 *)c->b_dfl_off.p, (u32)nb, (uint8_t *)c->b_dfl_packed.p);
        HIP_TRY(c, hipMemcpyAsync(out + written, c->b_dfl_packed.p, (size_t)total, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        written += (int64_t)total;
    }
    *out_bytes = written;
    if (c->ktime) ev_collect(c, MISC_POOL);
    return PJB_OK;
}

static int ingest_parse(pjb_ctx *c, int32_t tid, OpenContig &oc, const uint8_t *d_out, size_t n_blocks, int64_t comp_bytes, int64_t total,
                        int32_t first_uoffset, int64_t *n_records, double t_scan, double t_up, double t_inf);

// the part of pjb_submit_bam behind the upload: `d_comp` holds the target's BGZF bytes (padded), `blocks` their layout
static int ingest_staged(pjb_ctx *c, int32_t tid, OpenContig &oc, const uint8_t *d_comp, const std::vector<InfBlock> &blocks, int64_t comp_bytes,
                         int64_t total, int32_t first_uoffset, int64_t *n_records, double t_scan, double t_up) {
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    int rc;
    if ((rc = ensure(c, c->b_inf_out, (size_t)total + 64))) return rc;
    HIP_TRY(c, hipMemsetAsync((uint8_t *)c->b_inf_out.p + total, 0, 64, c->stream));
    if ((rc = inflate_on_device(c, d_comp, blocks, (uint8_t *)c->b_inf_out.p))) return rc;
    return ingest_parse(c, tid, oc, (const uint8_t *)c->b_inf_out.p, blocks.size(), comp_bytes, total, first_uoffset, n_records, t_scan, t_up, now() - t0);
}

// the inflated bytes of one target's region (d_out, `total` of them followed by 64 zero bytes) -> the SoA batch of the target
static int ingest_parse(pjb_ctx *c, int32_t tid, OpenContig &oc, const uint8_t *d_out, size_t n_blocks, int64_t comp_bytes, int64_t total,
                        int32_t first_uoffset, int64_t *n_records, double t_scan, double t_up, double t_inf) {
    const bool prof = getenv("PJB_PROFILE_HOST") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now(), t_walk;
    hipStream_t st = c->stream;
    int rc;
    // ---- record boundaries
    BamRegion R;
    R.U = d_out;
    R.total = (iu64)total;
    R.first = (iu64)first_uoffset;
    R.tid = tid;
    R.ref_len = c->ref_len[(size_t)tid];
    R.n_ref = (int32_t)c->ref_len.size();
    const uint32_t n_seg = (uint32_t)(((iu64)total + BAM_SEG - 1) / BAM_SEG);
    // seg_start u64 | seg_base u64 | land u64 | seg_n u32
    if ((rc = ensure(c, c->b_bam_seg, (size_t)n_seg * 28 + 64))) return rc;
    if ((rc = ensure(c, c->b_bam_ctl, 64))) return rc;
    iu64 *seg_start = (iu64 *)c->b_bam_seg.p;
    iu64 *seg_base = seg_start + n_seg;
    iu64 *land = seg_base + n_seg;
    iu32 *seg_n = (iu32 *)(land + n_seg);
    iu32 *ctl = (iu32 *)c->b_bam_ctl.p; // [0..2] end / mismatch / bad segment, [4..5] u64 total of a scan
    iu64 *d_total = (iu64 *)(ctl + 4);
    HIP_TRY(c, hipMemsetAsync(ctl, 0xff, 16, st));
    BamWalkOut O;
    O.seg_n = seg_n;
    O.land = land;
    O.rec_off = nullptr;
    O.seg_base = seg_base;
    O.ctl = ctl;
    LAUNCH(c, "bam_find_starts", bam_find_starts, dim3(n_seg), dim3(64), R, n_seg, seg_start);
    if (const char *e = getenv("PJB_TEST_FALSE_START")) { // test hook: damage the guessed start of one segment
        const uint32_t k = (uint32_t)atoi(e);
        if (k > 0 && k < n_seg) {
            iu64 v = 0;
            HIP_TRY(c, hipMemcpyAsync(&v, seg_start + k, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipStreamSynchronize(st));
            if (v != BAM_NONE) {
                v += 1;
                HIP_TRY(c, hipMemcpyAsync(seg_start + k, &v, 8, hipMemcpyHostToDevice, st));
                HIP_TRY(c, hipStreamSynchronize(st));
            }
        }
    }
    uint32_t h_ctl[8];
    uint32_t end_seg = 0xffffffffu;
    for (int attempt = 0;; attempt++) {
        HIP_TRY(c, hipMemsetAsync(ctl, 0xff, 16, st));
        HIP_TRY(c, hipMemsetAsync(ctl + 6, 0xff, 4, st));
        LAUNCH(c, "bam_walk_count", bam_walk<false>, dim3((n_seg + 255) / 256), dim3(256), R, n_seg, (const iu64 *)seg_start, O);
        HIP_TRY(c, hipMemcpyAsync(h_ctl, ctl, 32, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        end_seg = h_ctl[0];
        if (h_ctl[2] != 0xffffffffu && h_ctl[2] <= end_seg && (h_ctl[1] == 0xffffffffu || h_ctl[2] <= h_ctl[1]))
            return fail(c, PJB_ERR_BGZF, "Invalid BAM record layout on target %d (inflated offset %llu..)", tid,
                        (unsigned long long)h_ctl[2] * BAM_SEG);
        if (h_ctl[1] == 0xffffffffu || h_ctl[1] > end_seg) break; // every walk landed on the next start
        // a guessed start was not a record boundary: replace it by the boundary the verified walk reached, walk again
        if (attempt >= 16)
            return fail(c, PJB_ERR_BGZF, "BAM record chain of target %d is inconsistent near inflated offset %llu", tid,
                        (unsigned long long)h_ctl[1] * BAM_SEG);
        LAUNCH(c, "bam_repair_start", bam_repair_start, dim3(1), dim3(1), seg_start, n_seg, h_ctl[1], (const iu64 *)land, (iu64)total, ctl);
        HIP_TRY(c, hipMemcpyAsync(h_ctl, ctl, 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if (h_ctl[3] != 0xffffffffu)
            return fail(c, PJB_ERR_BGZF, "Invalid BAM record on target %d (inflated offset %llu..)", tid, (unsigned long long)h_ctl[3] * BAM_SEG);
    }
    // The data ends inside a record of this target and no record of another target (or past the target's end) was seen:
    // the bytes handed over stop short of the target's last alignment (a stale index, a truncated file).  The reference
    // fails on a truncated file too (bgzf_read / bam_read1); dropping the tail silently would change counts.
    if (end_seg == 0xffffffffu && h_ctl[6] != 0xffffffffu)
        return fail(c, PJB_ERR_BGZF, "the data for target %d ends inside an alignment record (inflated offset %llu..): truncated "
                                     "file, or the index's span for the target is too short", tid, (unsigned long long)h_ctl[6] * BAM_SEG);
    LAUNCH(c, "bam_trim_segments", bam_trim_segments, dim3((n_seg + 255) / 256), dim3(256), seg_n, n_seg, (const iu32 *)ctl);
    if ((rc = run_scan(c, "bam_seg", SegCountFn{seg_n}, SegBaseSink{seg_base}, n_seg, d_total))) return rc;
    HIP_TRY(c, hipMemcpyAsync(h_ctl, ctl, 24, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    t_walk = now() - t0;
    t0 = now();
    iu64 n64;
    memcpy(&n64, &h_ctl[4], 8);
    if (n64 == 0) return PJB_OK;
    if (n64 >= 0xffffff00ull) return fail(c, PJB_ERR_ARG, "submit_bam: more than 2^32 alignments on one target are not supported");
    const size_t n = (size_t)n64;
    if ((rc = ensure(c, c->b_bam_rec, n * 8))) return rc;
    O.rec_off = (iu64 *)c->b_bam_rec.p;
    LAUNCH(c, "bam_walk_fill", bam_walk<true>, dim3((n_seg + 255) / 256), dim3(256), R, n_seg, (const iu64 *)seg_start, O);

    // ---- SoA arrays in the target's slabs (same packing as a host-submitted batch)
    // (150-base paired-end records: fields + operations + 4- and 2-bit bases of the spliced third are 0.28 of the inflated bytes)
    if (oc.slabs.empty()) oc.slab_hint = (((size_t)total / 100 * 32) + ((size_t)32 << 20)) & ~(((size_t)1 << 20) - 1);
    const size_t fixed[8] = {n * 4, n * 2, n, n, n * 4, n * 4, n * 4, (n + 1) * 4}; // pos flag mapq xs l_qseq mtid mpos cig_off
    size_t offs[9], tot_b = 0;
    for (int k = 0; k < 8; k++) {
        offs[k] = tot_b;
        tot_b += (std::max<size_t>(fixed[k], 16) + 255) & ~(size_t)255;
    }
    offs[8] = tot_b; // seq_off
    tot_b += (((n + 1) * 4) + 255) & ~(size_t)255;
    uint8_t *dev = (uint8_t *)slab_alloc(c, oc, tot_b);
    if (!dev) return fail(c, PJB_ERR_NOMEM, "submit_bam: out of device memory for %zu alignments", n);
    BamSoA B;
    B.pos = (int32_t *)(dev + offs[0]);
    B.flag = (uint16_t *)(dev + offs[1]);
    B.mapq = dev + offs[2];
    B.xs = dev + offs[3];
    B.l_qseq = (int32_t *)(dev + offs[4]);
    B.mtid = (int32_t *)(dev + offs[5]);
    B.mpos = (int32_t *)(dev + offs[6]);
    B.cig_off = (iu32 *)(dev + offs[7]);
    B.seq_off = (iu32 *)(dev + offs[8]);
    B.cigar = nullptr;
    B.seq4 = nullptr;
    B.name_hash = nullptr;
    B.seq2 = nullptr;
    B.seq_exc = nullptr;
    if (c->extra) {
        B.name_hash = (iu64 *)slab_alloc(c, oc, n * 8 + 16);
        if (!B.name_hash) return fail(c, PJB_ERR_NOMEM, "submit_bam: out of device memory for name codes");
    }
    if ((rc = run_scan(c, "bam_sizes", BamSizesFn{R.U, (const iu64 *)c->b_bam_rec.p}, BamOffsetsSink{B.cig_off, B.seq_off}, n, d_total)))
        return rc;
    iu64 tot = 0;
    HIP_TRY(c, hipMemcpyAsync(&tot, d_total, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    const uint32_t n_ops = (uint32_t)(tot >> 32), n_words = (uint32_t)tot;
    // (a carry out of the low half would mean 2^32 sequence words: 16 GB of bases on one target)
    const uint32_t tails[2] = {n_ops, n_words};
    HIP_TRY(c, hipMemcpyAsync(B.cig_off + n, &tails[0], 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(B.seq_off + n, &tails[1], 4, hipMemcpyHostToDevice, st));
    B.cigar = (iu32 *)slab_alloc(c, oc, (size_t)n_ops * 4 + 16);
    B.seq4 = (uint8_t *)slab_alloc(c, oc, (size_t)n_words * 4 + 16);
    if (!B.cigar || !B.seq4) return fail(c, PJB_ERR_NOMEM, "submit_bam: out of device memory for CIGARs / bases");
    static const bool no_seq2 = getenv("PJB_NO_SEQ2") && atoi(getenv("PJB_NO_SEQ2")) != 0;
    if (!no_seq2) { // the bases in 2 bits as well (what pjb_batch.seq2 / .seq_exc hold): written where the 4-bit bases are
        B.seq2 = (unsigned short *)slab_alloc(c, oc, ((size_t)n_words + 2) * 2 + 16);
        B.seq_exc = (iu32 *)slab_alloc(c, oc, ((n + 31) / 32) * 4 + 16);
        if (!B.seq2 || !B.seq_exc) return fail(c, PJB_ERR_NOMEM, "submit_bam: out of device memory for the 2-bit bases");
    }
    LAUNCH(c, "bam_transcode", bam_transcode, dim3((unsigned)((n + 255) / 256)), dim3(256), R.U, (const iu64 *)c->b_bam_rec.p, (iu64)n, B);
    HIP_TRY(c, hipStreamSynchronize(st)); // `tails` is on this stack frame
    if (c->ktime) ev_collect(c, MISC_POOL);
    if (prof)
        fprintf(stderr, "[host profile] submit_bam tid %d: %zu blocks, %.1f MB -> %.1f MB, %zu records: header scan %.3f, upload %.3f, inflate %.3f, "
                        "boundaries %.3f, fill+sizes+transcode %.3f s\n",
                tid, n_blocks, comp_bytes / 1e6, total / 1e6, n, t_scan, t_up, t_inf, t_walk, now() - t0);
    DevBatch d;
    memset(&d, 0, sizeof d);
    d.n = (int64_t)n;
    d.base = 0;
    d.pos = B.pos; d.flag = B.flag; d.mapq = B.mapq; d.xs = B.xs; d.l_qseq = B.l_qseq; d.mtid = B.mtid; d.mpos = B.mpos;
    d.cig_off = B.cig_off; d.cigar = B.cigar; d.seq_off = B.seq_off; d.seq4 = B.seq4;
    d.name_hash = (const u64 *)B.name_hash;
    d.seq2 = (const uint32_t *)B.seq2;
    d.seq_exc = B.seq_exc;
    oc.on_main_stream = true;
    oc.batches.push_back(d);
    oc.last_known.push_back(0);
    oc.last_pos.push_back(INT32_MIN);
    if (n_records) *n_records = (int64_t)n;
    return PJB_OK;
}

extern "C" int pjb_submit_bam(pjb_ctx *c, int32_t tid, const uint8_t *comp, int64_t comp_bytes, int32_t first_uoffset,
                              int64_t *n_records) {
    if (!c) return PJB_ERR_ARG;
    if (n_records) *n_records = 0;
    if (comp_bytes < 0 || (comp_bytes && !comp) || first_uoffset < 0) return fail(c, PJB_ERR_ARG, "submit_bam: bad arguments");
    if (tid < 0 || (size_t)tid >= c->ref_len.size()) return fail(c, PJB_ERR_ARG, "submit_bam: bad tid %d", tid);
    c->cur_tid = tid;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    OpenContig &oc = c->open[tid];
    if (!oc.batches.empty()) return fail(c, PJB_ERR_STATE, "submit_bam: target %d already has batches (one call per target)", tid);
    const bool prof = getenv("PJB_PROFILE_HOST") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now(), t_scan, t_up;
    std::vector<InfBlock> blocks;
    int64_t total = 0;
    int rc = scan_bgzf(c, comp, comp_bytes, blocks, total);
    if (rc) return rc;
    if (total == 0 || (int64_t)first_uoffset >= total) return PJB_OK;
    t_scan = now() - t0;
    t0 = now();
    hipStream_t st = c->stream;
    if ((rc = ensure(c, c->b_inf_comp, (size_t)comp_bytes + INF_PAD))) return rc;
    {
        // page-locked input (pjb_host_alloc): one DMA, no staging copy
        hipPointerAttribute_t at;
        const bool pinned = hipPointerGetAttributes(&at, comp) == hipSuccess && at.type == hipMemoryTypeHost;
        if (!pinned) (void)hipGetLastError();
        if (pinned) HIP_TRY(c, hipMemcpyAsync(c->b_inf_comp.p, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, st));
        else if ((rc = upload_staged(c, c->b_inf_comp.p, comp, (size_t)comp_bytes))) return rc;
    }
    HIP_TRY(c, hipMemsetAsync((uint8_t *)c->b_inf_comp.p + comp_bytes, 0, INF_PAD, st));
    if (prof) (void)hipStreamSynchronize(st);
    t_up = now() - t0;
    return ingest_staged(c, tid, oc, (const uint8_t *)c->b_inf_comp.p, blocks, comp_bytes, total, first_uoffset, n_records, t_scan, t_up);
}

// ---- the same in pieces -----------------------------------------------------------------------------------------
// (pjb_bam_begin / pjb_bam_piece / pjb_bam_pieces_done / pjb_bam_end, see the header)
struct BamStage {
    Buf dev;                 // the target's BGZF bytes on the device (from the context's pool)
    int64_t total = 0, got = 0;
    std::vector<InfBlock> blocks;
    int64_t total_out = 0;
    int64_t next = 0;        // file-relative offset of the next block header to look at
    uint8_t keep[65536 + 64]; // bytes [keep_at, got) of what arrived, for a block whose header or footer straddles two pieces
    int64_t keep_at = 0, keep_n = 0;
    double t_scan = 0, t_up = 0;
    // the inflate launched at the last piece (launched: ev_inf follows the kernel on its stream)
    bool launched = false;
    iu32 ctl[2] = {0, 0}; // { "some block failed", lanes }: copied to the device asynchronously, so it lives here and not on a stack
    Buf out, d_blocks, d_status, d_scratch, d_bitmap;
    hipEvent_t ev_inf = nullptr, ev_last = nullptr;
};

// a buffer of at least `bytes` from a pool (the smallest that fits), else a new one
static int pool_take(pjb_ctx *c, std::vector<Buf> &pool, Buf &b, size_t bytes) {
    int best = -1;
    for (size_t k = 0; k < pool.size(); k++)
        if (pool[k].cap >= bytes && (best < 0 || pool[k].cap < pool[(size_t)best].cap)) best = (int)k;
    if (best >= 0) {
        b = pool[(size_t)best];
        pool.erase(pool.begin() + best);
        return PJB_OK;
    }
    return ensure(c, b, bytes);
}
static void pool_give(std::vector<Buf> &pool, Buf &b) {
    if (b.p) pool.push_back(b);
    b.p = nullptr;
    b.cap = 0;
}
static void stage_release(pjb_ctx *c, BamStage &st) { // (after the work that uses the buffers has completed)
    pool_give(c->stage_pool, st.dev);
    pool_give(c->out_pool, st.out);
    pool_give(c->misc_pool, st.d_blocks);
    pool_give(c->misc_pool, st.d_status);
    pool_give(c->misc_pool, st.d_scratch);
    pool_give(c->out_pool, st.d_bitmap); // (output-sized: an eighth of the inflated bytes)
    if (st.ev_inf) (void)hipEventDestroy(st.ev_inf);
    if (st.ev_last) (void)hipEventDestroy(st.ev_last);
    st.ev_inf = st.ev_last = nullptr;
}

// every byte of the target has been queued for copying and every block header seen: inflate on a stream of its own, behind
// the last copy.  Nothing here waits; a failure just leaves the inflate to pjb_bam_end.
static void inflate_early(pjb_ctx *c, BamStage &st) {
    const size_t nb = st.blocks.size();
    if (st.launched || nb == 0 || st.total_out <= 0) return;
    size_t lanes = std::min<size_t>((nb + 63) / 64 * 64, (size_t)c->inflate_lanes);
    if (const char *e = getenv("PJB_INF_BLOCKS_PER_LAUNCH")) lanes = std::min<size_t>((nb + 63) / 64 * 64, (size_t)std::max(64, atoi(e)) / 64 * 64);
    if (pool_take(c, c->out_pool, st.out, (size_t)st.total_out + 64) || pool_take(c, c->misc_pool, st.d_blocks, nb * sizeof(InfBlock)) ||
        pool_take(c, c->misc_pool, st.d_status, nb * 4 + 16) || pool_take(c, c->misc_pool, st.d_scratch, lanes * INF_SCRATCH_PER_LANE) ||
        pool_take(c, c->out_pool, st.d_bitmap, nb * INF_BITMAP_WORDS * 8)) {
        std::lock_guard<std::mutex> lk(c->err_mu); // (a failure here just leaves the inflate to pjb_bam_end)
        c->err.clear();
        return;
    }
    hipStream_t &is = c->inf_streams[c->inf_next++ & 3u];
    if (!is) {
        // The inflate streams have the lowest priority: a launch holds every LDS byte of the chip for ~50 ms, and the short
        // kernels beside it -- record parsing, genome uploads, the junc chains of the targets before it -- are what the one
        // host thread that serves all targets waits for (end to end 2.73 -> 2.44 s)
        int lo = 0, hi = 0; // (least, greatest priority)
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&is, hipStreamNonBlocking, lo) != hipSuccess) return;
    }
    if (hipEventCreateWithFlags(&st.ev_last, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&st.ev_inf, hipEventDisableTiming) != hipSuccess) return;
    int *d_status = (int *)st.d_status.p;
    int *d_any = d_status + nb;
    iu32 *d_next = (iu32 *)(d_any + 1);
    st.ctl[0] = 0u;
    st.ctl[1] = (iu32)lanes;
    bool ok = hipEventRecord(st.ev_last, c->stream_up) == hipSuccess && hipStreamWaitEvent(is, st.ev_last, 0) == hipSuccess &&
              hipMemsetAsync((uint8_t *)st.out.p + st.total_out, 0, 64, is) == hipSuccess &&
              hipMemcpyAsync(st.d_blocks.p, st.blocks.data(), nb * sizeof(InfBlock), hipMemcpyHostToDevice, is) == hipSuccess &&
              hipMemcpyAsync(d_any, st.ctl, 8, hipMemcpyHostToDevice, is) == hipSuccess;
    if (ok) {
        ok = hipMemsetAsync(st.d_bitmap.p, 0, nb * INF_BITMAP_WORDS * 8, is) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(bgzf_decode, dim3((unsigned)(lanes / 64)), dim3(64), I3_LDS_BYTES, is, (const uint8_t *)st.dev.p, (const InfBlock *)st.d_blocks.p,
                               (iu32)nb, (uint8_t *)st.out.p, (uint8_t *)st.d_scratch.p, d_status, d_any, d_next, (iu64 *)st.d_bitmap.p, 8);
            hipLaunchKernelGGL(bgzf_resolve, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, is, (const InfBlock *)st.d_blocks.p, (iu32)nb, (uint8_t *)st.out.p,
                               (const iu64 *)st.d_bitmap.p, (const int *)d_status);
            ok = hipGetLastError() == hipSuccess && hipEventRecord(st.ev_inf, is) == hipSuccess;
        }
    }
    if (!ok) { // whatever was queued must be over before the buffers are used again
        (void)hipStreamSynchronize(is);
        (void)hipGetLastError();
        return;
    }
    st.launched = true;
}

// block headers that are complete with the bytes received so far (the last `avail` bytes of the stream are at `p`, the
// first of them is byte `p_at` of the target's bytes); leaves st.next at the first block it cannot finish yet
static int stage_scan(pjb_ctx *c, BamStage &st, const uint8_t *p, int64_t p_at, int64_t avail) {
    auto byte_at = [&](int64_t off) -> int { // a byte of the stream that is still in reach (this piece or the kept tail)
        if (off >= p_at && off < p_at + avail) return p[off - p_at];
        if (off >= st.keep_at && off < st.keep_at + st.keep_n) return st.keep[off - st.keep_at];
        return -1;
    };
    const int64_t end = p_at + avail;
    while (st.next < st.total) {
        const int64_t off = st.next;
        if (off + 18 > end) break;
        uint8_t h[18];
        for (int k = 0; k < 18; k++) {
            const int v = byte_at(off + k);
            if (v < 0) return fail(c, PJB_ERR_STATE, "bam_piece: internal: header byte %lld out of reach", (long long)(off + k));
            h[k] = (uint8_t)v;
        }
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return fail(c, PJB_ERR_BGZF, "not a BGZF block header at byte %lld", (long long)off);
        const uint32_t xlen = h[10] | (uint32_t)h[11] << 8;
        if (off + 12 + xlen > end) break;
        int64_t bsize = -1;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            int f[6];
            for (int k = 0; k < 6; k++) f[k] = x + (uint32_t)k < xlen ? byte_at(off + 12 + x + k) : 0;
            if (f[0] < 0 || f[1] < 0 || f[2] < 0 || f[3] < 0) return fail(c, PJB_ERR_STATE, "bam_piece: internal: extra field out of reach");
            const uint32_t slen = (uint32_t)f[2] | (uint32_t)f[3] << 8;
            if (f[0] == 'B' && f[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = (int64_t)((uint32_t)f[4] | (uint32_t)f[5] << 8) + 1;
            x += 4 + slen;
        }
        if (bsize < 0) return fail(c, PJB_ERR_BGZF, "BGZF block at byte %lld has no BC field", (long long)off);
        if (bsize < (int64_t)xlen + 20 || off + bsize > st.total)
            return fail(c, PJB_ERR_BGZF, "BGZF block at byte %lld has an impossible size %lld", (long long)off, (long long)bsize);
        if (off + bsize > end) break; // its footer has not arrived
        uint32_t isize = 0;
        for (int k = 0; k < 4; k++) {
            const int v = byte_at(off + bsize - 4 + k);
            if (v < 0) return fail(c, PJB_ERR_STATE, "bam_piece: internal: footer byte out of reach");
            isize |= (uint32_t)v << (8 * k);
        }
        if (isize > 65536u) return fail(c, PJB_ERR_BGZF, "BGZF block at byte %lld declares %u inflated bytes", (long long)off, isize);
        InfBlock b;
        b.in_off = (iu64)(off + 12 + xlen);
        b.in_len = (iu32)(bsize - xlen - 20);
        b.out_off = (iu64)st.total_out;
        b.out_len = isize;
        st.blocks.push_back(b);
        st.total_out += isize;
        st.next = off + bsize;
    }
    return PJB_OK;
}

void bam_stage_clear(pjb_ctx *c) {
    for (auto &is : c->inf_streams)
        if (is) (void)hipStreamSynchronize(is);
    for (auto &kv : c->bam_stage) {
        stage_release(c, *kv.second);
        delete kv.second;
    }
    c->bam_stage.clear();
    for (auto *pool : {&c->stage_pool, &c->out_pool, &c->misc_pool}) {
        for (auto &b : *pool) release(b);
        pool->clear();
    }
    for (auto &is : c->inf_streams)
        if (is) (void)hipStreamDestroy(is);
    for (auto &ev : c->up_events)
        if (ev) (void)hipEventDestroy(ev);
    if (c->ev_up) (void)hipEventDestroy(c->ev_up);
    if (c->stream_up) (void)hipStreamDestroy(c->stream_up);
}

extern "C" int pjb_bam_begin(pjb_ctx *c, int32_t tid, int64_t total_bytes) {
    if (!c) return PJB_ERR_ARG;
    if (tid < 0 || (size_t)tid >= c->ref_len.size() || total_bytes <= 0) return fail(c, PJB_ERR_ARG, "bam_begin: bad arguments (tid %d)", tid);
    std::lock_guard<std::mutex> lk(c->bam_mu);
    if (c->bam_stage.count(tid)) return fail(c, PJB_ERR_STATE, "bam_begin: target %d is being staged already", tid);
    // (that the target has no batches yet is checked by pjb_bam_end, on the thread that owns the open targets)
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    std::unique_ptr<BamStage> st(new (std::nothrow) BamStage());
    if (!st) return fail(c, PJB_ERR_NOMEM, "bam_begin: out of host memory");
    // device buffer: the smallest free one of the pool that fits, else a new one
    const size_t need = (size_t)total_bytes + INF_PAD;
    int best = -1;
    for (size_t k = 0; k < c->stage_pool.size(); k++)
        if (c->stage_pool[k].cap >= need && (best < 0 || c->stage_pool[k].cap < c->stage_pool[(size_t)best].cap)) best = (int)k;
    if (best >= 0) {
        st->dev = c->stage_pool[(size_t)best];
        c->stage_pool.erase(c->stage_pool.begin() + best);
    } else {
        int rc = ensure(c, st->dev, need);
        if (rc) return rc;
    }
    st->total = total_bytes;
    hipError_t he = hipSuccess;
    if (!c->stream_up) he = hipStreamCreateWithFlags(&c->stream_up, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipMemsetAsync((uint8_t *)st->dev.p + total_bytes, 0, INF_PAD, c->stream_up);
    if (he != hipSuccess) {
        pool_give(c->stage_pool, st->dev);
        return fail(c, PJB_ERR_HIP, "bam_begin: %s", hipGetErrorString(he));
    }
    c->bam_stage[tid] = st.release();
    return PJB_OK;
}

static int bam_piece_body(pjb_ctx *c, int32_t tid, BamStage &st, const uint8_t *piece, int64_t bytes, int64_t *ticket);

extern "C" int pjb_bam_piece(pjb_ctx *c, int32_t tid, const uint8_t *piece, int64_t bytes, int64_t *ticket) {
    if (!c || !piece || bytes <= 0) return fail(c, PJB_ERR_ARG, "bam_piece: bad arguments");
    std::lock_guard<std::mutex> lk(c->bam_mu);
    auto it = c->bam_stage.find(tid);
    if (it == c->bam_stage.end()) return fail(c, PJB_ERR_STATE, "bam_piece: target %d was not begun (pjb_bam_begin)", tid);
    const int rc = bam_piece_body(c, tid, *it->second, piece, bytes, ticket);
    if (rc) { // the target's staging is dropped: it has to be begun again
        (void)hipStreamSynchronize(c->stream_up);
        for (auto &is : c->inf_streams)
            if (is) (void)hipStreamSynchronize(is);
        stage_release(c, *it->second);
        delete it->second;
        c->bam_stage.erase(it);
    }
    return rc;
}

static int bam_piece_body(pjb_ctx *c, int32_t tid, BamStage &st, const uint8_t *piece, int64_t bytes, int64_t *ticket) {
    if (st.got + bytes > st.total) return fail(c, PJB_ERR_ARG, "bam_piece: target %d: more bytes than announced", tid);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    // the copy first (asynchronous, on the upload stream), the header hop meanwhile
    HIP_TRY(c, hipMemcpyAsync((uint8_t *)st.dev.p + st.got, piece, (size_t)bytes, hipMemcpyHostToDevice, c->stream_up));
    const int64_t tk = ++c->up_ticket;
    hipEvent_t &ev = c->up_events[(size_t)(tk % (int64_t)PJB_UP_EVENTS)];
    if (!ev) HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    else if (tk - c->up_done >= (int64_t)PJB_UP_EVENTS) { // the ring is full: its oldest copy must have completed
        HIP_TRY(c, hipEventSynchronize(ev));
        c->up_done = std::max<int64_t>(c->up_done, tk - (int64_t)PJB_UP_EVENTS);
    }
    HIP_TRY(c, hipEventRecord(ev, c->stream_up));
    st.t_up += now() - t0;
    t0 = now();
    int rc = stage_scan(c, st, piece, st.got, bytes);
    if (rc) return rc;
    st.got += bytes;
    // keep what the next piece's first block may still need: everything from st.next on, if it is short (a block is
    // at most 64 KB), else nothing (the block then starts in a later piece)
    if (st.next < st.got) {
        const int64_t from = st.next;
        const int64_t n = st.got - from;
        if (n > (int64_t)sizeof st.keep) return fail(c, PJB_ERR_BGZF, "BGZF block at byte %lld is longer than 64 KB", (long long)from);
        uint8_t tmp[sizeof st.keep];
        for (int64_t k = 0; k < n; k++) {
            const int64_t off = from + k;
            tmp[k] = off >= st.got - bytes ? piece[off - (st.got - bytes)] : st.keep[off - st.keep_at];
        }
        memcpy(st.keep, tmp, (size_t)n);
        st.keep_at = from;
        st.keep_n = n;
    } else
        st.keep_n = 0;
    st.t_scan += now() - t0;
    if (ticket) *ticket = tk;
    if (st.got == st.total && st.next == st.total) inflate_early(c, st);
    return PJB_OK;
}

extern "C" int pjb_bam_inflate_done(pjb_ctx *c, int32_t tid) {
    if (!c) return 1;
    std::lock_guard<std::mutex> lk(c->bam_mu);
    auto it = c->bam_stage.find(tid);
    if (it == c->bam_stage.end() || !it->second->launched) return 1; // (nothing in flight: pjb_bam_end does all the work)
    const bool done = hipEventQuery(it->second->ev_inf) == hipSuccess;
    (void)hipGetLastError();
    return done ? 1 : 0;
}

extern "C" int pjb_bam_pieces_done(pjb_ctx *c, int64_t *completed_ticket) {
    if (!c || !completed_ticket) return PJB_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->bam_mu);
    while (c->up_done < c->up_ticket) {
        hipEvent_t ev = c->up_events[(size_t)((c->up_done + 1) % (int64_t)PJB_UP_EVENTS)];
        if (!ev || hipEventQuery(ev) != hipSuccess) break;
        c->up_done++;
    }
    (void)hipGetLastError(); // (hipErrorNotReady is not an error here)
    *completed_ticket = c->up_done;
    return PJB_OK;
}

extern "C" int pjb_bam_end(pjb_ctx *c, int32_t tid, int32_t first_uoffset, int64_t *n_records) {
    if (!c) return PJB_ERR_ARG;
    if (n_records) *n_records = 0;
    std::unique_ptr<BamStage> st;
    {
        std::lock_guard<std::mutex> lk(c->bam_mu);
        auto it = c->bam_stage.find(tid);
        if (it == c->bam_stage.end()) return fail(c, PJB_ERR_STATE, "bam_end: target %d was not begun (pjb_bam_begin)", tid);
        st.reset(it->second);
        c->bam_stage.erase(it);
    }
    struct Return { // the device buffer goes back to the pool whatever happens (after the work that reads it)
        pjb_ctx *c;
        BamStage *st;
        ~Return() {
            (void)hipStreamSynchronize(c->stream);
            if (st->launched) (void)hipEventSynchronize(st->ev_inf); // (the inflate waited for the target's last copy)
            else if (c->stream_up) (void)hipStreamSynchronize(c->stream_up); // early returns: the copies may still read the caller's buffers
            std::lock_guard<std::mutex> lk(c->bam_mu);
            stage_release(c, *st);
        }
    } ret{c, st.get()};
    c->cur_tid = tid;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (first_uoffset < 0) return fail(c, PJB_ERR_ARG, "bam_end: bad first_uoffset");
    if (st->got != st->total) return fail(c, PJB_ERR_ARG, "bam_end: target %d: %lld of %lld bytes arrived", tid, (long long)st->got, (long long)st->total);
    if (st->next != st->total) return fail(c, PJB_ERR_BGZF, "truncated BGZF block at byte %lld", (long long)st->next);
    OpenContig &oc = c->open[tid];
    if (!oc.batches.empty()) return fail(c, PJB_ERR_STATE, "bam_end: target %d already has batches (one call per target)", tid);
    if (st->total_out == 0 || (int64_t)first_uoffset >= st->total_out) return PJB_OK;
    if (st->launched) { // the inflate started with the last piece: wait for it, look at its status words, go on with the records
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = now();
        HIP_TRY(c, hipEventSynchronize(st->ev_inf));
        int rc = inflate_status(c, st->blocks, (const int *)st->d_status.p);
        if (rc) return rc;
        return ingest_parse(c, tid, oc, (const uint8_t *)st->out.p, st->blocks.size(), st->total, st->total_out, first_uoffset, n_records, st->t_scan, st->t_up,
                            now() - t0);
    }
    // the service stream picks up behind the last copy
    if (!c->ev_up) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->ev_up, c->stream_up));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_up, 0));
    return ingest_staged(c, tid, oc, (const uint8_t *)st->dev.p, st->blocks, st->total, st->total_out, first_uoffset, n_records, st->t_scan, st->t_up);
}

