// pjb_extra.hip.h -- `junc --extra` on the device: the three metric families the
// reference computes after the main pass, single-threaded, by re-reading BAM files it has just split off
// (src/junction_builder.cc:152-226,293-312):
//   mm_score          Junction::calcMultipleMappingScore              lib/src/junction.cc:914-921
//   up_aln, down_aln  Junction::processJunctionVicinity               lib/src/junction.cc:651-677
//   coverage          DepthParser::loadNextBatch + Junction::calcCoverage
//                                                                     lib/src/depth_parser.cc:112-164, junction.cc:923-951
// Here they come from the alignment records that are already in HBM for the main pass:
//   * "unspliced.bam" = records without an N operation that are mapped (junction_builder.cc:168-186);
//   * per-base depth = difference array over their M / = / X runs + one scan (replaces htslib's pileup,
//     deps/htslib-1.3/sam.c:1853-1975, including its 8000-record cap -- see kx_cap_*);
//   * flanking counts = two rank queries: records starting before the intron minus records ending before
//     the left anchor; records starting inside the right anchor;
//   * name multiplicities = one open-addressing table over the 64-bit codes of every spliced record.
#pragma once

#include "pjb_kernels.hip.h"

namespace pjb {

// ---- std::hash<std::string> (libstdc++ _Hash_bytes, 64-bit: a MurmurHash64A variant, seed 0xc70f6907) of
// BamAlignment::deriveName() (lib/src/bam_alignment.cc:233-242; lib/include/portcullis/junction.hpp:158).
// `name` has `len` bytes without the NUL.  Used by the device record transcoder; the host transcoder has the
// same function (portcullis/bam/name_hash.hpp).
__host__ __device__ inline u64 std_hash_shift_mix(u64 v) { return v ^ (v >> 47); }
__host__ __device__ inline u64 derive_name_hash(const uint8_t *name, u32 len, u32 flag) {
    const u64 mul = (((u64)0xc6a4a793UL) << 32) + (u64)0x5bd1e995UL;
    uint8_t suf[3] = {'_', 'R', '?'};
    u32 total = len;
    if (flag & 0x1u) {
        suf[2] = (flag & 0x40u) ? '1' : (flag & 0x80u) ? '2' : '?';
        total += 3;
    }
    auto at = [&](u32 i) -> u64 { return i < len ? name[i] : suf[i - len]; };
    u64 hash = 0xc70f6907ULL ^ ((u64)total * mul);
    const u32 aligned = total & ~7u;
    for (u32 p = 0; p < aligned; p += 8) {
        u64 w = 0;
        for (int k = 7; k >= 0; k--) w = (w << 8) | at(p + (u32)k); // little-endian unaligned load
        const u64 data = std_hash_shift_mix(w * mul) * mul;
        hash ^= data;
        hash *= mul;
    }
    if (total & 7u) {
        u64 data = 0;
        for (int n = (int)(total & 7u) - 1; n >= 0; n--) data = (data << 8) + at(aligned + (u32)n);
        hash ^= data;
        hash *= mul;
    }
    hash = std_hash_shift_mix(hash) * mul;
    hash = std_hash_shift_mix(hash);
    return hash;
}

constexpr u32 PLP_MAXCNT = 8000; // bam_plp_init, deps/htslib-1.3/sam.c:1622

struct ExtraCounters { // one per contig, device memory
    u32 n_zero;       // unspliced mapped records with no reference-consuming op (zlist entries)
    u32 n_spliced;    // spliced records appended to the name-code list
    u32 max_buffered; // max over unspliced records of the pileup's buffered-record upper bound (cap detection)
    u32 n_unspliced;  // unspliced mapped records with a reference span
    u32 hot_first, hot_last; // first / last record ordinal whose bound reaches the cap
    u32 n_dropped;
    u32 _pad;
};

// KX1: one thread per record (every record of the contig).  Classifies the record as the reference's
// separateBams does, records its position / exclusive end for the rank queries, counts its end in `ce`
// and adds its M / = / X runs to the depth difference array; the name codes of spliced records are appended
// (order is irrelevant: they only feed a multiset).
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(256) void kx_classify(DevBatch b, int32_t ref_len, int32_t *x_pos,
                                                    int32_t *x_endx, uint8_t *q_flag, u32 *ce, int32_t *dd, u32 *zlist,
                                                    u32 zcap, ExtraCounters *cnt) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool on = r < b.n;
    bool spliced = false;
    if (on) {
        const u32 g = b.base + (u32)r;
        const u32 c0 = b.cig_off[r], c1 = b.cig_off[r + 1];
        const int32_t pos = b.pos[r];
        int32_t aligned = 0;
        for (u32 k = c0; k < c1; k++) {
            const u32 op = b.cigar[k];
            const u32 ty = op & 15u;
            if (ty == OP_N) spliced = true;
            if (op_consumes_ref(ty)) aligned += (int32_t)(op >> 4);
        }
        const bool mapped = !(b.flag[r] & 0x4u);
        const bool unspliced = !spliced && mapped; // a record of unspliced.bam
        const bool spans = unspliced && aligned > 0 && pos >= 0;
        x_pos[g] = pos;
        x_endx[g] = pos + (aligned > 0 ? aligned : 1); // bam_endpos
        q_flag[g] = spans ? 1 : 0;
        if (unspliced && aligned == 0) { // getEnd() == pos - 1: handled one by one in kx_flank (practically never)
            const u32 z = atomicAdd(&cnt->n_zero, 1u);
            if (z < zcap) zlist[z] = (u32)pos;
        }
        if (spans) {
            int32_t e = pos + aligned - 1; // getEnd()
            if (e > ref_len) e = ref_len;
            atomicAdd(&ce[e], 1u);
            int32_t x = pos;
            for (u32 k = c0; k < c1; k++) {
                const u32 op = b.cigar[k];
                const u32 ty = op & 15u;
                const int32_t ln = (int32_t)(op >> 4);
                if (ty == OP_M || ty == 7u || ty == 8u) {
                    int32_t a = x, bb = x + ln;
                    if (a < 0) a = 0;
                    if (bb > ref_len) bb = ref_len;
                    if (a < bb) {
                        atomicAdd(&dd[a], 1);
                        atomicAdd(&dd[bb], -1);
                    }
                }
                if (op_consumes_ref(ty)) x += ln;
            }
        }
    }
}
#endif // PJB_KERNELS_EXTRA

// The name codes of the spliced records, in BAM order, without a single atomic: k1_count left, per 1024-record tile,
// the ordered list of its spliced records (spl_idx) and their number (tile_stats); one block scans the tile counts,
// then a block per tile copies its records' codes to the tile's place.  (A wave-aggregated atomicAdd on one counter
// per wavefront of records cost 1.1 ms per 10 M records: 156 k returning atomics on ONE address.)
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(1024) void kx_spliced_offsets(const TileStats *ts, u32 n_tiles, u32 *off, ExtraCounters *cnt) {
    __shared__ u32 wsum[16];
    __shared__ u32 carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (u32 base = 0; base < n_tiles; base += 1024) {
        const u32 i = base + threadIdx.x;
        const u32 v = i < n_tiles ? ts[i].spliced : 0u;
        const u32 inc = wave_iscan(v);
        const int w = threadIdx.x >> 6;
        if (lane_id() == 63) wsum[w] = inc;
        __syncthreads();
        u32 wb = 0, tot = 0;
        for (int k = 0; k < 16; k++) {
            const u32 t = wsum[k];
            if (k < w) wb += t;
            tot += t;
        }
        const u32 carry = carry_s;
        if (i < n_tiles) off[i] = carry + wb + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) cnt->n_spliced = carry_s;
}
#endif // PJB_KERNELS_EXTRA
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(256) void kx_spliced_codes(DevBatch b, const TileStats *ts, const u32 *spl_idx, const u32 *off, u64 *spl_codes) {
    const u32 tile = b.tile_base + blockIdx.x;
    const u32 nspl = ts[tile].spliced, o = off[tile];
    for (u32 ks = threadIdx.x; ks < nspl; ks += 256) spl_codes[o + ks] = b.name_hash[spl_idx[(size_t)tile * K1_TILE + ks]];
}
#endif // PJB_KERNELS_EXTRA


// scan functors -----------------------------------------------------------------------------------------------
struct ArrU32Fn {
    const u32 *a;
    __device__ u64 operator()(u64 i) const { return a[i]; }
};
struct ArrU8Fn {
    const uint8_t *a;
    __device__ u64 operator()(u64 i) const { return a[i]; }
};
struct ArrI32Fn { // signed terms, summed modulo 2^64
    const int32_t *a;
    __device__ u64 operator()(u64 i) const { return (u64)(int64_t)a[i]; }
};
struct ExclusiveU32Sink { // out[i] = sum of the terms before i (may alias the input)
    u32 *out;
    __device__ void operator()(u64 i, u64, u64 ex) const { out[i] = (u32)ex; }
};
struct InclusiveU32Sink { // out[i] = sum of the terms up to and including i (may alias the input)
    u32 *out;
    __device__ void operator()(u64 i, u64 v, u64 ex) const { out[i] = (u32)(ex + v); }
};

// KX3: the pileup's record cap.  bam_plp_push (sam.c:1906) drops a record that starts where the iterator
// stands -- the start of the last record it kept -- while more than maxcnt list nodes exist: the kept records
// with bam_endpos >= that position, plus two.  An upper bound for record i: every earlier unspliced record with
// endpos >= pos_i, kept or not = (unspliced records before i) - (unspliced records with endpos < pos_i).  If
// the bound stays below maxcnt - 1 everywhere, nothing is dropped and the difference array already holds the
// reference's depth (the normal case: it takes 8000-fold coverage of unspliced records).
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(256) void kx_cap_bound(const int32_t *x_pos, const uint8_t *q_flag, const u32 *prefix_q,
                                                     const u32 *pe, u32 n, int32_t ref_len, u32 *bound, ExtraCounters *cnt) {
    const u32 g = blockIdx.x * 256 + threadIdx.x;
    u32 u = 0;
    bool hot = false;
    if (g < n && q_flag[g]) {
        const int32_t p = x_pos[g];
        // records with endpos < p  <=>  getEnd() < p - 1
        const int32_t idx = p - 1 < 0 ? 0 : (p - 1 > ref_len + 1 ? ref_len + 1 : p - 1);
        u = prefix_q[g] - pe[idx];
        if (bound) bound[g] = u;
        hot = u + 2 > PLP_MAXCNT;
    }
    // (one counter for every wavefront of records: look first -- the maximum settles after a few waves, and 156 k
    // atomics on one address are served one at a time; a stale look only costs an atomic)
    const u32 wm = wave_max(u);
    if (lane_id() == 0 && wm > cnt->max_buffered) atomicMax(&cnt->max_buffered, wm);
    if (hot) {
        atomicMin(&cnt->hot_first, g);
        atomicMax(&cnt->hot_last, g);
    }
}
#endif // PJB_KERNELS_EXTRA

// KX3b: exact replay of the cap over the span of records whose bound reaches it (outside the span every record
// is kept whatever happened before: its list is at most its bound).  Sequential by nature -- whether a record is
// kept depends on which earlier ones were -- so one lane walks the span; `de` (zeroed, ref_len + 2 entries)
// counts dropped records by endpos.  Kept records inside the list = bound - dropped records still inside.
#ifdef PJB_KERNELS_EXTRA
__global__ void kx_cap_replay(const int32_t *x_pos, const int32_t *x_endx, const uint8_t *q_flag, const u32 *bound, u32 n,
                              int32_t ref_len, u32 *de, uint8_t *dropped, ExtraCounters *cnt) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const u32 g0 = cnt->hot_first, g1 = cnt->hot_last;
    if (g0 > g1 || g0 >= n) return;
    // start of the last unspliced record before the span (it was kept: nothing before the span is dropped)
    int32_t last_kept_pos = -1;
    for (u32 g = g0; g-- > 0;)
        if (q_flag[g]) {
            last_kept_pos = x_pos[g];
            break;
        }
    u32 dtotal = 0, dgone = 0;
    int32_t sweep = x_pos[g0];
    for (u32 g = g0; g <= g1; g++) {
        if (!q_flag[g]) continue;
        const int32_t p = x_pos[g];
        while (sweep < p && sweep <= ref_len + 1) dgone += de[sweep++];
        const u32 in_list = bound[g] - (dtotal - dgone);
        if (p == last_kept_pos && in_list + 2 > PLP_MAXCNT) {
            dropped[g] = 1;
            int32_t e = x_endx[g];
            if (e > ref_len + 1) e = ref_len + 1;
            if (e < sweep) e = sweep;
            de[e]++;
            dtotal++;
        } else
            last_kept_pos = p;
    }
    cnt->n_dropped = dtotal;
}
#endif // PJB_KERNELS_EXTRA

// KX3c: take the dropped records' M / = / X runs back out of the difference array (before the scan).
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(256) void kx_undo_dropped(DevBatch b, int32_t ref_len, const uint8_t *dropped, int32_t *dd) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= b.n || !dropped[b.base + (u32)r]) return;
    int32_t x = b.pos[r];
    for (u32 k = b.cig_off[r]; k < b.cig_off[r + 1]; k++) {
        const u32 op = b.cigar[k];
        const u32 ty = op & 15u;
        const int32_t ln = (int32_t)(op >> 4);
        if (ty == OP_M || ty == 7u || ty == 8u) {
            int32_t a = x, bb = x + ln;
            if (a < 0) a = 0;
            if (bb > ref_len) bb = ref_len;
            if (a < bb) {
                atomicAdd(&dd[a], -1);
                atomicAdd(&dd[bb], 1);
            }
        }
        if (op_consumes_ref(ty)) x += ln;
    }
}
#endif // PJB_KERNELS_EXTRA

__device__ __forceinline__ u32 lower_bound_i32(const int32_t *a, u32 n, int32_t v) { // first index with a[i] >= v
    u32 lo = 0, hi = n;
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

struct ExtraRow { // what pjb_extra_finish hands back, parallel to the junction rows
    double mm_score, coverage;
    u32 up_aln, down_aln;
    u32 m_sum; // sum of name multiplicities (uint32 arithmetic as in junction.cc:916-919)
    u32 _pad;
};

// KX4: flanking alignment counts, one thread per junction (processJunctionVicinity, junction.cc:651-677):
//   up   = unspliced records with  intron.start > pos  and  leftAncStart <= getEnd()
//   down = unspliced records with  rightAncEnd >= pos  and  intron.end < pos
// A record with a reference span has getEnd() >= pos, so "getEnd() < left" implies "pos < intron.start" and
//   up = #{pos < start} - #{getEnd() < left};  down = #{pos <= right} - #{pos <= end}.
// prefix_q[i] = such records among the first i records (all records are position sorted); pe[x] = such records
// with getEnd() < x.  The reference's region query (sam_itr_queryi over [left - maxQueryLength - 1, right +
// maxQueryLength + 1)) cannot exclude a record either test accepts.  Records without a reference span
// (getEnd() == pos - 1) are tested one by one from `zlist`.
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(256) void kx_flank(const pjb_junction_row *rows, u32 n_rows, const int32_t *x_pos, u32 n_reads,
                                                 const u32 *prefix_q, const u32 *pe, int32_t ref_len, const u32 *zlist,
                                                 const ExtraCounters *cnt, u32 zcap, ExtraRow *out) {
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_rows) return;
    const int32_t s = rows[j].start, e = rows[j].end, l = rows[j].left, r = rows[j].right;
    auto before = [&](int32_t v) -> u32 { return prefix_q[lower_bound_i32(x_pos, n_reads, v)]; }; // #{pos < v}
    auto clampx = [&](int32_t v) -> int32_t { return v < 0 ? 0 : (v > ref_len + 1 ? ref_len + 1 : v); };
    u32 up = before(s) - pe[clampx(l)];
    u32 down = before(r == INT32_MAX ? r : r + 1) - before(e == INT32_MAX ? e : e + 1);
    u32 nz = cnt->n_zero;
    if (nz > zcap) nz = zcap;
    for (u32 k = 0; k < nz; k++) {
        const int32_t pos = (int32_t)zlist[k], end = pos - 1;
        if (s > pos && l <= end) up++;
        if (r >= pos && e < pos) down++;
    }
    out[j].up_aln = up;
    out[j].down_aln = down;
}
#endif // PJB_KERNELS_EXTRA

// KX5: per sorted pair, the name code of its record and the global row it belongs to (kept until pjb_extra_finish)
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(256) void kx_pair_codes(const u32 *sidx, const u32 *jid_of, const u32 *pair_g, const DevBatch *batches,
                                                      int n_batches, u32 n, u32 row_base,
                                                      u64 *pair_code, u32 *pair_row) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32 g = pair_g[sidx[i]];
    int bi = n_batches - 1;
    while (bi > 0 && g < batches[bi].base) bi--;
    pair_code[i] = batches[bi].name_hash[g - batches[bi].base];
    pair_row[i] = row_base + jid_of[i];
}
#endif // PJB_KERNELS_EXTRA

// ---- the sparse path (round 3): no array of the target's length --------------------------------------------------
// The depth of the unspliced records is only ever READ at 42 positions per junction (Junction::calcCoverage,
// junction.cc:935-951) and the ends histogram at one position per junction (processJunctionVicinity) -- while building
// them costs two memsets, 3 atomics per record and two scans over the target's LENGTH (0.5 ms each per 100 Mb).  The
// records are sorted by position, and a record that covers x starts within max_span bases before x: so both questions
// are answered from the records themselves, by a thread per junction that walks the few records starting in
// (x - max_span, x].  What is kept of a target until pjb_extra_finish: the unspliced records with a span, compacted --
// pos[k], end[k] (exclusive) in file order -- and the list of their D operations ("gaps": inside the span, not counted by
// the pileup) with one offset per 256 records.  (The spliced records are left out: a junction's own alignments start
// right before it, 300 k of them at the deepest junction of configs[1], and would all be walked for nothing.)
//   htslib's 8000-record cap (kx_cap_*) is only possible where 7999 consecutive unspliced records start within max_span
// bases (kx_cap_check); such a target -- and one with more gaps than the list holds -- goes through the dense path
// above instead.
// (SparseCounters, XOut: pjb_kernels.hip.h -- k1_count writes them when the records go through it; kx_classify_sparse is
// the same classification for the records of a target that went through k1_walk)
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(256) void kx_classify_sparse(DevBatch b, int32_t *s_pos, int32_t *s_end, uint8_t *q_flag, u32 *zlist, u32 zcap,
                                                           SparseCounters *cnt) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    u32 span = 0, gapmax = 0;
    bool many = false;
    if (r < b.n) {
        const u32 g = b.base + (u32)r;
        const u32 c0 = b.cig_off[r], c1 = b.cig_off[r + 1];
        const int32_t pos = b.pos[r];
        int32_t aligned = 0;
        u32 ngap = 0;
        bool spliced = false;
        for (u32 k = c0; k < c1; k++) {
            const u32 op = b.cigar[k];
            const u32 ty = op & 15u;
            if (ty == OP_N) spliced = true;
            if (op_consumes_ref(ty)) {
                aligned += (int32_t)(op >> 4);
                if (ty == 2u && (op >> 4)) { // D: inside the span, no depth
                    ngap++;
                    gapmax = max(gapmax, op >> 4);
                }
            }
        }
        const bool mapped = !(b.flag[r] & 0x4u);
        const bool unspliced = !spliced && mapped;
        const bool spans = unspliced && aligned > 0 && pos >= 0;
        s_pos[g] = pos;
        s_end[g] = spans ? pos + aligned : pos;
        if (!spans) ngap = 0, gapmax = 0;
        if (ngap > SPARSE_GAP_MAX) many = true, ngap = SPARSE_GAP_MAX;
        q_flag[g] = (uint8_t)((spans ? 1u : 0u) | (ngap << 1));
        if (spans) span = (u32)aligned;
        if (unspliced && aligned == 0) {
            const u32 z = atomicAdd(&cnt->n_zero, 1u);
            if (z < zcap) zlist[z] = (u32)pos;
        }
    }
    span = wave_max(span);
    gapmax = wave_max(gapmax);
    if (lane_id() == 0) { // (look first: the maxima settle after a few waves)
        if (span > cnt->max_span) atomicMax(&cnt->max_span, span);
        if (gapmax > cnt->max_gap) atomicMax(&cnt->max_gap, gapmax);
    }
    if (many) atomicOr(&cnt->need_dense, 2u);
}
#endif // PJB_KERNELS_EXTRA

// one scan over the records: (records with a span) | (gaps) << 32
struct SparseFn {
    const uint8_t *q;
    __device__ u64 operator()(u64 i) const {
        const u32 v = q[i];
        return (u64)(v & 1u) | ((u64)(v >> 1) << 32);
    }
};
struct Gap {
    int32_t start, end; // [start, end)
};
struct SparseSink { // the records with a span, compacted in rank order; gaps before every 256th record and every 256th rank
    int32_t *comp_pos, *comp_end;
    u32 *gapoff_rec, *gapoff_rank;
    const int32_t *s_pos, *s_end;
    const uint8_t *q;
    __device__ void operator()(u64 i, u64, u64 ex) const {
        const u32 rank = (u32)ex, gbefore = (u32)(ex >> 32);
        if ((i & 255u) == 0) gapoff_rec[i >> 8] = gbefore;
        if (q[i] & 1u) {
            comp_pos[rank] = s_pos[i];
            comp_end[rank] = s_end[i];
            if ((rank & 255u) == 0) gapoff_rank[rank >> 8] = gbefore;
        }
    }
};
// the gaps (D operations) of the unspliced records in record order: a block per 256 records (global ordinals, so that
// gapoff[block] is the block's first entry), an exclusive scan of the records' gap counts inside the block
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(256) void kx_gaps(DevBatch b, const uint8_t *q, u32 n_total, const u32 *gapoff, Gap *gaps, u32 gap_cap, SparseCounters *cnt,
                                                u32 n_blocks) {
    __shared__ u32 wsum[4];
    if ((cnt->total >> 32) == 0) return; // (no gap in the whole target: the usual case for short reads; the grid is small for that)
    const u32 first_block = b.base >> 8;
    for (u32 blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) { // (uniform per block: the barriers are safe)
        const u64 g = ((u64)(first_block + blk) << 8) + threadIdx.x; // global record ordinal
        const u32 ngap = g < n_total ? (u32)(q[g] >> 1) : 0u;
        const u32 inc = wave_iscan(ngap);
        const int w = threadIdx.x >> 6;
        __syncthreads();
        if (lane_id() == 63) wsum[w] = inc;
        __syncthreads();
        u32 before = gapoff[first_block + blk] + inc - ngap;
        for (int k = 0; k < w; k++) before += wsum[k];
        if (!ngap || g < b.base || g >= (u64)b.base + (u64)b.n) continue; // (a neighbouring batch's record: it only counts)
        if (before + ngap > gap_cap) {
            atomicOr(&cnt->need_dense, 4u);
            continue;
        }
        const u32 r = (u32)g - b.base;
        int32_t x = b.pos[r];
        u32 k_out = 0;
        for (u32 k = b.cig_off[r]; k < b.cig_off[r + 1] && k_out < ngap; k++) {
            const u32 op = b.cigar[k], ty = op & 15u;
            const int32_t ln = (int32_t)(op >> 4);
            if (ty == 2u && ln) gaps[before + k_out++] = Gap{x, x + ln};
            if (op_consumes_ref(ty)) x += ln;
        }
    }
}
#endif // PJB_KERNELS_EXTRA

// the pileup's cap can only bite where PLP_MAXCNT - 1 records with a span start within max_span bases of each other
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(256) void kx_cap_check(const int32_t *comp_pos, SparseCounters *cnt) {
    const u32 n = (u32)cnt->total, K = PLP_MAXCNT - 1;
    const u32 r = blockIdx.x * 256 + threadIdx.x + K;
    if (r >= n) return;
    if ((int64_t)comp_pos[r] - (int64_t)comp_pos[r - K] <= (int64_t)cnt->max_span) atomicOr(&cnt->need_dense, 1u);
}
#endif // PJB_KERNELS_EXTRA

__device__ __forceinline__ u32 lower_bound_i64(const int32_t *a, u32 n, int64_t v) { // first index with a[i] >= v
    u32 lo = 0, hi = n;
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if ((int64_t)a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// Every lane has a range [i0, i1) of its own and wants acc(i, p) summed over it, p being the lane's parameters.  The ranges are
// a handful of records where coverage is ordinary -- the lane walks them itself -- and thousands in a pile-up: those are taken
// one after the other by the whole wavefront.  (A wavefront per junction was tried first: its 64-wide searches touched 64 cache
// lines per probe, 0.8 ms per 50 k junctions; a search per lane shares its upper levels with the neighbouring junctions'.)
// Every lane of the wavefront must call this.
constexpr u32 RANGE_SHORT = 48;
struct Win { // two windows of positions, [p1lo, p1hi] and [p2lo, p2hi] (empty: hi < lo)
    int32_t p1lo, p1hi, p2lo, p2hi;
};
template <class F>
__device__ __forceinline__ void ranges_sum2(u32 i0, u32 i1, const Win w, F acc, u32 &s1, u32 &s2) {
    const bool big = i1 > i0 && i1 - i0 > RANGE_SHORT;
    if (!big)
        for (u32 i = i0; i < i1; i++) acc(i, w, s1, s2);
    u64 m = __ballot(big);
    const int lane = lane_id();
    while (m) {
        const int src = __ffsll((unsigned long long)m) - 1;
        m &= m - 1;
        const u32 j0 = (u32)__shfl((int)i0, src, 64), j1 = (u32)__shfl((int)i1, src, 64);
        Win ws;
        ws.p1lo = __shfl(w.p1lo, src, 64);
        ws.p1hi = __shfl(w.p1hi, src, 64);
        ws.p2lo = __shfl(w.p2lo, src, 64);
        ws.p2hi = __shfl(w.p2hi, src, 64);
        u32 t1 = 0, t2 = 0;
        for (u32 i = j0 + (u32)lane; i < j1; i += 64) acc(i, ws, t1, t2);
        t1 = wave_total<DppAdd>(t1); // (every lane gets the sum)
        t2 = wave_total<DppAdd>(t2);
        if (lane == src) {
            s1 += t1;
            s2 += t2;
        }
    }
}

// KX4 without the ends histogram: a thread per junction.
//   #{getEnd() < x} = (records with a span that start before x - max_span: all of them) + (those among the records starting
//   in [x - max_span, x) whose last base lies before x); a record starting at or after x ends at or after x.
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(256) void kx_flank_sparse(const pjb_junction_row *rows, u32 n_rows, const int32_t *s_pos, const int32_t *s_end,
                                                        int32_t ref_len, const u32 *zlist, const SparseCounters *cnt, u32 zcap, ExtraRow *out) {
    const u32 n_reads = (u32)cnt->total; // (s_pos / s_end: the records with a span only, in rank order)
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    const bool on = j < n_rows;
    int32_t s = 0, e = 0, l = 0, r = 0;
    if (on) s = rows[j].start, e = rows[j].end, l = rows[j].left, r = rows[j].right;
    auto before = [&](int32_t v) -> u32 { return lower_bound_i32(s_pos, n_reads, v); }; // #{pos < v}
    const int32_t lc = l < 0 ? 0 : (l > ref_len + 1 ? ref_len + 1 : l);
    u32 ended = 0, i0 = 0, i1 = 0;
    if (on) {
        if (lc == ref_len + 1) ended = n_reads; // (the histogram clamps every end to ref_len)
        else {
            i0 = lower_bound_i64(s_pos, n_reads, (int64_t)lc - (int64_t)cnt->max_span);
            i1 = lower_bound_i32(s_pos, n_reads, lc);
            ended = i0;
        }
    }
    u32 part = 0, unused = 0;
    ranges_sum2(i0, i1, Win{lc, 0, 0, 0}, [&](u32 i, const Win w, u32 &a, u32 &) {
        if (s_end[i] - 1 < w.p1lo) a++;
    }, part, unused);
    if (!on) return;
    u32 up = before(s) - (ended + part);
    u32 down = before(r == INT32_MAX ? r : r + 1) - before(e == INT32_MAX ? e : e + 1);
    u32 nz = cnt->n_zero;
    if (nz > zcap) nz = zcap;
    for (u32 k = 0; k < nz; k++) {
        const int32_t pos = (int32_t)zlist[k], end = pos - 1;
        if (s > pos && l <= end) up++;
        if (r >= pos && e < pos) down++;
    }
    out[j].up_aln = up;
    out[j].down_aln = down;
}
#endif // PJB_KERNELS_EXTRA

// what a target keeps for pjb_extra_finish (device pointers): the records with a span in rank order, their gaps (record
// order = rank order) and the number of gaps before every 256th of them
struct SparseDepth {
    const int32_t *s_pos, *s_end;
    const Gap *gaps;
    const u32 *gapoff;
    u32 n_reads, n_gaps, max_span, max_gap;
};
__device__ __forceinline__ u32 span_overlap(int64_t x0, int64_t x1, int32_t plo, int32_t phi) { // |[x0, x1) n [plo, phi]|
    const int64_t lo = x0 > plo ? x0 : (int64_t)plo, hi = (x1 - 1) < phi ? (x1 - 1) : (int64_t)phi;
    return hi >= lo ? (u32)(hi - lo + 1) : 0u;
}
// sum over i in [a, b], 1 <= i < len, of the depth at position i - 1 -- for two windows [a1, b1], [a2, b2] at once; every lane of
// the wavefront calls this (`on`: the lane has a junction)
__device__ __forceinline__ void sparse_cov2(const SparseDepth D, int32_t len, bool on, int32_t a1, int32_t b1, int32_t a2, int32_t b2, u32 &sum1,
                                            u32 &sum2) {
    // window [a, b] in i -> positions [max(a, 1) - 1, min(b, len - 1) - 1]
    auto lo_of = [&](int32_t a) -> int32_t { return (a < 1 ? 1 : a) - 1; };
    auto hi_of = [&](int32_t b) -> int32_t { return b < 0 ? -1 : (b > len - 1 ? len - 1 : b) - 1; };
    const Win w{lo_of(a1), hi_of(b1), lo_of(a2), hi_of(b2)};
    const bool e1 = w.p1hi >= w.p1lo, e2 = w.p2hi >= w.p2lo;
    int64_t lo = 0, hi = -1;
    if (e1) lo = w.p1lo, hi = w.p1hi;
    if (e2) {
        lo = e1 && lo < w.p2lo ? lo : (int64_t)w.p2lo;
        hi = e1 && hi > w.p2hi ? hi : (int64_t)w.p2hi;
    }
    u32 i0 = 0, i1 = 0, g0 = 0, g1 = 0;
    if (on && hi >= lo) {
        i0 = lower_bound_i64(D.s_pos, D.n_reads, lo - (int64_t)D.max_span + 1);
        i1 = lower_bound_i64(D.s_pos, D.n_reads, hi + 1);
        if (D.n_gaps && i1 > i0) { // a gap that covers a position of the window belongs to a record starting in (lo - max_span, hi]
            const u32 k1 = (i1 + 255u) >> 8;
            g0 = D.gapoff[i0 >> 8];
            g1 = (u64)k1 * 256u < (u64)D.n_reads ? D.gapoff[k1] : D.n_gaps;
        }
    }
    u32 s1 = 0, s2 = 0, m1 = 0, m2 = 0;
    ranges_sum2(i0, i1, w, [&](u32 i, const Win ww, u32 &a, u32 &b) {
        const int64_t x0 = D.s_pos[i], x1 = D.s_end[i];
        a += span_overlap(x0, x1, ww.p1lo, ww.p1hi);
        b += span_overlap(x0, x1, ww.p2lo, ww.p2hi);
    }, s1, s2);
    ranges_sum2(g0, g1, w, [&](u32 k, const Win ww, u32 &a, u32 &b) {
        const Gap g = D.gaps[k];
        a += span_overlap(g.start, g.end, ww.p1lo, ww.p1hi);
        b += span_overlap(g.start, g.end, ww.p2lo, ww.p2hi);
    }, m1, m2);
    sum1 = s1 - m1;
    sum2 = s2 - m2;
}
// Junction::calcCoverage (junction.cc:923-951) from a target's records instead of its depth vector: a thread per junction
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(256) void kx_coverage_sparse(const pjb_junction_row *rows, u32 row0, u32 n, SparseDepth D, int32_t len_src, ExtraRow *out) {
    const u32 k = blockIdx.x * 256 + threadIdx.x;
    const bool on = k < n;
    const u32 j = row0 + (on ? k : 0u);
    int32_t s = 0, e = 0;
    if (on) s = rows[j].start, e = rows[j].end;
    u32 d1, d2, a1, a2;
    sparse_cov2(D, len_src, on, s - 20, s - 11, s - 10, s, d1, d2);
    sparse_cov2(D, len_src, on, e + 10, e + 20, e, e + 9, a1, a2);
    if (!on) return;
    const double donor = (1.0 / 9.0) * (double)d1 - (1.0 / 10.0) * (double)d2;
    const double acceptor = (1.0 / 10.0) * (double)a1 - (1.0 / 9.0) * (double)a2;
    out[j].coverage = donor + acceptor;
}
#endif // PJB_KERNELS_EXTRA

// ---- phase 2 (all contigs done) ---------------------------------------------------------------------------
constexpr u64 NAME_EMPTY = ~0ull;
__device__ __forceinline__ u64 name_slot_key(u64 code) { return code == NAME_EMPTY ? code - 1 : code; }
// (any number of slots, not a power of two: the table of a file is sized to its names, and 144 MB stay in the 256 MB of
// Infinity Cache where 268 MB do not)
__device__ __forceinline__ u32 name_slot_of(u64 key, u32 slots) { return (u32)((((key * 0x9e3779b97f4a7c15ULL) >> 32) * (u64)slots) >> 32); }
__device__ __forceinline__ u32 name_next(u32 h, u32 slots) { return h + 1 == slots ? 0u : h + 1; }

// splicedAlignmentMap[code]++ (junction_builder.cc:173-174) for every spliced record of the file.  One 16-byte slot per
// name: key and count share a sector (a lookup is one random access, not two); the table is filled with 0xff, so an empty
// slot's key is NAME_EMPTY and a slot's count is `count + 1` in 32-bit arithmetic.
struct NameSlot {
    u64 key;
    u32 count, _pad;
};
__device__ __forceinline__ void name_add(NameSlot *tab, u32 slots, u64 key, u32 n) {
    u32 h = name_slot_of(key, slots);
    for (;;) {
        u64 cur = tab[h].key;
        if (cur == NAME_EMPTY) cur = atomicCAS((unsigned long long *)&tab[h].key, (unsigned long long)NAME_EMPTY, (unsigned long long)key);
        if (cur == NAME_EMPTY || cur == key) {
            atomicAdd(&tab[h].count, n);
            return;
        }
        h = name_next(h, slots);
    }
}
// the table has grown: every name of the old one into the new one
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(256) void kx_name_rehash(const NameSlot *old_tab, u32 old_slots, NameSlot *tab, u32 slots) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= old_slots) return;
    const NameSlot o = old_tab[i];
    if (o.key != NAME_EMPTY) name_add(tab, slots, o.key, o.count + 1u);
}
#endif // PJB_KERNELS_EXTRA
// M of every junction: sum over its alignments of the map entry of their code (junction.cc:916-919, uint32).  Four pairs
// per thread, a wavefront's 256 consecutive pairs in four rounds: the four random probes of a thread are in flight together.
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(256) void kx_name_sum(const u64 *pair_code, const u32 *pair_row, u32 n, const NameSlot *tab, u32 slots, ExtraRow *out) {
    const u32 base = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 256 + (u32)lane_id();
    u64 key[4];
    u32 h[4], row[4], c[4];
    NameSlot sl[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { // (unconditional loads from a clamped index, masked after: they travel together)
        const u32 i = base + 64 * k, ic = i < n ? i : n - 1;
        const u64 code = pair_code[ic];
        const u32 rw = pair_row[ic];
        key[k] = i < n ? name_slot_key(code) : 0ull;
        row[k] = i < n ? rw : 0xffffffffu;
        h[k] = name_slot_of(key[k], slots);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) sl[k] = tab[h[k]];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u32 i = base + 64 * k;
        while (i < n && sl[k].key != key[k] && sl[k].key != NAME_EMPTY) {
            h[k] = name_next(h[k], slots);
            sl[k] = tab[h[k]];
        }
        c[k] = i < n && sl[k].key == key[k] ? sl[k].count + 1u : 0u;
    }
    // pairs are stored junction by junction: fold equal rows inside the wave before the atomic
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u32 i = base + 64 * k;
        const u32 cc = seg_reduce_to_head(c[k], row[k], OpAdd());
        const u32 prev = __shfl_up(row[k], 1, 64);
        if (i < n && (lane_id() == 0 || prev != row[k])) atomicAdd(&out[row[k]].m_sum, cc);
    }
}
#endif // PJB_KERNELS_EXTRA
// four codes per thread into the table (the probes of a thread in flight together)
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(256) void kx_name_insert4(const u64 *codes, u32 n, NameSlot *tab, u32 slots) {
    const u32 base = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 256 + (u32)lane_id();
    u64 key[4];
    u32 h[4];
    u64 cur[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u32 i = base + 64 * k;
        const u64 code = codes[i < n ? i : n - 1];
        key[k] = i < n ? name_slot_key(code) : 0ull;
        h[k] = name_slot_of(key[k], slots);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) cur[k] = tab[h[k]].key; // (a plain look first: most codes of a deep file are in the table already)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (base + 64 * k >= n) continue;
        if (cur[k] == key[k]) atomicAdd(&tab[h[k]].count, 1u);
        else name_add(tab, slots, key[k], 1u);
    }
}
#endif // PJB_KERNELS_EXTRA

// Junction::calcCoverage (junction.cc:923-951) for the rows [row0, row0 + n) against the depth vector of one
// target: levels[i] = cover[i - 1] (DepthParser stores a position's depth at pos + 1, depth_parser.cc:127,147),
// entries outside [0, len_src) are skipped as the reference's bounds test does.
__device__ __forceinline__ double cov_window(const u32 *cover, int32_t len, int32_t a, int32_t b) {
    const double multiplier = 1.0 / (double)(b - a);
    u32 readCount = 0;
    for (int32_t i = a; i <= b; i++)
        if (i >= 1 && i < len) readCount += cover[i - 1];
    return multiplier * (double)readCount;
}
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(256) void kx_coverage(const pjb_junction_row *rows, u32 row0, u32 n, const u32 *cover, int32_t len_src, ExtraRow *out) {
    const u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const u32 j = row0 + k;
    const int32_t s = rows[j].start, e = rows[j].end;
    const double donor = cov_window(cover, len_src, s - 20, s - 11) - cov_window(cover, len_src, s - 10, s);
    const double acceptor = cov_window(cover, len_src, e + 10, e + 20) - cov_window(cover, len_src, e, e + 9);
    out[j].coverage = donor + acceptor;
}
#endif // PJB_KERNELS_EXTRA
// the finished columns of every junction, in the layout pjb_extra_finish hands out (mm_score = N / M, junction.cc:920)
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(256) void kx_rows_out(const pjb_junction_row *rows, const ExtraRow *x, u32 n, pjb_extra_row *out) {
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    pjb_extra_row o;
    o.mm_score = (double)rows[j].nb_raw / (double)x[j].m_sum;
    o.coverage = x[j].coverage;
    o.up_aln = x[j].up_aln;
    o.down_aln = x[j].down_aln;
    out[j] = o;
}
#endif // PJB_KERNELS_EXTRA

// ---- filt feature rows: ModelFeatures::setRow, lib/src/model_features.cc:161-212 ------------------
struct DevModels {
    const double *exon, *intron, *don_t, *don_f, *acc_t, *acc_f, *don_pw, *acc_pw; // device tables, nullptr = untrained
    int exon_size, intron_size, don_pw_size, acc_pw_size;
};
struct GenomeRef {
    const uint8_t *d; // upper-cased bases
    int32_t len;
};
// SeqUtils::makeClean code of the base at window position i of [b, e] (faidx-clamped), read backwards and complemented
// when the junction's consensus strand is negative (SeqUtils::reverseComplement of the fetched string)
__device__ __forceinline__ int window_code(const uint8_t *g, int32_t b, int32_t e, int neg, int32_t i) {
    uint8_t c = neg ? revcomp_char(g[e - i]) : g[b + i];
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
}
// KmerMarkovModel::getScore, markov_model.cc:57-78 (order 5)
__device__ double kmer_score(const double *tab, const uint8_t *g, int32_t glen, int32_t beg, int32_t end, int neg) {
    int32_t b = beg, e = end;
    fetch_clamp(glen, b, e);
    const int32_t n = glen > 0 ? e - b + 1 : 0;
    double score = 1.0;
    u32 no_count = 0;
    u32 ctx = 0;
    for (int32_t i = 0; i < n; i++) {
        const int c = window_code(g, b, e, neg, i);
        if (i >= PJB_KMER_ORDER) {
            const double m = tab ? tab[(size_t)ctx * 5 + (u32)c] : 0.0;
            if (m != 0.0) score = __dmul_rn(score, m);
            else no_count++;
        }
        ctx = (ctx * 5u + (u32)c) % 3125u;
    }
    if (score == 0.0) return -100.0;
    if (no_count > 2) score = score / ((double)no_count * 0.5);
    return log(score);
}
// PosMarkovModel::getScore, markov_model.cc:101-115 (order 1); *len_out = length of the window
__device__ double pos_score(const double *tab, const uint8_t *g, int32_t glen, int32_t beg, int32_t end, int neg, int32_t *len_out) {
    int32_t b = beg, e = end;
    fetch_clamp(glen, b, e);
    const int32_t n = glen > 0 ? e - b + 1 : 0;
    *len_out = n;
    double score = 1.0;
    for (int32_t i = 1; i < n && i < PJB_PW_LEN; i++) {
        const int c = window_code(g, b, e, neg, i);
        score = __dmul_rn(score, tab ? tab[i * 5 + c] : 0.0);
    }
    if (score == 0.0) return -300.0;
    return log(score);
}
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(256) void kg_features(const pjb_junction_row *rows, u32 n, const GenomeRef *genomes, int n_refs, DevModels M,
                                                    double mean_read_length, u32 l95, double *out, int *bad) {
    const u32 r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const pjb_junction_row &j = rows[r];
    double *f = out + (size_t)r * PJB_N_FEATURES;
    if (j.refid < 0 || j.refid >= n_refs || !genomes[j.refid].d) {
        atomicOr(bad, 1);
        return;
    }
    const uint8_t *g = genomes[j.refid].d;
    const int32_t gl = genomes[j.refid].len;
    const int neg = j.cons_strand == PJB_STRAND_NEG;
    const int32_t s = j.start, e = j.end;
    // calcSplicingScores, junction.cc:1361-1382: left = [start-3, start+20], right = [end-20, end+2]; on the negative
    // strand the reverse complement of `right` is the donor and that of `left` the acceptor
    const int32_t db = neg ? e - 20 : s - 3, de = neg ? e + 2 : s + 20;
    const int32_t ab = neg ? s - 3 : e - 20, ae = neg ? s + 20 : e + 2;
    int32_t don_len, acc_len;
    const double pws = pos_score(M.don_pw, g, gl, db, de, neg, &don_len) + pos_score(M.acc_pw, g, gl, ab, ae, neg, &acc_len);
    const double ss = (kmer_score(M.don_t, g, gl, db, de, neg) - kmer_score(M.don_f, g, gl, db, de, neg)) +
                      (kmer_score(M.acc_t, g, gl, ab, ae, neg) - kmer_score(M.acc_f, g, gl, ab, ae, neg));
    const u32 size = (u32)(e - s + 1);
    f[0] = 0.0; // isGenuine()
    f[1] = (double)(j.nb_raw - j.nb_ms);
    f[2] = (double)j.nb_dist;
    f[3] = (double)j.nb_rel;
    f[4] = j.entropy;
    f[5] = (double)j.nb_rel / (double)j.nb_raw;
    f[6] = (double)j.max_min_anc;
    f[7] = (double)j.maxmmes;
    f[8] = (double)(u32)j.sum_mismatches / (double)j.nb_raw; // nbMismatches is a uint32 in the reference (junction.cc:893)
    f[9] = l95 == 0 ? 0.0 : (size <= l95 ? 0.0 : log((double)(size - l95)));
    f[10] = (double)(j.hamming5p < j.hamming3p ? j.hamming5p : j.hamming3p);
    if (M.exon_size == 0 || M.intron_size == 0) f[11] = 0.0;
    else { // calcCodingPotential, junction.cc:1328-1359
        double cp = kmer_score(M.exon, g, gl, s - 82, s - 2, neg) - kmer_score(M.intron, g, gl, s - 82, s - 2, neg);
        cp = cp + (kmer_score(M.intron, g, gl, s, s + 80, neg) - kmer_score(M.exon, g, gl, s, s + 80, neg));
        cp = cp + (kmer_score(M.intron, g, gl, e - 80, e, neg) - kmer_score(M.exon, g, gl, e - 80, e, neg));
        cp = cp + (kmer_score(M.exon, g, gl, e + 1, e + 81, neg) - kmer_score(M.intron, g, gl, e + 1, e + 81, neg));
        f[11] = cp;
    }
    // isPWModelEmpty() is asked after calcSplicingScores, whose lookups (operator[]) have by then put every position
    // of a window longer than the model's order into an untrained model's map
    const bool pw_empty = (M.don_pw_size == 0 && don_len <= 1) || (M.acc_pw_size == 0 && acc_len <= 1);
    f[12] = pw_empty ? 0.0 : pws;
    f[13] = pw_empty ? 0.0 : ss;
#pragma unroll
    for (int i = 0; i < 20; i++) { // calcJunctionAnchorDepthLogDeviation, junction.cc:1384-1391
        double Ni = (double)j.jad[i];
        if (Ni == 0.0) Ni = 0.000000000001;
        const double Pi = 1.0 - ((double)i / (double)(mean_read_length / 2.0));
        const double Ei = (double)j.nb_raw * Pi;
        f[14 + i] = log2(Ni / Ei);
    }
}
#endif // PJB_KERNELS_EXTRA

// ---- bamfilt: BamFilter::filter's decision per alignment, src/bam_filter.cc:75-150,190-225.
// The walk is the reference's, including that it does not advance over an N operation (only the else-branch of
// :86-96 adds to lEnd): the introns after a read's first one are looked up short of the earlier introns' lengths.
#ifdef PJB_KERNELS_EXTRA
__global__ __launch_bounds__(256) void kf_filter(const int32_t *pos, const u32 *cig_off, const u32 *cigar, u32 n, const u64 *keys, u32 n_keys,
                                                  int clip_mode, uint8_t *codes) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int32_t lEnd = pos[i];
    u32 nN = 0;
    bool good = false;
    for (u32 k = cig_off[i]; k < cig_off[i + 1]; k++) {
        const u32 op = cigar[k];
        const u32 ty = op & 15u;
        const int32_t ln = (int32_t)(op >> 4);
        if (ty == OP_N) {
            nN++;
            const u64 key = ((u64)(u32)lEnd << 32) | (u64)(u32)(lEnd + ln - 1);
            u32 lo = 0, hi = n_keys; // JunctionSystem::getJunction: is the intron in the set?
            while (lo < hi) {
                const u32 mid = lo + ((hi - lo) >> 1);
                if (keys[mid] < key) lo = mid + 1;
                else hi = mid;
            }
            good |= lo < n_keys && keys[lo] == key;
        } else if (op_consumes_ref(ty)) {
            lEnd += ln;
        }
    }
    uint8_t code = 1;                                    // not spliced: written as is (:221-224)
    if (nN) {
        if (clip_mode == PJB_CLIP_COMPLETE || nN <= 1) code = good ? 2 : 0;   // containsJunctionInSystem (:196-202)
        else code = good ? 3 : 0;                         // clipMSR: !allBad (:204-218)
    }
    codes[i] = code;
}
#endif // PJB_KERNELS_EXTRA

} // namespace pjb
