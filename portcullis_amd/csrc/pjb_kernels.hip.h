// pjb_kernels.hip.h -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the
// Portcullis `junc` hot path.  Integer / byte work bounded by HBM bandwidth; no MFMA.
//
// The chain of one target or group of targets (DESIGN.md section 4 has the table, the data layout and the byte counts):
//   K1   k1_count, k1_scan_tiles   per-read CIGAR walk: N-op count, length stats, sortedness; the tiles' spliced lists     (a1,a2)
//        k1_emit                   the spliced reads of the closed-form shapes: pairs (key, 32-byte record) complete       (a3,a5,a8,a12)
//        k1_generic                every other spliced read: its operations walked, pairs, closed form where possible      (a3,a5,a8,a12)
//   K2d  kd_*                      ordered dense junction ids from candidate keys; junction keys and anchors               (a3,a5)
//   K4b  k4b_generic               window check of the closed forms / the padded query-genome walks (side stream)          (a12,a13)
//   K2   rs_*                      stable LSD radix sort of (junction id, pair index)                                       (a3 grouping)
//   K4   k4_pairs                  sorted pairs -> fragment heads; head / run masks                                         (a8,a13)
//   K2s  k2_runs, k2_expand        position runs per junction (or k2_heads + kf_* on full keys)                             (a3,a7)
//   K5   k5_*                      fragment -> junction reduce, entropy, splice motif, hamming -> rows                      (a6,a7,a9-a11,a13)
//   K6   k6_rows_out               rows to the table; control block to the host
// References in comments are file:line in the reference checkout.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/portcullis_amd.h"

namespace pjb {

typedef unsigned long long u64;
typedef unsigned int u32;

// ---------------------------------------------------------------------------------------------
// device-side structures
// ---------------------------------------------------------------------------------------------
struct DevBatch {
    const int32_t *pos;
    const uint16_t *flag;
    const uint8_t *mapq;
    const uint8_t *xs;
    const int32_t *l_qseq;
    const int32_t *mtid;
    const int32_t *mpos;
    const uint32_t *cig_off;
    const uint32_t *cigar;
    const uint32_t *seq_off;
    const uint8_t *seq4;
    const u64 *name_hash; // --extra only (nullptr otherwise)
    int64_t n;
    uint32_t base;      // global read ordinal of record 0 within the contig
    uint32_t tile_base; // first K1 tile index of this batch
    int32_t prev_pos;   // pos of the last record of the previous batch (sortedness across batches)
    int32_t member;     // the batch's target: its index among the chain's members (GroupTab)
    const int32_t *prev_pos_ptr; // where that position is, when only the device knows it (nullptr: prev_pos holds it)
    const uint32_t *seq2;        // pjb_batch.seq2 seen as words (two 16-bit granules each; nullptr: the batch has 4-bit bases only)
    const uint32_t *seq_exc;     // pjb_batch.seq_exc
};

// error word: min over (ordinal << 8 | -code); ~0 = no error
__device__ inline void set_error(u64 *err, u32 ordinal, int code) {
    atomicMin(err, ((u64)ordinal << 8) | (u64)(u32)(-code));
}

struct TileStats { // per K1 tile
    u32 spliced, unspliced;
    u64 sum_len;
    int32_t min_len, max_len;
    int32_t max_end;   // max(pos + alignedLength)
    int32_t max_nlen;  // longest N op
    int32_t min_pos;
    int32_t _pad;
};

// Per-contig control block in device memory.  The host sizes buffers and grids from LIMITS it chooses before anything
// runs (pairs, junctions, key format); the kernels read the actual counts from here, and a count that exceeds its limit
// raises an overflow bit and zeroes the count so that everything downstream does nothing.  pjb_finish_contig reads the
// block back once, at the end, and repeats the contig with larger limits if a bit is set.
enum : u32 { OVF_PAIRS = 1u, OVF_KEYFMT = 2u, OVF_JUNC = 4u, OVF_DENSE = 8u, OVF_LISTS = 16u };
struct ContigStats {
    u64 spliced, unspliced, sum_len;
    int32_t min_len, max_len;
    int32_t max_end, max_nlen, min_pos;
    u32 n_tiles;
    u64 n_pairs;   // pairs found (whatever the limit)
    u64 err;
    u32 n_junc, n_runs; // junctions / position runs found (whatever the limit)
    u32 P;         // pairs the pipeline works on: n_pairs, or 0 after an overflow
    u32 J, R;      // junctions / runs the pipeline works on
    u32 n_slots;   // J + ceil(P / 64) fragment slots
    u32 overflow;  // OVF_*
    u32 n_cand;    // K2d: keys in the candidate list (every junction at least once, few of them more often)
    u32 n_slices;  // ceil(P / 64): 64-pair slices of the sorted pair array (fragments, run masks)
    u32 list_need; // OVF_LISTS: the fullest sub-list of the read lists (EmitLists) wanted this many entries
};

// A pair = one N operation walked (JunctionSystem::addJunctions, junction_system.cc:140-210).  k1_emit writes, in BAM order,
// the pair's intron key (its own array: kd_assign and the sort's first pass stream over the keys alone) and ONE 32-byte record
// with everything the per-junction reductions need; every later kernel that works in sorted order fetches a pair with one
// 32-byte gather (two 16-byte loads from one sector).
struct __attribute__((aligned(16))) PairRec {
    u64 aux;         // per-pair match statistics (pack_res): written by k1_emit for the [S] M N M [S] shape, by k4b_generic for the rest
    int32_t lstart;  // lStart  (left anchor start of this pair)
    int32_t rend;    // rEndExc-1
    int32_t pos;     // read position   (entropy / distinct-alignment runs)
    int32_t aend;    // read end = pos + alignedLength - 1
    u32 meta;        // bit field, see META_*
    u32 updown;      // upjuncs | downjuncs << 16
};
static_assert(sizeof(PairRec) == 32, "PairRec is one 32-byte sector");
struct Pairs {
    u64 *key;     // packed intron key (see make_key), BAM order
    PairRec *rec; // BAM order
    u32 *g;       // global read ordinal of the pair's record -- written for PJB_FLAG_EXTRA contexts only (nullptr otherwise)
};
__device__ __forceinline__ void rec_store(PairRec *dst, const PairRec &r) {
    uint4 *q = reinterpret_cast<uint4 *>(dst);
    q[0] = make_uint4((u32)r.aux, (u32)(r.aux >> 32), (u32)r.lstart, (u32)r.rend);
    q[1] = make_uint4((u32)r.pos, (u32)r.aend, r.meta, r.updown);
}
__device__ __forceinline__ PairRec rec_load(const PairRec *src) {
    const uint4 *q = reinterpret_cast<const uint4 *>(src);
    const uint4 a = q[0], b = q[1];
    PairRec r;
    r.aux = (u64)a.x | ((u64)a.y << 32);
    r.lstart = (int32_t)a.z;
    r.rend = (int32_t)a.w;
    r.pos = (int32_t)b.x;
    r.aend = (int32_t)b.y;
    r.meta = b.z;
    r.updown = b.w;
    return r;
}

enum : u32 {
    META_CAT_MASK = 3u,      // 0 r1pos, 1 r1neg, 2 r2pos, 3 r2neg   (junction.cc:483-498)
    META_MULTI = 1u << 2,    // read has > 1 N op                    (junction.cc:499)
    META_XS_SHIFT = 3,       // 2 bits: 0 unknown, 1 '+', 2 '-'
    META_UM = 1u << 5,       // mapq >= 30                           (junction.cc:773)
    META_BPP = 1u << 6,      // BAM proper-pair flag                 (junction.cc:780)
    META_PPP = 1u << 7,      // calcIfProperPair                     (junction.cc:784)
    META_REL = 1u << 8,      // reliable                             (junction.cc:792)
    META_SIMPLE = 1u << 9,   // CIGAR is [S] M N M [S] and l_qseq matches it: both anchors are single contiguous compares
                             // that do not depend on the junction-level window (k1_emit compares them itself)
};
// per-pair match statistics packed in 64 bits: minMatch | mmes << 20 | mismatches << 40
__device__ __forceinline__ u64 pack_res(u32 minMatch, u32 mmes, u32 mis) {
    return (u64)(minMatch & 0xfffffu) | ((u64)(mmes & 0xfffffu) << 20) | ((u64)mis << 40);
}
constexpr u32 RES_FIELD_MAX = 0xfffffu; // anchors longer than this take the generic path

// ---------------------------------------------------------------------------------------------
// Target GROUPS ("super-chains").  A chain of 45 kernels over one 8 M-read target leaves most of the chip idle in most of
// its kernels; several targets finished together are ONE chain over a virtual sequence in which member i occupies
// [voff_i, voff_i + len_i) (offsets 64-aligned, a gap between members).  k1_emit adds the offset to every coordinate it
// emits, so keys, sort, grouping, anchors and reductions never see the difference -- an intron key still names one
// junction of one target, and key order is (member, start, end).  Only what touches a target's OWN data converts back:
// the genome of a pair / junction (k1_generic, k4b_generic, k5_finalize look the member up by index / position) and the rows
// (refid, local coordinates).  A single target is a group of one with offset 0.
// ---------------------------------------------------------------------------------------------
constexpr int GROUP_MAX = 32;
constexpr int32_t GROUP_GAP = 4096;
struct GroupTab {
    int32_t n;
    int32_t voff[GROUP_MAX]; // ascending
    int32_t len[GROUP_MAX];
    int32_t tid[GROUP_MAX];
    const uint8_t *d[GROUP_MAX];   // upper-cased bases
    const u32 *codes[GROUP_MAX];   // 4-bit codes (nullptr: exotic member)
    const u32 *codes2[GROUP_MAX];  // 2-bit codes and, behind them, the bitmap of the 64-base stretches that hold a character outside ACGT
                                   // (k0_encode2; nullptr with codes)
    u32 exc_members;               // bit m: member m's bitmap has a bit set at all (else k1_emit does not look at it)
};
struct Member {
    int32_t idx, voff, len, tid;
    const uint8_t *d;
    const u32 *codes;
};
__device__ __forceinline__ Member member_of(const GroupTab &T, int32_t vpos) {
    int m = 0;
    if (T.n > 1) {
#pragma unroll
        for (int s = GROUP_MAX / 2; s >= 1; s >>= 1)
            if (m + s < T.n && T.voff[m + s] <= vpos) m += s;
    }
    Member M;
    M.idx = m;
    M.voff = T.voff[m];
    M.len = T.len[m];
    M.tid = T.tid[m];
    M.d = T.d[m];
    M.codes = T.codes[m];
    return M;
}
// per-member counters of a group (what pjb_region_result reports per target)
struct MemberStats {
    u64 spliced, unspliced, sum_len, n_pairs;
    int32_t min_len, max_len;
    u32 n_junc, _pad;
};

// key packing: normal case (start << lbits) | intron_len, fallback raw (start << 32) | (u32)end
struct KeyFmt {
    int raw;   // 1 = raw 64-bit (weird coordinates present)
    int lbits; // bits of intron length
    int total_bits;
};
__device__ __host__ inline u64 make_key(const KeyFmt &f, int32_t istart, int32_t iend) {
    if (f.raw) return ((u64)(u32)istart << 32) | (u64)(u32)iend;
    return ((u64)(u32)istart << f.lbits) | (u64)(u32)(iend - istart + 1);
}
__device__ __host__ inline void unpack_key(const KeyFmt &f, u64 k, int32_t &istart, int32_t &iend) {
    if (f.raw) {
        istart = (int32_t)(u32)(k >> 32);
        iend = (int32_t)(u32)k;
    } else {
        istart = (int32_t)(u32)(k >> f.lbits);
        iend = istart + (int32_t)(u32)(k & ((1ull << f.lbits) - 1)) - 1;
    }
}

// CIGAR op classes by BAM op code "MIDNSHP=XB" (bam_alignment.hpp:75-99)
__device__ __forceinline__ bool op_consumes_ref(u32 op) { return (0x18Du >> op) & 1u; }   // M D N = X
__device__ __forceinline__ bool op_consumes_query(u32 op) { return (0x193u >> op) & 1u; } // M I S = X
enum : u32 { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };

// ---------------------------------------------------------------------------------------------
// wave / block primitives (wave = 64 lanes)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// A barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence as well: hipcc drains vmcnt before it, which
// makes a wave wait for every load it has in flight and for its STORES to be acknowledged -- k1_emit keeps the next trip's loads in
// flight across its barriers on purpose.  Nothing that other waves of the block read from global memory may depend on this.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// A pointer that was READ from memory (a batch descriptor, a pair's sequence address) is a generic pointer to the compiler:
// its loads are flat_load, which count against the LDS counter too and are waited for one by one.  Everything such pointers
// name here is device memory: gload reads through a global pointer.
template <class T>
__device__ __forceinline__ T gload(const T *p) {
    T v;
    __builtin_memcpy(&v, (const __attribute__((address_space(1))) void *)p, sizeof(T));
    return v;
}
// A batch descriptor fetched from device memory (one launch per chain: the block looks its batch up): the same fields as
// pointers into GLOBAL memory, so that what is read through them are global_load / s_load instructions.
#define PJB_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const PJB_GLOBAL T *as_global(const T *p) {
    return (const PJB_GLOBAL T *)p;
}
template <class T, class U>
__device__ __forceinline__ T gload_as(const PJB_GLOBAL U *p) { // a T at a global address
    T v;
    __builtin_memcpy(&v, (const PJB_GLOBAL void *)p, sizeof(T));
    return v;
}
struct GBatch {
    const PJB_GLOBAL int32_t *pos;
    const PJB_GLOBAL uint16_t *flag;
    const PJB_GLOBAL uint8_t *mapq;
    const PJB_GLOBAL uint8_t *xs;
    const PJB_GLOBAL int32_t *l_qseq;
    const PJB_GLOBAL int32_t *mtid;
    const PJB_GLOBAL int32_t *mpos;
    const PJB_GLOBAL uint32_t *cig_off;
    const PJB_GLOBAL uint32_t *cigar;
    const PJB_GLOBAL uint32_t *seq_off;
    const PJB_GLOBAL uint8_t *seq4;
    int64_t n;
    uint32_t base, tile_base;
    int32_t prev_pos, member;
    const PJB_GLOBAL int32_t *prev_pos_ptr;
    const PJB_GLOBAL uint32_t *seq2, *seq_exc;
};
#define PJB_CONSTANT __attribute__((address_space(4)))
__device__ __forceinline__ GBatch load_batch(const DevBatch *d) { // (through the constant address space: a uniform index gives scalar loads)
    const PJB_CONSTANT DevBatch *g = (const PJB_CONSTANT DevBatch *)d;
    GBatch b;
    b.pos = as_global(g->pos);
    b.flag = as_global(g->flag);
    b.mapq = as_global(g->mapq);
    b.xs = as_global(g->xs);
    b.l_qseq = as_global(g->l_qseq);
    b.mtid = as_global(g->mtid);
    b.mpos = as_global(g->mpos);
    b.cig_off = as_global(g->cig_off);
    b.cigar = as_global(g->cigar);
    b.seq_off = as_global(g->seq_off);
    b.seq4 = as_global(g->seq4);
    b.n = g->n;
    b.base = g->base;
    b.tile_base = g->tile_base;
    b.prev_pos = g->prev_pos;
    b.member = g->member;
    b.prev_pos_ptr = as_global(g->prev_pos_ptr);
    b.seq2 = as_global(g->seq2);
    b.seq_exc = as_global(g->seq_exc);
    return b;
}
// four consecutive words at a 4-byte aligned address (global_load_dwordx4 accepts that)
struct __attribute__((packed, aligned(4))) Words4 {
    u32 x, y, z, w;
};

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T t = __shfl_down(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T t = __shfl_down(v, o, 64);
        v = t < v ? t : v;
    }
    return v;
}
// inclusive scan across the wave
template <typename T>
__device__ __forceinline__ T wave_iscan(T v) {
    int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        T t = __shfl_up(v, o, 64);
        if (l >= o) v += t;
    }
    return v;
}

// Whole-wave reductions on the DPP path (no LDS traffic, one VALU instruction per step; __shfl_* goes through
// ds_bpermute): quads, half rows, rows of 16, then row_bcast:15 / row_bcast:31 carry the row totals up -- the result
// is in lane 63 and is read back as a scalar.  `IDENT` is what lanes that a step does not write contribute.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u32 dpp_move(u32 ident, u32 v) {
    return (u32)__builtin_amdgcn_update_dpp((int)ident, (int)v, CTRL, ROW_MASK, 0xf, false);
}
struct DppAdd {
    static constexpr u32 ident = 0u;
    __device__ __forceinline__ static u32 op(u32 a, u32 b) { return a + b; }
};
struct DppMax {
    static constexpr u32 ident = 0u;
    __device__ __forceinline__ static u32 op(u32 a, u32 b) { return a > b ? a : b; }
};
struct DppMin {
    static constexpr u32 ident = 0xffffffffu;
    __device__ __forceinline__ static u32 op(u32 a, u32 b) { return a < b ? a : b; }
};
template <typename Op>
__device__ __forceinline__ u32 wave_total(u32 v) {
    v = Op::op(v, dpp_move<0xB1, 0xf>(Op::ident, v));  // quad_perm [1,0,3,2]
    v = Op::op(v, dpp_move<0x4E, 0xf>(Op::ident, v));  // quad_perm [2,3,0,1]
    v = Op::op(v, dpp_move<0x141, 0xf>(Op::ident, v)); // row_half_mirror
    v = Op::op(v, dpp_move<0x140, 0xf>(Op::ident, v)); // row_mirror: every lane of a row holds the row's result
    v = Op::op(v, dpp_move<0x142, 0xa>(Op::ident, v)); // row_bcast:15 into rows 1 and 3
    v = Op::op(v, dpp_move<0x143, 0xc>(Op::ident, v)); // row_bcast:31 into rows 2 and 3
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}

// exclusive scan over the 256 threads of a block in thread order; returns exclusive prefix, total in *total.
// smem: at least 4 elements of T.  Contains __syncthreads (call uniformly).
template <typename T>
__device__ __forceinline__ T block_escan_256(T v, T *smem, T *total) {
    T inc = wave_iscan(v);
    int w = threadIdx.x >> 6, l = lane_id();
    __syncthreads();
    if (l == 63) smem[w] = inc;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        T s = smem[i];
        if (i < w) base += s;
        tot += s;
    }
    *total = tot;
    return base + inc - v;
}

// the same over NW wavefronts (NW = 1: no barrier, no shared memory traffic)
template <int NW, typename T, bool LDS_ONLY = false>
__device__ __forceinline__ T block_escan(T v, T *smem, T *total) {
    T inc = wave_iscan(v);
    if constexpr (NW == 1) {
        *total = __shfl(inc, 63, 64);
        return inc - v;
    } else {
        int w = threadIdx.x >> 6, l = lane_id();
        if constexpr (LDS_ONLY) lds_barrier();
        else __syncthreads();
        if (l == 63) smem[w] = inc;
        if constexpr (LDS_ONLY) lds_barrier();
        else __syncthreads();
        T base = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) {
            T s = smem[i];
            if (i < w) base += s;
            tot += s;
        }
        *total = tot;
        return base + inc - v;
    }
}

// ---------------------------------------------------------------------------------------------
// generic multi-block exclusive scan of u64 values produced by a functor (3 kernels)
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_TILE = 2048; // 256 threads x 8

template <typename F>
__global__ __launch_bounds__(256) void scan_reduce_kernel(F f, u64 n, u64 *tile_sums, const u32 *np) {
    __shared__ u64 sm[4];
    if (np) { // length known on the device only: the grid covers the host's limit, and a count beyond it (there is none) must not reach past the buffers
        const u64 d = *np;
        n = d < n ? d : n;
    }
    u64 base = (u64)blockIdx.x * SCAN_TILE;
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        u64 i = base + (u64)k * 256 + threadIdx.x;
        if (i < n) s += f(i); // (fetching all eight terms first, unconditionally, as k1_count does, changed nothing here: 49 vs 45 us)
    }
    s = wave_sum(s);
    if (lane_id() == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// single block: in-place exclusive scan of tile sums; writes grand total to *total
// (a template only so that every translation unit that scans -- chain, extra, ingest -- instantiates it for itself)
template <int UNUSED = 0>
__global__ __launch_bounds__(1024) void scan_tiles_kernel(u64 *tile_sums, u32 n_tiles, u64 *total) {
    __shared__ u64 wsum[16];
    __shared__ u64 carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (u32 base = 0; base < n_tiles; base += 1024) {
        u32 i = base + threadIdx.x;
        u64 v = i < n_tiles ? tile_sums[i] : 0;
        u64 inc = wave_iscan(v);
        int w = threadIdx.x >> 6;
        if (lane_id() == 63) wsum[w] = inc;
        __syncthreads();
        u64 wb = 0, tot = 0;
        for (int k = 0; k < 16; k++) {
            u64 s = wsum[k];
            if (k < w) wb += s;
            tot += s;
        }
        u64 carry = carry_s;
        if (i < n_tiles) tile_sums[i] = carry + wb + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

// third kernel: recompute values, exclusive prefix handed to the sink g(i, value, exclusive_prefix)
template <typename F, typename G>
__global__ __launch_bounds__(256) void scan_apply_kernel(F f, G g, u64 n, const u64 *tile_sums, const u32 *np) {
    __shared__ u64 sm[4];
    if (np) {
        const u64 d = *np;
        n = d < n ? d : n;
    }
    if ((u64)blockIdx.x * SCAN_TILE >= n) return;
    u64 base = (u64)blockIdx.x * SCAN_TILE;
    u64 run = tile_sums[blockIdx.x];
    // thread order within the tile must equal element order: round k covers [base+k*256, +256)
#pragma unroll
    for (int k = 0; k < 8; k++) {
        u64 i = base + (u64)k * 256 + threadIdx.x;
        u64 v = i < n ? f(i) : 0;
        u64 tot;
        u64 ex = block_escan_256<u64>(v, sm, &tot);
        if (i < n) g(i, v, run + ex);
        run += tot;
    }
}

// ---------------------------------------------------------------------------------------------
// The same scan in TWO kernels for contig-sized inputs: every block of the apply kernel adds up the sums of the
// tiles before it itself (at most SCAN2_MAX_TILES plain loads from an L2-resident array, 16 per thread) instead of
// waiting for a single-block kernel in between -- one dependent launch less per scan.  (A one-kernel scan with
// decoupled look-back was built and measured: 2 600 resident tiles polling each other's granules through the fabric
// cost 47-76 us against 29 us for three kernels, with or without fences; it is in the history, not in the tree.)
// ---------------------------------------------------------------------------------------------
constexpr u32 SCAN2_MAX_TILES = 4096;
template <typename F, typename G>
__global__ __launch_bounds__(256) void scan_apply2_kernel(F f, G g, u64 n, const u64 *tile_sums, const u32 *np, u64 *total) {
    __shared__ u64 sm[4];
    __shared__ u64 s_pref[4];
    if (np) {
        const u64 d = *np;
        n = d < n ? d : n;
    }
    const u64 n_tiles = n == 0 ? 1 : (n + SCAN_TILE - 1) / SCAN_TILE; // (an empty input still gets its total written)
    if (blockIdx.x >= n_tiles) return;
    u64 acc = 0;
    for (u32 t = threadIdx.x; t < blockIdx.x; t += 256) acc += tile_sums[t];
    acc = wave_sum(acc);
    if (lane_id() == 0) s_pref[threadIdx.x >> 6] = acc;
    __syncthreads();
    u64 run = s_pref[0] + s_pref[1] + s_pref[2] + s_pref[3];
    if (blockIdx.x == n_tiles - 1 && threadIdx.x == 0) *total = run + tile_sums[blockIdx.x];
    const u64 base = (u64)blockIdx.x * SCAN_TILE;
    // thread order within the tile must equal element order: round k covers [base+k*256, +256)
#pragma unroll
    for (int k = 0; k < 8; k++) {
        u64 i = base + (u64)k * 256 + threadIdx.x;
        u64 v = i < n ? f(i) : 0;
        u64 tot;
        u64 ex = block_escan_256<u64>(v, sm, &tot);
        if (i < n) g(i, v, run + ex);
        run += tot;
    }
}

// ---------------------------------------------------------------------------------------------
// K0: upper-case contig bases in place (boost::to_upper on fetched strings, junction.cc:586-587,635-638)
// ---------------------------------------------------------------------------------------------
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void k0_upper(uint8_t *g, int64_t n, int do_upper, int *has_x) {
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    bool x = false;
    for (int64_t j = i; j < n && j < i + 16; j++) {
        uint8_t c = g[j];
        x |= (c == 'X' || (do_upper && c == 'x'));
    }
    if (__ballot(x) && lane_id() == 0) atomicOr(has_x, 1);
    if (!do_upper) return;
    if (i + 16 <= n && ((uintptr_t)(g + i) & 15) == 0) {
        uint4 v = *reinterpret_cast<uint4 *>(g + i);
        u32 *w = reinterpret_cast<u32 *>(&v);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32 x = w[k], r = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                u32 c = (x >> (8 * b)) & 0xff;
                if (c >= 'a' && c <= 'z') c -= 32;
                r |= c << (8 * b);
            }
            w[k] = r;
        }
        *reinterpret_cast<uint4 *>(g + i) = v;
    } else {
        for (int64_t j = i; j < n && j < i + 16; j++) {
            uint8_t c = g[j];
            if (c >= 'a' && c <= 'z') g[j] = c - 32;
        }
    }
}
#endif // PJB_KERNELS_CHAIN

// K0f: the sequence lines of one FASTA record, as they are in the file, -> its bases.  With the .fai's geometry (line_blen
// bases per line, line_len bytes per line) base i sits at byte (i / line_blen) * line_len + i % line_blen; every base must be
// a graphic character and every line terminator byte not one -- the test faidx's loader implies (deps/htslib-1.3/faidx.c
// reads with isgraph).  Anything else raises `bad` and the host falls back to filtering the characters itself.
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void k0_fasta(const uint8_t *raw, int64_t raw_bytes, int64_t n, int32_t line_blen, int32_t line_len,
                                                 uint8_t *out, int *bad) {
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i0 >= n) return;
    int64_t line = i0 / line_blen;
    int32_t col = (int32_t)(i0 - line * line_blen);
    int64_t src = line * line_len + col;
    bool wrong = false;
    const int64_t i1 = i0 + 16 < n ? i0 + 16 : n;
    for (int64_t i = i0; i < i1; i++) {
        uint8_t ch = 0;
        if (src < raw_bytes) ch = raw[src];
        else wrong = true;
        wrong |= ch <= 32 || ch >= 127;
        out[i] = ch;
        src++;
        if (++col == line_blen) { // end of a full line: its terminator bytes follow (if another base does)
            if (i + 1 < n)
                for (int32_t k = 0; k < line_len - line_blen; k++) {
                    const uint8_t t = src + k < raw_bytes ? raw[src + k] : (uint8_t)'?';
                    wrong |= t > 32 && t < 127;
                }
            src += line_len - line_blen;
            col = 0;
        }
    }
    if (__ballot(wrong) && lane_id() == 0) atomicOr(bad, 1);
}
#endif // PJB_KERNELS_CHAIN

// K0b: contig bases -> 4-bit nt16 codes, two per byte, LOW nibble first (base i at bits 4*(i&7) of
// word i>>3).  A byte outside the 16-letter alphabet "=ACMGRSVTWYHKDBN" has no code: the contig is
// flagged "exotic" and k4 then uses its byte-wise path, so equality semantics stay exact.
__device__ __forceinline__ u32 nt16_code(u32 c, bool &exotic) {
    switch (c) {
    case '=': return 0;
    case 'A': return 1;
    case 'C': return 2;
    case 'M': return 3;
    case 'G': return 4;
    case 'R': return 5;
    case 'S': return 6;
    case 'V': return 7;
    case 'T': return 8;
    case 'W': return 9;
    case 'Y': return 10;
    case 'H': return 11;
    case 'K': return 12;
    case 'D': return 13;
    case 'B': return 14;
    case 'N': return 15;
    default: exotic = true; return 0;
    }
}
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void k0_encode(const uint8_t *g, int64_t n, u32 *codes, int64_t n_words, int *exotic_flag) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool exotic = false;
    if (w < n_words) {
        u32 out = 0;
        const int64_t b0 = w * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int64_t i = b0 + k;
            if (i < n) out |= nt16_code(g[i], exotic) << (4 * k);
        }
        codes[w] = out;
    }
    if (__ballot(exotic) && lane_id() == 0) atomicOr(exotic_flag, 1);
}
#endif // PJB_KERNELS_CHAIN

// K0c: the same bases in TWO bits (A 0, C 1, G 2, T 3; base i at bits 2 (i & 15) of word i >> 4; anything else is stored as 0) and
// the exceptions: bit i >> 6 of the bitmap is set when one of the bases 64 (i >> 6) .. 64 (i >> 6) + 63 is not A, C, G or T.  k1_emit
// compares a block of read bases in 2 bits only where the bitmap is clear under it (a read of pure ACGT against a stretch of pure
// ACGT: characters are equal exactly where the 2-bit codes are) and takes the 4-bit codes for every other block.  A thread owns one
// 64-base stretch: four words of codes (one 16-byte store), one bit; a wavefront's bits are one u64 of the bitmap.
//   Layout of the allocation: codes2[0 .. n2w) | K0_CODES2_PAD zero words | bitmap u32[(n + 2047) / 2048] | K0_GEXC_PAD zero words.
constexpr int K0_CODES2_PAD = 8, K0_GEXC_PAD = 4;
__host__ __device__ inline int64_t codes2_words(int64_t n) { return ((n + 63) / 64) * 4; }          // (whole stretches)
__host__ __device__ inline int64_t gexc_words(int64_t n) { return (((n + 63) / 64 + 63) / 64) * 2; } // (whole u64s)
__host__ __device__ inline int64_t codes2_alloc_words(int64_t n) { return codes2_words(n) + K0_CODES2_PAD + gexc_words(n) + K0_GEXC_PAD; }
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void k0_encode2(const uint8_t *g, int64_t n, u32 *codes2, u32 *gexc, int *any_exc) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; // stretch
    const int64_t n_str = (n + 63) / 64;
    bool exc = false;
    if (s < n_str) {
        u32 w[4] = {0, 0, 0, 0};
        const int64_t b0 = s * 64;
        if (b0 + 64 <= n && ((uintptr_t)(g + b0) & 15) == 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 v = *reinterpret_cast<const uint4 *>(g + b0 + 16 * q);
                const u32 x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const u32 c = (x[k >> 2] >> (8 * (k & 3))) & 0xffu;
                    const u32 code = c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u;
                    exc |= !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
                    w[q] |= code << (2 * k);
                }
            }
        } else {
            for (int k = 0; k < 64; k++) {
                const int64_t i = b0 + k;
                if (i < n) {
                    const u32 c = g[i];
                    const u32 code = c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u;
                    exc |= !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
                    w[k >> 4] |= code << (2 * (k & 15));
                }
            }
        }
        *reinterpret_cast<uint4 *>(codes2 + 4 * s) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    const u64 m = __ballot(exc);
    if (lane_id() == 0 && s < ((n_str + 63) / 64) * 64) reinterpret_cast<u64 *>(gexc)[s >> 6] = m;
    if (m && lane_id() == 0) atomicOr(any_exc, 1); // (a target without a single exception -- a telomere-to-telomere assembly, a bacterium -- is never asked)
}
#endif // PJB_KERNELS_CHAIN

// ---------------------------------------------------------------------------------------------
// K1a: per-read CIGAR walk, pass 1 (BamAlignment::init bam_alignment.cc:71-100, findJuncs length
// stats src/junction_builder.cc:333-343).  One thread per read; a tile is 1024 consecutive reads.
// ---------------------------------------------------------------------------------------------
constexpr int K1_TILE = 1024;

// --extra by-products of the first pass (pjb_extra.hip.h, "the sparse path"): which records belong to unspliced.bam and what
// they span -- k1_count has every record's CIGAR in registers anyway
struct SparseCounters { // one per target, device memory (zeroed)
    u32 n_zero;       // unspliced mapped records with no reference-consuming op (zlist entries)
    u32 max_span;     // longest reference span of an unspliced mapped record
    u32 max_gap;      // longest D operation among them
    u32 need_dense;   // bit 0: the pileup cap may bite; bit 1: a record with more than 126 gaps; bit 2: gap list full
    u64 total;        // written by the scan: unspliced records with a span | gaps << 32
};
constexpr u32 SPARSE_GAP_MAX = 126;
struct XOut {
    int32_t *s_pos, *s_end; // per record (global ordinal): position, exclusive end of the span (= pos: no span / not unspliced.bam)
    uint8_t *q;             // bit 0: has a span, bits 1-7: D operations
    u32 *zlist;
    u32 zcap;
    SparseCounters *cnt;
};

// chk_ref_len > 0 (members of a group): a tile with an alignment that ends past the target reports max_end = INT32_MAX, so
// that k1_scan_tiles sees "weird coordinates" whatever the group's virtual length is (the host then finishes the
// members one by one).
// spl_rec: per spliced read of the tile's list, what this pass holds in registers anyway and k1_emit would have to gather again --
// {first operation's index, position, first word of the bases, operations (15 bits) | bases present (bit 15) | l_qseq (16 bits)};
// an operation count or l_qseq that does not fit is stored as all-ones and fetched by k1_emit.
__device__ __forceinline__ u32 spl_nlq(u32 n, bool seq_ok, int32_t lq) {
    return (n < 0x7fffu ? n : 0x7fffu) | (seq_ok ? 0x8000u : 0u) | ((u32)lq < 0xffffu ? (u32)lq << 16 : 0xffff0000u);
}
#ifndef K1C_T
#define K1C_T 256 // threads of a k1_count block (a tile is K1_TILE reads: 4 per thread).  512 threads x 2 reads need 64 registers instead of
                  // 84 but took 88 against 63 us a launch beside the other chains' kernels: a block of 8 wavefronts waits for 8 free slots on ONE CU
#endif
#ifndef K1C_NO_ROWS
#define K1C_NO_ROWS 0 // 1: every tile takes the rounds (A/B builds)
#endif
#ifndef K1C_WAVES
#define K1C_WAVES 7 // waves per SIMD the register allocation aims at: 5 / 6 / 7 / 8 = 423 / 412 / 408 / 490 us a chain (8: 27 registers spilled)
#endif
// ONE launch per chain: block = one tile of the chain's tile space; its batch is the last one whose tile_base it has reached
// (the descriptors live in device memory, the index is uniform: scalar loads).  chk_members (groups): see chk_ref_len below.
__device__ __forceinline__ int batch_of_tile(const DevBatch *batches, int n_batches, u32 tile) {
    // lane k looks at batch base + k: one load and one ballot per 64 batches
    int found = 0;
    for (int base = 0; base < n_batches; base += 64) {
        const int k = base + lane_id();
        const u32 tb = k < n_batches ? gload(&batches[k].tile_base) : 0xffffffffu;
        const u64 m = __ballot(tb <= tile);
        if (m == 0) break;
        found = base + 63 - __clzll((long long)m);
    }
    return __builtin_amdgcn_readfirstlane(found);
}
// inclusive scan across the wave on the DPP path (row shifts, then the row totals carried up: no LDS traffic, ~14 instructions)
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ u32 dpp_shift0(u32 v) { // the source lane's value, 0 where there is none
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ u32 wave_iscan_dpp(u32 v) {
    u32 o = v + dpp_shift0<0x111, 0xf, 0xf>(v) + dpp_shift0<0x112, 0xf, 0xf>(v) + dpp_shift0<0x113, 0xf, 0xf>(v); // row_shr:1..3
    o += dpp_shift0<0x114, 0xf, 0xe>(o); // row_shr:4
    o += dpp_shift0<0x118, 0xf, 0xc>(o); // row_shr:8
    o += dpp_shift0<0x142, 0xa, 0xf>(o); // row_bcast:15
    o += dpp_shift0<0x143, 0xc, 0xf>(o); // row_bcast:31
    return o;
}
#define PJB_LDS __attribute__((address_space(3)))
#ifndef K1C_OPSW
#define K1C_OPSW 3072 // operations of a tile that the fast path of k1_count keeps in LDS (a tile of 1024 reads has ~1 700)
#endif
// The FAST path of a tile (all 1024 reads there, its operations fit in LDS): a thread owns four CONSECUTIVE reads, so every
// fixed-width field is one 16-byte load (the kernel was bound by VALU issue -- 83 % busy, round 5's SQ counters: 16 address
// computations and word loads for the operations, four 64-bit shuffle scans), the tile's operations come with coalesced loads
// straight to LDS and are read from there, and ONE scan per wavefront orders the spliced reads.
template <bool EXTRA>
__device__ __forceinline__ void k1_count_rows(const GBatch &b, const u32 tile_local, const u32 cA, const u32 n_ops, u32 *tile_cnt, TileStats *tile_stats, u32 *spl_idx,
                                              u32 *spl_poff, uint4 *spl_rec, u64 *err, const int32_t chk_ref_len, const XOut &X) {
    constexpr int NW = K1C_T / 64;
    static_assert(K1C_T * 4 == K1_TILE, "four reads a thread");
    __shared__ __attribute__((aligned(16))) u32 s_ops[K1C_OPSW + 8];
    __shared__ u32 s_scan[2][NW];
    __shared__ u64 r64[NW][2];
    __shared__ int32_t r32[NW][6];
    const int t = threadIdx.x, w = t >> 6;
    const int64_t r0 = (int64_t)tile_local * K1_TILE + 4 * t; // the thread's first read
    // ---- the tile's operations -> LDS (16 bytes a lane straight to LDS; the span starts at the 16-byte boundary below its first word)
    const uintptr_t a0 = (uintptr_t)(b.cigar + cA);
    const u32 shift = (u32)(a0 >> 2) & 3u;
    {
        const PJB_GLOBAL uint4 *p = (const PJB_GLOBAL uint4 *)(a0 - 4u * shift);
        const u32 total = n_ops + shift; // words from the aligned start; they all exist but for the last 16 bytes' tail: the array ends at cig_off[n]
        const u32 exist = b.cig_off[b.n] - cA + shift;
#pragma unroll
        for (int it = 0; it < (K1C_OPSW + 3 + K1C_T * 4 - 1) / (K1C_T * 4); it++) {
            if ((u32)it * (K1C_T * 4) < total) {
                const u32 i = (u32)it * (K1C_T * 4) + (u32)t * 4u;
                if (i + 4u <= exist) {
                    if (i < total) __builtin_amdgcn_global_load_lds(p + (i >> 2), (PJB_LDS void *)(s_ops + it * (K1C_T * 4) + w * 256), 16, 0, 0);
                } else if (i < exist) {
                    const PJB_GLOBAL u32 *q = (const PJB_GLOBAL u32 *)p + i;
                    s_ops[i] = q[0];
                    if (i + 1u < exist) s_ops[i + 1] = q[1];
                    if (i + 2u < exist) s_ops[i + 2] = q[2];
                }
            }
        }
    }
    // ---- the four reads' fields: one 16-byte load each
    const Words4 co = gload_as<Words4>(b.cig_off + r0);
    const u32 co4 = b.cig_off[r0 + 4];
    const Words4 po = gload_as<Words4>(b.pos + r0);
    const int32_t pprev = r0 > 0 ? b.pos[r0 - 1] : (b.prev_pos_ptr ? *b.prev_pos_ptr : b.prev_pos);
    const Words4 lqv = gload_as<Words4>(b.l_qseq + r0);
    const Words4 sov = gload_as<Words4>(b.seq_off + r0);
    const u32 so4 = b.seq_off[r0 + 4];
    const u32 xs4 = gload_as<u32>(b.xs + r0);
    u64 fl4 = 0;
    if (EXTRA) fl4 = gload_as<u64>(b.flag + r0);
    const u32 c[5] = {co.x, co.y, co.z, co.w, co4}, so[5] = {sov.x, sov.y, sov.z, sov.w, so4};
    const int32_t pos[4] = {(int32_t)po.x, (int32_t)po.y, (int32_t)po.z, (int32_t)po.w}, lq[4] = {(int32_t)lqv.x, (int32_t)lqv.y, (int32_t)lqv.z, (int32_t)lqv.w};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the LDS-DMA: hipcc does not wait for it before a ds_read of its own accord)
    __syncthreads();
    u32 cnt = 0, spl = 0, uns = 0;
    u64 sum = 0;
    int32_t mn = INT32_MAX, mx = 0, max_end = 0, max_nlen = 0, min_pos = INT32_MAX;
    u32 cN[4], nlq[4];
    u32 xspan = 0, xgapmax = 0;
    bool xmany = false;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int64_t r = r0 + i;
        const int32_t p = pos[i], lenv = lq[i];
        const int32_t prv = i == 0 ? pprev : pos[i > 0 ? i - 1 : 0];
        if (p < prv) set_error(err, b.base + (u32)r, PJB_ERR_UNSORTED);
        if (((xs4 >> (8 * i)) & 0xffu) > 2u) set_error(err, b.base + (u32)r, PJB_ERR_BAD_XS);
        mn = lenv < mn ? lenv : mn;
        mx = lenv > mx ? lenv : mx;
        sum += (u64)(int64_t)lenv;
        const u32 n = c[i + 1] - c[i];
        nlq[i] = spl_nlq(n, (u64)(so[i + 1] - so[i]) * 8ull >= (u64)(int64_t)lenv, lenv);
        const u32 *ops = s_ops + (c[i] - cA + shift);
        u32 cc = 0, ngap = 0, gmax = 0;
        int32_t al = 0;
        auto count_op = [&](u32 op) {
            const u32 ty = op & 15u;
            const int32_t ln = (int32_t)(op >> 4);
            if (op_consumes_ref(ty)) al += ln;
            if (ty == OP_N) {
                cc++;
                max_nlen = ln > max_nlen ? ln : max_nlen;
            }
            if (EXTRA && ty == OP_D && ln) { // inside the span, no depth
                ngap++;
                gmax = max(gmax, (u32)ln);
            }
        };
        const u32 o0 = ops[0], o1 = ops[1], o2 = ops[2], o3 = ops[3]; // (what lies behind the read's last operation is masked: 0M has no effect)
        count_op(0u < n ? o0 : 0u);
        count_op(1u < n ? o1 : 0u);
        count_op(2u < n ? o2 : 0u);
        count_op(3u < n ? o3 : 0u);
        for (u32 k = 4; k < n; k++) count_op(ops[k]);
        cnt += cc;
        if (cc) {
            spl++;
            const int32_t e = p + al;
            max_end = e > max_end ? e : max_end;
            min_pos = p < min_pos ? p : min_pos;
        } else
            uns++;
        cN[i] = cc;
        if (EXTRA) { // the record's part in unspliced.bam (junction_builder.cc:168-186)
            const u32 g = b.base + (u32)r;
            const u32 fv = (u32)(fl4 >> (16 * i)) & 0xffffu;
            const bool unspliced = cc == 0 && !(fv & 0x4u);
            const bool spans = unspliced && al > 0 && p >= 0;
            X.s_pos[g] = p;
            X.s_end[g] = spans ? p + al : p;
            if (!spans) ngap = 0, gmax = 0;
            if (ngap > SPARSE_GAP_MAX) xmany = true, ngap = SPARSE_GAP_MAX;
            X.q[g] = (uint8_t)((spans ? 1u : 0u) | (ngap << 1));
            if (spans) xspan = max(xspan, (u32)al);
            xgapmax = max(xgapmax, gmax);
            if (unspliced && al == 0) {
                const u32 z = atomicAdd(&X.cnt->n_zero, 1u);
                if (z < X.zcap) X.zlist[z] = (u32)p;
            }
        }
    }
    if (EXTRA) {
        xspan = wave_max(xspan);
        xgapmax = wave_max(xgapmax);
        if (lane_id() == 0) { // (look first: the maxima settle after a few waves)
            if (xspan > X.cnt->max_span) atomicMax(&X.cnt->max_span, xspan);
            if (xgapmax > X.cnt->max_gap) atomicMax(&X.cnt->max_gap, xgapmax);
        }
        if (xmany) atomicOr(&X.cnt->need_dense, 2u);
    }
    // ---- ordered compaction of the tile's spliced reads: slot k holds the k-th of them (batch-local index) and the tile-local offset
    // of its first pair.  Thread order is read order: one scan of the threads' counts.
    {
        const u32 inc_s = wave_iscan_dpp(spl), inc_p = wave_iscan_dpp(cnt);
        if (lane_id() == 63) {
            s_scan[0][w] = inc_s;
            s_scan[1][w] = inc_p;
        }
        __syncthreads();
        u32 ex_s = inc_s - spl, ex_p = inc_p - cnt;
#pragma unroll
        for (int k = 0; k < NW; k++)
            if (k < w) {
                ex_s += s_scan[0][k];
                ex_p += s_scan[1][k];
            }
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (cN[i]) {
                const size_t slot = (size_t)blockIdx.x * K1_TILE + ex_s;
                spl_idx[slot] = (u32)(r0 + i);
                spl_poff[slot] = ex_p;
                spl_rec[slot] = make_uint4(c[i], (u32)pos[i], so[i], nlq[i]);
                ex_s++;
                ex_p += cN[i];
            }
    }
    // ---- the tile's statistics (whole-wave reductions on the DPP path; the length sum in 16-bit halves: nothing overflows 32 bits)
    const u32 w_cnt = wave_total<DppAdd>(cnt), w_su = wave_total<DppAdd>((spl << 16) | uns);
    sum = (u64)wave_total<DppAdd>((u32)(sum & 0xffffu)) + ((u64)wave_total<DppAdd>((u32)((sum >> 16) & 0xffffu)) << 16) +
          ((u64)wave_total<DppAdd>((u32)(sum >> 32)) << 32);
    auto smin = [](int32_t v) { return (int32_t)(wave_total<DppMin>((u32)v ^ 0x80000000u) ^ 0x80000000u); };
    auto smax = [](int32_t v) { return (int32_t)(wave_total<DppMax>((u32)v ^ 0x80000000u) ^ 0x80000000u); };
    mn = smin(mn);
    mx = smax(mx);
    max_end = smax(max_end);
    max_nlen = smax(max_nlen);
    min_pos = smin(min_pos);
    if (lane_id() == 0) {
        r64[w][0] = ((u64)w_cnt << 32) | w_su;
        r64[w][1] = sum;
        r32[w][0] = mn;
        r32[w][1] = mx;
        r32[w][2] = max_end;
        r32[w][3] = max_nlen;
        r32[w][4] = min_pos;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 pk = 0;
        TileStats ts;
        ts.sum_len = 0;
        ts.min_len = INT32_MAX, ts.max_len = 0, ts.max_end = 0, ts.max_nlen = 0, ts.min_pos = INT32_MAX;
        for (int i = 0; i < NW; i++) {
            pk += r64[i][0];
            ts.sum_len += r64[i][1];
            ts.min_len = min(ts.min_len, r32[i][0]);
            ts.max_len = max(ts.max_len, r32[i][1]);
            ts.max_end = max(ts.max_end, r32[i][2]);
            ts.max_nlen = max(ts.max_nlen, r32[i][3]);
            ts.min_pos = min(ts.min_pos, r32[i][4]);
        }
        ts.spliced = (u32)((pk >> 16) & 0xffff);
        ts.unspliced = (u32)(pk & 0xffff);
        if (chk_ref_len > 0 && ts.max_end > chk_ref_len) ts.max_end = INT32_MAX;
        ts._pad = 0;
        tile_stats[blockIdx.x] = ts;
        tile_cnt[blockIdx.x] = (u32)(pk >> 32);
    }
}
template <bool EXTRA>
__global__ __launch_bounds__(K1C_T) __attribute__((amdgpu_waves_per_eu(K1C_WAVES, K1C_WAVES))) void k1_count(const DevBatch *batches, int n_batches, u32 *tile_cnt, TileStats *tile_stats, u32 *spl_idx,
                                                 u32 *spl_poff, uint4 *spl_rec, u64 *err, GroupTab G, int chk_members, XOut X) {
    constexpr int NW = K1C_T / 64, RPT = K1_TILE / K1C_T; // wavefronts of the block; reads per thread
    __shared__ u64 sm64[NW];
    __shared__ u64 sm_scan4[RPT][NW];
    __shared__ int32_t smi[NW][6];
    const GBatch b = load_batch(batches + batch_of_tile(batches, n_batches, blockIdx.x));
    const u32 tile_local = blockIdx.x - b.tile_base;
    const int32_t chk_ref_len = chk_members ? max(G.len[b.member], 1) : 0;
    int64_t base = (int64_t)tile_local * K1_TILE;
#if !K1C_NO_ROWS
    if (base + K1_TILE <= b.n) { // (a whole tile whose operations fit in LDS: the fast path; else the rounds below)
        const u32 cA = b.cig_off[base], cB = b.cig_off[base + K1_TILE];
        if (cB - cA + 3u <= (u32)K1C_OPSW) {
            k1_count_rows<EXTRA>(b, tile_local, cA, cB - cA, tile_cnt, tile_stats, spl_idx, spl_poff, spl_rec, err, chk_ref_len, X);
            return;
        }
    }
#endif
    u32 cnt = 0, spl = 0, uns = 0;
    u64 sum = 0;
    int32_t mn = INT32_MAX, mx = 0, max_end = 0, max_nlen = 0, min_pos = INT32_MAX;
    // all loads of the thread's 4 reads are issued before anything is consumed: offsets, then the first
    // K1_OPS ops of every CIGAR (longer CIGARs continue from global memory), then the per-read scalars
    constexpr int K1_OPS = 4;
    u32 c0[RPT], nop[RPT], ops[RPT][K1_OPS];
    int32_t pos4[RPT];
    u32 c4[RPT], so4[RPT], nlq4[RPT]; // (nlq4: spl_nlq -- what the spliced list keeps of operations count, l_qseq and the presence of bases)
    u32 bad = 0;                // bit it: read `it` lies before its predecessor; bit 4 + it: its XS code is not one
    u32 xflag4[RPT] = {}, xspan = 0, xgapmax = 0; // (EXTRA)
    bool xmany = false;
    // (Every load below is UNCONDITIONAL -- a lane past the batch's end reads the last record, an operation past a CIGAR's end
    // reads a word that is always there -- and the value is masked afterwards.  Written as `cond ? load : 0` the compiler
    // may not speculate the load, turns it into a branch and waits for each one before issuing the next: the 16 CIGAR loads
    // of a thread were 16 round trips in a row.)
#pragma unroll
    for (int it = 0; it < RPT; it++) {
        const int64_t r = base + it * K1C_T + threadIdx.x;
        const bool on = r < b.n;
        const int64_t rr = on ? r : b.n - 1, rp = rr > 0 ? rr - 1 : 0;
        const u32 c0v = b.cig_off[rr], c1v = b.cig_off[rr + 1];
        const int32_t posv = b.pos[rr], prevv = b.pos[rp], lenv = b.l_qseq[rr];
        const u32 xsv = (u32)b.xs[rr], sov = b.seq_off[rr], so1v = b.seq_off[rr + 1];
        // (the scalars are folded as they arrive -- the operations' loads below need the offsets of this same batch of loads anyway --
        // so that only what the list and the walk need stays in registers: 58 of them, eight tiles a CU)
        so4[it] = sov;
        c0[it] = on ? c0v : 0u;
        nop[it] = on ? c1v - c0v : 0u;
        pos4[it] = on ? posv : 0;
        const int32_t prv = r > 0 ? prevv : (b.prev_pos_ptr ? *b.prev_pos_ptr : b.prev_pos);
        if (on && posv < prv) bad |= 1u << it;
        if (on && xsv > 2) bad |= 16u << it;
        if (on) {
            mn = lenv < mn ? lenv : mn;
            mx = lenv > mx ? lenv : mx;
            sum += (u64)(int64_t)lenv;
        }
        nlq4[it] = spl_nlq(c1v - c0v, (u64)(so1v - sov) * 8ull >= (u64)(int64_t)lenv, lenv);
        if (EXTRA) {
            const u32 fv = (u32)b.flag[rr];
            xflag4[it] = on ? fv : 0u;
        }
    }
    // (one 16-byte load per read instead of four words was measured: 55 against 52 us a launch -- consecutive reads' operations
    // are neighbours, the word loads coalesce)
#pragma unroll
    for (int it = 0; it < RPT; it++)
#pragma unroll
        for (int k = 0; k < K1_OPS; k++) {
            const bool has = (u32)k < nop[it];
            const u32 v = *(has ? b.cigar + c0[it] + k : b.cig_off);
            ops[it][k] = has ? v : 0u;
        }
#pragma unroll
    for (int it = 0; it < RPT; it++) {
        const int64_t r = base + it * K1C_T + threadIdx.x;
        u32 cthis = 0;
        if (r < b.n) {
            const int32_t p = pos4[it];
            if (bad & (1u << it)) set_error(err, b.base + (u32)r, PJB_ERR_UNSORTED);
            if (bad & (16u << it)) set_error(err, b.base + (u32)r, PJB_ERR_BAD_XS);
            u32 c = 0, ngap = 0, gmax = 0;
            int32_t al = 0;
            auto count_op = [&](u32 op) {
                const u32 ty = op & 15u;
                const int32_t ln = (int32_t)(op >> 4);
                if (op_consumes_ref(ty)) al += ln;
                if (ty == OP_N) {
                    c++;
                    max_nlen = ln > max_nlen ? ln : max_nlen;
                }
                if (EXTRA && ty == OP_D && ln) { // inside the span, no depth
                    ngap++;
                    gmax = max(gmax, (u32)ln);
                }
            };
#pragma unroll
            for (int k = 0; k < K1_OPS; k++) count_op(ops[it][k]); // padding ops are 0M: no effect
            for (u32 k = K1_OPS; k < nop[it]; k++) count_op(b.cigar[c0[it] + k]);
            cnt += c;
            if (c) {
                spl++;
                int32_t e = p + al;
                max_end = e > max_end ? e : max_end;
                min_pos = p < min_pos ? p : min_pos;
            } else
                uns++;
            cthis = c;
            if (EXTRA) { // the record's part in unspliced.bam (junction_builder.cc:168-186)
                const u32 g = b.base + (u32)r;
                const bool unspliced = c == 0 && !(xflag4[it] & 0x4u);
                const bool spans = unspliced && al > 0 && p >= 0;
                X.s_pos[g] = p;
                X.s_end[g] = spans ? p + al : p;
                if (!spans) ngap = 0, gmax = 0;
                if (ngap > SPARSE_GAP_MAX) xmany = true, ngap = SPARSE_GAP_MAX;
                X.q[g] = (uint8_t)((spans ? 1u : 0u) | (ngap << 1));
                if (spans) xspan = max(xspan, (u32)al);
                xgapmax = max(xgapmax, gmax);
                if (unspliced && al == 0) {
                    const u32 z = atomicAdd(&X.cnt->n_zero, 1u);
                    if (z < X.zcap) X.zlist[z] = (u32)p;
                }
            }
        }
        c4[it] = cthis;
    }
    if (EXTRA) {
        xspan = wave_max(xspan);
        xgapmax = wave_max(xgapmax);
        if (lane_id() == 0) { // (look first: the maxima settle after a few waves)
            if (xspan > X.cnt->max_span) atomicMax(&X.cnt->max_span, xspan);
            if (xgapmax > X.cnt->max_gap) atomicMax(&X.cnt->max_gap, xgapmax);
        }
        if (xmany) atomicOr(&X.cnt->need_dense, 2u);
    }
    // ordered compaction of the spliced reads of this tile: slot k holds the k-th spliced read
    // (batch-local index) and the tile-local offset of its first pair.  Read order is round-major
    // (r = base + it * 256 + thread): one wave scan per round, one barrier for all four.
    {
        u64 inc[RPT];
        const int w = threadIdx.x >> 6;
#pragma unroll
        for (int it = 0; it < RPT; it++) {
            inc[it] = wave_iscan<u64>(((u64)c4[it] << 16) | (u64)(c4[it] ? 1u : 0u));
            if (lane_id() == 63) sm_scan4[it][w] = inc[it];
        }
        __syncthreads();
        u64 run = 0; // (pairs << 16 | spliced reads) of everything before, in read order
#pragma unroll
        for (int it = 0; it < RPT; it++) {
            u64 before = run;
#pragma unroll
            for (int i = 0; i < NW; i++) {
                const u64 t = sm_scan4[it][i];
                if (i < w) before += t;
                run += t;
            }
            if (c4[it]) {
                const u64 ex = before + inc[it] - (((u64)c4[it] << 16) | 1u);
                const size_t slot = (size_t)blockIdx.x * K1_TILE + (u32)(ex & 0xffffu);
                spl_idx[slot] = (u32)(base + it * K1C_T + threadIdx.x);
                spl_poff[slot] = (u32)(ex >> 16);
                spl_rec[slot] = make_uint4(c0[it], (u32)pos4[it], so4[it], nlq4[it]);
            }
        }
    }
    // block reduce
    // whole-wave reductions on the DPP path (every lane gets the result); per wave: spl, uns <= 256, cnt <= 256 * 65535,
    // and the length sum goes in two 16-bit halves so that nothing can overflow 32 bits
    u64 packed = ((u64)wave_total<DppAdd>(cnt) << 32) | (u64)wave_total<DppAdd>((spl << 16) | uns);
    sum = (u64)wave_total<DppAdd>((u32)(sum & 0xffffu)) + ((u64)wave_total<DppAdd>((u32)((sum >> 16) & 0xffffu)) << 16) +
          ((u64)wave_total<DppAdd>((u32)(sum >> 32)) << 32);
    auto smin = [](int32_t v) { return (int32_t)(wave_total<DppMin>((u32)v ^ 0x80000000u) ^ 0x80000000u); };
    auto smax = [](int32_t v) { return (int32_t)(wave_total<DppMax>((u32)v ^ 0x80000000u) ^ 0x80000000u); };
    mn = smin(mn);
    mx = smax(mx);
    max_end = smax(max_end);
    max_nlen = smax(max_nlen);
    min_pos = smin(min_pos);
    int w = threadIdx.x >> 6;
    __shared__ u64 smp[NW];
    if (lane_id() == 0) {
        smp[w] = packed;
        sm64[w] = sum;
        smi[w][0] = mn;
        smi[w][1] = mx;
        smi[w][2] = max_end;
        smi[w][3] = max_nlen;
        smi[w][4] = min_pos;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 p = 0;
        TileStats t;
        t.sum_len = 0;
        t.min_len = INT32_MAX, t.max_len = 0, t.max_end = 0, t.max_nlen = 0, t.min_pos = INT32_MAX;
        for (int i = 0; i < NW; i++) {
            p += smp[i];
            t.sum_len += sm64[i];
            t.min_len = min(t.min_len, smi[i][0]);
            t.max_len = max(t.max_len, smi[i][1]);
            t.max_end = max(t.max_end, smi[i][2]);
            t.max_nlen = max(t.max_nlen, smi[i][3]);
            t.min_pos = min(t.min_pos, smi[i][4]);
        }
        t.spliced = (u32)((p >> 16) & 0xffff);
        t.unspliced = (u32)(p & 0xffff);
        if (chk_ref_len > 0 && t.max_end > chk_ref_len) t.max_end = INT32_MAX;
        t._pad = 0;
        tile_stats[blockIdx.x] = t;
        tile_cnt[blockIdx.x] = (u32)(p >> 32);
    }
}

// exclusive scan of per-tile pair counts (in place) + reduction of tile stats
// (tile_desc != nullptr: the tiles of k1_walk placed their pairs themselves -- only the total is taken, from their
// descriptors, and nothing is written back)
// tile_soff / chunk_tile (both may be nullptr): the tiles' spliced reads seen as ONE dense list -- tile_soff[t] = spliced reads
// before tile t (n_tiles + 1 entries), chunk_tile[c] = the tile that holds list entry 256 c -- for k1_emit's dense mapping.
// A FEW blocks of 256 threads (at most K1S_BLOCKS), each over a contiguous range of tiles: a block first sums its range and
// publishes the sums (ScanPart, flag = this launch's epoch: nothing to reset between launches), then adds up what the blocks
// before it published -- one lane per predecessor, all of them produced at the same time: no chain -- and scans its range
// from there; the last block reduces the statistics and writes the ContigStats.  (Workgroups start in the order of their
// index: a block only ever waits for blocks that started before it.)  It was ONE block: 28 k tiles of a chain of configs[2] in
// seven rounds of dependent loads, 100 us alone and 310 us beside the other chains' kernels, on every chain's critical path.
// (And one block of 1024 before that: a workgroup of 16 wavefronts needs 16 free wave slots on ONE CU at once, and beside
// the inflate and other chains' kernels in the end-to-end run it waited for them -- 4 ms per target instead of 60 us.)
constexpr int K1S_THREADS = 256, K1S_PER = 16, K1S_BLOCKS = 64;
struct ScanPart { // what one block of k1_scan_tiles found in its range of tiles
    u64 pairs, spl, uns, sum;
    int32_t mn, mx, max_end, max_nlen, min_pos;
    u32 flag; // == the launch's epoch: the part is complete
    u32 _pad[2];
};
static_assert(sizeof(ScanPart) == 64, "ScanPart layout");
__host__ __device__ inline u32 k1s_blocks(u32 n_tiles) { // ~1024 tiles a block
    const u32 g = (n_tiles + 1023u) / 1024u;
    return g < 1u ? 1u : (g > (u32)K1S_BLOCKS ? (u32)K1S_BLOCKS : g);
}
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(K1S_THREADS) void k1_scan_tiles(u32 *tile_cnt, const TileStats *ts, u32 n_tiles, ContigStats *out, u32 pair_limit,
                                                              KeyFmt kf, int32_t ref_len, const u64 *tile_desc, u32 *tile_soff, u32 *chunk_tile,
                                                              ScanPart *parts, u32 epoch) {
    constexpr int NW = K1S_THREADS / 64;
    __shared__ u64 wsum[NW];
    __shared__ u32 wsum2[NW];
    __shared__ u32 carry2_s;
    __shared__ u64 carry_s;
    __shared__ u64 r_pairs[NW], r_spl[NW], r_uns[NW], r_sum[NW];
    __shared__ int32_t r_i[NW][5];
    __shared__ ScanPart total_s; // (last block: everything before it)
    const u32 nblk = gridDim.x, blk = blockIdx.x;
    // the block's range: whole rounds of 16 tiles per thread so that ranges start at multiples of 16
    const u32 per_blk = ((n_tiles + nblk - 1) / nblk + (u32)K1S_PER - 1) / (u32)K1S_PER * (u32)K1S_PER;
    const u32 lo = min(n_tiles, blk * per_blk), hi = min(n_tiles, lo + per_blk);
    const int w = threadIdx.x >> 6;
    // ---- pass A: the range's sums
    u64 pairs = 0, spl = 0, uns = 0, sum = 0;
    int32_t mn = INT32_MAX, mx = 0, max_end = 0, max_nlen = 0, min_pos = INT32_MAX;
    for (u32 base = lo; base < hi; base += K1S_THREADS * K1S_PER) {
        const u32 i0 = base + K1S_PER * threadIdx.x;
#pragma unroll 4
        for (int q = 0; q < K1S_PER; q++) { // (loads from a clamped index, masked after: they travel together, see k1_count)
            const u32 i = i0 + q, ic = i < hi ? i : hi - 1;
            const u64 cnt = tile_desc ? (tile_desc[ic] & ((1ull << 40) - 1)) : (u64)tile_cnt[ic];
            const TileStats t = ts[ic];
            if (i < hi) {
                pairs += cnt;
                spl += t.spliced;
                uns += t.unspliced;
                sum += t.sum_len;
                mn = min(mn, t.min_len);
                mx = max(mx, t.max_len);
                max_end = max(max_end, t.max_end);
                max_nlen = max(max_nlen, t.max_nlen);
                min_pos = min(min_pos, t.min_pos);
            }
        }
    }
    pairs = wave_sum(pairs);
    spl = wave_sum(spl);
    uns = wave_sum(uns);
    sum = wave_sum(sum);
    mn = wave_min(mn);
    mx = wave_max(mx);
    max_end = wave_max(max_end);
    max_nlen = wave_max(max_nlen);
    min_pos = wave_min(min_pos);
    if (lane_id() == 0) {
        r_pairs[w] = pairs;
        r_spl[w] = spl;
        r_uns[w] = uns;
        r_sum[w] = sum;
        r_i[w][0] = mn;
        r_i[w][1] = mx;
        r_i[w][2] = max_end;
        r_i[w][3] = max_nlen;
        r_i[w][4] = min_pos;
    }
    __syncthreads();
    ScanPart mine;
    mine.pairs = mine.spl = mine.uns = mine.sum = 0;
    mine.mn = INT32_MAX, mine.mx = 0, mine.max_end = 0, mine.max_nlen = 0, mine.min_pos = INT32_MAX;
    for (int k = 0; k < NW; k++) {
        mine.pairs += r_pairs[k];
        mine.spl += r_spl[k];
        mine.uns += r_uns[k];
        mine.sum += r_sum[k];
        mine.mn = min(mine.mn, r_i[k][0]);
        mine.mx = max(mine.mx, r_i[k][1]);
        mine.max_end = max(mine.max_end, r_i[k][2]);
        mine.max_nlen = max(mine.max_nlen, r_i[k][3]);
        mine.min_pos = min(mine.min_pos, r_i[k][4]);
    }
    if (threadIdx.x == 0 && blk + 1 < nblk) { // (nobody reads the last block's part)
        ScanPart *o = parts + blk;
        o->pairs = mine.pairs;
        o->spl = mine.spl;
        o->uns = mine.uns;
        o->sum = mine.sum;
        o->mn = mine.mn;
        o->mx = mine.mx;
        o->max_end = mine.max_end;
        o->max_nlen = mine.max_nlen;
        o->min_pos = mine.min_pos;
        __hip_atomic_store(&o->flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- what the blocks before this one hold: lane k of the first wavefront waits for block k
    if (w == 0) {
        const u32 l = lane_id();
        ScanPart b;
        b.pairs = b.spl = b.uns = b.sum = 0;
        b.mn = INT32_MAX, b.mx = 0, b.max_end = 0, b.max_nlen = 0, b.min_pos = INT32_MAX;
        if (l < blk) {
            const ScanPart *o = parts + l;
            while (__hip_atomic_load(&o->flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(1);
            b.pairs = __hip_atomic_load(&o->pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b.spl = __hip_atomic_load(&o->spl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (blk + 1 == nblk) {
                b.uns = __hip_atomic_load(&o->uns, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b.sum = __hip_atomic_load(&o->sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b.mn = __hip_atomic_load(&o->mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b.mx = __hip_atomic_load(&o->mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b.max_end = __hip_atomic_load(&o->max_end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b.max_nlen = __hip_atomic_load(&o->max_nlen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b.min_pos = __hip_atomic_load(&o->min_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        b.pairs = wave_sum(b.pairs);
        b.spl = wave_sum(b.spl);
        if (blk + 1 == nblk) {
            b.uns = wave_sum(b.uns);
            b.sum = wave_sum(b.sum);
            b.mn = wave_min(b.mn);
            b.mx = wave_max(b.mx);
            b.max_end = wave_max(b.max_end);
            b.max_nlen = wave_max(b.max_nlen);
            b.min_pos = wave_min(b.min_pos);
        }
        if (l == 0) {
            carry_s = b.pairs;
            carry2_s = (u32)b.spl;
            total_s = b;
        }
    }
    __syncthreads();
    // ---- pass B: the scan of the range, from the sums before it (the tiles' counts once more: L2 hits)
    for (u32 base = lo; base < hi; base += K1S_THREADS * K1S_PER) {
        const u32 i0 = base + K1S_PER * threadIdx.x;
        u64 v[K1S_PER];
        u32 v2[K1S_PER];
        u64 vs = 0;
        u32 vs2 = 0;
#pragma unroll
        for (int q = 0; q < K1S_PER; q++) {
            const u32 i = i0 + q, ic = i < hi ? i : hi - 1;
            const u64 cnt = tile_desc ? (tile_desc[ic] & ((1ull << 40) - 1)) : (u64)tile_cnt[ic];
            const u32 sp = ts[ic].spliced;
            v[q] = i < hi ? cnt : 0;
            v2[q] = i < hi ? sp : 0u;
            vs += v[q];
            vs2 += v2[q];
        }
        const u64 inc = wave_iscan(vs);
        const u32 inc2 = wave_iscan(vs2);
        if (lane_id() == 63) {
            wsum[w] = inc;
            wsum2[w] = inc2;
        }
        __syncthreads();
        u64 wb = 0, tot = 0;
        u32 wb2 = 0, tot2 = 0;
        for (int k = 0; k < NW; k++) {
            const u64 s = wsum[k];
            const u32 s2 = wsum2[k];
            if (k < w) {
                wb += s;
                wb2 += s2;
            }
            tot += s;
            tot2 += s2;
        }
        const u64 carry = carry_s;
        const u32 carry2 = carry2_s;
        // NOTE: exclusive offsets are stored as 32-bit: a contig is limited to < 2^32 pairs
        u64 ex = carry + wb + inc - vs;
        u32 so = carry2 + wb2 + inc2 - vs2;
#pragma unroll
        for (int q = 0; q < K1S_PER; q++) {
            const u32 i = i0 + q;
            if (i < hi) {
                if (!tile_desc) tile_cnt[i] = (u32)ex;
                if (tile_soff) {
                    tile_soff[i] = so;
                    for (u32 c = (so + 255u) >> 8; (c << 8) < so + v2[q]; c++) chunk_tile[c] = i; // list entries 256 c that fall into this tile
                }
            }
            ex += v[q];
            so += v2[q];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            carry_s = carry + tot;
            carry2_s = carry2 + tot2;
        }
        __syncthreads();
    }
    if (blk + 1 != nblk) return;
    if (threadIdx.x == 0) {
        if (tile_soff) tile_soff[n_tiles] = carry2_s;
        const ScanPart b = total_s;
        const u64 n_pairs = carry_s;
        const int32_t m0 = min(b.mn, mine.mn), m1 = max(b.mx, mine.mx), m2 = max(b.max_end, mine.max_end), m3 = max(b.max_nlen, mine.max_nlen),
                      m4 = min(b.min_pos, mine.min_pos);
        out->spliced = b.spl + mine.spl;
        out->unspliced = b.uns + mine.uns;
        out->sum_len = b.sum + mine.sum;
        out->min_len = m0;
        out->max_len = m1;
        out->max_end = m2;
        out->max_nlen = m3;
        out->min_pos = m4;
        out->n_tiles = n_tiles;
        out->n_pairs = n_pairs;
        // limits the host assumed: pair capacity and key format (the keys of k1_emit must fit the digits it planned)
        u32 ovf = 0;
        if (n_pairs > (u64)pair_limit) ovf |= OVF_PAIRS;
        if (n_pairs > 0) {
            const bool weird = m4 < 0 || m2 > ref_len || m2 < 0;
            if (!kf.raw) {
                int need = 0;
                for (u32 v = (u32)m3; v; v >>= 1) need++;
                if (weird || need > kf.lbits) ovf |= OVF_KEYFMT;
            }
        }
        out->overflow = ovf;
        out->P = ovf ? 0u : (u32)n_pairs;
        out->J = out->R = out->n_slots = out->n_slices = 0;
        out->n_junc = out->n_runs = 0;
        out->list_need = 0;
        out->n_cand = 0;
    }
}
#endif // PJB_KERNELS_CHAIN

// ---------------------------------------------------------------------------------------------
// Packed-base compare helpers (used by k1_emit for the simple shape and by the generic walks of k4b_generic)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 nt16_ascii(u32 c) { // seq_nt16_str "=ACMGRSVTWYHKDBN"
    // bytes: '=' 3D, 'A' 41, 'C' 43, 'M' 4D, 'G' 47, 'R' 52, 'S' 53, 'V' 56 | 'T' 54,'W' 57,'Y' 59,'H' 48,'K' 4B,'D' 44,'B' 42,'N' 4E
    const u64 t0 = 0x565352474D43413DULL;
    const u64 t1 = 0x4E42444B48595754ULL;
    const u64 t = (c & 8u) ? t1 : t0;
    return (u32)(t >> ((c & 7u) * 8)) & 0xffu;
}

// Read nibbles [qi, qi+l) (BAM order: high nibble first) against genome codes [gi, gi+l) (low nibble first), 64 bases
// = 9 words of each per round; mismatch positions are reported relative to `out_base`.
//   A lane's words are consecutive but the next lane's are somewhere else, so every load instruction of the wave
// touches 64 cache lines whatever its width, and the number of load INSTRUCTIONS sets the pace (measured in
// k4a_simple: 19 per-word loads per read cost 80 us per contig, the compare itself nothing).  So the words come as two
// 16-byte loads and one 4-byte load per stream -- 4-byte aligned (reads start on word boundaries), which
// global_load_dwordx4 accepts -- guarded so that nothing is read past the read's last word / the contig's last code
// word; the short tail of a stream takes guarded word loads.
__device__ __forceinline__ u32 swap_nibbles(u32 x) { return ((x & 0x0F0F0F0Fu) << 4) | ((x >> 4) & 0x0F0F0F0Fu); }
// A chunk = NW consecutive words of each stream = (NW - 1) * 8 anchor bases per round (the extra word feeds the funnel
// shift of the last one): NW = 9 -> two 16-byte loads and one word per stream, NW = 5 -> one 16-byte load and one word.
template <int NW>
struct CmpChunkT {
    u32 qw[NW], gg[NW];
};
// d[k] = p[first + k] for 0 <= first + k <= last, else 0
template <int NW>
__device__ __forceinline__ void load_words(u32 (&d)[NW], const u32 *p, int32_t first, int32_t last) {
    static_assert(NW == 9 || NW == 8 || NW == 5, "chunk width");
    if (first >= 0 && first + NW - 1 <= last) {
        const Words4 a = gload(reinterpret_cast<const Words4 *>(p + first));
        d[0] = a.x, d[1] = a.y, d[2] = a.z, d[3] = a.w;
        if constexpr (NW >= 8) {
            const Words4 b = gload(reinterpret_cast<const Words4 *>(p + first + 4));
            d[4] = b.x, d[5] = b.y, d[6] = b.z, d[7] = b.w;
        }
        if constexpr (NW != 8) d[NW - 1] = gload(p + first + NW - 1); // (NW = 8: two 16-byte loads and nothing else -- 56 bases a round)
    } else {
#pragma unroll
        for (int k = 0; k < NW; k++) d[k] = (first + k >= 0 && first + k <= last) ? gload(p + first + k) : 0u;
    }
}
// anchor bases [t, t + 8 (NW - 1)) of an anchor of l bases: read bases from qi (words of the read up to word q_last may be
// touched: whatever lies past the anchor only feeds bits that the length mask removes), genome from gi
// TO_BUFFER_END: q_last is the last word of the batch's bases (relative to the read) and the genome may be read up to its last
// code word, not only to the anchor's: the last round of an anchor then takes the 16-byte loads too instead of one guarded
// load per word (what lies past the anchor is masked either way).
template <int NW, bool TO_BUFFER_END = false>
__device__ __forceinline__ void chunk_load(CmpChunkT<NW> &C, const u32 *seqw, int32_t qi, int32_t q_last, const u32 *gw, int32_t gi, int32_t g_words,
                                           int32_t l, int32_t t) {
    const int32_t lastg = (gi + l - 1) >> 3;
    load_words<NW>(C.qw, seqw, (qi + t) >> 3, q_last);
    // (genome words outside [0, g_words) and -- but for TO_BUFFER_END -- past the anchor's last word read as 0, as they always did)
    load_words<NW>(C.gg, gw, (gi + t) >> 3, !TO_BUFFER_END && lastg < g_words - 1 ? lastg : g_words - 1);
}
template <int NW>
__device__ __forceinline__ void chunk_cmp(CmpChunkT<NW> &C, int32_t qi, int32_t gi, int32_t l, int32_t t, int32_t out_base, int32_t &mism,
                                          int32_t &first_mis, int32_t &last_mis) {
    const u32 shq = (u32)(qi & 7) * 4u, shg = (u32)(gi & 7) * 4u; // (t is a multiple of 8: the shifts do not move)
#pragma unroll
    for (int k = 0; k < NW; k++) C.qw[k] = swap_nibbles(C.qw[k]);
#pragma unroll
    for (int c = 0; c < NW - 1; c++) {
        const int32_t rem = l - t - 8 * c;
        if (rem > 0) {
            const u32 q = __builtin_amdgcn_alignbit(C.qw[c + 1], C.qw[c], shq);
            const u32 g = __builtin_amdgcn_alignbit(C.gg[c + 1], C.gg[c], shg);
            const u32 x = q ^ g;
            u32 m = (((x & 0x77777777u) + 0x77777777u) | x) & 0x88888888u; // top bit of every nibble that differs
            if (rem < 8) m &= (1u << (4 * rem)) - 1u;
            if (m) {
                mism += __popc(m);
                if (first_mis < 0) first_mis = out_base + t + 8 * c + ((__ffs((int)m) - 1) >> 2);
                last_mis = out_base + t + 8 * c + ((31 - __clz((int)m)) >> 2);
            }
        }
    }
}
// v_ffbl_b32 / v_ffbh_u32 as the hardware defines them: 0xffffffff for 0 (the language's count-zeros builtins leave 0 undefined or
// cost a second instruction)
__device__ __forceinline__ u32 ffbl_hw(u32 x) {
    u32 r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ u32 ffbh_hw(u32 x) {
    u32 r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
// chunk_cmp without a branch: the words' flags (top bit of every nibble that differs) are cut to the anchor's length by a shift pair
// (no compare / select, nothing skipped), and the first / last mismatch are kept as BIT positions 4 * base + 3 from the anchor's
// start -- fbit: the lowest (0xffffffff: none), lbit1: the highest + 1 (0: none) -- folded with saturating adds and min / max, which
// make a word without mismatches neutral (v_ffbl / v_ffbh return 0xffffffff for it).  base = (int32_t)fbit >> 2 and
// ((int32_t)lbit1 - 1) >> 2, -1 for none.  20 vector instructions a word where chunk_cmp takes 26 and seven scalar ones.
template <int NW>
__device__ __forceinline__ void chunk_cmp_bits(CmpChunkT<NW> &C, int32_t qi, int32_t gi, int32_t l, int32_t t, u32 &mism, u32 &fbit, u32 &lbit1) {
    const u32 shq = (u32)(qi & 7) * 4u, shg = (u32)(gi & 7) * 4u;
#pragma unroll
    for (int k = 0; k < NW; k++) C.qw[k] = swap_nibbles(C.qw[k]);
    const int32_t rem2 = 2 * (l - t); // twice the bases left from the chunk's first
    u32 fw = 0xffffffffu, lw = 0;
#pragma unroll
    for (int c = 0; c < NW - 1; c++) {
        int32_t w2 = rem2 - 16 * c; // twice the word's valid bases, 0 .. 16
        w2 = w2 < 0 ? 0 : (w2 > 16 ? 16 : w2);
        const u32 inval = (0xffffffffu << (u32)w2) << (u32)w2; // (two shifts: a shift by 32 would not move)
        const u32 q = __builtin_amdgcn_alignbit(C.qw[c + 1], C.qw[c], shq);
        const u32 g = __builtin_amdgcn_alignbit(C.gg[c + 1], C.gg[c], shg);
        const u32 x = q ^ g;
        const u32 m = (((x & 0x77777777u) + 0x77777777u) | x) & (0x88888888u & ~inval);
        mism += (u32)__popc(m);
        fw = min(fw, __builtin_elementwise_add_sat(ffbl_hw(m), 32u * (u32)c));
        lw = max(lw, __builtin_elementwise_sub_sat(32u * (u32)c + 32u, ffbh_hw(m)));
    }
    fbit = min(fbit, __builtin_elementwise_add_sat(fw, 4u * (u32)t));
    lbit1 = lw ? lw + 4u * (u32)t : lbit1; // (rounds ascend)
}
// one stretch of l bases
template <int NW, bool TO_BUFFER_END = false>
__device__ __forceinline__ void cmp_words(const u32 *seqw, int32_t qi, int32_t q_last, const u32 *gw, int32_t gi, int32_t g_words, int32_t l,
                                          int32_t out_base, int32_t &mism, int32_t &first_mis, int32_t &last_mis) {
    for (int32_t t = 0; t < l; t += 8 * (NW - 1)) {
        CmpChunkT<NW> C;
        chunk_load<NW, TO_BUFFER_END>(C, seqw, qi, q_last, gw, gi, g_words, l, t);
        chunk_cmp<NW>(C, qi, gi, l, t, out_base, mism, first_mis, last_mis);
    }
}


// ---------------------------------------------------------------------------------------------
// K1b: per-read CIGAR walk, pass 2: emit one pair per N op in BAM order
// (JunctionSystem::addJunctions junction_system.cc:140-210 restated iteratively:
//  after an N op the next left anchor starts at the CLAMPED rStart; rEndExc is the UNCLAMPED
//  rStart plus the reference-consuming ops up to the next N, then clamped).
// Per-pair predicates of Junction::addJunctionAlignment (junction.cc:477-502) and
// calcAlignmentStats (junction.cc:755-814) are evaluated here, where the read's fixed-width
// fields are read coalesced, and packed into `meta`.  The same thread then compares the anchors
// of a read of the common [S] M N M [S] shape with the genome (AlignmentInfo::calcMatchStats,
// junction.cc:147-240): the pair's record leaves this kernel complete.
// ---------------------------------------------------------------------------------------------
// CIGAR ops of one alignment: the first OPS_LDS ops are staged in LDS (one column per thread), the
// rest (long reads) are read from global memory
constexpr int OPS_LDS = 8;
template <int STRIDE, int NLDS>
struct OpsViewT {
    const uint32_t *g;
    const u32 *lds; // &s_ops[0][column]: NLDS rows of STRIDE columns
    __device__ __forceinline__ u32 operator[](u32 k) const { return k < (u32)NLDS ? lds[k * STRIDE] : gload(g + k); }
};
typedef OpsViewT<256, OPS_LDS> OpsView;

struct NCursor { // walks the N ops of one CIGAR yielding the unclamped position after each N
    u32 i;
    int32_t acc;
    bool has;
    int32_t peek;
};
template <typename Ops>
__device__ __forceinline__ void ncursor_advance(NCursor &c, const Ops cig, u32 n) {
    c.has = false;
    while (c.i < n) {
        u32 op = cig[c.i++];
        u32 ty = op & 15u;
        if (op_consumes_ref(ty)) c.acc += (int32_t)(op >> 4);
        if (ty == OP_N) {
            c.has = true;
            c.peek = c.acc;
            return;
        }
    }
}

#ifndef PJB_CLOSED_INDELS
#define PJB_CLOSED_INDELS 1 // blocks with I / D operations in closed form too (emit_read_pairs): the walk list all but empties -- k4b_generic
                            // 224 -> 140 us a chain, k1_generic 122 -> 162 -- 8.82 -> 8.64 ms a step.  (While these reads were walked by a few
                            // lanes of k1_emit's blocks the same switch cost 60 us a k1_emit launch: 9.75 against 9.1 ms.)
#endif
#ifndef PJB_SIMPLE_NW
#define PJB_SIMPLE_NW 8 // two 16-byte loads per stream and round, 56 bases a round (5: one 16-byte load and a word, 32 bases)
#endif
// (profiles/r04e_compare_variants.txt: 32 bases a round with both anchors' words in flight together / 56 together / 32 one anchor
// after the other / 56 one after the other = 10.56 / 11.56 / 10.07 / 9.97 ms a step; since then an anchor is a block compared on
// its own, one after the other)
constexpr int SIMPLE_NW = PJB_SIMPLE_NW;
// what the walks find comparing one anchor (a block of emitted positions): its length, mismatches, first and last mismatch
struct CmpBlock {
    int32_t len, mism, first, last;
};
// a pair's packed statistics from its left and right anchor (junction.cc:263-272: matches from the left anchor's end / the right
// anchor's start; mmes; mismatches)
__device__ __forceinline__ u64 cmp_blocks_res(const CmpBlock &L, const CmpBlock &R) {
    const u32 upM = L.last < 0 ? (u32)L.len : (u32)(L.len - 1 - L.last);
    const u32 downM = R.first < 0 ? (u32)R.len : (u32)R.first;
    const u32 tu = (u32)(L.len - L.mism), td = (u32)(R.len - R.mism);
    return pack_res(upM < downM ? upM : downM, tu < td ? tu : td, (u32)(L.mism + R.mism));
}
// One spliced read's pairs (JunctionSystem::addJunctions junction_system.cc:140-210) -- everything that follows from
// the read's fixed-width fields is in R, the CIGAR behind `cig`.  Shape test, walk with the two monotone cursors for
// the up/down junction counts (junction.cc:795-812), one 32-byte record per pair.
struct EmitRead {
    u32 n;        // CIGAR operations
    int32_t pos;
    u32 g;        // global read ordinal
    u32 meta;     // per-read predicates (category, XS, UM, BPP, PPP, REL, MULTI)
    u32 nN;       // N operations
    int32_t aend; // pos + aligned length - 1
    int32_t lq;   // l_qseq
    bool seq_ok;  // the record carries at least lq bases
    u32 off;      // index of the read's first pair
    // A read of the shape [S] M (N M)+ [S] (l_qseq matching, bases present, nothing clamped): the anchors of pair k are the M
    // blocks either side of its N operation -- UNLESS another read's alignment makes the junction's anchor window reach over
    // a neighbouring intron of this read (bam_alignment.cc:359: the walks only stop at an N operation that leaves the
    // window).  The window is known after K2d; the block compares are done here, where the read's operations and bases are
    // at hand (closed != nullptr), and k4b_generic checks the window for the reads on its second list.
    const u32 *closed_seqw; // the read's packed bases (nullptr: not of that shape -- the generic walks fill PairRec::aux in)
    int32_t q_limit;        // last word behind closed_seqw that may be read (cmp_words)
    const u32 *gcodes;      // the target's 4-bit codes
    int32_t glen, voff;     // the target's length and its offset in the group's virtual sequence
};
// (reads of the simple shape never come here: k1_emit finishes them in closed form.)  on_pair(key, lStart, rEnd) is called
// for every pair once its record is complete; the match statistics (PairRec::aux) of a read that is not `closed` are
// k4b_generic's to fill in.
template <typename Ops, typename PairFn>
__device__ __forceinline__ void emit_read_pairs(const Ops cig, const EmitRead R, const Pairs P, const KeyFmt kf, const int32_t ref_len,
                                                u64 *err, PairFn &&on_pair) {
    const u32 n = R.n, g = R.g, nN = R.nN;
    const int32_t pos = R.pos, aend = R.aend;
    u32 meta = R.meta;
    // ---- walk: pairs (junction_system.cc:140-210) and, with two monotone cursors over the read's own
    // introns, the up/down junction counts (junction.cc:795-812)
    NCursor U = {0, pos, false, 0}, D = {0, pos, false, 0};
    ncursor_advance(U, cig, n);
    ncursor_advance(D, cig, n);
    u32 cntU = 0, cntD = 0;
    int32_t lStart = pos, lEndExc = pos, sumAfter = 0, prevRStartU = 0, prevIend = 0, prevIstart = 0;
    int32_t qAcc = 0, prevQ = 0; // query bases (soft clips included) before the operation / before the pending pair's N
    int64_t prev = -1;
    u64 prev_key = 0;
    const bool closed = R.closed_seqw != nullptr;
    PairRec pend;
    pend.aux = 0;
    pend.lstart = pend.rend = 0;
    pend.pos = pos;
    pend.aend = aend;
    pend.meta = closed ? meta | META_SIMPLE : meta;
    pend.updown = 0;
    // closed: the read is [S] B (N B)+ [S], every block B a run of M = X I D operations that starts and ends with bases on both
    // sides.  What the walks emit for a block -- compared bases for M = X, read letters against 'X' padding for I, 'X' padding
    // against genome bases for D (never equal: such targets hold no 'X') -- is accumulated ONCE per block as the operations go
    // by: a block is the right anchor of the pair before it and the left anchor of the pair behind it (bam_alignment.cc:341-462
    // with the whole block inside the window, which k4b_generic checks).
    typedef CmpBlock Block;
    Block blkL = {0, 0, -1, -1}, blk = {0, 0, -1, -1}; // the pending pair's left block; the block being walked
    auto closed_stats = [&]() { return cmp_blocks_res(blkL, blk); };
    (void)prevIstart;
    (void)prevQ;
    u32 k = 0;
    for (u32 i = 0; i < n; i++) {
        const u32 op = cig[i];
        const u32 ty = op & 15u;
        const int32_t ln = (int32_t)(op >> 4);
        if (ty == OP_N) {
            if (prev >= 0) {
                int32_t rEndExc = prevRStartU + sumAfter;
                if (rEndExc - 1 >= ref_len) rEndExc = ref_len; // junction_system.cc:172-174
                pend.rend = rEndExc - 1;
                if (closed) pend.aux = closed_stats();
                rec_store(P.rec + prev, pend);
                on_pair(prev_key, pend.lstart, pend.rend);
                if (rEndExc - 1 < prevIend) set_error(err, g, PJB_ERR_MIN_ANCHOR); // intron.cc:76
            }
            const int32_t istart = lEndExc;
            const int32_t rStartU = lEndExc + ln;
            int32_t rStart = rStartU;
            if (rStart - 1 >= ref_len) rStart = ref_len - 1; // junction_system.cc:169-171
            const int32_t iend = rStart - 1;
            while (U.has && U.peek < istart) {
                cntU++;
                ncursor_advance(U, cig, n);
            }
            while (D.has && D.peek <= iend + 1) {
                cntD++;
                ncursor_advance(D, cig, n);
            }
            const int64_t idx = (int64_t)R.off + k;
            const u64 key = make_key(kf, istart, iend);
            P.key[idx] = key;
            if (P.g) P.g[idx] = g;
            pend.lstart = lStart;
            pend.updown = cntU | ((nN - cntD) << 16);
            if (lStart > istart) set_error(err, g, PJB_ERR_MIN_ANCHOR); // intron.cc:68
            prev = idx;
            prev_key = key;
            blkL = blk; // (the block that ends here is this pair's left anchor)
            blk = Block{0, 0, -1, -1};
            prevIstart = istart;
            prevQ = qAcc;
            prevIend = iend;
            prevRStartU = rStartU;
            sumAfter = 0;
            lStart = rStart;
            lEndExc = rStart;
            k++;
        } else {
            if (closed && ty != OP_S) {
                const bool cq = op_consumes_query(ty), cr = op_consumes_ref(ty);
                if (cq && cr) // M = X: bases against bases
                    cmp_words<SIMPLE_NW, true>(R.closed_seqw, qAcc, R.q_limit, R.gcodes, lEndExc - R.voff, (R.glen + 7) / 8 + 1, ln, blk.len, blk.mism, blk.first, blk.last);
                else { // I / D: ln positions that never match
                    blk.mism += ln;
                    if (blk.first < 0) blk.first = blk.len;
                    blk.last = blk.len + ln - 1;
                }
                blk.len += ln;
            }
            if (op_consumes_ref(ty)) {
                lEndExc += ln;
                sumAfter += ln;
            }
        }
        if (op_consumes_query(ty)) qAcc += ln;
    }
    if (prev >= 0) {
        int32_t rEndExc = prevRStartU + sumAfter;
        if (rEndExc - 1 >= ref_len) rEndExc = ref_len;
        pend.rend = rEndExc - 1;
        if (closed) pend.aux = closed_stats();
        rec_store(P.rec + prev, pend);
        on_pair(prev_key, pend.lstart, pend.rend);
        if (rEndExc - 1 < prevIend) set_error(err, g, PJB_ERR_MIN_ANCHOR);
    }
}

// per-read predicates of a pair's `meta` word (Junction::addJunctionAlignment junction.cc:477-502, calcAlignmentStats
// junction.cc:755-814, BamAlignment::calcIfProperPair bam_alignment.cc:271-292); MULTI and SIMPLE are added by the caller
__device__ __forceinline__ u32 read_meta(u32 flag, u32 xs, u32 mapq, int32_t pos, int32_t mtid, int32_t mpos, int32_t tid, int orientation) {
    const bool pp_check = orientation == PJB_OR_FR || orientation == PJB_OR_RF || orientation == PJB_OR_FF;
    const bool first = flag & 0x40, rev = flag & 0x10;
    u32 meta = (first ? 0u : 2u) + (rev ? 1u : 0u);
    meta |= (xs & 3u) << META_XS_SHIFT;
    const bool um = mapq >= 30;
    if (um) meta |= META_UM;
    if (flag & 0x2) meta |= META_BPP;
    bool ppp = false;
    if (pp_check) {
        const bool paired = flag & 0x1, mate_mapped = !(flag & 0x8);
        if (paired && mate_mapped && tid == mtid) {
            const bool mrev = flag & 0x20;
            const bool diff = rev != mrev;
            const bool gap = !rev ? pos < mpos : pos > mpos;
            ppp = orientation == PJB_OR_FR ? (diff && gap) : orientation == PJB_OR_RF ? (diff && !gap) : (!diff && gap);
        }
    }
    if (ppp) meta |= META_PPP;
    if (um && (!pp_check || ppp)) meta |= META_REL;
    return meta;
}

// The common shapes [S] M N M [S] and [S] M N M N M [S] (coordinates of the read's own target): an anchor is one block of bases,
// read[q, q + len) against genome[g, g + len) (cmp_words); neither depends on the junction-level window -- the walk rules of
// bam_alignment.cc:341-462 reduce to exactly this for the shapes (for two introns: unless a window reaches over the other
// intron, which k4b_generic checks).
// Thread per spliced read of the batch, DENSE: the tiles' spliced lists (k1_count compacted them per tile) are walked as one
// list -- entry s lies in the tile t with tile_soff[t] <= s < tile_soff[t + 1], found from chunk_tile (the tile of entry
// 256 * chunk) and a short walk -- so every thread of every block has a read.  The read's CIGAR is fetched once (8
// independent loads into an LDS column, the walks then run at LDS latency).  The grid is fixed; blocks stride over the
// batch's chunks of 256 list entries.
//   voff: the target's offset in its group's virtual sequence (0 for a single target); every coordinate a pair carries is
// virtual, the per-read predicates and the base compare work on the record's own coordinates.
//   gcodes: the target's 4-bit codes; nullptr (exotic characters, an 'X' in the sequence): no read is "simple", every
// pair goes through k4b_generic's byte-wise walks.
// By-products, so that no later kernel has to stream over the pairs for them:
//   * K2d's candidate keys: the block keeps the keys it emits in a small hash set in LDS -- with the smallest lStart and the
//     largest rEnd of the pairs behind each key (junction.cc:477-529: the junction anchors' first level of reduction) -- and
//     appends the distinct ones to the candidate list when the set is a quarter full (and when the block leaves); a
//     junction appears once per residency;
//   * the list of reads that need the generic walks (k4b_generic), in GEN_SHARDS sub-lists (one returning atomic per
//     wavefront, spread over 256 addresses).
constexpr int K1E_LOOK = 16;
#ifndef K1E_WAVES
#define K1E_WAVES 4 // wavefronts per SIMD the register allocation aims at (tools/build_variants.sh builds the others for A/B runs)
#endif
#ifndef K1E_SET
#define K1E_SET 1 // slots of the block's candidate set per thread (2: 8.82 against 8.75 ms a step)
#endif
constexpr int K1E_T = 256, K1E_SHIFT = 8; // threads of a block = list entries of one trip
constexpr int KC_SLOTS = K1E_T * K1E_SET;     // the block's candidate set (LDS), flushed when a quarter full
constexpr u64 KD_EMPTY = ~0ull; // no key: a packed key has fewer than 64 bits
constexpr u32 GEN_SHARDS = 256, GEN_CNT_STRIDE = 32;
constexpr int KD_PAGE_SHIFT = 6; // a page of the start bitmap: 64 words, one wavefront
struct EmitLists {
    u64 *cand;      // candidate keys (nullptr: the chain sorts the full keys and wants none); their count is ContigStats::n_cand
    u64 *bitmap;    // K2d's bitmap of intron starts (all-clear at rest): every candidate sets its start's bit as it is listed
    u32 *page_cnt;  // starts per PAGE of the bitmap (64 words = 4 096 bases; all-zero at rest): counted as their bits are set for the first time
    u64 *cand_anc;  // per candidate: min lStart | max rEnd << 32 over the pairs it stands for (the junction anchors' first level)
    u64 *gen_list;  // [3][GEN_SHARDS][gen_cap].  Lists 1 and 2, for k4b_generic: global read ordinal | index of the read's first pair
                    // << 32 -- the reads whose pairs need the generic walks; the reads whose closed form waits for the window check.
                    // List 3, for k1_generic: the read's slot in the tiles' spliced lists | index of its first pair << 32.
    u32 *gen_cnt;   // [GEN_SHARDS][GEN_CNT_STRIDE]: words 2 l - 2, 2 l - 1 = entries of list l's sub-list, pairs of those reads -- a cache
                    // line per shard: atomics on one line are served one after the other, whatever the address in it
    u32 gen_cap;    // room of one sub-list
    u32 pack_nn;    // 1 (chains of fewer than 2^28 reads): entries of the second list carry min(N operations, 15) in bits 28-31 of the read ordinal
};
// Room of a sub-list (before anything says otherwise): k1_emit deals its chunks of 256 spliced reads round-robin to the sub-lists
// (up to 256 entries a chunk on list 2 and on list 3), k1_generic its chunks of 256 items of list 3 (up to 256 entries on list 1
// or 2).  A sub-list that several full chunks of both kernels hit can want more than this: the appends count on, nothing is
// written past the room, the chain is closed (OVF_LISTS, ContigStats::list_need) and repeated with the room it asked for.
__host__ __device__ inline u32 gen_list_cap(u32 pair_limit) {
    const u32 chunks = pair_limit / 256u + 2u;
    return ((chunks + GEN_SHARDS - 1) / GEN_SHARDS) * 256u;
}
// the fullest sub-list of the three lists' sub-lists `shard` (0: all within their room)
__device__ __forceinline__ u32 lists_over(const u32 *gen_cnt, u32 shard, u32 cap) {
    const u32 a = gen_cnt[shard * GEN_CNT_STRIDE], b = gen_cnt[shard * GEN_CNT_STRIDE + 2], c = gen_cnt[shard * GEN_CNT_STRIDE + 4];
    // (the entries of list 3 that found no room were not walked by k1_generic: each of them would have gone on list 1 or 2 -- an
    // upper bound, so that ONE repeat settles the lists)
    const u32 dropped = c > cap ? c - cap : 0u;
    const u32 ab = (a > b ? a : b) + dropped;
    const u32 m = ab > c ? ab : c;
    return m > cap ? m : 0u;
}
// What k1_emit and k1_generic share: the block's candidate set (LDS) and the appends to k4b_generic's / k1_generic's lists.
struct EmitShared {
    u64 set[KC_SLOTS];
    int32_t lo[KC_SLOTS], hi[KC_SLOTS];
    u32 set_n, base, scan[4];
};
// A kernel argument fetched where it is needed, not kept (-DK1E_LAZY_ARGS=1): k1_emit's rarely used arguments (the candidate list's pointers,
// the error word, the control block) cost it twelve scalar registers that it spills into VGPR lanes and fetches back eight times a trip.  The
// scalar load from the kernel-argument segment is cached; the empty asm keeps the compiler from hoisting it out of the rare branch.
// (Both measured in round 6, profiles/r06_k1_experiments.txt section 6: fetching the rare arguments lazily takes the kernel's v_readlane from 150
// to 59 a trip and makes it 15 % SLOWER -- 1 105 - 1 140 us a chain in the step against 940 - 970 --, not unrolling the probes halves its code and
// changes nothing: both are off.)
#ifndef K1E_PROBE_UNROLL
#define K1E_PROBE_UNROLL 1
#endif
#ifndef K1E_LAZY_ARGS
#define K1E_LAZY_ARGS 0
#endif
template <class T>
__device__ __forceinline__ T kernarg_at(u32 byte_off) {
    const PJB_CONSTANT char *p = (const PJB_CONSTANT char *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    T v;
    __builtin_memcpy(&v, p + byte_off, sizeof(T));
    return v;
}
// LAZY_E / LAZY_CS: byte offsets of the EmitLists / the ContigStats pointer among the kernel's arguments (-1: the members below are used)
template <int LAZY_E = -1, int LAZY_CS = -1>
struct EmitCtxT {
    EmitShared &sh;
    const EmitLists &E; // (LAZY_E >= 0: only gen_list, gen_cnt, gen_cap, pack_nn are read through this)
    const KeyFmt kf;
    ContigStats *cs_;
    const bool want_cand;
    __device__ __forceinline__ EmitLists cold() const {
        if constexpr (LAZY_E >= 0) return kernarg_at<EmitLists>((u32)LAZY_E);
        else return E;
    }
    __device__ __forceinline__ ContigStats *cs() const {
        if constexpr (LAZY_CS >= 0) return kernarg_at<ContigStats *>((u32)LAZY_CS);
        else return cs_;
    }
    __device__ __forceinline__ void init() {
        if (!want_cand) return;
#pragma unroll
        for (int i = 0; i < KC_SLOTS / K1E_T; i++) {
            sh.set[i * K1E_T + threadIdx.x] = KD_EMPTY;
            sh.lo[i * K1E_T + threadIdx.x] = INT32_MAX;
            sh.hi[i * K1E_T + threadIdx.x] = INT32_MIN;
        }
        if (threadIdx.x == 0) sh.set_n = 0;
    }
    __device__ __forceinline__ void cand_mark(const EmitLists &C, u64 k) const { // (what kd_mark did in a launch of its own)
        int32_t ms, me;
        unpack_key(kf, k, ms, me);
        const u32 w = (u32)ms >> 6;
        const u64 bit = 1ull << (ms & 63);
        const u64 old = atomicOr((unsigned long long *)(C.bitmap + w), (unsigned long long)bit);
        if (!(old & bit)) atomicAdd(&C.page_cnt[w >> KD_PAGE_SHIFT], 1u); // (the ranks below are scanned over the pages, not over the words)
    }
    __device__ __forceinline__ void cand_insert(u64 k, int32_t lstart, int32_t rend) const {
        if (!want_cand) return;
        u32 h = (u32)((k * 0x9E3779B97F4A7C15ull) >> 40) & (KC_SLOTS - 1);
#if !K1E_PROBE_UNROLL
#pragma nounroll // (unrolled 24 times -- twice, a pair each -- the probes were half of k1_emit's code: the first probe is the one that runs)
#endif
        for (int probe = 0; probe < 24; probe++) { // look first: most keys are there already, and a read of one address by many lanes is a broadcast
            u64 old = sh.set[h];
            if (old == KD_EMPTY) {
                old = atomicCAS((unsigned long long *)&sh.set[h], (unsigned long long)KD_EMPTY, (unsigned long long)k);
                if (old == KD_EMPTY) {
                    atomicAdd(&sh.set_n, 1u);
                    old = k;
                }
            }
            if (old == k) {
                if (lstart < sh.lo[h]) atomicMin(&sh.lo[h], lstart);
                if (rend > sh.hi[h]) atomicMax(&sh.hi[h], rend);
                return;
            }
            h = (h + 1) & (KC_SLOTS - 1);
        }
        // a crowded set (reads with hundreds of introns): straight to the list, where duplicates do no harm
        const EmitLists C = cold();
        const u32 at = atomicAdd(&cs()->n_cand, 1u);
        C.cand[at] = k;
        C.cand_anc[at] = (u64)(u32)lstart | ((u64)(u32)rend << 32);
        cand_mark(C, k);
    }
    // appends the wavefront's entries to list `kind` (1: reads for k4b_generic's walks, 2: for its window check, 3: reads for
    // k1_generic): one returning atomic per wavefront; sub-list `shard` (callers deal 256-entry chunks round-robin: gen_list_cap)
    // the same in two halves: the returning atomic early (list_reserve), the stores once its answer is there (list_write) -- a wavefront
    // that appends and stores at once waits a memory round trip for the atomic
    // (`pairs`: the same for every entry of the call -- k1_emit's lists hold reads of two pairs, or count none)
    __device__ __forceinline__ u32 list_reserve(u32 kind, bool mine, u32 pairs, u32 shard) const {
        const u64 gm2 = __ballot(mine);
        if (!gm2) return 0u;
        const u32 pairs_w = pairs * (u32)__popcll(gm2);
        const int leader = __ffsll((long long)gm2) - 1;
        const u32 w0 = shard * GEN_CNT_STRIDE + (kind - 1) * 2;
        u32 base = 0;
        if (lane_id() == leader) {
            base = atomicAdd(&E.gen_cnt[w0], (u32)__popcll(gm2));
            if (pairs_w) atomicAdd(&E.gen_cnt[w0 + 1], pairs_w);
        }
        return base; // (in the leader's lane)
    }
    __device__ __forceinline__ void list_write(u32 kind, bool mine, u32 base, u64 entry, u32 shard) const {
        const u64 gm2 = __ballot(mine);
        if (!gm2) return;
        base = (u32)__builtin_amdgcn_readlane((int)base, __ffsll((long long)gm2) - 1);
        const u32 at = base + (u32)__popcll(gm2 & ((1ull << lane_id()) - 1));
        if (mine && at < E.gen_cap) E.gen_list[((size_t)(kind - 1) * GEN_SHARDS + shard) * E.gen_cap + at] = entry;
    }
    __device__ __forceinline__ void list_append(u32 kind, bool mine, u32 pairs, u64 entry, u32 shard) const {
        const u64 gm2 = __ballot(mine);
        if (!gm2) return;
        const u32 pairs_w = wave_total<DppAdd>(mine ? pairs : 0u);
        const int leader = __ffsll((long long)gm2) - 1;
        const u32 w0 = shard * GEN_CNT_STRIDE + (kind - 1) * 2;
        u32 base = 0;
        if (lane_id() == leader) {
            base = atomicAdd(&E.gen_cnt[w0], (u32)__popcll(gm2));
            atomicAdd(&E.gen_cnt[w0 + 1], pairs_w);
        }
        base = (u32)__builtin_amdgcn_readlane((int)base, leader);
        const u32 at = base + (u32)__popcll(gm2 & ((1ull << lane_id()) - 1));
        if (mine && at < E.gen_cap) E.gen_list[((size_t)(kind - 1) * GEN_SHARDS + shard) * E.gen_cap + at] = entry;
    }
    // candidate keys: the set is flushed when it fills up, and before the block leaves (every thread of the block calls)
    __device__ __forceinline__ void cand_flush(bool force) const {
        if (!want_cand) return;
        lds_barrier();
        if (sh.set_n > (u32)KC_SLOTS / 4 || force) {
            u64 mine[KC_SLOTS / K1E_T], anc[KC_SLOTS / K1E_T];
            u32 cnt = 0;
#pragma unroll
            for (int i = 0; i < KC_SLOTS / K1E_T; i++) {
                const int at = i * K1E_T + threadIdx.x;
                mine[i] = sh.set[at];
                anc[i] = (u64)(u32)sh.lo[at] | ((u64)(u32)sh.hi[at] << 32);
                cnt += mine[i] != KD_EMPTY;
                sh.set[at] = KD_EMPTY;
                sh.lo[at] = INT32_MAX;
                sh.hi[at] = INT32_MIN;
            }
            u32 total;
            const u32 excl = block_escan<K1E_T / 64, u32, true>(cnt, sh.scan, &total);
            if (threadIdx.x == 0) {
                sh.base = total ? atomicAdd(&cs()->n_cand, total) : 0u;
                sh.set_n = 0;
            }
            lds_barrier();
            const EmitLists C = cold();
            u32 o = sh.base + excl;
#pragma unroll
            for (int i = 0; i < KC_SLOTS / K1E_T; i++)
                if (mine[i] != KD_EMPTY) {
                    C.cand[o] = mine[i];
                    C.cand_anc[o++] = anc[i];
                    cand_mark(C, mine[i]);
                }
        }
    }
};
typedef EmitCtxT<> EmitCtx;

// ---- bases in TWO bits (pjb_batch.seq2 / GroupTab::codes2).  Round 5 took the compares apart (profiles/r05_k1_experiments.txt sections
// 9, 10, 13): they are bound twice -- by the texture addresser, which takes a 4-byte-aligned 16-byte gather one LANE a cycle (16 gathers a
// read), and by ~20 vector instructions per 8 bases.  In 2 bits a 16-byte gather holds 64 bases and a word 16: a round of two gathers per
// stream covers 112 bases where the 4-bit round covers 56, and a word costs 13 instructions.  A lane compares in 2 bits when its read
// is pure ACGT (seq_exc clear), lies inside its target, and the target's exception bitmap is clear under every block of its bases
// (checked behind the compare: the bitmap words are asked for before the first round and looked at after the last); every other lane
// takes the 4-bit rounds as before -- characters are equal exactly where 2-bit codes are when both sides are pure ACGT.
//   Read base i of a read whose seq4 words start at so: bit 16 (so & 1) + 2 i from word so >> 1 of seq2 (a 16-bit granule per seq4 word).
struct __attribute__((packed, aligned(4))) Words2 {
    u32 x, y;
};
#ifndef C2_SKIP_HI
#define C2_SKIP_HI 1 // a round whose block has 48 bases or fewer left asks for ONE 16-byte gather per stream (the addresser takes a gather a LANE a cycle: lanes that do not ask cost nothing)
#endif
constexpr int C2_NW = 8;                      // words per stream and round: two 16-byte gathers, 7 words = 112 bases compared
constexpr int32_t C2_ROUND = 16 * (C2_NW - 1);
constexpr int32_t C2_MAX_BLOCK = 1900;        // a longer block of bases (31 stretches of the bitmap: one 8-byte load) takes the 4-bit rounds
template <int NW>
__device__ __forceinline__ void chunk2_cmp_bits(const u32 (&qw)[NW], const u32 (&gw)[NW], u32 shq, u32 shg, int32_t l, int32_t t, u32 &mism, u32 &fbit,
                                                u32 &lbit1) {
    const int32_t rem = l - t; // bases left from the chunk's first
    u32 fw = 0xffffffffu, lw = 0;
#pragma unroll
    for (int c = 0; c < NW - 1; c++) {
        int32_t v = rem - 16 * c; // the word's valid bases, 0 .. 16
        v = v < 0 ? 0 : (v > 16 ? 16 : v);
        const u32 inval = (0xffffffffu << (u32)v) << (u32)v; // (two shifts: a shift by 32 would not move)
        const u32 x = __builtin_amdgcn_alignbit(qw[c + 1], qw[c], shq) ^ __builtin_amdgcn_alignbit(gw[c + 1], gw[c], shg);
        const u32 m = (x | (x >> 1)) & (0x55555555u & ~inval); // bit 2 j: base j differs
        mism += (u32)__popc(m);
        fw = min(fw, __builtin_elementwise_add_sat(ffbl_hw(m), 32u * (u32)c));
        lw = max(lw, __builtin_elementwise_sub_sat(32u * (u32)c + 32u, ffbh_hw(m)));
    }
    fbit = min(fbit, __builtin_elementwise_add_sat(fw, 2u * (u32)t));
    lbit1 = lw ? lw + 2u * (u32)t : lbit1; // (rounds ascend)
}

// ONE launch per chain (25 launches of 13 - 690 us became 3): a block takes a range of the chain's consecutive trips -- (batch, chunk
// of 256 list entries), a batch after the other; a chunk that two batches share is visited once for each, its lanes masked.  The
// batch descriptors and the members' table are read through uniform indices (scalar loads).
//   The kernel waits for memory, not for bandwidth (round 4's SQ counters: three quarters of the wave-cycles parked on s_waitcnt),
// and a trip is a chain of dependent round trips: tile offsets -> list record -> operations and per-read fields -> bases -> stores.
// So the trips are software-pipelined: while trip v is compared, the list records of trip v + 2 and the operations / fields of
// trip v + 1 are already on their way (in registers).
//   (Round 5 also built the version that stages the wavefront's bases and genome windows through LDS with coalesced loads: 2.4 x fewer
// vector-memory instructions, the same time -- profiles/r05_k1_experiments.txt section 1; it left the tree with round 6, git has it.)
#if defined(K1E_PROF) && defined(PJB_KERNELS_CHAIN) // (debug builds: wave-cycles per section of k1_emit, summed over every wavefront; printed by pjb_destroy)
__device__ unsigned long long g_k1e_prof[16];
#define K1E_T0() unsigned long long prof_t = __builtin_amdgcn_s_memtime()
#define K1E_MARK(i)                                                                  \
    do {                                                                             \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                \
        if (lane_id() == 0) atomicAdd(&g_k1e_prof[i], now_ - prof_t);                \
        prof_t = now_;                                                               \
    } while (0)
#else
#define K1E_T0() do {} while (0)
#define K1E_MARK(i) do {} while (0)
#endif
#if defined(K1E_HIST) && defined(PJB_KERNELS_CHAIN) // (debug builds: per wavefront and trip, compare rounds run (the longest lane's) against the lanes' mean -- pjb_destroy prints the table)
__device__ unsigned long long g_k1e_hist[4][32]; // [0]: trips by rounds run, [1]: sum of active lanes' rounds by rounds run, [2]: active lanes, [3]: 4-bit rounds run
#endif
constexpr int K1E_MAXB = 64; // batches one launch takes (the host splits longer lists)
struct EmitRec { // a trip's list records (per lane)
    bool on;
    u32 slot, r, toff, poff;
    uint4 sr;
};
struct EmitOps { // and what the records lead to: the read's first operations and its fixed-width fields
    u32 op[OPS_LDS];
    u32 n, flag, xs, mapq;
    u32 excw; // the word of seq_exc that holds the read's bit (all ones: no 2-bit bases in the batch).  Kept as it was loaded: anything computed from
              // it here would make fetch_ops WAIT for its loads -- they are asked for a trip ahead so that nothing waits for them
    int32_t mtid, mpos, lq;
};
struct EmitTrip { // (uniform)
    int bi;
    u32 chunk, s_begin, s_end;
};
#ifdef PJB_KERNELS_CHAIN
// k1_emit's arguments as the kernel-argument segment lays them out (explicit arguments one after the other, each at its natural alignment):
// what kernarg_at reads the rarely used ones from.  k1_emit's own parameter list below MUST stay in this order.
struct K1EmitArgs {
    const DevBatch *batches;
    int n_batches;
    u32 n_tiles_total;
    const u32 *tile_off, *tile_soff, *chunk_tile, *spl_idx, *spl_poff;
    const uint4 *spl_rec;
    Pairs P;
    EmitLists E;
    KeyFmt kf;
    GroupTab G;
    int use_codes, orientation;
    u64 *err;
    ContigStats *cs;
};
__global__ __launch_bounds__(K1E_T) __attribute__((amdgpu_waves_per_eu(K1E_WAVES, K1E_WAVES))) void k1_emit(const DevBatch *batches, int n_batches, u32 n_tiles_total, const u32 *tile_off, const u32 *tile_soff,
                                                const u32 *chunk_tile, const u32 *spl_idx, const u32 *spl_poff, const uint4 *spl_rec, Pairs P, EmitLists E, KeyFmt kf,
                                                GroupTab G, int use_codes, int orientation, u64 *err_unused, ContigStats *cs_unused) {
    // (the error word, the control block, the candidate list's pointers and the read ordinals' array are fetched from the kernel-argument
    // segment where they are used -- rare branches, the block's last flush --, not kept in scalar registers: see kernarg_at)
#if K1E_LAZY_ARGS
    (void)err_unused;
    (void)cs_unused;
    auto err_ptr = [] { return kernarg_at<u64 *>((u32)offsetof(K1EmitArgs, err)); };
    auto pg_ptr = [] { return kernarg_at<u32 *>((u32)(offsetof(K1EmitArgs, P) + offsetof(Pairs, g))); };
    ContigStats *const cs = kernarg_at<ContigStats *>((u32)offsetof(K1EmitArgs, cs)); // (once: the test below)
#else
    auto err_ptr = [=] { return err_unused; };
    auto pg_ptr = [=] { return P.g; };
    ContigStats *const cs = cs_unused;
#endif
    __shared__ u32 s_soff[K1E_LOOK];
    __shared__ EmitShared sh;
    __shared__ u32 s_cfirst[K1E_MAXB + 1]; // trips before batch i
    __shared__ u32 s_sbegin[K1E_MAXB + 1]; // list entries before batch i (batches follow each other in the tiles' space)
    __shared__ u32 s_seqw[K1E_MAXB], s_cigw[K1E_MAXB], s_tbase[K1E_MAXB]; // words of packed bases / operations in batch i; its first tile
    __shared__ u32 s_wsum[4];
    if (cs->P == 0) return; // no pairs, or a limit was exceeded: the contig is repeated with larger buffers
    K1E_T0();
    const bool want_g = pg_ptr() != nullptr; // (--extra contexts: the pairs' read ordinals)
#if K1E_LAZY_ARGS
    const bool want_cand = kernarg_at<u64 *>((u32)(offsetof(K1EmitArgs, E) + offsetof(EmitLists, cand))) != nullptr;
    EmitCtxT<(int)offsetof(K1EmitArgs, E), (int)offsetof(K1EmitArgs, cs)> ctx{sh, E, kf, nullptr, want_cand};
#else
    const bool want_cand = E.cand != nullptr;
    EmitCtx ctx{sh, E, kf, cs, want_cand};
#endif
    ctx.init();
    auto cand_insert = [&](u64 k, int32_t lstart, int32_t rend) { ctx.cand_insert(k, lstart, rend); };
    {
        u32 trips = 0;
        if ((int)threadIdx.x < n_batches) {
            const PJB_GLOBAL DevBatch *d = as_global(batches + threadIdx.x);
            const u32 tb = d->tile_base, nt = (u32)((d->n + K1_TILE - 1) / K1_TILE);
            const u32 sb = tile_soff[tb], se = tile_soff[tb + nt];
            trips = sb == se ? 0u : ((se + (u32)K1E_T - 1u) >> K1E_SHIFT) - (sb >> K1E_SHIFT);
            s_sbegin[threadIdx.x] = sb;
            if ((int)threadIdx.x == n_batches - 1) s_sbegin[n_batches] = se;
            s_seqw[threadIdx.x] = as_global(d->seq_off)[d->n];
            s_cigw[threadIdx.x] = as_global(d->cig_off)[d->n];
            s_tbase[threadIdx.x] = tb;
        }
        u32 total;
        const u32 ex = block_escan<K1E_T / 64>(trips, s_wsum, &total);
        if (threadIdx.x < (u32)K1E_MAXB) s_cfirst[threadIdx.x] = ex;
        if (threadIdx.x == K1E_T - 1) s_cfirst[K1E_MAXB] = total;
        __syncthreads();
    }
    const u32 n_trips = s_cfirst[K1E_MAXB];
    // a block's trips are consecutive: the window of tile offsets (s_soff: 16 tiles ~ 18 trips) is fetched once and serves the trips
    // behind it; consecutive trips share junctions, so the block's candidate set lists each of them once
    const u32 per_block = (n_trips + gridDim.x - 1) / gridDim.x;
    const u32 v_lo = min(n_trips, blockIdx.x * per_block), v_hi = min(n_trips, v_lo + per_block);
    if (v_lo >= v_hi) return;
    u32 win_t0 = 0xffffffffu; // the tile of s_soff[0] (none yet)
    int win_bi = -1;
    // ---- the trip's batch: the last one with s_cfirst <= v (uniform)
    auto locate = [&](u32 v) {
        static_assert(K1E_MAXB <= 64, "one ballot");
        const int k = lane_id();
        const u64 m = __ballot(k < n_batches && s_cfirst[k] <= v);
        EmitTrip T;
        T.bi = __builtin_amdgcn_readfirstlane(63 - __clzll((long long)m));
        T.s_begin = s_sbegin[T.bi];
        T.s_end = s_sbegin[T.bi + 1];
        T.chunk = (T.s_begin >> K1E_SHIFT) + (v - s_cfirst[T.bi]);
        return T;
    };
    // ---- a trip's list records (k1_count's by-product: no gathers of cig_off, pos, l_qseq, seq_off); every thread of the block calls
    auto fetch_rec = [&](const EmitTrip &T) {
        const u32 s_last = min((T.chunk + 1u) << K1E_SHIFT, T.s_end) - 1u;
        // (the window is only written between two barriers below: reading it here needs none)
        const bool reload = win_bi != T.bi || win_t0 == 0xffffffffu || s_last >= s_soff[K1E_LOOK - 1];
        if (reload) { // the window of tile offsets: from the tile of the trip's first entry (for a batch's first chunk: of the batch's first entry)
            win_t0 = (T.chunk << K1E_SHIFT) < T.s_begin ? s_tbase[T.bi] : chunk_tile[T.chunk >> (8 - K1E_SHIFT)]; // (chunk_tile: the tile of entry 256 c)
            win_bi = T.bi;
            const u32 mine = threadIdx.x < K1E_LOOK && win_t0 + threadIdx.x <= n_tiles_total ? tile_soff[win_t0 + threadIdx.x] : 0xffffffffu;
            lds_barrier(); // (everybody is done with the old window)
            if (threadIdx.x < K1E_LOOK) s_soff[threadIdx.x] = mine;
            lds_barrier();
        }
        const u32 s = (T.chunk << K1E_SHIFT) + threadIdx.x;
        EmitRec R;
        R.on = s >= T.s_begin && s < T.s_end;
        R.slot = R.r = R.toff = R.poff = 0;
        R.sr = make_uint4(0, 0, 0, 0);
        // The entry's tile = the number of the window's offsets it has reached (they ascend).  A wavefront's 64 entries are consecutive: they
        // lie in one tile, seldom in two or three -- so lane m holds offset m, two ballots say how many offsets the wavefront's first and last
        // entry have reached, and only the offsets in between (none, mostly) are compared lane by lane.  (It was fifteen LDS reads and
        // compares a lane: the kernel is bound by instruction issue -- round 6's SQ counters: its four wavefronts keep a SIMD's issue port
        // busy 104 % of the time between them --, so every instruction of a trip counts.)
        const u32 w_first = (T.chunk << K1E_SHIFT) + (threadIdx.x & ~63u);
        const u32 off_m = lane_id() < K1E_LOOK ? s_soff[lane_id() & (K1E_LOOK - 1)] : 0xffffffffu;
        const u32 m_lo = (u32)__popcll(__ballot(off_m <= w_first)), m_hi = (u32)__popcll(__ballot(off_m <= w_first + 63u));
        if (R.on) {
            u32 k = m_lo ? m_lo - 1u : 0u; // (m_lo >= 1: offset 0 belongs to the tile of the trip's first entry or an earlier one)
            for (u32 m = m_lo ? m_lo : 1u; m < m_hi; m++) k += s >= (u32)__builtin_amdgcn_readlane((int)off_m, (int)m) ? 1u : 0u;
            u32 tile = win_t0 + k, soff = s_soff[k];
            if (k + 1 == (u32)K1E_LOOK && s >= soff) { // (a run of tiles without spliced reads longer than the window: search)
                u32 lo = tile, hi = n_tiles_total; // tile_soff[lo] <= s < tile_soff[hi]
                while (hi - lo > 1) {
                    const u32 mid = (lo + hi) >> 1;
                    if (tile_soff[mid] <= s) lo = mid;
                    else hi = mid;
                }
                tile = lo;
                soff = tile_soff[lo];
            }
            R.slot = tile * (u32)K1_TILE + (s - soff);
            R.sr = spl_rec[R.slot];
            R.r = spl_idx[R.slot];
            R.toff = tile_off[tile];
            R.poff = spl_poff[R.slot];
        }
        return R;
    };
    // ---- and what they lead to: the read's first eight operations (two 16-byte loads; whatever lies behind its last one is masked
    // by the consumer), its five fields and its bit of the batch's exception bitmap
    auto fetch_ops = [&](const EmitTrip &T, const EmitRec &R) {
        EmitOps O;
#pragma unroll
        for (int q = 0; q < OPS_LDS; q++) O.op[q] = 0;
        O.n = O.flag = O.xs = O.mapq = 0;
        O.excw = 0xffffffffu;
        O.mtid = O.mpos = O.lq = 0;
        if (R.on) {
            const GBatch b = load_batch(batches + T.bi);
            const u32 c0 = R.sr.x, cig_words = s_cigw[T.bi];
            const int64_t r = R.r;
            u32 n = R.sr.w & 0x7fffu;
            if (n == 0x7fffu) n = b.cig_off[r + 1] - c0;
            O.n = n;
            O.flag = b.flag[r];
            O.xs = b.xs[r];
            O.mapq = b.mapq[r];
            O.mtid = b.mtid[r];
            O.mpos = b.mpos[r];
            if (b.seq2 != nullptr) O.excw = b.seq_exc[r >> 5]; // (uniform test: the batch either has 2-bit bases or not)
            O.lq = (int32_t)(R.sr.w >> 16);
            if (O.lq == 0xffff) O.lq = b.l_qseq[r];
            static_assert(OPS_LDS == 8, "two 16-byte loads");
            if (cig_words >= (u32)OPS_LDS) {
                // (the batch's last operations are read from eight words before its end: the loads never leave the array)
                const u32 c0c = c0 + (u32)OPS_LDS <= cig_words ? c0 : cig_words - (u32)OPS_LDS;
                const Words4 lo4 = gload_as<Words4>(b.cigar + c0c), hi4 = gload_as<Words4>(b.cigar + c0c + 4);
                u32 w[OPS_LDS] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
                if (c0c != c0) { // (shifted: at most seven words)
                    const u32 sh = c0 - c0c;
#pragma unroll
                    for (int q = 0; q < OPS_LDS; q++) {
                        u32 v = 0;
#pragma unroll
                        for (int j = q; j < OPS_LDS; j++) v = (u32)(j - q) == sh ? w[j] : v;
                        w[q] = v;
                    }
                }
#pragma unroll
                for (int q = 0; q < OPS_LDS; q++) O.op[q] = w[q];
            } else { // (a batch of fewer than eight operations: word by word, guarded)
#pragma unroll
                for (int q = 0; q < OPS_LDS; q++) O.op[q] = c0 + (u32)q < cig_words ? b.cigar[c0 + (u32)q] : 0u;
            }
        }
        return O;
    };
    K1E_MARK(0); // block start: tables
    EmitTrip T = locate(v_lo);
    EmitRec R = fetch_rec(T);
    EmitOps O = fetch_ops(T, R);
    // (the list records run TWO trips ahead, the operations and fields one: what a trip asks for has had a whole trip to arrive)
    EmitTrip T1 = T;
    EmitRec R1 = R;
    if (v_lo + 1 < v_hi) {
        T1 = locate(v_lo + 1);
        R1 = fetch_rec(T1);
    }
    K1E_MARK(1); // prologue: first trip's records and operations issued
    for (u32 v = v_lo; v < v_hi; v++) {
        const GBatch b = load_batch(batches + T.bi);
        const int mem = b.member;
        const int32_t voff = G.voff[mem], ref_len = G.len[mem], tid = G.tid[mem];
        const PJB_GLOBAL u32 *gcodes = use_codes ? as_global(G.codes[mem]) : (const PJB_GLOBAL u32 *)nullptr;
        const PJB_GLOBAL u32 *gcodes2 = use_codes && b.seq2 != nullptr ? as_global(G.codes2[mem]) : (const PJB_GLOBAL u32 *)nullptr;
        const int32_t vlen = voff + ref_len; // the target's end in the group's virtual sequence
        const u32 seq_words = s_seqw[T.bi];  // words of packed bases in the batch: nothing reads past them
        const int32_t g_words = (ref_len + 7) / 8 + 1;
        const u32 chunk = T.chunk;
        const bool on = R.on;
        const u32 so = R.sr.z;
        const int32_t pos = (int32_t)R.sr.y, lq = O.lq;
        // A read of the shape [S] M N M [S] or [S] M N M N M [S] (l_qseq matching, bases present) is finished here, in closed form
        // (junction_system.cc:140-210 for one or two N operations); any other read goes on k1_generic's list: a few lanes of every
        // wavefront walking their reads kept the whole block waiting (84 of 307 us a launch for one read in twenty).
        bool generic = false, simple = false, two = false;
        u32 dS = 0, a = 0, nl = 0, b2 = 0, nl2 = 0, b3 = 0, meta = 0, off = 0, g = 0;
        if (on) {
            const u32 n = O.n;
            // (what lies behind the read's last operation -- the next read's operations -- is never looked at: every use below is behind a
            // test of n)
            const u32 (&op)[OPS_LDS] = O.op;
            g = b.base + R.r;
            off = R.toff + R.poff;
            meta = read_meta(O.flag, O.xs, O.mapq, pos, O.mtid, O.mpos, tid, orientation);
            const bool seq_ok = (R.sr.w & 0x8000u) != 0;
            // ---- shape: [S] M N M [S], or -- two introns, nothing clamped -- [S] M N M N M [S]
            if (gcodes != nullptr && n >= 3 && n <= 7) {
                const bool clipF = (op[0] & 15u) == OP_S;
                const u32 i0 = clipF ? 1u : 0u;
                const u32 oa = clipF ? op[1] : op[0], on_ = clipF ? op[2] : op[1], ob = clipF ? op[3] : op[2];
                const u32 o3 = clipF ? op[4] : op[3], o4 = clipF ? op[5] : op[4], o5 = clipF ? op[6] : op[5]; // (0 past the last operation)
                two = n >= i0 + 5u && (o3 & 15u) == OP_N;
                const u32 body = i0 + (two ? 5u : 3u);
                const u32 oL = two ? o5 : o3;
                const bool clipL = n == body + 1u && (oL & 15u) == OP_S;
                if (n == body || clipL) {
                    dS = clipF ? op[0] >> 4 : 0u;
                    const u32 dE = clipL ? oL >> 4 : 0u;
                    a = oa >> 4;
                    nl = on_ >> 4;
                    b2 = ob >> 4;
                    nl2 = two ? o3 >> 4 : 0u;
                    b3 = two ? o4 >> 4 : 0u;
                    simple = (oa & 15u) == OP_M && (on_ & 15u) == OP_N && (ob & 15u) == OP_M && a > 0 && b2 > 0 && a <= RES_FIELD_MAX && b2 <= RES_FIELD_MAX &&
                             dS <= RES_FIELD_MAX && lq > 1 && (u64)lq == (u64)dS + a + b2 + b3 + dE && seq_ok;
                    if (two) // (the clamps of junction_system.cc:169-174 are written out for one intron only: an alignment that leaves its target walks)
                        simple = simple && (o4 & 15u) == OP_M && b3 > 0 && b3 <= RES_FIELD_MAX && nl > 0 && nl2 > 0 && pos >= 0 &&
                                 (int64_t)pos + a + nl + b2 + nl2 + b3 <= (int64_t)ref_len;
                }
            }
            two = two && simple;
            generic = !simple;
        }
        const u32 shard = (chunk >> (8 - K1E_SHIFT)) % GEN_SHARDS;
        const u32 base2 = ctx.list_reserve(2, two, 2u, shard), base3 = ctx.list_reserve(3, generic, 0u, shard);
        K1E_MARK(2); // shapes (waits for the operations)
        // ---- in 2 bits: the read is pure ACGT, lies inside its target (nothing clamped: block k of its bases starts at g2s[k]), no block
        // longer than the bitmap load covers.  The exception bitmap under the blocks -- bits g2s >> 6 .. (g2s + len - 1) >> 6, at most 31 of
        // them, in the two words from word g2s >> 11 -- is asked for here and looked at just before the rounds.
        bool use2 = simple && gcodes2 != nullptr && !((O.excw >> (R.r & 31u)) & 1u) && pos >= 0 && (int64_t)pos + a + nl + b2 + nl2 + b3 <= (int64_t)ref_len &&
                    a <= (u32)C2_MAX_BLOCK && b2 <= (u32)C2_MAX_BLOCK && b3 <= (u32)C2_MAX_BLOCK;
        const int64_t n2w = codes2_words(ref_len);
        const bool any_exc = (G.exc_members >> mem) & 1u; // (uniform)
        Words2 x0 = {0, 0}, x1 = {0, 0}, x2 = {0, 0};
        if (use2 && any_exc) {
            const PJB_GLOBAL u32 *gexc = gcodes2 + n2w + K0_CODES2_PAD;
            const u32 g1 = (u32)pos + a + nl;
            x0 = gload_as<Words2>(gexc + ((u32)pos >> 11));
            x1 = gload_as<Words2>(gexc + (g1 >> 11));
            if (two) x2 = gload_as<Words2>(gexc + ((g1 + b2 + nl2) >> 11));
        }
        // the next trips' loads go out before this trip's compares: its list records (two ahead), then the operations and fields (one
        // ahead) -- they are in flight while this trip is compared
        const bool more = v + 1 < v_hi, more2 = v + 2 < v_hi;
        EmitTrip T2 = T1;
        EmitRec R2 = R1;
        EmitOps On = O;
        if (more2) {
            T2 = locate(v + 2);
            R2 = fetch_rec(T2);
        }
        K1E_MARK(4); // records of the trip after the next issued
        if (more) On = fetch_ops(T1, R1);
        K1E_MARK(6); // next operations issued
        // One read's pairs in closed form (junction_system.cc:140-210 for one or two N operations): its blocks of bases compared with
        // gathers, in 2 bits where that is exact and on the 4-bit codes elsewhere.
        const u32 *seqw = (const u32 *)b.seq4 + so;
        const int32_t q_limit = (int32_t)min(seq_words - 1u - so, 0x7fffffffu);
        if (simple) {
            const int32_t vpos = pos + voff;
            const int32_t aend_all = vpos + (int32_t)(a + nl + b2 + nl2 + b3) - 1;
            // ---- the pairs' geometry (junction_system.cc:140-210 for one or two N operations)
            const u32 npairs = two ? 2u : 1u;
            int32_t ist_[2], lst_[2], rend_[2], iend_[2];
            u32 ud_[2];
            {
                int32_t lst = vpos; // the left block of the pair: where it starts, its length; intron; right block
                u32 la = a, ln_ = nl, lb = b2;
#pragma unroll
                for (u32 pr = 0; pr < 2; pr++) {
                    const int32_t istart = lst + (int32_t)la;
                    const int32_t rStartU = istart + (int32_t)ln_;
                    int32_t rStart = rStartU;
                    if (rStart - 1 >= vlen) rStart = vlen - 1; // junction_system.cc:169-171
                    const int32_t iend = rStart - 1;
                    int32_t rEndExc = rStartU + (int32_t)lb;
                    if (rEndExc - 1 >= vlen) rEndExc = vlen; // junction_system.cc:172-174
                    if (pr < npairs && rEndExc - 1 < iend) set_error(err_ptr(), g, PJB_ERR_MIN_ANCHOR); // intron.cc:76
                    ist_[pr] = istart;
                    lst_[pr] = lst;
                    rend_[pr] = rEndExc - 1;
                    iend_[pr] = iend;
                    // junction.cc:795-812.  One N operation: nothing upstream; "downstream" counts the operation itself unless its end was
                    // clamped.  Two: the first has the second downstream, the second the first upstream.
                    ud_[pr] = two ? (pr == 0 ? (1u << 16) : 1u) : (rStartU <= iend + 1 ? 0u : (1u << 16));
                    lst = rStart;
                    la = lb;
                    ln_ = nl2;
                    lb = b3;
                }
            }
            // ---- the anchors' match statistics: every block of bases is compared once -- the block between two introns is the first
            // pair's right anchor and the second pair's left one.  Block k: read [bq, bq + bl) against the target's [bg, bg + bl).
            const int32_t bq[3] = {(int32_t)dS, (int32_t)(dS + a), (int32_t)(dS + a + b2)};
            const int32_t bg[3] = {lst_[0] - voff, iend_[0] + 1 - voff, iend_[1] + 1 - voff};
            const int32_t bl[3] = {(int32_t)a, rend_[0] - iend_[0], rend_[1] - iend_[1]};
            CmpBlock res[3] = {{bl[0], 0, -1, -1}, {bl[1], 0, -1, -1}, {bl[2], 0, -1, -1}};
            // The gathers, a ROUND at a time, every lane through its own blocks one after the other: a lane with a long left anchor
            // has a short right one, so the wavefront needs max (rounds of all blocks of a lane) iterations where block after block it
            // needed max (rounds of block 0) + max (rounds of block 1) + ...; each iteration is a memory round trip as well.
            const int nb = two ? 3 : 2;
            { // a stretch with a character outside ACGT under one of the blocks: the lane compares on the 4-bit codes
                auto hit = [&](const Words2 &w, int32_t gi, int32_t l) {
                    const u32 b0 = (u32)gi >> 6, cnt = (((u32)(gi + l - 1)) >> 6) - b0 + 1u; // 1 .. 31 stretches
                    const u64 bits = (((u64)1 << cnt) - 1u) << (b0 & 31u);
                    return (((u64)w.x | ((u64)w.y << 32)) & bits) != 0;
                };
                if (any_exc && use2 && (hit(x0, bg[0], bl[0]) || hit(x1, bg[1], bl[1]) || (two && hit(x2, bg[2], bl[2])))) use2 = false;
            }
#ifdef K1E_HIST
            u32 h_rounds = 0, h_mine = 0, h_rounds4 = 0;
#endif
            if (__ballot(use2)) {
                const u32 *seq2w = (const u32 *)b.seq2 + (so >> 1);
                const int32_t q2_last = (int32_t)min(((seq_words + 1u) >> 1) - 1u - (so >> 1), 0x7fffffffu); // the batch's last word of 2-bit bases, from the read's first
                const int32_t g2_last = (int32_t)(n2w + K0_CODES2_PAD - 1);
                const u32 qodd = (so & 1u) * 16u;
                int k = use2 ? 0 : nb;
                int32_t t = 0;
                u32 mism = 0, fbit = 0xffffffffu, lbit1 = 0;
                for (;;) {
                    const int32_t l = k == 0 ? bl[0] : k == 1 ? bl[1] : bl[2];
                    const bool act = k < nb && t < l;
                    if (act) {
                        const int32_t qi = k == 0 ? bq[0] : k == 1 ? bq[1] : bq[2], gi = k == 0 ? bg[0] : k == 1 ? bg[1] : bg[2];
                        const u32 qbit = qodd + 2u * (u32)(qi + t), gbit = 2u * (u32)(gi + t);
                        u32 qw[C2_NW], gw[C2_NW];
                        const int32_t qf = (int32_t)(qbit >> 5), gf = (int32_t)(gbit >> 5);
                        if (C2_SKIP_HI && qf + C2_NW - 1 <= q2_last) { // (the genome's words are always there: K0_CODES2_PAD)
                            const Words4 qa = gload(reinterpret_cast<const Words4 *>(seq2w + qf)), ga = gload_as<Words4>(gcodes2 + gf);
                            Words4 qb = {0, 0, 0, 0}, gb = {0, 0, 0, 0};
                            if (l - t > 48) { // (words 4 .. 7 feed the bases from the 49th on)
                                qb = gload(reinterpret_cast<const Words4 *>(seq2w + qf + 4));
                                gb = gload_as<Words4>(gcodes2 + gf + 4);
                            }
                            qw[0] = qa.x, qw[1] = qa.y, qw[2] = qa.z, qw[3] = qa.w, qw[4] = qb.x, qw[5] = qb.y, qw[6] = qb.z, qw[7] = qb.w;
                            gw[0] = ga.x, gw[1] = ga.y, gw[2] = ga.z, gw[3] = ga.w, gw[4] = gb.x, gw[5] = gb.y, gw[6] = gb.z, gw[7] = gb.w;
                        } else {
                            load_words<C2_NW>(qw, seq2w, qf, q2_last);
                            load_words<C2_NW>(gw, (const u32 *)gcodes2, gf, g2_last);
                        }
                        chunk2_cmp_bits<C2_NW>(qw, gw, qbit & 31u, gbit & 31u, l, t, mism, fbit, lbit1);
                        t += C2_ROUND;
#ifdef K1E_HIST
                        h_mine++;
#endif
                    }
                    if (k < nb && t >= l) { // the lane's block is finished: its results, the next block
                        const int32_t first = (int32_t)fbit >> 1, last = ((int32_t)lbit1 - 1) >> 1; // (-1: no mismatch)
                        if (k == 0) res[0].mism = (int32_t)mism, res[0].first = first, res[0].last = last;
                        if (k == 1) res[1].mism = (int32_t)mism, res[1].first = first, res[1].last = last;
                        if (k == 2) res[2].mism = (int32_t)mism, res[2].first = first, res[2].last = last;
                        k++;
                        t = 0, mism = 0, fbit = 0xffffffffu, lbit1 = 0;
                    }
#ifdef K1E_HIST
                    h_rounds++;
#endif
                    if (!__ballot(k < nb)) break;
                }
            }
            if (__ballot(!use2)) { // ---- on the 4-bit codes (rounds of 56 bases)
                int k = use2 ? nb : 0;
                int32_t t = 0;
                u32 mism = 0, fbit = 0xffffffffu, lbit1 = 0;
                for (;;) {
                    // (blocks without bases -- a clamped end -- are stepped over)
                    const int32_t l = k == 0 ? bl[0] : k == 1 ? bl[1] : bl[2];
                    const bool act = k < nb && t < l;
                    if (act) {
                        const int32_t qi = k == 0 ? bq[0] : k == 1 ? bq[1] : bq[2], gi = k == 0 ? bg[0] : k == 1 ? bg[1] : bg[2];
                        CmpChunkT<SIMPLE_NW> C;
                        chunk_load<SIMPLE_NW, true>(C, seqw, qi, q_limit, (const u32 *)gcodes, gi, g_words, l, t);
                        chunk_cmp_bits<SIMPLE_NW>(C, qi, gi, l, t, mism, fbit, lbit1);
                        t += 8 * (SIMPLE_NW - 1);
                    }
                    if (k < nb && t >= l) { // the lane's block is finished (or empty): its results, the next block
                        const int32_t first = (int32_t)fbit >> 2, last = ((int32_t)lbit1 - 1) >> 2; // (-1: no mismatch)
                        if (k == 0) res[0].mism = (int32_t)mism, res[0].first = first, res[0].last = last;
                        if (k == 1) res[1].mism = (int32_t)mism, res[1].first = first, res[1].last = last;
                        if (k == 2) res[2].mism = (int32_t)mism, res[2].first = first, res[2].last = last;
                        k++;
                        t = 0, mism = 0, fbit = 0xffffffffu, lbit1 = 0;
                    }
#ifdef K1E_HIST
                    h_rounds4++;
#endif
                    if (!__ballot(k < nb)) break;
                }
            }
#ifdef K1E_HIST
            {
                const u32 lanes = (u32)__popcll(__ballot(true)), sum = wave_total<DppAdd>(h_mine);
                const u32 r_ = min(h_rounds, 31u), r4 = min(wave_total<DppMax>(h_rounds4), 31u);
                if (lane_id() == (int)(__ffsll((long long)__ballot(true)) - 1)) {
                    atomicAdd(&g_k1e_hist[0][r_], 1ull);
                    atomicAdd(&g_k1e_hist[1][r_], (unsigned long long)sum);
                    atomicAdd(&g_k1e_hist[2][r_], (unsigned long long)lanes);
                    atomicAdd(&g_k1e_hist[3][r4], 1ull);
                }
            }
#endif
#pragma unroll
            for (u32 pr = 0; pr < 2; pr++) {
                if (pr >= npairs) break;
                PairRec Q;
                Q.lstart = lst_[pr];
                Q.rend = rend_[pr];
                Q.pos = vpos;
                Q.aend = aend_all;
                Q.meta = meta | META_SIMPLE | (two ? META_MULTI : 0u);
                Q.updown = ud_[pr];
                Q.aux = cmp_blocks_res(res[pr], res[pr + 1]);
                const u64 key = make_key(kf, ist_[pr], iend_[pr]);
                P.key[off + pr] = key;
                if (want_g) pg_ptr()[off + pr] = g;
                rec_store(P.rec + off + pr, Q);
                if (want_cand) cand_insert(key, Q.lstart, Q.rend);
            }
        }
        const u64 p1_entry = (u64)(E.pack_nn ? g | (2u << 28) : g) | ((u64)off << 32);
        K1E_MARK(7); // compares, records, candidates
        ctx.list_write(2, two, base2, p1_entry, shard);
        ctx.list_write(3, generic, base3, (u64)R.slot | ((u64)off << 32), shard); // (the read's place in the tiles' lists, its first pair)
        K1E_MARK(8); // list entries
        // (the candidate set is flushed when the block leaves: a set that fills up before that sends its keys straight to the list --
        // cand_insert -- and no barrier ties the block's wavefronts together trip by trip)
        if (!more) ctx.cand_flush(true);
        K1E_MARK(9); // candidate flush (barrier)
        T = T1;
        R = R1;
        O = On;
        T1 = T2;
        R1 = R2;
    }
}
#endif // PJB_KERNELS_CHAIN

// The reads k1_emit did not finish (three and more introns, indels, = X P H operations, clamped ends, exotic targets, SEQ '*'):
// a thread per read of the chain's third list, dense.  The read's operations are walked (emit_read_pairs): keys and records
// of its pairs, candidates, and -- for [S] B (N B)+ [S] -- the blocks' compares; then the read goes on one of k4b_generic's
// lists.  One launch per chain, behind the chain's k1_emit launches.
__device__ __forceinline__ const DevBatch &find_batch_by_tile(const DevBatch *batches, int n_batches, u32 tile) {
    int lo = 0, hi = n_batches - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (batches[mid].tile_base <= tile) lo = mid;
        else hi = mid - 1;
    }
    return batches[lo];
}
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(K1E_T) void k1_generic(const DevBatch *batches, int n_batches, const u32 *spl_idx, const uint4 *spl_rec, Pairs P, EmitLists E, KeyFmt kf,
                                                    GroupTab G, int use_codes, int orientation, u64 *err, ContigStats *cs) {
    __shared__ EmitShared sh;
    __shared__ u32 s_ops[OPS_LDS][K1E_T];
    __shared__ u32 s_first[GEN_SHARDS + 1];
    __shared__ u32 s_wsum[4];
    if (cs->P == 0) return;
    const bool want_cand = E.cand != nullptr;
    EmitCtx ctx{sh, E, kf, cs, want_cand};
    ctx.init();
    static_assert(GEN_SHARDS == K1E_T, "a sub-list per thread");
    {
        const u32 c = E.gen_cnt[threadIdx.x * GEN_CNT_STRIDE + 4];
        u32 total;
        const u32 ex = block_escan<K1E_T / 64>(c < E.gen_cap ? c : E.gen_cap, s_wsum, &total);
        s_first[threadIdx.x] = ex;
        if (threadIdx.x == K1E_T - 1) s_first[GEN_SHARDS] = total;
        __syncthreads();
    }
    const u32 n_items = s_first[GEN_SHARDS];
    for (u32 item0 = blockIdx.x * K1E_T; item0 < n_items; item0 += gridDim.x * K1E_T) {
        const u32 item = item0 + threadIdx.x;
        u32 gen_pairs = 0, gen_kind = 0;
        u64 gen_entry = 0;
        if (item < n_items) {
            u32 sub = 0; // the last sub-list with s_first[sub] <= item
#pragma unroll
            for (u32 step = GEN_SHARDS / 2; step > 0; step >>= 1)
                if (sub + step < GEN_SHARDS && s_first[sub + step] <= item) sub += step;
            const u64 e3 = E.gen_list[((size_t)2 * GEN_SHARDS + sub) * E.gen_cap + (item - s_first[sub])];
            const u32 slot = (u32)e3;
            const DevBatch &b = find_batch_by_tile(batches, n_batches, slot / (u32)K1_TILE);
            const u32 r = spl_idx[slot];
            const uint4 sr = spl_rec[slot];
            const int m = b.member;
            const int32_t voff = G.voff[m], ref_len = G.len[m], vlen = voff + ref_len;
            const u32 *gcodes = use_codes ? G.codes[m] : (const u32 *)nullptr;
            EmitRead R;
            R.n = sr.w & 0x7fffu;
            if (R.n == 0x7fffu) R.n = gload(b.cig_off + r + 1) - sr.x;
            R.lq = (int32_t)(sr.w >> 16);
            if (R.lq == 0xffff) R.lq = gload(b.l_qseq + r);
            const int32_t pos = (int32_t)sr.y;
            R.pos = pos + voff;
            R.g = b.base + r;
            R.meta = read_meta(gload(b.flag + r), (u32)gload(b.xs + r), gload(b.mapq + r), pos, gload(b.mtid + r), gload(b.mpos + r), G.tid[m], orientation);
            R.seq_ok = (sr.w & 0x8000u) != 0;
            R.off = (u32)(e3 >> 32);
            OpsViewT<K1E_T, OPS_LDS> cig;
            cig.g = b.cigar + sr.x;
            cig.lds = &s_ops[0][threadIdx.x];
#pragma unroll
            for (int q = 0; q < OPS_LDS; q++) { // (unconditional loads, masked: see k1_count)
                const bool has = (u32)q < R.n;
                const u32 v = gload(has ? cig.g + q : b.cig_off);
                s_ops[q][threadIdx.x] = has ? v : 0u;
            }
            // operations counted, and the shape [S] B (N B)+ [S] recognised, B = M = X I D operations that begin and end with bases on
            // both sides: state 0 at the first operation, 5 after a leading S, 1 after M = X, 6 after I or D, 2 after an N, 3 after the
            // closing S, 4 any other shape
            u32 nN = 0, shape = 0;
            int32_t aligned = 0;
            int64_t qsum = 0, bsum = 0; // query bases; positions the walks emit for the current block
            for (u32 q = 0; q < R.n; q++) {
                const u32 o = cig[q], ty = o & 15u, ln = o >> 4;
                nN += (ty == OP_N);
                if (op_consumes_ref(ty)) aligned += (int32_t)ln;
                if (op_consumes_query(ty)) qsum += ln;
                const bool len_ok = ln > 0 && ln <= RES_FIELD_MAX;
                if (ty == OP_M || ty == OP_EQ || ty == OP_X) shape = (shape == 0 || shape == 5 || shape == 2 || shape == 1 || shape == 6) && len_ok ? 1u : 4u;
                else if (ty == OP_I || ty == OP_D) shape = PJB_CLOSED_INDELS && shape == 1 && len_ok ? 6u : 4u;
                else if (ty == OP_N) shape = shape == 1 && ln > 0 ? 2u : 4u;
                else if (ty == OP_S) shape = shape == 0 && len_ok ? 5u : shape == 1 && q + 1 == R.n ? 3u : 4u;
                else shape = 4;
                bsum = ty == OP_N ? 0 : ty == OP_S ? bsum : bsum + ln;
                if (bsum > (int64_t)RES_FIELD_MAX) shape = 4;
            }
            if (nN > 1) R.meta |= META_MULTI;
            R.nN = nN;
            R.aend = R.pos + aligned - 1;
            // (nothing clamped: the alignment lies inside its target -- junction_system.cc:169-174)
            const bool closed = gcodes != nullptr && (shape == 1 || shape == 3) && nN > 0 && R.seq_ok && R.lq > 1 && qsum == (int64_t)R.lq &&
                                R.pos >= voff && R.aend < vlen;
            R.closed_seqw = closed ? reinterpret_cast<const u32 *>(b.seq4) + sr.z : nullptr;
            R.q_limit = (int32_t)min(gload(b.seq_off + b.n) - 1u - sr.z, 0x7fffffffu);
            R.gcodes = gcodes;
            R.glen = ref_len;
            R.voff = voff;
            emit_read_pairs(cig, R, P, kf, vlen, err, [&](u64 key, int32_t lstart, int32_t rend) { ctx.cand_insert(key, lstart, rend); });
            gen_pairs = nN;
            gen_kind = closed ? 2u : 1u;
            gen_entry = (u64)(closed && E.pack_nn ? R.g | ((nN < 15u ? nN : 15u) << 28) : R.g) | ((u64)R.off << 32);
        }
        // the read goes on k4b_generic's first list (the walks) or on its second (closed form done, window to be checked)
        const u32 shard = (item0 >> K1E_SHIFT) % GEN_SHARDS;
        ctx.list_append(1, gen_kind == 1, gen_pairs, gen_entry, shard);
        ctx.list_append(2, gen_kind == 2, gen_pairs, gen_entry, shard);
        ctx.cand_flush(item0 + gridDim.x * K1E_T >= n_items);
    }
}
#endif // PJB_KERNELS_CHAIN

// per-member counters of a group, from the tile statistics of the member's tiles (before k1_scan_tiles turns the tile pair
// counts into offsets): one block per member
// (behind k1_scan_tiles and off the chain's K1 stage: a member's pairs are the difference of the scanned counts at its first tile and
// the next member's)
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void kg_member_stats(const u32 *tile_off, const TileStats *ts, const u32 *tile_lo, int n_members, MemberStats *out, u32 n_tiles,
                                                       const ContigStats *cs) {
    __shared__ u64 sm[4][4];
    __shared__ int32_t smi[4][2];
    const int m = blockIdx.x;
    if (m >= n_members) return;
    u64 spl = 0, uns = 0, sum = 0, pairs = 0;
    int32_t mn = INT32_MAX, mx = 0;
    for (u32 t = tile_lo[m] + threadIdx.x; t < tile_lo[m + 1]; t += 256) {
        const TileStats x = ts[t];
        spl += x.spliced;
        uns += x.unspliced;
        sum += x.sum_len;
        mn = min(mn, x.min_len);
        mx = max(mx, x.max_len);
    }
    spl = wave_sum(spl);
    uns = wave_sum(uns);
    sum = wave_sum(sum);
    {
        const u32 t0 = tile_lo[m], t1 = tile_lo[m + 1], total = (u32)cs->n_pairs;
        pairs = threadIdx.x == 0 ? (u64)((t1 < n_tiles ? tile_off[t1] : total) - (t0 < n_tiles ? tile_off[t0] : total)) : 0ull;
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0) {
        sm[w][0] = spl;
        sm[w][1] = uns;
        sm[w][2] = sum;
        sm[w][3] = pairs;
        smi[w][0] = mn;
        smi[w][1] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        MemberStats S;
        S.spliced = sm[0][0] + sm[1][0] + sm[2][0] + sm[3][0];
        S.unspliced = sm[0][1] + sm[1][1] + sm[2][1] + sm[3][1];
        S.sum_len = sm[0][2] + sm[1][2] + sm[2][2] + sm[3][2];
        S.n_pairs = sm[0][3] + sm[1][3] + sm[2][3] + sm[3][3];
        S.min_len = min(min(smi[0][0], smi[1][0]), min(smi[2][0], smi[3][0]));
        S.max_len = max(max(smi[0][1], smi[1][1]), max(smi[2][1], smi[3][1]));
        S.n_junc = 0; // (counted by k5_finalize)
        S._pad = 0;
        out[m] = S;
    }
}
#endif // PJB_KERNELS_CHAIN

// ---------------------------------------------------------------------------------------------
// K2d: ordered dense junction ids.  The intron key is 46-48 bits wide (contig coordinate + intron length): five radix
// passes, the first of them over digits that differ for every junction of a tile.  But a contig has 10^4-10^5
// distinct introns, and their RANK in (start, end) order needs 15-19 bits: two passes -- and because the ranks of
// the junctions a tile touches are neighbours (pairs arrive in BAM order, ranks are ordered by start), both passes
// scatter into a few long runs per tile.  The rank comes without sorting anything:
//   k1_emit    pairs -> candidate keys (distinct per block residency; see there)
//   (k1_emit / k1_generic also set one bit per contig base where an intron of a listed candidate starts)
//   scan       prefix popcount over the bitmap words  -> rank of a start among the distinct starts
//   kd_ends    per start rank, the distinct intron ends seen (alternative acceptors: a handful; DENSE_ENDS slots)
//   scan       number of ends per start rank           -> first junction id of every start (+ the anchors' rest state)
//   kd_table   junction id -> intron key and the junction's anchors (min lStart, max rEnd over its pairs, junction.cc:477-529),
//              from the candidates; closes the chain if a limit was exceeded
//   kd_assign  id of a pair = first id of its start + number of that start's ends below its own end
// Grouping by id is grouping by (start, end), id order is (start, end) order, so everything downstream -- segment
// heads, fragments, row order -- works on the ids as it did on the keys.  A start with more than DENSE_ENDS different
// ends raises OVF_DENSE and the contig is repeated with the full-key sort.
// ---------------------------------------------------------------------------------------------
constexpr int DENSE_ENDS = 8;
constexpr u32 DENSE_EMPTY = 0xffffffffu;

__device__ __forceinline__ u32 start_rank(const u64 *bitmap, const u32 *wrank, int32_t start) {
    const u32 w = (u32)start >> 6;
    return wrank[w] + (u32)__popcll(bitmap[w] & ((1ull << (start & 63)) - 1ull));
}
// Pairs arrive in BAM order, so the pairs of a junction sit close together, and a deep junction is one key repeated 10^5
// times.  Operations on device memory that many waves aim at one address (or one cache line: the bitmap words of
// neighbouring starts) are served one after the other by a single L2 channel -- measured, 0.5 ns each, 1.4 ms for the
// pairs of one contig.  So only the CANDIDATE list (1-3x the number of junctions) touches the bitmap and the end slots.
// candidates -> one bit per contig base: an intron starts here
struct PopcFn {
    const u64 *words;
    __device__ u64 operator()(u64 i) const { return (u64)__popcll(words[i]); }
};
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void kd_ends(const u64 *cand, KeyFmt kf, const u64 *bitmap, const u32 *wrank, u32 junc_limit, u32 *ends,
                                               u32 *cand_rank, ContigStats *cs) {
    const u32 n = cs->n_cand; // (a few candidates per junction: the grid is small and strides)
    for (u32 p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
        int32_t s, e;
        unpack_key(kf, cand[p], s, e);
        const u32 rs = start_rank(bitmap, wrank, s);
        cand_rank[p] = rs;
        if (rs >= junc_limit) {
            atomicOr(&cs->overflow, OVF_JUNC);
            continue;
        }
        u32 *slot = ends + (size_t)rs * DENSE_ENDS;
        const u32 ue = (u32)e;
        bool placed = false;
        for (int k = 0; k < DENSE_ENDS && !placed; k++) {
            const u32 cur = atomicCAS(&slot[k], DENSE_EMPTY, ue);
            placed = cur == DENSE_EMPTY || cur == ue; // else: the slot holds another end (slots never change once set)
        }
        if (!placed) atomicOr(&cs->overflow, OVF_DENSE);
    }
}
#endif // PJB_KERNELS_CHAIN
// The ranks of the bitmap's words, page by page: the starts are few (a third of the pages of a human-sized chain hold one, fewer
// where genes cluster), so the prefix sum runs over the PAGES' counts (k1_emit / k1_generic count a start when its bit is set for
// the first time) and only the pages that hold a start are read: a wavefront per page, lane = word, the word's rank = the page's
// rank + the popcounts of the page's words before it.  Words of pages without a start keep whatever rank they had: nobody asks
// for it (kd_ends, kd_assign look up the words of their own starts).  (Until round 5 a three-kernel scan read all of the bitmap
// twice and wrote every word's rank: 320 MB and 90 us a 1-Gb chain.)
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void kd_rank_pages(const u64 *bitmap, const u32 *page_cnt, const u32 *page_rank, u32 *wrank, u32 n_pages, u32 n_words) {
    const u32 page = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (page >= n_pages) return;
    if (page_cnt[page] == 0) return; // (uniform per wavefront)
    const u32 w = (page << KD_PAGE_SHIFT) + (u32)lane_id();
    const u32 c = w < n_words ? (u32)__popcll(bitmap[w]) : 0u;
    const u32 inc = wave_iscan(c);
    if (w < n_words) wrank[w] = page_rank[page] + inc - c;
}
#endif // PJB_KERNELS_CHAIN
// Bitmap, page counts and end slots are all-clear at rest: instead of memsets over contig-sized buffers per contig, the
// candidates wipe exactly what they set (after kd_assign has read it).
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void kd_reset(const u64 *cand, const u32 *cand_rank, KeyFmt kf, u32 junc_limit, const ContigStats *cs,
                                                u64 *bitmap, u32 *ends, u32 *page_cnt) {
    const u32 n = cs->n_cand;
    for (u32 p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
        int32_t s, e;
        unpack_key(kf, cand[p], s, e);
        bitmap[(u32)s >> 6] = 0;
        page_cnt[(u32)s >> (6 + KD_PAGE_SHIFT)] = 0;
        const u32 rs = cand_rank[p];
        if (rs < junc_limit) {
            uint4 *q = reinterpret_cast<uint4 *>(ends + (size_t)rs * DENSE_ENDS);
            q[0] = q[1] = make_uint4(DENSE_EMPTY, DENSE_EMPTY, DENSE_EMPTY, DENSE_EMPTY);
        }
    }
}
#endif // PJB_KERNELS_CHAIN
struct EndsCountFn {
    const u32 *ends;
    __device__ u64 operator()(u64 rs) const {
        const uint4 *q = reinterpret_cast<const uint4 *>(ends + rs * DENSE_ENDS);
        const uint4 a = q[0], b = q[1];
        return (u64)((a.x != DENSE_EMPTY) + (a.y != DENSE_EMPTY) + (a.z != DENSE_EMPTY) + (a.w != DENSE_EMPTY) + (b.x != DENSE_EMPTY) +
                     (b.y != DENSE_EMPTY) + (b.z != DENSE_EMPTY) + (b.w != DENSE_EMPTY));
    }
};
// sink of the scan over the start ranks (junc_limit entries): first junction id of the start, and -- entry i of the
// junction-sized anchor arrays -- the rest state kd_assign's atomics start from
struct FirstIdSink {
    u32 *first_id;
    int32_t *anc_l, *anc_r;
    __device__ void operator()(u64 i, u64, u64 ex) const {
        first_id[i] = (u32)ex;
        anc_l[i] = INT32_MAX;
        anc_r[i] = INT32_MIN;
    }
};
__device__ __forceinline__ u32 ends_below(const u32 *ends, u32 rs, u32 ue) { // (DENSE_EMPTY slots compare as larger than any end)
    const uint4 *q = reinterpret_cast<const uint4 *>(ends + (size_t)rs * DENSE_ENDS);
    const uint4 a = q[0], b = q[1];
    return (a.x < ue) + (a.y < ue) + (a.z < ue) + (a.w < ue) + (b.x < ue) + (b.y < ue) + (b.z < ue) + (b.w < ue);
}
// Junction anchors from pairs that sit in the lanes of a wavefront (any order): leftAncStart = min lStart, rightAncEnd = max
// rEnd (junction.cc:477-529).  In BAM order the pairs of the two or three junctions of a locus alternate from lane to lane, so
// the wavefront takes its DISTINCT junctions one after the other -- all lanes of one junction are folded on the DPP path,
// whatever lies between them -- and one lane per junction touches the arrays, and only if the value would move them (a
// read at L2 first: the arrays only ever move one way, so an older value errs on the side of one atomic too many).  (Folding
// neighbouring lanes only left one atomic per lane wherever junctions alternate: 1.4 ms per chain on one L2 channel.)
// segmented (by key) reduce towards the segment's FIRST lane; equal keys are contiguous across the lanes
template <typename T, typename OP>
__device__ __forceinline__ T seg_reduce_to_head(T v, u32 segkey, OP op) {
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        T t = __shfl_down(v, o, 64);
        u32 k = __shfl_down(segkey, o, 64);
        if (l + o < 64 && k == segkey) v = op(v, t);
    }
    return v;
}
struct OpMin { template <typename T> __device__ T operator()(T a, T b) const { return a < b ? a : b; } };
struct OpMax { template <typename T> __device__ T operator()(T a, T b) const { return a > b ? a : b; } };
struct OpAdd { template <typename T> __device__ T operator()(T a, T b) const { return a + b; } };
__device__ __forceinline__ void anchors_fold(bool valid, u32 j, int32_t l, int32_t r, int32_t *anc_l, int32_t *anc_r) {
    const u32 lk = (u32)l ^ 0x80000000u, rk = (u32)r ^ 0x80000000u; // (signed order through the sign bit)
    u64 todo = __ballot(valid);
    while (todo) {
        const int first = __ffsll((long long)todo) - 1;
        const u32 jc = (u32)__builtin_amdgcn_readlane((int)j, first);
        const bool mine = valid && j == jc;
        const u64 m = __ballot(mine);
        const int32_t lo = (int32_t)(wave_total<DppMin>(mine ? lk : 0xffffffffu) ^ 0x80000000u);
        const int32_t hi = (int32_t)(wave_total<DppMax>(mine ? rk : 0u) ^ 0x80000000u);
        if (lane_id() == first) { // (the look goes to L2, where the atomics are done: a CU's L1 would keep showing it the line it fetched first)
            if (lo < __hip_atomic_load(&anc_l[jc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&anc_l[jc], lo);
            if (hi > __hip_atomic_load(&anc_r[jc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&anc_r[jc], hi);
        }
        todo &= ~m;
    }
}

// junction id -> intron key (every candidate writes its junction's entry: duplicates write the same value) and the junction's
// anchors from the candidates' partial ones; thread 0 closes the chain -- P = 0, nothing downstream runs, the host repeats the
// contig -- if a limit was exceeded while the ids were built
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void kd_table(const u64 *cand, const u64 *cand_anc, const u32 *cand_rank, KeyFmt kf, u32 junc_limit, const u32 *ends,
                                                const u32 *first_id, const u64 *total, u64 *jkey, int32_t *anc_l, int32_t *anc_r, ContigStats *cs,
                                                const u32 *gen_cnt, u32 gen_cap, u32 sort_limit) {
    const u32 p = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0) { // (the read lists: every sub-list within its room?)
        static_assert(GEN_SHARDS == 256, "a shard per thread of the first block");
        const u32 over = lists_over(gen_cnt, threadIdx.x, gen_cap);
        if (over) atomicMax(&cs->list_need, over);
        if (__syncthreads_or((int)over) && threadIdx.x == 0) cs->overflow |= OVF_LISTS;
    }
    if (p == 0) {
        const u64 J = *total;
        u32 ovf = cs->overflow;
        // (sort_limit: the ids the sort's digits were planned for -- what chains of this context have had so far, with room; junc_limit:
        // what the buffers hold)
        if (cs->P != 0 && (J > (u64)junc_limit || J > (u64)sort_limit)) {
            ovf |= OVF_JUNC;
            cs->overflow = ovf;
            cs->n_junc = (u32)(J < 0xffffffffull ? J : 0xffffffffull);
        }
        if (ovf) cs->P = 0;
        else if (cs->P) { // the ids ARE the junctions: their number and the fragment slots are known from here on
            const u32 n_pairs = cs->P;
            cs->n_junc = cs->J = (u32)J;
            cs->n_slices = (n_pairs + 63) / 64;
            cs->n_slots = (u32)J + (n_pairs + 63) / 64;
        }
    }
    const u32 n = cs->n_cand;
    for (u32 base = blockIdx.x * 256u; base < n; base += gridDim.x * 256u) {
        const u32 q = base + threadIdx.x;
        const bool on = q < n;
        const u32 pc = on ? q : n - 1;
        const u32 rs = cand_rank[pc];
        const u64 k = cand[pc], a = cand_anc[pc];
        bool valid = on && rs < junc_limit;
        u32 id = 0xffffffffu;
        if (valid) {
            int32_t s, e;
            unpack_key(kf, k, s, e);
            id = first_id[rs] + ends_below(ends, rs, (u32)e);
            valid = id < junc_limit;
            if (valid) jkey[id] = k;
        }
        // (a junction has a candidate or two: the lanes of a wavefront hold different junctions, there is nothing to fold -- each lane
        // moves its junction's anchors itself, and only if its value would move them)
        if (valid) {
            const int32_t lo = (int32_t)(u32)a, hi = (int32_t)(u32)(a >> 32);
            if (lo < __hip_atomic_load(&anc_l[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&anc_l[id], lo);
            if (hi > __hip_atomic_load(&anc_r[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&anc_r[id], hi);
        }
    }
}
#endif // PJB_KERNELS_CHAIN

// fragment record of the per-junction reductions: 48 words (see k4_pairs)
enum {
    F_N = 0, F_R1P, F_R1N, F_R2P, F_R2N, F_MS, F_XSP, F_XSN, F_UM, F_BPP, F_PPP, F_REL, F_DIST, // sums
    F_MAXMINANC, F_UP, F_DOWN, F_MAXMMES, F_MAXMINMATCH,                                       // max
    F_FIRSTMIS,                                                                                 // min
    F_PAD19,                                                                                    // keeps the next pair 8-byte aligned
    F_MISM_LO, F_MISM_HI,                                                                       // 64-bit sum
    F_JAD0,                                                                                     // 20 sums
    F_WORDS = 48
};
__device__ __forceinline__ void acc_rest_state(u32 *acc, u64 n_junc) { // what k5_frag_reduce's atomics start from: sums 0, max 0, min 100000000
    const u64 n = n_junc * F_WORDS, step = (u64)gridDim.x * blockDim.x;
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += step) acc[t] = (t % F_WORDS) == F_FIRSTMIS ? 100000000u : 0u;
}

// (the sort's tile and its digit counter, defined with the sort below: kd_assign counts the first digit of the ids it hands out)
constexpr int KDA_TILE = 4096; // = RS_TILE
__device__ __forceinline__ void wave_hist_add(u32 *h, u32 d, bool valid);
constexpr int KDA_PER = 4;
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void kd_assign(const u64 *key, const u32 *np, KeyFmt kf, const u64 *bitmap, const u32 *wrank, const u32 *ends,
                                                 const u32 *first_id, u32 junc_limit, const u64 *total, u32 *jid_bam, u32 *acc, const u64 *jkey,
                                                 const int32_t *anc_l, const int32_t *anc_r, u64 *err, ContigStats *cs_chk, int hist_bits, u32 *hist) {
    // A block takes one TILE of the sort (4 096 pairs, four chunks of 1 024) and leaves the tile's counts of the ids' first digit
    // where rs_hist would have put them (hist_bits > 0): the sort's first pass starts at its scan -- the ids are not read a second
    // time to be counted.
    __shared__ u32 s_h[4096]; // (RS_MAX_BINS)
    const u32 n = *np;
    if (n == 0) { // (a chain without pairs, or closed by an overflow: the tile's counts are still this kernel's to write -- zeros --, the
                  // panel kernels behind it read them)
        if (hist_bits > 0)
            for (u32 d = threadIdx.x; d < (1u << hist_bits); d += 256) hist[(size_t)blockIdx.x * (1u << hist_bits) + d] = 0;
        return;
    }
    {
        const u64 J = *total;
        acc_rest_state(acc, J < (u64)junc_limit ? J : (u64)junc_limit);
    }
    const u32 nb = hist_bits > 0 ? 1u << hist_bits : 0u;
    for (u32 d = threadIdx.x; d < nb; d += 256) s_h[d] = 0;
    if (nb) __syncthreads();
    for (u32 chunk = blockIdx.x * (KDA_TILE / (KDA_PER * 256)); chunk < (blockIdx.x + 1) * (KDA_TILE / (KDA_PER * 256)); chunk++) {
    if (chunk * (KDA_PER * 256) >= n) break;
    // KDA_PER pairs a thread, their loads side by side: a pair is a chain of three dependent look-ups (key -> bitmap word and rank ->
    // end slots and first id), and a wavefront with one pair per lane (651 k of them a chain) spent its life waiting for them one
    // after the other -- 134 us a chain (round 4) for 267 MB
    // Pairs arrive in BAM order: the lanes of a wavefront mostly hold the pairs of one or two junctions.  Only the first lane of a run
    // of equal keys (a "leader") looks its junction up -- the look-ups are gathers of 56 bytes a pair, and their cost follows the
    // lanes that take part -- the others take the leader's answer (one ds_bpermute).
    u64 kk[KDA_PER];
    u32 pp[KDA_PER];
#pragma unroll
    for (int q = 0; q < KDA_PER; q++) {
        pp[q] = (chunk * KDA_PER + q) * 256 + threadIdx.x;
        kk[q] = key[pp[q] < n ? pp[q] : n - 1];
    }
    bool lead[KDA_PER];
    int32_t ss[KDA_PER], ee[KDA_PER];
    u64 bw[KDA_PER];
    u32 wr[KDA_PER];
#pragma unroll
    for (int q = 0; q < KDA_PER; q++) {
        const u32 plo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)kk[q], 0x138, 0xf, 0xf, false);         // wave_shr:1
        const u32 phi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(kk[q] >> 32), 0x138, 0xf, 0xf, false);
        lead[q] = lane_id() == 0 || plo != (u32)kk[q] || phi != (u32)(kk[q] >> 32);
        unpack_key(kf, kk[q], ss[q], ee[q]);
        bw[q] = 0;
        wr[q] = 0;
        if (lead[q]) {
            bw[q] = bitmap[(u32)ss[q] >> 6];
            wr[q] = wrank[(u32)ss[q] >> 6];
        }
    }
    u32 rs_[KDA_PER], fi[KDA_PER], eb[KDA_PER];
#pragma unroll
    for (int q = 0; q < KDA_PER; q++) {
        rs_[q] = wr[q] + (u32)__popcll(bw[q] & ((1ull << (ss[q] & 63)) - 1ull));
        fi[q] = eb[q] = 0;
        if (lead[q]) {
            fi[q] = first_id[rs_[q]];
            eb[q] = ends_below(ends, rs_[q], (u32)ee[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < KDA_PER; q++) {
        const u64 lm = __ballot(lead[q]) & ((2ull << lane_id()) - 1ull); // (lane 0 always leads)
        const u32 id = (u32)__shfl((int)(fi[q] + eb[q]), 63 - __clzll((long long)lm), 64);
        fi[q] = id;
        eb[q] = 0;
        if (pp[q] < n) jid_bam[pp[q]] = id; // (the sort's first pass reads the ids from here; k4b_generic looks its pairs' junctions up)
        if (nb) wave_hist_add(s_h, id & (nb - 1u), pp[q] < n);
    }
#ifdef PJB_SELFCHECK
    const u32 p = pp[0];
    if (p < n) {
    const int32_t s = ss[0], e = ee[0];
    const u32 rs = rs_[0], id = fi[0] + eb[0]; // (debug builds: the junction table kd_table made must know this pair's junction -- its start bit, its end slot, its key,
                     // anchors that enclose the intron; a failure stops the chain (P = 0) so that the report gets out)
    {
        int bad = 0;
        const uint4 *q = reinterpret_cast<const uint4 *>(ends + (size_t)rs * DENSE_ENDS);
        const uint4 a = q[0], b = q[1];
        const u32 ue = (u32)e;
        if (!((bitmap[(u32)s >> 6] >> (s & 63)) & 1ull)) bad = 6;
        else if (!(a.x == ue || a.y == ue || a.z == ue || a.w == ue || b.x == ue || b.y == ue || b.z == ue || b.w == ue)) bad = 7;
        else if (id >= junc_limit || id >= (u32)*total) bad = 1;
        else if (jkey[id] != key[p]) bad = 2;
        else if (anc_l[id] > s || anc_r[id] < e) bad = 3;
        if (bad) {
            set_error(err, 0xfffff000u + (u32)bad, PJB_ERR_HIP);
            cs_chk->P = 0;
        }
    }
    }
#endif
    } // chunks
    if (nb) {
        __syncthreads();
        for (u32 d = threadIdx.x; d < nb; d += 256) hist[(size_t)blockIdx.x * nb + d] = s_h[d];
    }
}
#endif // PJB_KERNELS_CHAIN

// ---------------------------------------------------------------------------------------------
// K2: stable LSD radix sort of (key, pair index).  Classic 3-step passes: per-tile digit
// histogram -> exclusive scan of the bin-major count matrix -> ranked scatter.  Stability comes
// from ranking in memory order: wave w of a tile owns a contiguous 1/4 of it, rounds are
// contiguous 64-element slices, and equal digits inside a round are ranked by lane with a
// ballot-based match-any.
// ---------------------------------------------------------------------------------------------
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = 256 * RS_ITEMS; // 4096 keys per block
constexpr int RS_MAX_BITS = 12;
constexpr int RS_MAX_BINS = 1 << RS_MAX_BITS;
static_assert(KDA_TILE == RS_TILE && RS_MAX_BINS <= 4096, "kd_assign counts the sort's first digit tile by tile");

// Histogram add of one digit per lane.  Keys arrive nearly sorted and deep junctions repeat one key
// thousands of times, so most lanes of a wave often hold the same digit: the two most common
// leading digits are counted by one lane each (no 64-way same-address LDS conflict), the rest
// with plain LDS atomics.
__device__ __forceinline__ void wave_hist_add(u32 *h, u32 d, bool valid) {
    u64 rem = __ballot(valid);
    const int lane = lane_id();
#pragma unroll
    for (int it = 0; it < 2; it++) {
        if (rem == 0) return;
        const int first = __ffsll((long long)rem) - 1;
        const u32 d0 = __shfl(d, first, 64);
        const u64 m = __ballot(valid && d == d0) & rem;
        if (lane == first) atomicAdd(&h[d0], (u32)__popcll(m));
        rem &= ~m;
    }
    if ((rem >> lane) & 1ull) atomicAdd(&h[d], 1u);
}

template <typename K>
__global__ __launch_bounds__(256) void rs_hist(const K *keys, const u32 *np, int shift, int bits, u32 *hist, u32 n_tiles) {
    __shared__ u32 h[RS_MAX_BINS];
    const u32 n = *np; // the grid covers the host's limit; tiles past the data count nothing
    const u32 nb = 1u << bits;
    for (u32 d = threadIdx.x; d < nb; d += 256) h[d] = 0;
    __syncthreads();
    const u32 base = blockIdx.x * RS_TILE;
    const u32 mask = nb - 1;
    K kk[RS_ITEMS];
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const u32 i = base + k * 256 + threadIdx.x;
        kk[k] = i < n ? keys[i] : (K)0;
    }
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const u32 i = base + k * 256 + threadIdx.x;
        wave_hist_add(h, (u32)(kk[k] >> shift) & mask, i < n);
    }
    __syncthreads();
    // tile-major: a tile's counts are one contiguous run (bin-major, every 4-byte store was a write granule of its own:
    // twice the algorithmic traffic, PMC round 2), and rs_scatter reads its tile's scanned counts the same way
    for (u32 d = threadIdx.x; d < nb; d += 256) hist[(size_t)blockIdx.x * nb + d] = h[d];
}

// Exclusive scan over the tiles of every digit's counts (tile order) and the digit totals, on the tile-major matrix, in two
// small kernels over PANELS of RSP_TILES tiles x 64 digits (one wavefront each, lane = digit, so every load is 256
// contiguous bytes of a tile's row): rs_panel_sums adds up each panel's columns; rs_panel_scan lets every panel add the
// sums of the panels before it itself (a few hundred at most) and writes its tiles' exclusive counts; the last panel
// also leaves the digit totals.  (One block per digit walking all tiles -- round 2 -- took 8 us for one chromosome's 650
// tiles and 144 us for a 0.5 Gb chain's 4 400.)  The scatter adds the number of keys with a smaller digit itself (a
// block scan of the 2^bits totals), so the pass needs neither a scan of the whole matrix nor global atomics.
constexpr u32 RSP_TILES = 64;
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void rs_panel_sums(const u32 *hist, u32 n_tiles, u32 nb, u32 *psum) {
    const u32 d = blockIdx.y * 256 + threadIdx.x;
    if (d >= nb) return;
    const u32 t0 = blockIdx.x * RSP_TILES, t1 = t0 + RSP_TILES < n_tiles ? t0 + RSP_TILES : n_tiles;
    u32 sum = 0;
    for (u32 t = t0; t < t1; t += 8) { // eight loads in flight per lane
        u32 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = t + k < t1 ? hist[(size_t)(t + k) * nb + d] : 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) sum += v[k];
    }
    psum[(size_t)blockIdx.x * nb + d] = sum;
}
#endif // PJB_KERNELS_CHAIN
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void rs_panel_scan(const u32 *hist, const u32 *psum, u32 n_tiles, u32 nb, u32 *hist_scan, u32 *row_total) {
    const u32 d = blockIdx.y * 256 + threadIdx.x;
    if (d >= nb) return;
    const u32 panel = blockIdx.x;
    u32 run = 0;
    for (u32 q = 0; q < panel; q += 8) {
        u32 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = q + k < panel ? psum[(size_t)(q + k) * nb + d] : 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) run += v[k];
    }
    const u32 t0 = panel * RSP_TILES, t1 = t0 + RSP_TILES < n_tiles ? t0 + RSP_TILES : n_tiles;
    for (u32 t = t0; t < t1; t += 8) {
        u32 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = t + k < t1 ? hist[(size_t)(t + k) * nb + d] : 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (t + k < t1) hist_scan[(size_t)(t + k) * nb + d] = run;
            run += v[k];
        }
    }
    if (t1 == n_tiles) row_total[d] = run; // (the last panel)
}
#endif // PJB_KERNELS_CHAIN

// LDS of one rs_scatter block: the tile's keys in digit order (reused for the pair indices), the offset
// "global position - tile-local position" of every digit, and the digit counters of the 4 waves
// (16 bit: a wave owns 1024 keys, a tile 4096).
__host__ __device__ constexpr size_t rs_scatter_lds_bytes(int bits, size_t key_bytes = 8) {
    return (size_t)RS_TILE * key_bytes + ((size_t)4 << bits) + ((size_t)8 << bits);
}
constexpr int RS_PF = 2; // digits per thread whose scan entries are prefetched (9-bit digits: all of them)

// Lanes of the wave holding the same digit as this lane ("match any"), one ballot per digit bit:
// peers &= bit set ? ballot : ~ballot, written as peers &= ~(ballot ^ m) with m = 0 / -1 from a signed
// bit-field extract.  BITS > 0 unrolls the loop; BITS == 0 takes the width at run time.
template <int BITS>
__device__ __forceinline__ u64 wave_match_digit(u32 d, bool valid, int bits) {
    u64 peers = __ballot(valid);
    u32 lo = (u32)peers, hi = (u32)(peers >> 32);
    auto step = [&](int bit) {
        const int m = __builtin_amdgcn_sbfe((int)d, bit, 1); // -1 if the bit is set
        const u64 bb = __ballot(m != 0);
        lo &= ~((u32)bb ^ (u32)m);
        hi &= ~((u32)(bb >> 32) ^ (u32)m);
    };
    if constexpr (BITS > 0) {
#pragma unroll
        for (int bit = 0; bit < BITS; bit++) step(bit);
    } else {
        for (int bit = 0; bit < bits; bit++) step(bit);
    }
    return ((u64)hi << 32) | lo;
}

// One radix pass over a 4096-key tile: rank (stable, in memory order), exchange through LDS so that
// the tile leaves in digit order, write out with consecutive lanes on consecutive addresses inside a
// digit run.  The first output slot of (digit, tile) = keys of the whole array with a smaller digit
// (block scan over row_total) + keys with that digit in earlier tiles (hist_scan, bin-major).
// A single-launch variant (digit totals up front, earlier tiles' counts by decoupled look-back over
// 8-byte {epoch, count} granules) was measured at 0.064 ms per pass against 0.058 ms for
// hist + rowscan + scatter: with every tile resident at once the look-back chain costs more than the
// two small kernels, so it was dropped.
// K: u64 (full intron keys) or u32 (dense junction ids: half the key traffic, half the key registers)
template <int BITS, typename K>
__global__ __launch_bounds__(256, 3) void rs_scatter(const K *kin, const u32 *vin, K *kout, u32 *vout, const u32 *np, int shift,
                                                      int bits, const u32 *hist_scan, const u32 *row_total, u32 n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem[];
    __shared__ u64 s_scan[4];
    const u32 n = *np;
    if ((u64)blockIdx.x * RS_TILE >= n) return; // the grid covers the host's limit
    if (BITS > 0) bits = BITS;
    const u32 nb = 1u << bits;
    const u32 mask = nb - 1;
    K *kbuf = (K *)rs_smem;
    u32 *ibuf = (u32 *)rs_smem;
    u32 *delta = (u32 *)(rs_smem + (size_t)RS_TILE * sizeof(K));
    unsigned short *wcnt = (unsigned short *)(delta + nb); // [4][nb]
    for (u32 d = threadIdx.x; d < 2 * nb; d += 256) ((u32 *)wcnt)[d] = 0;
    __syncthreads();
    const u32 tile = blockIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id(); // w in an SGPR: scalar base addresses
    unsigned short *wc = wcnt + (size_t)w * nb;
    const u32 base = tile * RS_TILE + w * (RS_TILE / 4);
    const u64 lt = (1ull << lane) - 1;
    K key[RS_ITEMS];
    u32 val[RS_ITEMS];
    u32 rkp[RS_ITEMS / 2]; // tile-local ranks (< 4096), two per register: the kernel must stay under 128 VGPRs
    // the key / index buffers are allocated with one tile of slack, so the last tile loads unguarded too
    // (lanes past n are masked below): 16 independent loads from one scalar base
    const K *kp = kin + base;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) key[r] = kp[r * 64 + lane];
    // digits [d0, d0 + per) belong to this thread in the digit phase (consecutive, so that one block
    // scan orders them); their row totals / scanned counts are fetched now, behind the ranking
    const u32 per = (nb + 255) / 256;
    const u32 d0 = threadIdx.x * per;
    u32 pf_rt[RS_PF], pf_hs[RS_PF];
#pragma unroll
    for (int k = 0; k < RS_PF; k++) {
        const u32 d = d0 + k;
        const bool on = (u32)k < per && d < nb;
        pf_rt[k] = on ? row_total[d] : 0u;
        pf_hs[k] = on ? hist_scan[(size_t)tile * nb + d] : 0u;
    }
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const u32 i = base + r * 64 + lane;
        const bool valid = i < n;
        const u32 d = (u32)(key[r] >> shift) & mask;
        const u64 peers = wave_match_digit<BITS>(d, valid, bits);
        // only scalars of the peer mask stay live across the LDS round trip
        const u32 lower = (u32)__popcll(peers & lt), group = (u32)__popcll(peers);
        const int leader = valid ? __ffsll((long long)peers) - 1 : lane;
        u32 before = 0;
        if (valid && lane == leader) {
            before = wc[d];
            wc[d] = (unsigned short)(before + group);
        }
        const u32 rank = __shfl(before, leader, 64) + lower;
        rkp[r >> 1] = (r & 1) ? (rkp[r >> 1] | (rank << 16)) : rank;
    }
    // the pair indices are not needed before the second exchange: their latency hides behind the digit phase
    if (vin) {
        const u32 *vp = vin + base;
#pragma unroll
        for (int r = 0; r < RS_ITEMS; r++) val[r] = vp[r * 64 + lane];
    } else {
#pragma unroll
        for (int r = 0; r < RS_ITEMS; r++) val[r] = base + r * 64 + lane;
    }
    __syncthreads();
    u32 mine = 0, below = 0;
    for (u32 k = 0; k < per; k++) {
        const u32 d = d0 + k;
        if (d < nb) {
            mine += (u32)wcnt[d] + wcnt[nb + d] + wcnt[2 * nb + d] + wcnt[3 * nb + d];
            below += k < RS_PF ? pf_rt[k < RS_PF ? k : 0] : row_total[d];
        }
    }
    u64 tot; // one scan carries both: tile-local start of the digit | keys of the whole array with a smaller digit
    const u64 both = block_escan_256<u64>((u64)mine | ((u64)below << 32), s_scan, &tot);
    u32 tstart = (u32)both, dbase = (u32)(both >> 32);
    for (u32 k = 0; k < per; k++) {
        const u32 d = d0 + k;
        if (d >= nb) break;
        const u32 c0 = wcnt[d], c1 = wcnt[nb + d], c2 = wcnt[2 * nb + d], c3 = wcnt[3 * nb + d];
        // first output slot of this tile's keys with digit d
        const u32 gfirst = dbase + (k < RS_PF ? pf_hs[k < RS_PF ? k : 0] : hist_scan[(size_t)tile * nb + d]);
        dbase += k < RS_PF ? pf_rt[k < RS_PF ? k : 0] : row_total[d];
        delta[d] = gfirst - tstart;
        wcnt[d] = (unsigned short)tstart;
        wcnt[nb + d] = (unsigned short)(tstart + c0);
        wcnt[2 * nb + d] = (unsigned short)(tstart + c0 + c1);
        wcnt[3 * nb + d] = (unsigned short)(tstart + c0 + c1 + c2);
        tstart += c0 + c1 + c2 + c3;
    }
    __syncthreads();
    // exchange: keys to their tile-local sorted position
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const u32 i = base + r * 64 + lane;
        const u32 lp = ((rkp[r >> 1] >> ((r & 1) * 16)) & 0xffffu) + wc[(u32)(key[r] >> shift) & mask];
        rkp[r >> 1] = (r & 1) ? ((rkp[r >> 1] & 0xffffu) | (lp << 16)) : ((rkp[r >> 1] & 0xffff0000u) | lp);
        if (i < n) kbuf[lp] = key[r];
    }
    __syncthreads();
    const u32 cnt = min((u32)RS_TILE, n - tile * RS_TILE);
    u32 digp[RS_ITEMS / 2]; // digit of the key in slot j, two per register (for the second write-out)
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const u32 j = r * 256 + threadIdx.x;
        const K kk = j < cnt ? kbuf[j] : (K)0;
        const u32 d = (u32)(kk >> shift) & mask;
        digp[r >> 1] = (r & 1) ? (digp[r >> 1] | (d << 16)) : d;
        if (j < cnt) kout[delta[d] + j] = kk;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const u32 i = base + r * 64 + lane;
        if (i < n) ibuf[(rkp[r >> 1] >> ((r & 1) * 16)) & 0xffffu] = val[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const u32 j = r * 256 + threadIdx.x;
        if (j < cnt) vout[delta[(digp[r >> 1] >> ((r & 1) * 16)) & 0xffffu] + j] = ibuf[j];
    }
}

// ---------------------------------------------------------------------------------------------
// K2s: segment heads.  For sorted position i:  head_j = new intron key; head_r = head_j or read
// position differs from the previous pair of the junction (entropy runs, junction.cc:730-749).
// One u64 scan carries both counters (junction count << 32 | run count).
// ---------------------------------------------------------------------------------------------
struct HeadFn {
    const u64 *skey;
    const u32 *sidx;
    const PairRec *rec;
    __device__ u64 operator()(u64 i) const {
        const u64 im = i ? i - 1 : 0; // (every load unconditional)
        const u64 ka = skey[i], kb = skey[im];
        const u32 sa = sidx[i], sb = sidx[im];
        const int32_t pa = rec[sa].pos, pb = rec[sb].pos;
        const bool hj = i == 0 || ka != kb;
        const bool hr = hj || pa != pb;
        return ((u64)hj << 32) | (u64)hr;
    }
};
struct HeadSink {
    u32 *jid_of;    // [P]   junction id per sorted pair
    u32 *seg_off;   // [J+1] first sorted pair of junction
    u32 *run_first; // [J+1] first run of junction
    u32 *run_start; // [R+1] first sorted pair of run
    const u64 *skey;
    u64 *jkey;      // [J]   intron key of the junction -- when the sort ran on the full keys (nullptr: K2d left the table)
    u32 junc_limit; //       entries jkey has (the other arrays are pair-sized; k2_close stops the chain when there are more heads)
    __device__ void operator()(u64 i, u64 v, u64 ex) const {
        const u32 j = (u32)(ex >> 32) + (u32)(v >> 32) - 1; // inclusive count - 1
        const u32 r = (u32)ex + (u32)v - 1;
        jid_of[i] = j;
        if (v >> 32) {
            seg_off[j] = (u32)i;
            run_first[j] = r;
            if (jkey && j < junc_limit) jkey[j] = skey[i];
        }
        if ((u32)v) run_start[r] = (u32)i;
    }
};
#ifdef PJB_KERNELS_CHAIN
__global__ void k2_close(u64 *total, u32 *seg_off, u32 *run_first, u32 *run_start, ContigStats *cs, u32 junc_limit, const u32 *gen_cnt, u32 gen_cap) {
    const u32 n_pairs = cs->P;
    if (n_pairs == 0) return;
    {
        u32 need = 0;
        for (u32 sh = 0; sh < GEN_SHARDS; sh++) need = max(need, lists_over(gen_cnt, sh, gen_cap));
        if (need) { // a read list overflowed: the chain is repeated with more room
            cs->list_need = need;
            cs->overflow |= OVF_LISTS;
            cs->P = 0;
            return;
        }
    }
    const u32 J = (u32)(*total >> 32), R = (u32)*total;
    cs->n_junc = J;
    cs->n_runs = R;
    if (J > junc_limit) { // the junction-sized buffers are too small: everything downstream stands still
        cs->overflow |= OVF_JUNC;
        cs->P = 0;
        return;
    }
    seg_off[J] = n_pairs;
    run_first[J] = R;
    run_start[R] = n_pairs;
    cs->J = J;
    cs->R = R;
    cs->n_slots = J + (n_pairs + 63) / 64;
}
#endif // PJB_KERNELS_CHAIN

// K2s for the chain on dense ids: k4_pairs left two bits per sorted pair (junction starts / position run starts), a scan over the
// slices' popcounts numbers the runs, and this kernel writes what HeadSink writes -- from the masks and the sorted ids alone.
struct Popc64Fn {
    const u64 *words;
    __device__ u64 operator()(u64 i) const { return (u64)__popcll(words[i]); }
};
constexpr int K2E_PER = 4;
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void k2_expand(const u32 *sid, const u64 *head_mask, const u64 *run_mask, const u32 *run_base, const u64 *total,
                                                 u32 *seg_off, u32 *run_first, u32 *run_start, ContigStats *cs) {
    const u32 n = cs->P;
#pragma unroll
    for (int q = 0; q < K2E_PER; q++) { // (a wavefront takes K2E_PER slices: their mask words' loads travel together)
        const u32 i = (blockIdx.x * K2E_PER + q) * 256 + threadIdx.x;
        if (i >= n) continue;
        const u32 sl = i >> 6, bit = i & 63u;
        const u64 mr = run_mask[sl], mj = head_mask[sl];
        if ((mr >> bit) & 1ull) {
            const u32 r = run_base[sl] + (u32)__popcll(mr & ((1ull << bit) - 1ull));
            run_start[r] = i;
            if ((mj >> bit) & 1ull) {
                const u32 j = sid[i];
                seg_off[j] = i;
                run_first[j] = r;
            }
        }
    }
    if (n > 0 && (n - 1) / (256u * K2E_PER) == blockIdx.x && threadIdx.x == 0) { // (k2_close's part: once, by the block that holds the last pair)
        const u32 R = (u32)*total, J = cs->J;
        seg_off[J] = n;
        run_first[J] = R;
        run_start[R] = n;
        cs->n_runs = cs->R = R;
    }
}
#endif // PJB_KERNELS_CHAIN

// ---------------------------------------------------------------------------------------------
// The chain that sorted the FULL keys (PJB_DENSE_IDS off, raw keys, or a donor with more acceptors than K2d keeps) has its
// junction ids only now: two small kernels give it what kd_assign gives the usual chain -- the rest state of anchors and
// accumulators, the junction id of every pair in BAM order (k4b_generic works in BAM order) and the anchors.
// ---------------------------------------------------------------------------------------------
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void kf_init(u32 *acc, const u32 *n_junc_p, int32_t *anc_l, int32_t *anc_r) {
    const u32 n_junc = *n_junc_p;
    acc_rest_state(acc, n_junc);
    for (u64 t = (u64)blockIdx.x * 256 + threadIdx.x; t < n_junc; t += (u64)gridDim.x * 256) {
        anc_l[t] = INT32_MAX;
        anc_r[t] = INT32_MIN;
    }
}
#endif // PJB_KERNELS_CHAIN
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void kf_anchors(const u32 *sidx, const u32 *jid_of, const PairRec *rec, const u32 *np, u32 *jid_bam, int32_t *anc_l,
                                                   int32_t *anc_r) {
    const u32 n = *np;
    if (blockIdx.x * 256u >= n) return;
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    const u32 ic = valid ? i : n - 1;
    const u32 p = sidx[ic], j = jid_of[ic];
    const uint4 ra = *reinterpret_cast<const uint4 *>(rec + p);
    if (valid) jid_bam[p] = j;
    anchors_fold(valid, j, (int32_t)ra.z, (int32_t)ra.w, anc_l, anc_r);
}
#endif // PJB_KERNELS_CHAIN

// ---------------------------------------------------------------------------------------------
// K4: per-pair match statistics (AlignmentInfo::calcMatchStats junction.cc:147-240 on top of
// BamAlignment::getPaddedQuerySeq / getPaddedGenomeSeq bam_alignment.cc:341-462), computed by
// counting instead of building strings: the query walk and the genome walk are advanced in
// lock-step over the CIGAR and their emissions compared op by op.
// ---------------------------------------------------------------------------------------------
constexpr int GENERIC_NW = 8; // (as the closed form: two 16-byte loads per stream and round, loads may run to the buffers' ends)
struct Side {
    int32_t len, mism, first_mis, last_mis;
    int err;
};

// (k0, r0, q0): the operation the walks start at and the reference / query position there -- (0, position, 0), or, from the
// pair's hint, the first operation that starts inside the window: everything before it the walks only step over
__device__ Side anchor_side(const OpsView cig, u32 n, int32_t position, int32_t aligned, const uint8_t *seq, int32_t lq,
                            const uint8_t *genome, int32_t glen, bool genome_has_x, const u32 *gcodes, int32_t start,
                            int32_t end, u32 k0, int32_t r0, int32_t q0, int32_t q_limit /* last word behind seq that may be read */) {
    Side S;
    S.len = 0;
    S.mism = 0;
    S.first_mis = -1;
    S.last_mis = -1;
    S.err = 0;
    const int32_t get_end = position + aligned - 1;
    if (start > get_end || end < position) { // bam_alignment.cc:342
        S.err = PJB_ERR_NO_PRESENCE;
        return S;
    }
    // getQuerySeqAfterClipping, bam_alignment.cc:256-264 (size_t arithmetic incl. the "+1")
    int32_t dS = 0, dE = 0;
    if ((cig[0] & 15u) == OP_S) dS = (int32_t)(cig[0] >> 4);
    if ((cig[n - 1] & 15u) == OP_S) dE = (int32_t)(cig[n - 1] >> 4);
    if (dS > lq) {
        S.err = PJB_ERR_CLIP_RANGE;
        return S;
    }
    const u64 cnt = (u64)(int64_t)lq - (u64)(int64_t)dS - (u64)(int64_t)dE + 1ull;
    const u64 avail = (u64)(lq - dS);
    const int32_t clen = (int32_t)(cnt < avail ? cnt : avail);
    // ---- pass A: query walk over ops only -> where it stops (actual_start / actual_end)
    int32_t rPos = r0, qPos = q0;
    for (u32 k = k0; k < n; k++) {
        const u32 op = cig[k], ty = op & 15u;
        const int32_t ln = (int32_t)(op >> 4);
        const bool cRef = op_consumes_ref(ty);
        const bool cQry = op_consumes_query(ty) && ty != OP_S;
        if (rPos < start) {
            if (cRef) rPos += ln;
            if (cQry) qPos += ln;
            continue;
        }
        if ((rPos > end && ty != OP_I) || (ty == OP_N && rPos + ln > end)) break; // :359
        if (cQry) {
            const int32_t l = (rPos + ln > end && ty != OP_I) ? end - rPos + 1 : ln;
            if (l == 0) {
                S.err = PJB_ERR_ZERO_LEN_OP;
                return S;
            }
            if (qPos < 0 || qPos + l > clen) {
                S.err = PJB_ERR_QUERY_RANGE;
                return S;
            }
        }
        if (cRef) rPos += ln;
        if (cQry) qPos += ln;
    }
    const int32_t q_start = position > start ? position : start; // :400
    const int32_t q_end = rPos <= end ? rPos - 1 : end;          // :401
    if (q_start - start < 0 || end - q_end < 0) {                // :414-421 (unreachable, kept for parity)
        S.err = PJB_ERR_QREGION;
        return S;
    }
    const int32_t gsize = end - start + 1; // length of the fetched anchor string
    // ---- pass B: both walks in lock-step, comparing emissions
    rPos = r0;
    qPos = q0;
    bool qDone = false, gDone = false, diverged = false;
    int32_t qTot = 0, gTot = 0, mism = 0, first_mis = -1, last_mis = -1;
    for (u32 k = k0; k < n; k++) {
        const u32 op = cig[k], ty = op & 15u;
        const int32_t ln = (int32_t)(op >> 4);
        const bool cRef = op_consumes_ref(ty);
        const bool cQry = op_consumes_query(ty) && ty != OP_S;
        int32_t qEmit = 0, gEmit = 0;
        int qKind = 0, gKind = 0; // 1 = bases, 2 = 'X' padding
        if (!qDone && rPos >= start) {
            if ((rPos > end && ty != OP_I) || (ty == OP_N && rPos + ln > end)) qDone = true;
            else if (cQry) {
                qEmit = (rPos + ln > end && ty != OP_I) ? end - rPos + 1 : ln;
                qKind = 1;
            } else if (cRef) {
                qEmit = rPos + ln > end ? end - rPos + 1 : ln;
                qKind = 2;
            }
        }
        if (!gDone && rPos >= q_start) {
            if (rPos > q_end && ty != OP_I) gDone = true;
            else if (cRef) {
                const int32_t so = rPos - start;
                gEmit = rPos + ln > q_end ? q_end - rPos + 1 : ln;
                if (so < 0 || so + gEmit > gsize) { // :437
                    S.err = PJB_ERR_GENOME_RANGE;
                    return S;
                }
                gKind = 1;
            } else if (cQry) {
                gEmit = ln;
                gKind = 2;
            }
        }
        if (qEmit != gEmit) diverged = true;
        if (!diverged && qEmit > 0) {
            if (qKind == 1 && gKind == 1 && gcodes != nullptr && rPos >= 0 && rPos + qEmit <= glen) {
                cmp_words<GENERIC_NW, true>(reinterpret_cast<const u32 *>(seq), dS + qPos, q_limit, gcodes, rPos, (glen + 7) / 8 + 1, qEmit, qTot,
                                      mism, first_mis, last_mis);
            } else if (qKind == 1 && gKind == 1) {
                const int32_t qb = dS + qPos;
                for (int32_t t = 0; t < qEmit; t++) {
                    const int32_t qi = qb + t;
                    const u32 byte = gload(seq + (qi >> 1));
                    const u32 code = (qi & 1) ? (byte & 15u) : (byte >> 4);
                    const int32_t gi = rPos + t;
                    const u32 gc = (gi >= 0 && gi < glen) ? (u32)gload(genome + gi) : 0u;
                    if (nt16_ascii(code) != gc) {
                        mism++;
                        if (first_mis < 0) first_mis = qTot + t;
                        last_mis = qTot + t;
                    }
                }
            } else if (qKind == 1) { // read letters vs 'X' (insertion): never equal
                mism += qEmit;
                if (first_mis < 0) first_mis = qTot;
                last_mis = qTot + qEmit - 1;
            } else if (!genome_has_x) { // 'X' padding (deletion / enclosed intron) vs genome bases without any 'X'
                mism += qEmit;
                if (first_mis < 0) first_mis = qTot;
                last_mis = qTot + qEmit - 1;
            } else { // contig really contains 'X' characters: compare byte by byte
                for (int32_t t = 0; t < qEmit; t++) {
                    const int32_t gi = rPos + t;
                    const u32 gc = (gi >= 0 && gi < glen) ? (u32)gload(genome + gi) : 0u;
                    if (gc != (u32)'X') {
                        mism++;
                        if (first_mis < 0) first_mis = qTot + t;
                        last_mis = qTot + t;
                    }
                }
            }
        }
        qTot += qEmit;
        gTot += gEmit;
        if (cRef) rPos += ln;
        if (cQry) qPos += ln;
        if (qDone && gDone) break;
    }
    if (qTot != gTot || qTot == 0) { // junction.cc:192,208
        S.err = PJB_ERR_ANCHOR_MISMATCH;
        return S;
    }
    if (diverged) {
        S.err = PJB_ERR_DIVERGENT;
        return S;
    }
    S.len = qTot;
    S.mism = mism;
    S.first_mis = first_mis;
    S.last_mis = last_mis;
    return S;
}

// per-pair match statistics through the generic lock-step walks (any CIGAR)
__device__ __forceinline__ u64 pair_stats_generic(const OpsView cig, u32 nc, int32_t pos, int32_t aligned, const uint8_t *seq,
                                                  int32_t lq, const uint8_t *genome, int32_t glen, bool has_x, const u32 *gcodes,
                                                  int32_t left, int32_t istart, int32_t iend, int32_t right, u32 g, u64 *err, u32 opi,
                                                  int32_t qN, int32_t q_limit) {
    if (lq <= 1) { // junction.cc:168-185
        const u32 totUp = (u32)((istart - 1) - left + 1);
        const u32 totDown = (u32)(right - (iend + 1) + 1);
        return pack_res(0, totUp < totDown ? totUp : totDown, 0);
    }
    // Where the walks start.  The pair's N operation is operation `opi`; it starts at istart, qN query bases into the read.  The
    // left side walks BACK from it to the first operation that starts inside the window -- an operation that starts before the
    // window is stepped over whole by the walks, and so is everything before it --, the right side starts at the operation
    // behind the N.  (A read of 95 operations and 15 introns walked all 95 four times per pair.)
    u32 kL = opi, kR = 0;
    int32_t rL = istart, qL = qN, rR = pos, qR = 0;
    while (kL > 0) {
        const u32 op = cig[kL - 1], ty = op & 15u;
        const int32_t ln = (int32_t)(op >> 4);
        const int32_t rp = rL - (op_consumes_ref(ty) ? ln : 0);
        if (rp < left) break; // (it starts before the window: stepped over, like all before it)
        kL--;
        rL = rp;
        if (op_consumes_query(ty) && ty != OP_S) qL -= ln;
    }
    const int32_t after = istart + (int32_t)(cig[opi] >> 4); // the walks' position behind the N (iend + 1 unless it was clamped)
    if (after == iend + 1) {
        kR = opi + 1;
        rR = after;
        qR = qN;
    }
    const Side L = anchor_side(cig, nc, pos, aligned, seq, lq, genome, glen, has_x, gcodes, left, istart - 1, kL, rL, qL, q_limit);
    if (L.err) {
        set_error(err, g, L.err);
        return 0;
    }
    const Side R = anchor_side(cig, nc, pos, aligned, seq, lq, genome, glen, has_x, gcodes, iend + 1, right, kR, rR, qR, q_limit);
    if (R.err) {
        set_error(err, g, R.err);
        return 0;
    }
    const u32 upM = L.last_mis < 0 ? (u32)L.len : (u32)(L.len - 1 - L.last_mis);  // getNbMatchesFromEnd :272
    const u32 downM = R.first_mis < 0 ? (u32)R.len : (u32)R.first_mis;            // getNbMatchesFromStart :263
    const u32 tu = (u32)(L.len - L.mism), td = (u32)(R.len - R.mism);
    return pack_res(upM < downM ? upM : downM, tu < td ? tu : td, (u32)(L.mism + R.mism));
}

__device__ __forceinline__ const DevBatch &find_batch(const DevBatch *batches, int n_batches, u32 g) {
    int lo = 0, hi = n_batches - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (batches[mid].base <= g) lo = mid;
        else hi = mid - 1;
    }
    return batches[lo];
}

// K4b: every pair that is not of the simple shape (multi-junction reads, indels, = X P H ops, exotic contigs, SEQ '*'), one
// thread per READ of the list k1_emit compacted, in BAM order: the read's operations are fetched once (LDS column) and
// walked once; at every N operation the pair's junction-level anchors (kd_assign) are looked up and the two lock-step
// walks start right there (op index and query offset are at hand: no hint has to travel with the pair).  The result
// goes into the pair's record.  Runs on the side stream, beside the sort.
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void k4b_generic(const u64 *list, const u32 *gen_cnt, u32 cap, const u64 *key, PairRec *rec, const u32 *jid_bam,
                                                    KeyFmt kf, const DevBatch *batches, int n_batches, const int32_t *anc_l, const int32_t *anc_r,
                                                    GroupTab G, int genome_has_x, int use_codes, u64 *err, const ContigStats *cs, u32 pack_nn) {
    __shared__ u32 s_ops[OPS_LDS][256];
    __shared__ u32 s_first[2 * GEN_SHARDS + 1]; // item index of every sub-list's first entry (both lists, one index space)
    __shared__ u32 s_wsum[4];
    if (cs->P == 0) return; // (a limit was exceeded while the junction ids were built: there are no ids, the chain is repeated)
    // The sub-lists are filled to different heights (sub-list `shard` of list `l` occupies [(l GEN_SHARDS + shard) cap, ... + its
    // count)): every block numbers their entries -- an exclusive scan over the 512 counts -- and the grid strides over the items.
    {
        const u32 i0 = threadIdx.x * 2;
        const u32 c0 = gen_cnt[(i0 % GEN_SHARDS) * GEN_CNT_STRIDE + (i0 / GEN_SHARDS) * 2];
        const u32 c1 = gen_cnt[((i0 + 1) % GEN_SHARDS) * GEN_CNT_STRIDE + ((i0 + 1) / GEN_SHARDS) * 2];
        const u32 n0 = c0 < cap ? c0 : cap, n1 = c1 < cap ? c1 : cap;
        u32 total;
        const u32 ex = block_escan<4>(n0 + n1, s_wsum, &total);
        s_first[i0] = ex;
        s_first[i0 + 1] = ex + n0;
        if (threadIdx.x == 255) s_first[2 * GEN_SHARDS] = total;
        __syncthreads();
    }
    static_assert(GEN_SHARDS == 256, "two sub-lists per thread of the block");
    const u32 n_items = s_first[2 * GEN_SHARDS];
    for (u32 item0 = blockIdx.x * 256; item0 < n_items; item0 += gridDim.x * 256) {
    const u32 item = item0 + threadIdx.x;
    if (item >= n_items) continue; // (no barrier below)
    u32 sub = 0; // the last sub-list with s_first[sub] <= item
#pragma unroll
    for (u32 step = GEN_SHARDS; step > 0; step >>= 1)
        if (sub + step < 2 * GEN_SHARDS && s_first[sub + step] <= item) sub += step;
    const bool check_only = sub >= GEN_SHARDS;
    const u64 entry = list[(size_t)sub * cap + (item - s_first[sub])];
    const u32 p0 = (u32)(entry >> 32);
    const u32 g = check_only && pack_nn ? (u32)entry & 0x0fffffffu : (u32)entry;
    if (check_only) {
        // A read [S] M (N M)+ [S] whose pairs k1_emit finished with the M blocks as anchors.  That is what the walks produce
        // unless the junction's window reaches over a neighbouring intron of the read: on the left the walk starts at the
        // first operation that begins inside the window (the previous N begins at its istart), on the right it stops at an N
        // that ends outside it (bam_alignment.cc:359).  A read that fails the test takes the walks below -- all its pairs.
        // N operations of the read: in the entry, or from its first pair's record (no junction of the read ends before that pair, one is its own)
        const u32 code = pack_nn ? (u32)entry >> 28 : 15u;
        const u32 nN = code < 15u ? code : (reinterpret_cast<const uint4 *>(rec + p0)[1].w >> 16) + 1u;
        bool ok = true;
        int32_t prev_istart = 0, is, ie;
        unpack_key(kf, key[p0], is, ie);
        for (u32 k = 0; k < nN; k++) {
            const u32 j = jid_bam[p0 + k];
            int32_t nis = 0, nie = 0;
            if (k + 1 < nN) unpack_key(kf, key[p0 + k + 1], nis, nie);
            if (k > 0 && prev_istart >= anc_l[j]) ok = false;
            if (k + 1 < nN && nie + 1 <= anc_r[j]) ok = false;
            prev_istart = is;
            is = nis;
            ie = nie;
        }
        if (ok) continue;
    }
    const DevBatch &b = find_batch(batches, n_batches, g);
    const u32 r = g - b.base;
    const u32 *cig_off = b.cig_off;
    const u32 c0 = gload(cig_off + r), c1 = gload(cig_off + r + 1);
    const u32 nc = c1 - c0;
    OpsView cig;
    cig.g = b.cigar + c0;
    cig.lds = &s_ops[0][threadIdx.x];
#pragma unroll
    for (int k = 0; k < OPS_LDS; k++) { // (unconditional loads, masked: see k1_count)
        const bool has = (u32)k < nc;
        const u32 v = gload(has ? cig.g + k : cig_off);
        s_ops[k][threadIdx.x] = has ? v : 0u;
    }
    const uint4 rb = reinterpret_cast<const uint4 *>(rec + p0)[1]; // pos | aend | meta | updown of the read's first pair
    const int32_t vpos = (int32_t)rb.x, aend = (int32_t)rb.y;
    const Member M = member_of(G, vpos); // the read's target: everything below is in the target's own coordinates
    const int32_t pos = vpos - M.voff;
    const int32_t lq = gload(b.l_qseq + r);
    const u32 so0 = gload(b.seq_off + r), words = gload(b.seq_off + r + 1) - so0;
    if (lq > 1 && (u64)words * 8ull < (u64)lq) {
        set_error(err, g, PJB_ERR_NO_SEQ);
        continue; // (the records keep aux = 0)
    }
    const uint8_t *seq = b.seq4 + (size_t)so0 * 4;
    const int32_t q_limit = (int32_t)min(gload(b.seq_off + b.n) - 1u - so0, 0x7fffffffu); // (to the end of the batch's bases: chunk_load)
    u32 k = 0;
    int32_t qsum = 0; // query bases before the operation, soft clips not counted (anchor_side's qPos)
    for (u32 i = 0; i < nc; i++) {
        const u32 op = cig[i], ty = op & 15u;
        if (ty == OP_N) {
            const u32 p = p0 + k;
            const u32 j = jid_bam[p];
            int32_t istart, iend;
            unpack_key(kf, key[p], istart, iend);
            const u64 res = pair_stats_generic(cig, nc, pos, aend - vpos + 1, seq, lq, M.d, M.len, genome_has_x != 0, use_codes ? M.codes : (const u32 *)nullptr,
                                               anc_l[j] - M.voff, istart - M.voff, iend - M.voff, anc_r[j] - M.voff, g, err, i, qsum, q_limit);
            *reinterpret_cast<u64 *>(rec + p) = res; // PairRec::aux
            k++;
        }
        if (op_consumes_query(ty) && ty != OP_S) qsum += (int32_t)(op >> 4);
    }
    } // items
}
#endif // PJB_KERNELS_CHAIN

// K4: gather the pairs in sorted order -- one 32-byte record each -- and fold predicates and match statistics to fragment
// heads with a segmented wave reduction (junction.cc:862-909 accumulators, :755-814 counters).
// Fragments.  The sorted pair array is processed in fixed 64-pair slices (one wavefront each).  A fragment is a maximal
// run of one junction inside one slice; its slot is jid + slice index, which is unique and increasing along the array.
// A slot is skipped when a junction starts exactly on a slice boundary, and the very last slot is never used: the
// wavefront that sees either marks the slot unused (frag_j = -1) itself, so nothing has to be initialised.
// A second small kernel reduces slots -> junctions (segmented again, then one atomic per wave and junction), so a
// junction with 10^6 pairs costs ~250 same-address atomics, not 10^6.
// masks (nullptr: the chain that sorted the full keys has its runs from the head scan): per 64-pair slice, the lanes where a
// junction starts and the lanes where a run of equal read positions starts (entropy, junction.cc:730-749) -- this kernel holds
// every pair's record anyway, k2_expand turns the bits into seg_off / run_first / run_start without touching a pair.
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void k4_pairs(const u32 *sidx, const u32 *jid_of, const PairRec *rec, const u64 *jkey, KeyFmt kf, const u32 *np,
                                                 u32 *frag, int32_t *frag_j, u64 *head_mask, u64 *run_mask, const ContigStats *cs_chk, u64 *err_chk) {
    const u32 n = *np;
    if (blockIdx.x * 256u >= n) return;
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    const int lane = lane_id();
    const u32 ic = valid ? i : n - 1; // (loads unconditional, masked after: see k1_count)
    const u32 p = sidx[ic];
    const u32 jv = jid_of[ic];
#ifdef PJB_SELFCHECK // (debug builds: what the sort hands over must be a permutation of the pairs with ids below J that ascend in steps of 0 or 1)
    {
        const u32 Jc = cs_chk->J;
        const u32 jb = ic ? jid_of[ic - 1] : jv;
        int bad = 0;
        if (p >= n) bad = 1;
        else if (jv >= Jc) bad = 2;
        else if (jv != jb && jv != jb + 1) bad = 3;
        else if (ic == 0 && jv != 0) bad = 4;
        else if (ic == n - 1 && jv != Jc - 1) bad = 5;
        else if (jv + (ic >> 6) + 1 >= cs_chk->n_slots) bad = 6;
        else if ((ic >> 6) >= cs_chk->n_slices) bad = 7;
        else if (cs_chk->n_slots != Jc + (n + 63) / 64) bad = 8;
        if (__ballot(bad != 0)) {
            if (bad) set_error(err_chk, 0xffffe000u + (u32)bad, PJB_ERR_HIP);
            return;
        }
    }
#endif
    const PairRec Rc = rec_load(rec + p);
    const u64 jk = jkey[jv];
    // the pair before this one in sorted order: the neighbouring lane's -- lane 0 fetches it
    u32 jprev_v = __shfl_up(jv, 1, 64);
    int32_t pos_prev = __shfl_up(Rc.pos, 1, 64), aend_prev = __shfl_up(Rc.aend, 1, 64);
    if (lane == 0 && valid && i > 0) { // (valid: a wavefront past the end must not follow whatever lies behind the sorted indices)
        jprev_v = jid_of[i - 1];
        const uint4 q = reinterpret_cast<const uint4 *>(rec + sidx[i - 1])[1];
        pos_prev = (int32_t)q.x;
        aend_prev = (int32_t)q.y;
    }
    u32 j = 0xffffffffu;
    u32 cnt[4] = {0, 0, 0, 0}, jadp[5] = {0, 0, 0, 0, 0}, mx[5] = {0, 0, 0, 0, 0};
    u32 first_mis = 100000000u; // junction.cc:864
    u64 mism64 = 0;
    if (valid) {
        j = jv;
        int32_t istart, iend;
        unpack_key(kf, jk, istart, iend);
        const u64 rs = Rc.aux;
        const u32 minMatch = (u32)(rs & 0xfffffu), mmes = (u32)((rs >> 20) & 0xfffffu), nbMis = (u32)(rs >> 40);
        const u32 meta = Rc.meta;
        const u32 cat = meta & META_CAT_MASK;
        const u32 xs = (meta >> META_XS_SHIFT) & 3u;
        // distinct alignment runs in BAM order (junction.cc:763-771): compare with the previous pair of the junction
        const bool dist_head = i == 0 || jprev_v != j || pos_prev != Rc.pos || aend_prev != Rc.aend;
        // 8-bit lanes: a wavefront adds at most 64 per field
        cnt[0] = 1u | ((u32)(cat == 0) << 8) | ((u32)(cat == 1) << 16) | ((u32)(cat == 2) << 24);
        cnt[1] = (u32)(cat == 3) | ((u32)((meta & META_MULTI) != 0) << 8) | ((u32)(xs == 1) << 16) | ((u32)(xs == 2) << 24);
        cnt[2] = (u32)((meta & META_UM) != 0) | ((u32)((meta & META_BPP) != 0) << 8) | ((u32)((meta & META_PPP) != 0) << 16) |
                 ((u32)((meta & META_REL) != 0) << 24);
        cnt[3] = (u32)dist_head;
#pragma unroll
        for (int w = 0; w < 5; w++) // junction.cc:875-877: JAD[k]++ for k < min(20, minMatch)
            jadp[w] = (u32)((u32)(4 * w) < minMatch) | ((u32)((u32)(4 * w + 1) < minMatch) << 8) |
                      ((u32)((u32)(4 * w + 2) < minMatch) << 16) | ((u32)((u32)(4 * w + 3) < minMatch) << 24);
        const int32_t la = istart - Rc.lstart, ra = Rc.rend - iend; // Intron::minAnchorLength intron.cc:81-83
        mx[0] = (u32)(la < ra ? la : ra);
        mx[1] = Rc.updown & 0xffffu;
        mx[2] = Rc.updown >> 16;
        mx[3] = mmes;
        mx[4] = minMatch;
        first_mis = minMatch > 0 ? minMatch : 100000000u;
        mism64 = nbMis;
        // unused fragment slots: the one skipped when a junction starts on a slice boundary, and the last one
        if (lane == 0 && i > 0 && jprev_v != j) frag_j[j + (i >> 6) - 1] = -1;
        if (i == n - 1) frag_j[j + (i >> 6) + 1] = -1;
    }
    if (head_mask) {
        const bool hj = valid && (i == 0 || jprev_v != j);
        const bool hr = valid && (hj || pos_prev != Rc.pos);
        const u64 mj = __ballot(hj), mr = __ballot(hr);
        if (lane == 0 && i < n) { // (a wavefront wholly past the end has no slice: the arrays hold ceil(n / 64) words)
            head_mask[i >> 6] = mj;
            run_mask[i >> 6] = mr;
        }
    }
    auto store_fragment = [&](u32 hi_word) {
        const u32 slot = j + (i >> 6);
        u32 vals[F_WORDS];
#pragma unroll
        for (int k = 0; k < F_WORDS; k++) vals[k] = 0;
        // a full wavefront of one junction makes the first field 64: it still fits its byte (max 64 < 256)
#pragma unroll
        for (int k = 0; k < 13; k++) vals[F_N + k] = (cnt[k >> 2] >> (8 * (k & 3))) & 0xffu;
        vals[F_MAXMINANC] = mx[0];
        vals[F_UP] = mx[1];
        vals[F_DOWN] = mx[2];
        vals[F_MAXMMES] = mx[3];
        vals[F_MAXMINMATCH] = mx[4];
        vals[F_FIRSTMIS] = first_mis;
        vals[F_MISM_LO] = (u32)mism64;
        vals[F_MISM_HI] = hi_word;
#pragma unroll
        for (int k = 0; k < 20; k++) vals[F_JAD0 + k] = (jadp[k >> 2] >> (8 * (k & 3))) & 0xffu;
        uint4 *dst = reinterpret_cast<uint4 *>(frag + (size_t)slot * F_WORDS);
#pragma unroll
        for (int k = 0; k < F_WORDS / 4; k++) dst[k] = make_uint4(vals[4 * k], vals[4 * k + 1], vals[4 * k + 2], vals[4 * k + 3]);
        frag_j[slot] = (int32_t)j;
    };
    // ---- a wavefront that holds pairs of ONE junction (the usual case: a junction has a few hundred pairs): plain
    // whole-wave reductions on the DPP path, 16 x 6 VALU steps instead of 115 ds_bpermute round trips
    const u32 j0 = (u32)__builtin_amdgcn_readfirstlane((int)j);
    if (__ballot(valid && j != j0) == 0) {
#pragma unroll
        for (int w = 0; w < 4; w++) cnt[w] = wave_total<DppAdd>(cnt[w]);
#pragma unroll
        for (int w = 0; w < 5; w++) jadp[w] = wave_total<DppAdd>(jadp[w]);
#pragma unroll
        for (int w = 0; w < 5; w++) mx[w] = wave_total<DppMax>(mx[w]);
        first_mis = wave_total<DppMin>(first_mis);
        // (a pair has fewer than 2^24 mismatches: 64 of them fit 32 bits)
        mism64 = (u64)wave_total<DppAdd>((u32)mism64);
        if (valid && lane == 0) store_fragment(0u);
        return;
    }
    // ---- several junctions in the wavefront: segmented wave reduce to fragment heads; the "same junction at distance
    // o" tests are done once
    u32 take = 0;
#pragma unroll
    for (int sft = 0; sft < 6; sft++) {
        const int o = 1 << sft;
        const u32 kk = __shfl_down(j, o, 64);
        if (lane + o < 64 && kk == j) take |= 1u << sft;
    }
    auto red_add = [&](u32 v) {
#pragma unroll
        for (int sft = 0; sft < 6; sft++) {
            const u32 t = __shfl_down(v, 1 << sft, 64);
            if ((take >> sft) & 1u) v += t;
        }
        return v;
    };
    auto red_max = [&](u32 v) {
#pragma unroll
        for (int sft = 0; sft < 6; sft++) {
            const u32 t = __shfl_down(v, 1 << sft, 64);
            if (((take >> sft) & 1u) && t > v) v = t;
        }
        return v;
    };
#pragma unroll
    for (int w = 0; w < 4; w++) cnt[w] = red_add(cnt[w]);
#pragma unroll
    for (int w = 0; w < 5; w++) jadp[w] = red_add(jadp[w]);
#pragma unroll
    for (int w = 0; w < 5; w++) mx[w] = red_max(mx[w]);
#pragma unroll
    for (int sft = 0; sft < 6; sft++) {
        const u32 t = __shfl_down(first_mis, 1 << sft, 64);
        const u64 t64 = __shfl_down(mism64, 1 << sft, 64);
        if ((take >> sft) & 1u) {
            first_mis = t < first_mis ? t : first_mis;
            mism64 += t64;
        }
    }
    const u32 jprev = __shfl_up(j, 1, 64);
    if (valid && (lane == 0 || jprev != j)) store_fragment((u32)(mism64 >> 32));
}
#endif // PJB_KERNELS_CHAIN

// K5a: fragment slots -> junction accumulators (acc pre-initialised: sums 0, max 0, min 100000000).
// One wavefront walks 64 consecutive slots; lane k owns word k of the 48-word record, so every
// load is one coalesced 192-byte row and a junction's run of slots is folded in registers; the run
// is flushed with one atomic per word when the junction changes (same-address chain <= P_j/4096).
__device__ __forceinline__ u32 frag_combine(int k, u32 a, u32 b) {
    if (k >= F_MAXMINANC && k <= F_MAXMINMATCH) return a > b ? a : b;
    if (k == F_FIRSTMIS) return a < b ? a : b;
    return a + b; // sums; F_MISM_LO/HI are handled as one 64-bit add by the caller
}
constexpr int FRAG_SLOTS_PER_WAVE = 16; // short per-wave chains keep enough wavefronts in flight
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void k5_frag_reduce(const u32 *frag, const int32_t *frag_j, const u32 *n_slots_p, u32 *acc) {
    const u32 n_slots = *n_slots_p;
    const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int k = lane_id();
    const u32 s0 = wave * FRAG_SLOTS_PER_WAVE;
    if (s0 >= n_slots) return;
    const int32_t myj = (k < FRAG_SLOTS_PER_WAVE && s0 + k < n_slots) ? frag_j[s0 + k] : -1;
    int32_t cur = -1;
    u32 v = 0, carry_lo = 0; // lane F_MISM_HI also keeps the low word to do the 64-bit add
    auto flush = [&](int32_t j) {
        if (j < 0 || k >= F_JAD0 + 20) return;
        u32 *dst = acc + (size_t)j * F_WORDS + k;
        if (k == F_MISM_LO) return; // done by lane F_MISM_HI as one 64-bit atomic
        if (k == F_MISM_HI) {
            atomicAdd(reinterpret_cast<u64 *>(dst - 1), ((u64)v << 32) | carry_lo);
        } else if (k >= F_MAXMINANC && k <= F_MAXMINMATCH) atomicMax(dst, v);
        else if (k == F_FIRSTMIS) atomicMin(dst, v);
        else if (v) atomicAdd(dst, v);
    };
    // (the rows of all sixteen slots are asked for before the first is folded: sixteen loads on their way at once, not a chain of
    // sixteen round trips)
    u32 row[FRAG_SLOTS_PER_WAVE];
#pragma unroll
    for (int q = 0; q < FRAG_SLOTS_PER_WAVE; q++) {
        const int32_t jq = __shfl(myj, q, 64);
        row[q] = (jq >= 0 && k < F_WORDS) ? frag[(size_t)(s0 + (u32)q) * F_WORDS + k] : 0u;
    }
#pragma unroll
    for (int q = 0; q < FRAG_SLOTS_PER_WAVE; q++) { // (no break / continue: the loop unrolls and row[] stays in registers)
        const int32_t j = __shfl(myj, q, 64);
        if (j >= 0) { // (unused slots and those past the last have -1)
            const u32 w = row[q];
            const u32 wlo = __shfl(w, F_MISM_LO, 64);
            if (j != cur) {
                flush(cur);
                cur = j;
                v = w;
                carry_lo = wlo;
            } else if (k == F_MISM_HI) {
                const u64 sum = (((u64)v << 32) | carry_lo) + (((u64)w << 32) | wlo);
                v = (u32)(sum >> 32);
                carry_lo = (u32)sum;
            } else {
                v = frag_combine(k, v, w);
            }
        }
    }
    flush(cur);
}
#endif // PJB_KERNELS_CHAIN

// ---------------------------------------------------------------------------------------------
// K5b: one thread per junction: strand from reads, entropy over position runs, splice motif,
// hamming scores, suspicious flag -> output row.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint8_t revcomp_char(uint8_t c) { // REVCOMP_LOOKUP seq_utils.hpp:33-40 (NUL outside A-Z)
    switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'D': return 'H';
    case 'G': return 'C';
    case 'H': return 'D';
    case 'M': return 'K';
    case 'N': return 'N';
    case 'R': return 'Y';
    case 'S': return 'W';
    case 'T': return 'A';
    case 'U': return 'A';
    case 'V': return 'B';
    case 'W': return 'S';
    case 'X': return 'X';
    case 'Y': return 'R';
    default: return 0;
    }
}

// faidx_fetch_seq clamping (deps/htslib-1.3/faidx.c:453-457): returns clamped [b,e]
__device__ __forceinline__ void fetch_clamp(int32_t glen, int32_t &b, int32_t &e) {
    if (e < b) b = e;
    if (b < 0) b = 0;
    else if (glen <= b) b = glen - 1;
    if (e < 0) e = 0;
    else if (glen <= e) e = glen - 1;
}

// Entropy terms, one thread per position run: p*log2(p) with the reference's grouping
// (junction.cc:730-749): with runs r_0..r_m of equal read position the flush rule yields the counts
// (r_0+1, r_1, ..., r_{m-1}, r_m-1); a zero count contributes nothing.  k5_finalize adds the terms
// of a junction sequentially, in run order, so the sum rounds like the reference's loop.
// Entropy of a junction (junction.cc:730-749) = |sum over its position runs of p log2 p|, p = run length / pairs of the junction
// with the reference's grouping (first run + 1, last run - 1 when there are several), added one after the other in run
// order so that the sum rounds like the reference's loop (:742-748).  Sixteen lanes per junction: 16 runs at a time -- each
// lane its run's term from two neighbouring run starts -- folded in lane order (the chain of dependent adds is the
// reference's, there is no chain of dependent loads).  (The terms had a kernel and an array of their own until
// round 4.)
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void k5_entropy_sum(const u32 *seg_off, const u32 *run_first, const u32 *run_start, const u32 *n_junc_p,
                                                       double *ent_sum) {
    // sixteen lanes per junction, four junctions per wavefront: most junctions have a handful of runs (a whole wavefront each spent
    // its time on 60 idle lanes), the deep ones a hundred or two (read positions within a read length of the intron)
    const int lane = lane_id(), sl = lane & 15;
    const u32 j = ((blockIdx.x * 4 + (threadIdx.x >> 6)) << 2) + (u32)(lane >> 4);
    const bool valid = j < *n_junc_p;
    const u32 rf = valid ? run_first[j] : 0u, rl = valid ? run_first[j + 1] : 0u;
    const u32 n = valid ? seg_off[j + 1] - seg_off[j] : 0u;
    double sum = 0.0;
    for (u32 r0 = rf; __any(r0 < rl); r0 += 16) {
        const u32 r = r0 + (u32)sl;
        double t = 0.0;
        if (r < rl) {
            u32 c = run_start[r + 1] - run_start[r];
            if (rl - rf > 1) {
                if (r == rf) c += 1;
                else if (r == rl - 1) c -= 1;
            }
            if (c != 0 && n > 1) {
                const double pI = (double)c / (double)n;
                t = __dmul_rn(pI, log2(pI));
            }
        }
        const u32 cnt = r0 < rl ? (rl - r0 < 16u ? rl - r0 : 16u) : 0u;
#pragma unroll
        for (u32 k = 0; k < 16; k++) { // (in run order: the sum rounds like the reference's loop)
            const double v = __shfl(t, (int)k, 16);
            if (k < cnt) sum = __dadd_rn(sum, v);
        }
    }
    if (valid && sl == 0) ent_sum[j] = sum;
}
#endif // PJB_KERNELS_CHAIN

#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void k5_finalize(const u64 *jkey, const u32 *seg_off, const u32 *run_first,
                                                    const u32 *run_start, const u32 *acc, const int32_t *anc_l,
                                                    const int32_t *anc_r, KeyFmt kf, GroupTab G, const u32 *n_junc_p, const double *ent_sum,
                                                    pjb_junction_row *rows, u64 *err, u32 *member_junc) {
    const u32 n_junc = *n_junc_p;
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_junc) return;
    const u32 *a = acc + (size_t)j * F_WORDS;
    pjb_junction_row R;
    memset(&R, 0, sizeof R);
    int32_t istart, iend;
    unpack_key(kf, jkey[j], istart, iend);
    // the junction's target: rows carry the target's own coordinates
    const Member M = member_of(G, istart);
    const uint8_t *genome = M.d;
    const int32_t glen = M.len;
    istart -= M.voff;
    iend -= M.voff;
    if (G.n > 1) { // junctions per member: neighbouring junctions share their member, so one add per wavefront and member
        const int m0 = __builtin_amdgcn_readfirstlane(M.idx);
        const u64 same = __ballot(M.idx == m0);
        if (M.idx == m0) {
            if (lane_id() == __ffsll((long long)same) - 1) atomicAdd(&member_junc[m0], (u32)__popcll(same));
        } else
            atomicAdd(&member_junc[M.idx], 1u);
    }
    R.refid = M.tid;
    R.start = istart;
    R.end = iend;
    R.left = anc_l[j] - M.voff;
    R.right = anc_r[j] - M.voff;
    const u32 n = a[F_N];
    R.nb_raw = n;
    R.nb_dist = a[F_DIST];
    R.nb_ms = a[F_MS];
    R.nb_um = a[F_UM];
    R.nb_bpp = a[F_BPP];
    R.nb_ppp = a[F_PPP];
    R.nb_rel = a[F_REL];
    R.r1pos = a[F_R1P];
    R.r1neg = a[F_R1N];
    R.r2pos = a[F_R2P];
    R.r2neg = a[F_R2N];
    R.max_min_anc = a[F_MAXMINANC];
    R.maxmmes = a[F_MAXMMES];
    R.nb_up_juncs = a[F_UP];
    R.nb_down_juncs = a[F_DOWN];
    const u64 mism = (u64)a[F_MISM_LO] | ((u64)a[F_MISM_HI] << 32);
    R.sum_mismatches = mism;
    for (int k = 0; k < 20; k++) R.jad[k] = a[F_JAD0 + k];
    // suspicious, junction.cc:897-908.  (nbMismatches is a uint32 in the reference; the test is > 0)
    {
        const u32 first_mis = a[F_FIRSTMIS];
        if ((u32)mism > 0 && first_mis < 20 && !(a[F_MAXMINMATCH] > first_mis)) R.suspicious = 1;
    }
    // determineStrandFromReads, junction.cc:531-559
    {
        const u32 np = a[F_XSP], nn = a[F_XSN];
        const double tot = (double)n;
        if ((double)np / tot >= 0.95) R.read_strand = PJB_STRAND_POS;
        else if ((double)nn / tot >= 0.95) R.read_strand = PJB_STRAND_NEG;
        else R.read_strand = PJB_STRAND_UNK;
    }
    // calcEntropy, junction.cc:730-749: the per-run terms summed in run order (k5_entropy_sum), then fabs
    R.entropy = n > 1 ? fabs(ent_sum[j]) : 0.0;
    // processJunctionWindow, junction.cc:561-649
    {
        int32_t b = istart, e = istart + 1;
        fetch_clamp(glen, b, e);
        int32_t b2 = iend - 1, e2 = iend;
        fetch_clamp(glen, b2, e2);
        if (e - b + 1 != 2 || e2 - b2 + 1 != 2 || glen <= 0) {
            set_error(err, 0xffffff00u + 0, PJB_ERR_SPLICE_SITE_LEN);
            rows[j] = R;
            return;
        }
        uint8_t d0 = genome[b], d1 = genome[b + 1], a0 = genome[b2], a1 = genome[b2 + 1];
        const u32 m4 = ((u32)d0 << 24) | ((u32)d1 << 16) | ((u32)a0 << 8) | (u32)a1;
        const u32 GTAG = 0x47544147u, CTAC = 0x43544143u, ATAC = 0x41544143u, GTAT = 0x47544154u, GCAG = 0x47434147u,
                  CTGC = 0x43544743u;
        int css, ss;
        if (m4 == GTAG || m4 == CTAC) css = PJB_CSS_CANONICAL;
        else if (m4 == ATAC || m4 == GTAT || m4 == GCAG || m4 == CTGC) css = PJB_CSS_SEMI;
        else css = PJB_CSS_NO;
        if (m4 == GTAG || m4 == ATAC || m4 == GCAG) ss = PJB_STRAND_POS;
        else if (m4 == CTAC || m4 == GTAT || m4 == CTGC) ss = PJB_STRAND_NEG;
        else ss = PJB_STRAND_UNK;
        const int rs = R.read_strand;
        const int cons = rs == ss ? rs : rs == PJB_STRAND_UNK ? ss : ss == PJB_STRAND_UNK ? rs : PJB_STRAND_UNK;
        R.canonical = (uint8_t)css;
        R.ss_strand = (uint8_t)ss;
        R.cons_strand = (uint8_t)cons;
        if (cons == PJB_STRAND_NEG) {
            R.da1[0] = revcomp_char(a1);
            R.da1[1] = revcomp_char(a0);
            R.da2[0] = revcomp_char(d1);
            R.da2[1] = revcomp_char(d0);
        } else {
            R.da1[0] = d0;
            R.da1[1] = d1;
            R.da2[0] = a0;
            R.da2[1] = a1;
        }
        // anchors / intron flanks
        int32_t lb = R.left, le = istart - 1;
        fetch_clamp(glen, lb, le);
        int32_t rb = iend + 1, re = R.right;
        fetch_clamp(glen, rb, re);
        int32_t lib = istart, lie = istart + 9;
        fetch_clamp(glen, lib, lie);
        int32_t rib = iend - 9, rie = iend;
        fetch_clamp(glen, rib, rie);
        const int32_t leftAncLen = le - lb + 1, rightAncLen = re - rb + 1;
        const int32_t expL = istart - R.left, expR = R.right - iend;
        if ((leftAncLen != expL && expL > 0) || (rightAncLen != expR && expR > 0)) {
            set_error(err, 0xffffff00u + 1, PJB_ERR_ANCHOR_LEN);
            rows[j] = R;
            return;
        }
        if (lie - lib + 1 != 10 || rie - rib + 1 != 10) {
            set_error(err, 0xffffff00u + 2, PJB_ERR_INTRON_FLANK_LEN);
            rows[j] = R;
            return;
        }
        // calcHammingScores, junction.cc:823-857
        const int32_t nla = leftAncLen < 10 ? leftAncLen : 10;  // last <=10 of left anchor
        const int32_t la_b = leftAncLen < 10 ? lb : le - 9;
        const int32_t nra = rightAncLen < 10 ? rightAncLen : 10; // first <=10 of right anchor
        const int32_t ra_b = rb;
        const int32_t nli = 10, nri = 10;
        const int32_t leftDelta = nla - nri;
        const int32_t leftOffset = leftDelta <= 0 ? 0 : leftDelta;
        const int32_t leftLen = nla < nri ? nla : nri;
        const int32_t rightLen = nli < nra ? nli : nra;
        // la' = nla > leftLen ? la.substr(leftOffset,leftLen) : la   (nla <= 10 = nri so never longer)
        const int32_t zla = nla, la_o = 0;
        const int32_t zli = nli > rightLen ? rightLen : nli, li_o = 0;
        const int32_t zri = nri > leftLen ? leftLen : nri, ri_o = nri > leftLen ? leftOffset : 0;
        const int32_t zra = nra > rightLen ? rightLen : nra, ra_o = 0;
        (void)la_o; (void)li_o; (void)ra_o;
        u32 h5 = 0, h3 = 0;
        if (zla != zri || zra != zli) {
            set_error(err, 0xffffff00u + 3, PJB_ERR_HAMMING_LEN);
            rows[j] = R;
            return;
        }
        if (cons == PJB_STRAND_NEG) {
            // anchor5p = rc(ra), intron3p = rc(li): H over reversed+complemented strings
            for (int32_t t = 0; t < zra; t++) {
                const uint8_t x = revcomp_char(genome[ra_b + (zra - 1 - t)]);
                const uint8_t y = revcomp_char(genome[lib + (zli - 1 - t)]);
                h5 += x != y;
            }
            for (int32_t t = 0; t < zla; t++) {
                const uint8_t x = revcomp_char(genome[la_b + (zla - 1 - t)]);
                const uint8_t y = revcomp_char(genome[rib + ri_o + (zri - 1 - t)]);
                h3 += x != y;
            }
        } else {
            for (int32_t t = 0; t < zla; t++) h5 += genome[la_b + t] != genome[rib + ri_o + t];
            for (int32_t t = 0; t < zra; t++) h3 += genome[ra_b + t] != genome[lib + t];
        }
        R.hamming5p = h5;
        R.hamming3p = h3;
    }
    rows[j] = R;
}
#endif // PJB_KERNELS_CHAIN

// K6: the contig's rows leave the device inside the kernel chain -- the host does not know the row count when it
// queues the work, so it cannot size a copy: 8-byte units go straight into page-locked host memory (mapped into the
// device's address space; consecutive lanes write consecutive addresses, PCIe posted writes) and, if the caller set a
// row mirror, into its exchange slot; the control block follows as the last thing on the stream.
constexpr int ROW_U64 = (int)(sizeof(pjb_junction_row) / 8);
static_assert(sizeof(pjb_junction_row) % 8 == 0, "rows are copied in 8-byte units");
// Where a contig's rows go is known to the host only when the contigs before it have been collected.  With two contigs
// queued (pjb_finish_contig_begin) the second one's place follows from the first one's junction count, which lives on
// the device: a cursor (rows written so far: host table, exchange slot), read by k6_rows_out and advanced by
// publish_chain (the last block of k6_rows_out), on the rows stream and so in contig order.  base < 0: take the cursor.
struct RowCursor {
    u32 rows, mirror_rows;
    u32 blocks_done; // k6_rows_out: blocks of the running launch that have written their rows (0 at rest)
    u32 _pad;
};
__device__ __forceinline__ void publish_chain(const ContigStats *cs, u64 *err, u32 *gen_cnt, uint8_t *host, int64_t base, int64_t mirror_base,
                                              RowCursor *cur, const MemberStats *members, u32 *member_junc, int n_members);
// Rows -> the row table, then -- the block that finishes last -- the chain's control block -> page-locked host memory, rest
// states restored, the cursor advanced (publish_chain; a launch of its own until round 4).  The rows stream runs these
// launches one after the other, so one counter in the cursor serves every slot.
#ifdef PJB_KERNELS_CHAIN
__global__ __launch_bounds__(256) void k6_rows_out(const u64 *rows, const ContigStats *cs, u64 *host_table, int64_t base, int64_t mirror_base,
                                                   RowCursor *cur, u64 *mirror_table, u32 mirror_room, u64 *err, u32 *gen_cnt, uint8_t *host_pub,
                                                   const MemberStats *members, u32 *member_junc, int n_members) {
    // a grid of a few dozen blocks walks the rows: the kernel runs beside the next contig's first kernels, its stores
    // wait on PCIe, and it should not sit on their wave slots meanwhile
    const u32 nj = cs->J;
    const u64 n = (u64)nj * ROW_U64;
    const u64 at = base >= 0 ? (u64)base : (u64)cur->rows;
    const u64 mat = mirror_base >= 0 ? (u64)mirror_base : (u64)cur->mirror_rows;
    const bool to_mirror = mirror_table && mat + nj <= (u64)mirror_room; // (the slot is left alone by a contig that does not fit: the host reports it)
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const u64 v = rows[i];
        host_table[at * ROW_U64 + i] = v;
        if (to_mirror) mirror_table[mat * ROW_U64 + i] = v;
    }
    __shared__ u32 s_last;
    // (The rows -- host memory, and a mirror that may be a peer's -- must be out before the control block that announces them: a
    // system-scope fence in every block before it counts itself done, and once more in the last block before it publishes.  Today
    // every consumer waits for the stream's event; a consumer that polls the control block would see the rows too.  The one
    // blocks_done counter serves every slot because the rows stream runs these launches one after the other.)
    __threadfence_system();
    __syncthreads(); // (every thread of the block has read the cursor)
    if (threadIdx.x == 0) s_last = atomicAdd(&cur->blocks_done, 1u) + 1u == gridDim.x ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence_system();
    if (threadIdx.x == 0) cur->blocks_done = 0;
    publish_chain(cs, err, gen_cnt, host_pub, base, mirror_base, cur, members, member_junc, n_members);
}
#endif // PJB_KERNELS_CHAIN

// The last kernel of a contig: control block, error word and list counters go to page-locked host memory in one go
// (three small copies otherwise), error word and counters return to their rest state for the contig that uses this
// control slot next, and the row cursor moves on.
constexpr int PUB_BASE_AT = 240, PUB_ERR_AT = 256, PUB_XCNT_AT = 320 /* --extra: the target's counters, 64 bytes */, PUB_CHECKED_AT = 384 /* reads on k4b_generic's second list */, PUB_GEN_AT = 512, PUB_MEMBERS_AT = 1536, PUB_GREADS_AT = 3072, PUB_BYTES = 4096; // byte offsets in the published block
static_assert(PUB_MEMBERS_AT + GROUP_MAX * sizeof(MemberStats) <= PUB_BYTES && sizeof(MemberStats) % 8 == 0, "control block layout");
__device__ __forceinline__ void publish_chain(const ContigStats *cs, u64 *err, u32 *gen_cnt, uint8_t *host, int64_t base, int64_t mirror_base,
                                              RowCursor *cur, const MemberStats *members, u32 *member_junc, int n_members) {
    const u32 t = threadIdx.x;
    if (n_members > 1 && t < (u32)n_members) { // a group: the members' own counters
        MemberStats S = members[t];
        S.n_junc = member_junc[t];
        reinterpret_cast<MemberStats *>(host + PUB_MEMBERS_AT)[t] = S;
    }
    static_assert(sizeof(ContigStats) % 8 == 0 && sizeof(ContigStats) <= PUB_BASE_AT, "control block layout");
    static_assert(PUB_GEN_AT + GEN_SHARDS * 4 <= PUB_MEMBERS_AT && PUB_MEMBERS_AT + GROUP_MAX * sizeof(MemberStats) <= PUB_GREADS_AT &&
                      PUB_GREADS_AT + GEN_SHARDS * 4 <= PUB_BYTES,
                  "control block layout");
    if (t < sizeof(ContigStats) / 8) reinterpret_cast<u64 *>(host)[t] = reinterpret_cast<const u64 *>(cs)[t];
    __shared__ u32 s_chk[4];
    {
        const u32 chk = wave_total<DppAdd>(t < GEN_SHARDS ? gen_cnt[t * GEN_CNT_STRIDE + 2] : 0u);
        if (lane_id() == 0) s_chk[t >> 6] = chk;
    }
    if (t < GEN_SHARDS) { // pairs that took the generic walks, per sub-list; both counters back to their rest state
        reinterpret_cast<u32 *>(host + PUB_GEN_AT)[t] = gen_cnt[t * GEN_CNT_STRIDE + 1];
        reinterpret_cast<u32 *>(host + PUB_GREADS_AT)[t] = gen_cnt[t * GEN_CNT_STRIDE];
        reinterpret_cast<uint4 *>(gen_cnt + t * GEN_CNT_STRIDE)[0] = make_uint4(0, 0, 0, 0);
        reinterpret_cast<uint4 *>(gen_cnt + t * GEN_CNT_STRIDE)[1] = make_uint4(0, 0, 0, 0); // (the third list's)
    }
    if (t == 0) {
        *reinterpret_cast<u64 *>(host + PUB_ERR_AT) = *err;
        *err = ~0ull;
        const u32 at = base >= 0 ? (u32)base : cur->rows, mat = mirror_base >= 0 ? (u32)mirror_base : cur->mirror_rows;
        reinterpret_cast<u32 *>(host + PUB_BASE_AT)[0] = at;
        reinterpret_cast<u32 *>(host + PUB_BASE_AT)[1] = mat;
        cur->rows = at + cs->J;
        cur->mirror_rows = mat + cs->J;
    }
    __syncthreads();
    if (t < (u32)GROUP_MAX) member_junc[t] = 0; // (k5_finalize counts into it; rest state for the chain that uses the slot next)
    if (t == 0) *reinterpret_cast<u32 *>(host + PUB_CHECKED_AT) = s_chk[0] + s_chk[1] + s_chk[2] + s_chk[3];
}

} // namespace pjb
