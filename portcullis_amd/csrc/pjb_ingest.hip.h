// Device-side ingest of prepared BAM files: BGZF inflate and BAM record parsing on the GPU.
//
// bgzf_inflate replaces, for the junc path, htslib's bgzf_read_block / inflate_block
// (deps/htslib-1.3/bgzf.c:292-316, 421-540): every BGZF block is an independent raw-DEFLATE stream of
// at most 64 KB of output, so a file is tens of thousands of independent decodes.  One LANE owns one
// BGZF block (a 64-lane workgroup = 64 blocks): DEFLATE is serial inside a stream, the parallelism
// is across streams.  Like inflate_block, the CRC32 of the footer is not checked; unlike it, the
// inflated size must equal the footer's ISIZE (it fixes where the block lands in the output).
// All per-lane state lives in LDS, laid out [entry][lane] so the 64 lanes of a wave hit 64 different banks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pjb {

typedef unsigned long long iu64;
typedef uint32_t iu32;

constexpr int INF_PAD = 4096;        // zeroed bytes the compressed buffer carries after its last block
constexpr int INF_LENS = 320;        // code lengths being read (288 + 32)
constexpr size_t INF_SCRATCH_PER_LANE = INF_LENS; // global scratch per lane: the code lengths of the block header being read

enum : int { // per-block status
    INF_OK = 0,
    INF_ERR_HEADER = 1,    // not a BGZF block header
    INF_ERR_BTYPE = 2,     // reserved block type
    INF_ERR_STORED = 3,    // LEN / NLEN mismatch
    INF_ERR_CODELENS = 4,  // bad code length set (over-subscribed / bad repeat / too many symbols)
    INF_ERR_CODE = 5,      // invalid Huffman code in the data
    INF_ERR_DIST = 6,      // distance reaches before the start of the block
    INF_ERR_OVERRUN = 7,   // more output than ISIZE / more input than the block holds
    INF_ERR_SIZE = 8,      // stream ended before ISIZE bytes
};

struct InfBlock { // one BGZF block, filled by the host while it hops over the block headers
    iu64 in_off;  // first byte of the DEFLATE payload in the compressed buffer
    iu64 out_off; // where the block's bytes go in the inflated buffer
    iu32 in_len;  // payload bytes (BSIZE + 1 - XLEN - 20)
    iu32 out_len; // ISIZE
};

__constant__ unsigned short c_len_base[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                              31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ unsigned char c_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ unsigned short c_dist_base[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
                                               193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ unsigned char c_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ unsigned char c_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ iu64 load64u(const uint8_t *p) {
    iu64 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ void store64u(uint8_t *p, iu64 v) { __builtin_memcpy(p, &v, 8); }
__device__ __forceinline__ iu32 load32u(const uint8_t *p) {
    iu32 v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
// the low n (0..8) bytes of v in at most three stores (4 + 2 + 1) instead of n byte stores
__device__ __forceinline__ void store_low(uint8_t *p, iu64 v, iu32 n) {
    if (n >= 8) {
        store64u(p, v);
        return;
    }
    if (n & 4u) {
        const iu32 w = (iu32)v;
        __builtin_memcpy(p, &w, 4);
        p += 4;
        v >>= 32;
    }
    if (n & 2u) {
        const unsigned short h = (unsigned short)v;
        __builtin_memcpy(p, &h, 2);
        p += 2;
        v >>= 16;
    }
    if (n & 1u) *p = (uint8_t)v;
}

// LSB-first bit reader.  Input is fetched 8 bytes at a time, two words ahead of the bit buffer, so
// the load a refill depends on was issued at least 64 input bits earlier.
struct BitReader {
    const uint8_t *ip; // next byte to fetch into the reservoir
    iu64 bb;           // bit buffer
    int nb;            // valid bits in bb
    iu64 r0, r1;       // reservoir: r0 is consumed 32 bits at a time, r1 is the word after it
    int r0w;           // 32-bit halves left in r0 (2, 1)
    __device__ __forceinline__ void start(const uint8_t *p) {
        r0 = load64u(p);
        r1 = load64u(p + 8);
        ip = p + 16;
        r0w = 2;
        bb = 0;
        nb = 0;
    }
    __device__ __forceinline__ void refill() { // at least 33 valid bits afterwards
        if (nb <= 32) {
            bb |= (r0 & 0xffffffffull) << nb;
            nb += 32;
            r0 >>= 32;
            if (--r0w == 0) {
                r0 = r1;
                r0w = 2;
                r1 = load64u(ip);
                ip += 8;
            }
        }
    }
    // address of the next input byte not yet moved into the bit buffer's whole bytes
    __device__ __forceinline__ const uint8_t *byte_pos() const { return ip - 8 - 4 * r0w - (nb >> 3); }
    __device__ __forceinline__ iu32 peek(int n) const { return (iu32)bb & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(int n) {
        bb >>= n;
        nb -= n;
    }
    __device__ __forceinline__ iu32 take(int n) {
        const iu32 v = peek(n);
        drop(n);
        return v;
    }
};

__device__ __forceinline__ iu32 bitrev16(iu32 v, int len) { return __brev(v) >> (32 - len); }

// =================================================================================================
// bgzf_inflate: the memory accesses of the 64 lanes are lined up.
//
// A first version decoded straight from and to global memory; some lane then needs memory in practically
// every iteration (a fifth of all symbols are matches further back than any register window, plus input
// refills and second-level table lookups) and a wave waits as a whole: one memory round trip per
// iteration, 2 us each, 20 GB/s.  Here a lane never touches global memory while it decodes:
//   * the Huffman tables are complete in LDS: 8-bit / 5-bit first levels, longer codes by canonical
//     arithmetic (upper bound per code length, symbols in code order) without branches;
//   * input comes from a 64-byte LDS ring per lane, matches go to a 16-entry LDS queue per lane
//     (decoding does not depend on what a match copies); literals are plain byte stores;
//   * when any lane runs low on input or has a full queue, ALL lanes take a memory phase together:
//     one round trip tops up the rings and fetches the first 16 bytes of every queued match whose source
//     lies before the queue's first destination; the few remaining steps (overlapping, long or
//     short-period copies) follow in order, one step of every lane per round trip.
// Block headers and table builds still read the stream directly (a few per block).  52 GB/s of inflated
// bytes on the 10 M-read BAM (2.1 GB in 40 ms), 130x one zlib thread.
// =================================================================================================
constexpr int I2_QUEUE = 16;                    // queued matches per lane
#ifndef PJB_I2_LITS
#define PJB_I2_LITS 3
#endif
constexpr int I2_LITS = PJB_I2_LITS;            // literals decoded per iteration before the (one) length/distance pair
// u16 entry offsets inside a lane's LDS area.  There is no first-level lookup table: with 64 lanes some lane holds a
// long code in every step, so the canonical search ran every time anyway -- and the 576 bytes of first-level tables
// were what limited a CU to two workgroups (one wave on each of two SIMDs of four).  Every code is decoded
// canonically: limits per code length in registers, symbols in code order in LDS.  612 B per lane: four workgroups
// per CU, a wave on every SIMD.
constexpr iu32 I2_LADJ = 0, I2_DADJ = 16;       // per code length: slot of its first code minus that code (mod 2^16)
constexpr iu32 I2_LLONG = 32;                   // literal/length symbols in code order: 288 low bytes (144; the fixed code has 288) + 288 ninth bits (18)
constexpr iu32 I2_LLONG_HI = I2_LLONG + 144;
constexpr iu32 I2_DLONG = I2_LLONG + 162;       // distance symbols in code order: 30 bytes (15)
constexpr iu32 I2_RING = I2_DLONG + 16;         // 16 words of input; while a block header is read: the table build's scratch
constexpr iu32 I2_Q = I2_RING + 32;             // I2_QUEUE x 2 words
constexpr iu32 I2_LANE_U16 = I2_Q + 4 * I2_QUEUE;
constexpr int I2_LDS_BYTES = (int)I2_LANE_U16 * 2 * 64 + 256; // + the shared length / distance tables (2 x 32 words)
static_assert(I2_LANE_U16 % 2 == 0 && I2_RING % 2 == 0 && I2_Q % 2 == 0, "word-aligned LDS areas");
static_assert(4 * I2_LDS_BYTES <= 160 * 1024, "four workgroups per CU");

struct Lane2 {
    unsigned short *p; // &lds[lane * 2]
    __device__ __forceinline__ unsigned short &h(iu32 e) const { return p[(e >> 1) * 128 + (e & 1)]; }
    __device__ __forceinline__ iu32 &w(iu32 e_even) const { return *(iu32 *)(p + (e_even >> 1) * 128); } // 32-bit word at an even entry
    __device__ __forceinline__ iu32 byte(iu32 base, iu32 i) const { return (h(base + (i >> 1)) >> ((i & 1) * 8)) & 0xffu; }
    __device__ __forceinline__ void set_byte(iu32 base, iu32 i, iu32 v) const {
        unsigned short &x = h(base + (i >> 1));
        x = (unsigned short)((i & 1) ? ((x & 0x00ffu) | (v << 8)) : ((x & 0xff00u) | v));
    }
};

// Tables of one tree, canonical: with v = the next 15 bits, first bit on top, the code length is the smallest len with
// v < lim[len] (lim[len] = (first code + count of length len) << (15 - len), a non-decreasing sequence: 1 + the
// number of limits that v reaches), and the symbol is lng[(v >> (15 - len)) + adj[len]] where lng lists the symbols
// in code order (bytes, plus their 9th bit at lng_hi for the literal/length tree) and adj[len] = slot of the first
// code of that length - that first code (mod 2^16).  The limits live in registers (Lim), adj and lng in LDS.
// `codes`: the tree of the code-length alphabet.  Like zlib's inflate_table (inftrees.c), an over-subscribed set is an
// error and so is an incomplete one, except a literal/length or distance set that consists of a single 1-bit code.
struct Lim {
    iu32 v[16]; // [1..15]
};
__device__ int inf2_build(const Lane2 L, Lim &lim, iu32 adj, iu32 lng, iu32 lng_hi, const uint8_t *lens, int n, bool codes = false) {
    const iu32 cnt = I2_RING, run = I2_RING + 16; // scratch (the ring is dead while a header is read): codes per length, running slot
    for (int i = 0; i < 16; i++) L.h(cnt + i) = 0;
    for (int i = 0; i < n; i++) L.h(cnt + lens[i]) = L.h(cnt + lens[i]) + 1;
    L.h(cnt) = 0; // unused codes do not count
    int left = 1, max_len = 0;
    for (int len = 1; len <= 15; len++) {
        left <<= 1;
        left -= (int)L.h(cnt + len);
        if (left < 0) return INF_ERR_CODELENS;
        if (L.h(cnt + len)) max_len = len;
    }
    if (max_len > 0 && left > 0 && (codes || max_len != 1)) return INF_ERR_CODELENS; // incomplete set
    // symbols in code order: by length, then by symbol
    iu32 slot = 0;
    for (int len = 1; len <= 15; len++) {
        L.h(run + len) = (unsigned short)slot;
        slot += L.h(cnt + len);
    }
    for (int s = 0; s < n; s++) {
        const int len = lens[s];
        if (len == 0) continue;
        const iu32 at = L.h(run + len);
        L.h(run + len) = (unsigned short)(at + 1);
        L.set_byte(lng, at, (iu32)s & 0xffu);
        if (lng_hi) {
            unsigned short &m = L.h(lng_hi + (at >> 4));
            m = (unsigned short)((s & 0x100) ? (m | (1u << (at & 15))) : (m & ~(1u << (at & 15))));
        }
    }
    // limits and adjustments from the canonical first codes
    iu32 code = 0;
    slot = 0;
    lim.v[0] = 0;
#pragma unroll
    for (int len = 1; len <= 15; len++) {
        const iu32 c = L.h(cnt + len);
        code = (code + (len > 1 ? (iu32)L.h(cnt + len - 1) : 0u)) << 1; // first code of this length
        lim.v[len] = (code + c) << (15 - len);                           // <= 2^15
        L.h(adj + len) = (unsigned short)(slot - code);
        slot += c;
    }
    return 0;
}

// symbol << 4 | len of the code at the low end of `bits` (at least 15 valid bits), 0 if invalid
template <bool NINTH_BIT>
__device__ __forceinline__ iu32 inf2_decode(const Lane2 L, const Lim &lim, iu32 adj, iu32 lng, iu32 lng_hi, iu32 bits) {
    const iu32 v15 = __brev(bits) >> 17; // the next 15 bits, first bit in the top position
    iu32 len = 1;
#pragma unroll
    for (int l = 1; l <= 15; l++) len += v15 >= lim.v[l] ? 1u : 0u; // registers, no branches
    if (len > 15) return 0;
    const iu32 i = ((v15 >> (15 - len)) + L.h(adj + len)) & 0x1ffu; // < 286 for a valid code set
    iu32 sym = L.byte(lng, i);
    if (NINTH_BIT) sym |= ((L.h(lng_hi + (i >> 4)) >> (i & 15)) & 1u) << 8;
    return (sym << 4) | len;
}

// every vector memory operation of this wave has completed (vmcnt = 0, other counters untouched).  Placed where
// conditionally issued loads end, so that the compiler does not have to assume them pending later on -- it would
// then drain the stores of the decode loop whenever one of their registers is reused.
__device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt(0x0F70); }

struct Ring2 { // 16-word LDS ring + bit buffer
    iu32 ri, rf;       // read index, words available
    const uint8_t *gp; // next input byte to fetch into the ring
    iu64 bb;
    int nb;
};

// (the one-kernel bgzf_inflate of round 2 -- decode and copies in one lane-per-block kernel, 52 GB/s -- lived here until round 4;
// what follows replaced it.  The LDS layout above and the table builders are shared.)

// =================================================================================================
// Round 3: the inflate in two kernels.
//
// What bounded bgzf_inflate above (SQ counters, profiles/r02f_inflate_sq_counters.txt): a lane-block moved 64 KB through
// ~185 000 single-lane memory transactions -- one byte store per literal, 4 + 2 + 1-byte pieces per copy step, the
// loads of every copy -- and a CU retires those at about one per cycle: ~40 % of the kernel, all of it inside "memory
// phases" in which the whole wave stood still (50 % of wave cycles waiting).  LZ77 copies are exactly the part of
// DEFLATE that is NOT serial per block, so they leave the lane-per-block kernel:
//
//   bgzf_decode   lane per BGZF block, as before, but the lane only DECODES: literals go to their final place, a
//                 length/distance pair leaves a 3-byte token (length - 3, distance - 1) in the first bytes of the gap
//                 it will fill (a match is at least 3 bytes long, so the token always fits and needs no memory of its
//                 own) and one bit in a per-block bitmap of match starts (8 KB per block, a 64-bit word kept in a
//                 register and stored when the position leaves it).  No match queue (LDS per lane 612 -> 484 bytes:
//                 five workgroups per CU), no loads but the input refill, nothing that depends on what a copy yields.
//   bgzf_resolve  wave per BGZF block, no LDS (so it runs beside resident decode workgroups): walks the bitmap 1024
//                 positions at a time, deals the matches out 64 at a time in output order (lane = match), and copies
//                 in rounds: a match is ready when its source lies before the first unfinished match of the batch
//                 (everything before that is final: literals were in place, earlier matches are done); ready matches
//                 copy side by side in 16-byte steps, overlapping ones (distance < length) as periodic patterns read
//                 from their final first period only.  The first unfinished match is always ready, so a batch takes
//                 at most 64 rounds, typically 2-5.
// Both kernels keep the block status words of bgzf_inflate; resolve skips blocks that failed.
// =================================================================================================
constexpr iu32 I3_RING = I2_DLONG + 16;
constexpr iu32 I3_LANE_U16 = I3_RING + 32; // no match queue
constexpr int I3_LDS_BYTES = (int)I3_LANE_U16 * 2 * 64 + 256;
constexpr size_t INF_BITMAP_WORDS = 1024; // u64 words per block: one bit per output byte
static_assert(I3_LANE_U16 % 2 == 0 && I3_RING % 2 == 0, "word-aligned LDS areas");
static_assert(5 * I3_LDS_BYTES <= 160 * 1024, "five workgroups per CU");
static_assert(I3_RING == I2_RING, "inf2_build's scratch is the ring");

// 16 bytes at any address as two 8-byte words
struct U128 {
    iu64 lo, hi;
};
__device__ __forceinline__ U128 load128u(const uint8_t *p) {
    U128 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
__device__ __forceinline__ void store128u(uint8_t *p, U128 v) { __builtin_memcpy(p, &v, 16); }

__global__ __launch_bounds__(64) void bgzf_decode(const uint8_t *comp, const InfBlock *blocks, iu32 n_blocks, uint8_t *out, uint8_t *scratch,
                                                    int *status, int *any_error, iu32 *next_block, iu64 *bitmap, int inf_refill) {
    extern __shared__ __attribute__((aligned(16))) unsigned short inf_lds[];
    iu32 *s_len = (iu32 *)(inf_lds + (size_t)I3_LANE_U16 * 64); // base | extra bits << 16
    iu32 *s_dist = s_len + 32;
    if (threadIdx.x < 29) s_len[threadIdx.x] = (iu32)c_len_base[threadIdx.x] | ((iu32)c_len_extra[threadIdx.x] << 16);
    if (threadIdx.x < 30) s_dist[threadIdx.x] = (iu32)c_dist_base[threadIdx.x] | ((iu32)c_dist_extra[threadIdx.x] << 16);
    __syncthreads();
    iu32 b = blockIdx.x * 64 + threadIdx.x;
    bool have = b < n_blocks;
    Lane2 L;
    L.p = inf_lds + threadIdx.x * 2;
    uint8_t *lens = scratch + (size_t)(blockIdx.x * 64 + threadIdx.x) * INF_SCRATCH_PER_LANE;
    InfBlock B;
    B.in_off = B.out_off = 0;
    B.in_len = B.out_len = 0;
    if (have) B = blocks[b];
    const uint8_t *in0 = comp + B.in_off, *in_end = in0 + B.in_len;
    uint8_t *base = out + B.out_off;
    iu64 *bm = bitmap + (size_t)(have ? b : 0) * INF_BITMAP_WORDS;
    iu32 out_len = B.out_len;
    bool more = true;
    enum { ST_HEADER, ST_SYMBOLS, ST_DONE };
    int state = have ? ST_HEADER : ST_DONE;
    int err = 0;
    bool last = false;
    iu64 bitpos = 0;
    iu32 pos = 0;
    iu64 bmw = 0;          // match-start bits of the bitmap word the position is in
    iu32 bmi = 0xffffffffu; // index of that word (none yet)
    Ring2 R;
    R.ri = R.rf = 0;
    R.gp = in0;
    R.bb = 0;
    R.nb = 0;
    Lim limL, limD;
#pragma unroll
    for (int k = 0; k < 16; k++) limL.v[k] = limD.v[k] = 0;

    // The input rings are topped up WITHOUT standing still: when some lane's ring is getting low every lane issues the
    // loads of its next 32 input bytes (two 16-byte loads) and goes on decoding; the words are moved into the rings at
    // the top of the next iteration, by which time they have arrived.  A lane takes as many of its 8 words as its ring
    // has room for (the next fetch starts behind them).  Invariant: after the top of an iteration every lane holds at
    // least 5 words (one iteration uses at most 4): a lane below I3_LOW words has a fetch on its way that brings it
    // to 8 or more.
    U128 pf0, pf1;
    pf0.lo = pf0.hi = pf1.lo = pf1.hi = 0;
    bool pending = false; // (wave-uniform)
    auto refill_issue = [&]() {
        pf0 = load128u(R.gp); // unconditional: the buffer is padded
        pf1 = load128u(R.gp + 16);
        pending = true;
    };
    auto refill_commit = [&]() {
        wait_vm();
        pending = false;
        if (state == ST_SYMBOLS && R.gp > in_end + 128) { // (see bgzf_inflate: no lane reads more than 256 bytes past its payload)
            err = INF_ERR_OVERRUN;
            state = ST_DONE;
        }
        const iu32 take = state == ST_SYMBOLS ? (16u - R.rf < 8u ? 16u - R.rf : 8u) : 0u;
        const iu32 wi = R.ri + R.rf;
        const iu32 w[8] = {(iu32)pf0.lo, (iu32)(pf0.lo >> 32), (iu32)pf0.hi, (iu32)(pf0.hi >> 32),
                           (iu32)pf1.lo, (iu32)(pf1.lo >> 32), (iu32)pf1.hi, (iu32)(pf1.hi >> 32)};
#pragma unroll
        for (int k = 0; k < 8; k++)
            if ((iu32)k < take) L.w(I3_RING + 2 * ((wi + k) & 15u)) = w[k];
        R.rf += take;
        R.gp += 4 * take;
    };
    constexpr iu32 I3_LOW = 10; // a ring below this many words asks for more (an iteration uses at most 4, usually 1 or 2)

    for (;;) {
        if ((int)__popcll(__ballot(state == ST_DONE && more)) >= (__any(state != ST_DONE) ? inf_refill : 1)) {
            const bool fin = state == ST_DONE && more;
            if (fin && have) {
                if (bmi != 0xffffffffu) bm[bmi] = bmw;
                if (!err && pos != out_len) err = INF_ERR_SIZE;
                status[b] = err;
                if (err) atomicOr(any_error, 1);
            }
            const iu64 fm = __ballot(fin);
            iu32 first_new = 0;
            const int leader = __ffsll((long long)fm) - 1;
            if ((int)threadIdx.x == leader) first_new = atomicAdd(next_block, (iu32)__popcll(fm));
            first_new = __shfl(first_new, leader, 64);
            if (fin) {
                b = first_new + (iu32)__popcll(fm & ((1ull << threadIdx.x) - 1ull));
                have = b < n_blocks;
                more = have;
                if (have) {
                    B = blocks[b];
                    in0 = comp + B.in_off;
                    in_end = in0 + B.in_len;
                    base = out + B.out_off;
                    bm = bitmap + (size_t)b * INF_BITMAP_WORDS;
                    out_len = B.out_len;
                    err = 0;
                    last = false;
                    bitpos = 0;
                    pos = 0;
                    bmw = 0;
                    bmi = 0xffffffffu;
                    R.ri = R.rf = 0;
                    R.gp = in0;
                    R.bb = 0;
                    R.nb = 0;
                    state = ST_HEADER;
                }
            }
        }
        if (!__any(state != ST_DONE)) break;
        if (state == ST_HEADER) {
            // ---- block header and tables, read straight from the stream (as in bgzf_inflate)
            BitReader br;
            br.start(in0 + (bitpos >> 3));
            br.refill();
            br.drop((int)(bitpos & 7));
            iu64 used = bitpos & 7;
            auto take = [&](int n) -> iu32 {
                br.refill();
                used += (iu64)n;
                return br.take(n);
            };
            last = take(1);
            const iu32 type = take(2);
            if (type == 0) { // stored
                const int pad = (int)((8 - ((bitpos + 3) & 7)) & 7);
                (void)take(pad);
                const iu32 len = take(16), nlen = take(16);
                const uint8_t *src = in0 + ((bitpos + 3 + (iu64)pad + 32) >> 3);
                if ((len ^ 0xffffu) != nlen) err = INF_ERR_STORED;
                else if (src + len > in_end || pos + len > out_len) err = INF_ERR_OVERRUN;
                else {
                    for (iu32 i = 0; i < len; i++) base[pos + i] = src[i];
                    pos += len;
                    bitpos = (iu64)(src + len - in0) * 8;
                }
                if (err || last) state = ST_DONE;
            } else if (type == 3) {
                err = INF_ERR_BTYPE;
                state = ST_DONE;
            } else {
                int nlit = 288, ndist = 32;
                if (type == 1) {
                    for (int i = 0; i < 144; i++) lens[i] = 8;
                    for (int i = 144; i < 256; i++) lens[i] = 9;
                    for (int i = 256; i < 280; i++) lens[i] = 7;
                    for (int i = 280; i < 288; i++) lens[i] = 8;
                    for (int i = 288; i < 320; i++) lens[i] = 5;
                } else {
                    nlit = (int)take(5) + 257;
                    ndist = (int)take(5) + 1;
                    const int ncl = (int)take(4) + 4;
                    if (nlit > 286 || ndist > 30) err = INF_ERR_CODELENS;
                    if (!err) {
                        for (int i = 0; i < 19; i++) lens[i] = 0;
                        for (int i = 0; i < ncl; i++) lens[c_clen_order[i]] = (uint8_t)take(3);
                        err = inf2_build(L, limL, I2_LADJ, I2_LLONG, 0, lens, 19, true);
                    }
                    int i = 0;
                    while (!err && i < nlit + ndist) {
                        br.refill();
                        const iu32 e = inf2_decode<false>(L, limL, I2_LADJ, I2_LLONG, 0, (iu32)br.bb);
                        if (e == 0) {
                            err = INF_ERR_CODELENS;
                            break;
                        }
                        br.drop((int)(e & 15u));
                        used += e & 15u;
                        const iu32 sym = e >> 4;
                        if (sym < 16) {
                            lens[i++] = (uint8_t)sym;
                        } else {
                            iu32 rep, val = 0;
                            if (sym == 16) {
                                if (i == 0) {
                                    err = INF_ERR_CODELENS;
                                    break;
                                }
                                val = lens[i - 1];
                                rep = 3 + take(2);
                            } else if (sym == 17) {
                                rep = 3 + take(3);
                            } else {
                                rep = 11 + take(7);
                            }
                            if (i + (int)rep > nlit + ndist) {
                                err = INF_ERR_CODELENS;
                                break;
                            }
                            while (rep--) lens[i++] = (uint8_t)val;
                        }
                    }
                    if (!err && lens[256] == 0) err = INF_ERR_CODELENS;
                }
                if (!err) err = inf2_build(L, limD, I2_DADJ, I2_DLONG, 0, lens + nlit, ndist);
                if (!err) err = inf2_build(L, limL, I2_LADJ, I2_LLONG, I2_LLONG_HI, lens, nlit);
                if (err) {
                    state = ST_DONE;
                } else {
                    const iu64 sp = (bitpos & ~7ull) + used;
                    R.gp = in0 + (sp >> 3);
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const iu64 v = load64u(R.gp + 8 * k);
                        L.w(I3_RING + 4 * k) = (iu32)v;
                        L.w(I3_RING + 4 * k + 2) = (iu32)(v >> 32);
                    }
                    R.gp += 64;
                    R.bb = (iu64)L.w(I3_RING) | ((iu64)L.w(I3_RING + 2) << 32);
                    R.ri = 2;
                    R.rf = 14;
                    R.nb = 64 - (int)(sp & 7);
                    R.bb >>= (sp & 7);
                    state = ST_SYMBOLS;
                }
            }
        }
        wait_vm();
        pending = false; // (a fetch issued before the last lane left the symbols is dropped: lanes re-enter with primed rings)
        // ---- symbols
        for (;;) {
            const bool sym_on = state == ST_SYMBOLS;
            if (!__any(sym_on)) break;
            if (pending) refill_commit();
            if (__any(state == ST_SYMBOLS && R.rf < I3_LOW)) refill_issue();
            // Invariant at this point: a lane in ST_SYMBOLS holds more than 32 bits in its bit buffer and at least 5 words in
            // its ring.  The next ring word is fetched ahead, so that topping the buffer up never waits for LDS.
            iu32 nw = L.w(I3_RING + 2 * R.ri);
            auto topup = [&]() { // more than 32 bits again (twice only after an iteration that used all of them)
                if (R.nb <= 32) {
                    R.bb |= (iu64)nw << R.nb;
                    R.nb += 32;
                    R.ri = (R.ri + 1) & 15u;
                    R.rf--;
                    nw = L.w(I3_RING + 2 * R.ri);
                }
            };
            if (state == ST_SYMBOLS) {
                // ---- up to three literal/length codes at once.  The code LENGTHS come out of registers alone (limits per
                // length), so the three bit positions are known before anything is looked up; the lookups -- slot adjustment,
                // then symbol byte and ninth bit -- of all three then travel together: two LDS round trips for three
                // symbols instead of six.  Symbols after a length code or past the valid bits are thrown away.
                const iu32 b1 = (iu32)R.bb;
                const iu32 v1 = __brev(b1) >> 17;
                iu32 len1 = 1;
#pragma unroll
                for (int l = 1; l <= 15; l++) len1 += v1 >= limL.v[l] ? 1u : 0u;
                const iu32 b2 = (iu32)(R.bb >> (len1 & 31u));
                const iu32 v2 = __brev(b2) >> 17;
                iu32 len2 = 1;
#pragma unroll
                for (int l = 1; l <= 15; l++) len2 += v2 >= limL.v[l] ? 1u : 0u;
                const iu32 b3 = (iu32)(R.bb >> ((len1 + len2) & 63u));
                const iu32 v3 = __brev(b3) >> 17;
                iu32 len3 = 1;
#pragma unroll
                for (int l = 1; l <= 15; l++) len3 += v3 >= limL.v[l] ? 1u : 0u;
                const iu32 a1 = L.h(I2_LADJ + (len1 & 15u)), a2 = L.h(I2_LADJ + (len2 & 15u)), a3 = L.h(I2_LADJ + (len3 & 15u));
                const iu32 i1 = ((v1 >> ((15u - len1) & 15u)) + a1) & 0x1ffu, i2 = ((v2 >> ((15u - len2) & 15u)) + a2) & 0x1ffu,
                           i3 = ((v3 >> ((15u - len3) & 15u)) + a3) & 0x1ffu;
                const iu32 y1 = L.h(I2_LLONG + (i1 >> 1)), y2 = L.h(I2_LLONG + (i2 >> 1)), y3 = L.h(I2_LLONG + (i3 >> 1));
                const iu32 h1 = L.h(I2_LLONG_HI + (i1 >> 4)), h2 = L.h(I2_LLONG_HI + (i2 >> 4)), h3 = L.h(I2_LLONG_HI + (i3 >> 4));
                const iu32 s1 = ((y1 >> ((i1 & 1u) * 8u)) & 0xffu) | (((h1 >> (i1 & 15u)) & 1u) << 8);
                const iu32 s2 = ((y2 >> ((i2 & 1u) * 8u)) & 0xffu) | (((h2 >> (i2 & 15u)) & 1u) << 8);
                const iu32 s3 = ((y3 >> ((i3 & 1u) * 8u)) & 0xffu) | (((h3 >> (i3 & 15u)) & 1u) << 8);
                const iu32 c1 = len1, c2 = len1 + len2, c3 = c2 + len3; // (c2 <= 30 < the bits held)
                const bool lit1 = len1 <= 15u && s1 < 256u;
                const bool take2 = lit1, lit2 = take2 && len2 <= 15u && s2 < 256u;
                const bool take3 = lit2 && c3 <= (iu32)R.nb, lit3 = take3 && len3 <= 15u && s3 < 256u;
                const bool bad_code = len1 > 15u || (take2 && len2 > 15u) || (take3 && len3 > 15u);
                const iu32 nlit = (iu32)lit1 + (iu32)lit2 + (iu32)lit3;
                // the symbol that ends the run of literals (if one was decoded), and the bits used
                iu32 sym = 0, used = 0;
                if (!lit1) {
                    sym = s1;
                    used = c1;
                } else if (!lit2) {
                    sym = s2;
                    used = c2;
                } else if (take3 && !lit3) {
                    sym = s3;
                    used = c3;
                } else
                    used = take3 ? c3 : c2;
                if (bad_code) {
                    err = INF_ERR_CODE;
                    state = ST_DONE;
                } else if (pos + nlit > out_len) {
                    err = INF_ERR_OVERRUN;
                    state = ST_DONE;
                } else {
                    if (lit1) base[pos] = (uint8_t)s1;
                    if (lit2) base[pos + 1] = (uint8_t)s2;
                    if (lit3) base[pos + 2] = (uint8_t)s3;
                    pos += nlit;
                    R.bb >>= used;
                    R.nb -= (int)used;
                    topup();
                    topup();
                    if (sym == 256u) { // end of block: where the next header starts
                        const iu64 consumed = (iu64)(R.gp - in0) * 8 - 32ull * R.rf - (iu64)R.nb;
                        if (consumed > (iu64)B.in_len * 8) {
                            err = INF_ERR_OVERRUN;
                            state = ST_DONE;
                        } else {
                            bitpos = consumed;
                            state = last ? ST_DONE : ST_HEADER;
                        }
                    } else if (sym > 256u) {
                        // ---- length / distance pair: 5 + 15 + 13 bits at most, all in the buffer; the base / extra-bit tables
                        // of RFC 1951 3.2.5 in closed form (no lookups)
                        const iu32 ls = sym - 257u;
                        const iu32 xl = ls < 8u || ls >= 28u ? 0u : (ls >> 2) - 1u;
                        const iu32 lbase = ls < 8u ? ls + 3u : ls >= 28u ? 258u : ((4u + (ls & 3u)) << xl) + 3u;
                        const iu32 len = lbase + ((iu32)R.bb & ((1u << xl) - 1u));
                        R.bb >>= xl;
                        R.nb -= (int)xl;
                        const iu32 vd = __brev((iu32)R.bb) >> 17;
                        iu32 dl = 1;
#pragma unroll
                        for (int l = 1; l <= 15; l++) dl += vd >= limD.v[l] ? 1u : 0u;
                        const iu32 di = ((vd >> ((15u - dl) & 15u)) + L.h(I2_DADJ + (dl & 15u))) & 0x1ffu;
                        const iu32 ds = L.byte(I2_DLONG, di & 31u);
                        if (ls > 28u || dl > 15u || ds >= 30u) {
                            err = INF_ERR_CODE;
                            state = ST_DONE;
                        } else {
                            R.bb >>= dl;
                            R.nb -= (int)dl;
                            const iu32 xd = ds < 4u ? 0u : (ds >> 1) - 1u;
                            const iu32 dbase = ds < 4u ? ds + 1u : ((2u + (ds & 1u)) << xd) + 1u;
                            const iu32 dist = dbase + ((iu32)R.bb & ((1u << xd) - 1u));
                            R.bb >>= xd;
                            R.nb -= (int)xd;
                            if (dist > pos) {
                                err = INF_ERR_DIST;
                                state = ST_DONE;
                            } else if (pos + len > out_len) {
                                err = INF_ERR_OVERRUN;
                                state = ST_DONE;
                            } else {
                                // the match's token in the first three bytes of its own gap, its start in the bitmap
                                const iu32 tok = (len - 3u) | ((dist - 1u) << 8);
                                if (pos + 4u <= out_len) {
                                    // one 4-byte store: its last byte is beyond a 3-byte match and belongs to what this lane writes
                                    // NEXT (a literal, or the next token) -- a later store of the same lane to the same byte, which wins
                                    __builtin_memcpy(base + pos, &tok, 4);
                                } else {
                                    const unsigned short lo = (unsigned short)tok;
                                    __builtin_memcpy(base + pos, &lo, 2);
                                    base[pos + 2] = (uint8_t)(tok >> 16);
                                }
                                const iu32 w = pos >> 6;
                                if (w != bmi) {
                                    if (bmi != 0xffffffffu) bm[bmi] = bmw;
                                    bmi = w;
                                    bmw = 0;
                                }
                                bmw |= 1ull << (pos & 63u);
                                pos += len;
                            }
                        }
                        topup();
                        topup();
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void bgzf_resolve(const InfBlock *blocks, iu32 n_blocks, uint8_t *out, const iu64 *bitmap, const int *status) {
    const iu32 lane = threadIdx.x & 63u;
    const iu32 b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_blocks) return; // (whole waves leave; the kernel has no barrier)
    if (status[b] != 0) return;
    const InfBlock B = blocks[b];
    uint8_t *base = out + B.out_off;
    const iu64 *bm = bitmap + (size_t)b * INF_BITMAP_WORDS;
    const iu32 n_words = (B.out_len + 63u) >> 6;
    for (iu32 w0 = 0; w0 < n_words; w0 += 16) { // 1024 output positions per round, 16 per lane
        const iu32 wi = w0 + (lane >> 2);
        const iu64 word = wi < n_words ? bm[wi] : 0ull;
        const iu32 bits = (iu32)(word >> (16u * (lane & 3u))) & 0xffffu;
        const iu32 cnt = (iu32)__popc(bits);
        iu32 inc = cnt; // inclusive scan over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const iu32 t = __shfl_up(inc, d, 64);
            if ((int)lane >= d) inc += t;
        }
        const iu32 pre = inc - cnt;
        const iu32 T = __shfl(inc, 63, 64);
        for (iu32 m0 = 0; m0 < T; m0 += 64) { // 64 matches in output order, lane = match
            const iu32 m = m0 + lane;
            const bool have = m < T;
            // the lane whose 16 positions hold match m: the last one with pre <= m
            iu32 s = 0;
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                const iu32 t = s + (iu32)step;
                const iu32 p = __shfl(pre, (int)(t & 63u), 64);
                if (t < 64u && p <= m) s = t;
            }
            const iu32 k = m - __shfl(pre, (int)s, 64);
            iu32 bs = __shfl(bits, (int)s, 64);
#pragma unroll
            for (int i = 0; i < 5; i++) // (16 positions hold at most 6 match starts: a match is 3 bytes or longer)
                if ((iu32)i < k) bs &= bs - 1u;
            const iu32 p = (w0 + (s >> 2)) * 64u + 16u * (s & 3u) + (have && bs ? (iu32)__ffs((int)bs) - 1u : 0u);
            iu32 len = 0, dist = 1;
            if (have) {
                const iu32 tok = load32u(base + p) & 0x7fffffu; // (a byte past the token is read: the buffer is padded)
                len = (tok & 0xffu) + 3u;
                dist = (tok >> 8) + 1u;
            }
            const iu32 src = p - dist;
            const iu32 need = len < dist ? len : dist; // source bytes that must be final
            bool done = !have || dist > p || p + len > B.out_len; // (bgzf_decode has checked both for every token it wrote)
            // Which matches of the batch does this one wait for?  Those whose destination overlaps its source bytes: the
            // matches are in output order and their destinations do not overlap, so they are a run of lanes [i_lo, i_hi)
            // below this one -- i_lo = matches that end at or before the source, i_hi = matches that start before the
            // source's end.  Everything else the source touches is final: literals were in place, earlier batches are done.
            const iu32 kp = have ? p : 0xffffffffu, ke = have ? p + len : 0xffffffffu; // (lanes without a match sort last)
            const iu32 s_end = src + need;
            iu32 i_lo = 0, i_hi = 0;
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                const iu32 tl = i_lo + (iu32)step, th = i_hi + (iu32)step;
                const iu32 el = __shfl(ke, (int)((tl - 1u) & 63u), 64), ph = __shfl(kp, (int)((th - 1u) & 63u), 64);
                if (el <= src) i_lo = tl;
                if (ph < s_end) i_hi = th;
            }
            const iu64 deps = i_hi > i_lo ? (((1ull << i_hi) - 1ull) & ~((1ull << i_lo) - 1ull)) : 0ull; // (i_hi <= this lane < 64)
            for (int round = 0; round < 64; round++) { // (the first unfinished match waits for nobody: at most 64 rounds)
                const iu64 nd = __ballot(!done);
                if (!nd) break;
                const bool ready = !done && !(nd & deps);
                if (ready) {
                    uint8_t *d = base + p;
                    const uint8_t *sp = base + src;
                    if (dist >= len) { // no overlap: 16 bytes a step
                        iu32 i = 0;
                        for (; i + 16 <= len; i += 16) store128u(d + i, load128u(sp + i));
                        if (i < len) {
                            const U128 v = load128u(sp + i);
                            const iu32 r = len - i;
                            if (r >= 8) {
                                store64u(d + i, v.lo);
                                store_low(d + i + 8, v.hi, r - 8);
                            } else
                                store_low(d + i, v.lo, r);
                        }
                    } else if (dist < 8) { // short period: out of registers
                        const iu64 pmask = (1ull << (8 * dist)) - 1ull;
                        const iu64 per = load64u(sp) & pmask;
                        iu32 ph = 0;
                        for (iu32 i = 0; i < len; i += 8) {
                            iu64 y = ph ? ((per >> (8 * ph)) | (per << (8 * (dist - ph)))) & pmask : per; // rotated period
                            for (iu32 sh = 8 * dist; sh < 64; sh <<= 1) y |= y << sh;
                            store_low(d + i, y, len - i);
                            ph = (ph + 8) % dist;
                        }
                    } else { // 8 <= distance < length: chunk i is the period from offset i mod distance on, wrapping once at most
                        iu32 o = 0;
                        for (iu32 i = 0; i < len; i += 8) {
                            iu64 y = load64u(sp + o);
                            if (o + 8 > dist) {
                                const iu32 kk = dist - o; // 1..7 bytes before the wrap
                                y = (y & ((1ull << (8 * kk)) - 1ull)) | (load64u(sp) << (8 * kk));
                            }
                            store_low(d + i, y, len - i);
                            o += 8;
                            if (o >= dist) o -= dist;
                        }
                    }
                    done = true;
                }
                wait_vm(); // this round's stores are visible to the next round's (and the next batch's) loads
            }
        }
    }
}

// =================================================================================================
// BAM records on the device: the inflated bytes of one target's region -> the SoA batch of pjb_batch.
// Replaces BamReader::next + BamAlignment::init per record (lib/src/bam_reader.cc:134-142,
// lib/src/bam_alignment.cc:71-100).  Records are a linked list (block_size hops), so:
//   bam_find_starts : one wave per 64 KB segment finds the first byte that starts a record.  A candidate
//                     must look like a record (field ranges, sizes that add up, printable NUL-terminated
//                     name, legal CIGAR ops) and so must the next two records after it.  This is a guess,
//                     verified below.
//   bam_walk        : one thread per segment follows the list from its start to the next segment's start
//                     and must land on it exactly (that is the verification: by induction from the true
//                     first record every start is then a true record boundary; a start it contradicts is replaced
//                     by the boundary it reached, bam_repair_start, and the walk repeated), counting records; run a
//                     second time it writes the record offsets.
//   BamSizes scan   : CIGAR ops and sequence words per record -> cig_off / seq_off (generic u64 scan).
//   bam_transcode   : one thread per record writes the fixed-width fields, CIGAR, 4-bit bases of spliced
//                     reads and the XS code (same rules as the host transcoder's xsCode).
// =================================================================================================
constexpr iu32 BAM_SEG = 1u << 16;
constexpr iu64 BAM_NONE = ~0ull;

struct BamRegion {
    const uint8_t *U; // inflated bytes
    iu64 total;       // bytes in U
    iu64 first;       // offset of the first record (known from the index)
    int32_t tid, ref_len, n_ref;
};

__device__ __forceinline__ iu32 ld32u(const uint8_t *p) {
    iu32 v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ iu32 ld16u(const uint8_t *p) {
    unsigned short v;
    __builtin_memcpy(&v, p, 2);
    return v;
}

// header-level plausibility of a record at offset c; returns its block_size, 0 if implausible.
// A record cut off by the end of the buffer is plausible as far as it can be checked.
__device__ iu32 bam_header_ok(const BamRegion R, iu64 c) {
    if (c + 36 > R.total) return 0;
    const uint8_t *r = R.U + c + 4;
    const iu32 bs = ld32u(R.U + c);
    if (bs < 32 || bs > (1u << 28)) return 0;
    const int32_t rt = (int32_t)ld32u(r), rp = (int32_t)ld32u(r + 4);
    if (rt < -1 || rt >= R.n_ref || rp < -1) return 0;
    const iu32 l_name = r[8], n_cig = ld16u(r + 12);
    const int32_t l_seq = (int32_t)ld32u(r + 16);
    const int32_t mt = (int32_t)ld32u(r + 20), mp = (int32_t)ld32u(r + 24);
    if (l_name < 1 || l_seq < 0 || mt < -1 || mt >= R.n_ref || mp < -1) return 0;
    if (32ull + l_name + 4ull * n_cig + (iu64)((l_seq + 1) / 2) + (iu64)l_seq > bs) return 0;
    return bs;
}

__device__ bool bam_record_ok(const BamRegion R, iu64 c) {
    iu32 bs = bam_header_ok(R, c);
    if (!bs) return false;
    const uint8_t *r = R.U + c + 4;
    const iu32 l_name = r[8], n_cig = ld16u(r + 12);
    // name: printable, NUL-terminated; CIGAR: op codes 0..8 -- as far as the buffer reaches
    const iu64 avail = R.total - (c + 4);
    for (iu32 k = 0; k < l_name; k++) {
        if (32ull + k >= avail) return true;
        const uint8_t ch = r[32 + k];
        if (k + 1 == l_name ? ch != 0 : (ch < 33 || ch > 126)) return false;
    }
    for (iu32 k = 0; k < n_cig; k++) {
        if (32ull + l_name + 4ull * k + 4 > avail) return true;
        if ((r[32 + l_name + 4 * k] & 15u) > 8u) return false;
    }
    // the two records after it
    iu64 nx = c + 4 + bs;
    for (int hop = 0; hop < 2; hop++) {
        if (nx + 36 > R.total) return true;
        bs = bam_header_ok(R, nx);
        if (!bs) return false;
        nx += 4 + bs;
    }
    return true;
}

__global__ __launch_bounds__(64) void bam_find_starts(BamRegion R, iu32 n_seg, iu64 *seg_start) {
    const iu32 s = blockIdx.x;
    if (s >= n_seg) return;
    const iu64 lo = (iu64)s * BAM_SEG, hi = lo + BAM_SEG < R.total ? lo + BAM_SEG : R.total;
    if (R.first >= lo && R.first < hi) { // the one start that is known: nothing before it belongs to the target
        if (threadIdx.x == 0) seg_start[s] = R.first;
        return;
    }
    if (hi <= R.first) {
        if (threadIdx.x == 0) seg_start[s] = BAM_NONE;
        return;
    }
    for (iu64 c0 = lo; c0 < hi; c0 += 64) {
        const iu64 c = c0 + threadIdx.x;
        const bool ok = c < hi && bam_record_ok(R, c);
        const iu64 m = __ballot(ok);
        if (m) {
            if (threadIdx.x == 0) seg_start[s] = c0 + (iu32)(__ffsll((long long)m) - 1);
            return;
        }
    }
    if (threadIdx.x == 0) seg_start[s] = BAM_NONE; // a record longer than the segment covers it entirely
}

struct BamWalkOut {
    iu32 *seg_n;       // records per segment
    iu64 *land;        // (count pass) where the segment's walk stopped
    iu64 *rec_off;     // (fill pass) offset of every record
    const iu64 *seg_base; // (fill pass) exclusive scan of seg_n
    iu32 *ctl;         // [0] smallest segment index in which the target's records ended, [1] smallest segment whose
                       // walk did not land on the next start, [2] smallest segment with an invalid record (atomicMin)
};

template <bool FILL>
__global__ __launch_bounds__(256) void bam_walk(BamRegion R, iu32 n_seg, const iu64 *seg_start, BamWalkOut O) {
    const iu32 s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_seg) return;
    iu64 cur = seg_start[s];
    if (cur == BAM_NONE) {
        if (!FILL) O.seg_n[s] = 0;
        return;
    }
    iu64 limit = R.total;
    for (iu32 t = s + 1; t < n_seg; t++) { // next segment that has a start (almost always t = s + 1)
        const iu64 v = seg_start[t];
        if (v != BAM_NONE) {
            limit = v;
            break;
        }
    }
    iu32 n = 0;
    iu64 at = FILL ? O.seg_base[s] : 0;
    const iu32 n_take = FILL ? O.seg_n[s] : 0xffffffffu; // the fill pass stops where the (possibly trimmed) count says
    bool ended = false, partial = false;
    while (cur < limit && n < n_take) {
        if (cur + 36 > R.total) {
            partial = true;
            break;
        }
        const iu32 bs = ld32u(R.U + cur);
        const uint8_t *r = R.U + cur + 4;
        if (bs < 32) {
            if (!FILL) atomicMin(&O.ctl[2], s);
            break;
        }
        const int32_t rt = (int32_t)ld32u(r), rp = (int32_t)ld32u(r + 4);
        if (rt != R.tid || rp >= R.ref_len) { // first record that is not the target's: the region ends here
            ended = true;
            break;
        }
        if (cur + 4 + (iu64)bs > R.total) {
            partial = true;
            break;
        }
        const iu32 l_name = r[8], n_cig = ld16u(r + 12);
        const int32_t l_seq = (int32_t)ld32u(r + 16);
        if (l_seq < 0 || 32ull + l_name + 4ull * n_cig + (iu64)((l_seq + 1) / 2) + (iu64)l_seq > bs) {
            if (!FILL) atomicMin(&O.ctl[2], s);
            break;
        }
        if (FILL) O.rec_off[at + n] = cur;
        n++;
        cur += 4 + (iu64)bs;
    }
    if (!FILL) {
        O.seg_n[s] = n;
        O.land[s] = cur;
        if (ended) atomicMin(&O.ctl[0], s);
        else if (!partial && cur != limit) atomicMin(&O.ctl[1], s); // overshot the next start: that start was not a record
        if (partial && !ended) atomicMin(&O.ctl[6], s);             // the data ends inside one of the target's records
    }
}

// The walk of segment s (whose own start is verified: every earlier walk landed) ran past the start guessed for a later
// segment and stopped at `land[s]`, a true record boundary: the guess was not a record.  Segments the overshooting
// record covers entirely have no start, the one holding the boundary starts there.
__global__ void bam_repair_start(iu64 *seg_start, iu32 n_seg, iu32 s, const iu64 *land, iu64 total, iu32 *ctl) {
    if (blockIdx.x || threadIdx.x) return;
    const iu64 cur = land[s];
    iu32 t = s + 1;
    while (t < n_seg && seg_start[t] == BAM_NONE) t++;
    if (t >= n_seg || cur <= seg_start[t]) { // stopped short of the next start: a damaged record, not a false guess
        ctl[3] = s;
        return;
    }
    for (iu32 v = t; v < n_seg; v++) {
        const iu64 hi = (iu64)(v + 1) * BAM_SEG;
        if (cur >= hi || cur >= total) {
            seg_start[v] = BAM_NONE;
            if (cur >= total && hi >= total) break;
        } else {
            seg_start[v] = cur;
            break;
        }
    }
}

// records of segments after the one in which the target ended do not belong to it
__global__ void bam_trim_segments(iu32 *seg_n, iu32 n_seg, const iu32 *ctl) {
    const iu32 s = blockIdx.x * 256 + threadIdx.x;
    if (s < n_seg && s > ctl[0]) seg_n[s] = 0;
}

struct SegCountFn {
    const iu32 *seg_n;
    __device__ iu64 operator()(iu64 i) const { return seg_n[i]; }
};
struct SegBaseSink {
    iu64 *seg_base;
    __device__ void operator()(iu64 i, iu64, iu64 ex) const { seg_base[i] = ex; }
};

// per record: CIGAR ops << 32 | sequence words (4-byte words of packed bases, only for reads with an N op)
struct BamSizesFn {
    const uint8_t *U;
    const iu64 *rec_off;
    __device__ iu64 operator()(iu64 i) const {
        const uint8_t *r = U + rec_off[i] + 4;
        const iu32 l_name = r[8], n_cig = ld16u(r + 12);
        const int32_t l_seq = (int32_t)ld32u(r + 16);
        const uint8_t *cg = r + 32 + l_name;
        bool spl = false;
        for (iu32 k = 0; k < n_cig; k++) spl |= (cg[4 * k] & 15u) == 3u;
        const iu64 words = (spl && l_seq > 0) ? ((iu64)((l_seq + 1) / 2) + 3) / 4 : 0;
        return ((iu64)n_cig << 32) | words;
    }
};
struct BamOffsetsSink {
    iu32 *cig_off, *seq_off;
    __device__ void operator()(iu64 i, iu64, iu64 ex) const {
        cig_off[i] = (iu32)(ex >> 32);
        seq_off[i] = (iu32)ex;
    }
};

struct BamSoA {
    int32_t *pos;
    uint16_t *flag;
    uint8_t *mapq, *xs;
    int32_t *l_qseq, *mtid, *mpos;
    iu32 *cig_off, *cigar, *seq_off;
    uint8_t *seq4;
    iu64 *name_hash; // junc --extra only (nullptr otherwise): std::hash of deriveName(), see pjb_extra.hip.h
    unsigned short *seq2; // the bases in 2 bits, a 16-bit granule per seq4 word, and the reads' exception bitmap (pjb_batch.seq2 / .seq_exc;
    iu32 *seq_exc;        // nullptr: not written)
};

// XS:A aux tag -> 0 absent / '?' / '.', 1 '+', 2 '-', 3 anything else (same rules as the host transcoder)
__device__ uint8_t bam_xs_code(const uint8_t *p, const uint8_t *end) {
    while (p + 3 <= end) {
        const uint8_t t0 = p[0], t1 = p[1], ty = p[2];
        p += 3;
        const bool is_xs = t0 == 'X' && t1 == 'S';
        iu64 sz = 0;
        switch (ty) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'd': sz = 8; break;
        case 'Z': case 'H': {
            const uint8_t *q = p;
            while (q < end && *q) q++;
            sz = (iu64)(q - p) + 1;
            break;
        }
        case 'B': {
            if (p + 5 > end) return is_xs ? 3 : 0;
            const uint8_t sub = p[0];
            const iu32 cnt = ld32u(p + 1);
            const iu64 es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            sz = 5 + es * cnt;
            break;
        }
        default: return is_xs ? 3 : 0;
        }
        if (is_xs) {
            if (ty != 'A' || p >= end) return 3;
            const char ch = (char)p[0];
            return ch == '+' ? 1 : ch == '-' ? 2 : (ch == '?' || ch == '.') ? 0 : 3;
        }
        if (sz > (iu64)(end - p)) return 0;
        p += sz;
    }
    return 0;
}

// one seq4 word (eight bases, BAM order: high nibble first) -> its 16-bit granule of 2-bit codes (A 0, C 1, G 2, T 3, base j at bits 2j) and,
// in `bad`, bit 4 j + 3 for every base j that is none of the four
__device__ __forceinline__ iu32 seq4_word_to2(iu32 v, iu32 &bad) {
    const iu32 sw = ((v & 0x0F0F0F0Fu) << 4) | ((v >> 4) & 0x0F0F0F0Fu);                      // nibble j = base j
    const iu32 code = (((sw >> 1) & 0x77777777u) - ((sw >> 3) & 0x11111111u)) & 0x33333333u;  // 1 2 4 8 -> 0 1 2 3
    const iu32 pc = sw - ((sw >> 1) & 0x77777777u) - ((sw >> 2) & 0x33333333u) - ((sw >> 3) & 0x11111111u); // bits set per nibble
    const iu32 x = pc ^ 0x11111111u;                                                           // 0 where exactly one is
    bad = (((x & 0x77777777u) + 0x77777777u) | x) & 0x88888888u;
    iu32 y = (code | (code >> 2)) & 0x0F0F0F0Fu;
    y = (y | (y >> 4)) & 0x00FF00FFu;
    return (y | (y >> 8)) & 0xFFFFu;
}
__global__ __launch_bounds__(256) void bam_transcode(const uint8_t *U, const iu64 *rec_off, iu64 n, BamSoA B) {
    const iu64 i = (iu64)blockIdx.x * 256 + threadIdx.x;
    bool exc = true; // (a record without bases, or past the last one)
    if (i < n) {
        const iu64 off = rec_off[i];
        const iu32 bs = ld32u(U + off);
        const uint8_t *r = U + off + 4;
        const iu32 l_name = r[8], n_cig = ld16u(r + 12);
        const int32_t l_seq = (int32_t)ld32u(r + 16);
        const iu64 cig_at = 32 + l_name, seq_at = cig_at + 4ull * n_cig;
        const iu64 seq_bytes = (iu64)((l_seq + 1) / 2);
        const iu64 aux_at = seq_at + seq_bytes + (iu64)l_seq;
        B.pos[i] = (int32_t)ld32u(r + 4);
        B.mapq[i] = r[9];
        B.flag[i] = (uint16_t)ld16u(r + 14);
        B.l_qseq[i] = l_seq;
        B.mtid[i] = (int32_t)ld32u(r + 20);
        B.mpos[i] = (int32_t)ld32u(r + 24);
        B.xs[i] = bam_xs_code(r + aux_at, r + bs);
        if (B.name_hash) B.name_hash[i] = derive_name_hash(r + 32, l_name ? l_name - 1 : 0, ld16u(r + 14));
        const iu32 co = B.cig_off[i];
        for (iu32 k = 0; k < n_cig; k++) B.cigar[co + k] = ld32u(r + cig_at + 4 * k);
        const iu32 so = B.seq_off[i], words = B.seq_off[i + 1] - so;
        if (words) {
            iu32 *dst = (iu32 *)B.seq4 + so;
            const uint8_t *src = r + seq_at;
            iu32 any_bad = 0;
            for (iu32 w = 0; w < words; w++) {
                iu32 v = ld32u(src + 4 * w); // may read up to 3 bytes past the bases: still inside the record (qualities follow)
                const iu64 have = seq_bytes - 4ull * w;
                if (have < 4) v &= (1u << (8 * (iu32)have)) - 1u; // zero the padding of the last word
                dst[w] = v;
                if (B.seq2) {
                    iu32 bad;
                    B.seq2[so + w] = (unsigned short)seq4_word_to2(v, bad);
                    const int64_t left = (int64_t)l_seq - 8 * (int64_t)w; // bases of the read from this word on
                    if (left < 8) bad &= left > 0 ? (1u << (4 * (iu32)left)) - 1u : 0u;
                    any_bad |= bad;
                }
            }
            exc = any_bad != 0 || (iu64)words * 8 < (iu64)l_seq || l_seq <= 0;
        }
    }
    if (B.seq_exc) { // the wavefront's 64 records are two words of the bitmap
        const iu64 m = __ballot(exc);
        const iu64 i0 = (iu64)blockIdx.x * 256 + (threadIdx.x & ~63u);
        if ((threadIdx.x & 63u) == 0 && i0 < n) {
            B.seq_exc[i0 >> 5] = (iu32)m;
            if (i0 + 32 < n) B.seq_exc[(i0 >> 5) + 1] = (iu32)(m >> 32);
        }
    }
}

} // namespace pjb
