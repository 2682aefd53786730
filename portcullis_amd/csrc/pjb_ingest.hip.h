// Device-side ingest of prepared BAM files: BGZF inflate on the GPU.
//
// Replaces, for the junc path, htslib's bgzf_read_block / inflate_block
// (deps/htslib-1.3/bgzf.c:292-316, 421-540): every BGZF block is an independent raw-DEFLATE stream of
// at most 64 KB of output, so a file is tens of thousands of independent decodes.  One LANE owns one
// BGZF block (a 64-lane workgroup = 64 blocks): DEFLATE is serial inside a stream, the parallelism
// is across streams.  Like inflate_block, the CRC32 of the footer is not checked; unlike it, the
// inflated size must equal the footer's ISIZE (it fixes where the block lands in the output).
//
// Huffman decode: first-level tables (8 bits literal/length, 5 bits distance) live in LDS, laid out
// [entry][lane] so the 64 lanes of a wave hit 64 different banks; codes longer than the first level
// continue in second-level tables in a per-lane global scratch area.  Entries are 16 bit:
//   direct  : symbol << 4 | code length (1..15)          (0 = invalid code)
//   link    : 0x8000 | (sub-table offset / 2) << 4 | (sub-table index bits - 1)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pjb {

typedef unsigned long long iu64;
typedef uint32_t iu32;

constexpr int INF_ROOT_L = 8;        // first-level bits, literal/length code
constexpr int INF_ROOT_D = 5;        // first-level bits, distance code
constexpr int INF_SUB_L = 2048;      // second-level entries per lane, literal/length
constexpr int INF_SUB_D = 1024;      // second-level entries per lane, distance
constexpr int INF_PAD = 4096;        // zeroed bytes the compressed buffer carries after its last block
constexpr int INF_LENS = 320;        // code lengths being read (288 + 32)
constexpr int INF_LANE_U16 = (1 << INF_ROOT_L) + (1 << INF_ROOT_D) + 32; // + count[16] + next_code[16]
constexpr int INF_LDS_BYTES = INF_LANE_U16 * 2 * 64;                      // per 64-lane workgroup
constexpr size_t INF_SCRATCH_PER_LANE = (size_t)(INF_SUB_L + INF_SUB_D) * 2 + INF_LENS; // bytes

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
    INF_ERR_TABLE = 9,     // second-level table space exhausted
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

// per-lane LDS area, u16 entries, interleaved over the wave: entry e of lane l sits at ((e >> 1) * 64 + l) * 2 + (e & 1)
struct LaneLds {
    unsigned short *p; // &lds[lane * 2]
    __device__ __forceinline__ unsigned short &operator[](iu32 e) const { return p[(e >> 1) * 128 + (e & 1)]; }
};
constexpr iu32 L_LIT = 0, L_DIST = 1u << INF_ROOT_L, L_COUNT = L_DIST + (1u << INF_ROOT_D), L_NEXT = L_COUNT + 16;

struct BitReader {
    const uint8_t *ip; // next byte to fetch
    iu64 bb;           // bit buffer, LSB first
    int nb;            // valid bits in bb
    __device__ __forceinline__ void refill() { // at least 33 valid bits afterwards (the buffer is padded by 8 bytes)
        if (nb <= 32) {
            iu32 w;
            __builtin_memcpy(&w, ip, 4);
            bb |= (iu64)w << nb;
            ip += 4;
            nb += 32;
        }
    }
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

// Canonical Huffman tables from lens[0..n): first level in LDS at `root` (2^root_bits entries), longer
// codes in sub[0..sub_cap).  Returns 0 or an INF_ERR_* code.
__device__ int inf_build(const LaneLds L, iu32 root, int root_bits, unsigned short *sub, int sub_cap, const uint8_t *lens, int n) {
    for (int i = 0; i < 16; i++) L[L_COUNT + i] = 0;
    for (int i = 0; i < n; i++) L[L_COUNT + lens[i]] = L[L_COUNT + lens[i]] + 1;
    const iu32 rsize = 1u << root_bits;
    for (iu32 e = 0; e < rsize; e++) L[root + e] = 0;
    if (L[L_COUNT] == (unsigned short)n) return 0; // no codes at all: every lookup is invalid (legal for an unused distance tree)
    // over-subscription check and first code of every length
    int left = 1;
    iu32 code = 0;
    for (int len = 1; len <= 15; len++) {
        left <<= 1;
        left -= (int)L[L_COUNT + len];
        if (left < 0) return INF_ERR_CODELENS;
        code = (code + (len > 1 ? L[L_COUNT + len - 1] : 0)) << 1;
        L[L_NEXT + len] = (unsigned short)code;
    }
    // pass 1: the longest code under every first-level prefix that has long codes (kept in the root entry)
    for (int s = 0; s < n; s++) {
        const int len = lens[s];
        if (len > root_bits) {
            const iu32 c = L[L_NEXT + len];
            L[L_NEXT + len] = (unsigned short)(c + 1);
            const iu32 pre = bitrev16(c, len) & (rsize - 1);
            if (L[root + pre] < (unsigned short)len) L[root + pre] = (unsigned short)len;
        }
    }
    int used = 0;
    for (iu32 e = 0; e < rsize; e++) {
        const int mx = L[root + e];
        if (mx) {
            const int sb = mx - root_bits; // 1..7 (15 - 8) or 1..10 (15 - 5)
            if (used + (1 << sb) > sub_cap) return INF_ERR_TABLE;
            // link: offset / 2 in 11 bits (sub-table sizes are even, so offsets are), index bits - 1 in the low 4
            L[root + e] = (unsigned short)(0x8000u | ((iu32)(used >> 1) << 4) | (iu32)(sb - 1));
            for (int k = 0; k < (1 << sb); k++) sub[used + k] = 0;
            used += 1 << sb;
        }
    }
    // pass 2: fill
    code = 0;
    for (int len = 1; len <= 15; len++) {
        code = (code + (len > 1 ? L[L_COUNT + len - 1] : 0)) << 1;
        L[L_NEXT + len] = (unsigned short)code;
    }
    for (int s = 0; s < n; s++) {
        const int len = lens[s];
        if (len == 0) continue;
        const iu32 c = L[L_NEXT + len];
        L[L_NEXT + len] = (unsigned short)(c + 1);
        const iu32 rev = bitrev16(c, len);
        const unsigned short ent = (unsigned short)(((iu32)s << 4) | (iu32)len);
        if (len <= root_bits) {
            for (iu32 e = rev; e < rsize; e += 1u << len) L[root + e] = ent;
        } else {
            const iu32 link = L[root + (rev & (rsize - 1))];
            const int sb = (int)(link & 15u) + 1;
            const iu32 off = ((link >> 4) & 0x7ffu) << 1;
            const int rest = len - root_bits;
            for (iu32 k = rev >> root_bits; k < (1u << sb); k += 1u << rest) sub[off + k] = ent;
        }
    }
    return 0;
}

// decode one symbol; returns the entry (symbol << 4 | len) with the bits consumed, 0 if the code is invalid
__device__ __forceinline__ iu32 inf_decode(const LaneLds L, iu32 root, int root_bits, const unsigned short *sub, BitReader &br) {
    iu32 e = L[root + br.peek(root_bits)];
    if (e & 0x8000u) {
        const int sb = (int)(e & 15u) + 1;
        const iu32 off = ((e >> 4) & 0x7ffu) << 1;
        e = sub[off + (((iu32)(br.bb >> root_bits)) & ((1u << sb) - 1u))];
    }
    br.drop((int)(e & 15u));
    return e;
}

// One lane inflates one BGZF block.  `comp` must be readable INF_PAD bytes past the last payload (a
// truncated last block reads its code lengths before the overrun checks of the symbol loop apply).
__global__ __launch_bounds__(64) void bgzf_inflate(const uint8_t *comp, const InfBlock *blocks, iu32 n_blocks, uint8_t *out,
                                                    uint8_t *scratch, int *status, int *any_error) {
    extern __shared__ __attribute__((aligned(16))) unsigned short inf_lds[];
    const iu32 b = blockIdx.x * 64 + threadIdx.x;
    if (b >= n_blocks) return;
    LaneLds L;
    L.p = inf_lds + threadIdx.x * 2;
    uint8_t *my = scratch + (size_t)b * INF_SCRATCH_PER_LANE;
    unsigned short *sub_l = (unsigned short *)my;
    unsigned short *sub_d = sub_l + INF_SUB_L;
    uint8_t *lens = (uint8_t *)(sub_d + INF_SUB_D);
    const InfBlock B = blocks[b];
    BitReader br;
    br.ip = comp + B.in_off;
    br.bb = 0;
    br.nb = 0;
    const uint8_t *in_end = br.ip + B.in_len;
    uint8_t *op0 = out + B.out_off, *op = op0, *op_end = op0 + B.out_len;
    int err = 0;
    bool last = false;
    while (!last && !err) {
        br.refill();
        last = br.take(1);
        const iu32 type = br.take(2);
        if (type == 0) { // stored: byte-align, LEN, NLEN, raw bytes
            br.drop(br.nb & 7);
            br.refill();
            const iu32 len = br.take(16);
            br.refill();
            const iu32 nlen = br.take(16);
            if ((len ^ 0xffffu) != nlen) {
                err = INF_ERR_STORED;
                break;
            }
            const uint8_t *src = br.ip - (br.nb >> 3); // bytes still in the buffer are the next input bytes
            if (src + len > in_end || op + len > op_end) {
                err = INF_ERR_OVERRUN;
                break;
            }
            for (iu32 i = 0; i < len; i++) op[i] = src[i];
            op += len;
            br.ip = src + len;
            br.bb = 0;
            br.nb = 0;
            continue;
        }
        if (type == 3) {
            err = INF_ERR_BTYPE;
            break;
        }
        int nlit, ndist;
        if (type == 1) { // fixed codes
            for (int i = 0; i < 144; i++) lens[i] = 8;
            for (int i = 144; i < 256; i++) lens[i] = 9;
            for (int i = 256; i < 280; i++) lens[i] = 7;
            for (int i = 280; i < 288; i++) lens[i] = 8;
            for (int i = 288; i < 320; i++) lens[i] = 5;
            nlit = 288;
            ndist = 32;
        } else { // dynamic codes
            br.refill();
            nlit = (int)br.take(5) + 257;
            ndist = (int)br.take(5) + 1;
            const int ncl = (int)br.take(4) + 4;
            if (nlit > 286 || ndist > 30) {
                err = INF_ERR_CODELENS;
                break;
            }
            for (int i = 0; i < 19; i++) lens[i] = 0;
            for (int i = 0; i < ncl; i++) {
                br.refill();
                lens[c_clen_order[i]] = (uint8_t)br.take(3);
            }
            // the code-length code: 7-bit first level in the literal root area, never needs a second level
            if ((err = inf_build(L, L_LIT, 7, sub_l, 0, lens, 19))) break;
            int i = 0;
            while (i < nlit + ndist) {
                br.refill();
                const iu32 e = inf_decode(L, L_LIT, 7, sub_l, br);
                if (e == 0) {
                    err = INF_ERR_CODELENS;
                    break;
                }
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
                        rep = 3 + br.take(2);
                    } else if (sym == 17) {
                        rep = 3 + br.take(3);
                    } else {
                        rep = 11 + br.take(7);
                    }
                    if (i + (int)rep > nlit + ndist) {
                        err = INF_ERR_CODELENS;
                        break;
                    }
                    while (rep--) lens[i++] = (uint8_t)val;
                }
            }
            if (err) break;
            if (lens[256] == 0) { // no end-of-block code
                err = INF_ERR_CODELENS;
                break;
            }
        }
        // lens[] is overwritten by nothing below, but the distance lengths start at nlit: build distance first
        if ((err = inf_build(L, L_DIST, INF_ROOT_D, sub_d, INF_SUB_D, lens + nlit, ndist))) break;
        if ((err = inf_build(L, L_LIT, INF_ROOT_L, sub_l, INF_SUB_L, lens, nlit))) break;
        // ---- symbols
        for (;;) {
            if (br.ip > in_end + 8) { // garbage can decode for a long time: never read far past the payload
                err = INF_ERR_OVERRUN;
                break;
            }
            br.refill();
            iu32 e = inf_decode(L, L_LIT, INF_ROOT_L, sub_l, br);
            if (e == 0) {
                err = INF_ERR_CODE;
                break;
            }
            iu32 sym = e >> 4;
            if (sym < 256) {
                if (op >= op_end) {
                    err = INF_ERR_OVERRUN;
                    break;
                }
                *op++ = (uint8_t)sym;
                continue;
            }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) {
                err = INF_ERR_CODE;
                break;
            }
            iu32 len = c_len_base[sym] + br.take(c_len_extra[sym]);
            br.refill();
            e = inf_decode(L, L_DIST, INF_ROOT_D, sub_d, br);
            if (e == 0 || (e >> 4) >= 30) {
                err = INF_ERR_CODE;
                break;
            }
            const iu32 ds = e >> 4;
            const iu32 dist = c_dist_base[ds] + br.take(c_dist_extra[ds]);
            if (dist > (iu32)(op - op0)) {
                err = INF_ERR_DIST;
                break;
            }
            if (op + len > op_end) {
                err = INF_ERR_OVERRUN;
                break;
            }
            const uint8_t *src = op - dist;
            if (dist >= 8) { // 8 bytes per step; source and destination do not overlap within a step
                while (len >= 8) {
                    iu64 v;
                    __builtin_memcpy(&v, src, 8);
                    __builtin_memcpy(op, &v, 8);
                    src += 8;
                    op += 8;
                    len -= 8;
                }
                while (len--) *op++ = *src++;
            } else { // short period: the pattern is read once, then only stores
                iu64 pat = 0;
                for (iu32 k = 0; k < dist; k++) pat |= (iu64)src[k] << (8 * k);
                iu32 ph = 0;
                while (len--) {
                    *op++ = (uint8_t)(pat >> (8 * ph));
                    ph = ph + 1 == dist ? 0 : ph + 1;
                }
            }
        }
        if (!err && br.ip - (br.nb >> 3) > in_end) err = INF_ERR_OVERRUN;
    }
    if (!err && op != op_end) err = INF_ERR_SIZE;
    status[b] = err;
    if (err) atomicOr(any_error, 1);
}

} // namespace pjb
