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

__device__ __forceinline__ iu64 load64u(const uint8_t *p) {
    iu64 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ void store64u(uint8_t *p, iu64 v) { __builtin_memcpy(p, &v, 8); }

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

// Output of one block: bytes are collected 8 at a time before they are stored, and the last 16 bytes
// stay in registers so that matches with distance <= 16 never read memory (and longer ones never read
// bytes that are still pending).  BAM blocks are full of them: every quality-less read carries a run of
// 0xff as long as the read (distance 1).
struct OutWriter {
    uint8_t *base;  // first byte of the block
    iu32 pos;       // bytes produced
    iu64 acc;       // bytes of the current 8-byte group [pos & ~7, pos)
    iu64 wlo, whi;  // the last 16 bytes, oldest in the low byte of wlo
    __device__ __forceinline__ void put(iu32 x) {
        acc |= (iu64)x << (8 * (pos & 7u));
        wlo = (wlo >> 8) | (whi << 56);
        whi = (whi >> 8) | ((iu64)x << 56);
        pos++;
        if ((pos & 7u) == 0) {
            store64u(base + pos - 8, acc);
            acc = 0;
        }
    }
    __device__ __forceinline__ void put8(iu64 v) {
        const iu32 k = pos & 7u;
        if (k == 0) {
            store64u(base + pos, v);
        } else {
            store64u(base + pos - k, acc | (v << (8 * k)));
            acc = v >> (64 - 8 * k);
        }
        wlo = whi;
        whi = v;
        pos += 8;
    }
    // n (1..7) bytes, low bytes of v first
    __device__ __forceinline__ void putn(iu64 v, iu32 n) {
        v &= (1ull << (8 * n)) - 1ull;
        const iu32 k = pos & 7u;
        acc |= v << (8 * k);
        if (k + n >= 8) {
            store64u(base + pos - k, acc);
            acc = k ? v >> (64 - 8 * k) : 0ull;
        }
        wlo = (wlo >> (8 * n)) | (whi << (64 - 8 * n));
        whi = (whi >> (8 * n)) | (v << (64 - 8 * n));
        pos += n;
    }
    // the next 8 bytes of a match with distance 1..16, from the register window (periodic if dist < 8)
    __device__ __forceinline__ iu64 ahead(iu32 dist) const {
        if (dist >= 8) {
            const iu32 sh = 8 * (16 - dist); // 0..64
            return sh == 0 ? wlo : sh == 64 ? whi : (wlo >> sh) | (whi << (64 - sh));
        }
        iu64 v = whi >> (8 * (8 - dist)); // the last `dist` bytes
        for (iu32 sh = 8 * dist; sh < 64; sh <<= 1) v |= v << sh;
        return v;
    }
    __device__ __forceinline__ void flush() {
        const iu32 k = pos & 7u;
        for (iu32 i = 0; i < k; i++) base[pos - k + i] = (uint8_t)(acc >> (8 * i));
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
        // consume the load inside the branch: the wait for it (which also drains this lane's pending stores)
        // must not sit at the join, where every first-level hit would pay it too
        asm volatile("" : "+v"(e));
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
    br.start(comp + B.in_off);
    const uint8_t *in_end = comp + B.in_off + B.in_len;
    OutWriter ow;
    ow.base = out + B.out_off;
    ow.pos = 0;
    ow.acc = 0;
    ow.wlo = ow.whi = 0;
    const iu32 out_len = B.out_len;
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
            const uint8_t *src = br.byte_pos();
            if (src + len > in_end || ow.pos + len > out_len) {
                err = INF_ERR_OVERRUN;
                break;
            }
            iu32 i = 0;
            for (; i + 8 <= len; i += 8) ow.put8(load64u(src + i));
            for (; i < len; i++) ow.put(src[i]);
            br.start(src + len);
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
        if ((err = inf_build(L, L_DIST, INF_ROOT_D, sub_d, INF_SUB_D, lens + nlit, ndist))) break;
        if ((err = inf_build(L, L_LIT, INF_ROOT_L, sub_l, INF_SUB_L, lens, nlit))) break;
        // ---- symbols.  One flat loop: an iteration either decodes a symbol or moves up to 8 bytes of the
        // pending match, so a lane in a long copy does not stall the 63 others for its whole length.
        iu32 mlen = 0, mdist = 0;
        for (;;) {
            if (mlen) {
                const iu32 n = mlen < 8 ? mlen : 8;
                // distance >= 17: the 8 bytes read were stored at least 9 positions back (at most 7 are pending)
                const iu64 v = mdist <= 16 ? ow.ahead(mdist) : load64u(ow.base + ow.pos - mdist);
                if (n == 8) ow.put8(v);
                else ow.putn(v, n);
                mlen -= n;
                continue;
            }
            if (br.ip > in_end + 32) { // garbage can decode for a long time: never read far past the payload
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
                if (ow.pos >= out_len) {
                    err = INF_ERR_OVERRUN;
                    break;
                }
                ow.put(sym);
                continue;
            }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) {
                err = INF_ERR_CODE;
                break;
            }
            const iu32 len = c_len_base[sym] + br.take(c_len_extra[sym]);
            br.refill();
            e = inf_decode(L, L_DIST, INF_ROOT_D, sub_d, br);
            if (e == 0 || (e >> 4) >= 30) {
                err = INF_ERR_CODE;
                break;
            }
            const iu32 ds = e >> 4;
            const iu32 dist = c_dist_base[ds] + br.take(c_dist_extra[ds]);
            if (dist > ow.pos) {
                err = INF_ERR_DIST;
                break;
            }
            if (ow.pos + len > out_len) {
                err = INF_ERR_OVERRUN;
                break;
            }
            mlen = len;
            mdist = dist;
        }
        if (!err && br.byte_pos() > in_end) err = INF_ERR_OVERRUN;
    }
    if (!err && ow.pos != out_len) err = INF_ERR_SIZE;
    ow.flush();
    status[b] = err;
    if (err) atomicOr(any_error, 1);
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
//                     first record every start is then a true record boundary), counting records; run a
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
        if (ended) atomicMin(&O.ctl[0], s);
        else if (!partial && cur != limit) atomicMin(&O.ctl[1], s); // overshot the next start: that start was not a record
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

__global__ __launch_bounds__(256) void bam_transcode(const uint8_t *U, const iu64 *rec_off, iu64 n, BamSoA B) {
    const iu64 i = (iu64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
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
    const iu32 co = B.cig_off[i];
    for (iu32 k = 0; k < n_cig; k++) B.cigar[co + k] = ld32u(r + cig_at + 4 * k);
    const iu32 so = B.seq_off[i], words = B.seq_off[i + 1] - so;
    if (words) {
        iu32 *dst = (iu32 *)B.seq4 + so;
        const uint8_t *src = r + seq_at;
        for (iu32 w = 0; w < words; w++) {
            iu32 v = ld32u(src + 4 * w); // may read up to 3 bytes past the bases: still inside the record (qualities follow)
            const iu64 have = seq_bytes - 4ull * w;
            if (have < 4) v &= (1u << (8 * (iu32)have)) - 1u; // zero the padding of the last word
            dst[w] = v;
        }
    }
}

} // namespace pjb
