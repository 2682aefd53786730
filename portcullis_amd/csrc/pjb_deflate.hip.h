// pjb_deflate.hip.h -- BGZF deflate on the device: the writing side of the BAM files the stages emit (bamfilt's filtered
// BAM, lib/src/bam_writer.cc:58-60 -> bam_write1 -> bgzf_write -> deflate_block, deps/htslib-1.3/bgzf.c:216-262,580-613).
// A BGZF file is a sequence of independent gzip members of at most 0xff00 input bytes each, so a block is a wavefront's
// work and a file is tens of thousands of them:
//   1. LZ77 parse, 64 positions per round: every lane hashes the four bytes at its position, takes the candidate the
//      rounds before left in a 4 K-entry table (LDS), extends the match 8 bytes a step; runs of one byte (quality
//      strings) are found against the byte before.  Then the round's greedy walk -- position, match length, next position
//      -- is a scalar loop over v_readlane (with zlib's one-step laziness: a longer match at the next position demotes
//      this one to a literal), the chosen lanes write their symbols and count them.
//   2. Two dynamic Huffman codes (RFC 1951 3.2.7) from the counts: rank sort by the wavefront, the two-queue merge and
//      zlib's length limiter (trees.c gen_bitlen) by one lane, canonical codes; the code-length code the same way.  A
//      block that would not shrink is stored (BTYPE 00).
//   3. The bits: 64 symbols per round, a wave scan of their widths, ds_or into a few staging words, whole words out.
//   4. CRC-32 of the input by 64 lanes on 64 segments, folded with zlib's crc32_combine arithmetic (x^(8 len) mod P).
// The result inflates to the input with any inflater (tests: zlib and this library's own bgzf_decode); it is NOT the byte
// stream zlib -6 would write -- one hash candidate instead of a chain of 128 -- and is a few per cent larger.
#pragma once

#include "pjb_ingest.hip.h"

namespace pjb {

constexpr u32 DFL_IN_MAX = 0xff00;         // input bytes per block (BGZF_BLOCK_SIZE, bgzf.c:42)
constexpr u32 DFL_SLOT = 65536 + 64;       // bytes of output room per block; the member starts at byte 2 (its data at 20: aligned)
constexpr u32 DFL_SLOT_AT = 2;
constexpr int DFL_HASH_BITS = 12;
constexpr u32 DFL_MIN_MATCH = 4, DFL_MAX_MATCH = 258;
constexpr u32 DFL_SYM_STRIDE = DFL_IN_MAX + 64; // symbols of one block (one per input byte at most, + end of block)
constexpr u32 DFL_STAGE_WORDS = 192;

// (the hash table is only needed by the parse, the tree and staging areas only after it: they share their LDS -- 10.6 KB per
// wavefront, 15 wavefronts per CU.  Measured (rocprofv3: 23.8 ms per launch of 4096 blocks, profiles/r03ai_*): 6 or 15 resident
// wavefronts per CU, 8 K or 4 K table entries, 8 or 16 bytes per extension step, skipping positions an earlier round's match
// covers -- none of it moves the kernel's 11 GB/s.  Every block of a launch is resident at once and the parse is bound by
// instruction issue: ~1 300 instructions per 64-position round, the wavefront pays for its slowest lane in the match
// extension and for every symbol in the scalar walk.  The next step is a lane per 1 KB segment walking it like zlib does
// (positions inside a match skipped) against chains built by a first pass; not built.)
struct DflAfterParse {
    // tree construction
    unsigned short sorted[288], parent[576];
    u32 nodefreq[576];
    uint8_t nodelen[576];
    // code-length sequence
    uint8_t cl_sym[320], cl_ext[320];
    u32 stage[DFL_STAGE_WORDS];
    u32 crc_tab[256];
    u32 crc_part[64];
    u32 bl_count[16], next_code[16]; // (indexed by lengths read from memory: LDS, not registers)
};
struct DflShared {
    union {
        unsigned short head[1 << DFL_HASH_BITS];
        DflAfterParse a;
    };
    u32 freq_ll[288], freq_d[32], freq_cl[20];
    unsigned short code_ll[288], code_d[32], code_cl[20];
    uint8_t len_ll[288], len_d[32], len_cl[20];
    u32 scalars[16]; // 0 n_cl, 1 hlit, 2 hdist, 3 hclen, 4 header bits, 6 stored flag
};

__device__ __forceinline__ u32 bitrev16(u32 v, u32 n) { return __brev(v) >> (32 - n); }

// symbol counts -> code lengths (<= max_bits) and canonical codes (bit-reversed: deflate sends Huffman codes MSB first
// into an LSB-first stream).  The whole wavefront calls it; `n` <= 288.
__device__ void dfl_build_code(DflShared &S, u32 *freq, int n, int max_bits, uint8_t *len, unsigned short *code) {
    const int lane = lane_id();
    // at least two symbols in use (zlib: "force at least two codes of non zero frequency", trees.c build_tree)
    if (lane == 0) {
        int used = 0;
        for (int i = 0; i < n; i++) used += freq[i] != 0;
        for (int i = 0; used < 2 && i < n; i++)
            if (freq[i] == 0) {
                freq[i] = 1;
                used++;
            }
    }
    __syncthreads();
    // rank sort of the symbols in use, ascending (count, symbol)
    int m = 0;
    for (int i = 0; i < n; i++) m += freq[i] != 0; // (uniform: every lane counts the same LDS words)
    for (int i = lane; i < n; i += 64) {
        len[i] = 0;
        const u32 f = freq[i];
        if (!f) continue;
        int rank = 0;
        for (int j = 0; j < n; j++) {
            const u32 g = freq[j];
            rank += g != 0 && (g < f || (g == f && j < i));
        }
        S.a.sorted[rank] = (unsigned short)i;
    }
    __syncthreads();
    if (lane == 0) {
        // two-queue merge: leaves 0 .. m-1 in sorted order, internal nodes m .. 2m-2 in the order they are made
        for (int i = 0; i < m; i++) S.a.nodefreq[i] = freq[S.a.sorted[i]];
        int a = 0, b = m, made = m;
        for (int k = 0; k < m - 1; k++) {
            int pick[2];
            for (int t = 0; t < 2; t++) {
                const bool leaf = a < m && (b >= made || S.a.nodefreq[a] <= S.a.nodefreq[b]);
                pick[t] = leaf ? a++ : b++;
            }
            S.a.nodefreq[made] = S.a.nodefreq[pick[0]] + S.a.nodefreq[pick[1]];
            S.a.parent[pick[0]] = S.a.parent[pick[1]] = (unsigned short)made;
            made++;
        }
        // depths from the root down, limited like zlib's gen_bitlen (trees.c): a node below max_bits stays at max_bits and
        // is counted; then leaves move down from shorter lengths until the code is complete again
        u32 *bl_count = S.a.bl_count, *next_code = S.a.next_code;
        for (int i = 0; i < 16; i++) bl_count[i] = 0;
        const int root = 2 * m - 2;
        S.a.nodelen[root] = 0;
        int overflow = 0;
        for (int node = root - 1; node >= 0; node--) {
            int bits = S.a.nodelen[S.a.parent[node]] + 1;
            if (bits > max_bits) {
                bits = max_bits;
                overflow++;
            }
            S.a.nodelen[node] = (uint8_t)bits;
            if (node < m) bl_count[bits]++;
        }
        while (overflow > 0) {
            int bits = max_bits - 1;
            while (bl_count[bits] == 0) bits--;
            bl_count[bits]--;
            bl_count[bits + 1] += 2;
            bl_count[max_bits]--;
            overflow -= 2;
        }
        // the longest codes to the rarest symbols
        int idx = 0;
        for (int bits = max_bits; bits >= 1; bits--)
            for (u32 c = 0; c < bl_count[bits]; c++) len[S.a.sorted[idx++]] = (uint8_t)bits;
        // canonical codes (RFC 1951 3.2.2)
        u32 c = 0;
        bl_count[0] = 0;
        for (int bits = 1; bits <= max_bits; bits++) {
            c = (c + bl_count[bits - 1]) << 1;
            next_code[bits] = c;
        }
        for (int i = 0; i < n; i++) {
            const int l = len[i];
            code[i] = l ? (unsigned short)bitrev16(next_code[l]++, (u32)l) : 0;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ u32 dfl_len_index(u32 len) { // index into c_len_base / c_len_extra (symbol 257 + index)
    if (len == 258) return 28;
    const u32 t = len - 3;
    if (t < 8) return t;
    const u32 nb = 31u - (u32)__clz((int)t);
    return 8 + 4 * (nb - 3) + ((t >> (nb - 2)) & 3u);
}
__device__ __forceinline__ u32 dfl_dist_code(u32 dist) {
    const u32 t = dist - 1;
    if (t < 4) return t;
    const u32 nb = 31u - (u32)__clz((int)t);
    return 2 * nb + ((t >> (nb - 1)) & 1u);
}

// zlib's crc32_combine arithmetic (crc32.c multmodp / x2nmodp): polynomials over GF(2) modulo the CRC-32 polynomial, reflected
__device__ __forceinline__ u32 crc_multmodp(u32 a, u32 b) {
    u32 m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xedb88320u : b >> 1;
    }
    return p;
}
__device__ __forceinline__ u32 crc_x2nmodp(u32 n, u32 k) { // x^(n * 2^k) mod P
    u32 p = 1u << 31, sq = 1u << 30; // x^0; x^1
    for (u32 i = 0; i < k; i++) sq = crc_multmodp(sq, sq);
    while (n) {
        if (n & 1u) p = crc_multmodp(sq, p);
        n >>= 1;
        sq = crc_multmodp(sq, sq);
    }
    return p;
}

struct DflBits { // lane 0's sequential writer into S.stage
    u32 *stage;
    u32 pos;
    __device__ __forceinline__ void put(u32 v, u32 n) {
        if (!n) return;
        const u32 w = pos >> 5, sh = pos & 31u;
        const iu64 x = (iu64)v << sh;
        stage[w] |= (u32)x;
        if (sh + n > 32) stage[w + 1] |= (u32)(x >> 32);
        pos += n;
    }
};

// One wavefront per block.  in: the stream; block b covers bytes [b * block_bytes, min(n_bytes, (b + 1) * block_bytes)).
// sym: DFL_SYM_STRIDE words per block (scratch); out: DFL_SLOT bytes per block, the member from byte DFL_SLOT_AT on;
// out_size[b]: its length.
__global__ __launch_bounds__(64) void bgzf_deflate(const uint8_t *in, iu64 n_bytes, u32 block_bytes, u32 n_blocks, u32 *sym_all, uint8_t *out_all,
                                                    u32 *out_size) {
    __shared__ DflShared S;
    const u32 blk = blockIdx.x;
    if (blk >= n_blocks) return;
    const int lane = lane_id();
    const uint8_t *src = in + (iu64)blk * block_bytes;
    const iu64 left = n_bytes - (iu64)blk * block_bytes;
    const u32 n = left < block_bytes ? (u32)left : block_bytes;
    u32 *sym = sym_all + (size_t)blk * DFL_SYM_STRIDE;
    uint8_t *slot = out_all + (size_t)blk * DFL_SLOT;
    uint8_t *member = slot + DFL_SLOT_AT, *data = member + 18;

    for (int i = lane; i < (1 << DFL_HASH_BITS); i += 64) S.head[i] = 0xffffu;
    for (int i = lane; i < 288; i += 64) S.freq_ll[i] = 0;
    if (lane < 32) S.freq_d[lane] = 0;
    if (lane < 20) S.freq_cl[lane] = 0;
    __syncthreads();

    // ---- 1. LZ77 parse
    u32 cur = 0, nsym = 0;
    u32 w_next = (u32)lane + 4 <= n ? load32u(src + lane) : 0u;
    for (u32 base = 0; base < n; base += 64) {
        const u32 p = base + (u32)lane;
        const bool can = p + 4 <= n;
        const u32 w = w_next;
        w_next = p + 64 + 4 <= n ? load32u(src + p + 64) : 0u; // (the next round's word is on its way while this round works)
        const u32 h = (w * 2654435761u) >> (32 - DFL_HASH_BITS);
        const u32 cand = can ? (u32)S.head[h] : 0xffffu;
        if (can) S.head[h] = (unsigned short)p;
        u32 mlen = 0, mdist = 0;
        const u32 maxl = can ? (n - p < DFL_MAX_MATCH ? n - p : DFL_MAX_MATCH) : 0u;
        const bool open = can && p >= cur; // (a position inside a match chosen in an earlier round only feeds the table)
        if (open && cand != 0xffffu && p - cand <= 32768u && load32u(src + cand) == w) {
            u32 l = 4;
            while (l + 16 <= maxl && load64u(src + p + l) == load64u(src + cand + l) && load64u(src + p + l + 8) == load64u(src + cand + l + 8)) l += 16;
            if (l + 8 <= maxl && load64u(src + p + l) == load64u(src + cand + l)) l += 8;
            while (l < maxl && src[p + l] == src[cand + l]) l++;
            mlen = l;
            mdist = p - cand;
        }
        if (open && p >= 1) { // a run of the byte before
            const u32 b = src[p - 1];
            if (w == b * 0x01010101u) {
                const iu64 bb = (iu64)b * 0x0101010101010101ull;
                u32 l = 4;
                while (l + 8 <= maxl && load64u(src + p + l) == bb) l += 8;
                while (l < maxl && src[p + l] == b) l++;
                if (l > mlen) {
                    mlen = l;
                    mdist = 1;
                }
            }
        }
        // the round's greedy walk (uniform): which positions start a symbol, which of them stay literals
        iu64 sel = 0, lit = 0;
        const u32 stop = base + 64 < n ? base + 64 : n;
        while (cur < stop) {
            const u32 l = cur - base;
            const u32 ml = (u32)__builtin_amdgcn_readlane((int)mlen, (int)l);
            const u32 ml1 = l < 63 ? (u32)__builtin_amdgcn_readlane((int)mlen, (int)(l + 1)) : 0u;
            sel |= 1ull << l;
            if (ml >= DFL_MIN_MATCH && ml1 <= ml) cur += ml;
            else {
                lit |= 1ull << l;
                cur += 1;
            }
        }
        const bool chosen = (sel >> lane) & 1ull, as_lit = (lit >> lane) & 1ull;
        if (chosen) {
            const u32 rank = (u32)__popcll(sel & ((1ull << lane) - 1ull));
            if (as_lit) {
                const u32 byte = can ? (w & 0xffu) : (u32)src[p];
                sym[nsym + rank] = byte;
                atomicAdd(&S.freq_ll[byte], 1u);
            } else {
                sym[nsym + rank] = 0x80000000u | ((mlen - 3) << 16) | (mdist - 1);
                atomicAdd(&S.freq_ll[257 + dfl_len_index(mlen)], 1u);
                atomicAdd(&S.freq_d[dfl_dist_code(mdist)], 1u);
            }
        }
        nsym += (u32)__popcll(sel);
    }
    if (lane == 0) {
        sym[nsym] = 256u; // end of block
        S.freq_ll[256] = 1;
    }
    nsym++;
    __syncthreads();
    for (int i = lane; i < 256; i += 64) { // (the hash table's LDS is free now)
        u32 c = (u32)i;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        S.a.crc_tab[i] = c;
    }
    __syncthreads();

    // ---- 2. the codes
    dfl_build_code(S, S.freq_ll, 286, 15, S.len_ll, S.code_ll);
    dfl_build_code(S, S.freq_d, 30, 15, S.len_d, S.code_d);
    if (lane == 0) {
        int hlit = 286, hdist = 30;
        while (hlit > 257 && S.len_ll[hlit - 1] == 0) hlit--;
        while (hdist > 1 && S.len_d[hdist - 1] == 0) hdist--;
        // run-length code of the hlit + hdist lengths (RFC 1951 3.2.7: 16 = previous 3-6 times, 17 = zeros 3-10, 18 = zeros 11-138)
        int ncl = 0;
        const int total = hlit + hdist;
        auto at = [&](int i) -> int { return i < hlit ? S.len_ll[i] : S.len_d[i - hlit]; };
        for (int i = 0; i < total;) {
            const int v = at(i);
            int run = 1;
            while (i + run < total && at(i + run) == v) run++;
            if (v == 0 && run >= 3) {
                const int r = run > 138 ? 138 : run;
                S.a.cl_sym[ncl] = r <= 10 ? 17 : 18;
                S.a.cl_ext[ncl++] = (uint8_t)(r <= 10 ? r - 3 : r - 11);
                i += r;
            } else if (v != 0 && run >= 4) { // the value once, then repeats of 3-6
                S.a.cl_sym[ncl] = (uint8_t)v;
                S.a.cl_ext[ncl++] = 0;
                int rest = run - 1;
                i += 1;
                while (rest >= 3) {
                    const int r = rest > 6 ? 6 : rest;
                    S.a.cl_sym[ncl] = 16;
                    S.a.cl_ext[ncl++] = (uint8_t)(r - 3);
                    rest -= r;
                    i += r;
                }
                // (what is left, < 3, comes round again as single values)
            } else {
                S.a.cl_sym[ncl] = (uint8_t)v;
                S.a.cl_ext[ncl++] = 0;
                i += 1;
            }
        }
        for (int i = 0; i < ncl; i++) S.freq_cl[S.a.cl_sym[i]]++;
        S.scalars[0] = (u32)ncl;
        S.scalars[1] = (u32)hlit;
        S.scalars[2] = (u32)hdist;
    }
    __syncthreads();
    dfl_build_code(S, S.freq_cl, 19, 7, S.len_cl, S.code_cl);
    if (lane == 0) {
        int hclen = 19;
        while (hclen > 4 && S.len_cl[c_clen_order[hclen - 1]] == 0) hclen--;
        S.scalars[3] = (u32)hclen;
        iu64 bits = 3 + 5 + 5 + 4 + 3ull * (u32)hclen;
        const int ncl = (int)S.scalars[0];
        for (int i = 0; i < ncl; i++) {
            const int s = S.a.cl_sym[i];
            bits += S.len_cl[s] + (s == 16 ? 2 : s == 17 ? 3 : s == 18 ? 7 : 0);
        }
        for (int i = 0; i < 286; i++) bits += (iu64)S.freq_ll[i] * (S.len_ll[i] + (i >= 257 ? (u32)c_len_extra[i - 257] : 0u));
        for (int i = 0; i < 30; i++) bits += (iu64)S.freq_d[i] * (S.len_d[i] + (u32)c_dist_extra[i]);
        // (symbols that were given a count only to have two codes in use were never sent: the estimate is an upper bound)
        S.scalars[6] = (bits + 7) / 8 >= (iu64)n + 5 ? 1u : 0u;
    }
    __syncthreads();
    const bool stored = S.scalars[6] != 0;
    u32 data_bytes;

    if (stored) { // BFINAL = 1, BTYPE = 00, LEN, NLEN, the bytes
        if (lane == 0) {
            data[0] = 1;
            data[1] = (uint8_t)(n & 0xff);
            data[2] = (uint8_t)(n >> 8);
            data[3] = (uint8_t)(~n & 0xff);
            data[4] = (uint8_t)((~n >> 8) & 0xff);
        }
        for (u32 i = (u32)lane; i < n; i += 64) data[5 + i] = src[i];
        data_bytes = 5 + n;
    } else {
        // ---- 3. the bits: header by one lane, symbols 64 at a time
        for (int i = lane; i < (int)DFL_STAGE_WORDS; i += 64) S.a.stage[i] = 0;
        __syncthreads();
        u32 word_at = 0; // words of `data` written so far
        u32 bitpos;      // bits in the staging area
        if (lane == 0) {
            DflBits B{S.a.stage, 0};
            const int ncl = (int)S.scalars[0], hlit = (int)S.scalars[1], hdist = (int)S.scalars[2], hclen = (int)S.scalars[3];
            B.put(5, 3); // BFINAL 1, BTYPE 10
            B.put((u32)(hlit - 257), 5);
            B.put((u32)(hdist - 1), 5);
            B.put((u32)(hclen - 4), 4);
            for (int i = 0; i < hclen; i++) B.put(S.len_cl[c_clen_order[i]], 3);
            for (int i = 0; i < ncl; i++) {
                const int s = S.a.cl_sym[i];
                B.put(S.code_cl[s], S.len_cl[s]);
                if (s >= 16) B.put(S.a.cl_ext[i], s == 16 ? 2 : s == 17 ? 3 : 7);
            }
            S.scalars[4] = B.pos; // (<= 17 + 57 + 316 * 14 bits: inside the staging area)
        }
        __syncthreads();
        bitpos = S.scalars[4];
        auto flush_words = [&]() { // whole words of the staging area -> data; the partial word moves to the front
            const u32 full = bitpos >> 5;
            for (u32 i = (u32)lane; i < full; i += 64) reinterpret_cast<u32 *>(data)[word_at + i] = S.a.stage[i];
            __syncthreads();
            const u32 tail = S.a.stage[full];
            __syncthreads();
            for (u32 i = (u32)lane; i <= full + 2 && i < DFL_STAGE_WORDS; i += 64) S.a.stage[i] = 0;
            __syncthreads();
            if (lane == 0) S.a.stage[0] = tail;
            __syncthreads();
            word_at += full;
            bitpos &= 31u;
        };
        flush_words();
        for (u32 sbase = 0; sbase < nsym; sbase += 64) {
            const u32 i = sbase + (u32)lane;
            iu64 bits = 0;
            u32 nb = 0;
            if (i < nsym) {
                const u32 v = sym[i];
                if (v & 0x80000000u) {
                    const u32 len = ((v >> 16) & 0xffu) + 3, dist = (v & 0x7fffu) + 1;
                    const u32 li = dfl_len_index(len), dc = dfl_dist_code(dist);
                    bits = S.code_ll[257 + li];
                    nb = S.len_ll[257 + li];
                    bits |= (iu64)(len - c_len_base[li]) << nb;
                    nb += c_len_extra[li];
                    bits |= (iu64)S.code_d[dc] << nb;
                    nb += S.len_d[dc];
                    bits |= (iu64)(dist - c_dist_base[dc]) << nb;
                    nb += c_dist_extra[dc];
                } else {
                    bits = S.code_ll[v];
                    nb = S.len_ll[v];
                }
            }
            const u32 inc = wave_iscan(nb);
            const u32 total = (u32)__shfl((int)inc, 63, 64);
            if (nb) {
                const u32 at = bitpos + inc - nb, w = at >> 5, sh = at & 31u;
                const u32 lo = (u32)(bits << sh);
                const iu64 hi = sh ? bits >> (32 - sh) : bits >> 32;
                if (sh == 0) {
                    atomicOr(&S.a.stage[w], (u32)bits);
                    if (nb > 32) atomicOr(&S.a.stage[w + 1], (u32)(bits >> 32));
                } else {
                    atomicOr(&S.a.stage[w], lo);
                    if (sh + nb > 32) atomicOr(&S.a.stage[w + 1], (u32)hi);
                    if (sh + nb > 64) atomicOr(&S.a.stage[w + 2], (u32)(hi >> 32));
                }
            }
            __syncthreads();
            bitpos += total;
            flush_words();
        }
        // the last partial word
        const u32 rest = (bitpos + 7) >> 3;
        if (lane == 0) {
            const u32 wlast = S.a.stage[0];
            for (u32 k = 0; k < rest; k++) data[(size_t)word_at * 4 + k] = (uint8_t)(wlast >> (8 * k));
        }
        data_bytes = word_at * 4 + rest;
    }

    // ---- 4. CRC-32 of the input, gzip header and trailer (bgzf.c:216-262)
    {
        const u32 seg = (((n + 63) / 64) + 3u) & ~3u;
        const u32 a = (u32)lane * seg, b = a + seg < n ? a + seg : n;
        u32 c = 0xffffffffu;
        if (a < b) {
            u32 i = a;
            for (; i + 4 <= b; i += 4) {
                u32 wv = *reinterpret_cast<const u32 *>(src + i); // (block starts and segments are multiples of 4)
                for (int k = 0; k < 4; k++) {
                    c = S.a.crc_tab[(c ^ wv) & 0xffu] ^ (c >> 8);
                    wv >>= 8;
                }
            }
            for (; i < b; i++) c = S.a.crc_tab[(c ^ src[i]) & 0xffu] ^ (c >> 8);
        }
        S.a.crc_part[lane] = c ^ 0xffffffffu;
        __syncthreads();
        if (lane == 0) {
            u32 crc = n ? S.a.crc_part[0] : 0u;
            if (n > seg) {
                const u32 xs = crc_x2nmodp(seg, 3);
                for (u32 k = 1; k * seg < n; k++) {
                    const u32 lenk = (k + 1) * seg <= n ? seg : n - k * seg;
                    crc = crc_multmodp(lenk == seg ? xs : crc_x2nmodp(lenk, 3), crc) ^ S.a.crc_part[k];
                }
            }
            const u32 total = 18 + data_bytes + 8;
            const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
            for (int k = 0; k < 16; k++) member[k] = hdr[k];
            member[16] = (uint8_t)((total - 1) & 0xff);
            member[17] = (uint8_t)((total - 1) >> 8);
            uint8_t *t = data + data_bytes;
            for (int k = 0; k < 4; k++) t[k] = (uint8_t)(crc >> (8 * k));
            for (int k = 0; k < 4; k++) t[4 + k] = (uint8_t)(n >> (8 * k));
            out_size[blk] = total;
        }
    }
}

// the members, back to back: a block per member
__global__ __launch_bounds__(256) void bgzf_pack(const uint8_t *slots, const u32 *size, const iu64 *offset, u32 n_blocks, uint8_t *packed) {
    const u32 b = blockIdx.x;
    if (b >= n_blocks) return;
    const uint8_t *s = slots + (size_t)b * DFL_SLOT + DFL_SLOT_AT;
    uint8_t *d = packed + offset[b];
    const u32 n = size[b];
    for (u32 i = threadIdx.x; i < n; i += 256) d[i] = s[i];
}

} // namespace pjb
