#include <portcullis/bam/fast_inflate.hpp>

#include <cstring>
#include <mutex>

namespace portcullis {
namespace bam {
namespace {

// ---- table entries: value << 16 | flags << 8 | extra << 4 | len
//   len    bits this lookup consumes (a second-level entry: the code's bits beyond the first-level index)
//   extra  extra bits that follow the code (lengths, distances); first-level entry of a long code: the second level's index bits
//   value  literal / base length / base distance / start of the second-level table
//   F_LIT2: two literals in one entry (value = second << 8 | first, len = both codes' bits) -- where the index bits behind a
//   literal's code hold a second literal's whole code
constexpr uint32_t F_LITERAL = 1u << 8, F_EOB = 2u << 8, F_SUB = 4u << 8, F_INVALID = 8u << 8, F_LIT2 = 16u << 8;
constexpr int LIT_BITS = 11, DIST_BITS = 8, PRE_BITS = 7;
constexpr int LIT_CAP = (1 << LIT_BITS) + 1280, DIST_CAP = (1 << DIST_BITS) + 256, PRE_CAP = 1 << PRE_BITS;

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t litlenInfo(int s) {
    if (s < 256) return ((uint32_t)s << 16) | F_LITERAL;
    if (s == 256) return F_EOB;
    if (s < 286) return ((uint32_t)LEN_BASE[s - 257] << 16) | ((uint32_t)LEN_EXTRA[s - 257] << 4);
    return F_INVALID;  // 286, 287: in the fixed code, never in a stream
}
inline uint32_t distInfo(int s) { return s < 30 ? ((uint32_t)DIST_BASE[s] << 16) | ((uint32_t)DIST_EXTRA[s] << 4) : F_INVALID; }
inline uint32_t preInfo(int s) { return (uint32_t)s << 16; }

inline uint32_t reverseBits(uint32_t code, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; i++) r |= ((code >> i) & 1u) << (len - 1 - i);
    return r;
}

// Canonical Huffman code of `n` symbols (lens[s] in 0 .. 15) into a two-level table indexed by the next bits of the stream,
// least significant bit first.  false: over-subscribed, or incomplete where zlib does not allow it (inftrees.c: an
// incomplete code passes only if it is a single code of one bit, never for the code-length code), or the table would not
// fit (cannot happen for 15-bit codes and these capacities).
template <class Info>
bool buildTable(const uint8_t* lens, int n, int mainBits, uint32_t* table, int cap, bool isPrecode, Info info) {
    int count[16] = {0};
    for (int s = 0; s < n; s++) count[lens[s]]++;
    int maxLen = 15;
    while (maxLen > 0 && count[maxLen] == 0) maxLen--;
    const int mainSize = 1 << mainBits;
    for (int i = 0; i < mainSize; i++) table[i] = F_INVALID | 1u;
    if (maxLen == 0) return !isPrecode;  // no codes at all: any lookup fails (zlib: the same, when a code is asked for)
    int left = 1;
    for (int len = 1; len <= 15; len++) {
        left <<= 1;
        left -= count[len];
        if (left < 0) return false;
    }
    if (left > 0 && (isPrecode || maxLen != 1)) return false;
    uint32_t next[16];  // the first code of every length (RFC 1951, 3.2.2)
    {
        uint32_t c = 0;
        next[0] = 0;
        for (int len = 1; len <= 15; len++) {
            c = (c + (len > 1 ? (uint32_t)count[len - 1] : 0u)) << 1;
            next[len] = c;
        }
    }
    // second-level tables: as many index bits as the longest code behind the first-level entry needs
    uint8_t subBits[1 << LIT_BITS];
    if (maxLen > mainBits) memset(subBits, 0, (size_t)mainSize);
    uint32_t revOf[288];
    for (int s = 0; s < n; s++) {
        const int len = lens[s];
        if (!len) continue;
        const uint32_t rev = reverseBits(next[len]++, len);
        revOf[s] = rev;
        if (len > mainBits) {
            uint8_t& b = subBits[rev & (uint32_t)(mainSize - 1)];
            if (len - mainBits > b) b = (uint8_t)(len - mainBits);
        }
    }
    int used = mainSize;
    if (maxLen > mainBits)
        for (int i = 0; i < mainSize; i++)
            if (subBits[i]) {
                const int size = 1 << subBits[i];
                if (used + size > cap) return false;
                table[i] = ((uint32_t)used << 16) | F_SUB | ((uint32_t)subBits[i] << 4) | (uint32_t)mainBits;
                for (int k = 0; k < size; k++) table[used + k] = F_INVALID | 1u;
                used += size;
            }
    for (int s = 0; s < n; s++) {
        const int len = lens[s];
        if (!len) continue;
        const uint32_t rev = revOf[s], e = info(s);
        if (len <= mainBits) {
            for (uint32_t i = rev; i < (uint32_t)mainSize; i += 1u << len) table[i] = e | (uint32_t)len;
        } else {
            const uint32_t m = table[rev & (uint32_t)(mainSize - 1)];
            const uint32_t start = m >> 16, size = 1u << ((m >> 4) & 15u);
            for (uint32_t i = rev >> mainBits; i < size; i += 1u << (len - mainBits)) table[start + i] = e | (uint32_t)(len - mainBits);
        }
    }
    return true;
}

// BAM blocks are mostly literals (packed bases, qualities): entries whose index bits hold two whole literal codes yield both.
void pairLiterals(uint32_t* lit) {
    uint32_t one[1 << LIT_BITS];
    memcpy(one, lit, sizeof one);
    for (uint32_t i = 0; i < (1u << LIT_BITS); i++) {
        const uint32_t e = one[i];
        if (!(e & F_LITERAL)) continue;
        const uint32_t l1 = e & 15u, e2 = one[i >> l1];  // (the bits above the index are not known: taken as zeros ...)
        if ((e2 & F_LITERAL) && l1 + (e2 & 15u) <= (uint32_t)LIT_BITS)  // (... which is right if the second code ends inside the index)
            lit[i] = ((e2 >> 16) << 24) | (e & 0x00ff0000u) | F_LITERAL | F_LIT2 | (l1 + (e2 & 15u));
    }
}

struct FixedTables {
    uint32_t lit[LIT_CAP], dist[DIST_CAP];
    FixedTables() {
        uint8_t l[288], d[32];
        for (int s = 0; s < 288; s++) l[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
        for (int s = 0; s < 32; s++) d[s] = 5;
        (void)buildTable(l, 288, LIT_BITS, lit, LIT_CAP, false, litlenInfo);
        (void)buildTable(d, 32, DIST_BITS, dist, DIST_CAP, false, distInfo);
        pairLiterals(lit);
    }
};
const FixedTables& fixedTables() {
    static const FixedTables t;  // (thread-safe initialisation)
    return t;
}

inline uint64_t load64(const uint8_t* p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;  // (little-endian hosts only, like the rest of the reader)
}
inline void store64(uint8_t* p, uint64_t v) { memcpy(p, &v, 8); }

}  // namespace

bool fastInflate(const uint8_t* in, size_t inLen, uint8_t* out0, size_t outLen) {
    const uint8_t* p = in;
    const uint8_t* const inEnd = in + inLen;
    uint8_t* out = out0;
    uint8_t* const outEnd = out0 + outLen;
    uint64_t bitbuf = 0;
    unsigned bitcnt = 0;
    size_t overrun = 0;  // bytes past the end that were fed as zeros (an error only if their bits are consumed)
    uint32_t litTab[LIT_CAP], distTab[DIST_CAP];

#define REFILL()                                                  \
    do {                                                          \
        if (inEnd - p >= 8) {                                     \
            bitbuf |= load64(p) << bitcnt;                        \
            p += (63 - bitcnt) >> 3;                              \
            bitcnt |= 56;                                         \
        } else {                                                  \
            while (bitcnt < 56) {                                 \
                if (p < inEnd) bitbuf |= (uint64_t)*p++ << bitcnt; \
                else overrun++;                                   \
                bitcnt += 8;                                      \
            }                                                     \
        }                                                         \
    } while (0)
// (no branch on F_LIT2: both bytes are stored where two fit, and `out` moves by one or two)
#define PUT_LITERALS(e)                                          \
    do {                                                         \
        if (outEnd - out >= 2) {                                 \
            out[0] = (uint8_t)((e) >> 16);                       \
            out[1] = (uint8_t)((e) >> 24);                       \
            out += 1 + (((e) >> 12) & 1u);                       \
        } else {                                                 \
            if (out == outEnd || ((e) & F_LIT2)) return false;   \
            *out++ = (uint8_t)((e) >> 16);                       \
        }                                                        \
    } while (0)
#define BITS(n) ((uint32_t)(bitbuf & ((1ull << (n)) - 1)))
#define DROP(n)            \
    do {                   \
        bitbuf >>= (n);    \
        bitcnt -= (n);     \
    } while (0)

    for (;;) {
        REFILL();
        const uint32_t last = BITS(1), type = (uint32_t)(bitbuf >> 1) & 3u;
        DROP(3);
        const uint32_t *lit, *dist;
        if (type == 0) {
            // ---- stored: to the next byte boundary, LEN, ~LEN, the bytes
            DROP(bitcnt & 7);
            if ((bitcnt >> 3) < overrun) return false;
            p -= (bitcnt >> 3) - overrun;  // (whole bytes that were buffered but not consumed)
            overrun = 0;
            bitbuf = 0;
            bitcnt = 0;
            if (inEnd - p < 4) return false;
            const uint32_t len = (uint32_t)p[0] | ((uint32_t)p[1] << 8), nlen = (uint32_t)p[2] | ((uint32_t)p[3] << 8);
            p += 4;
            if ((len ^ 0xffffu) != nlen) return false;
            if ((size_t)(inEnd - p) < len || (size_t)(outEnd - out) < len) return false;
            memcpy(out, p, len);
            out += len;
            p += len;
            if (last) break;
            continue;
        } else if (type == 1) {
            const FixedTables& f = fixedTables();
            lit = f.lit;
            dist = f.dist;
        } else if (type == 2) {
            // ---- dynamic: the code-length code, then the two codes' lengths
            const uint32_t nlit = BITS(5) + 257, ndist = (uint32_t)(bitbuf >> 5 & 31) + 1, npre = (uint32_t)(bitbuf >> 10 & 15) + 4;
            DROP(14);
            if (nlit > 286 || ndist > 30) return false;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t plen[19] = {0};
            for (uint32_t i = 0; i < npre; i++) {
                if (bitcnt < 3) REFILL();
                plen[order[i]] = (uint8_t)BITS(3);
                DROP(3);
            }
            uint32_t preTab[PRE_CAP];
            if (!buildTable(plen, 19, PRE_BITS, preTab, PRE_CAP, true, preInfo)) return false;
            uint8_t lens[286 + 30 + 138];
            const uint32_t total = nlit + ndist;
            uint32_t i = 0;
            while (i < total) {
                REFILL();  // (a code of at most 7 bits and at most 7 extra bits)
                const uint32_t e = preTab[BITS(PRE_BITS)];
                if (e & F_INVALID) return false;
                DROP(e & 15u);
                const uint32_t sym = e >> 16;
                if (sym < 16) {
                    lens[i++] = (uint8_t)sym;
                } else if (sym == 16) {
                    if (i == 0) return false;
                    const uint32_t rep = 3 + BITS(2);
                    DROP(2);
                    if (i + rep > total) return false;
                    memset(lens + i, lens[i - 1], rep);
                    i += rep;
                } else {
                    const uint32_t rep = sym == 17 ? 3 + BITS(3) : 11 + BITS(7);
                    DROP(sym == 17 ? 3 : 7);
                    if (i + rep > total) return false;
                    memset(lens + i, 0, rep);
                    i += rep;
                }
            }
            if (lens[256] == 0) return false;  // no end-of-block code
            if (!buildTable(lens, (int)nlit, LIT_BITS, litTab, LIT_CAP, false, litlenInfo)) return false;
            if (!buildTable(lens + nlit, (int)ndist, DIST_BITS, distTab, DIST_CAP, false, distInfo)) return false;
            pairLiterals(litTab);
            lit = litTab;
            dist = distTab;
        } else
            return false;

        // ---- the block's symbols
        for (;;) {
            REFILL();
            uint32_t e = lit[BITS(LIT_BITS)];
            if (e & F_SUB) {
                DROP(LIT_BITS);
                e = lit[(e >> 16) + BITS((e >> 4) & 15u)];
            }
            DROP(e & 15u);
            if (e & F_LITERAL) {
                // up to three literals per refill (15 bits each at most)
                PUT_LITERALS(e);
                e = lit[BITS(LIT_BITS)];
                if (e & F_SUB) {
                    DROP(LIT_BITS);
                    e = lit[(e >> 16) + BITS((e >> 4) & 15u)];
                }
                DROP(e & 15u);
                if (e & F_LITERAL) {
                    PUT_LITERALS(e);
                    e = lit[BITS(LIT_BITS)];
                    if (e & F_SUB) {
                        DROP(LIT_BITS);
                        e = lit[(e >> 16) + BITS((e >> 4) & 15u)];
                    }
                    DROP(e & 15u);
                    if (e & F_LITERAL) {
                        PUT_LITERALS(e);
                        continue;
                    }
                }
                REFILL();  // (a match follows: up to 5 + 15 + 13 more bits)
            }
            if (e & (F_EOB | F_INVALID)) {
                if (e & F_INVALID) return false;
                break;
            }
            const uint32_t lx = (e >> 4) & 15u;
            const size_t length = (e >> 16) + BITS(lx);
            DROP(lx);
            uint32_t d = dist[BITS(DIST_BITS)];
            if (d & F_SUB) {
                DROP(DIST_BITS);
                d = dist[(d >> 16) + BITS((d >> 4) & 15u)];
            }
            if (d & F_INVALID) return false;
            DROP(d & 15u);
            const uint32_t dx = (d >> 4) & 15u;
            const size_t distance = (d >> 16) + BITS(dx);
            DROP(dx);
            if (distance > (size_t)(out - out0) || length > (size_t)(outEnd - out)) return false;
            const uint8_t* src = out - distance;
            if (distance >= 8 && length + 8 <= (size_t)(outEnd - out)) {
                size_t k = 0;
                do {
                    store64(out + k, load64(src + k));
                    k += 8;
                } while (k < length);
            } else if (distance == 1) {
                memset(out, *src, length);
            } else {
                for (size_t k = 0; k < length; k++) out[k] = src[k];
            }
            out += length;
        }
        if (last) break;
    }
#undef REFILL
#undef PUT_LITERALS
#undef BITS
#undef DROP
    // the stream ended: every byte of the output is there, and no bit was taken from beyond the input
    return out == outEnd && overrun * 8 <= bitcnt;
}

}  // namespace bam
}  // namespace portcullis
