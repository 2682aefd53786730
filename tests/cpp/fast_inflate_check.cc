// fastInflate against zlib (no GPU; built with ASan + UBSan by tests/test_host_fast_inflate.py):
//   fast_inflate_check <seed> <cases>    -- differential cases: zlib's own output at every level / strategy / window, stored
//                                            blocks, multi-block streams, then damaged copies of each (truncated, bits flipped)
//   fast_inflate_check speed <MB> [threads] -- MB/s of both decoders on BAM-like data in 64 KB blocks
// A case passes if fastInflate either declines (the callers then ask zlib) or returns exactly what zlib's inflate returns
// for the same input and output size; on undamaged streams it must not decline.
#include <portcullis/bam/fast_inflate.hpp>

#include <zlib.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

using portcullis::bam::fastInflate;

static std::vector<uint8_t> deflateRaw(const std::vector<uint8_t>& in, int level, int strategy, int memLevel, bool multi) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, memLevel, strategy) != Z_OK) abort();
    std::vector<uint8_t> out(in.size() + in.size() / 8 + 1024);
    zs.next_out = out.data();
    zs.avail_out = (uInt)out.size();
    if (multi && in.size() > 64) {  // several blocks, of different kinds: full flushes in the middle
        const size_t a = in.size() / 3, b = 2 * in.size() / 3;
        zs.next_in = const_cast<uint8_t*>(in.data());
        zs.avail_in = (uInt)a;
        if (deflate(&zs, Z_FULL_FLUSH) != Z_OK) abort();
        zs.avail_in = (uInt)(b - a);
        if (deflate(&zs, Z_SYNC_FLUSH) != Z_OK) abort();
        zs.avail_in = (uInt)(in.size() - b);
    } else {
        zs.next_in = const_cast<uint8_t*>(in.data());
        zs.avail_in = (uInt)in.size();
    }
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) abort();
    out.resize(out.size() - zs.avail_out);
    deflateEnd(&zs);
    return out;
}

// what the reader's zlib path does: inflate(Z_FINISH) into exactly n bytes
static bool zlibInflate(const uint8_t* in, size_t inLen, uint8_t* out, size_t n) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) abort();
    zs.next_in = const_cast<uint8_t*>(in);
    zs.avail_in = (uInt)inLen;
    zs.next_out = out;
    zs.avail_out = (uInt)n;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.avail_out == 0;
    inflateEnd(&zs);
    return ok;
}

static std::vector<uint8_t> makeData(std::mt19937_64& rng, int kind, size_t n) {
    std::vector<uint8_t> d(n);
    switch (kind) {
    case 0:  // random bytes (incompressible: stored blocks at some levels)
        for (auto& c : d) c = (uint8_t)rng();
        break;
    case 1:  // BAM-like: records of repeated structure, 4-letter sequences, runs of quality values
        for (size_t i = 0; i < n;) {
            const size_t len = 80 + rng() % 200;
            const uint8_t q = (uint8_t)(30 + rng() % 10);
            for (size_t k = 0; k < len && i < n; k++, i++) d[i] = k < 36 ? (uint8_t)(k * 7 + (rng() % 3 == 0)) : k < len / 2 ? (uint8_t)("\x11\x12\x14\x18\x21\x22\x24\x28\x41\x42\x44\x48\x81\x82\x84\x88"[rng() % 16]) : q;
        }
        break;
    case 2:  // long runs (distance 1, length 258)
        for (size_t i = 0; i < n; i++) d[i] = (uint8_t)((i / 5000) * 31);
        break;
    case 3:  // text with far repeats (long distances)
        for (size_t i = 0; i < n; i++) d[i] = i >= 30000 && rng() % 8 ? d[i - 30000 + (rng() % 3)] : (uint8_t)('a' + rng() % 26);
        break;
    default:  // few symbols: short codes, incomplete distance codes
        for (auto& c : d) c = (uint8_t)("ab"[rng() % 2]);
    }
    return d;
}

int main(int argc, char** argv) {
    if (argc >= 3 && std::string(argv[1]) == "speed") {
        std::mt19937_64 rng(7);
        const size_t total = (size_t)atoi(argv[2]) << 20, blk = 0xff00;
        const std::vector<uint8_t> data = makeData(rng, 1, total);
        std::vector<std::vector<uint8_t>> comp;
        for (size_t o = 0; o < total; o += blk) comp.push_back(deflateRaw(std::vector<uint8_t>(data.begin() + (long)o, data.begin() + (long)std::min(total, o + blk)), 6, Z_DEFAULT_STRATEGY, 8, false));
        std::vector<uint8_t> out(total);
        if (argc >= 4) {  // speed <MB> <threads>: the reader's loop -- threads take blocks from a counter, outputs side by side
            const int nt = atoi(argv[3]);
            for (int which = 0; which < 2; which++) {
                const auto t0 = std::chrono::steady_clock::now();
                for (int rep = 0; rep < 3; rep++) {
                    std::atomic<size_t> next(0);
                    std::vector<std::thread> th;
                    for (int t = 0; t < nt; t++)
                        th.emplace_back([&] {
                            for (;;) {
                                const size_t b = next.fetch_add(1);
                                if (b >= comp.size()) break;
                                const size_t o = b * blk, n = std::min(blk, total - o);
                                if (which) (void)fastInflate(comp[b].data(), comp[b].size(), out.data() + o, n);
                                else (void)zlibInflate(comp[b].data(), comp[b].size(), out.data() + o, n);
                            }
                        });
                    for (auto& x : th) x.join();
                }
                const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                printf("%d threads, %s: %.0f MB/s%s\n", nt, which ? "fastInflate" : "zlib       ", 3.0 * (double)total / s / 1e6, out == data ? "" : "  WRONG");
            }
            return 0;
        }
        for (int which = 0; which < 2; which++) {
            const auto t0 = std::chrono::steady_clock::now();
            bool ok = true;
            for (int rep = 0; rep < 3; rep++)
                for (size_t b = 0, o = 0; b < comp.size(); b++, o += blk) {
                    const size_t n = std::min(blk, total - o);
                    ok &= which ? fastInflate(comp[b].data(), comp[b].size(), out.data() + o, n) : zlibInflate(comp[b].data(), comp[b].size(), out.data() + o, n);
                }
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("%s: %.0f MB/s%s\n", which ? "fastInflate" : "zlib       ", 3.0 * (double)total / s / 1e6, ok && out == data ? "" : "  WRONG");
        }
        return 0;
    }
    const uint64_t seed = argc >= 2 ? strtoull(argv[1], nullptr, 10) : 1;
    const int cases = argc >= 3 ? atoi(argv[2]) : 200;
    std::mt19937_64 rng(seed);
    long clean = 0, damaged = 0, declinedDamaged = 0;
    for (int c = 0; c < cases; c++) {
        const int kind = (int)(rng() % 5);
        const size_t sizes[] = {0, 1, 2, 3, 9, 70, 300, 4000, 65280, 65536, 20000 + rng() % 40000};
        const size_t n = sizes[rng() % (sizeof sizes / sizeof sizes[0])];
        const std::vector<uint8_t> data = makeData(rng, kind, n);
        const int levels[] = {0, 1, 2, 4, 6, 9};
        const int strategies[] = {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
        const std::vector<uint8_t> comp = deflateRaw(data, levels[rng() % 6], strategies[rng() % 5], 1 + (int)(rng() % 9), rng() % 3 == 0);
        // ---- undamaged: must be taken, must be right; guard bytes either side of the output stay untouched
        std::vector<uint8_t> out(n + 32, 0xA5);
        // (the input in a buffer of its exact size: a read past its end is ASan's to report)
        std::vector<uint8_t> exact(comp);
        if (!fastInflate(exact.data(), exact.size(), out.data() + 16, n)) {
            printf("case %d: declined an undamaged stream (kind %d, %zu bytes)\n", c, kind, n);
            return 1;
        }
        if (n && memcmp(out.data() + 16, data.data(), n) != 0) {
            printf("case %d: wrong bytes (kind %d, %zu bytes)\n", c, kind, n);
            return 1;
        }
        for (int k = 0; k < 16; k++)
            if (out[(size_t)k] != 0xA5 || out[16 + n + (size_t)k] != 0xA5) {
                printf("case %d: wrote outside the output\n", c);
                return 1;
            }
        clean++;
        // a wrong output size must be declined, like zlib's Z_FINISH does
        if (n > 0 && fastInflate(exact.data(), exact.size(), out.data() + 16, n - 1)) {
            printf("case %d: accepted a stream longer than the output\n", c);
            return 1;
        }
        {
            std::vector<uint8_t> big(n + 1 + 32, 0xA5);
            if (fastInflate(exact.data(), exact.size(), big.data() + 16, n + 1)) {
                printf("case %d: accepted a stream shorter than the output\n", c);
                return 1;
            }
        }
        // ---- damaged copies: whatever fastInflate accepts, zlib accepts with the same bytes
        for (int dmg = 0; dmg < 12; dmg++) {
            std::vector<uint8_t> bad(comp);
            if (dmg % 3 == 0 && bad.size() > 1) bad.resize(rng() % bad.size());
            else if (!bad.empty())
                for (int f = 0; f < 1 + dmg % 3; f++) bad[rng() % bad.size()] ^= (uint8_t)(1u << (rng() % 8));
            std::vector<uint8_t> o1(n + 32, 0xA5), o2(n + 1, 0);  // (never a null next_out: zlib calls that a stream error)
            const bool a = fastInflate(bad.data(), bad.size(), o1.data() + 16, n);
            damaged++;
            for (int k = 0; k < 16; k++)
                if (o1[(size_t)k] != 0xA5 || o1[16 + n + (size_t)k] != 0xA5) {
                    printf("case %d/%d: wrote outside the output\n", c, dmg);
                    return 1;
                }
            if (!a) {
                declinedDamaged++;
                continue;
            }
            const bool b = zlibInflate(bad.data(), bad.size(), o2.data(), n);
            if (!b || (n && memcmp(o1.data() + 16, o2.data(), n) != 0)) {
                printf("case %d/%d: accepted what zlib %s (kind %d, %zu bytes)\n", c, dmg, b ? "decodes differently" : "rejects", kind, n);
                return 1;
            }
        }
    }
    printf("ok: %ld undamaged streams, %ld damaged ones (%ld declined)\n", clean, damaged, declinedDamaged);
    return 0;
}
