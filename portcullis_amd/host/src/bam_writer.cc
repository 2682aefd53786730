// BamWriter: see bam_writer.hpp.  BGZF per the SAM specification section 4.1; BAI per section 5.2.
#include <portcullis/bam/bam_writer.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <thread>
#include <zlib.h>

namespace portcullis {
namespace bam {

// PORTCULLIS_PROFILE=1: where the writer's time goes (stderr, at close)
namespace {
struct WriterProfile {
    bool on = getenv("PORTCULLIS_PROFILE") != nullptr;
    double gather = 0, compress = 0, fwrite_ = 0, index = 0, tail = 0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
} g_prof;
}  // namespace

// ---- buffer hooks: every block remembers who releases it (the hooks may change while buffers are alive)
namespace {
BufferHooks g_hooks;
struct BlockHead {
    void (*release)(void*);
    void* base;
    uint64_t pad_[6];  // (the payload stays 64-byte aligned)
};
}  // namespace
void setBufferHooks(const BufferHooks& h) { g_hooks = h; }
void* hookedAlloc(size_t bytes) {
    const BufferHooks h = g_hooks;
    void* base = h.alloc ? h.alloc(bytes + sizeof(BlockHead)) : nullptr;
    void (*rel)(void*) = h.alloc ? h.release : nullptr;
    if (!base) {
        base = malloc(bytes + sizeof(BlockHead));
        rel = nullptr;
    }
    if (!base) throw std::bad_alloc();
    BlockHead* bh = static_cast<BlockHead*>(base);
    bh->release = rel;
    bh->base = base;
    return bh + 1;
}
void hookedFree(void* p) {
    if (!p) return;
    BlockHead* bh = static_cast<BlockHead*>(p) - 1;
    if (bh->release) bh->release(bh->base);
    else free(bh->base);
}

template <class B>
static void put32(B& b, uint32_t v) {
    for (int k = 0; k < 4; k++) b.push_back((uint8_t)(v >> (8 * k)));
}
template <class B>
static void put64(B& b, uint64_t v) {
    for (int k = 0; k < 8; k++) b.push_back((uint8_t)(v >> (8 * k)));
}
static inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

static int reg2bin(int64_t beg, int64_t end) {  // SAM specification section 5.3
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

static inline int32_t recordEnd(const uint8_t* rec, size_t len) {  // pos + reference span (at least 1), like bam_endpos
    const int32_t pos = (int32_t)rd32(rec + 8);
    const uint32_t l_name = rec[12], n_cig = rd16(rec + 16);
    int64_t span = 0;
    const uint8_t* cg = rec + 36 + l_name;
    if (36 + (size_t)l_name + 4ull * n_cig <= len)
        for (uint32_t k = 0; k < n_cig; k++) {
            const uint32_t op = rd32(cg + 4 * k), ty = op & 15u;
            if (ty == 0 || ty == 2 || ty == 3 || ty == 7 || ty == 8) span += op >> 4;
        }
    return (int32_t)(pos + (span > 0 ? span : 1));
}

BamWriter::~BamWriter() {
    try {
        if (inflight.valid()) inflight.wait();
        if (writing.valid()) writing.wait();
        if (fp) close();
    } catch (...) {
    }
}

void BamWriter::open(const std::string& headerText, const std::vector<RefSeq>& targets) {
    fp = fopen(path.c_str(), "wb");
    if (!fp) throw BamException("Could not open output BAM file: " + path);
    setvbuf(fp, nullptr, _IOFBF, 4 << 20);
    nTargets = targets.size();
    bins.assign(nTargets, {});
    lin.assign(nTargets, {});
    cacheChunks = nullptr;
    cacheTid = -1;
    pending.clear();
    flushing.clear();
    flushRecs.clear();
    ubase = uflushed;
    pending.insert(pending.end(), {'B', 'A', 'M', 1});
    put32(pending, (uint32_t)headerText.size());
    pending.insert(pending.end(), headerText.begin(), headerText.end());
    put32(pending, (uint32_t)targets.size());
    for (const auto& t : targets) {
        put32(pending, (uint32_t)t.name.size() + 1);
        pending.insert(pending.end(), t.name.begin(), t.name.end());
        pending.push_back(0);
        put32(pending, (uint32_t)t.length);
    }
    flush(true);  // like bam_hdr_write + bgzf_flush: the header has blocks of its own
}

void BamWriter::write(const uint8_t* rec, size_t len) {
    if (!fp) throw BamException("BamWriter::write: file is not open");
    settle();
    if (len < 36) throw BamException("BamWriter::write: not a BAM record");
    if (wantIndex) {
        RecInfo r;
        r.tid = (int32_t)rd32(rec + 4);
        r.pos = (int32_t)rd32(rec + 8);
        r.end = recordEnd(rec, len);
        r.ustart = ubase + pending.size();
        recs.push_back(r);
    }
    pending.insert(pending.end(), rec, rec + len);
    if (pending.size() >= BLOCK * 64 * (size_t)threads) flush(false);
}

void BamWriter::writeRecords(const uint8_t* data, const std::vector<const std::vector<uint64_t>*>& slices, const uint8_t* codes, uint8_t only,
                             PhasePool& workers) {
    if (!fp) throw BamException("BamWriter::writeRecords: file is not open");
    const bool handOver = asyncFlush && (bool)compressor;  // (the flush thread has buffers of its own: gather beside it)
    if (!handOver) settle();
    const size_t ns = slices.size();
    if (ns == 0) return;
    auto keep = [&](size_t flat) { return !codes || (only ? codes[flat] == only : codes[flat] != 0); };
    std::vector<size_t> base(ns + 1, 0), kb(ns + 1, 0), kr(ns + 1, 0);
    for (size_t s = 0; s < ns; s++) base[s + 1] = base[s] + slices[s]->size();
    workers.run(ns, [&](size_t s) {
        size_t bytes = 0, n = 0;
        const std::vector<uint64_t>& off = *slices[s];
        for (size_t k = 0; k < off.size(); k++)
            if (keep(base[s] + k)) {
                bytes += 4 + (size_t)rd32(data + off[k]);
                n++;
            }
        kb[s + 1] = bytes;
        kr[s + 1] = n;
    });
    for (size_t s = 0; s < ns; s++) {
        kb[s + 1] += kb[s];
        kr[s + 1] += kr[s];
    }
    if (kr[ns] == 0) return;
    const double tg0 = WriterProfile::now();
    const size_t p0 = pending.size(), r0 = recs.size();
    if (pending.capacity() < p0 + kb[ns]) pending.reserve(p0 + kb[ns] + (kb[ns] >> 2) + (1u << 20));  // (one allocation for a file's pieces, not a doubling series)
    pending.resize(p0 + kb[ns]);
    if (wantIndex) recs.resize(r0 + kr[ns]);
    workers.run(ns, [&](size_t s) {
        const std::vector<uint64_t>& off = *slices[s];
        uint8_t* dst = pending.data() + p0 + kb[s];
        size_t i = r0 + kr[s];
        for (size_t k = 0; k < off.size(); k++) {
            if (!keep(base[s] + k)) continue;
            const uint8_t* rec = data + off[k];
            const size_t len = 4 + (size_t)rd32(rec);
            memcpy(dst, rec, len);
            if (wantIndex) {
                RecInfo r;
                r.tid = (int32_t)rd32(rec + 4);
                r.pos = (int32_t)rd32(rec + 8);
                r.end = recordEnd(rec, len);
                r.ustart = ubase + (uint64_t)(dst - pending.data());
                recs[i++] = r;
            }
            dst += len;
        }
    });
    g_prof.gather += WriterProfile::now() - tg0;
    if (pending.size() < BLOCK * flushBlocks) return;
    if (handOver) {
        waitFlush();  // (the piece before: `flushing` is free again, flushRecs holds the records it could not close)
        const size_t take = pending.size() / BLOCK * BLOCK, rest = pending.size() - take;
        flushing.swap(pending);
        pending.clear();
        if (pending.capacity() < flushing.capacity()) pending.reserve(flushing.capacity());
        pending.resize(rest);
        if (rest) memcpy(pending.data(), flushing.data() + take, rest);
        flushing.resize(take);
        flushRecs.insert(flushRecs.end(), recs.begin(), recs.end());
        recs.clear();
        ubase += take;
        inflight = std::async(std::launch::async, [this] { flushBuf(flushing, flushRecs, false); });
        return;
    }
    pool = &workers;
    flush(false);
    pool = nullptr;
}

void BamWriter::settle() {
    waitFlush();
    if (!flushRecs.empty()) {
        flushRecs.insert(flushRecs.end(), recs.begin(), recs.end());
        recs.swap(flushRecs);
        flushRecs.clear();
    }
}

void BamWriter::indexRecord(const RecInfo& r, uint64_t vs, uint64_t ve) {
    if (r.tid < 0 || (size_t)r.tid >= nTargets) return;
    const uint32_t bin = (uint32_t)reg2bin(r.pos, r.end);
    if (r.tid != cacheTid || bin != cacheBin || !cacheChunks) {
        cacheChunks = &bins[(size_t)r.tid][bin];  // (std::map: references stay valid while other bins are added)
        cacheTid = r.tid;
        cacheBin = bin;
    }
    auto& ch = *cacheChunks;
    if (!ch.empty() && ch.back().second == vs) ch.back().second = ve;
    else ch.push_back({vs, ve});
    const size_t w0 = (size_t)(std::max(r.pos, 0) >> 14), w1 = (size_t)(std::max(r.end - 1, 0) >> 14);
    auto& L = lin[(size_t)r.tid];
    if (L.size() <= w1) L.resize(w1 + 1, 0);
    for (size_t w = w0; w <= w1; w++)
        if (L[w] == 0) L[w] = vs;
}

// Compresses every complete 0xff00-byte block of `pending` (all of it when final) and writes the blocks in order.
void BamWriter::flush(bool final) {
    flushBuf(pending, recs, final);
    ubase = uflushed;
}

void BamWriter::flushBuf(ByteBuf& pending, std::vector<RecInfo>& recs, bool final) {
    const size_t nblk = final ? (pending.size() + BLOCK - 1) / BLOCK : pending.size() / BLOCK;
    if (nblk == 0) {
        if (final && wantIndex) {  // nothing left to compress, but the last records' ends are now known: the EOF block
            for (size_t i = 0; i < recs.size() && recs[i].vsKnown; i++)
                indexRecord(recs[i], recs[i].vs, i + 1 < recs.size() && recs[i + 1].vsKnown ? recs[i + 1].vs : cwritten << 16);
            recs.clear();
        }
        return;
    }
    const size_t take = std::min(pending.size(), nblk * BLOCK);
    std::vector<uint64_t> coff(nblk + 1, cwritten);
    bool external = false;
    double tp0 = WriterProfile::now();
    ByteBuf& co = cout_[coutCur];
    // (a few blocks -- the header, the last records of a file -- are zlib's: no reason to wait for a device)
    if (compressor && nblk >= 4 && compressor(pending.data(), take, BLOCK, co, csizes_)) {
        if (csizes_.size() != nblk) throw BamException("BamWriter: the block compressor returned the wrong number of blocks");
        for (size_t b = 0; b < nblk; b++) coff[b + 1] = coff[b] + csizes_[b];
        if (coff[nblk] - cwritten != co.size()) throw BamException("BamWriter: the block compressor's sizes do not add up");
        g_prof.compress += WriterProfile::now() - tp0;
        if (g_prof.on && getenv("PORTCULLIS_PROFILE_PIECES"))
            fprintf(stderr, "[writer piece] %s: compress %.3f .. %.3f (%zu blocks)\n", path.c_str(), fmod(tp0, 1000.0), fmod(WriterProfile::now(), 1000.0), nblk);
        // the piece goes to the file beside the index work below and the next piece's compression (the other buffer)
        waitWrite();
        writing = std::async(std::launch::async, [this, &co] {
            const double tw0 = WriterProfile::now();
            if (fwrite(co.data(), 1, co.size(), fp) != co.size()) throw BamException("BamWriter: write failed: " + path);
            g_prof.fwrite_ += WriterProfile::now() - tw0;
        });
        coutCur ^= 1;
        external = true;
    }
    if (!external) waitWrite();
    std::vector<std::vector<uint8_t>> cblk(external ? 0 : nblk);
    std::atomic<size_t> next(0);
    auto work = [&]() {
        std::vector<uint8_t> out(70000);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return;  // (the blocks stay empty: reported below)
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= nblk) break;
            const size_t off = b * BLOCK, len = std::min(BLOCK, take - off);
            deflateReset(&zs);
            zs.next_in = &pending[off];
            zs.avail_in = (uInt)len;
            zs.next_out = out.data();
            zs.avail_out = (uInt)out.size();
            if (deflate(&zs, Z_FINISH) != Z_STREAM_END) continue;
            const size_t clen = out.size() - zs.avail_out;
            std::vector<uint8_t>& o = cblk[b];
            o.reserve(clen + 26);
            const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
            o.insert(o.end(), hdr, hdr + 16);
            o.push_back((uint8_t)((clen + 25) & 0xff));
            o.push_back((uint8_t)((clen + 25) >> 8));
            o.insert(o.end(), out.begin(), out.begin() + (long)clen);
            put32(o, (uint32_t)crc32(crc32(0L, Z_NULL, 0), &pending[off], (uInt)len));
            put32(o, (uint32_t)len);
        }
        deflateEnd(&zs);
    };
    if (external) {
    } else if (pool) pool->run(std::min<size_t>(pool->size() ? pool->size() : 1, nblk), [&](size_t) { work(); });
    else {
        const int nt = (int)std::min<size_t>((size_t)threads, nblk);
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++) th.emplace_back(work);
        work();
        for (auto& t : th) t.join();
    }
    for (size_t b = 0; b < nblk && !external; b++) {
        if (cblk[b].empty()) throw BamException("BamWriter: deflate failed");
        coff[b + 1] = coff[b] + cblk[b].size();
        if (fwrite(cblk[b].data(), 1, cblk[b].size(), fp) != cblk[b].size()) throw BamException("BamWriter: write failed: " + path);
    }
    // virtual offsets: a record's start is known once the block holding its first byte is written, its end is the
    // start of the record behind it (or of the EOF block)
    tp0 = WriterProfile::now();
    if (wantIndex) {
        for (auto& r : recs) {
            if (r.vsKnown) continue;
            const uint64_t rel = r.ustart - uflushed;
            if (rel >= take) break;
            const size_t bk = (size_t)(rel / BLOCK);
            r.vs = (coff[bk] << 16) | (rel - bk * BLOCK);
            r.vsKnown = true;
        }
        size_t done = 0;
        for (size_t i = 0; i < recs.size(); i++) {
            if (!recs[i].vsKnown) break;
            uint64_t ve;
            if (i + 1 < recs.size()) {
                if (!recs[i + 1].vsKnown) break;
                ve = recs[i + 1].vs;
            } else if (final) {
                ve = coff[nblk] << 16;
            } else
                break;
            indexRecord(recs[i], recs[i].vs, ve);
            done = i + 1;
        }
        recs.erase(recs.begin(), recs.begin() + (long)done);
    }
    g_prof.index += WriterProfile::now() - tp0;
    tp0 = WriterProfile::now();
    cwritten = coff[nblk];
    uflushed += take;
    pending.erase(pending.begin(), pending.begin() + (long)take);
    g_prof.tail += WriterProfile::now() - tp0;
}

void BamWriter::close() {
    if (!fp) return;
    settle();
    flush(true);
    waitWrite();
    static const uint8_t eof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (fwrite(eof, 1, 28, fp) != 28) throw BamException("BamWriter: write failed: " + path);
    fclose(fp);
    fp = nullptr;
    if (g_prof.on)
        fprintf(stderr, "[writer profile] %s: gather %.3f s, compress %.3f s, fwrite %.3f s, index %.3f s, tail %.3f s\n", path.c_str(), g_prof.gather,
                g_prof.compress, g_prof.fwrite_, g_prof.index, g_prof.tail);
    if (!wantIndex) return;
    std::vector<uint8_t> o = {'B', 'A', 'I', 1};
    put32(o, (uint32_t)nTargets);
    for (size_t c = 0; c < nTargets; c++) {
        put32(o, (uint32_t)bins[c].size());
        for (auto& kv : bins[c]) {
            put32(o, kv.first);
            put32(o, (uint32_t)kv.second.size());
            for (auto& ch : kv.second) {
                put64(o, ch.first);
                put64(o, ch.second);
            }
        }
        put32(o, (uint32_t)lin[c].size());
        uint64_t last = 0;
        for (uint64_t v : lin[c]) {
            if (v) last = v;
            put64(o, last);
        }
    }
    FILE* f = fopen((path + ".bai").c_str(), "wb");
    if (!f) throw BamException("Could not write BAM index: " + path + ".bai");
    fwrite(o.data(), 1, o.size(), f);
    fclose(f);
}

}  // namespace bam
}  // namespace portcullis
