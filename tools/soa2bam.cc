// soa2bam -- test/bench tooling: turn structure-of-arrays alignment records (the layout of
// pjb_batch, as dumped by tools/e2e_bench.py from portcullis_amd.synth) into a Portcullis prep
// directory: coordinate-sorted BGZF BAM + BAI, FASTA + .fai.  Written from the SAM/BAM spec on zlib.
//
//   soa2bam <prep_dir> <threads> <contig_dir>...
// each <contig_dir> holds: name.txt, genome.u8 and (optionally, if the contig has reads)
// pos.i32 flag.u16 mapq.u8 xs.u8 l_qseq.i32 mtid.i32 mpos.i32 cig_off.u32 cigar.u32 seq_off.u32 seq4.u8
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <thread>
#include <vector>
#include <zlib.h>

template <typename T>
static std::vector<T> slurp(const std::string& path, bool optional = false) {
    std::vector<T> v;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        if (optional) return v;
        fprintf(stderr, "cannot open %s\n", path.c_str());
        exit(2);
    }
    fseeko(f, 0, SEEK_END);
    const off_t sz = ftello(f);
    fseeko(f, 0, SEEK_SET);
    v.resize((size_t)sz / sizeof(T));
    if (sz && fread(v.data(), 1, (size_t)sz, f) != (size_t)sz) exit(2);
    fclose(f);
    return v;
}

static int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

struct Rec {  // where a record starts in the uncompressed stream, for the index
    int32_t tid, pos, end;
    uint64_t ustart;
};

static void put32(std::vector<uint8_t>& b, uint32_t v) {
    for (int k = 0; k < 4; k++) b.push_back((uint8_t)(v >> (8 * k)));
}
static void put16(std::vector<uint8_t>& b, uint16_t v) {
    b.push_back((uint8_t)v);
    b.push_back((uint8_t)(v >> 8));
}

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: soa2bam <prep_dir> <threads> <contig_dir>...\n");
        return 2;
    }
    const std::string prep = argv[1];
    const int nthreads = std::max(1, atoi(argv[2]));
    const int ncontig = argc - 3;
    std::vector<std::string> names(ncontig);
    std::vector<int64_t> lens(ncontig);
    auto parallel_for = [&](size_t n, size_t grain, const std::function<void(size_t, size_t)>& fn) {
        std::atomic<size_t> next(0);
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++)
            th.emplace_back([&] {
                for (;;) {
                    const size_t a = next.fetch_add(grain);
                    if (a >= n) break;
                    fn(a, std::min(n, a + grain));
                }
            });
        for (auto& t : th) t.join();
    };
    // ---- FASTA + fai (60 bases per line; every contig's text is laid out in parallel, one write per contig)
    {
        FILE* fa = fopen((prep + "/portcullis.genome.fa").c_str(), "wb");
        FILE* fai = fopen((prep + "/portcullis.genome.fa.fai").c_str(), "wb");
        if (!fa || !fai) {
            fprintf(stderr, "cannot write into %s\n", prep.c_str());
            return 2;
        }
        int64_t off = 0;
        for (int c = 0; c < ncontig; c++) {
            const std::string d = argv[3 + c];
            auto nm = slurp<char>(d + "/name.txt");
            names[c] = std::string(nm.begin(), nm.end());
            while (!names[c].empty() && isspace((unsigned char)names[c].back())) names[c].pop_back();
            auto g = slurp<uint8_t>(d + "/genome.u8");
            lens[c] = (int64_t)g.size();
            off += fprintf(fa, ">%s\n", names[c].c_str());
            fprintf(fai, "%s\t%lld\t%lld\t60\t61\n", names[c].c_str(), (long long)g.size(), (long long)off);
            const size_t nlines = (g.size() + 59) / 60;
            std::vector<char> text(g.size() + nlines);
            parallel_for(nlines, 1 << 16, [&](size_t l0, size_t l1) {
                for (size_t l = l0; l < l1; l++) {
                    const size_t i = l * 60, n = std::min<size_t>(60, g.size() - i);
                    memcpy(&text[l * 61], &g[i], n);
                    text[l * 61 + n] = '\n';
                }
            });
            fwrite(text.data(), 1, text.size(), fa);
            off += (int64_t)text.size();
        }
        fclose(fa);
        fclose(fai);
    }
    // ---- uncompressed BAM stream: record sizes first (so every record knows its offset), then all records are
    // laid out in parallel.  Bases of unspliced reads and all qualities come from a generator seeded by the
    // record's ordinal, so the file does not depend on the thread count.
    std::vector<uint8_t> u;
    {
        std::string text = "@HD\tVN:1.4\tSO:coordinate\n";
        for (int c = 0; c < ncontig; c++) text += "@SQ\tSN:" + names[c] + "\tLN:" + std::to_string(lens[c]) + "\n";
        u.insert(u.end(), {'B', 'A', 'M', 1});
        put32(u, (uint32_t)text.size());
        u.insert(u.end(), text.begin(), text.end());
        put32(u, (uint32_t)ncontig);
        for (int c = 0; c < ncontig; c++) {
            put32(u, (uint32_t)names[c].size() + 1);
            u.insert(u.end(), names[c].begin(), names[c].end());
            u.push_back(0);
            put32(u, (uint32_t)lens[c]);
        }
    }
    struct Contig {
        std::vector<int32_t> pos, lq, mtid, mpos;
        std::vector<uint16_t> flag;
        std::vector<uint8_t> mapq, xs, seq4;
        std::vector<uint32_t> cig_off, cigar, seq_off;
        size_t first = 0;  // index of its first record in `recs`
    };
    std::vector<Contig> cs(ncontig);
    size_t total = 0;
    for (int c = 0; c < ncontig; c++) {
        const std::string d = argv[3 + c];
        Contig& k = cs[c];
        k.first = total;
        k.pos = slurp<int32_t>(d + "/pos.i32", true);
        if (k.pos.empty()) continue;
        k.flag = slurp<uint16_t>(d + "/flag.u16");
        k.mapq = slurp<uint8_t>(d + "/mapq.u8");
        k.xs = slurp<uint8_t>(d + "/xs.u8");
        k.lq = slurp<int32_t>(d + "/l_qseq.i32");
        k.mtid = slurp<int32_t>(d + "/mtid.i32");
        k.mpos = slurp<int32_t>(d + "/mpos.i32");
        k.cig_off = slurp<uint32_t>(d + "/cig_off.u32");
        k.cigar = slurp<uint32_t>(d + "/cigar.u32");
        k.seq_off = slurp<uint32_t>(d + "/seq_off.u32");
        k.seq4 = slurp<uint8_t>(d + "/seq4.u8");
        total += k.pos.size();
    }
    std::vector<Rec> recs(total);
    const int NAME_LEN = 12;  // "s%010llu" + NUL
    for (int c = 0; c < ncontig; c++) {
        Contig& k = cs[c];
        parallel_for(k.pos.size(), 1 << 16, [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; i++) {
                const uint32_t c0 = k.cig_off[i], c1 = k.cig_off[i + 1];
                int64_t span = 0;
                for (uint32_t q = c0; q < c1; q++) {
                    const uint32_t op = k.cigar[q] & 15u;
                    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += k.cigar[q] >> 4;
                }
                const int32_t l = k.lq[i];
                const bool has_xs = k.xs[i] == 1 || k.xs[i] == 2;
                const uint64_t bs = 32 + NAME_LEN + 4ull * (c1 - c0) + (uint64_t)(l + 1) / 2 + (uint64_t)l + (has_xs ? 4 : 0) + 4;
                recs[k.first + i] = {c, k.pos[i], (int32_t)(k.pos[i] + (span > 0 ? span : 1)), bs + 4};  // size for now
            }
        });
    }
    {
        uint64_t at = u.size();
        for (auto& r : recs) {
            const uint64_t sz = r.ustart;
            r.ustart = at;
            at += sz;
        }
        u.resize(at);
    }
    for (int c = 0; c < ncontig; c++) {
        Contig& k = cs[c];
        parallel_for(k.pos.size(), 1 << 15, [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; i++) {
                const size_t ordinal = k.first + i;
                const Rec& r = recs[ordinal];
                uint8_t* w = &u[r.ustart];
                const uint64_t next = ordinal + 1 < recs.size() ? recs[ordinal + 1].ustart : u.size();
                auto w32 = [&](uint32_t v) { memcpy(w, &v, 4); w += 4; };
                auto w16 = [&](uint16_t v) { memcpy(w, &v, 2); w += 2; };
                const uint32_t c0 = k.cig_off[i], c1 = k.cig_off[i + 1];
                const int32_t l = k.lq[i];
                const size_t sb = (size_t)(l + 1) / 2;
                const bool has_xs = k.xs[i] == 1 || k.xs[i] == 2;
                w32((uint32_t)(next - r.ustart - 4));
                w32((uint32_t)c);
                w32((uint32_t)k.pos[i]);
                *w++ = (uint8_t)NAME_LEN;
                *w++ = k.mapq[i];
                w16((uint16_t)reg2bin(r.pos, r.end));
                w16((uint16_t)(c1 - c0));
                w16(k.flag[i]);
                w32((uint32_t)l);
                w32((uint32_t)k.mtid[i]);
                w32((uint32_t)k.mpos[i]);
                w32(0);
                char name[16];
                snprintf(name, sizeof name, "s%010llu", (unsigned long long)ordinal);
                memcpy(w, name, NAME_LEN);
                w += NAME_LEN;
                memcpy(w, &k.cigar[c0], 4ull * (c1 - c0));
                w += 4ull * (c1 - c0);
                uint32_t lcg = (uint32_t)(ordinal * 2654435761ull) ^ 12345u;
                const uint32_t words = k.seq_off[i + 1] - k.seq_off[i];
                if ((size_t)words * 4 >= sb && sb) {
                    memcpy(w, &k.seq4[(size_t)k.seq_off[i] * 4], sb);
                    w += sb;
                } else
                    for (size_t q = 0; q < sb; q++) {  // unspliced reads carry no bases in the SoA: synthesise some
                        lcg = lcg * 1664525u + 1013904223u;
                        const uint8_t a = (uint8_t)(1u << ((lcg >> 24) & 3)), b = (uint8_t)(1u << ((lcg >> 26) & 3));
                        *w++ = (uint8_t)((a << 4) | b);
                    }
                for (int32_t q = 0; q < l; q++) {  // qualities with realistic entropy
                    lcg = lcg * 1664525u + 1013904223u;
                    *w++ = (uint8_t)(2 + ((lcg >> 20) % 39));
                }
                if (has_xs) {
                    *w++ = 'X'; *w++ = 'S'; *w++ = 'A';
                    *w++ = k.xs[i] == 1 ? '+' : '-';
                }
                *w++ = 'N'; *w++ = 'H'; *w++ = 'C'; *w++ = 1;
                if ((uint64_t)(w - &u[0]) != next) {
                    fprintf(stderr, "soa2bam: record %zu laid out wrong\n", ordinal);
                    abort();
                }
            }
        });
        k = Contig();  // the SoA arrays of this contig are no longer needed
    }
    // ---- BGZF, compressed in parallel
    const size_t BLK = 0xff00;
    const size_t nblk = (u.size() + BLK - 1) / BLK;
    std::vector<std::vector<uint8_t>> cblk(nblk);
    std::atomic<size_t> next(0);
    auto work = [&]() {
        std::vector<uint8_t> out(70000);
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= nblk) break;
            const size_t off = b * BLK, len = std::min(BLK, u.size() - off);
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            static const int level = getenv("SOA2BAM_LEVEL") ? atoi(getenv("SOA2BAM_LEVEL")) : 1;  // samtools writes level 6
            deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
            zs.next_in = &u[off];
            zs.avail_in = (uInt)len;
            zs.next_out = out.data();
            zs.avail_out = (uInt)out.size();
            deflate(&zs, Z_FINISH);
            const size_t clen = out.size() - zs.avail_out;
            deflateEnd(&zs);
            std::vector<uint8_t>& o = cblk[b];
            const uint8_t hdr[12] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0};
            o.insert(o.end(), hdr, hdr + 12);
            o.insert(o.end(), {'B', 'C', 2, 0});
            put16(o, (uint16_t)(clen + 25));
            o.insert(o.end(), out.begin(), out.begin() + (long)clen);
            put32(o, (uint32_t)crc32(crc32(0L, Z_NULL, 0), &u[off], (uInt)len));
            put32(o, (uint32_t)len);
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
    std::vector<uint64_t> coff(nblk + 1, 0);
    for (size_t b = 0; b < nblk; b++) coff[b + 1] = coff[b] + cblk[b].size();
    {
        FILE* f = fopen((prep + "/portcullis.sorted.alignments.bam").c_str(), "wb");
        for (size_t b = 0; b < nblk; b++) fwrite(cblk[b].data(), 1, cblk[b].size(), f);
        static const uint8_t eof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        fwrite(eof, 1, 28, f);
        fclose(f);
    }
    // ---- BAI
    auto voff = [&](uint64_t us) -> uint64_t {
        if (us >= u.size()) return coff[nblk] << 16;
        const size_t b = us / BLK;
        return (coff[b] << 16) | (us - b * BLK);
    };
    std::vector<std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>>> bins(ncontig);
    std::vector<std::vector<uint64_t>> lin(ncontig);
    std::vector<std::pair<uint64_t, uint64_t>>* cur = nullptr;
    int32_t cur_tid = -1;
    uint32_t cur_bin = 0;
    for (size_t i = 0; i < recs.size(); i++) {
        const Rec& r = recs[i];
        const uint64_t vs = voff(r.ustart), ve = voff(i + 1 < recs.size() ? recs[i + 1].ustart : u.size());
        const uint32_t bin = (uint32_t)reg2bin(r.pos, r.end);
        if (r.tid != cur_tid || bin != cur_bin) {
            cur = &bins[r.tid][bin];
            cur_tid = r.tid;
            cur_bin = bin;
        }
        auto& ch = *cur;
        if (!ch.empty() && ch.back().second == vs) ch.back().second = ve;
        else ch.push_back({vs, ve});
        const size_t w0 = (size_t)(r.pos >> 14), w1 = (size_t)((r.end - 1) >> 14);
        if (lin[r.tid].size() <= w1) lin[r.tid].resize(w1 + 1, 0);
        for (size_t w = w0; w <= w1; w++)
            if (lin[r.tid][w] == 0) lin[r.tid][w] = vs;
    }
    {
        FILE* f = fopen((prep + "/portcullis.sorted.alignments.bam.bai").c_str(), "wb");
        std::vector<uint8_t> o = {'B', 'A', 'I', 1};
        put32(o, (uint32_t)ncontig);
        for (int c = 0; c < ncontig; c++) {
            put32(o, (uint32_t)bins[c].size());
            for (auto& kv : bins[c]) {
                put32(o, kv.first);
                put32(o, (uint32_t)kv.second.size());
                for (auto& ch : kv.second) {
                    for (int k = 0; k < 8; k++) o.push_back((uint8_t)(ch.first >> (8 * k)));
                    for (int k = 0; k < 8; k++) o.push_back((uint8_t)(ch.second >> (8 * k)));
                }
            }
            put32(o, (uint32_t)lin[c].size());
            uint64_t last = 0;
            for (uint64_t v : lin[c]) {
                if (v) last = v;
                for (int k = 0; k < 8; k++) o.push_back((uint8_t)(last >> (8 * k)));
            }
        }
        fwrite(o.data(), 1, o.size(), f);
        fclose(f);
    }
    fprintf(stderr, "soa2bam: %zu records, %.1f MB uncompressed, %.1f MB BAM\n", recs.size(), u.size() / 1e6, coff[nblk] / 1e6);
    return 0;
}
