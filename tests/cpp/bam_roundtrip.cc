// Host BAM I/O under AddressSanitizer / UBSan (no GPU): BamReader's record-at-a-time view (next / current), its raw
// sequential walk (rewind / nextRecord), and BamWriter (parallel BGZF, in-process .bai) -- the file written here is
// read back through its own index and must hold the same records.
//   bam_roundtrip <in.bam> <out.bam> <threads> [bulk <chunk bytes> <drop every k-th record, 0: none> [pieces the scan runs ahead [async]]]
// async: the writer's hand-over route (a block compressor -- zlib here, the device in BamFilter -- on a thread of the writer's).
// bulk: the same through the many-records-at-once route of `bamfilt` (BamReader::scanRecordsParallel -> BamWriter::writeRecords).
#include <portcullis/bam/bam_reader.hpp>
#include <portcullis/bam/bam_writer.hpp>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
#include <zlib.h>
#include <cstdlib>
#include <string>
#include <iostream>

using namespace portcullis::bam;

struct Sum {
    unsigned long long n = 0, h = 1469598103934665603ull;
    void add(const BamAlignment& a) {
        n++;
        auto mix = [&](unsigned long long v) { h = (h ^ v) * 1099511628211ull; };
        mix((unsigned long long)a.getPosition());
        mix(a.getAlignmentFlag());
        mix((unsigned long long)a.getEnd());
        mix(a.getXsCode());
        for (const auto& op : a.getCigar()) mix(((unsigned long long)op.length << 8) | (unsigned char)op.type);
        for (char c : a.deriveName()) mix((unsigned char)c);
        for (char c : a.getQuerySeq()) mix((unsigned char)c);
    }
};

static Sum walk(const std::string& path) {
    BamReader r(path);
    r.open();
    Sum s;
    auto refs = r.createRefList();
    for (size_t t = 0; t < refs->size(); t++) {
        if (!r.hasAlignments((int32_t)t)) continue;
        r.setRegion((int32_t)t);
        while (r.next()) s.add(r.current());
    }
    return s;
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    try {
        const bool bulk = argc >= 7 && std::string(argv[4]) == "bulk";
        const Sum a = bulk ? Sum() : walk(argv[1]);  // (bulk: straight to the route under test, also for damaged files)
        unsigned long long raw = 0;
        {
            BamReader r(argv[1]);
            r.open();
            BamWriter w(argv[2], atoi(argv[3]));
            w.open(r.getHeaderText(), r.getTargets());
            if (argc >= 7 && std::string(argv[4]) == "bulk") {
                const int threads = atoi(argv[3]), drop = atoi(argv[6]);
                PhasePool workers(threads > 1 ? threads : 0);
                if (argc >= 9 && std::string(argv[8]) == "async") {
                    w.setBlockCompressor([](const uint8_t* in, size_t n, size_t block, ByteBuf& out, std::vector<uint32_t>& sizes) -> bool {
                        out.clear();
                        sizes.clear();
                        for (size_t off = 0; off < n; off += block) {
                            const size_t len = std::min(block, n - off);
                            std::vector<uint8_t> z(len + 1024);
                            z_stream zs;
                            memset(&zs, 0, sizeof zs);
                            if (deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
                            zs.next_in = const_cast<uint8_t*>(in + off);
                            zs.avail_in = (uInt)len;
                            zs.next_out = z.data();
                            zs.avail_out = (uInt)z.size();
                            const int rc = deflate(&zs, Z_FINISH);
                            const size_t clen = z.size() - zs.avail_out;
                            deflateEnd(&zs);
                            if (rc != Z_STREAM_END) return false;
                            const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
                            const size_t at = out.size();
                            out.resize(at + 18 + clen + 8);
                            memcpy(&out[at], hdr, 16);
                            out[at + 16] = (uint8_t)((clen + 25) & 0xff);
                            out[at + 17] = (uint8_t)((clen + 25) >> 8);
                            memcpy(&out[at + 18], z.data(), clen);
                            const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), in + off, (uInt)len), isz = (uint32_t)len;
                            memcpy(&out[at + 18 + clen], &crc, 4);
                            memcpy(&out[at + 22 + clen], &isz, 4);
                            sizes.push_back((uint32_t)(clen + 26));
                        }
                        return true;
                    });
                    w.setAsyncFlush(true);
                    w.setFlushBlocks(2);
                }
                std::vector<uint8_t> codes;
                r.scanRecordsParallel(threads, (size_t)atoll(argv[5]), [&](const BamReader::FileChunk& fc) {
                    codes.assign(fc.records, 1);
                    if (drop)
                        for (size_t i = 0; i < fc.records; i++)
                            if ((raw + i) % (size_t)drop == 0) codes[i] = 0;
                    raw += fc.records;
                    w.writeRecords(fc.data, fc.slices, codes.data(), 0, workers);
                }, argc >= 8 ? atoi(argv[7]) : 0);
                w.close();
                printf("bulk raw=%llu\n", raw);
                return 0;
            }
            std::vector<uint8_t> rec;
            r.rewind();
            while (r.nextRecord(rec)) {
                w.write(rec.data(), rec.size());
                raw++;
            }
            w.close();
        }
        const Sum b = walk(argv[2]);
        printf("placed=%llu raw=%llu hash_in=%llx hash_out=%llx\n", a.n, raw, a.h, b.h);
        return a.n == b.n && a.h == b.h && raw >= a.n ? 0 : 1;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 3;
    }
}
