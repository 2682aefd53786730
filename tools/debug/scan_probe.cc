// BamReader::scanRecordsParallel alone (the sink counts records): how fast do the pieces come on this box, with how many
// threads, pinned to which cores?   scan_probe <in.bam> <threads> [ahead]     (PORTCULLIS_PROFILE_PIECES=1: a line per piece)
#include <portcullis/bam/bam_reader.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    portcullis::bam::BamReader r(argv[1]);
    r.open();
    for (int rep = 0; rep < 3; rep++) {
        size_t n = 0, bytes = 0;
        const auto t0 = std::chrono::steady_clock::now();
        r.scanRecordsParallel(atoi(argv[2]), (size_t)256 << 20, [&](const portcullis::bam::BamReader::FileChunk& fc) {
            n += fc.records;
            bytes += fc.bytes;
        }, argc >= 4 ? atoi(argv[3]) : 0);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%s threads: %zu records, %.0f MB in %.3f s = %.1f GB/s\n", argv[2], n, bytes / 1e6, s, bytes / s / 1e9);
    }
    return 0;
}
