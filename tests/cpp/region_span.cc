// Host side of the device ingest: BamReader::regionSpan / readSpan / readRegionBytes on a prepared BAM.
// Prints, per target, "tid fileOff bytes firstU md5-less checksum" so that the Python test can compare with its own
// parse of the same file.
#include <portcullis/bam/bam_reader.hpp>

#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const bool csi = atoi(argv[2]) != 0;
    portcullis::bam::BamReader r(argv[1]);
    r.open(csi);
    auto refs = r.createRefList();
    for (size_t t = 0; t < refs->size(); t++) {
        uint64_t off = 0;
        size_t n = 0;
        uint32_t firstU = 0;
        if (!r.regionSpan((int32_t)t, off, n, firstU)) {
            printf("%zu none\n", t);
            continue;
        }
        size_t n2 = 0;
        uint32_t f2 = 0;
        uint8_t* a = r.readRegionBytes((int32_t)t, 3, n2, f2);
        uint8_t* b = (uint8_t*)portcullis::bam::bigAlloc(n + 64);
        r.readSpan(off, n, b, 1);
        unsigned long long sum = 1469598103934665603ull;  // FNV-1a over the bytes
        bool same = n2 == n && f2 == firstU;
        for (size_t i = 0; i < n; i++) {
            sum = (sum ^ b[i]) * 1099511628211ull;
            same = same && a[i] == b[i];
        }
        printf("%zu %llu %zu %u %llu %d\n", t, (unsigned long long)off, n, firstU, sum, same ? 1 : 0);
        portcullis::bam::bigFree(a);
        portcullis::bam::bigFree(b);
    }
    return 0;
}
