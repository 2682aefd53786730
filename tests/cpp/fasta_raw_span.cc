// Host side of pjb_upload_contig_fasta: GenomeMapper::rawSpan / readRaw on a FASTA file with its .fai.
// For every record of the index: "name offset bytes lineBases lineWidth length", and two files per record in <outdir>:
// <name>.raw (the bytes readRaw returns) and <name>.seq (fetchContig's bases), for the Python test to compare.
#include <portcullis/bam/genome_mapper.hpp>

#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    portcullis::bam::GenomeMapper g(argv[1]);
    g.buildFastaIndex();
    g.loadFastaIndex();
    const std::string out = argv[2];
    for (int k = 3; k < argc; k++) {
        const std::string name = argv[k];
        portcullis::bam::GenomeMapper::RawSpan s;
        if (!g.rawSpan(name, s)) {
            printf("%s none\n", name.c_str());
            continue;
        }
        std::vector<uint8_t> raw(s.bytes + 1, 0xEE);
        const bool whole = g.readRaw(s, raw.data(), 3);
        if (raw[s.bytes] != 0xEE) return 3;  // (wrote past the span)
        if (!whole) {
            printf("%s short\n", name.c_str());
            continue;
        }
        std::ofstream(out + "/" + name + ".raw", std::ios::binary).write((const char*)raw.data(), (std::streamsize)s.bytes);
        const std::string seq = g.fetchContig(name);
        std::ofstream(out + "/" + name + ".seq", std::ios::binary).write(seq.data(), (std::streamsize)seq.size());
        printf("%s %llu %zu %d %d %lld\n", name.c_str(), (unsigned long long)s.fileOffset, s.bytes, s.lineBases, s.lineWidth, (long long)s.length);
    }
    return 0;
}
