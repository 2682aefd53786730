// GenomeMapper: indexed FASTA access (role of lib/src/genome_mapper.cc + htslib faidx in the
// reference).  fetchBases keeps faidx_fetch_seq's clamping rules (deps/htslib-1.3/faidx.c:439-476);
// fetchContig returns a whole target sequence for upload to the GPU.
#pragma once

#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "bam_master.hpp"

namespace portcullis {
namespace bam {

class GenomeMapper {
    struct Entry {
        std::string name;
        int64_t len = 0, offset = 0;
        int32_t line_blen = 0, line_len = 0;
    };
    std::string genomeFile;
    std::vector<Entry> entries;
    std::unordered_map<std::string, size_t> byName;
    FILE* fp = nullptr;

    std::string readSpan(const Entry& e, int64_t beg, int64_t count) const;

public:
    explicit GenomeMapper(const std::string& path) : genomeFile(path) {}
    ~GenomeMapper();
    GenomeMapper(const GenomeMapper&) = delete;
    GenomeMapper& operator=(const GenomeMapper&) = delete;

    std::string getFastaIndexFile() const { return genomeFile + ".fai"; }
    void buildFastaIndex();  // writes <genome>.fai
    void loadFastaIndex();   // reads <genome>.fai and opens the FASTA
    int getNbSeqs() const { return (int)entries.size(); }
    bool hasSeq(const std::string& name) const { return byName.count(name) != 0; }
    int64_t getSeqLength(const std::string& name) const;

    // 0-based inclusive; end < beg fetches one base; both ends are clamped to the sequence
    std::string fetchBases(const char* name, int start, int end) const;
    // every base of the sequence (graphic characters only, case preserved)
    std::string fetchContig(const std::string& name) const;

    // where the sequence lines of a record lie in the file and how they are laid out (the .fai's columns); false if the
    // record is unknown or has no line geometry
    struct RawSpan {
        uint64_t fileOffset = 0;
        size_t bytes = 0;  // from the first base to the last (line terminators in between included)
        int32_t lineBases = 0, lineWidth = 0;
        int64_t length = 0;
    };
    bool rawSpan(const std::string& name, RawSpan& out) const;
    // the bytes of a raw span, read with a few threads (pread on the open file); false if the file ends before the span
    // does (a record that is not laid out as its index line says: the caller filters the characters instead)
    bool readRaw(const RawSpan& span, uint8_t* dst, int nthreads) const;
};

}  // namespace bam
}  // namespace portcullis
