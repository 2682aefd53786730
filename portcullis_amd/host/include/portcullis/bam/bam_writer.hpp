// BamWriter: BGZF + BAM writer for the stages that emit alignments (bamfilt; lib/src/bam_writer.cc of the reference:
// bgzf_open("w") + bam_hdr_write + bam_write1).  Records are appended as their raw BAM bytes; blocks of 0xff00
// uncompressed bytes are deflated by a pool of threads and written in order.  close() adds the BGZF EOF block and,
// where the reference shells out to `samtools index` (src/bam_filter.cc:236-244), writes the .bai itself.
#pragma once

#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "bam_master.hpp"
#include "phase_pool.hpp"

namespace portcullis {
namespace bam {

class BamWriter {
    struct RecInfo {
        int32_t tid, pos, end;
        uint64_t ustart;  // offset of the record in the uncompressed stream
        uint64_t vs = 0;  // its virtual file offset, once the block holding its first byte is written
        bool vsKnown = false;
    };
    std::string path;
    FILE* fp = nullptr;
    int threads = 1, level = 6;
    bool wantIndex = true;
    std::vector<uint8_t> pending;      // uncompressed bytes not yet flushed
    uint64_t uflushed = 0;             // uncompressed bytes already compressed and written
    uint64_t cwritten = 0;             // compressed bytes written
    std::vector<RecInfo> recs;         // records since the last flush
    size_t nTargets = 0;
    std::vector<std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>>> bins;
    std::vector<std::vector<uint64_t>> lin;
    bool havePrev = false;             // the previous record's end offset is still open
    int32_t prevTid = -1;
    uint32_t prevBin = 0;
    void flush(bool final);
    void indexRecord(const RecInfo& r, uint64_t vs, uint64_t ve);
    int32_t cacheTid = -1;             // indexRecord: the bin of the previous record (records come sorted: nearly always the same)
    uint32_t cacheBin = 0;
    std::vector<std::pair<uint64_t, uint64_t>>* cacheChunks = nullptr;
    PhasePool* pool = nullptr;         // writeRecords' workers also compress (else: threads started per flush)

public:
    static constexpr size_t BLOCK = 0xff00;
    BamWriter(const std::string& path, int threads = 1, int level = 6) : path(path), threads(threads < 1 ? 1 : threads), level(level) {}
    ~BamWriter();
    void setWriteIndex(bool on) { wantIndex = on; }
    const std::string& getPath() const { return path; }
    void open(const std::string& headerText, const std::vector<RefSeq>& targets);
    // one alignment record: the 4-byte block_size followed by block_size bytes, exactly as in the input file
    void write(const uint8_t* rec, size_t len);
    // Many records at once: the records at data + (*slices[s])[k], slice after slice, of which those are written whose code
    // (codes[flat index]; nullptr: every record) is non-zero -- or equals `only` when `only` is non-zero.  The bytes are
    // gathered and the blocks compressed by `workers`.
    void writeRecords(const uint8_t* data, const std::vector<const std::vector<uint64_t>*>& slices, const uint8_t* codes, uint8_t only,
                      PhasePool& workers);
    void close();
    bool isOpen() const { return fp != nullptr; }
};

}  // namespace bam
}  // namespace portcullis
