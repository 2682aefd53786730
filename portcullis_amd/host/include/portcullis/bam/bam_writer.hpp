// BamWriter: BGZF + BAM writer for the stages that emit alignments (bamfilt; lib/src/bam_writer.cc of the reference:
// bgzf_open("w") + bam_hdr_write + bam_write1).  Records are appended as their raw BAM bytes; blocks of 0xff00
// uncompressed bytes are deflated by a pool of threads and written in order.  close() adds the BGZF EOF block and,
// where the reference shells out to `samtools index` (src/bam_filter.cc:236-244), writes the .bai itself.
#pragma once

#include <cstdint>
#include <cstdio>
#include <functional>
#include <future>
#include <map>
#include <new>
#include <utility>
#include <string>
#include <vector>

#include "bam_master.hpp"
#include "phase_pool.hpp"

namespace portcullis {
namespace bam {

// The writer's large buffers (uncompressed records waiting for their blocks, compressed blocks) come from a pluggable
// allocator -- BamFilter plugs in page-locked memory (pjb_host_alloc) so that the device reads and writes them by DMA -- and
// are never filled with zeros when they grow.
struct BufferHooks {
    void* (*alloc)(size_t) = nullptr;  // nullptr: malloc / free
    void (*release)(void*) = nullptr;
};
void setBufferHooks(const BufferHooks& h);  // affects buffers allocated afterwards
void* hookedAlloc(size_t bytes);
void hookedFree(void* p);
template <class T>
struct HookAlloc {
    typedef T value_type;
    HookAlloc() = default;
    template <class U>
    HookAlloc(const HookAlloc<U>&) {}
    T* allocate(size_t n) { return static_cast<T*>(hookedAlloc(n * sizeof(T))); }
    void deallocate(T* p, size_t) { hookedFree(p); }
    template <class U>
    void construct(U*) noexcept {}  // (default-initialised: resize() does not touch the new bytes)
    template <class U, class A0, class... A>
    void construct(U* p, A0&& a0, A&&... a) {
        ::new ((void*)p) U(std::forward<A0>(a0), std::forward<A>(a)...);
    }
    template <class U>
    bool operator==(const HookAlloc<U>&) const { return true; }
    template <class U>
    bool operator!=(const HookAlloc<U>&) const { return false; }
};
typedef std::vector<uint8_t, HookAlloc<uint8_t>> ByteBuf;

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
    ByteBuf pending;                   // uncompressed bytes not yet flushed
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
    // compresses `n` bytes cut into blocks of `block` bytes (the last one shorter) into complete BGZF members, back to back in
    // `out`, their lengths in `sizes`; false: not available (zlib takes over)
    std::function<bool(const uint8_t* in, size_t n, size_t block, ByteBuf& out, std::vector<uint32_t>& sizes)> compressor;
    ByteBuf cout_[2];                  // the compressor's output: one is written to the file while the other is filled
    int coutCur = 0;
    std::future<void> writing;         // the fwrite of the last compressed piece
    void waitWrite() {
        if (writing.valid()) writing.get();  // (rethrows a failed write)
    }
    size_t flushBlocks = 64;
    bool asyncFlush = false;           // writeRecords returns while the blocks are compressed and written by another thread
    std::future<void> inflight;
    void waitFlush() {
        if (inflight.valid()) inflight.get();  // (rethrows what the flush threw)
    }
    // With asyncFlush, writeRecords gathers the next piece into `pending` while the flush thread works on `flushing` /
    // `flushRecs` (whole blocks only; records whose end is not known yet stay in flushRecs for the next flush).
    ByteBuf flushing;
    std::vector<RecInfo> flushRecs;
    uint64_t ubase = 0;                // offset of pending[0] in the uncompressed stream (== uflushed when no flush is in flight)
    void settle();                     // wait for the flush thread and take its leftover records back
    void flushBuf(ByteBuf& buf, std::vector<RecInfo>& rs, bool final);
    std::vector<uint32_t> csizes_;

public:
    static constexpr size_t BLOCK = 0xff00;
    BamWriter(const std::string& path, int threads = 1, int level = 6) : path(path), threads(threads < 1 ? 1 : threads), level(level) {}
    ~BamWriter();
    void setWriteIndex(bool on) { wantIndex = on; }
    // BGZF blocks compressed somewhere else (BamFilter: on the device, pjb_deflate_bgzf) instead of by zlib in this process
    void setBlockCompressor(std::function<bool(const uint8_t*, size_t, size_t, ByteBuf&, std::vector<uint32_t>&)> f) { compressor = std::move(f); }
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
    // With a block compressor set: writeRecords hands the gathered records to a thread of the writer's own and returns; the
    // next call (and close) waits for it.  The compressor then runs on that thread.
    void setAsyncFlush(bool on) { asyncFlush = on; }
    // writeRecords compresses once this many blocks are waiting (64: 4 MB; tests: a few, for many hand-overs in a small file)
    void setFlushBlocks(size_t n) { flushBlocks = n < 1 ? 1 : n; }
};

}  // namespace bam
}  // namespace portcullis
