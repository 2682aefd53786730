// BamReader: BGZF + BAM + BAI reader that transcodes alignment records straight into the
// fixed-width structure-of-arrays batches the device path consumes (pjb_batch), without building
// per-record objects.  Plays the role of lib/src/bam_reader.cc + BamAlignment::init
// (lib/src/bam_alignment.cc:71-100) of the reference; written from the SAM/BAM specification on
// top of zlib only.
#pragma once

#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "bam_alignment.hpp"
#include "bam_master.hpp"

struct pjb_batch;

namespace portcullis {
namespace bam {

// Large host arrays (decode buffers, batch arrays) come from 2 MiB-aligned blocks that ask for
// transparent huge pages: with 100+ decode threads faulting in fresh 4 KiB pages the kernel's
// address-space lock, not zlib, sets the pace.
void* bigAlloc(size_t bytes);
void bigFree(void* p);
template <class T>
struct BigAllocator {
    typedef T value_type;
    BigAllocator() = default;
    template <class U>
    BigAllocator(const BigAllocator<U>&) {}
    T* allocate(size_t n) { return static_cast<T*>(bigAlloc(n * sizeof(T))); }
    void deallocate(T* p, size_t) { bigFree(p); }
    template <class U>
    bool operator==(const BigAllocator<U>&) const { return true; }
    template <class U>
    bool operator!=(const BigAllocator<U>&) const { return false; }
};
template <class T>
using BatchVector = std::vector<T, BigAllocator<T>>;

// One batch of alignment records of one target sequence, in file order (layout of pjb_batch).
struct ReadBatch {
    BatchVector<int32_t> pos, l_qseq, mtid, mpos;
    BatchVector<uint16_t> flag;
    BatchVector<uint8_t> mapq, xs;
    BatchVector<uint32_t> cig_off, cigar, seq_off;
    BatchVector<uint8_t> seq4;
    BatchVector<uint64_t> name_hash;  // only filled by a reader with setNameHashes(true) (junc --extra)
    uint64_t n_refskip = 0;

    size_t size() const { return pos.size(); }
    void clear();
    void reserve(size_t n);
    void view(pjb_batch& out) const;  // fill a pjb_batch with pointers into this object
};

// Sequential reader over the inflated byte stream of a BGZF file with virtual-offset seeking.
class BgzfStream {
    FILE* fp = nullptr;
    std::vector<uint8_t> comp, block;
    size_t block_pos = 0;
    uint64_t block_coffset = 0, next_coffset = 0;
    bool at_eof = false;
    bool loadBlock();

public:
    ~BgzfStream() { close(); }
    void open(const std::string& path);
    void close();
    bool isOpen() const { return fp != nullptr; }
    void seek(uint64_t voffset);
    uint64_t tell() const { return block_pos >= block.size() ? next_coffset << 16 : (block_coffset << 16) | (uint64_t)block_pos; }
    size_t read(void* dst, size_t n);  // returns bytes read (< n only at end of file)
};

class BamReader {
    std::string bamFile;
    BgzfStream bgzf;
    std::string headerText;
    std::vector<RefSeq> targets;
    std::vector<uint64_t> firstOffset;  // per target: virtual offset of its first record, ~0 = none
    std::vector<uint64_t> lastOffset;   // per target: largest chunk end in the index (just past its last record)
    std::vector<std::vector<uint64_t>> restart;  // per target: sorted virtual offsets the index names (all are record starts)
    bool indexLoaded = false;
    std::function<bool(const uint8_t*, size_t, uint8_t*, size_t)> blockInflater;
    int32_t regionTid = -1;
    int32_t regionLen = 0;
    bool regionDone = true;
    bool wantNames = false;
    uint64_t firstRecordVoffset = 0;
    std::vector<uint8_t> rec;
    BamAlignment cur;      // next() / current(): the record-at-a-time view of the reference's reader
    ReadBatch one;

    void loadIndex(bool useCsi);

public:
    explicit BamReader(const std::string& path) : bamFile(path) {}

    void open(bool useCsi = false);
    void close() { bgzf.close(); }
    // also transcode std::hash(deriveName()) of every record (pjb_batch.name_hash; junc --extra)
    void setNameHashes(bool on) { wantNames = on; }

    std::shared_ptr<RefSeqPtrList> createRefList() const;
    std::shared_ptr<RefSeqPtrIndexMap> createRefMap(const RefSeqPtrList& refs) const;
    std::string bamDetails() const;
    const std::string& getHeaderText() const { return headerText; }

    // true if the index lists at least one alignment on the target
    bool hasAlignments(int32_t tid) const { return firstOffset[(size_t)tid] != ~0ull; }

    // Visit every alignment placed on `tid` with pos < length(tid), like
    // BamReader::setRegion(tid, 0, len) + next() of the reference (src/junction_builder.cc:321-322):
    // no flag is filtered, unplaced reads are never seen.
    void setRegion(int32_t tid);
    // Append up to maxRecords records of the region to `out`; false when the region is exhausted
    // and nothing was appended.
    bool nextBatch(ReadBatch& out, size_t maxRecords);
    // Record at a time, as lib/src/bam_reader.cc:134-142 of the reference: next() advances inside the region set by
    // setRegion(tid), current() is a reference to an internal object that the next call overwrites.
    bool next();
    const BamAlignment& current() const { return cur; }
    // Every record of the file in file order, unplaced ones included (the reader loop of BamFilter::filter,
    // src/bam_filter.cc:190): rewind() goes back to the first record, nextRecord() hands out the raw record (4-byte
    // block_size + body).  false at the end of the file.
    void rewind();
    bool nextRecord(std::vector<uint8_t>& rec);
    const std::vector<RefSeq>& getTargets() const { return targets; }

    // Same visit as setRegion(tid) + nextBatch(...) but with `nthreads` workers inside the target:
    // BGZF blocks are located from their headers, inflated in parallel into a contiguous buffer,
    // and the records are transcoded in parallel into batches of at most maxRecords alignments,
    // which are handed to `sink` in file order.  (The reference can only use one thread per target
    // sequence, src/junction_builder.cc:109-112; this is SURVEY row f1, host ingest.)
    void decodeRegionParallel(int32_t tid, int nthreads, size_t maxRecords, const std::function<void(ReadBatch&)>& sink);

    // Every record of the file in file order (unplaced ones included) for callers that keep records as byte spans
    // (BamFilter): the file is taken in pieces of about `chunkBytes` inflated bytes -- blocks inflated by `nthreads`
    // workers, record starts found by all of them at once (each walks the block_size chain from a record start the
    // index names to the next one; the unplaced tail of a file, which no index covers, is one walk) -- and `sink` gets
    // one FileChunk per piece: whole records only, `slices` = their offsets in `data`, slice after slice in file order.
    // The chunk's memory is the reader's: it is overwritten by the next piece.
    struct FileChunk {
        const uint8_t* data = nullptr;
        size_t bytes = 0;
        std::vector<const std::vector<uint64_t>*> slices;  // record offsets (each: 4-byte block_size + body)
        size_t records = 0;
    };
    // ahead > 0: the scan runs on a thread of its own (with the workers), up to `ahead` pieces before the one `sink` -- still
    // on the calling thread, one piece at a time, in file order -- is looking at; a piece stays valid until `sink` returns.
    void scanRecordsParallel(int nthreads, size_t chunkBytes, const std::function<void(const FileChunk&)>& sink, int ahead = 0);
    // scanRecordsParallel's blocks inflated somewhere else (BamFilter: on the device, pjb_inflate_bgzf) instead of by zlib on
    // the workers: `comp` holds `n` bytes of whole consecutive BGZF blocks whose inflated bytes (`outBytes` of them, the sum
    // of their ISIZE fields) go to `out`.  false: not available right now (zlib takes this piece); an error is thrown.
    // With it set, the scan's two large buffers come from the writer's buffer hooks (page-locked memory: the device reads
    // and writes them by DMA).
    void setBlockInflater(std::function<bool(const uint8_t* comp, size_t n, uint8_t* out, size_t outBytes)> f) { blockInflater = std::move(f); }

    // The file bytes that hold target `tid`'s records, untouched (whole BGZF blocks: from the block with its
    // first record through the block in which the next target starts, or the end of the file), read with
    // `nthreads` parallel preads into one bigAlloc buffer the caller frees with bigFree.  firstU = offset of
    // the first record inside the first block's inflated bytes.  Feeds pjb_submit_bam (device-side ingest).
    // Returns nullptr if the target has no records.
    uint8_t* readRegionBytes(int32_t tid, int nthreads, size_t& bytes, uint32_t& firstU);
    // The same in two steps, for callers that bring their own (e.g. page-locked) buffer: where the bytes are ...
    bool regionSpan(int32_t tid, uint64_t& fileOff, size_t& bytes, uint32_t& firstU);
    // ... and `bytes` bytes from `fileOff` into dst with `nthreads` parallel preads.
    void readSpan(uint64_t fileOff, size_t bytes, uint8_t* dst, int nthreads);
    // A read-only mapping of the whole file (made once per process and kept; nullptr if it cannot be had): callers that
    // page-lock pieces of it hand the page cache itself to the device.
    static const uint8_t* mapFile(const std::string& path, size_t& bytes);
};

}  // namespace bam
}  // namespace portcullis
