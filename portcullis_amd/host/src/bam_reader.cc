#include <portcullis/bam/bam_reader.hpp>
#include <portcullis/bam/phase_pool.hpp>
#include <portcullis/bam/bam_writer.hpp>
#include <portcullis/bam/fast_inflate.hpp>
#include <portcullis/bam/name_hash.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <sys/resource.h>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <cmath>
#include <cstring>
#include <fcntl.h>
#include <sstream>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <cstdlib>
#include <new>
#include <zlib.h>

#include "../../../include/portcullis_amd.h"

namespace portcullis {
namespace bam {

// blocks go through fastInflate first (fast_inflate.hpp); PORTCULLIS_ZLIB_INFLATE=1: zlib only
static const bool g_fastInflate = getenv("PORTCULLIS_ZLIB_INFLATE") == nullptr;

// ------------------------------------------------------------------ big allocations
void* bigAlloc(size_t bytes) {
    void* p = nullptr;
    if (bytes >= (1u << 20)) {
        const size_t rounded = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
        if (posix_memalign(&p, 2u << 20, rounded) != 0) throw std::bad_alloc();
        (void)madvise(p, rounded, MADV_HUGEPAGE);
    } else {
        p = malloc(bytes ? bytes : 1);
        if (!p) throw std::bad_alloc();
    }
    return p;
}

void bigFree(void* p) { free(p); }

// ------------------------------------------------------------------ ReadBatch
void ReadBatch::clear() {
    pos.clear(); l_qseq.clear(); mtid.clear(); mpos.clear(); flag.clear(); mapq.clear(); xs.clear();
    cig_off.assign(1, 0);
    cigar.clear();
    seq_off.assign(1, 0);
    seq4.clear();
    name_hash.clear();
    n_refskip = 0;
}

void ReadBatch::reserve(size_t n) {
    pos.reserve(n); l_qseq.reserve(n); mtid.reserve(n); mpos.reserve(n); flag.reserve(n); mapq.reserve(n); xs.reserve(n);
    cig_off.reserve(n + 1);
    seq_off.reserve(n + 1);
    cigar.reserve(n * 2);
}

void ReadBatch::view(pjb_batch& b) const {
    b.n_reads = (int64_t)pos.size();
    b.pos = pos.data(); b.flag = flag.data(); b.mapq = mapq.data(); b.xs = xs.data(); b.l_qseq = l_qseq.data();
    b.mtid = mtid.data(); b.mpos = mpos.data(); b.cig_off = cig_off.data(); b.cigar = cigar.data();
    b.seq_off = seq_off.data(); b.seq4 = seq4.data();
    b.name_hash = name_hash.size() == pos.size() && !pos.empty() ? name_hash.data() : nullptr;
    b.seq2 = nullptr; // (the host decoder hands over BAM's 4-bit bases only: the device ingest, which is the default, writes both)
    b.seq_exc = nullptr;
}

// ------------------------------------------------------------------ BGZF
static inline uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

void BgzfStream::open(const std::string& path) {
    close();
    fp = fopen(path.c_str(), "rb");
    if (!fp) throw BamException("Could not open BAM file: " + path);
    setvbuf(fp, nullptr, _IOFBF, 1 << 20);
    block.clear();
    block_pos = 0;
    block_coffset = next_coffset = 0;
    at_eof = false;
}

void BgzfStream::close() {
    if (fp) fclose(fp);
    fp = nullptr;
}

bool BgzfStream::loadBlock() {
    uint8_t hdr[18];
    block_coffset = next_coffset;
    size_t got = fread(hdr, 1, 18, fp);
    if (got == 0) {
        at_eof = true;
        block.clear();
        block_pos = 0;
        return false;
    }
    if (got != 18 || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4))
        throw BamException("Invalid BGZF block header");
    uint16_t xlen = le16(hdr + 10);
    // find the BC subfield (normally the first and only one)
    std::vector<uint8_t> extra(xlen);
    memcpy(extra.data(), hdr + 12, std::min<size_t>(6, xlen));
    if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, fp) != (size_t)(xlen - 6)) throw BamException("Truncated BGZF block");
    int bsize = -1;
    for (size_t o = 0; o + 4 <= extra.size();) {
        uint16_t slen = le16(&extra[o + 2]);
        if (extra[o] == 'B' && extra[o + 1] == 'C' && slen == 2 && o + 6 <= extra.size()) bsize = le16(&extra[o + 4]);
        o += 4 + slen;
    }
    if (bsize < 0) throw BamException("BGZF block without BC field");
    const size_t total = (size_t)bsize + 1;
    if (total < (size_t)xlen + 20) throw BamException("Invalid BGZF block: BSIZE is smaller than the block's header and footer");
    const size_t cdata = total - 12 - xlen - 8;
    comp.resize(cdata + 8);
    if (fread(comp.data(), 1, cdata + 8, fp) != cdata + 8) throw BamException("Truncated BGZF block");
    const uint32_t isize = le32(&comp[cdata + 4]);
    if (isize > 65536) throw BamException("Invalid BGZF block: ISIZE exceeds 64 KiB");
    block.resize(isize);
    if (isize && !(g_fastInflate && fastInflate(comp.data(), cdata, block.data(), isize))) {  // (declined: zlib has the last word)
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) throw BamException("inflateInit2 failed");
        zs.next_in = comp.data();
        zs.avail_in = (uInt)cdata;
        zs.next_out = block.data();
        zs.avail_out = isize;
        int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || zs.avail_out != 0) throw BamException("BGZF inflate failed");
    }
    next_coffset = block_coffset + total;
    block_pos = 0;
    return true;
}

void BgzfStream::seek(uint64_t voffset) {
    const uint64_t co = voffset >> 16;
    const size_t uo = (size_t)(voffset & 0xffff);
    if (fseeko(fp, (off_t)co, SEEK_SET) != 0) throw BamException("BGZF seek failed");
    next_coffset = co;
    at_eof = false;
    if (!loadBlock()) return;
    if (uo > block.size()) throw BamException("BGZF virtual offset beyond block");
    block_pos = uo;
}

size_t BgzfStream::read(void* dst, size_t n) {
    uint8_t* d = (uint8_t*)dst;
    size_t done = 0;
    while (done < n) {
        if (block_pos >= block.size()) {
            if (at_eof || !loadBlock()) break;
            continue;
        }
        const size_t take = std::min(n - done, block.size() - block_pos);
        memcpy(d + done, block.data() + block_pos, take);
        block_pos += take;
        done += take;
    }
    return done;
}

// ------------------------------------------------------------------ BAM
void BamReader::open(bool useCsi) {
    bgzf.open(bamFile);
    uint8_t b4[4];
    if (bgzf.read(b4, 4) != 4 || memcmp(b4, "BAM\1", 4) != 0) throw BamException("Not a BAM file: " + bamFile);
    if (bgzf.read(b4, 4) != 4) throw BamException("Truncated BAM header");
    const uint32_t l_text = le32(b4);
    headerText.resize(l_text);
    if (l_text && bgzf.read(&headerText[0], l_text) != l_text) throw BamException("Truncated BAM header");
    if (bgzf.read(b4, 4) != 4) throw BamException("Truncated BAM header");
    const uint32_t n_ref = le32(b4);
    targets.clear();
    for (uint32_t i = 0; i < n_ref; i++) {
        if (bgzf.read(b4, 4) != 4) throw BamException("Truncated BAM header");
        const uint32_t l_name = le32(b4);
        std::string name(l_name, 0);
        if (bgzf.read(&name[0], l_name) != l_name || bgzf.read(b4, 4) != 4) throw BamException("Truncated BAM header");
        if (!name.empty() && name.back() == 0) name.pop_back();
        targets.emplace_back((int32_t)i, name, (int32_t)le32(b4));
    }
    firstRecordVoffset = bgzf.tell();
    loadIndex(useCsi);
}

void BamReader::rewind() {
    bgzf.seek(firstRecordVoffset);
    regionDone = true;
}

bool BamReader::nextRecord(std::vector<uint8_t>& out) {
    uint8_t b4[4];
    const size_t got = bgzf.read(b4, 4);
    if (got == 0) return false;
    if (got != 4) throw BamException("Truncated BAM record");
    const uint32_t bs = le32(b4);
    if (bs < 32) throw BamException("Invalid BAM record");
    out.resize(4 + (size_t)bs);
    memcpy(out.data(), b4, 4);
    if (bgzf.read(out.data() + 4, bs) != bs) throw BamException("Truncated BAM record");
    return true;
}

// BAI (SAM spec section 5.2) or CSI (htslib CSIv1: BGZF-compressed, bins carry loffset, no linear
// index).  Only two things are kept per target: the smallest chunk start (where its records begin)
// and every virtual offset the index names, all of which are record starts (decodeRegionParallel).
void BamReader::loadIndex(bool useCsi) {
    const std::string path = bamFile + (useCsi ? ".csi" : ".bai");
    std::vector<uint8_t> buf;
    {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) throw BamException("Could not open BAM index: " + path);
        fseeko(f, 0, SEEK_END);
        const off_t sz = ftello(f);
        fseeko(f, 0, SEEK_SET);
        buf.resize((size_t)sz);
        const bool ok = !sz || fread(buf.data(), 1, (size_t)sz, f) == (size_t)sz;
        fclose(f);
        if (!ok) throw BamException("Could not read BAM index: " + path);
    }
    if (buf.size() >= 2 && buf[0] == 31 && buf[1] == 139) {  // BGZF container (CSI files are compressed)
        BgzfStream z;
        z.open(path);
        std::vector<uint8_t> out;
        uint8_t tmp[65536];
        for (;;) {
            const size_t n = z.read(tmp, sizeof tmp);
            out.insert(out.end(), tmp, tmp + n);
            if (n < sizeof tmp) break;
        }
        buf.swap(out);
    }
    const bool csi = buf.size() >= 4 && memcmp(buf.data(), "CSI\1", 4) == 0;
    const bool bai = buf.size() >= 4 && memcmp(buf.data(), "BAI\1", 4) == 0;
    if (!csi && !bai) throw BamException("Not a BAI or CSI index: " + path);
    size_t o = 4;
    uint32_t pseudoBin = 37450;
    if (csi) {
        if (o + 12 > buf.size()) throw BamException("Truncated CSI index");
        const int32_t depth = (int32_t)le32(&buf[o + 4]);
        const uint32_t l_aux = le32(&buf[o + 8]);
        o += 12 + l_aux;
        pseudoBin = (uint32_t)(((1ull << (depth * 3 + 3)) - 1) / 7 + 1);
    }
    if (o + 4 > buf.size()) throw BamException("Truncated BAM index");
    const uint32_t n_ref = le32(&buf[o]);
    o += 4;
    firstOffset.assign(targets.size(), ~0ull);
    lastOffset.assign(targets.size(), 0ull);
    restart.assign(targets.size(), std::vector<uint64_t>());
    for (uint32_t r = 0; r < n_ref; r++) {
        std::vector<uint64_t> pts;
        if (o + 4 > buf.size()) throw BamException("Truncated BAM index");
        const uint32_t n_bin = le32(&buf[o]);
        o += 4;
        uint64_t first = ~0ull, last = 0;
        for (uint32_t b = 0; b < n_bin; b++) {
            if (o + (csi ? 16 : 8) > buf.size()) throw BamException("Truncated BAM index");
            const uint32_t bin = le32(&buf[o]);
            o += 4;
            if (csi) o += 8;  // loffset: the start of the first record overlapping the bin, not of a record IN it
            const uint32_t n_chunk = le32(&buf[o]);
            o += 4;
            if (o + 16ull * n_chunk > buf.size()) throw BamException("Truncated BAM index");
            if (bin != pseudoBin)  // the pseudo-bin holds metadata, not chunks
                for (uint32_t c = 0; c < n_chunk; c++) {
                    const uint64_t v = le64(&buf[o + 16 * c]);
                    first = std::min(first, v);
                    last = std::max(last, le64(&buf[o + 16 * c + 8]));  // chunk end: just past the chunk's last record
                    pts.push_back(v);
                }
            o += 16ull * n_chunk;
        }
        if (!csi) {
            if (o + 4 > buf.size()) throw BamException("Truncated BAI index");
            const uint32_t n_intv = le32(&buf[o]);
            o += 4;
            if (o + 8ull * n_intv > buf.size()) throw BamException("Truncated BAI index");
            for (uint32_t k = 0; k < n_intv; k++) {
                const uint64_t v = le64(&buf[o + 8ull * k]);
                if (v) pts.push_back(v);  // first alignment overlapping each 16 kb window
            }
            o += 8ull * n_intv;
        }
        if (r < firstOffset.size()) {
            firstOffset[r] = first;
            lastOffset[r] = last;
            std::sort(pts.begin(), pts.end());
            pts.erase(std::unique(pts.begin(), pts.end()), pts.end());
            restart[r] = std::move(pts);
        }
    }
    indexLoaded = true;
}

std::shared_ptr<RefSeqPtrList> BamReader::createRefList() const {
    auto l = std::make_shared<RefSeqPtrList>();
    for (const RefSeq& t : targets) l->push_back(std::make_shared<RefSeq>(t));
    return l;
}

std::shared_ptr<RefSeqPtrIndexMap> BamReader::createRefMap(const RefSeqPtrList& refs) const {
    auto m = std::make_shared<RefSeqPtrIndexMap>();
    for (const auto& r : refs) (*m)[r->index] = r;
    return m;
}

std::string BamReader::bamDetails() const {
    std::stringstream ss;
    ss << "BAM details:" << std::endl << " - File: " << bamFile << std::endl << " - # Target sequences: " << targets.size() << std::endl;
    return ss.str();
}

void BamReader::setRegion(int32_t tid) {
    if (tid < 0 || (size_t)tid >= targets.size()) throw BamException("setRegion: target out of range");
    regionTid = tid;
    regionLen = targets[(size_t)tid].length;
    regionDone = firstOffset[(size_t)tid] == ~0ull;
    if (!regionDone) bgzf.seek(firstOffset[(size_t)tid]);
}

// XS:A aux tag -> code (0 absent / '?' / '.', 1 '+', 2 '-', 3 anything else incl. a non-'A' typed XS,
// for which bam_aux2A returns 0 and strandFromChar throws in the reference)
static uint8_t xsCode(const uint8_t* aux, const uint8_t* end) {
    const uint8_t* p = aux;
    while (p + 3 <= end) {
        const uint8_t t0 = p[0], t1 = p[1], ty = p[2];
        p += 3;
        const bool isXS = t0 == 'X' && t1 == 'S';
        size_t sz = 0;
        switch (ty) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'd': sz = 8; break;
        case 'Z': case 'H': {
            const uint8_t* q = p;
            while (q < end && *q) q++;
            sz = (size_t)(q - p) + 1;
            break;
        }
        case 'B': {
            if (p + 5 > end) return isXS ? 3 : 0;
            const uint8_t sub = p[0];
            const uint32_t cnt = le32(p + 1);
            size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            sz = 5 + es * cnt;
            break;
        }
        default: return isXS ? 3 : 0;  // malformed aux block
        }
        if (isXS) {
            if (ty != 'A' || p >= end) return 3;
            const char c = (char)p[0];
            return c == '+' ? 1 : c == '-' ? 2 : (c == '?' || c == '.') ? 0 : 3;
        }
        p += sz;
    }
    return 0;
}

// ------------------------------------------------------------------ BamAlignment (library-level entry)
static const char* const CIGAR_LETTERS = "MIDNSHP=XB";
static const char* const NT16_LETTERS = "=ACMGRSVTWYHKDBN";

BamAlignment::BamAlignment(const std::string& nm, int32_t ref, int32_t pos, uint16_t flag, uint8_t mq, const std::string& cigarText,
                           const std::string& bases, char xs, int32_t mId, int32_t mPos) {
    name = nm;
    refId = ref;
    position = pos;
    alFlag = flag;
    mapq = mq;
    mateId = mId;
    matePos = mPos;
    xsCode = xs == '+' ? 1 : xs == '-' ? 2 : (xs == 0 || xs == '?' || xs == '.') ? 0 : 3;
    int64_t num = 0;
    bool any = false;
    for (char ch : cigarText) {
        if (ch >= '0' && ch <= '9') {
            num = num * 10 + (ch - '0');
            any = true;
            continue;
        }
        const char* at = strchr(CIGAR_LETTERS, ch);
        if (!at || !any || ch == '*') throw BamException("BamAlignment: bad CIGAR text: " + cigarText);
        cigar.emplace_back(ch, (int32_t)num);
        rawCigar.push_back(((uint32_t)num << 4) | (uint32_t)(at - CIGAR_LETTERS));
        if (CigarOp::opConsumesReference(ch)) alignedLength += (int32_t)num;
        num = 0;
        any = false;
    }
    if (bases != "*" && !bases.empty()) {
        lQseq = (int32_t)bases.size();
        seq4.assign((bases.size() + 1) / 2, 0);
        for (size_t i = 0; i < bases.size(); i++) {
            const char* at = strchr(NT16_LETTERS, bases[i] >= 'a' && bases[i] <= 'z' ? bases[i] - 32 : bases[i]);
            const uint8_t code = at ? (uint8_t)(at - NT16_LETTERS) : 15;
            seq4[i / 2] |= (uint8_t)(code << ((i & 1) ? 0 : 4));
        }
    }
}

std::string BamAlignment::deriveName() const {
    return isPaired() ? name + (isFirstMate() ? "_R1" : isSecondMate() ? "_R2" : "_R?") : name;
}

bool BamAlignment::isSplicedRead() const {
    for (const auto& op : cigar)
        if (op.type == 'N') return true;
    return false;
}

uint32_t BamAlignment::getNbJunctionsInRead() const {
    uint32_t n = 0;
    for (const auto& op : cigar) n += op.type == 'N';
    return n;
}

std::string BamAlignment::getQuerySeq() const {
    std::string s((size_t)lQseq, '=');
    for (int32_t i = 0; i < lQseq && (size_t)(i / 2) < seq4.size(); i++) s[(size_t)i] = NT16_LETTERS[(seq4[(size_t)i / 2] >> ((i & 1) ? 0 : 4)) & 15];
    return s;
}

bool BamReader::next() {
    // one record through the batch transcoder, then unpacked into the object view (names are read separately:
    // the batch layout does not carry them)
    if (regionDone) return false;
    one.clear();
    const bool keep = wantNames;
    wantNames = true;
    bool ok = false;
    try {
        ok = nextBatch(one, 1);
    } catch (...) {
        wantNames = keep;
        throw;
    }
    wantNames = keep;
    if (!ok) return false;
    const uint8_t* r = rec.data();
    const uint32_t l_name = r[8];
    cur = BamAlignment();
    cur.name.assign((const char*)r + 32, l_name ? l_name - 1 : 0);
    cur.refId = regionTid;
    cur.position = one.pos[0];
    cur.alFlag = one.flag[0];
    cur.mapq = one.mapq[0];
    cur.xsCode = one.xs[0];
    cur.lQseq = one.l_qseq[0];
    cur.mateId = one.mtid[0];
    cur.matePos = one.mpos[0];
    cur.rawCigar.assign(one.cigar.begin(), one.cigar.end());
    for (uint32_t op : cur.rawCigar) {
        const char t = CIGAR_LETTERS[(op & 15u) < 10u ? (op & 15u) : 9u];
        cur.cigar.emplace_back(t, (int32_t)(op >> 4));
        if (CigarOp::opConsumesReference(t)) cur.alignedLength += (int32_t)(op >> 4);
    }
    const uint32_t n_cig = le16(r + 12);
    const size_t seq_at = 32 + (size_t)l_name + 4ull * n_cig;
    cur.seq4.assign(r + seq_at, r + seq_at + (size_t)((cur.lQseq + 1) / 2));
    return true;
}

bool BamReader::nextBatch(ReadBatch& out, size_t maxRecords) {
    if (out.cig_off.empty()) out.clear();
    size_t added = 0;
    while (!regionDone && added < maxRecords) {
        uint8_t b4[4];
        if (bgzf.read(b4, 4) != 4) {
            regionDone = true;
            break;
        }
        const uint32_t bs = le32(b4);
        if (bs < 32) throw BamException("Invalid BAM record");
        rec.resize(bs);
        if (bgzf.read(rec.data(), bs) != bs) throw BamException("Truncated BAM record");
        const uint8_t* r = rec.data();
        const int32_t tid = (int32_t)le32(r);
        const int32_t pos = (int32_t)le32(r + 4);
        if (tid != regionTid || pos >= regionLen) {
            regionDone = true;
            break;
        }
        const uint32_t l_name = r[8];
        const uint8_t mapq = r[9];
        const uint32_t n_cig = le16(r + 12);
        const uint16_t flag = le16(r + 14);
        const int32_t l_seq = (int32_t)le32(r + 16);
        const int32_t mtid = (int32_t)le32(r + 20);
        const int32_t mpos = (int32_t)le32(r + 24);
        const size_t cig_at = 32 + l_name;
        const size_t seq_at = cig_at + 4ull * n_cig;
        const size_t seq_bytes = (size_t)((l_seq + 1) / 2);
        const size_t aux_at = seq_at + seq_bytes + (size_t)(l_seq > 0 ? l_seq : 0);
        if (l_seq < 0 || aux_at > bs) throw BamException("Invalid BAM record layout");
        out.pos.push_back(pos);
        out.flag.push_back(flag);
        out.mapq.push_back(mapq);
        out.l_qseq.push_back(l_seq);
        out.mtid.push_back(mtid);
        out.mpos.push_back(mpos);
        out.xs.push_back(xsCode(r + aux_at, r + bs));
        if (wantNames) out.name_hash.push_back(deriveNameHash(r + 32, l_name ? l_name - 1 : 0, flag));
        bool spliced = false;
        for (uint32_t k = 0; k < n_cig; k++) {
            const uint32_t op = le32(r + cig_at + 4 * k);
            out.cigar.push_back(op);
            if ((op & 15u) == 3u) {
                spliced = true;
                out.n_refskip++;
            }
        }
        out.cig_off.push_back((uint32_t)out.cigar.size());
        if (spliced && seq_bytes) {  // only spliced alignments need their bases on the device
            const size_t words = (seq_bytes + 3) / 4;
            const size_t at = out.seq4.size();
            out.seq4.resize(at + words * 4, 0);
            memcpy(&out.seq4[at], r + seq_at, seq_bytes);
        }
        out.seq_off.push_back((uint32_t)(out.seq4.size() / 4));
        added++;
    }
    return added > 0;
}

// ------------------------------------------------------------------ parallel region decode
namespace {

struct Block {
    uint64_t coff;
    uint32_t csize, isize, xlen;
};

struct Mapped {
    const uint8_t* p = nullptr;
    size_t n = 0;
    int fd = -1;
    ~Mapped() {
        if (p) munmap((void*)p, n);
        if (fd >= 0) ::close(fd);
    }
};

template <typename F>
void parallelFor(int nthreads, size_t n, F f) {  // f(thread, begin, end) over contiguous slices
    if (nthreads <= 1 || n < 2) {
        f(0, (size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    const size_t per = (n + (size_t)nthreads - 1) / (size_t)nthreads;
    for (int t = 0; t < nthreads; t++) {
        const size_t a = std::min(n, per * (size_t)t), b = std::min(n, a + per);
        if (a < b) th.emplace_back(f, t, a, b);
    }
    for (auto& x : th) x.join();
}

}  // namespace

bool BamReader::regionSpan(int32_t tid, uint64_t& fileOff, size_t& bytes, uint32_t& firstU) {
    bytes = 0;
    firstU = 0;
    fileOff = 0;
    if (tid < 0 || (size_t)tid >= targets.size()) throw BamException("regionSpan: target out of range");
    if (firstOffset[(size_t)tid] == ~0ull) return false;
    const int fd = ::open(bamFile.c_str(), O_RDONLY);
    if (fd < 0) throw BamException("Could not open BAM file: " + bamFile);
    struct Closer {
        int fd;
        ~Closer() { ::close(fd); }
    } closer{fd};
    struct stat st;
    if (fstat(fd, &st) != 0) throw BamException("Could not stat BAM file: " + bamFile);
    const uint64_t fileSize = (uint64_t)st.st_size;
    const uint64_t start = firstOffset[(size_t)tid];
    uint64_t endCoff = fileSize;
    for (uint64_t fo : firstOffset)
        if (fo != ~0ull && fo > start) endCoff = std::min<uint64_t>(endCoff, fo >> 16);
    // the index also says where the target's last chunk ends (unmapped reads may follow the last target for gigabytes)
    if (lastOffset[(size_t)tid] > start) endCoff = std::min<uint64_t>(endCoff, lastOffset[(size_t)tid] >> 16);
    uint64_t end = fileSize;
    if (endCoff < fileSize) {  // the block in which the next target starts / the last chunk ends may hold the tail: include it
        uint8_t h[18];
        if (pread(fd, h, 18, (off_t)endCoff) != 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4) || le16(h + 10) != 6 ||
            h[12] != 'B' || h[13] != 'C')
            end = fileSize;  // unusual header layout: take everything (the device stops at the first foreign record)
        else
            end = std::min<uint64_t>(fileSize, endCoff + (uint64_t)le16(h + 16) + 1);
    }
    const uint64_t lo = start >> 16;
    if (end <= lo) return false;
    fileOff = lo;
    bytes = (size_t)(end - lo);
    firstU = (uint32_t)(start & 0xffff);
    return true;
}

const uint8_t* BamReader::mapFile(const std::string& path, size_t& bytes) {
    static std::mutex mu;
    static std::map<std::string, std::pair<const uint8_t*, size_t>> maps;
    std::lock_guard<std::mutex> lk(mu);
    auto it = maps.find(path);
    if (it == maps.end()) {
        const uint8_t* p = nullptr;
        size_t n = 0;
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd >= 0) {
            struct stat st;
            if (fstat(fd, &st) == 0 && st.st_size > 0) {
                void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) {
                    p = (const uint8_t*)m;
                    n = (size_t)st.st_size;
                }
            }
            ::close(fd);
        }
        it = maps.emplace(path, std::make_pair(p, n)).first;
    }
    bytes = it->second.second;
    return it->second.first;
}

void BamReader::readSpan(uint64_t fileOff, size_t want, uint8_t* buf, int nthreads) {
    const int fd = ::open(bamFile.c_str(), O_RDONLY);
    if (fd < 0) throw BamException("Could not open BAM file: " + bamFile);
    struct Closer {
        int fd;
        ~Closer() { ::close(fd); }
    } closer{fd};
    nthreads = std::max(1, nthreads);
    const size_t nsl = std::max<size_t>(1, std::min<size_t>((size_t)nthreads, want >> 22));
    std::atomic<bool> bad(false);
    auto readSlice = [&](size_t t) {
        const size_t per = (want + nsl - 1) / nsl, a = std::min(want, per * t), b = std::min(want, a + per);
        size_t got = 0;
        while (a + got < b) {
            const ssize_t r = pread(fd, buf + a + got, b - a - got, (off_t)(fileOff + a + got));
            if (r <= 0) {
                bad = true;
                return;
            }
            got += (size_t)r;
        }
    };
    if (nsl == 1) {
        readSlice(0);
    } else {
        std::vector<std::thread> th;
        for (size_t t = 0; t < nsl; t++) th.emplace_back(readSlice, t);
        for (auto& x : th) x.join();
    }
    if (bad) throw BamException("Could not read BAM file: " + bamFile);
}

uint8_t* BamReader::readRegionBytes(int32_t tid, int nthreads, size_t& bytes, uint32_t& firstU) {
    uint64_t lo = 0;
    if (!regionSpan(tid, lo, bytes, firstU)) return nullptr;
    uint8_t* buf = (uint8_t*)bigAlloc(bytes + 64);
    try {
        readSpan(lo, bytes, buf, nthreads);
    } catch (...) {
        bigFree(buf);
        throw;
    }
    return buf;
}

void BamReader::scanRecordsParallel(int nthreads, size_t chunkBytes, const std::function<void(const FileChunk&)>& sink, int ahead) {
    nthreads = std::max(1, nthreads);
    if (getenv("PORTCULLIS_PROFILE_PIECES"))
        fprintf(stderr, "[scan] called %.3f\n", fmod(std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(), 1000.0));
    PhasePool pool(nthreads > 1 ? nthreads : 0);
    Mapped m;
    m.fd = ::open(bamFile.c_str(), O_RDONLY);
    if (m.fd < 0) throw BamException("Could not open BAM file: " + bamFile);
    struct stat st;
    if (fstat(m.fd, &st) != 0) throw BamException("Could not stat BAM file: " + bamFile);
    const uint64_t fileSize = (uint64_t)st.st_size;
    // every record start the index names, file-wide
    std::vector<uint64_t> rpts;
    for (const auto& v : restart) rpts.insert(rpts.end(), v.begin(), v.end());
    std::sort(rpts.begin(), rpts.end());
    rpts.erase(std::unique(rpts.begin(), rpts.end()), rpts.end());
    const uint64_t CHUNK = std::max<uint64_t>(chunkBytes, 1u << 20);
    const bool hooked = (bool)blockInflater;  // (the device reads cbuf and writes buf)
    struct BigFree {
        bool hooked;
        void operator()(uint8_t* p) const {
            if (hooked) hookedFree(p);
            else bigFree(p);
        }
    };
    auto bufAlloc = [hooked](size_t n) { return (uint8_t*)(hooked ? hookedAlloc(n) : bigAlloc(n)); };
    const size_t CREAD = (size_t)std::min<uint64_t>(CHUNK, 96ull << 20) + (128u << 10);
    std::unique_ptr<uint8_t[], BigFree> cbuf(bufAlloc(CREAD), BigFree{hooked});
    // A piece lives in a slot (buffer, slices, FileChunk).  ahead == 0: one slot, `sink` is called between two pieces.
    // ahead > 0: the pieces are made by a thread of this call (with the workers) up to `ahead` pieces before the one `sink`
    // -- still on the calling thread, still one piece at a time and in file order -- is looking at.
    struct Slot {
        std::unique_ptr<uint8_t[], BigFree> buf;
        size_t cap = 0;
        std::vector<std::vector<uint64_t>> sl;
        FileChunk fc;
        bool full = false;  // delivered, `sink` has not returned yet
        explicit Slot(bool hooked) : buf(nullptr, BigFree{hooked}) {}
    };
    const size_t nslots = (size_t)std::max(0, ahead) + 1;
    std::vector<Slot> slots;
    for (size_t k = 0; k < nslots; k++) slots.emplace_back(hooked);
    std::mutex qmu;
    std::condition_variable qcv;
    std::deque<size_t> ready;   // delivered slots, oldest first
    bool producerDone = false, abortScan = false;
    std::exception_ptr producerError;
    auto deliver = [&](size_t k) {
        if (nslots == 1) {
            sink(slots[k].fc);
            return;
        }
        std::lock_guard<std::mutex> lk(qmu);
        slots[k].full = true;
        ready.push_back(k);
        qcv.notify_all();
    };
    const bool tracePieces = getenv("PORTCULLIS_PROFILE_PIECES") != nullptr;  // (stderr: when each piece was read, inflated, walked)
    auto tnow = [] { return fmod(std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(), 1000.0); };
    if (tracePieces) fprintf(stderr, "[scan] buffers and workers ready %.3f\n", tnow());
    auto produce = [&]() {
    size_t carry = 0, cHave = 0, iter = 0;
    const uint8_t* carrySrc = nullptr;  // the partial record behind the last piece's last whole one (in that piece's buffer)
    uint64_t cBase = firstRecordVoffset >> 16, fileOff = cBase;
    bool first = true;
    std::vector<Block> blocks;
    std::vector<uint64_t> uoff;
    std::vector<size_t> stop;
    for (;;) {
        const double tp0 = tnow();
        // ---- refill the compressed window
        if (fileOff < fileSize && cHave < CREAD) {
            const size_t want = (size_t)std::min<uint64_t>(CREAD - cHave, fileSize - fileOff);
            const size_t nsl = std::max<size_t>(1, std::min<size_t>((size_t)std::min(nthreads, 4), want >> 20));  // (four threads read the page cache fastest: profiles/r03ap_register_probe.txt)
            std::atomic<bool> ioBad(false);
            pool.run(nsl, [&](size_t t) {
                const size_t per = (want + nsl - 1) / nsl, a = std::min(want, per * t), b = std::min(want, a + per);
                size_t got = 0;
                while (a + got < b) {
                    const ssize_t r = pread(m.fd, cbuf.get() + cHave + a + got, b - a - got, (off_t)(fileOff + a + got));
                    if (r <= 0) {
                        ioBad = true;
                        return;
                    }
                    got += (size_t)r;
                }
            });
            if (ioBad) throw BamException("Could not read BAM file: " + bamFile);
            cHave += want;
            fileOff += want;
        }
        // ---- the blocks of this piece, from their headers
        blocks.clear();
        uoff.clear();
        uint64_t total = 0;
        size_t cpos = 0;
        while (cpos + 18 <= cHave) {
            const uint8_t* h = cbuf.get() + cpos;
            if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) throw BamException("Invalid BGZF block header");
            const uint32_t xlen = le16(h + 10);
            if (cpos + 12 + xlen > cHave) break;
            int bsize = -1;
            for (uint32_t o = 0; o + 4 <= xlen;) {
                const uint8_t* x = h + 12 + o;
                const uint32_t slen = le16(x + 2);
                if (x[0] == 'B' && x[1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = le16(x + 4);
                o += 4 + slen;
            }
            if (bsize < 0) throw BamException("BGZF block without BC field");
            const uint32_t tot = (uint32_t)bsize + 1;
            if (tot < xlen + 20) throw BamException("Invalid BGZF block: BSIZE is smaller than the block's header and footer");
            if (cpos + tot > cHave) break;
            const uint32_t isz = le32(h + tot - 4);
            if (isz > 65536) throw BamException("Invalid BGZF block: ISIZE exceeds 64 KiB");
            if (total != 0 && total + isz > CHUNK) break;
            blocks.push_back({(uint64_t)cpos, tot, isz, xlen});
            uoff.push_back(total);
            total += isz;
            cpos += tot;
        }
        if (blocks.empty()) {
            if (fileOff >= fileSize) {
                if (cHave != 0) throw BamException("Truncated BGZF block at the end of the file");
                break;
            }
            if (cHave >= CREAD) throw BamException("BGZF block larger than the read window");
            continue;
        }
        const size_t nb = blocks.size(), end = carry + (size_t)total;
        const double tp1 = tnow();
        Slot& S = slots[iter++ % nslots];
        if (nslots > 1) {  // (the slot's last piece may still be with the sink)
            std::unique_lock<std::mutex> lk(qmu);
            qcv.wait(lk, [&] { return !S.full || abortScan; });
            if (abortScan) return;
        }
        {
            std::unique_ptr<uint8_t[], BigFree> fresh(nullptr, BigFree{hooked});
            size_t ncap = S.cap;
            if (end + 8 > S.cap) {
                ncap = std::max<size_t>(end + 8, (size_t)CHUNK + (4u << 20));
                fresh.reset(bufAlloc(ncap));
            }
            uint8_t* to = fresh ? fresh.get() : S.buf.get();
            if (carry) memmove(to, carrySrc, carry);  // (one slot: within the same buffer)
            if (fresh) {
                S.buf.swap(fresh);
                S.cap = ncap;
            }
        }
        std::unique_ptr<uint8_t[], BigFree>& buf = S.buf;
        std::vector<std::vector<uint64_t>>& sl = S.sl;
        const double tp2 = tnow();
        if (!(blockInflater && blockInflater(cbuf.get(), cpos, buf.get() + carry, (size_t)total))) {  // ---- inflate
            std::atomic<size_t> next(0);
            std::atomic<bool> bad(false);
            uint8_t* base = buf.get() + carry;
            pool.run(std::min<size_t>((size_t)nthreads, nb), [&](size_t) {
                z_stream zs;
                memset(&zs, 0, sizeof zs);
                if (inflateInit2(&zs, -15) != Z_OK) {
                    bad = true;
                    return;
                }
                for (;;) {
                    const size_t b = next.fetch_add(1);
                    if (b >= nb) break;
                    const Block& k = blocks[b];
                    if (!k.isize) continue;
                    if (g_fastInflate && fastInflate(cbuf.get() + k.coff + 12 + k.xlen, k.csize - 12 - k.xlen - 8, base + uoff[b], k.isize)) continue;
                    inflateReset(&zs);
                    zs.next_in = cbuf.get() + k.coff + 12 + k.xlen;
                    zs.avail_in = k.csize - 12 - k.xlen - 8;
                    zs.next_out = base + uoff[b];
                    zs.avail_out = k.isize;
                    const int rc = inflate(&zs, Z_FINISH);
                    if (rc != Z_STREAM_END || zs.avail_out != 0) bad = true;
                }
                inflateEnd(&zs);
            });
            if (bad) throw BamException("BGZF inflate failed");
        }
        const double tp3 = tnow();
        // ---- record starts the index names inside this piece
        const size_t cur0 = first ? (size_t)(firstRecordVoffset & 0xffff) : 0;
        first = false;
        std::vector<size_t> pts;
        pts.push_back(cur0);
        {
            const uint64_t vlo = (cBase + blocks[0].coff) << 16, vhi = (cBase + blocks[nb - 1].coff + 1) << 16;
            auto it = std::lower_bound(rpts.begin(), rpts.end(), vlo);
            size_t bi = 0;
            for (; it != rpts.end() && *it < vhi; ++it) {
                const uint64_t co = *it >> 16;
                while (bi < nb && cBase + blocks[bi].coff < co) bi++;
                if (bi >= nb || cBase + blocks[bi].coff != co) continue;
                const size_t p = carry + (size_t)uoff[bi] + (size_t)(*it & 0xffff);
                if (p > pts.back() && p + 36 <= end) pts.push_back(p);
            }
        }
        std::vector<size_t> cut;
        {
            const int want = std::max(1, std::min<int>(nthreads * 4, (int)pts.size()));  // (more slices than threads: the tail of a file has no points)
            cut.push_back(pts[0]);
            for (int k = 1; k < want; k++) {
                const size_t target = cur0 + (size_t)((double)(end - cur0) * k / want);
                auto it = std::lower_bound(pts.begin(), pts.end(), target);
                if (it != pts.end() && *it > cut.back()) cut.push_back(*it);
            }
        }
        const size_t ns = cut.size();
        if (sl.size() < ns) sl.resize(ns);
        stop.assign(ns, 0);
        std::atomic<bool> badWalk(false);
        pool.run(ns, [&](size_t t) {
            std::vector<uint64_t>& off = sl[t];
            off.clear();
            const size_t limit = t + 1 < ns ? cut[t + 1] : end;
            off.reserve((limit - cut[t]) / 160 + 16);
            size_t cur = cut[t];
            const uint8_t* B = buf.get();
            while (cur < limit) {
                if (cur + 4 > end) break;
                const uint32_t bs = le32(B + cur);
                if (bs < 32) {
                    badWalk = true;
                    break;
                }
                if (cur + 4 + (size_t)bs > end) break;  // partial record at the end of the piece
                const uint8_t* r = B + cur + 4;
                if (32 + (size_t)r[8] + 4ull * le16(r + 12) > bs) {
                    badWalk = true;
                    break;
                }
                off.push_back(cur);
                cur += 4 + (size_t)bs;
            }
            if (t + 1 < ns && cur != limit) badWalk = true;  // the index named a non-boundary
            stop[t] = cur;
        });
        if (badWalk) throw BamException("Invalid BAM record (or the index names an offset that is not a record start)");
        FileChunk& fc = S.fc;
        fc = FileChunk();
        fc.data = buf.get();
        fc.bytes = stop[ns - 1];
        for (size_t t = 0; t < ns; t++) {
            fc.slices.push_back(&sl[t]);
            fc.records += sl[t].size();
        }
        // ---- what is left: the partial record behind the last whole one, the compressed bytes behind the last block
        const size_t used = stop[ns - 1];
        carry = end - used;
        carrySrc = buf.get() + used;
        if (tracePieces)
            fprintf(stderr, "[scan piece] %zu blocks: read %.3f .. %.3f, slot and carry .. %.3f, inflate .. %.3f, records .. %.3f\n", nb, tp0, tp1, tp2, tp3, tnow());
        if (fc.records) deliver((size_t)(&S - slots.data()));
        memmove(cbuf.get(), cbuf.get() + cpos, cHave - cpos);
        cHave -= cpos;
        cBase += cpos;
        if (fileOff >= fileSize && cHave == 0) {
            if (carry) throw BamException("Truncated BAM record at the end of the file");
            break;
        }
    }
    };  // produce
    if (nslots == 1) {
        produce();
        return;
    }
    std::thread producer([&] {
        try {
            produce();
        } catch (...) {
            std::lock_guard<std::mutex> lk(qmu);
            producerError = std::current_exception();
        }
        std::lock_guard<std::mutex> lk(qmu);
        producerDone = true;
        qcv.notify_all();
    });
    struct Joiner {  // (also when `sink` throws: the producer is told to stop, then joined)
        std::thread& t;
        std::mutex& mu;
        std::condition_variable& cv;
        bool& abort;
        ~Joiner() {
            {
                std::lock_guard<std::mutex> lk(mu);
                abort = true;
            }
            cv.notify_all();
            t.join();
        }
    } joiner{producer, qmu, qcv, abortScan};
    for (;;) {
        size_t k;
        {
            std::unique_lock<std::mutex> lk(qmu);
            qcv.wait(lk, [&] { return !ready.empty() || producerDone; });
            if (ready.empty()) {
                if (producerError) std::rethrow_exception(producerError);
                break;
            }
            k = ready.front();
            ready.pop_front();
        }
        sink(slots[k].fc);
        std::lock_guard<std::mutex> lk(qmu);
        slots[k].full = false;
        qcv.notify_all();
    }
}


void BamReader::decodeRegionParallel(int32_t tid, int nthreads, size_t maxRecords, const std::function<void(ReadBatch&)>& sink) {
    if (tid < 0 || (size_t)tid >= targets.size()) throw BamException("decodeRegionParallel: target out of range");
    if (firstOffset[(size_t)tid] == ~0ull) return;
    nthreads = std::max(1, nthreads);
    PhasePool pool(nthreads > 1 ? nthreads : 0);
    const int32_t refLen = targets[(size_t)tid].length;
    Mapped m;  // only the descriptor is used: compressed bytes are pread() into a reusable buffer
    m.fd = ::open(bamFile.c_str(), O_RDONLY);
    if (m.fd < 0) throw BamException("Could not open BAM file: " + bamFile);
    struct stat st;
    if (fstat(m.fd, &st) != 0) throw BamException("Could not stat BAM file: " + bamFile);
    const uint64_t fileSize = (uint64_t)st.st_size;
    const uint64_t start = firstOffset[(size_t)tid];
    // the target's records end no later than the block holding the next target's first record
    uint64_t endCoff = fileSize;
    for (uint64_t fo : firstOffset)
        if (fo != ~0ull && fo > start) endCoff = std::min<uint64_t>(endCoff, fo >> 16);
    // ---- chunks of blocks: one chunk = one batch.  Each chunk is inflated in parallel, then split
    // at record starts named by the index (chunk begins / linear index) so that the block_size
    // chain -- a pointer chase through DRAM -- is walked by all threads at once.
    const uint64_t CHUNK = std::min<uint64_t>(256ull << 20, std::max<uint64_t>(64ull << 10, (uint64_t)maxRecords * 192ull));
    const std::vector<uint64_t>& rpts = restart[(size_t)tid];
    size_t bufCap = 0;
    struct BigFree {
        void operator()(uint8_t* p) const { bigFree(p); }
    };
    std::unique_ptr<uint8_t[], BigFree> buf;  // raw storage: no zero fill, huge pages
    size_t carry = 0;
    bool done = false, first = true;
    ReadBatch batch;
    struct Slice {
        std::vector<uint64_t> off;
        uint64_t ops = 0, words = 0, skips = 0;
        size_t stop = 0;  // where the walk stopped
        bool ended = false, bad = false;
    };
    std::vector<Slice> sl;
    const bool prof = getenv("PJB_PROFILE_HOST") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tRefill = 0, tScan = 0, tInflate = 0, tWalk = 0, tFill = 0, tSink = 0, tMove = 0;
    size_t nChunks = 0;
    // compressed side: a window of the file, refilled by parallel pread
    const size_t CREAD = (size_t)std::min<uint64_t>(CHUNK, 96ull << 20) + (128u << 10);
    std::unique_ptr<uint8_t[], BigFree> cbuf((uint8_t*)bigAlloc(CREAD));
    size_t cHave = 0;                   // valid bytes in cbuf
    uint64_t cBase = start >> 16;       // file offset of cbuf[0]
    uint64_t fileOff = cBase;           // next file offset to read
    bool sawLastBlock = false;
    std::vector<Block> blocks;          // blocks of the current chunk; coff is the offset inside cbuf
    while (!done) {
        // ---- refill
        double t0 = now();
        nChunks++;
        if (fileOff < fileSize && cHave < CREAD) {
            const size_t want = (size_t)std::min<uint64_t>(CREAD - cHave, fileSize - fileOff);
            const size_t nsl = std::max<size_t>(1, std::min<size_t>((size_t)std::min(nthreads, 4), want >> 20));  // (four threads read the page cache fastest: profiles/r03ap_register_probe.txt)
            std::atomic<bool> ioBad(false);
            pool.run(nsl, [&](size_t t) {
                const size_t per = (want + nsl - 1) / nsl, a = std::min(want, per * t), b = std::min(want, a + per);
                size_t got = 0;
                while (a + got < b) {
                    const ssize_t r = pread(m.fd, cbuf.get() + cHave + a + got, b - a - got, (off_t)(fileOff + a + got));
                    if (r <= 0) {
                        ioBad = true;
                        return;
                    }
                    got += (size_t)r;
                }
            });
            if (ioBad) throw BamException("Could not read BAM file: " + bamFile);
            cHave += want;
            fileOff += want;
        }
        tRefill += now() - t0;
        t0 = now();
        // ---- blocks of this chunk, from their headers
        blocks.clear();
        std::vector<uint64_t> uoff;
        uint64_t total = 0;
        size_t cpos = 0;
        while (!sawLastBlock && cpos + 18 <= cHave) {
            const uint8_t* h = cbuf.get() + cpos;
            if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) throw BamException("Invalid BGZF block header");
            const uint32_t xlen = le16(h + 10);
            if (cpos + 12 + xlen > cHave) break;
            int bsize = -1;
            for (uint32_t o = 0; o + 4 <= xlen;) {
                const uint8_t* x = h + 12 + o;
                const uint32_t slen = le16(x + 2);
                if (x[0] == 'B' && x[1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = le16(x + 4);
                o += 4 + slen;
            }
            if (bsize < 0) throw BamException("BGZF block without BC field");
            const uint32_t tot = (uint32_t)bsize + 1;
            if (tot < xlen + 20) throw BamException("Invalid BGZF block: BSIZE is smaller than the block's header and footer");
            if (cpos + tot > cHave) break;  // incomplete in the window: next refill
            const uint32_t isz = le32(h + tot - 4);
            if (isz > 65536) throw BamException("Invalid BGZF block: ISIZE exceeds 64 KiB");
            if (total != 0 && total + isz > CHUNK) break;
            blocks.push_back({(uint64_t)cpos, tot, isz, xlen});
            uoff.push_back(total);
            total += isz;
            if (cBase + cpos >= endCoff) sawLastBlock = true;  // this block may still hold the tail of the target
            cpos += tot;
        }
        if (blocks.empty()) {
            if (fileOff >= fileSize || sawLastBlock) break;  // nothing more to decode
            if (cHave >= CREAD) throw BamException("BGZF block larger than the read window");
            continue;
        }
        const size_t b0 = 0, b1 = blocks.size();
        tScan += now() - t0;
        t0 = now();
        const size_t end = carry + total;
        if (end + 8 > bufCap) {
            const size_t ncap = std::max<size_t>(end + 8, (size_t)CHUNK + (4u << 20));
            std::unique_ptr<uint8_t[], BigFree> nb((uint8_t*)bigAlloc(ncap));
            if (carry) memcpy(nb.get(), buf.get(), carry);
            buf.swap(nb);
            bufCap = ncap;
        }
        {
            std::atomic<size_t> next(b0);
            std::atomic<bool> bad(false);
            uint8_t* base = buf.get() + carry;
            auto work = [&](int, size_t, size_t) {
                for (;;) {
                    const size_t b = next.fetch_add(1);
                    if (b >= b1) break;
                    const Block& k = blocks[b];
                    if (!k.isize) continue;
                    if (g_fastInflate && fastInflate(cbuf.get() + k.coff + 12 + k.xlen, k.csize - 12 - k.xlen - 8, base + uoff[b - b0], k.isize)) continue;
                    z_stream zs;
                    memset(&zs, 0, sizeof zs);
                    if (inflateInit2(&zs, -15) != Z_OK) {
                        bad = true;
                        break;
                    }
                    zs.next_in = cbuf.get() + k.coff + 12 + k.xlen;
                    zs.avail_in = k.csize - 12 - k.xlen - 8;
                    zs.next_out = base + uoff[b - b0];
                    zs.avail_out = k.isize;
                    const int rc = inflate(&zs, Z_FINISH);
                    inflateEnd(&zs);
                    if (rc != Z_STREAM_END || zs.avail_out != 0) bad = true;
                }
            };
            const size_t nt = std::min<size_t>((size_t)nthreads, b1 - b0);
            pool.run(nt, [&](size_t t) { work((int)t, 0, 0); });
            if (bad) throw BamException("BGZF inflate failed");
        }
        tInflate += now() - t0;
        t0 = now();
        // ---- split points inside this chunk
        const size_t cur0 = first ? (size_t)(start & 0xffff) : 0;
        first = false;
        std::vector<size_t> pts;
        pts.push_back(cur0);
        {
            const uint64_t vlo = (cBase + blocks[b0].coff) << 16, vhi = ((cBase + blocks[b1 - 1].coff + 1) << 16);
            auto it = std::lower_bound(rpts.begin(), rpts.end(), vlo);
            size_t bi = b0;
            for (; it != rpts.end() && *it < vhi; ++it) {
                const uint64_t co = *it >> 16;
                while (bi < b1 && cBase + blocks[bi].coff < co) bi++;
                if (bi >= b1 || cBase + blocks[bi].coff != co) continue;  // not a block start we know: ignore
                const size_t p = carry + (size_t)uoff[bi - b0] + (size_t)(*it & 0xffff);
                if (p > pts.back() && p + 36 <= end) pts.push_back(p);
            }
        }
        // thin the points to at most nthreads slices of similar byte size
        std::vector<size_t> cut;
        {
            const int want = std::max(1, std::min<int>(nthreads, (int)pts.size()));
            cut.push_back(pts[0]);
            for (int k = 1; k < want; k++) {
                const size_t target = cur0 + (size_t)((double)(end - cur0) * k / want);
                auto it = std::lower_bound(pts.begin(), pts.end(), target);
                if (it != pts.end() && *it > cut.back()) cut.push_back(*it);
            }
        }
        const size_t ns = cut.size();
        if (sl.size() < ns) sl.resize(ns);
        for (size_t t = 0; t < ns; t++) {  // reuse the offset arrays of the previous chunk (no malloc / munmap churn)
            sl[t].off.clear();
            sl[t].ops = sl[t].words = sl[t].skips = 0;
            sl[t].stop = 0;
            sl[t].ended = sl[t].bad = false;
        }
        auto walk = [&](int t, size_t, size_t) {
            Slice& S = sl[(size_t)t];
            const size_t limit = (size_t)t + 1 < ns ? cut[(size_t)t + 1] : end;
            const bool lastSlice = (size_t)t + 1 == ns;
            S.off.reserve((limit - cut[(size_t)t]) / 160 + 16);
            size_t cur = cut[(size_t)t];
            const uint8_t* B = buf.get();
            while (cur < limit) {
                if (cur + 4 > end) break;
                const uint32_t bs = le32(B + cur);
                if (bs < 32) {
                    S.bad = true;
                    break;
                }
                if (cur + 4 + (size_t)bs > end) break;  // partial record at the end of the chunk
                const uint8_t* r = B + cur + 4;
                const int32_t rt = (int32_t)le32(r), rp = (int32_t)le32(r + 4);
                if (rt != tid || rp >= refLen) {
                    S.ended = true;
                    break;
                }
                const uint32_t l_name = r[8], n_cig = le16(r + 12);
                const int32_t l_seq = (int32_t)le32(r + 16);
                if (32 + (size_t)l_name + 4ull * n_cig > bs) {
                    S.bad = true;
                    break;
                }
                const uint8_t* cg = r + 32 + l_name;
                bool spl = false;
                for (uint32_t k = 0; k < n_cig; k++)
                    if ((cg[4 * k] & 15u) == 3u) {
                        spl = true;
                        S.skips++;
                    }
                S.ops += n_cig;
                if (spl && l_seq > 0) S.words += ((size_t)((l_seq + 1) / 2) + 3) / 4;
                S.off.push_back(cur);
                cur += 4 + (size_t)bs;
            }
            if (!lastSlice && !S.ended && !S.bad && cur != limit) S.bad = true;  // the index named a non-boundary
            S.stop = cur;
        };
        pool.run(ns, [&](size_t t) { walk((int)t, 0, 0); });
        tWalk += now() - t0;
        t0 = now();
        // ---- assemble the batch: prefix sums over slices (stop at the first slice that saw the end)
        size_t nsUse = 0, nrec = 0;
        std::vector<uint64_t> recBase(ns + 1, 0), opBase(ns + 1, 0), wordBase(ns + 1, 0);
        uint64_t skips = 0;
        size_t stopAt = end;
        for (size_t t = 0; t < ns; t++) {
            if (sl[t].bad) throw BamException("Invalid BAM record (or the index names an offset that is not a record start)");
            nsUse = t + 1;
            recBase[t + 1] = recBase[t] + sl[t].off.size();
            opBase[t + 1] = opBase[t] + sl[t].ops;
            wordBase[t + 1] = wordBase[t] + sl[t].words;
            skips += sl[t].skips;
            stopAt = sl[t].stop;
            if (sl[t].ended) {
                done = true;
                break;
            }
        }
        nrec = (size_t)recBase[nsUse];
        if (nrec) {
            const size_t n = nrec;
            batch.pos.resize(n); batch.flag.resize(n); batch.mapq.resize(n); batch.xs.resize(n); batch.l_qseq.resize(n);
            batch.mtid.resize(n); batch.mpos.resize(n);
            batch.cig_off.resize(n + 1);
            batch.seq_off.resize(n + 1);
            batch.cigar.resize(opBase[nsUse]);
            batch.seq4.resize(wordBase[nsUse] * 4);
            if (wantNames) batch.name_hash.resize(n);
            else batch.name_hash.clear();
            batch.n_refskip = skips;
            std::atomic<bool> badRec(false);
            auto fill = [&](int t, size_t, size_t) {
                const Slice& S = sl[(size_t)t];
                uint64_t co = opBase[(size_t)t], so = wordBase[(size_t)t];
                size_t i = (size_t)recBase[(size_t)t];
                const uint8_t* B = buf.get();
                for (uint64_t off : S.off) {
                    const uint32_t bs = le32(B + off);
                    const uint8_t* r = B + off + 4;
                    const uint32_t l_name = r[8], n_cig = le16(r + 12);
                    const int32_t l_seq = (int32_t)le32(r + 16);
                    const size_t cig_at = 32 + l_name, seq_at = cig_at + 4ull * n_cig;
                    const size_t seq_bytes = (size_t)((l_seq + 1) / 2);
                    const size_t aux_at = seq_at + seq_bytes + (size_t)(l_seq > 0 ? l_seq : 0);
                    if (l_seq < 0 || aux_at > bs) {
                        badRec = true;
                        return;
                    }
                    batch.pos[i] = (int32_t)le32(r + 4);
                    batch.mapq[i] = r[9];
                    batch.flag[i] = le16(r + 14);
                    batch.l_qseq[i] = l_seq;
                    batch.mtid[i] = (int32_t)le32(r + 20);
                    batch.mpos[i] = (int32_t)le32(r + 24);
                    batch.xs[i] = xsCode(r + aux_at, r + bs);
                    if (wantNames) batch.name_hash[i] = deriveNameHash(r + 32, l_name ? l_name - 1 : 0, batch.flag[i]);
                    batch.cig_off[i] = (uint32_t)co;
                    batch.seq_off[i] = (uint32_t)so;
                    bool spl = false;
                    for (uint32_t k = 0; k < n_cig; k++) {
                        const uint32_t op = le32(r + cig_at + 4 * k);
                        batch.cigar[co++] = op;
                        spl |= (op & 15u) == 3u;
                    }
                    if (spl && seq_bytes) {
                        const size_t w = (seq_bytes + 3) / 4;
                        memset(&batch.seq4[(so + w - 1) * 4], 0, 4);  // zero the padding of the last word
                        memcpy(&batch.seq4[so * 4], r + seq_at, seq_bytes);
                        so += w;
                    }
                    i++;
                }
            };
            pool.run(nsUse, [&](size_t t) { fill((int)t, 0, 0); });
            if (badRec) throw BamException("Invalid BAM record layout");
            batch.cig_off[n] = (uint32_t)opBase[nsUse];
            batch.seq_off[n] = (uint32_t)wordBase[nsUse];
            tFill += now() - t0;
            t0 = now();
            sink(batch);
            tSink += now() - t0;
            t0 = now();
        }
        // ---- carry the partial record at the end of the chunk
        carry = done ? 0 : end - stopAt;
        if (carry) memmove(buf.get(), buf.get() + stopAt, carry);
        // ---- slide the compressed window
        const size_t consumed = (size_t)blocks.back().coff + blocks.back().csize;
        if (consumed < cHave) memmove(cbuf.get(), cbuf.get() + consumed, cHave - consumed);
        cHave -= consumed;
        cBase += consumed;
        tMove += now() - t0;
        if (sawLastBlock) break;
    }
    if (prof)
        fprintf(stderr, "[host profile] decode tid %d (%d threads, %zu chunks): refill %.3f scan %.3f inflate %.3f walk %.3f fill %.3f sink %.3f carry %.3f\n",
                tid, nthreads, nChunks, tRefill, tScan, tInflate, tWalk, tFill, tSink, tMove);
}

}  // namespace bam
}  // namespace portcullis
