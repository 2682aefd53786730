#include <sys/stat.h>
#include <unistd.h>
#include <thread>
#include <atomic>
#include <portcullis/bam/genome_mapper.hpp>

#include <algorithm>
#include <cctype>
#include <cstring>
#include <cstdio>
#include <fstream>
#include <sstream>

namespace portcullis {
namespace bam {

GenomeMapper::~GenomeMapper() {
    if (fp) fclose(fp);
}

void GenomeMapper::buildFastaIndex() {
    std::ifstream in(genomeFile.c_str(), std::ios::binary);
    if (!in) throw BamException("Could not open genome file: " + genomeFile);
    std::ofstream out(getFastaIndexFile().c_str());
    std::string line, name;
    int64_t off = 0, len = 0, seqOff = 0;
    int32_t lb = 0, lw = 0;
    auto flush = [&]() {
        if (!name.empty()) out << name << "\t" << len << "\t" << seqOff << "\t" << lb << "\t" << lw << "\n";
    };
    while (std::getline(in, line)) {
        const int64_t raw = (int64_t)line.size() + 1;
        if (!line.empty() && line[0] == '>') {
            flush();
            std::stringstream ss(line.substr(1));
            ss >> name;
            len = 0;
            lb = lw = 0;
            seqOff = off + raw;
        } else {
            size_t n = line.size();
            while (n && (line[n - 1] == '\r')) n--;
            if (lb == 0 && n) {
                lb = (int32_t)n;
                lw = (int32_t)raw;
            }
            len += (int64_t)n;
        }
        off += raw;
    }
    flush();
}

void GenomeMapper::loadFastaIndex() {
    std::ifstream in(getFastaIndexFile().c_str());
    if (!in) throw BamException("Could not open genome index: " + getFastaIndexFile());
    entries.clear();
    byName.clear();
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        std::stringstream ss(line);
        Entry e;
        std::string f;
        std::getline(ss, e.name, '\t');
        ss >> e.len >> e.offset >> e.line_blen >> e.line_len;
        if (!ss) throw BamException("Malformed line in " + getFastaIndexFile() + ": " + line);
        byName[e.name] = entries.size();
        entries.push_back(e);
    }
    if (fp) fclose(fp);
    fp = fopen(genomeFile.c_str(), "rb");
    if (!fp) throw BamException("Could not open genome file: " + genomeFile);
}

int64_t GenomeMapper::getSeqLength(const std::string& name) const {
    auto it = byName.find(name);
    return it == byName.end() ? -1 : entries[it->second].len;
}

// read `count` graphic characters starting at base `beg` of the sequence
std::string GenomeMapper::readSpan(const Entry& e, int64_t beg, int64_t count) const {
    std::string out;
    if (count <= 0 || e.line_blen <= 0) return out;
    out.reserve((size_t)count);
    const int64_t start = e.offset + beg / e.line_blen * e.line_len + beg % e.line_blen;
    if (fseeko(fp, (off_t)start, SEEK_SET) != 0) throw BamException("Seek failed in genome file");
    // upper bound of bytes to read: bases plus line terminators
    const int64_t lines = count / e.line_blen + 2;
    const int64_t maxBytes = count + lines * (e.line_len - e.line_blen) + 2;
    std::string buf((size_t)maxBytes, 0);
    const size_t got = fread(&buf[0], 1, (size_t)maxBytes, fp);
    // fast path: well-formed lines (line_blen graphic characters, then the terminator) are copied
    // line by line; anything else falls back to the character filter faidx applies
    {
        out.resize((size_t)count);
        size_t i = 0, o = 0;
        int64_t col = beg % e.line_blen;
        bool ok = true;
        while (o < (size_t)count && ok) {
            const size_t take = std::min<size_t>((size_t)(e.line_blen - col), (size_t)count - o);
            if (i + take > got) {
                ok = false;
                break;
            }
            const char* src = &buf[i];
            unsigned char acc_lo = 0xff, acc_hi = 0;
            for (size_t k = 0; k < take; k++) {
                const unsigned char ch = (unsigned char)src[k];
                acc_lo = ch < acc_lo ? ch : acc_lo;
                acc_hi = ch > acc_hi ? ch : acc_hi;
            }
            if (acc_lo <= 32 || acc_hi >= 127) {
                ok = false;
                break;
            }
            memcpy(&out[o], src, take);
            o += take;
            i += take;
            if (o < (size_t)count) {
                // skip the line terminator (line_len - line_blen bytes, none of them graphic)
                const size_t term = (size_t)(e.line_len - e.line_blen);
                for (size_t k = 0; k < term; k++)
                    if (i + k >= got || isgraph((unsigned char)buf[i + k])) ok = false;
                i += term;
            }
            col = 0;
        }
        if (ok) return out;
    }
    out.clear();
    for (size_t i = 0; i < got && (int64_t)out.size() < count; i++)
        if (isgraph((unsigned char)buf[i])) out.push_back(buf[i]);
    return out;
}

std::string GenomeMapper::fetchBases(const char* name, int start, int end) const {
    auto it = byName.find(name);
    if (it == byName.end()) return std::string();
    const Entry& e = entries[it->second];
    int64_t b = start, x = end;
    if (x < b) b = x;
    if (b < 0) b = 0;
    else if (e.len <= b) b = e.len - 1;
    if (x < 0) x = 0;
    else if (e.len <= x) x = e.len - 1;
    return readSpan(e, b, x - b + 1);
}

std::string GenomeMapper::fetchContig(const std::string& name) const {
    auto it = byName.find(name);
    if (it == byName.end()) throw BamException("The sequence \"" + name + "\" not found in " + genomeFile);
    const Entry& e = entries[it->second];
    return readSpan(e, 0, e.len);
}

bool GenomeMapper::rawSpan(const std::string& name, RawSpan& out) const {
    auto it = byName.find(name);
    if (it == byName.end()) return false;
    const Entry& e = entries[it->second];
    if (e.line_blen <= 0 || e.line_len < e.line_blen || e.len < 0 || e.offset < 0) return false;
    if (e.line_len - e.line_blen > 8) return false;  // (line ends are one or two bytes; anything odd is left to the character filter)
    out.fileOffset = (uint64_t)e.offset;
    out.lineBases = e.line_blen;
    out.lineWidth = e.line_len;
    out.length = e.len;
    out.bytes = e.len == 0 ? 0 : (size_t)((e.len - 1) / e.line_blen * e.line_len + (e.len - 1) % e.line_blen + 1);
    // an index line that describes more bytes than the file holds is not a layout to rely on (and nothing that large
    // should be page-locked on its word)
    struct stat st;
    if (!fp || fstat(fileno(fp), &st) != 0 || out.fileOffset + out.bytes > (uint64_t)st.st_size) return false;
    return true;
}

bool GenomeMapper::readRaw(const RawSpan& span, uint8_t* dst, int nthreads) const {
    const int fd = fileno(fp);
    const size_t want = span.bytes;
    const size_t nsl = std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, nthreads), want >> 22));
    std::atomic<bool> bad(false), eof(false);
    auto slice = [&](size_t t) {
        const size_t per = (want + nsl - 1) / nsl, a = std::min(want, per * t), b = std::min(want, a + per);
        size_t got = 0;
        while (a + got < b) {
            const ssize_t r = pread(fd, dst + a + got, b - a - got, (off_t)(span.fileOffset + a + got));
            if (r == 0) {
                eof = true;
                return;
            }
            if (r < 0) {
                bad = true;
                return;
            }
            got += (size_t)r;
        }
    };
    if (nsl == 1) slice(0);
    else {
        std::vector<std::thread> th;
        for (size_t t = 0; t < nsl; t++) th.emplace_back(slice, t);
        for (auto& x : th) x.join();
    }
    if (bad) throw BamException("Could not read genome file: " + genomeFile);
    return !eof;
}

}  // namespace bam
}  // namespace portcullis
