// JunctionSystem: merge, sort/index, calcJunctionStats, strandedness report, writers, loader.
// Behaviour follows lib/src/junction_system.cc of the reference (line numbers in comments).
#include <portcullis/junction_system.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <exception>
#include <fstream>
#include <thread>
#include <iostream>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../../include/portcullis_amd.h"

namespace portcullis {

std::string JunctionSystem::version = "";

// junction_system.cc:55-70: chain of consecutive junctions sharing a donor or acceptor with the
// previous member; returns the index of the last member
size_t JunctionSystem::createJunctionGroup(size_t index, std::vector<JunctionPtr>& group) {
    const JunctionPtr* cur = &junctionList[index];  // (by reference: a shared_ptr copy is two atomic operations, a quarter of a million times)
    group.push_back(*cur);
    for (size_t j = index + 1; j < junctionList.size(); j++) {
        const JunctionPtr& next = junctionList[j];
        if (!(*cur)->sharesDonorOrAcceptor(next)) return j - 1;
        group.push_back(next);
        cur = &next;
    }
    return junctionList.size() - 1;
}

void JunctionSystem::findJunctions(int32_t refId, JunctionList& subset) {
    subset.clear();
    for (const JunctionPtr& j : junctionList)
        if (j->getIntron()->ref.index == refId) subset.push_back(j);
}

void JunctionSystem::addJunction(JunctionPtr j) {
    distinctJunctions[*(j->getIntron())] = j;
    junctionList.push_back(j);
}

void JunctionSystem::append(JunctionSystem& other) {
    for (const auto& j : other.getJunctions()) addJunction(j);
}

// append() for a system that is not needed as a map afterwards (the per-target systems of findJunctions): its list is copied behind this
// one's, its map's NODES move over (no allocation, no copy of a key: a quarter of a million inserts were 40 ms of the merge) and whatever
// stays behind -- an intron this system has already -- overwrites this system's entry, as addJunction would have
void JunctionSystem::absorb(JunctionSystem& other) {
    junctionList.insert(junctionList.end(), other.junctionList.begin(), other.junctionList.end());
    distinctJunctions.merge(other.distinctJunctions);
    for (auto& kv : other.distinctJunctions) distinctJunctions[kv.first] = kv.second;
    other.distinctJunctions.clear();
}

void JunctionSystem::appendRows(const pjb_junction_row* rows, size_t n) {
    if (!refs) throw JunctionException("JunctionSystem::appendRows: reference sequence list is not set");
    junctionList.reserve(junctionList.size() + n);
    for (size_t i = 0; i < n; i++) addJunction(Junction::fromRow(rows[i], *refs));
}

bool JunctionSystem::addJunctions(const bam::BamAlignment& al) {
    if (!refs) throw JunctionException("JunctionSystem::addJunctions: reference sequence list is not set");
    const int32_t tid = al.getReferenceId();
    if (tid < 0 || (size_t)tid >= refs->size()) throw JunctionException("JunctionSystem::addJunctions: alignment is not placed on a known target");
    bam::ReadBatch& b = pending[tid];
    if (b.cig_off.empty()) b.clear();
    b.pos.push_back(al.getPosition());
    b.flag.push_back(al.getAlignmentFlag());
    b.mapq.push_back(al.getMapQuality());
    b.xs.push_back(al.getXsCode());
    b.l_qseq.push_back(al.getLength());
    b.mtid.push_back(al.getMateReferenceId());
    b.mpos.push_back(al.getMatePosition());
    for (uint32_t op : al.getRawCigar()) b.cigar.push_back(op);
    b.cig_off.push_back((uint32_t)b.cigar.size());
    const bool spliced = al.isSplicedRead();
    if (spliced && !al.getPackedSeq().empty()) {  // only spliced alignments need their bases on the device
        const std::vector<uint8_t>& s = al.getPackedSeq();
        const size_t words = (s.size() + 3) / 4, at = b.seq4.size();
        b.seq4.resize(at + words * 4, 0);
        memcpy(&b.seq4[at], s.data(), s.size());
        b.n_refskip += al.getNbJunctionsInRead();
    }
    b.seq_off.push_back((uint32_t)(b.seq4.size() / 4));
    return spliced;
}

size_t JunctionSystem::finish(const bam::GenomeMapper& gmap, Orientation orientation, int device) {
    if (!refs) throw JunctionException("JunctionSystem::finish: reference sequence list is not set");
    if (pending.empty()) return 0;
    pjb_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = PJB_ABI_VERSION;
    cfg.device = device;
    cfg.orientation = (int32_t)orientation;
    cfg.strandedness = PJB_SS_UNKNOWN;
    pjb_ctx* ctx = nullptr;
    if (pjb_create(&ctx, &cfg) != PJB_OK) throw JunctionException(std::string("pjb_create: ") + pjb_last_error(nullptr));
    struct Closer {
        pjb_ctx* c;
        ~Closer() { pjb_destroy(c); }
    } closer{ctx};
    std::vector<int32_t> lens;
    for (auto& r : *refs) lens.push_back(r->length);
    auto check = [&](int rc, const char* what) {
        if (rc != PJB_OK) throw JunctionException(std::string(what) + ": " + pjb_last_error(ctx));
    };
    check(pjb_set_refs(ctx, (int32_t)lens.size(), lens.data()), "pjb_set_refs");
    std::vector<int32_t> tids;
    for (auto& kv : pending) tids.push_back(kv.first);
    std::sort(tids.begin(), tids.end());
    size_t added = 0;
    for (int32_t tid : tids) {
        bam::ReadBatch& b = pending[tid];
        if (b.size() == 0) continue;
        const std::string contig = gmap.fetchContig(refs->at((size_t)tid)->name);
        check(pjb_upload_contig(ctx, tid, (const uint8_t*)contig.data(), (int64_t)contig.size()), "pjb_upload_contig");
        pjb_batch pb;
        b.view(pb);
        check(pjb_submit_batch(ctx, tid, &pb), "pjb_submit_batch");
        pjb_region_result rr;
        check(pjb_finish_contig(ctx, tid, &rr), "pjb_finish_contig");
        const pjb_junction_row* rows = nullptr;
        int64_t n = 0;
        check(pjb_collect(ctx, &rows, &n), "pjb_collect");
        appendRows(rows, (size_t)n);
        added += (size_t)n;
        check(pjb_clear_rows(ctx), "pjb_clear_rows");
        check(pjb_release_contig(ctx, tid), "pjb_release_contig");
    }
    pending.clear();
    return added;
}

JunctionPtr JunctionSystem::getJunction(const Intron& intron) const {
    auto it = distinctJunctions.find(intron);
    return it == distinctJunctions.end() ? nullptr : it->second;
}

// junction_system.cc:250-320
void JunctionSystem::calcJunctionStats() {
    if (junctionList.empty()) return;
    const size_t n = junctionList.size();
    std::vector<JunctionPtr> group;  // (one vector for every group: a quarter of a million allocations otherwise)
    for (size_t i = 0; i < n; i++) {
        group.clear();
        i = createJunctionGroup(i, group);
        uint32_t maxReads = 0;
        size_t maxIndex = 0;
        const bool unique = group.size() == 1;
        for (size_t k = 0; k < group.size(); k++) {
            if (maxReads < group[k]->getNbSplicedAlignments()) {
                maxReads = group[k]->getNbSplicedAlignments();
                maxIndex = k;
            }
            group[k]->setUniqueJunction(unique);
        }
        group[maxIndex]->setPrimaryJunction(true);
    }
    // distances to the neighbouring junctions; -1 (stored in a uint32_t) marks "none on this contig"
    size_t i = 0;
    bool lastdiffseq = false;
    while (i + 1 < n) {
        JunctionPtr first = junctionList[i], second = junctionList[i + 1];
        int32_t diff = second->getIntron()->start - first->getIntron()->end;
        diff = diff < 0 ? 0 : diff;
        if (first->getIntron()->ref.index != second->getIntron()->ref.index) {
            first->setDistanceToNextUpstreamJunction((uint32_t)-1);
            second->setDistanceToNextDownstreamJunction((uint32_t)-1);
            if (i == 0 || lastdiffseq) first->setDistanceToNextDownstreamJunction((uint32_t)-1);
            if (i == n - 2) second->setDistanceToNextUpstreamJunction((uint32_t)-1);
            lastdiffseq = true;
        } else if (i == 0) {
            first->setDistanceToNextDownstreamJunction((uint32_t)-1);
            first->setDistanceToNextUpstreamJunction((uint32_t)diff);
            second->setDistanceToNextDownstreamJunction((uint32_t)diff);
            lastdiffseq = false;
        } else if (i == n - 2) {
            first->setDistanceToNextUpstreamJunction((uint32_t)diff);
            second->setDistanceToNextDownstreamJunction((uint32_t)diff);
            second->setDistanceToNextUpstreamJunction((uint32_t)-1);
            lastdiffseq = false;
        } else {
            first->setDistanceToNextUpstreamJunction((uint32_t)diff);
            second->setDistanceToNextDownstreamJunction((uint32_t)diff);
            lastdiffseq = false;
        }
        i++;
    }
    for (auto& junc : junctionList) {
        const int32_t down = (int32_t)junc->getDistanceToNextDownstreamJunction();
        const int32_t up = (int32_t)junc->getDistanceToNextUpstreamJunction();
        junc->setDistanceToNearestJunction((uint32_t)((down == -1 || up == -1) ? std::max(down, up) : std::min(down, up)));
        junc->setMeanReadLength((uint32_t)this->meanQueryLength);
        if (junc->isSuspicious()) {
            const double prob = 1.0 - std::pow((junc->getMaxMMES() / (this->meanQueryLength / 2.0)), junc->getNbSplicedAlignments());
            if (prob > 0.99) junc->setPotentialFalsePositive(true);
        }
    }
}

void JunctionSystem::sort() {
    // the device returns every target's junctions in (start, end) order and targets are appended in index order, so
    // the merged list is usually sorted already: one linear check instead of n log n pointer-chasing compares
    if (!std::is_sorted(junctionList.begin(), junctionList.end(), JunctionComparator()))
        std::sort(junctionList.begin(), junctionList.end(), JunctionComparator());
}

void JunctionSystem::reserve(size_t n) {
    junctionList.reserve(n);
    distinctJunctions.reserve(n);
}

void JunctionSystem::index() {
    for (size_t i = 0; i < this->size() && i < junctionList.size(); i++) junctionList[i]->setId((uint32_t)i);
}

// junction_system.cc:455-560 (stdout report + inferred protocol)
std::pair<Orientation, Strandedness> JunctionSystem::determineStrandedness(bool verbose) const {
    uint32_t r1p_p = 0, r1n_p = 0, r2p_p = 0, r2n_p = 0, r1p_n = 0, r1n_n = 0, r2p_n = 0, r2n_n = 0;
    for (const JunctionPtr& j : junctionList) {
        if (j->getSpliceSiteStrand() == Strand::POSITIVE) {
            r1p_p += j->getNbR1PosAlignments();
            r1n_p += j->getNbR1NegAlignments();
            r2p_p += j->getNbR2PosAlignments();
            r2n_p += j->getNbR2NegAlignments();
        } else if (j->getSpliceSiteStrand() == Strand::NEGATIVE) {
            r1p_n += j->getNbR1PosAlignments();
            r1n_n += j->getNbR1NegAlignments();
            r2p_n += j->getNbR2PosAlignments();
            r2n_n += j->getNbR2NegAlignments();
        }
    }
    const double posr1 = ((double)((int32_t)r1p_p - (int32_t)r1n_p)) / ((double)(r1p_p + r1n_p));
    const double negr1 = ((double)((int32_t)r1n_n - (int32_t)r1p_n)) / ((double)(r1p_n + r1n_n));
    const double posr2 = ((double)((int32_t)r2p_p - (int32_t)r2n_p)) / ((double)(r2p_p + r2n_p));
    const double negr2 = ((double)((int32_t)r2n_n - (int32_t)r2p_n)) / ((double)(r2p_n + r2n_n));
    const uint32_t totalr1 = r1p_p + r1n_p + r1p_n + r1n_n;
    const uint32_t totalr2 = r2p_p + r2n_p + r2p_n + r2n_n;
    if (verbose) {
        using std::cout;
        using std::endl;
        cout << "Strand Analysis" << endl << "---------------" << endl << endl;
        cout << "Total Alignments:" << endl << " - R1:" << totalr1 << endl << " - R2:" << totalr2 << endl;
        cout << "Alignment counts when splice site suggests +ve strand:" << endl
             << " - R1+: " << r1p_p << endl << " - R1-: " << r1n_p << endl << " - R2+: " << r2p_p << endl << " - R2-: " << r2n_p << endl
             << "Alignment counts when splice site suggests -ve strand:" << endl
             << " - R1+: " << r1p_n << endl << " - R1-: " << r1n_n << endl << " - R2+: " << r2p_n << endl << " - R2-: " << r2n_n << endl;
        cout << "Correlation of read strand to splice site strand (1.0 = complete agreement, -1.0 = complete disagreement):" << endl
             << " - R1+: " << posr1 << endl << " - R1-: " << negr1 << endl << " - R2+: " << posr2 << endl << " - R2-: " << negr2 << endl << endl;
    }
    Strandedness s = Strandedness::UNKNOWN;
    Orientation o = Orientation::UNKNOWN;
    if (totalr1 == 0 && totalr2 == 0) {
    } else if (totalr2 == 0) {
        o = Orientation::SE;
        if (posr1 > 0.5 && negr1 > 0.5) s = Strandedness::SECONDSTRAND;
        else if (posr1 < -0.5 && negr1 < -0.5) s = Strandedness::FIRSTSTRAND;
    } else {
        o = Orientation::FR;
        if (posr1 > 0.5 && negr1 > 0.5 && posr2 < -0.5 && negr2 < -0.5) s = Strandedness::SECONDSTRAND;
        else if (posr1 < -0.5 && negr1 < -0.5 && posr2 > 0.5 && negr2 > 0.5) s = Strandedness::FIRSTSTRAND;
        else if (posr1 > 0.5 && negr1 > 0.5 && posr2 > 0.5 && negr2 > 0.5) {
            s = Strandedness::SECONDSTRAND;
            o = Orientation::FF;
        } else if (posr1 < -0.5 && negr1 < -0.5 && posr2 < -0.5 && negr2 < -0.5) {
            s = Strandedness::FIRSTSTRAND;
            o = Orientation::FF;
        }
    }
    if (std::fabs(posr1) <= 0.5 && std::fabs(negr1) <= 0.5 && std::fabs(posr2) <= 0.5 && std::fabs(negr2) <= 0.5)
        s = Strandedness::UNSTRANDED;
    return std::make_pair(o, s);
}

void JunctionSystem::saveAll(const std::string& outputPrefix, const std::string& source) {
    saveAll(outputPrefix, source, false, false, false);
}

namespace {
// text of items [0, n) in order, formatted by up to 16 threads on contiguous slices
template <typename F>
std::vector<std::string> formatSlices(size_t n, size_t bytesPerItem, F fn) {
    const size_t nt = n < 20000 ? 1 : std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<std::string> parts(nt);
    auto work = [&](size_t t) {
        const size_t a = n * t / nt, b = n * (t + 1) / nt;
        parts[t].reserve((b - a) * bytesPerItem + 64);
        for (size_t i = a; i < b; i++) fn(i, parts[t]);
    };
    if (nt == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (size_t t = 0; t < nt; t++) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    return parts;
}

// the parts, concatenated, as the content of `path`; large outputs are written by one thread per part (pwrite)
void writeParts(const std::string& path, const std::vector<std::string>& parts) {
    size_t total = 0;
    for (const auto& p : parts) total += p.size();
    const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) throw portcullis::JunctionException("Could not open output file: " + path);
    bool ok = true;
    auto put = [&](const std::string& p, size_t at) {
        size_t done = 0;
        while (done < p.size()) {
            const ssize_t w = pwrite(fd, p.data() + done, p.size() - done, (off_t)(at + done));
            if (w <= 0) {
                ok = false;
                return;
            }
            done += (size_t)w;
        }
    };
    if (total < ((size_t)16 << 20)) {
        size_t at = 0;
        for (const auto& p : parts) {
            put(p, at);
            at += p.size();
        }
    } else {
        std::vector<std::thread> th;
        size_t at = 0;
        for (const auto& p : parts) {
            th.emplace_back(put, std::cref(p), at);
            at += p.size();
        }
        for (auto& x : th) x.join();
    }
    ::close(fd);
    if (!ok) throw portcullis::JunctionException("Could not write output file: " + path);
}
}  // namespace

// junction_system.cc:336-383
void JunctionSystem::saveAll(const std::string& outputPrefix, const std::string& source, bool bedscore, bool outputExonGFF,
                             bool outputIntronGFF) {
    using std::cout;
    using std::endl;
    const std::string tabPath = outputPrefix + ".junctions.tab";
    const std::string exonGffPath = outputPrefix + ".junctions.exon.gff3";
    const std::string intronGffPath = outputPrefix + ".junctions.intron.gff3";
    const std::string bedPath = outputPrefix + ".junctions.bed";
    // the .bed is formatted and written beside the .tab (the messages keep the reference's order)
    std::exception_ptr bedError;
    std::thread bedThread([&] {
        try {
            outputBED(bedPath, CanonicalSS::ALL, source, bedscore);
        } catch (...) {
            bedError = std::current_exception();
        }
    });
    struct Join {
        std::thread& t;
        ~Join() {
            if (t.joinable()) t.join();
        }
    } joinBed{bedThread};
    cout << " - Saving junction table to: " << tabPath << " ... ";
    cout.flush();
    {
        // same bytes as `f << (*this) << endl` (header, rows, one more empty line), formatted without iostreams,
        // slices of the list on a few threads
        std::string head = Junction::junctionOutputHeader();
        head.push_back('\n');
        std::vector<std::string> parts = formatSlices(junctionList.size(), 320, [&](size_t i, std::string& out) {
            junctionList[i]->appendTabRow(out);
            out.push_back('\n');
        });
        parts.insert(parts.begin(), head);
        parts.push_back("\n");
        writeParts(tabPath, parts);
    }
    cout << "done." << endl;
    if (outputExonGFF) {
        cout << " - Saving junction GFF file to: " << exonGffPath << " ... ";
        cout.flush();
        std::ofstream f(exonGffPath.c_str());
        writeExonGFF(f, source);
        cout << "done." << endl;
    }
    if (outputIntronGFF) {
        cout << " - Saving intron GFF file to: " << intronGffPath << " ... ";
        cout.flush();
        std::ofstream f(intronGffPath.c_str());
        writeIntronGFF(f, source);
        cout << "done." << endl;
    }
    cout << " - Saving BED file with all junctions to: " << bedPath << " ... ";
    cout.flush();
    bedThread.join();
    if (bedError) std::rethrow_exception(bedError);
    cout << "done." << endl;
}

void JunctionSystem::outputDescription(std::ostream& strm) {
    for (const JunctionPtr& j : junctionList) {
        strm << "Junction " << j->getId() << ":" << std::endl;
        j->outputDescription(strm);
        strm << std::endl;
    }
}

std::ostream& operator<<(std::ostream& strm, const JunctionSystem& js) {
    strm << Junction::junctionOutputHeader() << "\n";
    for (const auto& j : js.junctionList) strm << *j << "\n";  // same bytes as endl, one flush at the end
    return strm;
}

void JunctionSystem::writeExonGFF(std::ostream& strm, const std::string& source) {
    for (const JunctionPtr& j : junctionList) j->outputJunctionGFF(strm, source);
}

void JunctionSystem::writeIntronGFF(std::ostream& strm, const std::string& source) {
    for (const JunctionPtr& j : junctionList) j->outputIntronGFF(strm, source);
}

void JunctionSystem::outputBED(const std::string& path, CanonicalSS type, const std::string& prefix, bool bedscore) {
    std::string out = "track name=\"junctions\" description=\"Portcullis V" + (version.empty() ? std::string("X.X.X") : version) +
                      " junctions\"\n";
    std::vector<std::string> parts = formatSlices(junctionList.size(), 96, [&](size_t i, std::string& o) {
        const JunctionPtr& j = junctionList[i];
        if (type == CanonicalSS::ALL || j->getSpliceSiteType() == type) j->appendBedRow(o, prefix, bedscore);
    });
    parts.insert(parts.begin(), out);
    writeParts(path, parts);
}

void JunctionSystem::outputBED(std::ostream& strm, CanonicalSS type, const std::string& prefix, bool bedscore) {
    strm << "track name=\"junctions\" description=\"Portcullis V" << (version.empty() ? "X.X.X" : version) << " junctions\""
         << std::endl;
    for (const JunctionPtr& j : junctionList)
        if (type == CanonicalSS::ALL || j->getSpliceSiteType() == type) j->outputBED(strm, prefix, bedscore);
}

// junction_system.cc:424-444: skips empty lines and any line containing "index"
void JunctionSystem::load(const std::string& junctionTabFile, bool simple) {
    struct stat st;
    if (stat(junctionTabFile.c_str(), &st) != 0)
        throw JunctionException("Could not find Portcullis junction tab file at: " + junctionTabFile);
    std::ifstream ifs(junctionTabFile.c_str());
    std::string line;
    while (std::getline(ifs, line)) {
        // boost::trim
        size_t a = 0, b = line.size();
        while (a < b && isspace((unsigned char)line[a])) a++;
        while (b > a && isspace((unsigned char)line[b - 1])) b--;
        line = line.substr(a, b - a);
        if (!line.empty() && line.find("index") == std::string::npos) {
            JunctionPtr j = Junction::parse(line);
            junctionList.push_back(j);
            if (!simple) distinctJunctions[*(j->getIntron())] = j;
        }
    }
}

}  // namespace portcullis
