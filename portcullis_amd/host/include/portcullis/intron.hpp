// Intron: the (reference sequence, start, end) key of a junction, 0-based inclusive.
// API of lib/include/portcullis/intron.hpp:44-149 / lib/src/intron.cc of the reference.
#pragma once

#include <algorithm>
#include <functional>
#include <ostream>
#include <sstream>

#include "bam/bam_master.hpp"

namespace portcullis {

struct IntronException : public PortcullisException {
    explicit IntronException(const std::string& m) : PortcullisException(m) {}
};

class Intron {
public:
    bam::RefSeq ref;
    int32_t start = 0;  // first base of the intron
    int32_t end = 0;    // last base of the intron

    Intron() = default;
    Intron(const bam::RefSeq& r, int32_t s, int32_t e) : ref(r), start(s), end(e) {}

    // equality ignores everything but (ref index, start, end)   (intron.hpp operator==)
    bool operator==(const Intron& o) const { return ref.index == o.ref.index && start == o.start && end == o.end; }
    bool operator!=(const Intron& o) const { return !(*this == o); }

    int32_t size() const { return end - start + 1; }

    // intron.cc:55-58 (strand is ignored, as in the reference)
    bool sharesDonorOrAcceptor(const Intron& o) const {
        return ref.index == o.ref.index && (start == o.start || end == o.end);
    }

    // intron.cc:67-83: min(left anchor length, right anchor length); throws if an anchor is inverted
    uint32_t minAnchorLength(int32_t leftAnchorStart, int32_t rightAnchorEnd) const {
        if (leftAnchorStart > start)
            throw IntronException("The intron start position must be greater than the left anchor start position: " +
                                  std::to_string(leftAnchorStart) + " **** " + toString() + " **** " +
                                  std::to_string(rightAnchorEnd));
        if (rightAnchorEnd < end)
            throw IntronException("The intron end position must be less than the right anchor end position: " +
                                  std::to_string(leftAnchorStart) + " **** " + toString() + " **** " +
                                  std::to_string(rightAnchorEnd));
        return (uint32_t)std::min(start - leftAnchorStart, rightAnchorEnd - end);
    }

    std::string toString() const {
        std::stringstream ss;
        ss << ref.name << "(" << start << "," << end << ")";
        return ss.str();
    }

    void outputDescription(std::ostream& strm, const std::string& delimiter = "; ") const {
        strm << "RefId: " << ref.index << delimiter << "RefName: " << ref.name << delimiter
             << "RefLength: " << ref.length << delimiter << "Start: " << start << delimiter << "End: " << end;
    }

    friend std::ostream& operator<<(std::ostream& strm, const Intron& l) {
        return strm << l.ref.index << "\t" << l.ref.name << "\t" << l.ref.length << "\t" << l.start << "\t" << l.end;
    }

    static std::string locationOutputHeader() { return "refid\trefname\treflen\tstart\tend"; }
};

struct IntronHasher {
    size_t operator()(const Intron& l) const {
        size_t seed = 0;
        auto mix = [&seed](size_t v) { seed ^= v + 0x9e3779b9 + (seed << 6) + (seed >> 2); };
        mix(std::hash<int32_t>()(l.ref.index));
        mix(std::hash<int32_t>()(l.start));
        mix(std::hash<int32_t>()(l.end));
        return seed;
    }
};

struct IntronComparator {  // order by (ref index, start, end)
    bool operator()(const Intron& l, const Intron& r) const {
        if (l.ref.index != r.ref.index) return l.ref.index < r.ref.index;
        if (l.start != r.start) return l.start < r.start;
        return l.end < r.end;
    }
};

}  // namespace portcullis
