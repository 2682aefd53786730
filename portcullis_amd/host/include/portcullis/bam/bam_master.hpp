// Enumerations, RefSeq and exception types of the junc path.
// Mirrors lib/include/portcullis/bam/bam_master.hpp:46-230 of the reference (same names,
// same enumerator order, same string forms) without Boost.
#pragma once

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <strings.h>
#include <unordered_map>
#include <vector>

namespace portcullis {

// Base of every exception of this library; carries the message the reference attaches through
// boost::error_info.
struct PortcullisException : public std::runtime_error {
    explicit PortcullisException(const std::string& m) : std::runtime_error(m) {}
};

namespace bam {

struct BamException : public PortcullisException {
    explicit BamException(const std::string& m) : PortcullisException(m) {}
};

enum class Strand { POSITIVE, NEGATIVE, UNKNOWN };

inline Strand strandFromBool(bool reverseStrand) { return reverseStrand ? Strand::NEGATIVE : Strand::POSITIVE; }

inline Strand strandFromChar(char strand) {  // bam_master.hpp:60-72 (throws on anything else)
    switch (strand) {
    case '+': return Strand::POSITIVE;
    case '-': return Strand::NEGATIVE;
    case '?':
    case '.': return Strand::UNKNOWN;
    }
    throw BamException(std::string("Unknown strand: ") + strand);
}

inline char strandToChar(Strand s) { return s == Strand::POSITIVE ? '+' : s == Strand::NEGATIVE ? '-' : '?'; }

inline std::string strandToString(Strand s) {
    return s == Strand::POSITIVE ? "POSITIVE" : s == Strand::NEGATIVE ? "NEGATIVE" : "UNKNOWN";
}

enum class Strandedness : std::uint8_t { UNSTRANDED, FIRSTSTRAND, SECONDSTRAND, UNKNOWN };

inline std::string strandednessToString(Strandedness ss) {
    switch (ss) {
    case Strandedness::UNSTRANDED: return "UNSTRANDED";
    case Strandedness::FIRSTSTRAND: return "FIRSTSTRAND";
    case Strandedness::SECONDSTRAND: return "SECONDSTRAND";
    case Strandedness::UNKNOWN: return "UNKNOWN";
    }
    return "[Unknown StrandSpecific type]";
}

inline std::string strandednessToLongString(Strandedness ss) {
    switch (ss) {
    case Strandedness::UNSTRANDED: return "Unstranded - can't determine transcript strand from read strand";
    case Strandedness::FIRSTSTRAND: return "Firststrand - R1 is not on transcript strand";
    case Strandedness::SECONDSTRAND: return "Secondstrand - R1 is on transcript strand";
    case Strandedness::UNKNOWN: return "Unknown strand protocol";
    }
    return "[Unknown StrandSpecific type]";
}

inline Strandedness strandednessFromString(const std::string& ss) {
    if (!strcasecmp(ss.c_str(), "UNSTRANDED")) return Strandedness::UNSTRANDED;
    if (!strcasecmp(ss.c_str(), "FIRSTSTRAND")) return Strandedness::FIRSTSTRAND;
    if (!strcasecmp(ss.c_str(), "SECONDSTRAND")) return Strandedness::SECONDSTRAND;
    if (!strcasecmp(ss.c_str(), "UNKNOWN")) return Strandedness::UNKNOWN;
    throw BamException("Unknown strandedness: " + ss);
}

enum class Orientation : std::uint8_t { SE, FR, RF, FF, UNKNOWN };

inline bool doProperPairCheck(Orientation o) {
    return o == Orientation::FR || o == Orientation::FF || o == Orientation::RF;
}

inline std::string orientationToString(Orientation o) {
    switch (o) {
    case Orientation::SE: return "SE";
    case Orientation::FR: return "FR";
    case Orientation::RF: return "RF";
    case Orientation::FF: return "FF";
    case Orientation::UNKNOWN: return "UNKNOWN";
    }
    return "[Unknown Orientation type]";
}

inline std::string orientationToLongString(Orientation o) {
    switch (o) {
    case Orientation::SE: return "Single-End (SE)";
    case Orientation::FR: return "Paired-End (FR): Forward Reverse (-> <-)";
    case Orientation::RF: return "Paired-End (RF): Reverse Forward (<- ->)";
    case Orientation::FF: return "Paired-End (FF): Forward Forward (-> ->)";
    case Orientation::UNKNOWN: return "Unknown";
    }
    return "[Unknown Orientation type]";
}

inline Orientation orientationFromString(const std::string& ss) {
    if (!strcasecmp(ss.c_str(), "SE")) return Orientation::SE;
    if (!strcasecmp(ss.c_str(), "FR")) return Orientation::FR;
    if (!strcasecmp(ss.c_str(), "RF")) return Orientation::RF;
    if (!strcasecmp(ss.c_str(), "FF")) return Orientation::FF;
    if (!strcasecmp(ss.c_str(), "UNKNOWN")) return Orientation::UNKNOWN;
    throw BamException("Unknown orientation: " + ss);
}

// Reference sequence descriptor (bam_master.hpp RefSeq)
struct RefSeq {
    int32_t index = -1;
    std::string name;
    int32_t length = 0;

    RefSeq() = default;
    RefSeq(int32_t i, const std::string& n, int32_t l) : index(i), name(n), length(l) {}

    std::string toString() const {
        return std::to_string(index) + ": " + name + " (" + std::to_string(length) + ")";
    }
};

typedef std::shared_ptr<RefSeq> RefSeqPtr;
typedef std::vector<RefSeqPtr> RefSeqPtrList;
typedef std::unordered_map<int32_t, RefSeqPtr> RefSeqPtrIndexMap;

}  // namespace bam
}  // namespace portcullis
