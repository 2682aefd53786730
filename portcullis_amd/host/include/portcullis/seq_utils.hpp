// SeqUtils: hamming distance / reverse complement helpers with the reference's semantics
// (lib/include/portcullis/seq_utils.hpp:33-118).
#pragma once

#include <string>

#include "bam/bam_master.hpp"

namespace portcullis {

struct SeqUtilsException : public PortcullisException {
    explicit SeqUtilsException(const std::string& m) : PortcullisException(m) {}
};

class SeqUtils {
public:
    static bool dnaNt(char c) { return c == 'A' || c == 'T' || c == 'G' || c == 'C'; }

    static char upper(char c) { return (c >= 'a' && c <= 'z') ? char(c - 32) : c; }

    static std::string makeClean(const std::string& s) {
        std::string o(s);
        for (char& c : o) {
            c = upper(c);
            if (!dnaNt(c)) c = 'N';
        }
        return o;
    }

    // upper-cases both sides; throws when the lengths differ
    static uint32_t hammingDistance(const std::string& a, const std::string& b) {
        if (a.size() != b.size())
            throw SeqUtilsException("Can't find hamming distance of strings that are not the same length.  s1: " +
                                    std::to_string(a.size()) + "\"" + a + "\"; s2: " + std::to_string(b.size()) + "\"" + b + "\"");
        uint32_t n = 0;
        for (size_t i = 0; i < a.size(); i++) n += upper(a[i]) != upper(b[i]);
        return n;
    }

    static std::string reverseSeq(const std::string& s) { return std::string(s.rbegin(), s.rend()); }

    // complement table of the reference (IUPAC aware; letters without an entry and anything
    // outside 'A'..'Z' become NUL -- the reference indexes its table out of bounds there)
    static char complement(char c) {
        switch (c) {
        case 'A': return 'T';
        case 'C': return 'G';
        case 'D': return 'H';
        case 'G': return 'C';
        case 'H': return 'D';
        case 'M': return 'K';
        case 'N': return 'N';
        case 'R': return 'Y';
        case 'S': return 'W';
        case 'T': return 'A';
        case 'U': return 'A';
        case 'V': return 'B';
        case 'W': return 'S';
        case 'X': return 'X';
        case 'Y': return 'R';
        default: return 0;
        }
    }

    static std::string reverseComplement(const std::string& s) {
        std::string o(s.size(), 0);
        for (size_t i = 0; i < s.size(); i++) o[s.size() - 1 - i] = complement(s[i]);
        return o;
    }
};

}  // namespace portcullis
