// BamAlignment: one alignment record as the reference's library callers see it
// (lib/include/portcullis/bam/bam_alignment.hpp:101-360): the fields the junc path reads
// (lib/src/bam_alignment.cc:71-100 caches exactly these) plus the CIGAR as (type, length) operations.
// It is the per-record door of the library-level entry JunctionSystem::addJunctions(const BamAlignment&)
// (lib/include/portcullis/junction_system.hpp:128-132); the bulk path never builds these objects.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "bam_master.hpp"

namespace portcullis {
namespace bam {

struct CigarOp {  // lib/include/portcullis/bam/bam_alignment.hpp:62-99
    char type;
    int32_t length;
    CigarOp(char t, int32_t l) : type(t), length(l) {}
    static bool opConsumesQuery(char op) { return op == 'M' || op == 'I' || op == 'S' || op == '=' || op == 'X'; }
    static bool opConsumesReference(char op) { return op == 'M' || op == 'D' || op == 'N' || op == '=' || op == 'X'; }
};

class BamAlignment {
    friend class BamReader;
    std::string name;
    int32_t refId = -1, position = -1, mateId = -1, matePos = -1, lQseq = 0, alignedLength = 0;
    uint16_t alFlag = 0;
    uint8_t mapq = 0, xsCode = 0;   // xs: 0 none / '?' / '.', 1 '+', 2 '-', 3 invalid
    std::vector<CigarOp> cigar;
    std::vector<uint32_t> rawCigar; // BAM-native
    std::vector<uint8_t> seq4;      // BAM-native 4-bit bases

public:
    BamAlignment() = default;
    // build a record by hand (tests, other front ends): cigar as text ("30M100N40M"), bases as letters or "*"
    BamAlignment(const std::string& name, int32_t refId, int32_t pos, uint16_t flag, uint8_t mapq, const std::string& cigarText,
                 const std::string& bases, char xs = 0, int32_t mateId = -1, int32_t matePos = -1);

    const std::string& getName() const { return name; }
    std::string deriveName() const;  // lib/src/bam_alignment.cc:233-242
    int32_t getReferenceId() const { return refId; }
    int32_t getPosition() const { return position; }
    int32_t getStart() const { return position; }
    int32_t getEnd() const { return position + alignedLength - 1; }
    int32_t getLength() const { return lQseq; }
    int32_t getMateReferenceId() const { return mateId; }
    int32_t getMatePosition() const { return matePos; }
    uint16_t getAlignmentFlag() const { return alFlag; }
    uint8_t getMapQuality() const { return mapq; }
    uint8_t getXsCode() const { return xsCode; }
    const std::vector<CigarOp>& getCigar() const { return cigar; }
    const std::vector<uint32_t>& getRawCigar() const { return rawCigar; }
    const std::vector<uint8_t>& getPackedSeq() const { return seq4; }
    bool isPaired() const { return alFlag & 0x1; }
    bool isMapped() const { return !(alFlag & 0x4); }
    bool isReverseStrand() const { return alFlag & 0x10; }
    bool isFirstMate() const { return alFlag & 0x40; }
    bool isSecondMate() const { return alFlag & 0x80; }
    bool isSplicedRead() const;                 // lib/src/bam_alignment.cc:294-301
    uint32_t getNbJunctionsInRead() const;      // lib/src/bam_alignment.cc:303-311
    std::string getQuerySeq() const;            // lib/src/bam_alignment.cc:244-250
};

}  // namespace bam
}  // namespace portcullis
