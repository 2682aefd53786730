// Junction: one candidate splice junction with every metric the `junc` stage reports.
// Same accessor surface as lib/include/portcullis/junction.hpp:189-1351 of the reference for the
// junc path; the per-alignment accumulation (addJunctionAlignment / calcMetrics /
// processJunctionWindow) is NOT done here -- it runs on the GPU behind the C ABI and arrives as a
// pjb_junction_row (see fromRow).
#pragma once

#include <array>
#include <memory>
#include <ostream>
#include <string>
#include <vector>

#include "bam/bam_master.hpp"
#include "intron.hpp"
#include "seq_utils.hpp"

struct pjb_junction_row;

namespace portcullis {

using bam::Strand;

const uint16_t MAP_QUALITY_THRESHOLD = 30;

struct JunctionException : public PortcullisException {
    explicit JunctionException(const std::string& m) : PortcullisException(m) {}
};

const std::string CANONICAL_SEQ = "GTAG";
const std::string SEMI_CANONICAL_SEQ1 = "ATAC";
const std::string SEMI_CANONICAL_SEQ2 = "GCAG";
const std::string CANONICAL_SEQ_RC = "CTAC";
const std::string SEMI_CANONICAL_SEQ1_RC = "GTAT";
const std::string SEMI_CANONICAL_SEQ2_RC = "CTGC";

enum class CanonicalSS { CANONICAL, SEMI_CANONICAL, NO, ALL };

inline CanonicalSS cssFromChar(char c) {
    return c == 'C' ? CanonicalSS::CANONICAL : c == 'S' ? CanonicalSS::SEMI_CANONICAL : CanonicalSS::NO;
}
inline char cssToChar(CanonicalSS c) {
    switch (c) {
    case CanonicalSS::CANONICAL: return 'C';
    case CanonicalSS::SEMI_CANONICAL: return 'S';
    case CanonicalSS::NO: return 'N';
    case CanonicalSS::ALL: return 'A';
    }
    return 'N';
}
inline std::string cssToString(CanonicalSS c) {
    switch (c) {
    case CanonicalSS::CANONICAL: return "Canonical";
    case CanonicalSS::SEMI_CANONICAL: return "Semi-canonical";
    case CanonicalSS::NO: return "No";
    case CanonicalSS::ALL: return "All";
    }
    return "No";
}

typedef std::shared_ptr<Intron> IntronPtr;

class Junction {
public:
    static const std::vector<std::string> METRIC_NAMES;
    static const std::vector<std::string> JAD_NAMES;
    static const std::vector<std::string> STRAND_NAMES;

private:
    IntronPtr intron;
    CanonicalSS canonicalSpliceSites = CanonicalSS::NO;
    uint32_t nbAlRaw = 0, nbAlDistinct = 0, nbAlMultiplySpliced = 0, nbAlUniquelyMapped = 0;
    uint32_t nbAlBamProperlyPaired = 0, nbAlPortcullisProperlyPaired = 0, nbAlReliable = 0;
    uint32_t nbAlR1Pos = 0, nbAlR1Neg = 0, nbAlR2Pos = 0, nbAlR2Neg = 0;
    double entropy = 0, meanMismatches = 0, meanReadLength = 0;
    uint32_t maxMinAnchor = 0, maxMMES = 0;
    double intronScore = 0;
    uint32_t hammingDistance5p = 10, hammingDistance3p = 10;
    double codingPotential = 0, positionWeightScore = 0, splicingSignal = 0;
    bool uniqueJunction = false, primaryJunction = false;
    uint32_t nbDownstreamJunctions = 0, nbUpstreamJunctions = 0;
    uint32_t distanceToNextDownstreamJunction = 0, distanceToNextUpstreamJunction = 0, distanceToNearestJunction = 0;
    double multipleMappingScore = 0, coverage = 0;
    uint32_t nbDownstreamFlankingAlignments = 0, nbUpstreamFlankingAlignments = 0, nbSamples = 1;
    bool suspicious = false, pfp = false;
    std::array<uint32_t, 20> junctionAnchorDepth{};
    Strand readStrand = Strand::UNKNOWN, ssStrand = Strand::UNKNOWN, consensusStrand = Strand::UNKNOWN;
    double score = 0;
    int32_t leftAncStart = 0, rightAncEnd = 0;
    std::string da1, da2;
    uint32_t id = 0;
    bool genuine = false;

public:
    Junction(IntronPtr location, int32_t leftAncStart, int32_t rightAncEnd);

    // Build from a device row (one junction after calcMetrics + processJunctionWindow).
    static std::shared_ptr<Junction> fromRow(const pjb_junction_row& row, const bam::RefSeqPtrList& refs);

    // ---- location
    IntronPtr getIntron() const { return intron; }
    uint32_t getIntronSize() const { return intron ? (uint32_t)intron->size() : 0; }
    int32_t getLeftAncStart() const { return leftAncStart; }
    int32_t getRightAncEnd() const { return rightAncEnd; }
    int32_t getLeftAnchorSize() const { return intron ? intron->start - leftAncStart : 0; }
    int32_t getRightAnchorSize() const { return intron ? rightAncEnd - intron->end : 0; }
    size_t size() const { return (size_t)(rightAncEnd - leftAncStart + 1); }
    bool sharesDonorOrAcceptor(const std::shared_ptr<Junction>& o) const { return intron->sharesDonorOrAcceptor(*o->intron); }
    void extendAnchors(int32_t otherStart, int32_t otherEnd);

    // ---- splice sites / strand
    CanonicalSS setDonorAndAcceptorMotif(std::string seq1, std::string seq2);
    CanonicalSS hasCanonicalSpliceSites(const std::string& seq1, const std::string& seq2) const;
    Strand predictedStrandFromSpliceSites(const std::string& seq1, const std::string& seq2) const;
    CanonicalSS getSpliceSiteType() const { return canonicalSpliceSites; }
    bool isCanonical() const { return canonicalSpliceSites == CanonicalSS::CANONICAL; }
    Strand getReadStrand() const { return readStrand; }
    Strand getSpliceSiteStrand() const { return ssStrand; }
    Strand getConsensusStrand() const { return consensusStrand; }
    void setReadStrand(Strand s) { readStrand = s; }
    void setSpliceSiteStrand(Strand s) { ssStrand = s; }
    void setConsensusStrand(Strand s) { consensusStrand = s; }
    const std::string& getDa1() const { return da1; }
    const std::string& getDa2() const { return da2; }
    void setDa1(const std::string& s) { da1 = s; }
    void setDa2(const std::string& s) { da2 = s; }

    // ---- static helper kept from the reference API
    static double calcEntropy(const std::vector<int32_t>& sortedJunctionPositions);

    // ---- getters (column names in comments)
    uint32_t getId() const { return id; }
    double getScore() const { return score; }
    bool isSuspicious() const { return suspicious; }
    bool isPotentialFalsePositive() const { return pfp; }
    bool isGenuine() const { return genuine; }
    uint32_t getNbSplicedAlignments() const { return nbAlRaw; }                      // nb_raw_aln
    uint32_t getNbDistinctAlignments() const { return nbAlDistinct; }                // nb_dist_aln
    uint32_t getNbUniquelySplicedAlignments() const { return nbAlRaw - nbAlMultiplySpliced; }  // nb_us_aln
    uint32_t getNbMultiplySplicedAlignments() const { return nbAlMultiplySpliced; }  // nb_ms_aln
    uint32_t getNbUniquelyMappedAlignments() const { return nbAlUniquelyMapped; }    // nb_um_aln
    uint32_t getNbMultiplyMappedAlignments() const { return nbAlRaw - nbAlUniquelyMapped; }  // nb_mm_aln
    uint32_t getNbBamProperlyPairedAlignments() const { return nbAlBamProperlyPaired; }
    uint32_t getNbPortcullisProperlyPairedAlignments() const { return nbAlPortcullisProperlyPaired; }
    uint32_t getNbReliableAlignments() const { return nbAlReliable; }
    double getReliable2RawAlignmentRatio() const { return (double)nbAlReliable / (double)nbAlRaw; }
    uint32_t getNbR1PosAlignments() const { return nbAlR1Pos; }
    uint32_t getNbR1NegAlignments() const { return nbAlR1Neg; }
    uint32_t getNbR2PosAlignments() const { return nbAlR2Pos; }
    uint32_t getNbR2NegAlignments() const { return nbAlR2Neg; }
    double getEntropy() const { return entropy; }
    double getMeanMismatches() const { return meanMismatches; }
    double getMeanReadLength() const { return meanReadLength; }
    uint32_t getMaxMinAnchor() const { return maxMinAnchor; }
    uint32_t getMaxMMES() const { return maxMMES; }
    double getIntronScore() const { return intronScore; }
    uint32_t getHammingDistance5p() const { return hammingDistance5p; }
    uint32_t getHammingDistance3p() const { return hammingDistance3p; }
    double getCodingPotential() const { return codingPotential; }
    double getPositionWeightScore() const { return positionWeightScore; }
    double getSplicingSignal() const { return splicingSignal; }
    bool isUniqueJunction() const { return uniqueJunction; }
    bool isPrimaryJunction() const { return primaryJunction; }
    uint32_t getNbUpstreamJunctions() const { return nbUpstreamJunctions; }
    uint32_t getNbDownstreamJunctions() const { return nbDownstreamJunctions; }
    uint32_t getDistanceToNextUpstreamJunction() const { return distanceToNextUpstreamJunction; }
    uint32_t getDistanceToNextDownstreamJunction() const { return distanceToNextDownstreamJunction; }
    uint32_t getDistanceToNearestJunction() const { return distanceToNearestJunction; }
    double getMultipleMappingScore() const { return multipleMappingScore; }
    double getCoverage() const { return coverage; }
    uint32_t getNbUpstreamFlankingAlignments() const { return nbUpstreamFlankingAlignments; }
    uint32_t getNbDownstreamFlankingAlignments() const { return nbDownstreamFlankingAlignments; }
    uint32_t getNbSamples() const { return nbSamples; }
    uint32_t getJunctionAnchorDepth(size_t i) const { return junctionAnchorDepth[i]; }

    // ---- setters
    void setId(uint32_t v) { id = v; }
    void setScore(double v) { score = v; }
    void setSuspicious(bool v) { suspicious = v; }
    void setPotentialFalsePositive(bool v) { pfp = v; }
    void setGenuine(bool v) { genuine = v; }
    void setNbSplicedAlignments(uint32_t v) { nbAlRaw = v; }
    void setNbDistinctAlignments(uint32_t v) { nbAlDistinct = v; }
    void setNbMultiplySplicedAlignments(uint32_t v) { nbAlMultiplySpliced = v; }
    void setNbUniquelyMappedAlignments(uint32_t v) { nbAlUniquelyMapped = v; }
    void setNbBamProperlyPairedAlignments(uint32_t v) { nbAlBamProperlyPaired = v; }
    void setNbPortcullisProperlyPairedAlignments(uint32_t v) { nbAlPortcullisProperlyPaired = v; }
    void setNbReliableAlignments(uint32_t v) { nbAlReliable = v; }
    void setNbR1PosAlignments(uint32_t v) { nbAlR1Pos = v; }
    void setNbR1NegAlignments(uint32_t v) { nbAlR1Neg = v; }
    void setNbR2PosAlignments(uint32_t v) { nbAlR2Pos = v; }
    void setNbR2NegAlignments(uint32_t v) { nbAlR2Neg = v; }
    void setEntropy(double v) { entropy = v; }
    void setMeanMismatches(double v) { meanMismatches = v; }
    // the reference's setter takes a uint32_t, so the mean is truncated (junction.hpp:928)
    void setMeanReadLength(uint32_t v) { meanReadLength = v; }
    void setMaxMinAnchor(int32_t v) { maxMinAnchor = (uint32_t)v; }
    void setMaxMMES(uint32_t v) { maxMMES = v; }
    void setIntronScore(double v) { intronScore = v; }
    void setHammingDistance5p(uint32_t v) { hammingDistance5p = v; }
    void setHammingDistance3p(uint32_t v) { hammingDistance3p = v; }
    void setCodingPotential(double v) { codingPotential = v; }
    void setPositionWeightScore(double v) { positionWeightScore = v; }
    void setSplicingSignal(double v) { splicingSignal = v; }
    void setUniqueJunction(bool v) { uniqueJunction = v; }
    void setPrimaryJunction(bool v) { primaryJunction = v; }
    void setNbUpstreamJunctions(uint32_t v) { nbUpstreamJunctions = v; }
    void setNbDownstreamJunctions(uint32_t v) { nbDownstreamJunctions = v; }
    void setDistanceToNextUpstreamJunction(uint32_t v) { distanceToNextUpstreamJunction = v; }
    void setDistanceToNextDownstreamJunction(uint32_t v) { distanceToNextDownstreamJunction = v; }
    void setDistanceToNearestJunction(uint32_t v) { distanceToNearestJunction = v; }
    void setMultipleMappingScore(double v) { multipleMappingScore = v; }
    void setCoverage(double v) { coverage = v; }
    void setNbUpstreamFlankingAlignments(uint32_t v) { nbUpstreamFlankingAlignments = v; }
    void setNbDownstreamFlankingAlignments(uint32_t v) { nbDownstreamFlankingAlignments = v; }
    void setNbSamples(uint32_t v) { nbSamples = v; }
    void setJunctionAnchorDepth(size_t i, uint32_t v) { junctionAnchorDepth[i] = v; }

    // ---- lookup by column name (reference: getValueFromName / getIntFromName)
    double getValueFromName(const std::string& name) const;

    // ---- output
    void outputDescription(std::ostream& strm, const std::string& delimiter = "\n") const;
    void condensedOutputDescription(std::ostream& strm, const std::string& delimiter = "\n") const;
    void outputIntronGFF(std::ostream& strm, const std::string& source) const;
    void outputJunctionGFF(std::ostream& strm, const std::string& source) const;
    void outputBED(std::ostream& strm, const std::string& prefix, bool bedscore) const;
    friend std::ostream& operator<<(std::ostream& strm, const Junction& j);
    // the same bytes as `strm << j` / outputBED(strm, ...), appended to a string without going through
    // iostream formatting (the writers of a few hundred thousand junctions spend their time there)
    void appendTabRow(std::string& out) const;
    void appendBedRow(std::string& out, const std::string& prefix, bool bedscore) const;

    static std::string junctionOutputHeader();
    static std::shared_ptr<Junction> parse(const std::string& line);
};

typedef std::shared_ptr<Junction> JunctionPtr;
typedef std::vector<JunctionPtr> JunctionList;

struct JunctionComparator {
    bool operator()(const JunctionPtr& a, const JunctionPtr& b) const {
        return IntronComparator()(*a->getIntron(), *b->getIntron());
    }
};

}  // namespace portcullis
