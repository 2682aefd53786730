// Junction: row construction, text formats and parsing.
// Formats follow the reference byte for byte (lib/include/portcullis/junction.hpp:1260-1319,
// lib/src/junction.cc:1020-1326): default-formatted ostream for the .tab, fixed/3 for BED,
// setprecision(4)/(9) quirks in the exon GFF.
#include <portcullis/junction.hpp>

#include <cmath>
#include <cstdio>
#include <iomanip>
#include <sstream>

#include "../../../include/portcullis_amd.h"

namespace portcullis {

const std::vector<std::string> Junction::METRIC_NAMES({
    "canonical_ss", "score", "suspicious", "pfp", "nb_raw_aln", "nb_dist_aln", "nb_us_aln", "nb_ms_aln", "nb_um_aln",
    "nb_mm_aln", "nb_bpp_aln", "nb_ppp_aln", "nb_rel_aln", "rel2raw", "nb_r1_pos", "nb_r1_neg", "nb_r2_pos", "nb_r2_neg",
    "entropy", "mean_mismatches", "mean_readlen", "max_min_anc", "maxmmes", "intron_score", "hamming5p", "hamming3p",
    "coding", "pws", "splice_sig", "uniq_junc", "primary_junc", "nb_up_juncs", "nb_down_juncs", "dist_2_up_junc",
    "dist_2_down_junc", "dist_nearest_junc", "mm_score", "coverage", "up_aln", "down_aln", "nb_samples"});

const std::vector<std::string> Junction::JAD_NAMES({"JAD01", "JAD02", "JAD03", "JAD04", "JAD05", "JAD06", "JAD07",
                                                    "JAD08", "JAD09", "JAD10", "JAD11", "JAD12", "JAD13", "JAD14",
                                                    "JAD15", "JAD16", "JAD17", "JAD18", "JAD19", "JAD20"});

const std::vector<std::string> Junction::STRAND_NAMES = {"read-strand", "ss-strand", "consensus-strand"};

Junction::Junction(IntronPtr location, int32_t left, int32_t right) : intron(location), leftAncStart(left), rightAncEnd(right) {
    maxMinAnchor = intron->minAnchorLength(left, right);
}

static Strand strandFromCode(uint8_t c) {
    return c == PJB_STRAND_POS ? Strand::POSITIVE : c == PJB_STRAND_NEG ? Strand::NEGATIVE : Strand::UNKNOWN;
}

std::shared_ptr<Junction> Junction::fromRow(const pjb_junction_row& r, const bam::RefSeqPtrList& refs) {
    const bam::RefSeq& ref = *refs.at((size_t)r.refid);
    auto in = std::make_shared<Intron>(ref, r.start, r.end);
    auto j = std::make_shared<Junction>(in, r.left, r.right);
    j->readStrand = strandFromCode(r.read_strand);
    j->ssStrand = strandFromCode(r.ss_strand);
    j->consensusStrand = strandFromCode(r.cons_strand);
    j->canonicalSpliceSites = r.canonical == PJB_CSS_CANONICAL ? CanonicalSS::CANONICAL
                              : r.canonical == PJB_CSS_SEMI    ? CanonicalSS::SEMI_CANONICAL
                                                               : CanonicalSS::NO;
    j->da1.assign((const char*)r.da1, 2);
    j->da2.assign((const char*)r.da2, 2);
    j->suspicious = r.suspicious != 0;
    j->nbAlRaw = r.nb_raw;
    j->nbAlDistinct = r.nb_dist;
    j->nbAlMultiplySpliced = r.nb_ms;
    j->nbAlUniquelyMapped = r.nb_um;
    j->nbAlBamProperlyPaired = r.nb_bpp;
    j->nbAlPortcullisProperlyPaired = r.nb_ppp;
    j->nbAlReliable = r.nb_rel;
    j->nbAlR1Pos = r.r1pos;
    j->nbAlR1Neg = r.r1neg;
    j->nbAlR2Pos = r.r2pos;
    j->nbAlR2Neg = r.r2neg;
    j->entropy = r.entropy;
    // (double) nbMismatches / (double) alignments.size(); nbMismatches is a uint32_t there (junction.cc:863,893)
    j->meanMismatches = (double)(uint32_t)r.sum_mismatches / (double)r.nb_raw;
    j->maxMinAnchor = r.max_min_anc;
    j->maxMMES = r.maxmmes;
    j->hammingDistance5p = r.hamming5p;
    j->hammingDistance3p = r.hamming3p;
    j->nbUpstreamJunctions = r.nb_up_juncs;
    j->nbDownstreamJunctions = r.nb_down_juncs;
    for (size_t k = 0; k < 20; k++) j->junctionAnchorDepth[k] = r.jad[k];
    return j;
}

void Junction::extendAnchors(int32_t otherStart, int32_t otherEnd) {
    leftAncStart = std::min(leftAncStart, otherStart);
    rightAncEnd = std::max(rightAncEnd, otherEnd);
    maxMinAnchor = std::max(maxMinAnchor, intron->minAnchorLength(otherStart, otherEnd));
}

CanonicalSS Junction::hasCanonicalSpliceSites(const std::string& seq1, const std::string& seq2) const {
    if (intron == nullptr || seq1.size() != 2 || seq2.size() != 2)
        throw JunctionException("Can't test for valid donor / acceptor when either string are not of length two, or the "
                                "intron location is not defined");
    const std::string seq = seq1 + seq2;
    if (seq == CANONICAL_SEQ || seq == CANONICAL_SEQ_RC) return CanonicalSS::CANONICAL;
    if (seq == SEMI_CANONICAL_SEQ1 || seq == SEMI_CANONICAL_SEQ1_RC || seq == SEMI_CANONICAL_SEQ2 || seq == SEMI_CANONICAL_SEQ2_RC)
        return CanonicalSS::SEMI_CANONICAL;
    return CanonicalSS::NO;
}

Strand Junction::predictedStrandFromSpliceSites(const std::string& seq1, const std::string& seq2) const {
    if (seq1.size() != 2 || seq2.size() != 2)
        throw JunctionException("Can't test donor / acceptor when either string are not of length two");
    const std::string seq = seq1 + seq2;
    if (seq == CANONICAL_SEQ || seq == SEMI_CANONICAL_SEQ1 || seq == SEMI_CANONICAL_SEQ2) return Strand::POSITIVE;
    if (seq == CANONICAL_SEQ_RC || seq == SEMI_CANONICAL_SEQ1_RC || seq == SEMI_CANONICAL_SEQ2_RC) return Strand::NEGATIVE;
    return Strand::UNKNOWN;
}

CanonicalSS Junction::setDonorAndAcceptorMotif(std::string seq1, std::string seq2) {
    canonicalSpliceSites = hasCanonicalSpliceSites(seq1, seq2);
    ssStrand = predictedStrandFromSpliceSites(seq1, seq2);
    consensusStrand = readStrand == ssStrand          ? readStrand
                      : readStrand == Strand::UNKNOWN ? ssStrand
                      : ssStrand == Strand::UNKNOWN   ? readStrand
                                                      : Strand::UNKNOWN;
    da1 = consensusStrand == Strand::NEGATIVE ? SeqUtils::reverseComplement(seq2) : seq1;
    da2 = consensusStrand == Strand::NEGATIVE ? SeqUtils::reverseComplement(seq1) : seq2;
    return canonicalSpliceSites;
}

// lib/src/junction.cc:730-749 including its grouping rule (a flush also swallows the first read of
// the next offset); positions must be sorted.
double Junction::calcEntropy(const std::vector<int32_t>& p) {
    const size_t n = p.size();
    if (n <= 1) return 0;
    double sum = 0.0;
    int32_t last = p[0];
    uint32_t at = 0;
    for (size_t i = 0; i < n; i++) {
        at++;
        if (p[i] != last || i == n - 1) {
            const double pI = (double)at / (double)n;
            sum += pI * log2(pI);
            last = p[i];
            at = 0;
        }
    }
    return std::fabs(sum);
}

double Junction::getValueFromName(const std::string& name) const {
    if (name == "nb_raw_aln") return nbAlRaw;
    if (name == "nb_dist_aln") return nbAlDistinct;
    if (name == "nb_us_aln") return getNbUniquelySplicedAlignments();
    if (name == "nb_ms_aln") return nbAlMultiplySpliced;
    if (name == "nb_um_aln") return nbAlUniquelyMapped;
    if (name == "nb_mm_aln") return getNbMultiplyMappedAlignments();
    if (name == "nb_bpp_aln") return nbAlBamProperlyPaired;
    if (name == "nb_ppp_aln") return nbAlPortcullisProperlyPaired;
    if (name == "nb_rel_aln") return nbAlReliable;
    if (name == "rel2raw") return getReliable2RawAlignmentRatio();
    if (name == "nb_r1_pos") return nbAlR1Pos;
    if (name == "nb_r1_neg") return nbAlR1Neg;
    if (name == "nb_r2_pos") return nbAlR2Pos;
    if (name == "nb_r2_neg") return nbAlR2Neg;
    if (name == "entropy") return entropy;
    if (name == "mean_mismatches") return meanMismatches;
    if (name == "mean_readlen") return meanReadLength;
    if (name == "max_min_anc") return maxMinAnchor;
    if (name == "maxmmes") return maxMMES;
    if (name == "intron_score") return intronScore;
    if (name == "hamming5p") return hammingDistance5p;
    if (name == "hamming3p") return hammingDistance3p;
    if (name == "coding") return codingPotential;
    if (name == "pws") return positionWeightScore;
    if (name == "splice_sig") return splicingSignal;
    if (name == "uniq_junc") return uniqueJunction;
    if (name == "primary_junc") return primaryJunction;
    if (name == "nb_up_juncs") return nbUpstreamJunctions;
    if (name == "nb_down_juncs") return nbDownstreamJunctions;
    if (name == "dist_2_up_junc") return distanceToNextUpstreamJunction;
    if (name == "dist_2_down_junc") return distanceToNextDownstreamJunction;
    if (name == "dist_nearest_junc") return distanceToNearestJunction;
    if (name == "mm_score") return multipleMappingScore;
    if (name == "coverage") return coverage;
    if (name == "up_aln") return nbUpstreamFlankingAlignments;
    if (name == "down_aln") return nbDownstreamFlankingAlignments;
    if (name == "nb_samples") return nbSamples;
    if (name == "size") return getIntronSize();
    if (name == "score") return score;
    if (name == "suspicious") return suspicious;
    if (name == "pfp") return pfp;
    for (size_t k = 0; k < JAD_NAMES.size(); k++)
        if (name == JAD_NAMES[k]) return junctionAnchorDepth[k];
    throw JunctionException("Unrecognised junction property: " + name);
}

// one .tab row
std::ostream& operator<<(std::ostream& strm, const Junction& j) {
    strm << j.id << "\t" << *(j.intron) << "\t" << j.getIntronSize() << "\t" << j.leftAncStart << "\t" << j.rightAncEnd << "\t"
         << bam::strandToChar(j.readStrand) << "\t" << bam::strandToChar(j.ssStrand) << "\t"
         << bam::strandToChar(j.consensusStrand) << "\t" << j.da1 << "\t" << j.da2 << "\t" << cssToChar(j.canonicalSpliceSites)
         << "\t" << j.score << "\t" << j.suspicious << "\t" << j.pfp << "\t" << j.nbAlRaw << "\t" << j.nbAlDistinct << "\t"
         << j.getNbUniquelySplicedAlignments() << "\t" << j.nbAlMultiplySpliced << "\t" << j.nbAlUniquelyMapped << "\t"
         << j.getNbMultiplyMappedAlignments() << "\t" << j.nbAlBamProperlyPaired << "\t" << j.nbAlPortcullisProperlyPaired
         << "\t" << j.nbAlReliable << "\t" << j.getReliable2RawAlignmentRatio() << "\t" << j.nbAlR1Pos << "\t" << j.nbAlR1Neg
         << "\t" << j.nbAlR2Pos << "\t" << j.nbAlR2Neg << "\t" << j.entropy << "\t" << j.meanMismatches << "\t"
         << j.meanReadLength << "\t" << j.maxMinAnchor << "\t" << j.maxMMES << "\t" << j.intronScore << "\t"
         << j.hammingDistance5p << "\t" << j.hammingDistance3p << "\t" << j.codingPotential << "\t" << j.positionWeightScore
         << "\t" << j.splicingSignal << "\t" << j.uniqueJunction << "\t" << j.primaryJunction << "\t" << j.nbUpstreamJunctions
         << "\t" << j.nbDownstreamJunctions << "\t" << j.distanceToNextUpstreamJunction << "\t"
         << j.distanceToNextDownstreamJunction << "\t" << j.distanceToNearestJunction << "\t" << j.multipleMappingScore << "\t"
         << j.coverage << "\t" << j.nbUpstreamFlankingAlignments << "\t" << j.nbDownstreamFlankingAlignments << "\t"
         << j.nbSamples;
    for (size_t i = 0; i < Junction::JAD_NAMES.size(); i++) strm << "\t" << j.junctionAnchorDepth[i];
    return strm;
}

namespace {
inline void putU(std::string& o, uint64_t v) {
    char b[24];
    int n = 0;
    do {
        b[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) o.push_back(b[--n]);
}
inline void putI(std::string& o, int64_t v) {
    if (v < 0) {
        o.push_back('-');
        putU(o, (uint64_t)(-v));
    } else
        putU(o, (uint64_t)v);
}
inline void putG(std::string& o, double v) {  // default ostream formatting of a double: %g
    if (v == 0.0 && !std::signbit(v)) {
        o.push_back('0');
        return;
    }
    char b[40];
    const int n = snprintf(b, sizeof b, "%g", v);
    o.append(b, (size_t)n);
}
inline void tab(std::string& o) { o.push_back('\t'); }
}  // namespace

void Junction::appendTabRow(std::string& o) const {
    putU(o, id); tab(o);
    putI(o, intron->ref.index); tab(o); o += intron->ref.name; tab(o); putI(o, intron->ref.length); tab(o);
    putI(o, intron->start); tab(o); putI(o, intron->end); tab(o);
    putU(o, getIntronSize()); tab(o); putI(o, leftAncStart); tab(o); putI(o, rightAncEnd); tab(o);
    o.push_back(bam::strandToChar(readStrand)); tab(o); o.push_back(bam::strandToChar(ssStrand)); tab(o);
    o.push_back(bam::strandToChar(consensusStrand)); tab(o);
    o += da1; tab(o); o += da2; tab(o); o.push_back(cssToChar(canonicalSpliceSites)); tab(o);
    putG(o, score); tab(o); o.push_back(suspicious ? '1' : '0'); tab(o); o.push_back(pfp ? '1' : '0'); tab(o);
    putU(o, nbAlRaw); tab(o); putU(o, nbAlDistinct); tab(o); putU(o, getNbUniquelySplicedAlignments()); tab(o);
    putU(o, nbAlMultiplySpliced); tab(o); putU(o, nbAlUniquelyMapped); tab(o); putU(o, getNbMultiplyMappedAlignments()); tab(o);
    putU(o, nbAlBamProperlyPaired); tab(o); putU(o, nbAlPortcullisProperlyPaired); tab(o); putU(o, nbAlReliable); tab(o);
    putG(o, getReliable2RawAlignmentRatio()); tab(o);
    putU(o, nbAlR1Pos); tab(o); putU(o, nbAlR1Neg); tab(o); putU(o, nbAlR2Pos); tab(o); putU(o, nbAlR2Neg); tab(o);
    putG(o, entropy); tab(o); putG(o, meanMismatches); tab(o); putG(o, meanReadLength); tab(o);
    putU(o, maxMinAnchor); tab(o); putU(o, maxMMES); tab(o); putG(o, intronScore); tab(o);
    putU(o, hammingDistance5p); tab(o); putU(o, hammingDistance3p); tab(o);
    putG(o, codingPotential); tab(o); putG(o, positionWeightScore); tab(o); putG(o, splicingSignal); tab(o);
    o.push_back(uniqueJunction ? '1' : '0'); tab(o); o.push_back(primaryJunction ? '1' : '0'); tab(o);
    putU(o, nbUpstreamJunctions); tab(o); putU(o, nbDownstreamJunctions); tab(o);
    putU(o, distanceToNextUpstreamJunction); tab(o); putU(o, distanceToNextDownstreamJunction); tab(o);
    putU(o, distanceToNearestJunction); tab(o);
    putG(o, multipleMappingScore); tab(o); putG(o, coverage); tab(o);
    putU(o, nbUpstreamFlankingAlignments); tab(o); putU(o, nbDownstreamFlankingAlignments); tab(o); putU(o, nbSamples);
    for (size_t i = 0; i < JAD_NAMES.size(); i++) {
        tab(o);
        putU(o, junctionAnchorDepth[i]);
    }
}

void Junction::appendBedRow(std::string& o, const std::string& prefix, bool bedscore) const {
    const char strand = consensusStrand == Strand::UNKNOWN ? '.' : bam::strandToChar(consensusStrand);
    o += intron->ref.name; tab(o); putI(o, leftAncStart); tab(o); putI(o, rightAncEnd + 1); tab(o);
    o += prefix; o.push_back('_'); putU(o, id); tab(o);
    {   // fixed, precision 3 (the stream state outputBED sets)
        char b[64];
        const int n = snprintf(b, sizeof b, "%.3f", bedscore ? getScore() : (double)getNbSplicedAlignments());
        o.append(b, (size_t)n);
    }
    tab(o); o.push_back(strand); tab(o); putI(o, intron->start); tab(o); putI(o, intron->end + 1); tab(o);
    o += "255,0,0\t2\t";
    putI(o, intron->start - leftAncStart); o.push_back(','); putI(o, rightAncEnd - intron->end); tab(o);
    o += "0,"; putI(o, intron->end - leftAncStart + 1); o.push_back('\n');
}

static std::string join(const std::vector<std::string>& v, const char* sep) {
    std::string o;
    for (size_t i = 0; i < v.size(); i++) {
        if (i) o += sep;
        o += v[i];
    }
    return o;
}

std::string Junction::junctionOutputHeader() {
    return std::string("index\t") + Intron::locationOutputHeader() + "\tsize\tleft\tright\t" + join(STRAND_NAMES, "\t") +
           "\tss1\tss2\t" + join(METRIC_NAMES, "\t") + "\t" + join(JAD_NAMES, "\t");
}

void Junction::outputDescription(std::ostream& strm, const std::string& d) const {
    strm << "*** Intron ***" << d;
    if (intron) {
        intron->outputDescription(strm, d);
        strm << d << "Intron Size: " << getIntronSize();
    } else
        strm << "No location set";
    strm << d << "*** Anchors ***" << d << "Anchor limits: (" << leftAncStart << ", " << rightAncEnd << ")" << d
         << "Anchor sizes: (" << getLeftAnchorSize() << ", " << getRightAnchorSize() << ")" << d << "*** Strand ***" << d
         << "Reads Strand: " << bam::strandToString(readStrand) << d << "Splice Site Strand: " << bam::strandToString(ssStrand)
         << d << "Consensus Strand: " << bam::strandToString(consensusStrand) << d << "*** Confidence ***" << d
         << "Canonical?: " << std::boolalpha << cssToString(canonicalSpliceSites) << "; Sequences: (" << da1 << " " << da2
         << ")" << d << "Filter score: " << score << d
         << "Suspicious? (no anchors extending beyond first mismatch): " << std::boolalpha << suspicious << d
         << "Potential False Positive? (Suspicious and MaxMMES should have been greater given junction depth): "
         << std::boolalpha << pfp << d << "*** Alignment counts ***" << d << "# Total Spliced Alignments: " << nbAlRaw << d
         << "# Distinct Alignments: " << nbAlDistinct << d << "# Uniquely Spliced Alignments: "
         << getNbUniquelySplicedAlignments() << d << "# Multiply Spliced Alignments: " << nbAlMultiplySpliced << d
         << "# Uniquely Mapped Alignments: " << nbAlUniquelyMapped << d << "# Multiply Mapped Alignments: "
         << getNbMultiplyMappedAlignments() << d << "# Properly paired (bam flag): " << nbAlBamProperlyPaired << d
         << "# Properly paired (portcullis): " << nbAlPortcullisProperlyPaired << d << "# Reliable (MapQ >="
         << MAP_QUALITY_THRESHOLD << " + portcullis properly paired) Alignments: " << nbAlReliable << d << "# R1 (+"
         << nbAlR1Pos << ",-" << nbAlR1Neg << "); # R2 (+" << nbAlR2Pos << ",-" << nbAlR2Neg << ")" << d
         << "*** RNA seq derived Junction stats ***" << d << "Entropy: " << entropy << d << "Mean mismatches: "
         << meanMismatches << d << "Mean read length: " << meanReadLength << d << "MaxMinAnchor: " << maxMinAnchor << d
         << "MaxMMES: " << maxMMES << d << "Intron score: " << intronScore << d << "*** Genome derived Junction stats ***" << d
         << "Hamming Distance 5': " << hammingDistance5p << d << "Hamming Distance 3': " << hammingDistance3p << d
         << "*** Junction group properties ***" << d << "Unique Junction: " << std::boolalpha << uniqueJunction << d
         << "Primary Junction: " << std::boolalpha << primaryJunction << d << "# Upstream Junctions: " << nbUpstreamJunctions
         << d << "# Downstream Junctions: " << nbDownstreamJunctions << d << "Distance to next upstream junction: "
         << distanceToNextUpstreamJunction << d << "Distance to next downstream junction: " << distanceToNextDownstreamJunction
         << d << "Distance to nearest junction: " << distanceToNearestJunction << d << "*** Extra metrics ***" << d
         << "Multiple mapping score: " << multipleMappingScore << d << "Coverage: " << coverage << d
         << "# Upstream Non-Spliced Alignments: " << nbUpstreamFlankingAlignments << d
         << "# Downstream Non-Spliced Alignments: " << nbDownstreamFlankingAlignments << d << "# Samples: " << nbSamples;
}

void Junction::condensedOutputDescription(std::ostream& strm, const std::string& d) const {
    strm << "Strand: " << bam::strandToString(consensusStrand) << d << "Canonical?=" << cssToString(canonicalSpliceSites) << d
         << "Score=" << score << d << "NbAlignments=" << getNbSplicedAlignments() << d << "NbDistinct=" << nbAlDistinct << d
         << "NbReliable=" << nbAlReliable << d << "Entropy=" << entropy << d << "MaxMMES=" << maxMMES << d
         << "HammingDistance5=" << hammingDistance5p << d << "HammingDistance3=" << hammingDistance3p << d
         << "UniqueJunction=" << std::boolalpha << uniqueJunction << d << "PrimaryJunction=" << std::boolalpha
         << primaryJunction << d;
}

void Junction::outputIntronGFF(std::ostream& strm, const std::string& source) const {
    const char strand = consensusStrand == Strand::UNKNOWN ? '?' : bam::strandToChar(consensusStrand);
    const std::string juncId = "junc_" + std::to_string(id);
    strm << intron->ref.name << "\t" << source << "\t"
         << "intron"
         << "\t" << intron->start + 1 << "\t" << intron->end + 1 << "\t" << nbAlRaw << "\t" << strand << "\t"
         << "."
         << "\t"
         << "mult=" << nbAlRaw << ";"
         << "grp=" << juncId << ";"
         << "src=E";
    strm << "\n";
}

void Junction::outputJunctionGFF(std::ostream& strm, const std::string& source) const {
    const char strand = consensusStrand == Strand::UNKNOWN ? '?' : bam::strandToChar(consensusStrand);
    const std::string juncId = "junc_" + std::to_string(id);
    strm << intron->ref.name << "\t" << source << "\t"
         << "match"
         << "\t" << leftAncStart + 1 << "\t" << rightAncEnd + 1 << "\t"
         << "0.0"
         << "\t" << strand << "\t"
         << "."
         << "\t"
         << "ID=" << juncId << ";"
         << "Name=" << juncId << ";"
         << "Note=cov:" << nbAlRaw << "|rel:" << nbAlReliable << "|ent:" << std::setprecision(4) << entropy
         << std::setprecision(9) << "|maxmmes:" << maxMMES << "|ham:" << std::min(hammingDistance3p, hammingDistance5p) << ";"
         << "mult=" << nbAlRaw << ";"
         << "grp=" << juncId << ";"
         << "src=E;";
    condensedOutputDescription(strm, ";");
    strm << "\n";
    strm << intron->ref.name << "\t" << source << "\t"
         << "match_part"
         << "\t" << leftAncStart + 1 << "\t" << (intron->start) << "\t"
         << "0.0"
         << "\t" << strand << "\t"
         << "."
         << "\t"
         << "ID=" << juncId << "_left"
         << ";"
         << "Parent=" << juncId << "\n";
    strm << intron->ref.name << "\t" << source << "\t"
         << "match_part"
         << "\t" << (intron->end + 2) << "\t" << rightAncEnd + 1 << "\t"
         << "0.0"
         << "\t" << strand << "\t"
         << "."
         << "\t"
         << "ID=" << juncId << "_right"
         << ";"
         << "Parent=" << juncId << "\n";
}

void Junction::outputBED(std::ostream& strm, const std::string& prefix, bool bedscore) const {
    const char strand = consensusStrand == Strand::UNKNOWN ? '.' : bam::strandToChar(consensusStrand);
    const std::string juncId = prefix + "_" + std::to_string(id);
    const int32_t sz1 = intron->start - leftAncStart;
    const int32_t sz2 = rightAncEnd - intron->end;
    const std::string blockSizes = std::to_string(sz1) + "," + std::to_string(sz2);
    const std::string blockStarts = std::to_string(0) + "," + std::to_string(intron->end - leftAncStart + 1);
    strm << std::fixed << std::setprecision(3);
    // the ternary mixes double and uint32_t, so the depth is printed as a double ("135.000")
    strm << intron->ref.name << "\t" << leftAncStart << "\t" << rightAncEnd + 1 << "\t" << juncId << "\t"
         << (bedscore ? this->getScore() : this->getNbSplicedAlignments()) << "\t" << strand << "\t" << intron->start << "\t"
         << intron->end + 1 << "\t"
         << "255,0,0"
         << "\t"
         << "2"
         << "\t" << blockSizes << "\t" << blockStarts << "\n";
}

static std::vector<std::string> splitTabsCompress(const std::string& line) {
    // boost::split(..., is_any_of("\t"), token_compress_on): runs of tabs collapse
    std::vector<std::string> parts;
    std::string cur;
    bool lastTab = false;
    for (char c : line) {
        if (c == '\t') {
            if (!lastTab) {
                parts.push_back(cur);
                cur.clear();
            }
            lastTab = true;
        } else {
            cur.push_back(c);
            lastTab = false;
        }
    }
    parts.push_back(cur);
    return parts;
}

std::shared_ptr<Junction> Junction::parse(const std::string& line) {
    std::vector<std::string> p = splitTabsCompress(line);
    const size_t expected = 11 + STRAND_NAMES.size() + METRIC_NAMES.size() + JAD_NAMES.size();
    if (p.size() != expected)
        throw JunctionException("Could not parse line due to incorrect number of columns.  This is probably a version "
                                "mismatch.  Check file and portcullis versions.  Expected " +
                                std::to_string(expected) + " columns.  Found " + std::to_string(p.size()) + ".");
    auto in = std::make_shared<Intron>(bam::RefSeq(std::stoi(p[1]), p[2], std::stoi(p[3])), std::stoi(p[4]), std::stoi(p[5]));
    auto j = std::make_shared<Junction>(in, std::stoi(p[7]), std::stoi(p[8]));
    j->setId((uint32_t)std::stoul(p[0]));
    size_t i = 9;
    j->readStrand = bam::strandFromChar(p[i++][0]);
    j->ssStrand = bam::strandFromChar(p[i++][0]);
    j->consensusStrand = bam::strandFromChar(p[i++][0]);
    j->setDa1(p[i++]);
    j->setDa2(p[i++]);
    j->canonicalSpliceSites = cssFromChar(p[i++][0]);
    j->setScore(std::stod(p[i++]));
    j->setSuspicious(p[i++] == "1");
    j->setPotentialFalsePositive(p[i++] == "1");
    j->setNbSplicedAlignments((uint32_t)std::stoul(p[i++]));
    j->setNbDistinctAlignments((uint32_t)std::stoul(p[i++]));
    i++;
    j->setNbMultiplySplicedAlignments((uint32_t)std::stoul(p[i++]));
    j->setNbUniquelyMappedAlignments((uint32_t)std::stoul(p[i++]));
    i++;
    j->setNbBamProperlyPairedAlignments((uint32_t)std::stoul(p[i++]));
    j->setNbPortcullisProperlyPairedAlignments((uint32_t)std::stoul(p[i++]));
    j->setNbReliableAlignments((uint32_t)std::stoul(p[i++]));
    i++;
    j->setNbR1PosAlignments((uint32_t)std::stoul(p[i++]));
    j->setNbR1NegAlignments((uint32_t)std::stoul(p[i++]));
    j->setNbR2PosAlignments((uint32_t)std::stoul(p[i++]));
    j->setNbR2NegAlignments((uint32_t)std::stoul(p[i++]));
    j->setEntropy(std::stod(p[i++]));
    j->setMeanMismatches(std::stod(p[i++]));
    j->meanReadLength = std::stod(p[i++]);
    j->setMaxMinAnchor(std::stoi(p[i++]));
    j->setMaxMMES((uint32_t)std::stoul(p[i++]));
    j->setIntronScore(std::stod(p[i++]));
    j->setHammingDistance5p((uint32_t)std::stoul(p[i++]));
    j->setHammingDistance3p((uint32_t)std::stoul(p[i++]));
    j->setCodingPotential(std::stod(p[i++]));
    j->setPositionWeightScore(std::stod(p[i++]));
    j->setSplicingSignal(std::stod(p[i++]));
    j->setUniqueJunction(p[i++] == "1");
    j->setPrimaryJunction(p[i++] == "1");
    j->setNbUpstreamJunctions((uint32_t)std::stoul(p[i++]));
    j->setNbDownstreamJunctions((uint32_t)std::stoul(p[i++]));
    j->setDistanceToNextUpstreamJunction((uint32_t)std::stoul(p[i++]));
    j->setDistanceToNextDownstreamJunction((uint32_t)std::stoul(p[i++]));
    j->setDistanceToNearestJunction((uint32_t)std::stoul(p[i++]));
    j->setMultipleMappingScore(std::stod(p[i++]));
    j->setCoverage(std::stod(p[i++]));
    j->setNbUpstreamFlankingAlignments((uint32_t)std::stoul(p[i++]));
    j->setNbDownstreamFlankingAlignments((uint32_t)std::stoul(p[i++]));
    j->setNbSamples((uint32_t)std::stoul(p[i++]));
    for (size_t k = 0; k < JAD_NAMES.size(); k++) j->setJunctionAnchorDepth(k, (uint32_t)std::stoul(p[i + k]));
    return j;
}

}  // namespace portcullis
