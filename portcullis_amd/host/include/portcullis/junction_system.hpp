// JunctionSystem: the ordered collection of junctions of a run, the cross-junction pass and the
// writers.  API of lib/include/portcullis/junction_system.hpp:48-181 of the reference for the junc
// path.  The bulk route is portcullis::JunctionBuilder (alignments go to the GPU in batches); the reference's
// per-alignment entry point addJunctions(const BamAlignment&) (junction_system.hpp:128-132) is kept for library
// callers: it queues the alignment, and finish() runs the device over everything queued.
#pragma once

#include <ostream>
#include <string>
#include <unordered_map>
#include <utility>

#include "bam/bam_reader.hpp"
#include "bam/genome_mapper.hpp"
#include "junction.hpp"

namespace portcullis {

using bam::Orientation;
using bam::Strandedness;

class JunctionSystem {
private:
    std::unordered_map<Intron, JunctionPtr, IntronHasher> distinctJunctions;
    JunctionList junctionList;
    std::shared_ptr<bam::RefSeqPtrList> refs;
    int32_t minQueryLength = 0;
    double meanQueryLength = 0.0;
    int32_t maxQueryLength = 0;
    std::unordered_map<int32_t, bam::ReadBatch> pending;  // addJunctions(): alignments queued per target, in arrival order

protected:
    size_t createJunctionGroup(size_t index, std::vector<JunctionPtr>& group);
    void findJunctions(int32_t refId, JunctionList& subset);

public:
    static std::string version;

    JunctionSystem() = default;
    explicit JunctionSystem(std::shared_ptr<bam::RefSeqPtrList> refs) : refs(refs) {}
    explicit JunctionSystem(const std::string& junctionFile) { load(junctionFile); }
    explicit JunctionSystem(JunctionList& jl) {
        for (auto& j : jl) addJunction(j);
    }

    const JunctionList& getJunctions() const { return junctionList; }
    size_t size() const { return distinctJunctions.size(); }
    void reserve(size_t n);  // room for n junctions (list and intron map) before a run of addJunction / append
    bool empty() const { return junctionList.empty(); }

    void setRefs(std::shared_ptr<bam::RefSeqPtrList> r) { refs = r; }
    std::shared_ptr<bam::RefSeqPtrList> getRefs() const { return refs; }

    void setQueryLengthStats(int32_t min, double mean, int32_t max) {
        minQueryLength = min;
        meanQueryLength = mean;
        maxQueryLength = max;
    }
    int32_t getMinQueryLength() const { return minQueryLength; }
    double getMeanQueryLength() const { return meanQueryLength; }
    int32_t getMaxQueryLength() const { return maxQueryLength; }

    // JunctionSystem::addJunctions(const BamAlignment&) of the reference (lib/src/junction_system.cc:140-210) found the
    // alignment's junctions on the spot; here the alignment is queued (alignments of one target must arrive in
    // coordinate order, as a region iterator delivers them) and the return value is the reference's: whether the
    // alignment is spliced.  Call finish() once every alignment has been added.
    bool addJunctions(const bam::BamAlignment& al);
    bool addJunctions(const bam::BamAlignment& al, size_t startOp, int32_t offset) { (void)startOp; (void)offset; return addJunctions(al); }
    // Runs the device path over the queued alignments, target by target (the genome of each comes from `gmap`), and
    // fills the junction list: what the reference's callers did with Junction::calcMetrics(orientation) +
    // processJunctionWindow(gmap) per junction (lib/include/portcullis/junction.hpp:1038-1100).  Returns the number of
    // junctions added.  Throws JunctionException for the data conditions under which the reference throws.
    size_t finish(const bam::GenomeMapper& gmap, Orientation orientation = Orientation::UNKNOWN, int device = 0);

    void addJunction(JunctionPtr j);
    void append(JunctionSystem& other);
    void absorb(JunctionSystem& other);  // append(), moving the other system's intron map over (the other keeps its list only)
    // Append junctions delivered by the device path (pjb_collect rows of one or more contigs).
    void appendRows(const pjb_junction_row* rows, size_t n);

    // cross-junction statistics: unique / primary junction groups, neighbour distances, pfp
    void calcJunctionStats();
    std::pair<Orientation, Strandedness> determineStrandedness(bool verbose) const;

    void sort();
    void index();

    void saveAll(const std::string& outputPrefix, const std::string& source);
    void saveAll(const std::string& outputPrefix, const std::string& source, bool bedscore, bool outputExonGFF,
                 bool outputIntronGFF);

    void outputDescription(std::ostream& strm);
    friend std::ostream& operator<<(std::ostream& strm, const JunctionSystem& js);
    void writeExonGFF(std::ostream& strm, const std::string& source);
    void writeIntronGFF(std::ostream& strm, const std::string& source);
    void outputBED(const std::string& path, CanonicalSS type, const std::string& prefix, bool bedscore);
    void outputBED(std::ostream& strm, CanonicalSS type, const std::string& prefix, bool bedscore);

    void load(const std::string& junctionTabFile) { load(junctionTabFile, false); }
    void load(const std::string& junctionTabFile, bool simple);

    JunctionPtr getJunctionAt(uint32_t index) const { return junctionList[index]; }
    JunctionPtr getJunction(const Intron& intron) const;
};

}  // namespace portcullis
