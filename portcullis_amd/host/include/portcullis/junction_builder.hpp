// JunctionBuilder: the `junc` stage driver.  Same construction / setter / process() surface as
// portcullis::JunctionBuilder (src/junction_builder.hpp:82-259 of the reference); the work that
// the reference does per alignment on CPU threads (findJuncs, src/junction_builder.cc:314-357)
// is streamed in batches to the MI355X through the C ABI in include/portcullis_amd.h.
#pragma once

#include <functional>
#include <future>
#include <string>
#include <vector>

#include "bam/bam_reader.hpp"
#include "bam/genome_mapper.hpp"
#include "junction_system.hpp"
#include "prepared_files.hpp"

struct pjb_junction_row;

namespace portcullis {

const std::string DEFAULT_JUNC_OUTPUT = "portcullis_junc/portcullis";
const std::string DEFAULT_JUNC_SOURCE = "portcullis";
const uint16_t DEFAULT_JUNC_THREADS = 1;

struct JunctionBuilderException : public PortcullisException {
    explicit JunctionBuilderException(const std::string& m) : PortcullisException(m) {}
};

// per target sequence outcome (RegionResult, src/junction_builder.hpp:62-76)
struct RegionResult {
    uint64_t splicedCount = 0;
    uint64_t unsplicedCount = 0;
    uint64_t sumQueryLengths = 0;
    int32_t minQueryLength = INT32_MAX;
    int32_t maxQueryLength = 0;
    std::string name;
    JunctionSystem js;
    size_t rowBase = 0;  // --extra: where this target's rows start in the device context's row table
};

class JunctionBuilder {
    PreparedFiles prepData;
    std::string outputDir;
    std::string outputPrefix;
    uint16_t threads = 1;
    bam::Strandedness strandSpecific = bam::Strandedness::UNKNOWN;
    bam::Orientation orientation = bam::Orientation::UNKNOWN;
    bool extra = false;
    bool separate = false;
    bool useCsi = false;
    bool outputExonGFF = false;
    bool outputIntronGFF = false;
    std::string source = DEFAULT_JUNC_SOURCE;
    bool verbose = false;
    int devices = 0;               // 0 = every visible GPU
    int hostThreads = 0;           // 0 = use `threads`; otherwise total host decode threads
    size_t batchRecords = 1 << 20; // alignments per batch sent to the device
    int innerThreads = 1;          // decode threads inside one target sequence (set by findJunctions)
    std::vector<int> transferRanks;  // per target: its place in the order the targets' file bytes should cross (set by findJunctions)
    int transferRank(int32_t tid) const { return tid >= 0 && (size_t)tid < transferRanks.size() ? transferRanks[(size_t)tid] : 1 << 30; }
    std::shared_ptr<class PinnedPool> pinnedPool;  // ring of page-locked pieces for the file bytes of large device-ingest runs
    std::shared_ptr<class PinnedPool> genomePool;  // a few page-locked buffers for the FASTA bytes of the target sequences (same runs)
    size_t pieceMinTarget = 0;                     // targets with fewer bytes go over in one (pageable) block
    bool backgroundTeardown = false;               // contexts and page-locked rings are taken down beside the merge and the writers (the program sets it: it leaves with _exit; a library caller that returns from main must not have a thread inside the runtime then)
    bool directPieces = true;                      // the threads that read the file hand the pieces to the device themselves (PORTCULLIS_DIRECT_PIECES=0: through the device thread's queue)
    bool deviceIngest = true;      // BGZF inflate + BAM record parse on the GPU (pjb_submit_bam); false: host threads

    std::shared_future<int> deviceCount;  // pjb_device_count() evaluated in the background

    JunctionSystem junctionSystem;
    std::shared_ptr<bam::RefSeqPtrList> refs;
    std::shared_ptr<bam::RefSeqPtrIndexMap> refMap;
    std::vector<RegionResult> results;

protected:
    void findJunctions();
    // decode one target sequence (worker body) and feed it to the thread that owns the GPU context
    void findJuncs(class DeviceThread& device, bam::BamReader& reader, bam::GenomeMapper& gmap, int32_t seq);
    // group chains (the program's default for large inputs): a worker does not wait for its target's chain -- the group is queued when its
    // last member has been asked for -- and findJunctions completes the targets once every one of them has been asked for
    void completeTarget(int32_t seq, struct DeferredTarget& dt);
    std::vector<std::shared_ptr<struct DeferredTarget>> deferredTargets;
    std::mutex deferredMu;

public:
    JunctionBuilder(const std::string& prepDir, const std::string& output);
    virtual ~JunctionBuilder() = default;

    std::string getRefName(int32_t seqId) { return refs->at((size_t)seqId)->name; }
    PreparedFiles& getPreparedFiles() { return prepData; }
    JunctionSystem& getJunctionSystem() { return junctionSystem; }

    bool isExtra() const { return extra; }
    void setExtra(bool v) { extra = v; }
    uint16_t getThreads() const { return threads; }
    void setThreads(uint16_t v) { threads = v; }
    bool isVerbose() const { return verbose; }
    void setVerbose(bool v) { verbose = v; }
    std::string getUnsplicedBamFile() const { return (outputDir.empty() ? std::string(".") : outputDir) + "/" + outputPrefix + ".unspliced.bam"; }
    std::string getSplicedBamFile() const { return (outputDir.empty() ? std::string(".") : outputDir) + "/" + outputPrefix + ".spliced.bam"; }
    std::string getUnmappedBamFile() const { return (outputDir.empty() ? std::string(".") : outputDir) + "/" + outputPrefix + ".unmapped.bam"; }
    // --separate (src/junction_builder.cc:152-226): the prepared BAM split into spliced / unspliced / unmapped files
    void separateBams();
    bool isSeparate() const { return separate; }
    void setSeparate(bool v) { separate = v; }
    std::string getSource() const { return source; }
    void setSource(const std::string& v) { source = v; }
    bam::Strandedness getStrandSpecific() const { return strandSpecific; }
    void setStrandSpecific(bam::Strandedness v) { strandSpecific = v; }
    bam::Orientation getOrientation() const { return orientation; }
    void setOrientation(bam::Orientation v) { orientation = v; }
    bool isUseCsi() const { return useCsi; }
    void setUseCsi(bool v) { useCsi = v; }
    bool isOutputExonGFF() const { return outputExonGFF; }
    void setOutputExonGFF(bool v) { outputExonGFF = v; }
    bool isOutputIntronGFF() const { return outputIntronGFF; }
    void setOutputIntronGFF(bool v) { outputIntronGFF = v; }
    // additions of this implementation
    int getDevices() const { return devices; }
    void setDevices(int n) { devices = n; }
    void setBatchRecords(size_t n) { batchRecords = n ? n : 1; }
    void setDeviceIngest(bool on) { deviceIngest = on; }
    bool isDeviceIngest() const { return deviceIngest; }
    // total host decode threads, independent of the number of target sequences (the reference's
    // --threads is capped at the number of targets; this one is not)
    void setHostThreads(int n) { hostThreads = n; }

    void process();

    static std::string title() { return "Portcullis Junction Builder Mode Help"; }
    static std::string description() {
        return std::string("Analyses all potential junctions found in the input BAM file.\n") +
               "Run \"portcullis prep ...\" to generate data suitable for junction finding\n" +
               "before running \"portcullis junc ...\"";
    }
    static std::string usage() { return "portcullis junc [options] <prep_data_dir>"; }
    static int main(int argc, char* argv[]);
};

}  // namespace portcullis
