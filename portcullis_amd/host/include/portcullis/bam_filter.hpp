// BamFilter: the `bamfilt` stage (src/bam_filter.{hpp,cc} of the reference): keeps the alignments of a BAM file that
// are unspliced or supported by a junction of the filtered junction file.  Same class surface (constructor, setters,
// filter(), main()); the per-alignment decision (BamFilter::containsJunctionInSystem / clipMSR, src/bam_filter.cc:75-150)
// runs on the GPU behind pjb_filter_batch, the BGZF output is written by portcullis::bam::BamWriter.
#pragma once

#include <string>

#include "bam/bam_master.hpp"

namespace portcullis {

struct BamFilterException : public PortcullisException {
    explicit BamFilterException(const std::string& m) : PortcullisException(m) {}
};

enum class ClipMode { HARD, SOFT, COMPLETE };  // src/bam_filter.hpp:50-54

inline std::string clipToString(ClipMode cm) { return cm == ClipMode::HARD ? "HARD" : cm == ClipMode::SOFT ? "SOFT" : "COMPLETE"; }
ClipMode clipFromString(const std::string& cm);  // throws BamFilterException("Unrecognised clip mode: ...")

class BamFilter {
    std::string junctionFile, bamFile, outputBam;
    ClipMode clipMode = ClipMode::HARD;
    bool saveMSRs = false, useCsi = false, verbose = false;
    int threads = 1, device = 0;
    // what the last filter() did (the reference prints these, src/bam_filter.cc:232-233)
    uint64_t nbReadsIn = 0, nbReadsOut = 0, nbReadsModifiedOut = 0;

public:
    BamFilter(const std::string& junctionFile, const std::string& bamFile, const std::string& outputBam);

    std::string getBamFile() const { return bamFile; }
    void setBamFile(const std::string& v) { bamFile = v; }
    std::string getJunctionFile() const { return junctionFile; }
    void setJunctionFile(const std::string& v) { junctionFile = v; }
    std::string getOutputBam() const { return outputBam; }
    void setOutputBam(const std::string& v) { outputBam = v; }
    ClipMode getClipMode() const { return clipMode; }
    void setClipMode(ClipMode v) { clipMode = v; }
    bool isSaveMSRs() const { return saveMSRs; }
    void setSaveMSRs(bool v) { saveMSRs = v; }
    bool isUseCsi() const { return useCsi; }
    void setUseCsi(bool v) { useCsi = v; }
    bool isVerbose() const { return verbose; }
    void setVerbose(bool v) { verbose = v; }
    void setThreads(int t) { threads = t < 1 ? 1 : t; }  // BGZF inflate of the input is one thread; deflate of the output uses these
    void setDevice(int d) { device = d; }
    uint64_t getNbReadsIn() const { return nbReadsIn; }
    uint64_t getNbReadsOut() const { return nbReadsOut; }
    uint64_t getNbReadsModifiedOut() const { return nbReadsModifiedOut; }

    void filter();

    static std::string helpMessage();
    static int main(int argc, char* argv[]);
};

}  // namespace portcullis
