// PreparedFiles: the path contract of a `portcullis prep` output directory (src/prepare.hpp:84-145
// and PreparedFiles::valid, src/prepare.cc:57-75).  `prep` itself is outside this library.
#pragma once

#include <string>
#include <sys/stat.h>

#include "bam/bam_master.hpp"

namespace portcullis {

struct PrepareException : public PortcullisException {
    explicit PrepareException(const std::string& m) : PortcullisException(m) {}
};

class PreparedFiles {
    std::string prepDir;

    static bool exists(const std::string& p) {
        struct stat st;
        return lstat(p.c_str(), &st) == 0;  // a (possibly dangling) symlink counts, as in the reference
    }

public:
    PreparedFiles() = default;
    explicit PreparedFiles(const std::string& dir) : prepDir(dir) {}

    const std::string& getPrepDir() const { return prepDir; }
    std::string getUnsortedBamFilePath() const { return prepDir + "/portcullis.unsorted.alignments.bam"; }
    std::string getSortedBamFilePath() const { return prepDir + "/portcullis.sorted.alignments.bam"; }
    std::string getBamIndexFilePath(bool useCsi) const { return getSortedBamFilePath() + (useCsi ? ".csi" : ".bai"); }
    std::string getGenomeFilePath() const { return prepDir + "/portcullis.genome.fa"; }
    std::string getGenomeIndexFilePath() const { return getGenomeFilePath() + ".fai"; }

    bool valid(bool useCsi) const {
        if (!exists(getSortedBamFilePath())) throw PrepareException("Could not find sorted BAM files at: " + getSortedBamFilePath());
        if (!exists(getBamIndexFilePath(useCsi))) throw PrepareException("Could not find BAM index at: " + getBamIndexFilePath(useCsi));
        if (!exists(getGenomeFilePath())) throw PrepareException("Could not find genome file at: " + getGenomeFilePath());
        if (!exists(getGenomeIndexFilePath())) throw PrepareException("Could not find genome index at: " + getGenomeIndexFilePath());
        return true;
    }
};

}  // namespace portcullis
