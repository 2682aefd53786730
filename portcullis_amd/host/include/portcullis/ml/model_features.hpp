// ModelFeatures: the feature side of the filt stage (lib/include/portcullis/ml/model_features.hpp,
// lib/src/model_features.cc:42-235): intron-size threshold, the Markov models trained from junction sets, and the
// feature matrix.  Training walks genome windows on the host (it is a k-mer count); the matrix -- every junction's
// windows scored against six k-mer and two position models -- comes from the device (pjb_filt_features).
// The random-forest side (ranger) is not part of this build: juncs2FeatureVectors returns a plain row-major matrix
// whose columns are VAR_NAMES + Junction::JAD_NAMES.
#pragma once

#include <string>
#include <vector>

#include "../bam/genome_mapper.hpp"
#include "../junction.hpp"
#include "markov_model.hpp"

namespace portcullis {
namespace ml {

extern const std::vector<std::string> VAR_NAMES;  // "Genuine", "rna_usrs", ... "dna_ss" (model_features.hpp:45-60)

class ModelFeatures {
    bam::GenomeMapper* gmap = nullptr;
    std::string genomeFile;
    int device = 0;
    std::string oriented(const JunctionPtr& j, int start, int end) const;  // fetchBases (+ reverse complement on the negative strand)

public:
    uint32_t L95 = 0;
    KmerMarkovModel exonModel, intronModel, donorTModel, donorFModel, acceptorTModel, acceptorFModel;
    PosMarkovModel donorPWModel, acceptorPWModel;

    ModelFeatures() {}
    ~ModelFeatures();
    ModelFeatures(const ModelFeatures&) = delete;
    ModelFeatures& operator=(const ModelFeatures&) = delete;

    void setDevice(int d) { device = d; }
    bool isCodingPotentialModelEmpty() { return exonModel.size() == 0 || intronModel.size() == 0; }
    bool isPWModelEmpty() { return donorPWModel.size() == 0 || acceptorPWModel.size() == 0; }
    void initGenomeMapper(const std::string& genomeFile);
    uint32_t calcIntronThreshold(const JunctionList& juncs);
    void trainCodingPotentialModel(const JunctionList& in);
    void trainSplicingModels(const JunctionList& pass, const JunctionList& fail);
    static std::vector<std::string> featureNames();
    // row-major [x.size()][featureNames().size()] -- ModelFeatures::setRow for every junction of x
    std::vector<double> juncs2FeatureVectors(const JunctionList& x);
};

}  // namespace ml
}  // namespace portcullis
