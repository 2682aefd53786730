// The filt stage's feature side driven as the reference's JunctionFilter drives it (src/junction_filter.cc: load the
// .tab, pick a positive and a negative set, ModelFeatures::calcIntronThreshold / trainCodingPotentialModel /
// trainSplicingModels / juncs2FeatureVectors).  Here the sets are simply "nb_raw >= 3" and the rest.
//   model_features <genome.fa> <junctions.tab> <out.txt>
#include <portcullis/junction_system.hpp>
#include <portcullis/ml/model_features.hpp>

#include <cstdio>
#include <iostream>

using namespace portcullis;

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    try {
        JunctionSystem js(argv[2]);
        JunctionList all = js.getJunctions(), pass, fail;
        for (auto& j : all) (j->getNbSplicedAlignments() >= 3 ? pass : fail).push_back(j);
        ml::ModelFeatures mf;
        mf.initGenomeMapper(argv[1]);
        mf.calcIntronThreshold(all);
        mf.trainCodingPotentialModel(pass);
        mf.trainSplicingModels(pass, fail);
        std::vector<double> m = mf.juncs2FeatureVectors(all);
        const size_t nf = ml::ModelFeatures::featureNames().size();
        FILE* f = fopen(argv[3], "w");
        fprintf(f, "# L95=%u exon=%zu intron=%zu donorPW=%zu\n", mf.L95, mf.exonModel.size(), mf.intronModel.size(), mf.donorPWModel.size());
        for (size_t i = 0; i < all.size(); i++) {
            for (size_t k = 0; k < nf; k++) fprintf(f, "%s%.17g", k ? "\t" : "", m[i * nf + k]);
            fprintf(f, "\n");
        }
        fclose(f);
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
