// ModelFeatures: see portcullis/ml/model_features.hpp.  Line numbers refer to lib/src/model_features.cc of the reference.
#include <portcullis/ml/model_features.hpp>

#include <algorithm>
#include <cstring>

#include <portcullis/seq_utils.hpp>

#include "../../../include/portcullis_amd.h"

namespace portcullis {
namespace ml {

const std::vector<std::string> VAR_NAMES = {"Genuine",       "rna_usrs",    "rna_dist",      "rna_rel",    "rna_entropy",
                                            "rna_rel2raw",   "rna_maxminanc", "rna_maxmmes", "rna_missmatch", "rna_intron",
                                            "dna_minhamm",   "dna_coding",  "dna_pws",       "dna_ss"};

ModelFeatures::~ModelFeatures() { delete gmap; }

void ModelFeatures::initGenomeMapper(const std::string& file) {  // :60-65
    delete gmap;
    genomeFile = file;
    gmap = new bam::GenomeMapper(file);
    gmap->loadFastaIndex();
}

uint32_t ModelFeatures::calcIntronThreshold(const JunctionList& juncs) {  // :67-75
    std::vector<uint32_t> sizes;
    for (const auto& j : juncs) sizes.push_back(j->getIntronSize());
    std::sort(sizes.begin(), sizes.end());
    L95 = sizes[(size_t)((double)sizes.size() * 0.95)];
    return L95;
}

// gmap.fetchBases(...) and SeqUtils::reverseComplement when the consensus strand is negative.  The bases are upper-cased
// first (the reference indexes its complement table with whatever case the FASTA has: out of bounds for lower case).
std::string ModelFeatures::oriented(const JunctionPtr& j, int start, int end) const {
    if (!gmap) throw JunctionException("ModelFeatures: initGenomeMapper was not called");
    std::string s = gmap->fetchBases(j->getIntron()->ref.name.c_str(), start, end);
    for (auto& c : s)
        if (c >= 'a' && c <= 'z') c = (char)(c - 32);
    if (j->getConsensusStrand() == bam::Strand::NEGATIVE) s = SeqUtils::reverseComplement(s);
    return s;
}

void ModelFeatures::trainCodingPotentialModel(const JunctionList& in) {  // :77-112
    std::vector<std::string> exons, introns;
    for (const auto& j : in) {
        const int s = j->getIntron()->start, e = j->getIntron()->end;
        exons.push_back(oriented(j, s - 202, s - 2));
        introns.push_back(oriented(j, s, e));
        exons.push_back(oriented(j, e + 1, e + 201));
    }
    exonModel.train(exons, 5);
    intronModel.train(introns, 5);
}

void ModelFeatures::trainSplicingModels(const JunctionList& pass, const JunctionList& fail) {  // :114-158
    std::vector<std::string> donors, acceptors;
    auto collect = [&](const JunctionList& l) {
        donors.clear();
        acceptors.clear();
        for (const auto& j : l) {
            const int s = j->getIntron()->start, e = j->getIntron()->end;
            std::string left = oriented(j, s - 3, s + 20), right = oriented(j, e - 20, e + 2);
            const bool neg = j->getConsensusStrand() == bam::Strand::NEGATIVE;
            donors.push_back(neg ? right : left);
            acceptors.push_back(neg ? left : right);
        }
    };
    collect(pass);
    donorPWModel.train(donors, 1);
    acceptorPWModel.train(acceptors, 1);
    donorTModel.train(donors, 5);
    acceptorTModel.train(acceptors, 5);
    collect(fail);
    donorFModel.train(donors, 5);
    acceptorFModel.train(acceptors, 5);
}

std::vector<std::string> ModelFeatures::featureNames() {
    std::vector<std::string> n = VAR_NAMES;
    n.insert(n.end(), Junction::JAD_NAMES.begin(), Junction::JAD_NAMES.end());
    return n;
}

std::vector<double> ModelFeatures::juncs2FeatureVectors(const JunctionList& x) {  // :214-230 with setRow :161-212
    if (!gmap) throw JunctionException("ModelFeatures: initGenomeMapper was not called");
    std::vector<double> out(x.size() * PJB_N_FEATURES, 0.0);
    if (x.empty()) return out;
    // one context, the genomes of the targets the junctions lie on
    pjb_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = PJB_ABI_VERSION;
    cfg.device = device;
    cfg.orientation = PJB_OR_UNKNOWN;
    cfg.strandedness = PJB_SS_UNKNOWN;
    pjb_ctx* ctx = nullptr;
    if (pjb_create(&ctx, &cfg) != PJB_OK) throw JunctionException(std::string("pjb_create: ") + pjb_last_error(nullptr));
    struct Closer {
        pjb_ctx* c;
        ~Closer() { pjb_destroy(c); }
    } closer{ctx};
    int32_t maxRef = 0;
    for (const auto& j : x) maxRef = std::max(maxRef, j->getIntron()->ref.index);
    std::vector<int32_t> lens((size_t)maxRef + 1, 0);
    std::vector<std::string> names((size_t)maxRef + 1);
    for (const auto& j : x) {
        lens[(size_t)j->getIntron()->ref.index] = j->getIntron()->ref.length;
        names[(size_t)j->getIntron()->ref.index] = j->getIntron()->ref.name;
    }
    auto check = [&](int rc, const char* what) {
        if (rc != PJB_OK) throw JunctionException(std::string(what) + ": " + pjb_last_error(ctx));
    };
    check(pjb_set_refs(ctx, (int32_t)lens.size(), lens.data()), "pjb_set_refs");
    for (size_t t = 0; t < lens.size(); t++) {
        if (names[t].empty()) continue;
        const std::string contig = gmap->fetchContig(names[t]);
        check(pjb_upload_contig(ctx, (int32_t)t, (const uint8_t*)contig.data(), (int64_t)contig.size()), "pjb_upload_contig");
    }
    std::vector<pjb_junction_row> rows(x.size());
    memset(rows.data(), 0, rows.size() * sizeof(pjb_junction_row));
    for (size_t i = 0; i < x.size(); i++) {
        const Junction& j = *x[i];
        pjb_junction_row& r = rows[i];
        r.refid = j.getIntron()->ref.index;
        r.start = j.getIntron()->start;
        r.end = j.getIntron()->end;
        r.left = j.getLeftAncStart();
        r.right = j.getRightAncEnd();
        r.cons_strand = (uint8_t)j.getConsensusStrand();
        r.nb_raw = j.getNbSplicedAlignments();
        r.nb_dist = j.getNbDistinctAlignments();
        r.nb_ms = j.getNbMultiplySplicedAlignments();
        r.nb_rel = j.getNbReliableAlignments();
        r.entropy = j.getEntropy();
        r.max_min_anc = j.getMaxMinAnchor();
        r.maxmmes = j.getMaxMMES();
        r.hamming5p = j.getHammingDistance5p();
        r.hamming3p = j.getHammingDistance3p();
        for (int k = 0; k < 20; k++) r.jad[k] = j.getJunctionAnchorDepth((size_t)k);
    }
    pjb_markov_models m;
    memset(&m, 0, sizeof m);
    m.exon = exonModel.table();
    m.intron = intronModel.table();
    m.donor_t = donorTModel.table();
    m.donor_f = donorFModel.table();
    m.acceptor_t = acceptorTModel.table();
    m.acceptor_f = acceptorFModel.table();
    m.donor_pw = donorPWModel.table();
    m.acceptor_pw = acceptorPWModel.table();
    m.exon_size = (int32_t)exonModel.size();
    m.intron_size = (int32_t)intronModel.size();
    m.donor_pw_size = (int32_t)donorPWModel.size();
    m.acceptor_pw_size = (int32_t)acceptorPWModel.size();
    check(pjb_filt_features(ctx, rows.data(), (int64_t)rows.size(), x[0]->getMeanReadLength(), L95, &m, out.data()), "pjb_filt_features");
    for (size_t i = 0; i < x.size(); i++) {
        out[i * PJB_N_FEATURES + 0] = x[i]->isGenuine() ? 1.0 : 0.0;
        out[i * PJB_N_FEATURES + 8] = x[i]->getMeanMismatches();  // the junction's own value (a row parsed from a .tab has no integer sum)
    }
    return out;
}

}  // namespace ml
}  // namespace portcullis
