// KmerMarkovModel / PosMarkovModel of the filt stage (lib/include/portcullis/ml/markov_model.hpp,
// lib/src/markov_model.cc): same interface (train / getScore / size / getOrder), stored as dense tables over the
// alphabet SeqUtils::makeClean leaves (A C G T N) instead of nested hash maps of strings -- which is also the form
// pjb_filt_features takes them in.
#pragma once

#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

namespace portcullis {
namespace ml {

inline int cleanCode(char c) {  // SeqUtils::makeClean, lib/include/portcullis/seq_utils.hpp:54-60
    if (c >= 'a' && c <= 'z') c = (char)(c - 32);
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
}

class KmerMarkovModel {
    uint16_t order = 5;
    std::vector<double> tab;    // [5^order][5]
    std::vector<uint8_t> seen;  // contexts the reference's map would hold
    size_t nCtx() const {
        size_t r = 1;
        for (int k = 0; k < order; k++) r *= 5;
        return r;
    }
    size_t ctx(const std::string& s, size_t at) const {
        size_t x = 0;
        for (size_t k = 0; k < order; k++) x = x * 5 + (size_t)cleanCode(s[at + k]);
        return x;
    }

public:
    KmerMarkovModel() {}
    explicit KmerMarkovModel(uint16_t o) : order(o) {}
    KmerMarkovModel(const std::vector<std::string>& input, uint16_t o) { train(input, o); }
    void train(const std::vector<std::string>& input, uint16_t o) {  // markov_model.cc:31-54
        order = o;
        tab.assign(nCtx() * 5, 0.0);
        seen.assign(nCtx(), 0);
        for (const auto& s : input)
            if (s.size() > (size_t)order + 1)
                for (size_t i = order; i < s.size(); i++) {
                    const size_t c = ctx(s, i - order);
                    tab[c * 5 + (size_t)cleanCode(s[i])] += 1.0;
                    seen[c] = 1;
                }
        for (size_t c = 0; c < seen.size(); c++) {
            double sum = 0;
            for (int k = 0; k < 5; k++) sum += tab[c * 5 + k];
            if (sum > 0)
                for (int k = 0; k < 5; k++) tab[c * 5 + k] /= sum;
        }
    }
    uint16_t getOrder() const { return order; }
    size_t size() const {
        size_t n = 0;
        for (uint8_t v : seen) n += v;
        return n;
    }
    double getScore(const std::string& s) {  // markov_model.cc:57-78
        if (tab.empty()) {
            tab.assign(nCtx() * 5, 0.0);
            seen.assign(nCtx(), 0);
        }
        double score = 1.0;
        uint32_t no_count = 0;
        for (size_t i = order; i < s.size(); i++) {
            const size_t c = ctx(s, i - order);
            seen[c] = 1;  // operator[] of the reference's map inserts what it looks up
            const double m = tab[c * 5 + (size_t)cleanCode(s[i])];
            if (m != 0.0) score *= m;
            else no_count++;
        }
        if (score == 0.0) return -100.0;
        if (no_count > 2) score /= ((double)no_count * 0.5);
        return std::log(score);
    }
    const double* table() const { return tab.empty() ? nullptr : tab.data(); }  // [5^order * 5], nullptr = never trained
};

class PosMarkovModel {
public:
    static const size_t LEN = 32;  // PJB_PW_LEN

private:
    uint16_t order = 1;
    std::vector<double> tab;  // [LEN][5]
    std::vector<uint8_t> seen;

public:
    PosMarkovModel() {}
    explicit PosMarkovModel(uint16_t o) : order(o) {}
    void train(const std::vector<std::string>& input, uint16_t o) {  // markov_model.cc:80-98
        order = o;
        tab.assign(LEN * 5, 0.0);
        seen.assign(LEN, 0);
        for (const auto& s : input)
            for (size_t i = order; i < s.size() && i < LEN; i++) {
                tab[i * 5 + (size_t)cleanCode(s[i])] += 1.0;
                seen[i] = 1;
            }
        for (size_t i = 0; i < LEN; i++) {
            double sum = 0;
            for (int k = 0; k < 5; k++) sum += tab[i * 5 + k];
            if (sum > 0)
                for (int k = 0; k < 5; k++) tab[i * 5 + k] /= sum;
        }
    }
    uint16_t getOrder() const { return order; }
    size_t size() const {
        size_t n = 0;
        for (uint8_t v : seen) n += v;
        return n;
    }
    double getScore(const std::string& s) {  // markov_model.cc:101-115
        if (tab.empty()) {
            tab.assign(LEN * 5, 0.0);
            seen.assign(LEN, 0);
        }
        double score = 1.0;
        for (size_t i = order; i < s.size() && i < LEN; i++) {
            seen[i] = 1;
            score *= tab[i * 5 + (size_t)cleanCode(s[i])];
        }
        if (score == 0.0) return -300.0;
        return std::log(score);
    }
    const double* table() const { return tab.empty() ? nullptr : tab.data(); }
};

}  // namespace ml
}  // namespace portcullis
