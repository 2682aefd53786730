// JunctionSystem::absorb (the merge of the per-target systems in findJunctions) against append, which mirrors the reference's
// JunctionSystem::append (lib/src/junction_system.cc:110-120): same list, same size(), same getJunction() answers -- also when the two
// systems share an intron (the absorbed system's junction takes the entry over, as addJunction would) -- and the same neighbour statistics
// from calcJunctionStats (lib/src/junction_system.cc:140-210) on the merged list.
#include <portcullis/junction_system.hpp>
#include "portcullis_amd.h"
#include <cstring>
#include <iostream>
#include <random>
#include <sstream>
using namespace portcullis;

static pjb_junction_row row(int32_t refid, int32_t start, int32_t end, uint32_t raw) {
    pjb_junction_row r;
    memset(&r, 0, sizeof r);
    r.refid = refid;
    r.start = start;
    r.end = end;
    r.left = start - 20;
    r.right = end + 20;
    r.da1[0] = 'G', r.da1[1] = 'T', r.da2[0] = 'A', r.da2[1] = 'G';
    r.nb_raw = raw;
    r.nb_dist = 1;
    return r;
}

int main() {
    auto refs = std::make_shared<bam::RefSeqPtrList>();
    for (int k = 0; k < 4; k++) refs->push_back(std::make_shared<bam::RefSeq>(k, "chr" + std::to_string(k), 1000000));
    std::mt19937_64 rng(11);
    int bad = 0;
    for (int round = 0; round < 20; round++) {
        // four per-target systems; in odd rounds target 2's system repeats a few introns of target 1's (a caller that absorbs overlapping systems)
        std::vector<std::vector<pjb_junction_row>> rows(4);
        for (int t = 0; t < 4; t++) {
            int32_t at = 1000;
            const int n = 50 + (int)(rng() % 400);
            for (int k = 0; k < n; k++) {
                at += 1 + (int32_t)(rng() % 300);
                rows[(size_t)t].push_back(row(t, at, at + 50 + (int32_t)(rng() % 2000), 1 + (uint32_t)(rng() % 100)));
                if (rng() % 4 == 0) rows[(size_t)t].push_back(row(t, at, at + 3000 + (int32_t)(rng() % 100), 1 + (uint32_t)(rng() % 100)));  // shares the donor
            }
        }
        if (round & 1)
            for (int k = 0; k < 5; k++) {
                pjb_junction_row r = rows[1][(size_t)k * 3];
                r.nb_raw += 1000;  // (told apart from the original by its count)
                rows[2].insert(rows[2].begin(), r);
            }
        JunctionSystem a(refs), b(refs);
        std::vector<std::unique_ptr<JunctionSystem>> pa, pb;
        for (int t = 0; t < 4; t++) {
            pa.emplace_back(new JunctionSystem(refs));
            pb.emplace_back(new JunctionSystem(refs));
            pa.back()->appendRows(rows[(size_t)t].data(), rows[(size_t)t].size());
            pb.back()->appendRows(rows[(size_t)t].data(), rows[(size_t)t].size());
        }
        for (int t = 0; t < 4; t++) {
            a.append(*pa[(size_t)t]);
            b.absorb(*pb[(size_t)t]);
        }
        if (a.size() != b.size() || a.getJunctions().size() != b.getJunctions().size()) {
            std::cerr << "round " << round << ": size " << a.size() << " / " << b.size() << ", list " << a.getJunctions().size() << " / " << b.getJunctions().size() << "\n";
            bad++;
            continue;
        }
        for (size_t i = 0; i < a.getJunctions().size(); i++) {
            const Intron& in = *a.getJunctions()[i]->getIntron();
            JunctionPtr ja = a.getJunction(in), jb = b.getJunction(in);
            if (!ja || !jb || ja->getNbSplicedAlignments() != jb->getNbSplicedAlignments()) bad++;
            if (pb[(size_t)in.ref.index]->getJunctions().empty()) bad++;  // (the absorbed systems keep their lists)
        }
        a.sort(), b.sort();
        a.index(), b.index();
        a.calcJunctionStats(), b.calcJunctionStats();
        std::ostringstream sa, sb;
        for (auto& j : a.getJunctions()) sa << *j << "\n";
        for (auto& j : b.getJunctions()) sb << *j << "\n";
        if (sa.str() != sb.str()) {
            std::cerr << "round " << round << ": the tables differ\n";
            bad++;
        }
    }
    std::cout << "mismatches: " << bad << std::endl;
    return bad != 0;
}
