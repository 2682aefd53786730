// The reference's library-level way of driving the junc path (lib/include/portcullis/junction_system.hpp:128-132,
// src/junction_builder.cc:314-357 written out by a caller): a reader loop handing alignments one at a time to
// JunctionSystem::addJunctions, then the per-junction metrics -- here JunctionSystem::finish() -- the merge steps and
// the writers.
//   library_entry <prep_dir> <out_prefix> <orientation>
#include <portcullis/bam/bam_reader.hpp>
#include <portcullis/bam/genome_mapper.hpp>
#include <portcullis/junction_system.hpp>
#include <portcullis/prepared_files.hpp>

#include <climits>
#include <cstdio>
#include <iostream>

using namespace portcullis;

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    try {
        PreparedFiles prep(argv[1]);
        bam::BamReader reader(prep.getSortedBamFilePath());
        reader.open();
        auto refs = reader.createRefList();
        bam::GenomeMapper gmap(prep.getGenomeFilePath());
        gmap.loadFastaIndex();
        JunctionSystem js(refs);
        JunctionSystem::version = "1.2.4";
        uint64_t spliced = 0, unspliced = 0, sumLen = 0;
        int32_t minLen = INT32_MAX, maxLen = 0;
        for (size_t t = 0; t < refs->size(); t++) {
            if (!reader.hasAlignments((int32_t)t)) continue;
            reader.setRegion((int32_t)t);
            while (reader.next()) {
                const bam::BamAlignment& al = reader.current();
                const int32_t len = al.getLength();
                minLen = std::min(minLen, len);
                maxLen = std::max(maxLen, len);
                sumLen += (uint64_t)len;
                if (js.addJunctions(al)) spliced++;
                else unspliced++;
            }
        }
        js.finish(gmap, bam::orientationFromString(argv[3]));
        js.sort();
        js.index();
        js.setQueryLengthStats(minLen, (double)sumLen / (double)(spliced + unspliced), maxLen);
        if (js.size() > 1) js.calcJunctionStats();
        js.saveAll(argv[2], "portcullis", false, false, false);
        printf("junctions=%zu spliced=%llu unspliced=%llu\n", js.size(), (unsigned long long)spliced, (unsigned long long)unspliced);
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
