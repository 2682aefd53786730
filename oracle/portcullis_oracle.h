/*
 * portcullis_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * A plain-C restatement of the reference's `junc` hot path
 * (portcullis::JunctionBuilder::findJuncs and everything below it), written
 * to mirror the reference's algorithm literally (string building, per-read
 * loops, hash-map grouping) so that it is an independent check on the HIP
 * path, which computes the same quantities by direct counting.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path (portcullis_amd/) never links, loads
 * or calls it.
 *
 * PARITY PIN: the oracle is pinned by (a) every known-answer test the
 * reference holds for this path (tests/bam_tests.cpp:181-248 padding KATs,
 * tests/intron_tests.cpp:29-65, tests/junction_tests.cpp:39-88,
 * tests/seq_utils_tests.cpp:30-48) and (b) outputs of the real reference
 * recorded in SURVEY.md Appendix A/B (two micro-fixtures on spombe.III.fa and
 * the clipped3.bam row), see tests/test_oracle_*.py.  The reference itself is
 * not buildable in this image (needs Boost, which is absent), so no
 * oracle/_ref binary exists.
 *
 * All `file:line` citations are relative to /root/reference.
 */
#ifndef PORTCULLIS_ORACLE_H
#define PORTCULLIS_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes: every condition under which the reference throws (or runs into
 * undefined behaviour) is reported as a negative code; orc_last_error() gives
 * a message. */
#define ORC_OK 0
#define ORC_ERR_BAD_XS (-1)          /* bam_master.hpp:60-72 strandFromChar throws */
#define ORC_ERR_NO_PRESENCE (-2)     /* bam_alignment.cc:342,406 */
#define ORC_ERR_ZERO_LEN_OP (-3)     /* bam_alignment.cc:363 */
#define ORC_ERR_QUERY_RANGE (-4)     /* bam_alignment.cc:376 */
#define ORC_ERR_GENOME_RANGE (-5)    /* bam_alignment.cc:437 */
#define ORC_ERR_QREGION (-6)         /* bam_alignment.cc:414-421 */
#define ORC_ERR_ANCHOR_MISMATCH (-7) /* junction.cc:192-223 (warning, then UB in calcMismatchStats) */
#define ORC_ERR_SPLICE_SITE_LEN (-8) /* junction.cc:570-585 */
#define ORC_ERR_ANCHOR_LEN (-9)      /* junction.cc:610-623 */
#define ORC_ERR_INTRON_FLANK_LEN (-10) /* junction.cc:624-633 */
#define ORC_ERR_MIN_ANCHOR (-11)     /* intron.cc:67-83 */
#define ORC_ERR_HAMMING_LEN (-12)    /* seq_utils.hpp:62-67 */
#define ORC_ERR_CLIP_RANGE (-13)     /* std::out_of_range from substr, bam_alignment.cc:263 */
#define ORC_ERR_UNSORTED (-14)       /* input contract: coordinate-sorted BAM */
#define ORC_ERR_NOMEM (-15)
#define ORC_ERR_ARG (-16)

/* Strand / canonical codes use the reference's enum order
 * (bam_master.hpp:50-54, junction.hpp:86-91). */
enum { ORC_STRAND_POS = 0, ORC_STRAND_NEG = 1, ORC_STRAND_UNK = 2 };
enum { ORC_CSS_CANONICAL = 0, ORC_CSS_SEMI = 1, ORC_CSS_NO = 2 };
/* Orientation (bam_master.hpp:133-139) */
enum { ORC_OR_SE = 0, ORC_OR_FR = 1, ORC_OR_RF = 2, ORC_OR_FF = 3, ORC_OR_UNKNOWN = 4 };

/* Alignment records of ONE contig in BAM file order, structure-of-arrays.
 * cigar is BAM-native (len<<4|op, op index into "MIDNSHP=XB"); seq4 is
 * BAM-native 4-bit packed (high nibble first); xs: 0 = no XS tag / '?' / '.',
 * 1 = '+', 2 = '-', 3 = anything else (reference throws). */
typedef struct orc_reads {
    int64_t n;
    const int32_t *pos;
    const uint16_t *flag;
    const uint8_t *mapq;
    const uint8_t *xs;
    const int32_t *l_qseq;
    const int32_t *mtid;
    const int32_t *mpos;
    const uint32_t *cig_off; /* n+1 */
    const uint32_t *cigar;
    const uint64_t *seq_off; /* n+1, byte offsets into seq4; may be empty for a read */
    const uint8_t *seq4;
} orc_reads;

typedef struct orc_row {
    uint32_t id;
    int32_t refid;
    int32_t start, end;          /* intron, 0-based inclusive */
    int32_t left, right;         /* leftAncStart, rightAncEnd */
    uint8_t read_strand, ss_strand, cons_strand;
    uint8_t canonical;
    uint8_t da1[2], da2[2];      /* may contain NUL bytes (REVCOMP_LOOKUP quirk) */
    uint8_t suspicious, pfp, uniq, primary;
    uint32_t nb_raw, nb_dist, nb_ms, nb_um, nb_bpp, nb_ppp, nb_rel;
    uint32_t r1pos, r1neg, r2pos, r2neg;
    double entropy, mean_mismatches, mean_readlen;
    uint32_t max_min_anc, maxmmes, hamming5p, hamming3p;
    uint32_t nb_up_juncs, nb_down_juncs;
    uint32_t dist_up, dist_down, dist_nearest;
    uint32_t jad[20];
    uint64_t sum_mismatches;     /* numerator of mean_mismatches (integer, for bit-exact checks) */
    /* --extra columns (0 unless orc_extra ran): junction.hpp:240-243 */
    double mm_score, coverage;
    uint32_t up_aln, down_aln;
} orc_row;

/* RegionResult (src/junction_builder.hpp:62-76) */
typedef struct orc_region {
    uint64_t spliced, unspliced, sum_len;
    int32_t min_len, max_len;
} orc_region;

/* ---- unit-level entry points (KAT surface) ---- */

/* BamAlignment::getPaddedQuerySeq(query_seq,start,end,actual_start,actual_end,false)
 * (lib/src/bam_alignment.cc:341-403).  query is the UNclipped read as letters.
 * Returns length written to out (NUL-terminated) or a negative error. */
int orc_padded_query_seq(const uint32_t *cigar, int n_cigar, int32_t position, int32_t aligned_len,
                         const char *query, int32_t start, int32_t end,
                         int32_t *actual_start, int32_t *actual_end, char *out, size_t out_cap);

/* BamAlignment::getPaddedGenomeSeq (lib/src/bam_alignment.cc:405-462). */
int orc_padded_genome_seq(const uint32_t *cigar, int n_cigar, int32_t position, int32_t aligned_len,
                          const char *genome_seq, int32_t start, int32_t end,
                          int32_t q_start, int32_t q_end, char *out, size_t out_cap);

/* SeqUtils (lib/include/portcullis/seq_utils.hpp:62-118) */
int orc_hamming(const char *a, size_t na, const char *b, size_t nb);
void orc_revcomp(const char *in, size_t n, char *out);

/* Intron::minAnchorLength (lib/src/intron.cc:67-83); negative error if it throws. */
int64_t orc_min_anchor(int32_t start, int32_t end, int32_t left, int32_t right);

/* Junction::setDonorAndAcceptorMotif (lib/src/junction.cc:289-326,504-516).
 * Returns canonical code; fills ss/consensus strand and da1/da2. */
int orc_donor_acceptor(const char *seq1, size_t n1, const char *seq2, size_t n2, int read_strand,
                       int *ss_strand, int *cons_strand, uint8_t da1[2], uint8_t da2[2]);

/* Junction::calcEntropy(vector<int32_t>) (lib/src/junction.cc:730-749); positions must be sorted. */
double orc_entropy(const int32_t *sorted_pos, size_t n);

/* Junction::calcHammingScores (lib/src/junction.cc:823-857). */
int orc_hamming_scores(const char *la, size_t nla, const char *li, size_t nli, const char *ri,
                       size_t nri, const char *ra, size_t nra, int cons_strand, uint32_t *h5,
                       uint32_t *h3);

/* ---- path-level entry points ---- */

/* JunctionBuilder::findJuncs for one contig (src/junction_builder.cc:314-357):
 * CIGAR walk, grouping, calcMetrics + processJunctionWindow for every junction.
 * genome = contig bases as faidx would return them (isgraph chars), NOT upper-cased
 * (the oracle upper-cases where the reference does).  Rows come back in
 * first-seen order (junctionList order); caller frees with orc_free_rows. */
int orc_find_juncs(int32_t tid, int32_t ref_len, const char *genome, const orc_reads *reads,
                   int orientation, orc_row **rows_out, int64_t *n_rows_out, orc_region *region_out);

void orc_free_rows(orc_row *rows);

/* Merge step of JunctionBuilder::findJunctions (src/junction_builder.cc:258-290):
 * sort by (refid,start,end), index, and if n>1 calcJunctionStats
 * (lib/src/junction_system.cc:250-320).  mean_query_len = sum/(spliced+unspliced). */
void orc_finalize(orc_row *rows, int64_t n, double mean_query_len);

/* ---- `junc --extra` (hidden flag; src/junction_builder.cc:152-226,293-312) ------------------------------
 * std::hash<std::string>()(BamAlignment::deriveName()) (lib/include/portcullis/junction.hpp:158,
 * lib/src/bam_alignment.cc:233-242).  std::hash<std::string> is not in the reference checkout: it is the
 * C++ standard library's (GNU libstdc++, the toolchain the reference's autotools build uses on Linux;
 * libstdc++-v3/libsupc++/hash_bytes.cc, 64-bit _Hash_bytes = a MurmurHash64A variant, seed 0xc70f6907),
 * restated here from its published algorithm.  Only equality of codes reaches the output (mm_score). */
uint64_t orc_name_hash(const char *qname, size_t n, uint16_t flag);

/* Per-base depth of the unspliced alignments of ONE contig as DepthParser::loadNextBatch builds it
 * (lib/src/depth_parser.cc:112-164) on top of htslib-1.3's pileup (deps/htslib-1.3/sam.c:1853-1975):
 * depth[x + 1] = alignments with an M/=/X base on x, among the records bam_plp_push accepts (mapped, and not
 * dropped by the 8000-read cap: a record starting at the pileup's current position while more than
 * maxcnt records are buffered).  `depth` has ref_len entries (the write for x + 1 == ref_len is out of
 * bounds in the reference and skipped here).  Returns the number of records that took part, or < 0. */
int64_t orc_depth(int32_t ref_len, const orc_reads *reads, uint32_t *depth);

/* Junction::calcCoverage(const vector<uint32_t>&) (lib/src/junction.cc:923-951). */
double orc_calc_coverage(int32_t start, int32_t end, const uint32_t *levels, size_t n_levels);

/* JunctionBuilder::separateBams' name map + calcExtraMetrics (src/junction_builder.cc:168-176,293-312):
 * mm_score (junction.cc:914-921), up_aln / down_aln (junction.cc:651-677) and coverage
 * (junction_system.cc:231-242 incl. the batch / contig pairing of DepthParser::getCurrentRefIndex) for
 * finalised rows (sorted by refid,start,end).  reads[t] / name_hash[t] are the records of target t in file
 * order (n = 0: none); max_query_len as JunctionSystem::setQueryLengthStats got it. */
int orc_extra(int32_t n_refs, const int32_t *ref_len, const orc_reads *reads, const uint64_t *const *name_hash,
              orc_row *rows, int64_t n_rows, int32_t max_query_len);

/* JunctionSystem::determineStrandedness (lib/src/junction_system.cc:455-560): orientation (ORC_OR_*) and
 * strandedness (bam_master.hpp:92-97 order: 0 UNSTRANDED, 1 FIRSTSTRAND, 2 SECONDSTRAND, 3 UNKNOWN) inferred from
 * the R1/R2 x splice-site-strand totals.  The unqualified `abs` of :553 is taken as std::abs(double) (what a
 * libstdc++ with <cmath> in scope resolves it to). */
void orc_determine_strandedness(const orc_row *rows, int64_t n, int *orientation, int *strandedness);

/* ---- `portcullis bamfilt` (src/bam_filter.cc:75-247) --------------------------------------
 * The decision BamFilter::filter makes for every record of one target, given the junctions of the filter's .tab
 * file on that target (js_start / js_end, n_js of them, any order):
 *   0  dropped
 *   1  kept: not spliced                                           (bam_filter.cc:221-224)
 *   2  kept: spliced, containsJunctionInSystem                     (:196-202, :75-100)
 *   3  kept: multiply spliced in HARD / SOFT mode, clipMSR found a good junction ("Modified" count, :204-218)
 * clip_mode: 0 HARD, 1 SOFT, 2 COMPLETE.  As written in the reference, the walk does not advance over an N
 * operation (:86-96: only the else-branch adds to lEnd), so the introns after a read's first one are looked up
 * at coordinates short of the earlier introns' lengths; and clipMSR edits only the cached CIGAR vector while
 * BamWriter::write emits the untouched bam1_t (lib/src/bam_writer.cc:58-60), so kept records leave unchanged in
 * every mode. */
int orc_bamfilt_flags(const orc_reads *reads, const int32_t *js_start, const int32_t *js_end, int64_t n_js, int clip_mode,
                      uint8_t *out);

/* ---- `portcullis filt` feature extraction -------------------------------------------------
 * ModelFeatures::setRow (lib/src/model_features.cc:161-212) with Junction::calcSplicingScores / calcCodingPotential
 * / calcJunctionAnchorDepthLogDeviation / calcIntronScore (lib/src/junction.cc:1328-1391,953-956) over the Markov
 * models of lib/src/markov_model.cc, trained as ModelFeatures does (model_features.cc:67-158):
 *   L95                calcIntronThreshold over the junctions l95_idx (0 of them: L95 = 0)
 *   exon / intron      trainCodingPotentialModel over cp_idx
 *   donor / acceptor   trainSplicingModels over pass_idx (true + position-weight models) and fail_idx (false models)
 * then one row of ORC_N_FEATURES doubles per junction, in the column order of VAR_NAMES + JAD_NAMES
 * (lib/include/portcullis/ml/model_features.hpp:45-60).  genomes[t] / ref_len[t]: the contigs as faidx returns them.
 * rows must be finalised (mean_readlen set).  `models_out` (optional, ORC_MODEL_DOUBLES doubles) receives the
 * trained tables in the dense layout the device path takes (see pjb_markov_models). */
#define ORC_N_FEATURES 34
#define ORC_KMER_TABLE (3125 * 5)                 /* order 5 over A C G T N: [context][next] */
#define ORC_PW_LEN 32                             /* positions of a position-weight model (windows are 24 / 23 long) */
#define ORC_MODEL_DOUBLES (6 * ORC_KMER_TABLE + 2 * ORC_PW_LEN * 5 + 8)
int orc_filt_features(int32_t n_refs, const int32_t *ref_len, const char *const *genomes, const orc_row *rows, int64_t n_rows,
                      const int64_t *l95_idx, int64_t n_l95, const int64_t *cp_idx, int64_t n_cp, const int64_t *pass_idx,
                      int64_t n_pass, const int64_t *fail_idx, int64_t n_fail, double *features_out, double *models_out,
                      uint32_t *l95_out);

/* Writers.  Return a malloc'd buffer (caller frees with orc_free_text) and its length.
 * ref_names[refid], ref_lens[refid].  (.tab: junction.hpp:1260-1319 + junction_system.hpp:154-160
 * + junction_system.cc:356; .bed: junction_system.cc:411-418 + junction.cc:1189-1214;
 * GFF: junction.cc:1102-1183) */
char *orc_write_tab(const orc_row *rows, int64_t n, const char *const *ref_names,
                    const int32_t *ref_lens, size_t *len_out);
char *orc_write_bed(const orc_row *rows, int64_t n, const char *const *ref_names,
                    const char *source, const char *version, size_t *len_out);
char *orc_write_intron_gff(const orc_row *rows, int64_t n, const char *const *ref_names,
                           const char *source, size_t *len_out);
char *orc_write_exon_gff(const orc_row *rows, int64_t n, const char *const *ref_names,
                         const char *source, size_t *len_out);
void orc_free_text(char *p);

const char *orc_last_error(void);
size_t orc_sizeof_row(void); /* binding check */

#ifdef __cplusplus
}
#endif
#endif
