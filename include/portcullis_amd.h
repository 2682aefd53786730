/*
 * portcullis_amd.h -- C ABI of the MI355X-native `junc` hot path.
 *
 * This is the drop-in boundary: everything HIP lives behind these entry
 * points; the host side (C++ JunctionBuilder / JunctionSystem mirror in
 * portcullis_amd/host/, or any other binding) only sees plain pointers and
 * sizes.  No exceptions cross the boundary: every call returns PJB_OK (0) or
 * a negative PJB_ERR_* code and pjb_last_error() gives the message.
 *
 * What each entry point replaces in the reference (paths relative to the
 * reference checkout):
 *
 *   pjb_set_refs          BamReader::createRefList           src/junction_builder.cc:103-108
 *   pjb_upload_contig     GenomeMapper::fetchBases (6 faidx fetches per junction)
 *                                                            lib/src/junction.cc:566-593,
 *                                                            lib/src/genome_mapper.cc:111-118
 *   pjb_submit_batch      the per-record body of findJuncs: BamAlignment::init,
 *                         length stats, JunctionSystem::addJunctions
 *                                                            src/junction_builder.cc:322-343,
 *                                                            lib/src/bam_alignment.cc:71-100,
 *                                                            lib/src/junction_system.cc:140-210
 *   pjb_finish_contig     junction finalisation for one target sequence:
 *                         Junction::calcMetrics + processJunctionWindow and the
 *                         RegionResult counters                src/junction_builder.cc:324-356,
 *                                                            lib/src/junction.cc:561-649,683-909
 *   pjb_collect           JunctionSystem::append of the per-contig systems
 *                                                            src/junction_builder.cc:258-269
 *   pjb_collect_device    the same rows, still in HBM (multi-GPU merge over xGMI)
 *   pjb_submit_bam        the reader loop itself: BamReader::setRegion / next and htslib's
 *                         bgzf_read_block / inflate_block / bam_read1 for one target
 *                                                            lib/src/bam_reader.cc:78-146,
 *                                                            deps/htslib-1.3/bgzf.c:292-316,421-540
 *   pjb_inflate_bgzf      inflate_block for a run of BGZF blocks
 *   pjb_deflate_bgzf      deflate_block for a stream of bytes (the BAM files the stages write)
 *                                                            deps/htslib-1.3/bgzf.c:292-316
 *
 * A context is bound to one HIP device and is not thread-safe; use one
 * context per GPU and call it from one thread.  Several contigs may be open at
 * once: the batches of one contig must arrive in BAM file order, batches of
 * different contigs may interleave, and pjb_finish_contig closes one contig.
 */
#ifndef PORTCULLIS_AMD_H
#define PORTCULLIS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PJB_ABI_VERSION 4 /* 4: pjb_batch.seq2 / .seq_exc (2-bit bases with their exception bitmap; a context created with abi_version 3 is
                           *    served as before: the two members are not read), pjb_merge_rows
                           * 2: pjb_batch.name_hash, PJB_FLAG_EXTRA, pjb_extra_finish
                           * 3: pjb_timing grew (generic_reads, position_runs, candidates, checked_reads); PJB_MAX_QUEUED 8; pjb_last_error is per calling
                           *    thread; additive since 2: PJB_FLAG_NO_CHAINS, pjb_finish_group_begin/_end, pjb_finish_ready, pjb_deflate_bgzf,
                           *    pjb_host_register/_unregister; the option "fused_k1" is gone */

/* ---- status codes ------------------------------------------------------ */
#define PJB_OK 0
/* data conditions under which the reference throws / crashes (fatal there, reported here) */
#define PJB_ERR_BAD_XS (-1)          /* strandFromChar, bam_master.hpp:60-72 */
#define PJB_ERR_NO_PRESENCE (-2)     /* bam_alignment.cc:342,406 */
#define PJB_ERR_ZERO_LEN_OP (-3)     /* bam_alignment.cc:363 */
#define PJB_ERR_QUERY_RANGE (-4)     /* bam_alignment.cc:376 */
#define PJB_ERR_GENOME_RANGE (-5)    /* bam_alignment.cc:437 */
#define PJB_ERR_QREGION (-6)         /* bam_alignment.cc:414-421 */
#define PJB_ERR_ANCHOR_MISMATCH (-7) /* junction.cc:192-223 */
#define PJB_ERR_SPLICE_SITE_LEN (-8) /* junction.cc:570-585 */
#define PJB_ERR_ANCHOR_LEN (-9)      /* junction.cc:610-623 */
#define PJB_ERR_INTRON_FLANK_LEN (-10) /* junction.cc:624-633 */
#define PJB_ERR_MIN_ANCHOR (-11)     /* intron.cc:67-83 */
#define PJB_ERR_HAMMING_LEN (-12)    /* seq_utils.hpp:62-67 */
#define PJB_ERR_CLIP_RANGE (-13)     /* substr out_of_range, bam_alignment.cc:263 */
#define PJB_ERR_UNSORTED (-14)       /* input contract: coordinate-sorted BAM */
/* API / runtime conditions */
#define PJB_ERR_NOMEM (-15)
#define PJB_ERR_ARG (-16)
#define PJB_ERR_HIP (-17)            /* a HIP runtime call failed */
#define PJB_ERR_NO_DEVICE (-18)      /* no usable gfx950 device: there is NO CPU fallback */
#define PJB_ERR_STATE (-19)          /* calls out of order */
#define PJB_ERR_DIVERGENT (-20)      /* malformed CIGAR: padded query/genome walks emit different
                                        lengths for one op (the reference would compare misaligned
                                        strings or crash; we refuse) */
#define PJB_ERR_BGZF (-22)           /* corrupt BGZF / DEFLATE data (bgzf.c:292-316 returns BGZF_ERR_ZLIB) */
#define PJB_ERR_NO_SEQ (-21)         /* a spliced read was submitted without its sequence bytes */

/* enums follow the reference's order (bam_master.hpp:50-54,92-97,133-139; junction.hpp:86-91) */
enum { PJB_STRAND_POS = 0, PJB_STRAND_NEG = 1, PJB_STRAND_UNK = 2 };
enum { PJB_CSS_CANONICAL = 0, PJB_CSS_SEMI = 1, PJB_CSS_NO = 2 };
enum { PJB_OR_SE = 0, PJB_OR_FR = 1, PJB_OR_RF = 2, PJB_OR_FF = 3, PJB_OR_UNKNOWN = 4 };
enum { PJB_SS_UNSTRANDED = 0, PJB_SS_FIRSTSTRAND = 1, PJB_SS_SECONDSTRAND = 2, PJB_SS_UNKNOWN = 3 };

typedef struct pjb_ctx pjb_ctx;

typedef struct pjb_config {
    int32_t abi_version;  /* PJB_ABI_VERSION */
    int32_t device;       /* HIP device ordinal */
    int32_t orientation;  /* PJB_OR_*  (JunctionBuilder::setOrientation, src/junction_builder.hpp:190) */
    int32_t strandedness; /* PJB_SS_*  accepted for API parity; like the reference it does not
                             change junc output (the reader's alignments carry
                             Strandedness::UNKNOWN, lib/src/bam_alignment.cc:154-165) */
    uint32_t flags;       /* PJB_FLAG_* */
} pjb_config;

/* record a HIP event pair around every kernel launch and accumulate per-kernel time
 * (pjb_get_kernel_timing); costs ~1 us per launch, off by default */
#define PJB_FLAG_KERNEL_TIMING 1u
/* `junc --extra` (JunctionBuilder::setExtra, src/junction_builder.hpp:166; calcExtraMetrics,
 * src/junction_builder.cc:293-312): every pjb_finish_contig also builds the contig's unspliced per-base depth,
 * the flanking alignment counts of its junctions and keeps the name codes of its spliced records; pjb_extra_finish
 * then yields mm_score / coverage / up_aln / down_aln for every row.  Batches must carry name_hash. */
#define PJB_FLAG_EXTRA 2u
/* a context that will not queue kernel chains (bamfilt: pjb_filter_batch, pjb_deflate_bgzf only): pjb_create leaves the chain
 * slots' streams and events (60 ms) to their first use */
#define PJB_FLAG_NO_CHAINS 4u

/* One batch of fixed-width alignment records of ONE contig, in BAM file order
 * (structure of arrays; BAM-native encodings):
 *   cigar    uint32 len<<4|op, op indexes "MIDNSHP=XB"
 *   cig_off  n_reads+1 offsets into cigar
 *   seq4     4-bit packed bases, high nibble first; each read's bytes start on a
 *            4-byte boundary
 *   seq_off  n_reads+1 offsets into seq4 in 4-byte WORDS; only reads with an N
 *            operation need sequence bytes, others may be empty
 *   xs       0 = no XS:A tag / '?' / '.', 1 = '+', 2 = '-', 3 = any other value
 */
typedef struct pjb_batch {
    int64_t n_reads;
    const int32_t *pos;
    const uint16_t *flag;
    const uint8_t *mapq;
    const uint8_t *xs;
    const int32_t *l_qseq;
    const int32_t *mtid;
    const int32_t *mpos;
    const uint32_t *cig_off;
    const uint32_t *cigar;
    const uint32_t *seq_off;
    const uint8_t *seq4;
    /* std::hash<std::string>()(BamAlignment::deriveName()) per record (lib/include/portcullis/junction.hpp:158,
     * lib/src/bam_alignment.cc:233-242: QNAME, plus "_R1" / "_R2" / "_R?" for paired reads); only read by
     * contexts created with PJB_FLAG_EXTRA, may be NULL otherwise. */
    const uint64_t *name_hash;
    /* ---- ABI 4 (read only when pjb_config.abi_version >= 4; both NULL: the compares run on seq4 as before) ----
     * seq2     the same bases in 2 bits (A 0, C 1, G 2, T 3), one uint16 per seq4 WORD: element seq_off[r] + k holds bases 8k .. 8k+7 of
     *          read r, base j at bits 2j, 2j+1 (what is stored for a base outside ACGT, or behind the read's last base, does not matter).
     *          As many elements as seq4 has words, rounded up to an even number (the device reads the array as 32-bit words); the array
     *          starts on a 4-byte boundary.
     * seq_exc  bit r (bit r & 31 of word r >> 5) set: read r is NOT to be compared in 2 bits -- one of its l_qseq bases is not A, C, G or
     *          T (BAM codes 1, 2, 4, 8), or it carries fewer than l_qseq bases.  (n_reads + 31) / 32 words.
     * Who fills them: the device ingest (pjb_submit_bam / pjb_bam_*) as it transcodes the records; a caller that decodes BAM itself
     * packs them where it copies SEQ (AlignmentInfo::calcMatchStats compares characters, lib/src/junction.cc:147-240, and a read of
     * pure ACGT against a stretch of pure ACGT compares the same in 2 bits -- k1_emit takes the 4-bit codes for everything else). */
    const uint16_t *seq2;
    const uint32_t *seq_exc;
} pjb_batch;

/* RegionResult (src/junction_builder.hpp:62-76) plus sizes of what was built */
typedef struct pjb_region_result {
    uint64_t spliced;
    uint64_t unspliced;
    uint64_t sum_len;
    int32_t min_len; /* INT32_MAX when the contig has no reads (src/junction_builder.cc:319) */
    int32_t max_len;
    int64_t n_reads;
    int64_t n_pairs;     /* (alignment, junction) pairs = N operations walked */
    int64_t n_junctions; /* distinct introns on this contig */
} pjb_region_result;

/* One junction as it stands after Junction::calcMetrics + processJunctionWindow
 * (i.e. before the cross-junction pass JunctionSystem::calcJunctionStats, which
 * is O(J) host work above this ABI).  Field names follow the .tab columns. */
typedef struct pjb_junction_row {
    int32_t refid;
    int32_t start, end;   /* intron, 0-based inclusive */
    int32_t left, right;  /* leftAncStart, rightAncEnd */
    uint8_t read_strand, ss_strand, cons_strand; /* PJB_STRAND_* */
    uint8_t canonical;                           /* PJB_CSS_* */
    uint8_t da1[2], da2[2];                      /* ss1 / ss2; may hold NUL (REVCOMP_LOOKUP quirk) */
    uint8_t suspicious;
    uint8_t _pad[3];
    uint32_t nb_raw, nb_dist, nb_ms, nb_um, nb_bpp, nb_ppp, nb_rel;
    uint32_t r1pos, r1neg, r2pos, r2neg;
    uint32_t max_min_anc, maxmmes, hamming5p, hamming3p;
    uint32_t nb_up_juncs, nb_down_juncs;
    uint32_t jad[20];
    uint32_t _pad2;
    uint64_t sum_mismatches; /* mean_mismatches = sum_mismatches / nb_raw (junction.cc:893) */
    double entropy;
} pjb_junction_row;

/* Device-side timing of the last pjb_finish_contig (HIP events on the context's stream) */
#define PJB_N_STAGES 8
typedef struct pjb_timing {
    float total_ms;               /* first kernel -> rows resident on host */
    float stage_ms[PJB_N_STAGES]; /* scan/emit, sort, group, anchors, pair stats, finalise, d2h, (spare): filled only under
                                     PJB_FLAG_KERNEL_TIMING with no kernel selection (an event between kernels costs a bubble) */
    int64_t sort_passes;
    int64_t generic_pairs; /* pairs that took the generic CIGAR walks (k4b_generic) instead of k1_emit's [S]MNM[S] fast path */
    int64_t generic_reads; /* the reads those pairs belong to */
    int64_t position_runs; /* runs of equal read position inside the junctions (the units of the entropy kernels) */
    int64_t candidates;    /* candidate keys the dense junction ids were built from (0: the chain sorted the full keys) */
    int64_t checked_reads; /* reads of the shape [S] M (N M)+ [S] with two or more introns: finished in closed form by k1_emit, their junctions'
                              anchor windows checked by k4b_generic (not counted in generic_reads unless the check failed... they never are: a
                              failed check walks the read without moving it to the other list) */
    /* ---- ABI 4 (not written for a context created with abi_version 3) ---- */
    int64_t repeats;        /* times the chain collected last was queued AGAIN because a limit it had been queued with turned out too small */
    int64_t repeat_reasons; /* which limits, OR-ed over those repeats: 1 pairs, 2 key format, 4 junctions (or the sort's digits), 8 dense ids, 16 read lists */
} pjb_timing;

/* ---- entry points ------------------------------------------------------ */

int pjb_create(pjb_ctx **out, const pjb_config *cfg);
void pjb_destroy(pjb_ctx *ctx);

/* Message of the last failing call the CALLING THREAD made on THIS context (ctx == NULL: its last failing pjb_create); the empty
 * string if that thread's last failure was on another context or it has had none.  Kept per thread: the piece calls below may run
 * beside the context's other calls, and a thread reads its own failure.  A successful call does not clear it. */
const char *pjb_last_error(const pjb_ctx *ctx);

/* Reference sequence lengths, indexed by BAM tid. */
int pjb_set_refs(pjb_ctx *ctx, int32_t n_refs, const int32_t *ref_len);

/* Bases of one contig exactly as faidx returns them (isgraph characters, any case);
 * copied to HBM and upper-cased there (junction.cc:586-587,635-638). */
int pjb_upload_contig(pjb_ctx *ctx, int32_t tid, const uint8_t *bases, int64_t len);
/* Same, for bases that are ALREADY upper-cased and resident in HBM; borrowed until
 * pjb_release_contig / pjb_destroy. */
int pjb_upload_contig_device(pjb_ctx *ctx, int32_t tid, const uint8_t *d_bases_upper, int64_t len);

/* The same upload from the FASTA file's own bytes: `raw` holds the record's sequence lines as they are in the file
 * (from the byte the .fai's OFFSET names, raw_bytes of them), line_blen / line_len are the .fai's LINEBASES / LINEWIDTH
 * and len its LENGTH.  The device takes the line terminators out (base i is byte (i / line_blen) * line_len + i % line_blen),
 * so the host neither parses nor copies the 250 MB of a human chromosome -- GenomeMapper::fetchBases's loader
 * (faidx_fetch_seq, deps/htslib-1.3/faidx.c:439-476, keeps the graphic characters) reduced to a pread.  A record that is
 * not laid out that way (a base position that holds a non-graphic byte, a graphic byte among the terminators, too few
 * bytes) is left alone: *well_formed = 0, nothing is uploaded, and the caller filters the characters itself
 * (pjb_upload_contig).  raw in page-locked memory (pjb_host_alloc) crosses in one DMA. */
int pjb_upload_contig_fasta(pjb_ctx *ctx, int32_t tid, const uint8_t *raw, int64_t raw_bytes, int32_t line_blen, int32_t line_len,
                            int64_t len, int *well_formed);
int pjb_release_contig(pjb_ctx *ctx, int32_t tid);

/* Append a batch to contig `tid` (opens it if it is not open).  The host
 * arrays are packed into one of the context's two page-locked staging buffers
 * and moved to HBM by DMA on the context's stream; the call returns as soon as
 * the arrays have been packed, so they may be reused at once while the DMA of
 * this batch overlaps the decoding of the next. */
int pjb_submit_batch(pjb_ctx *ctx, int32_t tid, const pjb_batch *host_batch);
/* Same for arrays already resident in HBM: borrowed until pjb_finish_contig (or pjb_finish_contig_end) returns. */
int pjb_submit_batch_device(pjb_ctx *ctx, int32_t tid, const pjb_batch *device_batch);

/* Run the device pipeline over everything submitted for contig `tid` and close it.  The contig's
 * genome must have been uploaded.  On return the contig's counters are in *result and its rows are appended to the
 * table pjb_collect returns (they travel to the host by DMA; pjb_collect waits for them). */
int pjb_finish_contig(pjb_ctx *ctx, int32_t tid, pjb_region_result *result);

/* The same in two halves, so that the device never waits for the host between contigs (the reference gets the
 * same effect from its thread pool: one findJuncs per target in flight per thread, src/junction_builder.cc:215-247).
 * _begin queues the whole kernel chain of the contig and returns without waiting; _end waits for the contig's rows
 * and control block, repeats the contig if a limit it was queued with turned out too small, and closes it --
 * pjb_finish_contig is _begin followed by _end.  Up to PJB_MAX_QUEUED contigs may be queued -- their kernel chains
 * run side by side on the device --; they are collected in the order they were queued and their rows land in that order.  Between the two calls the contig's batches (and device
 * arrays lent by pjb_submit_batch_device) must stay as they are; batches for OTHER targets may be submitted, genomes
 * uploaded.  pjb_collect covers collected contigs only; pjb_clear_rows / pjb_set_row_mirror need an empty queue.
 * PJB_FLAG_EXTRA contexts queue like any other (the target's extra metrics are queued when its chain is collected). */
#define PJB_MAX_QUEUED 8
int pjb_finish_contig_begin(pjb_ctx *ctx, int32_t tid);
int pjb_finish_contig_end(pjb_ctx *ctx, int32_t tid, pjb_region_result *result);

/* 1 if pjb_finish_contig_end / pjb_finish_group_end for the OLDEST queued chain would not wait for the device (its kernels
 * have completed, or nothing is queued, or the chain has to be queued again first), else 0.  Never blocks: a caller that
 * serves several targets collects finished chains when they are ready and does something else meanwhile. */
int pjb_finish_ready(pjb_ctx *ctx);

/* Several targets finished as ONE kernel chain (a "group").  A chain of ~45 kernels over one human chromosome's 8 M
 * alignments leaves most of the chip idle in most of its kernels; the reference's answer to many small targets is its
 * thread pool (one findJuncs per target and thread, src/junction_builder.cc:236-247), the device's is to walk several
 * targets' records in one pass: the group's targets are laid side by side in a virtual sequence, the intron keys carry
 * the position in it -- which is what the reference's (refId, start, end) key (lib/include/portcullis/intron.hpp:44-149)
 * amounts to -- and rows come back per target, in the order of `tids`, exactly as n calls of pjb_finish_contig would have
 * produced them (same rows, same per-target results).  Every target named must have had its batches submitted and its
 * genome uploaded; targets without alignments may be named.  _begin / _end pair up like pjb_finish_contig_begin / _end
 * and share their queue (PJB_MAX_QUEUED chains, collected in order; _end names the same targets in the same order).
 * PJB_ERR_ARG from _begin means "not as a group" (more than PJB_GROUP_MAX targets, 2^31 bases or more in all, a
 * PJB_FLAG_EXTRA context, a target whose genome holds characters outside the 16-letter nucleotide alphabet): finish the
 * targets one by one.  A member that turns out to hold alignments outside its own sequence makes _end finish the
 * members one by one itself. */
#define PJB_GROUP_MAX 32
int pjb_finish_group_begin(pjb_ctx *ctx, const int32_t *tids, int32_t n_tids);
int pjb_finish_group_end(pjb_ctx *ctx, const int32_t *tids, int32_t n_tids, pjb_region_result *results /* n_tids of them */);

/* The chain plan: which of `tids` (in the order given) go together as groups.  group_of[k] = index of the group tids[k] belongs to
 * (ascending, groups are runs of consecutive entries); returns the number of groups, a negative status on bad arguments.  No context
 * needed (host arithmetic).  The rule: a group holds consecutive targets whose sequences -- each with the gap the group's virtual
 * sequence leaves behind it, rounded up to 64 -- add up to at most max_bases (<= 0: 2^30, the measured optimum of profiles/
 * r04d_grouping_sweep.txt) and at most PJB_GROUP_MAX members; a set that would be ONE group of more than 0.6 Gb is planned again with 0.55 of its
 * bases as the limit (two chains, three when the targets do not divide that way: the first chain's tail then runs beside the second one's first kernels -- tools/rank_share.py, profiles/r06_rank_share.json).
 * The program (`portcullis_amd junc`), bench.py and the multi-GPU ranks all plan with this one function: the reference hands every target
 * to its thread pool (src/junction_builder.cc:236-247), the device gets chains that fill it. */
int pjb_plan_groups(const int32_t *ref_len, const int32_t *tids, int32_t n_tids, int64_t max_bases, int32_t *group_of);

/* Tuning switches (queue must be empty).  Results never depend on them.
 *   "overlap"    1 (default): a contig's kernels are spread over several HIP streams -- its first kernels beside the
 *                previous contig's last ones, match statistics and entropy beside the sort and the anchors; 0: one
 *                kernel at a time on one stream (clean per-kernel timings)
 *   "dense_ids"  1 (default): the sort works on ordered dense junction ids; 0: on the full intron keys
 *   "extra_dense" 0 (default): PJB_FLAG_EXTRA answers depth and flanking counts from the unspliced records themselves
 *                (a few records per junction) and builds a target's per-base depth vector only where htslib's
 *                8000-record pileup cap may bite; 1: the depth vector for every target (round 2's path)
 *   "sort_floor" 65536 (default): the sort's digits cover at least this many junction ids (and twice what the context's chains have
 *                had); n (test hook): a small floor, so that a chain with more junctions than planned for is repeated
 *   "list_cap"   0 (default): the kernels' read lists get the room the pair limit implies; n > 0 (test hook): the first attempt
 *                of every chain gets room for n entries per sub-list, so that the overflow-and-repeat path runs */
int pjb_set_option(pjb_ctx *ctx, const char *name, int64_t value);

/* All rows built so far, contig by contig in finish order, (start,end)-sorted
 * within a contig.  The pointer stays valid until the next finish/clear/destroy. */
int pjb_collect(pjb_ctx *ctx, const pjb_junction_row **rows, int64_t *n_rows);
/* The rows of the contig finished last, still in HBM (device pointer; valid until the next
 * pjb_finish_contig / pjb_destroy): what a multi-GPU merge all-gathers over xGMI without a detour
 * through host memory. */
int pjb_collect_device(pjb_ctx *ctx, const pjb_junction_row **device_rows, int64_t *n_rows);
/* A device buffer of the caller's (cap_bytes, 0 / NULL to stop) that every following pjb_finish_contig fills
 * before it returns: a 64-byte header of int64 { n_rows, spliced, unspliced, sum_len, min_len, max_len, 0, 0 }
 * followed by rows.  The buffer ACCUMULATES: the rows of every contig finished since the last
 * pjb_set_row_mirror / pjb_clear_rows are appended in finish order and the header holds the row total and the
 * folded counters (sums, min, max), so a rank that owns several contigs has ONE send slot per merge.  It is the
 * send slot of the multi-GPU merge (JunctionSystem::append + the counter sums of src/junction_builder.cc:258-269):
 * when pjb_finish_contig returns the slot can go straight into an all-gather, without a copy or a
 * synchronisation on the caller's side.  Rows that do not fit fail with PJB_ERR_ARG. */
#define PJB_MIRROR_HEADER_BYTES 64
int pjb_set_row_mirror(pjb_ctx *ctx, void *device_buffer, int64_t cap_bytes);
/* The RECEIVE side of that merge, for a caller that runs one process per GPU: `gathered` (HOST memory) holds n_ranks send slots as
 * pjb_set_row_mirror leaves them -- slot r at gathered + r * slot_stride_bytes: the 64-byte header, then its rows -- i.e. what an
 * all-gather of the ranks' slots (ncclAllGather over xGMI, MPI_Allgather, ...) followed by one copy to the host delivers.  Writes ONE
 * table to rows_out: every rank's rows, stably ordered by refid -- a target's rows are contiguous in its rank's slot and already
 * (start, end)-sorted, so this is JunctionSystem::sort's order (lib/src/junction_system.cc:322-330) over the appended systems
 * (src/junction_builder.cc:258-269) -- and folds the ranks' read-length counters into *totals (sums, min, max: src/junction_builder.cc:
 * 270-278; n_reads = spliced + unspliced, n_junctions = rows written, n_pairs = 0: the header does not carry it).  Host arithmetic
 * only: no context, no device, no collective library inside -- INTEGRATION.md section "multi-GPU" shows the lines around it.
 * PJB_ERR_ARG: a header with a negative count or more rows than the slot holds, or cap_rows too small (*n_rows then says how many). */
int pjb_merge_rows(const void *gathered, int32_t n_ranks, int64_t slot_stride_bytes, pjb_junction_row *rows_out, int64_t cap_rows,
                   int64_t *n_rows, pjb_region_result *totals);
int pjb_clear_rows(pjb_ctx *ctx);

/* The --extra columns of one junction (lib/include/portcullis/junction.hpp:240-243). */
typedef struct pjb_extra_row {
    double mm_score;  /* Junction::calcMultipleMappingScore, lib/src/junction.cc:914-921 */
    double coverage;  /* Junction::calcCoverage, lib/src/junction.cc:935-951, on DepthParser's vector
                         (lib/src/depth_parser.cc:112-164), handed out per JunctionSystem::calcCoverage
                         (lib/src/junction_system.cc:231-242) */
    uint32_t up_aln;  /* nbUpstreamFlankingAlignments, lib/src/junction.cc:651-677 */
    uint32_t down_aln;
} pjb_extra_row;
/* JunctionBuilder::calcExtraMetrics (src/junction_builder.cc:293-312) once every contig of the FILE has been
 * finished on this context (the name multiplicities and the depth hand-over between consecutive targets are
 * file-wide): one pjb_extra_row per row of pjb_collect, same order.  Needs PJB_FLAG_EXTRA.  The pointer stays
 * valid until the next pjb_clear_rows / pjb_destroy. */
int pjb_extra_finish(pjb_ctx *ctx, const pjb_extra_row **rows, int64_t *n_rows);

int pjb_get_timing(const pjb_ctx *ctx, pjb_timing *out);

/* Per-kernel device time (HIP events on the context's stream), accumulated over all
 * pjb_finish_contig calls since creation / the last reset.  Needs PJB_FLAG_KERNEL_TIMING. */
typedef struct pjb_kernel_time {
    char name[32];
    int64_t launches;
    double total_ms;
} pjb_kernel_time;
int pjb_get_kernel_timing(const pjb_ctx *ctx, pjb_kernel_time *out, int32_t cap, int32_t *n);
int pjb_reset_kernel_timing(pjb_ctx *ctx);
/* Restrict the event bracketing to the named kernels ("k4_pairs,rs_scatter"); "" or NULL = all. */
int pjb_select_timed_kernels(pjb_ctx *ctx, const char *comma_separated_names);

/* Page-locked host memory for batch arrays (hipHostMalloc); NULL if it cannot be had. */
void *pjb_host_alloc(size_t bytes);
void pjb_host_free(void *p);
/* Page-lock memory the caller already has -- e.g. a read-only mapping of the BAM file: the file's bytes then cross from the
 * page cache by DMA, without a copy by the CPU (hipHostRegister; `p` and `bytes` multiples of the page size).  The range
 * must not overlap another registered range; unregister before unmapping. */
int pjb_host_register(void *p, size_t bytes);
int pjb_host_unregister(void *p);

/* Number of visible HIP devices (0 if none); does not create a context. */
int pjb_device_count(void);

/* ---- device-side ingest -------------------------------------------------------
 * Inflate a run of whole BGZF blocks on the device: replaces the bgzf_read_block / inflate_block loop of
 * htslib (deps/htslib-1.3/bgzf.c:292-316, 421-540) that BamReader::next drives one block at a time
 * (lib/src/bam_reader.cc:134-142).  `comp` holds `comp_bytes` bytes of consecutive BGZF blocks (host
 * memory; page-locked memory from pjb_host_alloc makes the copy asynchronous); the inflated bytes of all
 * blocks are written back to `out` (capacity `out_cap`) and their count to *out_bytes.  Like
 * inflate_block, the footer CRC32 is not verified; a block whose inflated size differs from its ISIZE
 * field is an error (PJB_ERR_BGZF, message names the block).  An empty block (the BGZF EOF marker)
 * contributes nothing. */
int pjb_inflate_bgzf(pjb_ctx* ctx, const uint8_t* comp, int64_t comp_bytes, uint8_t* out, int64_t out_cap, int64_t* out_bytes);

/* The other direction: BGZF-compress `n_bytes` at `in` (host memory) on the device, for the BAM files the stages write
 * (BamWriter::write -> bam_write1 -> bgzf_write -> deflate_block, lib/src/bam_writer.cc:58-60,
 * deps/htslib-1.3/bgzf.c:216-262,580-613).  The input is cut into blocks of `block_bytes` bytes (a multiple of 4, at most
 * 0xff00 = BGZF_BLOCK_SIZE; the last block may be shorter); every block becomes one complete BGZF member (gzip header with
 * the BC field, a dynamic-Huffman or stored deflate block, CRC-32, ISIZE) and the members are written back to back to `out`
 * (capacity out_cap; n_blocks * 65536 always suffices), their total to *out_bytes, and -- if member_size is not NULL -- each
 * member's length to member_size[0 .. n_blocks) (a writer needs them for virtual file offsets).  No EOF block is added.
 * Any inflater reads the result; it is not the byte stream zlib would have produced (a single hash candidate per position). */
int pjb_deflate_bgzf(pjb_ctx* ctx, const uint8_t* in, int64_t n_bytes, int32_t block_bytes, uint8_t* out, int64_t out_cap, int64_t* out_bytes,
                     uint32_t* member_size);

/* One target's alignments straight from the file bytes: inflate + BAM record parse + transcode to the
 * pjb_batch layout, all on the device; the records are appended to target `tid` exactly as a
 * pjb_submit_batch of the same alignments would be.  Replaces the reader loop of
 * JunctionBuilder::findJuncs (src/junction_builder.cc:322-343: BamReader::setRegion / next /
 * BamAlignment::init, lib/src/bam_reader.cc:78-146, lib/src/bam_alignment.cc:71-100) for the whole target.
 *   comp, comp_bytes : consecutive whole BGZF blocks (host memory; page-locked memory from pjb_host_alloc is
 *                      DMA'd directly, anything else goes through the context's staging buffers), from the
 *                      block that holds the target's
 *                      first record through (at least) the block that holds its last one;
 *   first_uoffset    : offset of that first record inside the first block's inflated bytes (the low
 *                      16 bits of the index's virtual offset).
 * Records are taken until the first one whose refID is not `tid` (or whose position is past the
 * target's end, as hts_itr would stop) or the end of the data.  Data that ends inside one of the target's
 * records (before any record of another target) is an error: PJB_ERR_BGZF.  One call per target.  *n_records (optional) receives the number of alignments added.
 * Errors: PJB_ERR_BGZF for corrupt BGZF / DEFLATE / BAM record data. */
int pjb_submit_bam(pjb_ctx* ctx, int32_t tid, const uint8_t* comp, int64_t comp_bytes, int32_t first_uoffset, int64_t* n_records);

/* The same with the target's bytes arriving in PIECES, in file order, any sizes -- for a reader that fills a small ring
 * of page-locked buffers instead of one buffer per target (page-locking memory costs ~0.15 s per GB: three 3 GB buffers
 * were 1.4 s of a 5 s run over a 33 GB file):
 *   pjb_bam_begin        announces `total_bytes` for target `tid` and reserves device memory for them;
 *   pjb_bam_piece        queues the copy of the next `bytes` to the device (asynchronous if `piece` is page-locked)
 *                        and meanwhile hops over the BGZF block headers that this piece completes; *ticket (optional)
 *                        identifies the copy;
 *   pjb_bam_pieces_done  *completed_ticket = the highest ticket whose copy has left its host buffer (copies complete
 *                        in order; never blocks): the buffers of all pieces up to it may be reused;
 *   pjb_bam_end          inflate + parse + transcode of the staged bytes: from here on exactly pjb_submit_bam
 *                        (same result, same errors; a BGZF error found while hopping is reported by the piece call;
 *                        a failing piece call drops the target's staging: begin again).
 * Several targets may be between _begin and _end at once (each has its own device buffers, taken from pools the
 * context keeps).  The call that hands over a target's LAST piece also starts its inflate (own stream, lowest priority,
 * behind the copy): see pjb_bam_inflate_done.
 * Threads: pjb_bam_begin, _piece, _pieces_done and _inflate_done may be called from other threads than the one that makes
 * the context's other calls (a thread that reads the file hands its pieces over itself); pjb_bam_end belongs to that one
 * thread, like everything else. */
int pjb_bam_begin(pjb_ctx* ctx, int32_t tid, int64_t total_bytes);
int pjb_bam_piece(pjb_ctx* ctx, int32_t tid, const uint8_t* piece, int64_t bytes, int64_t* ticket);
int pjb_bam_pieces_done(pjb_ctx* ctx, int64_t* completed_ticket);
int pjb_bam_end(pjb_ctx* ctx, int32_t tid, int32_t first_uoffset, int64_t* n_records);
/* The target's inflate (bgzf_decode + bgzf_resolve) starts by itself when its last piece has been handed over (on a stream of its own, behind the
 * copy): 1 if it has finished -- or none is in flight -- i.e. pjb_bam_end will not wait for it, else 0.  A caller that
 * serves several targets can hand over other targets' pieces meanwhile; their inflates then run side by side (a launch
 * takes ~50 ms whatever its size, and most targets do not fill the chip). */
int pjb_bam_inflate_done(pjb_ctx *ctx, int32_t tid);

/* ---- `portcullis filt` feature rows -------------------------------------------------------
 * ModelFeatures::setRow (lib/src/model_features.cc:161-212) for a list of junctions: the columns of VAR_NAMES +
 * JAD_NAMES (lib/include/portcullis/ml/model_features.hpp:45-60), i.e. the row getters, calcIntronScore
 * (lib/src/junction.cc:953-956), Junction::calcCodingPotential / calcSplicingScores /
 * calcJunctionAnchorDepthLogDeviation (lib/src/junction.cc:1328-1391) with the Markov scores of
 * KmerMarkovModel::getScore / PosMarkovModel::getScore (lib/src/markov_model.cc:57-78,101-115) over genome windows.
 * The models arrive as dense tables over the alphabet makeClean leaves (A C G T N -> 0..4): a k-mer model of order
 * 5 is [5^5 contexts][5 next letters] probabilities (0 = never seen), a position model [PJB_PW_LEN positions][5].
 * A NULL table is an untrained model.  *_size: the reference's model.size() (contexts / positions trained; 0 =
 * empty -> the column is 0 as isCodingPotentialModelEmpty / isPWModelEmpty make it).  The genome of every target
 * the junctions lie on must have been uploaded (pjb_upload_contig).  mean_read_length: what
 * Junction::setMeanReadLength stored (the truncated mean, lib/include/portcullis/junction.hpp:928). */
#define PJB_N_FEATURES 34
#define PJB_KMER_ORDER 5
#define PJB_KMER_TABLE (3125 * 5)
#define PJB_PW_LEN 32
typedef struct pjb_markov_models {
    const double *exon, *intron;                               /* coding potential, ModelFeatures::exonModel / intronModel */
    const double *donor_t, *donor_f, *acceptor_t, *acceptor_f; /* splicing signal: true / false models */
    const double *donor_pw, *acceptor_pw;                      /* position weights, order 1 */
    int32_t exon_size, intron_size, donor_pw_size, acceptor_pw_size;
} pjb_markov_models;
int pjb_filt_features(pjb_ctx *ctx, const pjb_junction_row *rows, int64_t n_rows, double mean_read_length, uint32_t l95,
                      const pjb_markov_models *models, double *features_out /* n_rows x PJB_N_FEATURES, host */);

/* ---- `portcullis bamfilt` ----------------------------------------------------------------
 * The per-alignment decision of BamFilter::filter (src/bam_filter.cc:152-247): walk the CIGAR as
 * BamFilter::containsJunctionInSystem / clipMSR do (src/bam_filter.cc:75-150) and probe the set of junctions that
 * passed the filter.  The set of one target is uploaded as sorted keys; pjb_filter_batch then writes one code per
 * alignment of a batch of that target:
 *   0 dropped, 1 kept (not spliced), 2 kept (spliced, one of its introns is in the set),
 *   3 kept (multiply spliced read in HARD / SOFT clip mode with a good junction: the reference's "Modified" count).
 * Only pos, cig_off and cigar of the batch are read.  Kept records leave the reference unchanged in every clip
 * mode (its clipping edits a cached copy of the CIGAR that BamWriter never writes), so a code is all a writer needs.
 * Works on any context (no genome, no PJB_FLAG_*). */
enum { PJB_CLIP_HARD = 0, PJB_CLIP_SOFT = 1, PJB_CLIP_COMPLETE = 2 }; /* ClipMode, src/bam_filter.hpp:50-54 */
/* keys: (uint64)(uint32)start << 32 | (uint32)end of the target's passing junctions, ascending, host memory */
int pjb_filter_set_junctions(pjb_ctx *ctx, int32_t tid, const uint64_t *sorted_keys, int64_t n_keys);
int pjb_filter_batch(pjb_ctx *ctx, int32_t tid, const pjb_batch *host_batch, int32_t clip_mode, uint8_t *codes_out);

#ifdef __cplusplus
}
#endif
#endif
