// pjb_host.hip.h -- what the translation units of the C ABI share: the context (pjb_ctx) and its parts, error / allocation / launch helpers.
// The library is built from three units -- pjb_api.hip (contexts, uploads, batches, the kernel chains: pjb_kernels.hip.h), pjb_extra_api.hip
// (--extra, bamfilt, filt features: pjb_extra.hip.h) and pjb_ingest_api.hip (BGZF inflate / deflate, BAM records: pjb_ingest.hip.h, pjb_deflate.hip.h) --
// each of which compiles its own kernel family (the non-template kernels of a header are guarded by PJB_KERNELS_CHAIN / PJB_KERNELS_EXTRA), so an edit to
// one family rebuilds one unit.  Helpers are inline functions of this header: one definition, one set of thread-local error strings for all units.
#pragma once
#include "pjb_kernels.hip.h"
#include "pjb_extra.hip.h"

#include <algorithm>
#include <sys/mman.h>
#include <thread>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace pjb;


inline thread_local std::string g_create_error;
// The message of the last failing call of THIS thread (pjb_last_error returns it): pjb_bam_begin / _piece / _pieces_done /
// _inflate_done may run on other threads than the context's other calls, and a thread must neither read a string another
// thread is reassigning nor report another thread's failure.
inline thread_local std::string g_thread_error;
inline thread_local const void *g_thread_error_ctx = nullptr; // the context the message belongs to (a thread may drive several)

struct Buf {
    void *p = nullptr;
    size_t cap = 0;
};

struct Contig {
    uint8_t *d = nullptr;
    int64_t len = 0;
    bool owned = false;
    bool has_x = false;
    bool present = false;
    u32 *codes = nullptr; // packed 4-bit codes (k0_encode); nullptr when the contig is "exotic"
    u32 *codes2 = nullptr; // 2-bit codes | exception bitmap (k0_encode2): behind the 4-bit codes, in their allocation
    bool any_exc = true;   // the bitmap has a bit set (false: k1_emit never looks at it)
    size_t d_cap = 0, codes_cap = 0; // sizes of the allocations (they go back to the context's genome pool)
};

// The bases and codes of a released contig are kept for the next upload (targets come longest first, so the next genome
// fits): a hipMalloc / hipFree pair of 250 MB is ~10 ms on the thread that serves every target.
inline void free_contig(Contig &g, std::vector<Buf> *pool = nullptr) {
    auto give = [&](void *p, size_t cap) {
        if (!p) return;
        if (pool && cap > 0 && pool->size() < 12) {
            Buf b;
            b.p = p;
            b.cap = cap;
            pool->push_back(b);
        } else
            (void)hipFree(p);
    };
    if (g.owned) give(g.d, g.d_cap);
    give(g.codes, g.codes_cap); // (codes2 lies in the same allocation)
    g = Contig();
}
// device memory for a genome array: the smallest pooled buffer that fits, else a new one
inline void *genome_take(std::vector<Buf> &pool, size_t bytes, size_t &cap) {
    int best = -1;
    for (size_t k = 0; k < pool.size(); k++)
        if (pool[k].cap >= bytes && (best < 0 || pool[k].cap < pool[(size_t)best].cap)) best = (int)k;
    if (best >= 0) {
        void *p = pool[(size_t)best].p;
        cap = pool[(size_t)best].cap;
        pool.erase(pool.begin() + best);
        return p;
    }
    void *p = nullptr;
    cap = bytes;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
#ifdef PJB_DEBUG_ALLOC
    fprintf(stderr, "[alloc] genome: %p .. %p (%zu bytes)\n", p, (void *)((char *)p + bytes), bytes);
#endif
    return p;
}

struct Slab { // device memory for the batches copied in by pjb_submit_batch; reused contig after contig
    uint8_t *p = nullptr;
    size_t cap = 0, used = 0;
};

// a contig that has received batches and is not finished yet; several may be open at once
struct OpenContig {
    std::vector<DevBatch> batches;
    std::vector<int32_t> last_pos;  // pos of the last record of each batch (sortedness across batches)
    std::vector<char> last_known;   // 0 = must be read back from the device (device-resident batch)
    std::vector<Slab> slabs;        // device memory holding this contig's host-submitted batches
    bool on_main_stream = false;    // some batch was produced by work queued on the main stream (host copies, BAM ingest)
    size_t slab_hint = 0;           // what the target's records will take in all, roughly (device ingest: from the inflated bytes): its first slab
                                    // is this large -- ONE hipMalloc per target instead of one per 128 MB (targets finished as groups keep their
                                    // slabs until the group is collected: nothing comes back to the pool in between)
};


// --extra: what is kept of a finished contig until pjb_extra_finish
struct ExtraContig {
    int32_t tid = -1;
    int32_t len = 0;
    u32 *cover = nullptr;      // per-base depth of the unspliced records (len + 2 entries), nullptr: none
    bool has_unspliced = false;
    size_t row_base = 0, n_rows = 0;
    ExtraRow *xr = nullptr;    // n_rows entries (flanking counts now; m_sum / mm_score / coverage in phase 2)
    u64 *pair_code = nullptr;  // per sorted pair: name code of its record
    u32 *pair_row = nullptr;   //                  row (index into the context's row table)
    u32 n_pairs = 0;
    u64 *spl_codes = nullptr;  // name codes of the contig's spliced records
    u32 n_spl = 0;
    bool dense = false;        // went through the dense path: `cover` and the other pointers are allocations of their own
    bool codes_in_table = false; // the spliced records' codes are in the name table already
    SparseDepth sparse = {nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0}; // else: the records' spans (arena memory)
};

// what the targets of a PJB_FLAG_EXTRA context keep until pjb_extra_finish comes from a few large allocations that are
// reused by the next file (pjb_clear_rows): no hipMalloc / hipFree per target
struct XArena {
    struct Chunk {
        uint8_t *p;
        size_t cap;
    };
    std::vector<Chunk> chunks;
    size_t cur = 0, used = 0; // next byte: chunks[cur].p + used
};

// limits a contig is queued with (the kernels check them; see pjb_finish_contig_end)
struct ContigLimits {
    u32 pair_limit = 0, junc_limit = 0;
    u32 list_cap = 0; // room of a sub-list of the read lists (0: gen_list_cap(pair_limit))
    u32 sort_limit = 0; // junction ids the sort's digits are planned for (0: junc_limit).  The buffers hold junc_limit junctions -- a share of
                        // the pair limit, generous --, but digits planned for it made the sort count and scan 2048-entry tables per tile
                        // of 4096 pairs (61 MB a launch, round 4's PMC pass) where a chain has 2^17 junctions
    KeyFmt kf;
    bool dense = false; // sort ordered dense junction ids (K2d) instead of the full keys
};

// What two queued contigs must not share: control block, error word, list counters, batch descriptors and the device
// copy of the rows (the rows stream still reads them while the next contig's kernels run), the published block on the
// host, and the timing events.  Everything else is scratch of the main stream and protected by stream order.
struct CtlSlot {
    Buf cstats, err, gencount, batches, rows;
    Buf x_q, x_spos, x_send, x_gapoff, x_zlist, x_scnt, x_codes; // --extra: scratch of the target in this slot
    // what the first kernels of a contig (k1_count, k1_scan_tiles, k1_emit: the front stream) write and the rest of its
    // chain reads: the next contig's first kernels run beside this contig's last ones
    Buf tile_cnt, tile_stats, splidx, splpoff, splrec, tile_soff, chunk_tile;
    Buf scan_parts;     // k1_scan_tiles: ScanPart[K1S_BLOCKS], zeroed once; scan_epoch tells one launch's parts from the last one's
    u32 scan_epoch = 0;
    Buf members; // groups: MemberStats[GROUP_MAX] | member_junc u32[GROUP_MAX] | tile_lo u32[GROUP_MAX + 1]
    Buf okey, rec, g, jidbam; // the pairs (BAM order): intron keys, 32-byte records, [--extra: read ordinals], junction ids
    hipEvent_t ev_k1 = nullptr;
    hipEvent_t ev_xk1 = nullptr; // --extra: k1_count has left the records' spans (XOut) in the slot's scratch
    // the rest of the chain's scratch, and its streams: the chains of the two slots run side by side (most kernels of a
    // contig-sized chain are latency-bound and leave the chip half idle)
    Buf total, bitmap, wrank, ends, firstid, key[2], idx[2], hist, hist_scan, hist_part, bintotal, scan_tiles;
    Buf pagecnt, pagerank; // K2d: starts per page of the bitmap (all-zero at rest), their exclusive prefix
    Buf jid, seg, runfirst, runstart, ent, entsum, frag, fragj, acc, ancl, ancr, jkey, genlist, masks;
    bool dense_at_rest = false;
    hipStream_t main = nullptr, side = nullptr; // chain; match statistics / entropy beside it
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork2 = nullptr, ev_join2 = nullptr;
    uint8_t *pub = nullptr, *pub_dev = nullptr; // page-locked: what k6_rows_out (publish_chain) writes (host view, device view)
    DevBatch *batches_pinned = nullptr;         // page-locked staging of the batch descriptors
    size_t batches_pinned_cap = 0;
    bool at_rest = false;                       // error word / list counters are in their rest state (k6_rows_out (publish_chain) restores it)
    hipEvent_t ev[PJB_N_STAGES + 2] = {};
    hipEvent_t ev_rows = nullptr, ev_done = nullptr;
};
// optional per-kernel event brackets: one pool per control slot (collected when that contig is), one for everything
// launched outside a contig's chain (ingest, filters; collected when the timing table is read)
struct EvPool {
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    std::vector<int> name; // kernel-name index per event pair
};
constexpr int MISC_POOL = PJB_MAX_QUEUED;

// a target -- or a GROUP of targets finished as one chain (pjb_finish_group_begin) -- between _begin and _end
struct Flight {
    int32_t tid = -1;      // the first member (messages)
    std::vector<int32_t> tids;   // members, in the order of their virtual offsets (a single target: one entry)
    std::vector<int32_t> voff;   // offset of each member in the group's virtual sequence (GroupTab)
    int64_t vlen = 0;            // length of the virtual sequence
    std::vector<u32> tile_lo;    // first K1 tile of each member, + the total
    std::vector<int64_t> m_reads; // reads of each member
    std::vector<DevBatch> batches; // every member's batches, read ordinals and tile numbers running through the group
    std::vector<int> batch_member; // member of each batch
    int slot = 0;
    bool queued = false;   // its kernels are on the streams
    bool empty = false;    // no batches: nothing to queue
    bool forked = false;   // k4b_generic went to the side stream (the contig's batches must outlive it)
    ContigLimits lim;
    int64_t n_reads = 0;
    u32 n_tiles = 0;
    int attempt = 0, list_attempt = 0;
    int n_pass = 0;
    const u32 *sidx = nullptr;
    const u32 *jid_sorted = nullptr; // junction id of every sorted pair
    Pairs pr;
    // --extra: what the part that only needs the records (extra_pre) left for the part that needs the rows (extra_contig)
    bool x_pre = false;
    bool x_k1 = false; // k1_count classified the records (else: kx_classify_sparse)
    int32_t *x_spos = nullptr, *x_send = nullptr;
    u32 *x_gapoff = nullptr;
    Gap *x_gaps = nullptr;
    u32 x_gap_cap = 0;
};

constexpr size_t PJB_UP_EVENTS = 64;
struct pjb_ctx {
    pjb_config cfg;
    hipStream_t stream = nullptr;  // service stream: uploads, host batches, BAM ingest, filters, extra metrics; a contig's chain runs
                                   // on its slot's streams (CtlSlot::main / side)
    hipStream_t stream3 = nullptr; // rows stream: k6_rows_out + k6_rows_out (publish_chain) of a contig, beside the next contig's first kernels
    hipStream_t stream4 = nullptr; // header of the row mirror
    hipEvent_t ev_front = nullptr; // service stream -> chain stream
    CtlSlot sl[PJB_MAX_QUEUED];
    bool slot_busy[PJB_MAX_QUEUED] = {};
    Flight fl[PJB_MAX_QUEUED]; // FIFO: fl[0] is the oldest
    int n_fl = 0;
    int cur_slot = 0; // slot of the contig being queued / collected (extra)
    EvPool pools[PJB_MAX_QUEUED + 1];
    int cur_pool = MISC_POOL;
    Buf b_cursor;     // RowCursor
    // device ingest in pieces (pjb_bam_begin / _piece / _end)
    std::map<int32_t, struct BamStage *> bam_stage;
    // pjb_bam_begin / _piece / _pieces_done / _inflate_done may come from other threads than the context's other calls (the
    // threads that read the file hand their pieces over themselves): bam_mu guards the staging state below
    std::mutex bam_mu, err_mu;
    std::vector<Buf> genome_pool; // bases / codes of released contigs (free_contig, genome_take)
    std::vector<Buf> stage_pool;  // device buffers for staged BGZF bytes, reused target after target
    // pjb_bam_piece starts a target's bgzf_inflate as soon as its last piece is on its way (own stream, own buffers), so that
    // the inflates of several targets overlap each other and the copies: a launch takes ~50 ms whatever its size (a lane's
    // 64 KB block), and most targets fill less than the chip
    std::vector<Buf> out_pool, misc_pool;  // inflated bytes; block tables / status words / per-lane scratch
    hipStream_t inf_streams[4] = {};
    unsigned inf_next = 0;
    hipStream_t stream_up = nullptr;
    hipEvent_t ev_up = nullptr;
    hipEvent_t up_events[64] = {};
    int64_t up_ticket = 0, up_done = 0;
    std::string err;
    std::vector<int32_t> ref_len;
    std::vector<Contig> contigs;
    int32_t cur_tid = -1; // contig of the call in progress (error messages)
    std::map<int32_t, OpenContig> open;
    std::vector<Slab> slab_pool; // free slabs, reused contig after contig
    // two page-locked staging buffers: pjb_submit_batch packs the caller's arrays into one of them
    // (plain memcpy) and the DMA engine moves it to HBM while the caller decodes the next batch
    uint8_t *stage[2] = {nullptr, nullptr};
    size_t stage_cap[2] = {0, 0};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    bool stage_busy[2] = {false, false};
    unsigned stage_next = 0;
    // junction rows live in a grow-only pinned host buffer so the D2H copy is a single DMA
    pjb_junction_row *rows_pinned = nullptr;
    pjb_junction_row *rows_table = nullptr;      // the same table in HBM (k6_rows_out appends; a DMA per contig fills rows_pinned)
    bool rows_copy_pending = false;              // a DMA into rows_pinned is on stream4
    // buffers with a rest state that the kernel chain itself restores (no per-contig memsets): error word / list
    // counters (k6_rows_out (publish_chain); per slot), start bitmap / end slots (kd_reset).  false: set by a memset before use
    int last_slot = 0; // slot of the contig collected last (pjb_collect_device)
    int inflate_lanes = 512 * 64;                // lanes of one bgzf_inflate launch (2 workgroups x 256 CUs; set from the device at create)
    bool side_stream = true;                     // k4b_generic / entropy beside the main stream (pjb_set_option("overlap", 0): everything on one stream)
    bool dense_ids = true;                       // K2d (PJB_DENSE_IDS=0 sorts the full keys as round 1 did)
    u32 junc_seen = 0;                           // most junctions a contig has had so far (junction limit of the next contig)
    int lbits_seen = 18;                         // bits of the longest intron this context has met (key format of the next contig)
    size_t rows_n = 0, rows_cap = 0;             // rows collected so far; rows the HBM table holds
    size_t rows_pinned_cap = 0;                  // rows the host table holds: grown when a chain is collected, to what it brought (rows_pinned_reserve)
    std::vector<void *> graveyard;               // device buffers that were replaced by larger ones (ensure, the row table): freed by exhume()
    std::mutex grave_mu;                         // (pjb_bam_* run on the caller's threads beside the thread that queues the chains)
    size_t last_rows_n = 0; // rows of the contig finished last (still in b_rows)
    uint8_t *mirror = nullptr; // caller's device buffer filled by every finish (header + rows)
    size_t mirror_cap = 0;
    int64_t *mirror_hdr = nullptr; // page-locked staging of the header
    // the mirror accumulates: rows of every finish since the last pjb_set_row_mirror / pjb_clear_rows are appended and
    // the header holds the folded counters (a rank that owns several contigs sends ONE slot per merge)
    size_t mirror_rows = 0;
    int64_t mirror_acc[5] = {0, 0, 0, INT32_MAX, 0}; // spliced, unspliced, sum_len, min_len, max_len
    pjb_timing timing;
    int radix_max_bits = 11;
    double junc_per_read = 0;         // most junctions per read a chain of this context has had (the sort's digits of the next chain)
    u32 sort_floor = 1u << 16;        // pjb_set_option("sort_floor", n): the least number of junction ids the sort's digits are planned for (tests: small)
    u32 list_cap_forced = 0;          // pjb_set_option("list_cap", n): the read lists' first room (tests of the OVF_LISTS repeat)
    bool k1_serial = true;            // PJB_K1_SERIAL=0: the chains' K1 stages side by side
    hipEvent_t last_k1_ev = nullptr;  // the K1 stage of the chain queued last
    int k1s_blocks_forced = 0;                   // PJB_K1S_BLOCKS (tests): k1_scan_tiles on this many blocks -- 1: every tile in one block's rounds
    // optional per-kernel timing (the events live in the control slots)
    bool ktime = false;
    std::vector<std::string> knames;
    std::vector<int64_t> kcount;
    std::vector<double> kms;
    std::vector<std::string> ktime_only; // if non-empty, only these kernel names are bracketed
    // scratch
    Buf *scan_tiles = nullptr; // run_scan's tile sums: the service buffer, or the slot's while a chain is being queued
    Buf b_scan_tiles;
    Buf b_inf_comp, b_inf_out, b_inf_blocks, b_inf_status, b_inf_scratch, b_inf_bitmap; // device-side BGZF inflate
    Buf b_dfl_in, b_dfl_sym, b_dfl_slots, b_dfl_size, b_dfl_off, b_dfl_packed;           // device-side BGZF deflate
    Buf b_bam_seg, b_bam_rec, b_bam_ctl;                                  // device-side BAM record parse
    // --extra
    bool extra = false;
    std::vector<ExtraContig> xc;
    std::map<int32_t, std::pair<u64 *, u32>> filter_keys; // bamfilt: passing junctions per target (device, sorted)
    Buf f_pos, f_cigoff, f_cigar, f_codes;
    Buf g_rows, g_models, g_refs, g_out, g_bad; // filt feature rows
    Buf x_pos, x_endx, x_q, x_prefq, x_ce, x_bound, x_de, x_dropped, x_zlist, x_cnt, x_tabk, x_tabc, x_rs, x_re, x_rr, x_tileoff;
    Buf x_xrall, x_tab; // x_tab: the name table (NameSlot), x_tab_slots slots, holding the codes of x_tab_n spliced records
    size_t x_tab_slots = 0, x_tab_n = 0;
    XArena xarena;
    pjb_extra_row *xrows_pinned = nullptr;
    size_t xrows_pinned_cap = 0;
    bool extra_dense_only = false; // pjb_set_option("extra_dense", 1): the round-2 path for every target
    Buf b_hasx, b_xtotal;
    Buf b_fasta_raw; // pjb_upload_contig_fasta: the record's bytes as they are in the file
};


inline int fail(pjb_ctx *c, int code, const char *fmt, ...) {
    char tmp[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    if (c) {
        g_thread_error = tmp;
        g_thread_error_ctx = c;
        std::lock_guard<std::mutex> lk(c->err_mu);
        c->err = tmp;
    } else
        g_create_error = tmp;
    return code;
}

#define HIP_TRY(c, call)                                                                                   \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fail((c), PJB_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                                         \
    } while (0)

// the buffers ensure() has put aside go back to the device (hipFree waits for the device: call where that costs nothing)
inline void bury(pjb_ctx *c, void *p) {
    std::lock_guard<std::mutex> lk(c->grave_mu);
    c->graveyard.push_back(p);
}
inline bool exhume(pjb_ctx *c) { // (true: something was freed)
    std::vector<void *> g;
    {
        std::lock_guard<std::mutex> lk(c->grave_mu);
        g.swap(c->graveyard);
    }
    for (void *p : g) (void)hipFree(p);
    return !g.empty();
}
#ifdef PJB_DEBUG_ALLOC // (debug builds: every device buffer with its range on stderr, so that a "Memory access fault ... on address" can be placed)
#define ensure(c, b, bytes) ensure_named((c), (b), (bytes), #b, __LINE__)
inline int ensure_named(pjb_ctx *c, Buf &b, size_t bytes, const char *what, int line);
inline int ensure_impl(pjb_ctx *c, Buf &b, size_t bytes);
inline int ensure_named(pjb_ctx *c, Buf &b, size_t bytes, const char *what, int line) {
    const void *was = b.p;
    const int rc = ensure_impl(c, b, bytes);
    if (b.p != was) fprintf(stderr, "[alloc] %s (line %d): %p .. %p (%zu bytes, asked %zu)\n", what, line, b.p, (void *)((char *)b.p + b.cap), b.cap, bytes);
    return rc;
}
inline int ensure_impl(pjb_ctx *c, Buf &b, size_t bytes) {
#else
inline int ensure(pjb_ctx *c, Buf &b, size_t bytes) {
#endif
    if (bytes <= b.cap && b.p) return PJB_OK;
    // A buffer that has to grow is not freed here: hipFree waits for the whole device -- the inflate of the next targets, the copies in
    // flight -- and every thread that calls into the runtime meanwhile waits with it (end to end: 15 - 40 ms of nothing at the first
    // chains).  The old buffer goes to the context's graveyard, which is emptied where a wait costs nothing (exhume: pjb_clear_rows,
    // pjb_destroy) or when an allocation fails.  A buffer grows by a quarter at least, so what lies there is a few times the live size at most.
    if (b.p) bury(c, b.p);
    b.p = nullptr;
    b.cap = 0;
    size_t want = std::max<size_t>(bytes + bytes / 4, 256);
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (exhume(c)) e = hipMalloc(&b.p, want);
    }
    if (e != hipSuccess) {
        want = std::max<size_t>(bytes, 256);
        e = hipMalloc(&b.p, want);
        if (e != hipSuccess) return fail(c, PJB_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    // test hook (tests/test_gpu_poison.py): PJB_POISON=1 fills every new device buffer with a pattern.  Fresh device memory is
    // usually zero, and a kernel that reads what nobody wrote gets away with it until the allocator hands out a used page
    // (round 4: one run of `junc` in thirty died of it); with the pattern it fails every time.
    static const bool poison = getenv("PJB_POISON") != nullptr && strcmp(getenv("PJB_POISON"), "0") != 0;
    if (poison) {
        (void)hipMemset(b.p, 0xCD, want);
        (void)hipDeviceSynchronize();
    }
    return PJB_OK;
}

inline void release(Buf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

inline const char *err_text(int code) {
    switch (code) {
    case PJB_ERR_BAD_XS: return "Unknown strand: XS tag is not one of + - ? .";
    case PJB_ERR_NO_PRESENCE: return "Found an alignment that does not have a presence in the requested region";
    case PJB_ERR_ZERO_LEN_OP: return "Can't extract cigar op sequence from query string when length has been calculated as 0";
    case PJB_ERR_QUERY_RANGE: return "Can't extract cigar op sequence from query string";
    case PJB_ERR_GENOME_RANGE: return "Can't extract cigar op sequence from extracted genome region";
    case PJB_ERR_QREGION: return "Query region is outside the genomic region";
    case PJB_ERR_ANCHOR_MISMATCH: return "Anchor region for query and genome are not the same size";
    case PJB_ERR_SPLICE_SITE_LEN: return "Retrieved sequence for splice site of junction is not the expected length";
    case PJB_ERR_ANCHOR_LEN: return "Retrieved sequence for anchor of junction is not the expected length";
    case PJB_ERR_INTRON_FLANK_LEN: return "Retrieved sequence for intron region of junction is not the expected length";
    case PJB_ERR_MIN_ANCHOR: return "The intron must lie inside its anchors (Intron::minAnchorLength)";
    case PJB_ERR_HAMMING_LEN: return "Can't find hamming distance of strings that are not the same length";
    case PJB_ERR_CLIP_RANGE: return "Soft clip longer than the read (basic_string::substr)";
    case PJB_ERR_UNSORTED: return "Alignments are not coordinate sorted";
    case PJB_ERR_DIVERGENT: return "Malformed CIGAR: padded query and genome walks disagree";
    case PJB_ERR_NO_SEQ: return "A spliced alignment was submitted without its sequence";
    default: return "unknown error";
    }
}

inline int check_device_error(pjb_ctx *c, u64 e) {
    if (e == ~0ull) return PJB_OK;
    const int code = -(int)(e & 0xff);
    const unsigned long long ord = e >> 8;
    return fail(c, code, "%s (alignment ordinal %llu on target %d)", err_text(code), ord, c->cur_tid);
}

// host-side copy into page-locked staging memory, split over a few threads for large blocks
inline void parallel_copy(void *dst, const void *src, size_t bytes) {
    const size_t MIN_SLICE = (size_t)4 << 20;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t nthr = std::min<size_t>(std::min<size_t>(8, hw), bytes / MIN_SLICE);
    if (nthr <= 1) {
        memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> th;
    const size_t per = ((bytes + nthr - 1) / nthr + 63) & ~(size_t)63;
    for (size_t t = 0; t < nthr; t++) {
        const size_t a = std::min(bytes, per * t), b = std::min(bytes, a + per);
        if (a < b) th.emplace_back([=] { memcpy((uint8_t *)dst + a, (const uint8_t *)src + a, b - a); });
    }
    for (auto &x : th) x.join();
}

inline int bits_of(uint64_t v) {
    int b = 0;
    while (v) {
        b++;
        v >>= 1;
    }
    return b;
}

// kernel launch with optional event bracketing -------------------------------------------------
inline int kname_index(pjb_ctx *c, const char *name) {
    for (size_t i = 0; i < c->knames.size(); i++)
        if (c->knames[i] == name) return (int)i;
    c->knames.push_back(name);
    c->kcount.push_back(0);
    c->kms.push_back(0.0);
    return (int)c->knames.size() - 1;
}
inline void ev_begin(pjb_ctx *c, const char *name) {
    EvPool &S = c->pools[c->cur_pool];
    if (S.used + 2 > S.ev.size()) {
        S.ev.resize(S.used + 2);
        (void)hipEventCreate(&S.ev[S.used]);
        (void)hipEventCreate(&S.ev[S.used + 1]);
    }
    S.name.push_back(kname_index(c, name));
    (void)hipEventRecord(S.ev[S.used], c->stream);
}
inline void ev_end(pjb_ctx *c) {
    EvPool &S = c->pools[c->cur_pool];
    (void)hipEventRecord(S.ev[S.used + 1], c->stream);
    S.used += 2;
}
inline void ev_collect(pjb_ctx *c, int pool) { // the pool's events must have completed
    EvPool &S = c->pools[pool];
    for (size_t k = 0; k < S.name.size(); k++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, S.ev[2 * k], S.ev[2 * k + 1]) == hipSuccess) {
            c->kcount[(size_t)S.name[k]]++;
            c->kms[(size_t)S.name[k]] += ms;
        }
    }
    S.name.clear();
    S.used = 0;
}
inline void ev_drop(pjb_ctx *c, int pool) {
    c->pools[pool].name.clear();
    c->pools[pool].used = 0;
}
inline bool ktime_wanted(pjb_ctx *c, const char *name) {
    if (!c->ktime) return false;
    if (c->ktime_only.empty()) return true;
    for (auto &n : c->ktime_only)
        if (n == name) return true;
    return false;
}
// PJB_DEBUG_LAUNCH (a build flag, tools/build_variants.sh): every chain kernel is announced on stderr and waited for, so that the
// last name before a "Memory access fault" is the kernel that faulted
#ifdef PJB_DEBUG_LAUNCH
#define PJB_LAUNCH_TRACE(c, name)                                   \
    do {                                                            \
        (void)hipStreamSynchronize((c)->stream);                    \
        fprintf(stderr, "[launch] %s done\n", name);                \
    } while (0)
#define PJB_LAUNCH_ANNOUNCE(name) fprintf(stderr, "[launch] %s ...\n", name)
#else
#define PJB_LAUNCH_TRACE(c, name) do { } while (0)
#define PJB_LAUNCH_ANNOUNCE(name) do { } while (0)
#endif
#define LAUNCH_LDS(c, name, kern, grid, block, lds_bytes, ...)                       \
    do {                                                                            \
        const bool timed_ = ktime_wanted((c), name);                                \
        if (timed_) ev_begin((c), name);                                            \
        PJB_LAUNCH_ANNOUNCE(name);                                                  \
        hipLaunchKernelGGL(kern, grid, block, lds_bytes, (c)->stream, __VA_ARGS__); \
        if (timed_) ev_end((c));                                                    \
        HIP_TRY((c), hipGetLastError());                                            \
        PJB_LAUNCH_TRACE(c, name);                                                  \
    } while (0)
#define LAUNCH(c, name, kern, grid, block, ...) LAUNCH_LDS(c, name, kern, grid, block, 0, __VA_ARGS__)

// generic scan launchers ------------------------------------------------------------------------
template <typename F, typename G>
inline int run_scan(pjb_ctx *c, const char *tag, F f, G g, u64 n, u64 *d_total, const u32 *d_n = nullptr) {
    const u32 nt = std::max<u32>(1, (u32)((n + SCAN_TILE - 1) / SCAN_TILE)); // (an empty input still gets its total written)
    Buf &tiles = c->scan_tiles ? *c->scan_tiles : c->b_scan_tiles;
    int rc = ensure(c, tiles, (size_t)nt * 8);
    if (rc) return rc;
    u64 *ts = (u64 *)tiles.p;
    std::string t = tag;
    LAUNCH(c, (t + "_reduce").c_str(), (scan_reduce_kernel<F>), dim3(nt), dim3(256), f, n, ts, d_n);
    if (nt <= SCAN2_MAX_TILES) { // contig-sized: the apply blocks add up the tile sums before them themselves
        LAUNCH(c, (t + "_apply").c_str(), (scan_apply2_kernel<F, G>), dim3(nt), dim3(256), f, g, n, (const u64 *)ts, d_n, d_total);
        return PJB_OK;
    }
    LAUNCH(c, (t + "_tiles").c_str(), scan_tiles_kernel, dim3(1), dim3(1024), ts, nt, d_total);
    LAUNCH(c, (t + "_apply").c_str(), (scan_apply_kernel<F, G>), dim3(nt), dim3(256), f, g, n, (const u64 *)ts, d_n);
    return PJB_OK;
}

inline void *slab_alloc(pjb_ctx *c, OpenContig &oc, size_t bytes) {
    bytes = (std::max<size_t>(bytes, 16) + 255) & ~(size_t)255;
    for (auto &s : oc.slabs)
        if (s.cap - s.used >= bytes) {
            void *r = s.p + s.used;
            s.used += bytes;
            return r;
        }
    {
        // (a target's first slab: the smallest pooled one that holds what the target is expected to take in all, else a new one of that size)
        const size_t want = oc.slabs.empty() ? std::max(bytes, oc.slab_hint) : bytes;
        long best = -1;
        for (size_t k = 0; k < c->slab_pool.size(); k++)
            if (c->slab_pool[k].cap >= want && (best < 0 || c->slab_pool[k].cap < c->slab_pool[(size_t)best].cap)) best = (long)k;
        if (best >= 0) {
            Slab s = c->slab_pool[(size_t)best];
            c->slab_pool.erase(c->slab_pool.begin() + best);
            s.used = bytes;
            oc.slabs.push_back(s);
            return s.p;
        }
    }
    Slab s;
    s.cap = std::max<size_t>(bytes, std::max<size_t>((size_t)128 << 20, oc.slabs.empty() ? oc.slab_hint : 0));
    if (hipMalloc((void **)&s.p, s.cap) != hipSuccess) {
        s.cap = bytes;
        if (hipMalloc((void **)&s.p, s.cap) != hipSuccess) return nullptr;
    }
    s.used = bytes;
    oc.slabs.push_back(s);
#ifdef PJB_DEBUG_ALLOC
    fprintf(stderr, "[alloc] slab: %p .. %p (%zu bytes)\n", (void *)s.p, (void *)(s.p + s.cap), s.cap);
#endif
    return s.p;
}

inline void extra_clear(pjb_ctx *c) {
    for (auto &x : c->xc) {
        if (!x.dense) continue; // (everything else is arena memory)
        if (x.cover) (void)hipFree(x.cover);
        if (x.xr) (void)hipFree(x.xr);
        if (x.pair_code) (void)hipFree(x.pair_code);
        if (x.pair_row) (void)hipFree(x.pair_row);
        if (x.spl_codes) (void)hipFree(x.spl_codes);
    }
    c->xc.clear();
    c->xarena.cur = c->xarena.used = 0;
    c->x_tab_n = 0; // (the table is wiped when the next file's first codes arrive)
}

// `bytes` of arena memory (256-byte aligned), nullptr when the device is out of memory
inline void *xarena_alloc(pjb_ctx *c, size_t bytes) {
    XArena &A = c->xarena;
    bytes = (std::max<size_t>(bytes, 16) + 255) & ~(size_t)255;
    for (; A.cur < A.chunks.size(); A.cur++, A.used = 0)
        if (A.chunks[A.cur].cap - A.used >= bytes) {
            void *r = A.chunks[A.cur].p + A.used;
            A.used += bytes;
            return r;
        }
    XArena::Chunk ch;
    ch.cap = std::max<size_t>(bytes, (size_t)256 << 20);
    if (hipMalloc((void **)&ch.p, ch.cap) != hipSuccess) {
        ch.cap = bytes;
        if (hipMalloc((void **)&ch.p, ch.cap) != hipSuccess) return nullptr;
    }
    A.chunks.push_back(ch);
    A.cur = A.chunks.size() - 1;
    A.used = bytes;
    return ch.p;
}

inline int close_contig(pjb_ctx *c, int32_t tid) {
    auto it = c->open.find(tid);
    if (it == c->open.end()) return PJB_OK;
    for (auto &s : it->second.slabs) {
        s.used = 0;
        c->slab_pool.push_back(s);
    }
    c->open.erase(it);
    return PJB_OK;
}

int upload_staged(pjb_ctx *c, void *dst, const uint8_t *src, size_t bytes); // (defined with the ingest code)

// ---- defined in one unit, used by another
int upload_staged(pjb_ctx *c, void *dst, const uint8_t *src, size_t bytes);                                   // pjb_ingest_api.hip (FASTA bytes through the staging buffers)
void bam_stage_clear(pjb_ctx *c);                                                                              // pjb_ingest_api.hip (pjb_destroy)
int extra_pre(pjb_ctx *c, Flight &f);                                                                          // pjb_extra_api.hip (the chain queues a target's extra metrics)
int extra_contig(pjb_ctx *c, Flight &f, int32_t tid, u64 n_spliced, u32 P, u32 J, size_t row_base);            // pjb_extra_api.hip
// the host row table is complete up to rows_n
inline int rows_sync(pjb_ctx *c) {
    if (!c->rows_copy_pending) return PJB_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream4));
    c->rows_copy_pending = false;
    return PJB_OK;
}
constexpr u32 X_ZCAP = 1u << 20; // --extra: room of the list of unspliced records without a span
void ingest_kernel_attributes();  // pjb_ingest_api.hip: the inflate kernel's dynamic LDS (pjb_create sets the kernels' attributes on a thread of its own)
int ingest_lds_bytes();           // pjb_ingest_api.hip: LDS a workgroup of the inflate kernel takes (lanes of one launch)
