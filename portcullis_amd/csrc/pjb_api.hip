// pjb_api.hip -- C ABI (include/portcullis_amd.h) over the HIP kernels.
// One context = one HIP device + one stream + a grow-only scratch arena.
#include "pjb_kernels.hip.h"
#include "pjb_extra.hip.h"
#include "pjb_deflate.hip.h"
#include "pjb_ingest.hip.h"

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

namespace {

thread_local std::string g_create_error;
// The message of the last failing call of THIS thread (pjb_last_error returns it): pjb_bam_begin / _piece / _pieces_done /
// _inflate_done may run on other threads than the context's other calls, and a thread must neither read a string another
// thread is reassigning nor report another thread's failure.
thread_local std::string g_thread_error;
thread_local const void *g_thread_error_ctx = nullptr; // the context the message belongs to (a thread may drive several)

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
    size_t d_cap = 0, codes_cap = 0; // sizes of the allocations (they go back to the context's genome pool)
};

// The bases and codes of a released contig are kept for the next upload (targets come longest first, so the next genome
// fits): a hipMalloc / hipFree pair of 250 MB is ~10 ms on the thread that serves every target.
void free_contig(Contig &g, std::vector<Buf> *pool = nullptr) {
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
static void *genome_take(std::vector<Buf> &pool, size_t bytes, size_t &cap) {
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

} // namespace

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
    size_t rows_n = 0, rows_cap = 0;
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

namespace {

int fail(pjb_ctx *c, int code, const char *fmt, ...) {
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

#ifdef PJB_DEBUG_ALLOC // (debug builds: every device buffer with its range on stderr, so that a "Memory access fault ... on address" can be placed)
#define ensure(c, b, bytes) ensure_named((c), (b), (bytes), #b, __LINE__)
int ensure_named(pjb_ctx *c, Buf &b, size_t bytes, const char *what, int line);
int ensure_impl(pjb_ctx *c, Buf &b, size_t bytes);
int ensure_named(pjb_ctx *c, Buf &b, size_t bytes, const char *what, int line) {
    const void *was = b.p;
    const int rc = ensure_impl(c, b, bytes);
    if (b.p != was) fprintf(stderr, "[alloc] %s (line %d): %p .. %p (%zu bytes, asked %zu)\n", what, line, b.p, (void *)((char *)b.p + b.cap), b.cap, bytes);
    return rc;
}
int ensure_impl(pjb_ctx *c, Buf &b, size_t bytes) {
#else
int ensure(pjb_ctx *c, Buf &b, size_t bytes) {
#endif
    if (bytes <= b.cap && b.p) return PJB_OK;
    if (b.p) HIP_TRY(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = std::max<size_t>(bytes + bytes / 4, 256);
    hipError_t e = hipMalloc(&b.p, want);
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

void release(Buf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

const char *err_text(int code) {
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

int check_device_error(pjb_ctx *c, u64 e) {
    if (e == ~0ull) return PJB_OK;
    const int code = -(int)(e & 0xff);
    const unsigned long long ord = e >> 8;
    return fail(c, code, "%s (alignment ordinal %llu on target %d)", err_text(code), ord, c->cur_tid);
}

// host-side copy into page-locked staging memory, split over a few threads for large blocks
void parallel_copy(void *dst, const void *src, size_t bytes) {
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

int bits_of(uint64_t v) {
    int b = 0;
    while (v) {
        b++;
        v >>= 1;
    }
    return b;
}

// kernel launch with optional event bracketing -------------------------------------------------
int kname_index(pjb_ctx *c, const char *name) {
    for (size_t i = 0; i < c->knames.size(); i++)
        if (c->knames[i] == name) return (int)i;
    c->knames.push_back(name);
    c->kcount.push_back(0);
    c->kms.push_back(0.0);
    return (int)c->knames.size() - 1;
}
void ev_begin(pjb_ctx *c, const char *name) {
    EvPool &S = c->pools[c->cur_pool];
    if (S.used + 2 > S.ev.size()) {
        S.ev.resize(S.used + 2);
        (void)hipEventCreate(&S.ev[S.used]);
        (void)hipEventCreate(&S.ev[S.used + 1]);
    }
    S.name.push_back(kname_index(c, name));
    (void)hipEventRecord(S.ev[S.used], c->stream);
}
void ev_end(pjb_ctx *c) {
    EvPool &S = c->pools[c->cur_pool];
    (void)hipEventRecord(S.ev[S.used + 1], c->stream);
    S.used += 2;
}
void ev_collect(pjb_ctx *c, int pool) { // the pool's events must have completed
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
void ev_drop(pjb_ctx *c, int pool) {
    c->pools[pool].name.clear();
    c->pools[pool].used = 0;
}
bool ktime_wanted(pjb_ctx *c, const char *name) {
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
int run_scan(pjb_ctx *c, const char *tag, F f, G g, u64 n, u64 *d_total, const u32 *d_n = nullptr) {
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

void *slab_alloc(pjb_ctx *c, OpenContig &oc, size_t bytes) {
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

void extra_clear(pjb_ctx *c) {
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
void *xarena_alloc(pjb_ctx *c, size_t bytes) {
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

int close_contig(pjb_ctx *c, int32_t tid) {
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
} // namespace
static int slot_init(pjb_ctx *c, int k);
static void aux_streams(pjb_ctx *c) { // the rows stream and the row mirror's (15 - 20 ms each to create)
    if (!c->stream3) (void)hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking);
    if (!c->stream4) (void)hipStreamCreateWithFlags(&c->stream4, hipStreamNonBlocking);
}

extern "C" {

// Page-locked host memory.  hipHostMalloc zeroes and pins on the calling thread at ~4.6 GB/s (143 ms for the 768 MB ring of
// the end-to-end program, as much again for its genome buffers); anonymous huge-page memory touched by four threads and then
// registered is the same memory to a DMA (56.8 GB/s either way) after 23 ms (tools/debug/register_probe.hip,
// profiles/r03ap2_register_probe.txt).  Blocks below 8 MB, and anything mmap or the registration refuses, take hipHostMalloc.
namespace {
std::mutex g_host_mu;
std::map<void *, size_t> g_host_registered; // blocks of pjb_host_alloc that are mmap + hipHostRegister (value: mapped bytes)
} // namespace
void *pjb_host_alloc(size_t bytes) {
    if (bytes >= ((size_t)8 << 20)) {
        const size_t huge = (size_t)2 << 20, len = (bytes + huge - 1) & ~(huge - 1);
        void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p != MAP_FAILED) {
            (void)madvise(p, len, MADV_HUGEPAGE);
            const size_t nt = std::min<size_t>(4, len >> 24);
            auto touch = [&](size_t t) {
                const size_t per = ((len / std::max<size_t>(nt, 1)) + huge - 1) & ~(huge - 1), a = std::min(len, per * t), b = std::min(len, a + per);
                for (size_t o = a; o < b; o += 4096) ((volatile uint8_t *)p)[o] = 0;
            };
            if (nt <= 1) touch(0);
            else {
                std::vector<std::thread> th;
                for (size_t t = 0; t < nt; t++) th.emplace_back(touch, t);
                for (auto &x : th) x.join();
            }
            if (hipHostRegister(p, len, hipHostRegisterDefault) == hipSuccess) {
                std::lock_guard<std::mutex> lk(g_host_mu);
                g_host_registered[p] = len;
                return p;
            }
            (void)hipGetLastError();
            munmap(p, len);
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void pjb_host_free(void *p) {
    if (!p) return;
    size_t len = 0;
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        auto it = g_host_registered.find(p);
        if (it != g_host_registered.end()) {
            len = it->second;
            g_host_registered.erase(it);
        }
    }
    if (len) {
        (void)hipHostUnregister(p);
        munmap(p, len);
    } else
        (void)hipHostFree(p);
}

int pjb_host_register(void *p, size_t bytes) {
    if (!p || !bytes) return PJB_ERR_ARG;
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();
        return PJB_ERR_HIP;
    }
    return PJB_OK;
}
int pjb_host_unregister(void *p) {
    if (!p) return PJB_ERR_ARG;
    if (hipHostUnregister(p) != hipSuccess) {
        (void)hipGetLastError();
        return PJB_ERR_HIP;
    }
    return PJB_OK;
}

int pjb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pjb_create(pjb_ctx **out, const pjb_config *cfg) {
    if (!out || !cfg) return fail(nullptr, PJB_ERR_ARG, "pjb_create: null argument");
    *out = nullptr;
    if (cfg->abi_version != PJB_ABI_VERSION && cfg->abi_version != 3) // (3: the same entry points; pjb_batch ends at name_hash)
        return fail(nullptr, PJB_ERR_ARG, "pjb_create: ABI version %d, library is %d", cfg->abi_version, PJB_ABI_VERSION);
    if (cfg->orientation < PJB_OR_SE || cfg->orientation > PJB_OR_UNKNOWN)
        return fail(nullptr, PJB_ERR_ARG, "pjb_create: bad orientation %d", cfg->orientation);
    const bool ctrace = getenv("PJB_CREATE_TRACE") != nullptr; // (stderr: what the context's start is made of)
    const auto ct0 = std::chrono::steady_clock::now();
    auto cmark = [&](const char *what) {
        if (ctrace) fprintf(stderr, "[pjb_create] %7.1f ms: %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ct0).count(), what);
    };
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    cmark("hipGetDeviceCount");
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, PJB_ERR_NO_DEVICE,
                    "no HIP device available (%s); the junc hot path has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= n)
        return fail(nullptr, PJB_ERR_ARG, "pjb_create: device %d out of range (have %d)", cfg->device, n);
    e = hipSetDevice(cfg->device);
    if (e != hipSuccess) return fail(nullptr, PJB_ERR_HIP, "hipSetDevice(%d): %s", cfg->device, hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, cfg->device);
    if (e != hipSuccess) return fail(nullptr, PJB_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    cmark("hipSetDevice, hipGetDeviceProperties");
    const int n_cu = prop.multiProcessorCount;
    if (prop.warpSize != 64)
        return fail(nullptr, PJB_ERR_NO_DEVICE, "device %d (%s) is not a wave64 CDNA device", cfg->device, prop.gcnArchName);
    pjb_ctx *c = new (std::nothrow) pjb_ctx();
    if (!c) return fail(nullptr, PJB_ERR_NOMEM, "out of host memory");
    // The kernels' attributes -- the first call loads the code object, 30 - 70 ms -- are set by a thread of their own while
    // this one creates the streams (PJB_CREATE_SERIAL=1: afterwards, on this thread).
    const int dev = cfg->device;
    auto attributes = [dev] {
        (void)hipSetDevice(dev);
        (void)hipFuncSetAttribute((const void *)bgzf_decode, hipFuncAttributeMaxDynamicSharedMemorySize, I3_LDS_BYTES);
        // 12-bit digits need more dynamic LDS than the 64 KB a kernel gets without asking
        (void)hipFuncSetAttribute((const void *)rs_scatter<0, u64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rs_scatter_lds_bytes(RS_MAX_BITS));
        (void)hipFuncSetAttribute((const void *)rs_scatter<0, u32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rs_scatter_lds_bytes(RS_MAX_BITS, 4));
    };
    std::thread attr_thread(attributes);
    struct JoinAttr {
        std::thread &t;
        ~JoinAttr() {
            if (t.joinable()) t.join();
        }
    } join_attr{attr_thread};
    c->cfg = *cfg;
    memset(&c->timing, 0, sizeof c->timing);
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(nullptr, PJB_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    if (!(cfg->flags & PJB_FLAG_NO_CHAINS)) aux_streams(c); // (else: with the first chain)
    (void)hipEventCreateWithFlags(&c->ev_front, hipEventDisableTiming);
    c->scan_tiles = &c->b_scan_tiles;
    // The first four control slots get their streams and events now; the others when they are first used (slot_init).  Creating
    // a stream is ~10 ms on an idle device -- and blocked for 1.9 s once when it happened in the middle of an end-to-end run
    // (the runtime creates a hardware queue behind whatever the device is doing): a caller that wants more than four chains in
    // flight on a busy device queues that deep once, early.
    // (60 ms for four slots' streams and events; from four threads at once it is 70 - 100 ms: the runtime serialises them)
    cmark("main streams");
    if (!(cfg->flags & PJB_FLAG_NO_CHAINS))
        for (int k = 0; k < 4 && k < PJB_MAX_QUEUED; k++) (void)slot_init(c, k);
    cmark("chain slots");
    c->inflate_lanes = std::max(1, n_cu) * (160 * 1024 / I3_LDS_BYTES) * 64;
    c->ktime = (cfg->flags & PJB_FLAG_KERNEL_TIMING) != 0;
    c->extra = (cfg->flags & PJB_FLAG_EXTRA) != 0;
    if (const char *s = getenv("PJB_K1_SERIAL")) c->k1_serial = atoi(s) != 0;
    if (const char *s = getenv("PJB_K1S_BLOCKS")) c->k1s_blocks_forced = std::max(0, std::min(atoi(s), (int)K1S_BLOCKS));
    if (const char *s = getenv("PJB_RADIX_BITS")) {
        int v = atoi(s);
        if (v >= 4 && v <= RS_MAX_BITS) c->radix_max_bits = v;
    }
    cmark("options");
    attr_thread.join();
    cmark("kernel attributes");
    *out = c;
    return PJB_OK;
}

static void bam_stage_clear(pjb_ctx *c); // (defined with the staged ingest)

void pjb_destroy(pjb_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    (void)hipDeviceSynchronize(); // (contigs may still be queued)
#ifdef K1E_PROF
    {
        unsigned long long h[16] = {0};
        (void)hipDeviceSynchronize();
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(pjb::g_k1e_prof), sizeof h) == hipSuccess) {
            unsigned long long tot = 0;
            for (int i = 0; i < 12; i++) tot += h[i];
            static const char *nm[12] = {"block start", "prologue", "shapes (wait ops)", "windows issue", "next records issue", "stage wait", "next ops issue", "compare+emit", "lists", "cand flush", "span of bases", "bases issue"};
            for (int i = 0; i < 12; i++) fprintf(stderr, "[k1e_prof] %-20s %6.2f %%  %llu\n", nm[i], tot ? 100.0 * (double)h[i] / (double)tot : 0.0, h[i]);
        }
    }
#endif
#ifdef K1E_HIST
    { // compare rounds per wavefront and trip of k1_emit: what ran (the longest lane's) against what the lanes needed
        unsigned long long h[4][32] = {{0}};
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(pjb::g_k1e_hist), sizeof h) == hipSuccess) {
            unsigned long long trips = 0, run = 0, need = 0, lanes = 0;
            for (int r = 0; r < 32; r++) trips += h[0][r], run += h[0][r] * (unsigned long long)r, need += h[1][r], lanes += h[2][r];
            fprintf(stderr, "[k1e_hist] 2-bit rounds: %llu wavefront-trips, %.3f rounds run a trip, %.3f rounds a lane needs (max / mean %.3f)\n", trips,
                    trips ? (double)run / (double)trips : 0.0, lanes ? (double)need / (double)lanes : 0.0,
                    need ? ((double)run / (double)trips) / ((double)need / (double)lanes) : 0.0);
            for (int r = 0; r < 32; r++)
                if (h[0][r] || h[3][r])
                    fprintf(stderr, "[k1e_hist]   %2d rounds: %10llu trips (2-bit)  %10llu trips (4-bit)\n", r, h[0][r], h[3][r]);
        }
    }
#endif
    while (!c->open.empty()) close_contig(c, c->open.begin()->first);
    extra_clear(c);
    for (auto &kv : c->filter_keys)
        if (kv.second.first) (void)hipFree(kv.second.first);
    c->filter_keys.clear();
    for (auto &g : c->contigs) free_contig(g);
    for (auto &b : c->genome_pool) release(b);
    c->genome_pool.clear();
    for (auto &sl : c->slab_pool)
        if (sl.p) (void)hipFree(sl.p);
    if (c->rows_pinned) (void)hipHostFree(c->rows_pinned);
    if (c->rows_table) (void)hipFree(c->rows_table);
    bam_stage_clear(c);
    for (int k = 0; k < PJB_MAX_QUEUED; k++) {
        CtlSlot &S = c->sl[k];
        if (S.pub) (void)hipHostFree(S.pub);
        if (S.batches_pinned) (void)hipHostFree(S.batches_pinned);
        for (auto &ev : S.ev)
            if (ev) (void)hipEventDestroy(ev);
        if (S.ev_rows) (void)hipEventDestroy(S.ev_rows);
        if (S.ev_done) (void)hipEventDestroy(S.ev_done);
        Buf *sb[] = {&S.x_q, &S.x_spos, &S.x_send, &S.x_gapoff, &S.x_zlist, &S.x_scnt, &S.x_codes, &S.cstats, &S.err, &S.gencount, &S.batches, &S.rows, &S.tile_cnt, &S.tile_stats, &S.splidx, &S.splpoff, &S.splrec, &S.tile_soff, &S.chunk_tile, &S.scan_parts, &S.members, &S.okey, &S.g,
                     &S.rec, &S.jidbam, &S.jkey, &S.total, &S.bitmap, &S.wrank, &S.pagecnt, &S.pagerank, &S.ends, &S.firstid,
                     &S.key[0], &S.key[1], &S.idx[0], &S.idx[1], &S.hist, &S.hist_scan, &S.hist_part, &S.bintotal, &S.scan_tiles, &S.jid, &S.seg, &S.runfirst,
                     &S.runstart, &S.ent, &S.entsum, &S.frag, &S.fragj, &S.masks, &S.acc, &S.ancl, &S.ancr, &S.genlist};
        for (Buf *b : sb) release(*b);
        hipEvent_t evs[] = {S.ev_k1, S.ev_xk1, S.ev_fork, S.ev_join, S.ev_fork2, S.ev_join2};
        for (hipEvent_t e : evs)
            if (e) (void)hipEventDestroy(e);
        if (S.main) (void)hipStreamDestroy(S.main);
        if (S.side) (void)hipStreamDestroy(S.side);
    }
    if (c->mirror_hdr) (void)hipHostFree(c->mirror_hdr);
    for (int k = 0; k < 2; k++) {
        if (c->stage[k]) (void)hipHostFree(c->stage[k]);
        if (c->stage_ev[k]) (void)hipEventDestroy(c->stage_ev[k]);
    }
    Buf *all[] = {&c->b_cursor, &c->b_scan_tiles, &c->b_hasx, &c->b_xtotal, &c->b_fasta_raw, &c->b_inf_comp, &c->b_inf_out, &c->b_inf_blocks, &c->b_inf_status, &c->b_inf_scratch, &c->b_inf_bitmap,
                  &c->b_bam_seg, &c->b_bam_rec, &c->b_bam_ctl, &c->f_pos, &c->f_cigoff, &c->f_cigar, &c->f_codes, &c->g_rows, &c->g_models, &c->g_refs,
                  &c->g_out, &c->g_bad,
                  &c->x_pos, &c->x_endx, &c->x_q, &c->x_prefq, &c->x_ce, &c->x_bound, &c->x_de, &c->x_dropped, &c->x_zlist, &c->x_cnt,
                  &c->x_tabk, &c->x_tabc, &c->x_rs, &c->x_re, &c->x_rr, &c->x_tileoff, &c->x_xrall, &c->x_tab,
                  &c->b_dfl_in, &c->b_dfl_sym, &c->b_dfl_slots, &c->b_dfl_size, &c->b_dfl_off, &c->b_dfl_packed};
    for (Buf *b : all) release(*b);
    for (auto &ch : c->xarena.chunks) (void)hipFree(ch.p);
    if (c->xrows_pinned) (void)hipHostFree(c->xrows_pinned);
    for (auto &pool : c->pools)
        for (auto &ev : pool.ev) (void)hipEventDestroy(ev);
    if (c->ev_front) (void)hipEventDestroy(c->ev_front);
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    if (c->stream4) (void)hipStreamDestroy(c->stream4);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *pjb_last_error(const pjb_ctx *c) {
    if (!c) return g_create_error.c_str();
    return g_thread_error_ctx == c ? g_thread_error.c_str() : ""; // (this thread's last failure on THIS context; none: the empty string)
}

int pjb_set_refs(pjb_ctx *c, int32_t n_refs, const int32_t *ref_len) {
    if (!c) return PJB_ERR_ARG;
    if (n_refs < 0 || (n_refs > 0 && !ref_len)) return fail(c, PJB_ERR_ARG, "pjb_set_refs: bad arguments");
    if (!c->open.empty()) return fail(c, PJB_ERR_STATE, "pjb_set_refs: contig %d is still open", c->open.begin()->first);
    for (auto &g : c->contigs) free_contig(g);
    c->ref_len.assign(ref_len, ref_len + n_refs);
    c->contigs.assign((size_t)n_refs, Contig());
    return PJB_OK;
}

static int upload_common(pjb_ctx *c, int32_t tid, uint8_t *d, int64_t len, bool owned, bool do_upper, size_t d_cap = 0) {
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    int rc = ensure(c, c->b_hasx, sizeof(int));
    if (rc) return rc;
    HIP_TRY(c, hipMemsetAsync(c->b_hasx.p, 0, sizeof(int), c->stream));
    if (len > 0) {
        const int64_t nthreads = (len + 15) / 16;
        const unsigned nblk = (unsigned)((nthreads + 255) / 256);
        hipLaunchKernelGGL(k0_upper, dim3(nblk), dim3(256), 0, c->stream, d, len, do_upper ? 1 : 0, (int *)c->b_hasx.p);
    }
    int hx = 0;
    HIP_TRY(c, hipMemcpyAsync(&hx, c->b_hasx.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    // 4-bit codes for the word-parallel compare in k4 (after upper-casing)
    u32 *codes = nullptr;
    size_t codes_cap = 0;
    int exotic = 0;
    const int64_t n_words = (len + 7) / 8;
    static const bool no_seq2 = getenv("PJB_NO_SEQ2") && atoi(getenv("PJB_NO_SEQ2")) != 0;
    const size_t c2_at = ((size_t)(n_words + 2) + 3) & ~(size_t)3;                       // (the 2-bit codes start on a 16-byte boundary)
    const size_t c2_words = len > 0 && !no_seq2 ? (size_t)codes2_alloc_words(len) : 0;
    if (len > 0) {
        codes = (u32 *)genome_take(c->genome_pool, (c2_at + c2_words) * 4, codes_cap);
        if (!codes) return fail(c, PJB_ERR_NOMEM, "hipMalloc(genome codes, %zu bytes) failed", (c2_at + c2_words) * 4);
        (void)hipMemsetAsync(c->b_hasx.p, 0, sizeof(int), c->stream);
        (void)hipMemsetAsync(codes + n_words, 0, 8, c->stream);
        hipLaunchKernelGGL(k0_encode, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, c->stream, (const uint8_t *)d, len,
                           codes, n_words, (int *)c->b_hasx.p);
        (void)hipMemcpyAsync(&exotic, c->b_hasx.p, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    }
    // 2-bit codes and their exception bitmap for k1_emit's compares (PJB_NO_SEQ2=1: not built -- A/B runs): behind the 4-bit codes, in
    // the same allocation (a hipMalloc is milliseconds on the thread that serves every target)
    u32 *codes2 = nullptr;
    if (codes && c2_words) {
        codes2 = codes + c2_at;
        const int64_t n2w = codes2_words(len), nxw = gexc_words(len);
        (void)hipMemsetAsync(codes2 + n2w, 0, (size_t)K0_CODES2_PAD * 4, c->stream);
        (void)hipMemsetAsync(codes2 + n2w + K0_CODES2_PAD + nxw, 0, (size_t)K0_GEXC_PAD * 4, c->stream);
        hipLaunchKernelGGL(k0_encode2, dim3((unsigned)(((len + 63) / 64 + 255) / 256)), dim3(256), 0, c->stream, (const uint8_t *)d, len, codes2,
                           codes2 + n2w + K0_CODES2_PAD);
    }
    hipError_t se = hipStreamSynchronize(c->stream);
    if (se != hipSuccess) {
        if (codes) (void)hipFree(codes);
        return fail(c, PJB_ERR_HIP, "upload: %s", hipGetErrorString(se));
    }
    if (exotic && codes) {
        (void)hipFree(codes);
        codes = nullptr;
        codes2 = nullptr;
    }
    Contig &g = c->contigs[(size_t)tid];
    free_contig(g, &c->genome_pool);
    g.d = d;
    g.len = len;
    g.owned = owned;
    g.has_x = hx != 0;
    g.present = true;
    g.codes = codes;
    g.codes2 = codes2;
    g.d_cap = owned ? d_cap : 0;
    g.codes_cap = codes ? codes_cap : 0;
    return PJB_OK;
}

int pjb_upload_contig(pjb_ctx *c, int32_t tid, const uint8_t *bases, int64_t len) {
    if (!c) return PJB_ERR_ARG;
    if (tid < 0 || (size_t)tid >= c->contigs.size() || len < 0 || (len > 0 && !bases))
        return fail(c, PJB_ERR_ARG, "pjb_upload_contig: bad arguments (tid %d)", tid);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    size_t d_cap = 0;
    uint8_t *d = (uint8_t *)genome_take(c->genome_pool, (size_t)std::max<int64_t>(len, 16), d_cap);
    if (!d) return fail(c, PJB_ERR_NOMEM, "hipMalloc(genome %lld) failed", (long long)len);
    hipError_t e = hipSuccess;
    // through the page-locked staging buffers in 32 MiB pieces (a pageable hipMemcpy is several times slower)
    const size_t PIECE = (size_t)32 << 20;
    for (size_t off = 0; off < (size_t)len; off += PIECE) {
        const size_t n = std::min(PIECE, (size_t)len - off);
        const unsigned si = c->stage_next++ & 1u;
        if (c->stage_busy[si]) {
            (void)hipEventSynchronize(c->stage_ev[si]);
            c->stage_busy[si] = false;
        }
        if (c->stage_cap[si] < n) {
            if (c->stage[si]) (void)hipHostFree(c->stage[si]);
            c->stage[si] = nullptr;
            c->stage_cap[si] = 0;
            if (hipHostMalloc((void **)&c->stage[si], PIECE, hipHostMallocDefault) != hipSuccess) {
                (void)hipFree(d);
                return fail(c, PJB_ERR_NOMEM, "upload: cannot allocate page-locked staging memory");
            }
            c->stage_cap[si] = PIECE;
        }
        if (!c->stage_ev[si]) (void)hipEventCreateWithFlags(&c->stage_ev[si], hipEventDisableTiming);
        parallel_copy(c->stage[si], bases + off, n);
        e = hipMemcpyAsync(d + off, c->stage[si], n, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) {
            (void)hipFree(d);
            return fail(c, PJB_ERR_HIP, "hipMemcpy(genome): %s", hipGetErrorString(e));
        }
        (void)hipEventRecord(c->stage_ev[si], c->stream);
        c->stage_busy[si] = true;
    }
    int rc = upload_common(c, tid, d, len, true, true, d_cap);
    if (rc) (void)hipFree(d);
    return rc;
}

int pjb_upload_contig_fasta(pjb_ctx *c, int32_t tid, const uint8_t *raw, int64_t raw_bytes, int32_t line_blen, int32_t line_len, int64_t len,
                            int *well_formed) {
    if (!c) return PJB_ERR_ARG;
    if (tid < 0 || (size_t)tid >= c->contigs.size() || len < 0 || raw_bytes < 0 || (raw_bytes > 0 && !raw) || line_blen <= 0 ||
        line_len < line_blen || !well_formed)
        return fail(c, PJB_ERR_ARG, "pjb_upload_contig_fasta: bad arguments (tid %d)", tid);
    *well_formed = 0;
    if (line_len - line_blen > 64) return PJB_OK; // (line ends are a byte or two; the kernel's check of them is a loop per line)
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    int rc;
    if ((rc = ensure(c, c->b_fasta_raw, (size_t)std::max<int64_t>(raw_bytes, 16)))) return rc;
    if ((rc = ensure(c, c->b_hasx, sizeof(int)))) return rc;
    size_t d_cap = 0;
    uint8_t *d = (uint8_t *)genome_take(c->genome_pool, (size_t)std::max<int64_t>(len, 16), d_cap);
    if (!d) return fail(c, PJB_ERR_NOMEM, "hipMalloc(genome %lld) failed", (long long)len);
    struct Guard {
        uint8_t *d;
        ~Guard() {
            if (d) (void)hipFree(d);
        }
    } guard{d};
    // page-locked input (pjb_host_alloc): one DMA; otherwise through the staging buffers
    hipPointerAttribute_t at;
    const bool pinned = raw_bytes > 0 && hipPointerGetAttributes(&at, raw) == hipSuccess && at.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError();
    if (pinned) HIP_TRY(c, hipMemcpyAsync(c->b_fasta_raw.p, raw, (size_t)raw_bytes, hipMemcpyHostToDevice, c->stream));
    else if (raw_bytes > 0 && (rc = upload_staged(c, c->b_fasta_raw.p, raw, (size_t)raw_bytes))) return rc;
    int bad = 0;
    HIP_TRY(c, hipMemsetAsync(c->b_hasx.p, 0, sizeof(int), c->stream));
    if (len > 0) {
        const int64_t nthreads = (len + 15) / 16;
        hipLaunchKernelGGL(k0_fasta, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, c->stream, (const uint8_t *)c->b_fasta_raw.p, raw_bytes, len,
                           line_blen, line_len, d, (int *)c->b_hasx.p);
    }
    HIP_TRY(c, hipMemcpyAsync(&bad, c->b_hasx.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (bad) return PJB_OK; // (*well_formed stays 0: nothing was uploaded)
    rc = upload_common(c, tid, d, len, true, true, d_cap);
    if (rc) return rc;
    guard.d = nullptr;
    *well_formed = 1;
    return PJB_OK;
}

int pjb_upload_contig_device(pjb_ctx *c, int32_t tid, const uint8_t *d_bases_upper, int64_t len) {
    if (!c) return PJB_ERR_ARG;
    if (tid < 0 || (size_t)tid >= c->contigs.size() || len < 0 || (len > 0 && !d_bases_upper))
        return fail(c, PJB_ERR_ARG, "pjb_upload_contig_device: bad arguments (tid %d)", tid);
    return upload_common(c, tid, const_cast<uint8_t *>(d_bases_upper), len, false, false);
}

int pjb_release_contig(pjb_ctx *c, int32_t tid) {
    if (!c) return PJB_ERR_ARG;
    if (tid < 0 || (size_t)tid >= c->contigs.size()) return fail(c, PJB_ERR_ARG, "pjb_release_contig: bad tid %d", tid);
    Contig &g = c->contigs[(size_t)tid];
    (void)hipStreamSynchronize(c->stream);
    free_contig(g, &c->genome_pool);
    return PJB_OK;
}

static int add_batch(pjb_ctx *c, int32_t tid, const pjb_batch *b, bool device) {
    if (!c) return PJB_ERR_ARG;
    if (!b || b->n_reads < 0) return fail(c, PJB_ERR_ARG, "submit: bad batch");
    if (tid < 0 || (size_t)tid >= c->ref_len.size()) return fail(c, PJB_ERR_ARG, "submit: bad tid %d", tid);
    c->cur_tid = tid;
    OpenContig &oc = c->open[tid];
    if (b->n_reads == 0) return PJB_OK;
    if (!b->pos || !b->flag || !b->mapq || !b->xs || !b->l_qseq || !b->mtid || !b->mpos || !b->cig_off || !b->cigar ||
        !b->seq_off)
        return fail(c, PJB_ERR_ARG, "submit: null array in batch");
    if (c->extra && !b->name_hash) return fail(c, PJB_ERR_ARG, "submit: a PJB_FLAG_EXTRA context needs pjb_batch.name_hash");
    uint64_t total = 0;
    for (auto &x : oc.batches) total += (uint64_t)x.n;
    if (total + (uint64_t)b->n_reads >= 0xffffff00ull)
        return fail(c, PJB_ERR_ARG, "submit: more than 2^32 alignments on one target are not supported");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    DevBatch d;
    memset(&d, 0, sizeof d);
    d.n = b->n_reads;
    d.base = (uint32_t)total;
    if (device) {
        d.pos = b->pos; d.flag = b->flag; d.mapq = b->mapq; d.xs = b->xs; d.l_qseq = b->l_qseq; d.mtid = b->mtid;
        d.mpos = b->mpos; d.cig_off = b->cig_off; d.cigar = b->cigar; d.seq_off = b->seq_off; d.seq4 = b->seq4;
        d.name_hash = c->extra ? (const u64 *)b->name_hash : nullptr;
        if (c->cfg.abi_version >= 4 && b->seq2 && b->seq_exc) {
            if ((uintptr_t)b->seq2 & 3u) return fail(c, PJB_ERR_ARG, "submit: pjb_batch.seq2 must start on a 4-byte boundary");
            d.seq2 = (const uint32_t *)b->seq2;
            d.seq_exc = b->seq_exc;
        }
    } else {
        const size_t n = (size_t)b->n_reads;
        const size_t n_ops = b->cig_off[n], n_words = b->seq_off[n];
        const void *src[14] = {b->pos, b->flag, b->mapq, b->xs, b->l_qseq, b->mtid, b->mpos, b->cig_off, b->cigar, b->seq_off, b->seq4,
                               c->extra ? b->name_hash : nullptr, nullptr, nullptr};
        const bool two = c->cfg.abi_version >= 4 && b->seq2 && b->seq_exc;
        if (two) src[12] = b->seq2, src[13] = b->seq_exc;
        const size_t bytes[14] = {n * 4, n * 2, n, n, n * 4, n * 4, n * 4, (n + 1) * 4, n_ops * 4, (n + 1) * 4, n_words * 4,
                                  c->extra ? n * 8 : 0, two ? n_words * 2 : 0, two ? ((n + 31) / 32) * 4 : 0};
        // pack into a staging buffer, one DMA to a device slab region with the same packing
        size_t offs[14], total_b = 0;
        for (int k = 0; k < 14; k++) {
            offs[k] = total_b;
            total_b += (std::max<size_t>(bytes[k], 16) + 255) & ~(size_t)255;
        }
        const unsigned si = c->stage_next++ & 1u;
        if (c->stage_busy[si]) {
            HIP_TRY(c, hipEventSynchronize(c->stage_ev[si]));
            c->stage_busy[si] = false;
        }
        if (c->stage_cap[si] < total_b) {
            if (c->stage[si]) (void)hipHostFree(c->stage[si]);
            c->stage[si] = nullptr;
            c->stage_cap[si] = 0;
            const size_t want = total_b + total_b / 8;
            if (hipHostMalloc((void **)&c->stage[si], want, hipHostMallocDefault) != hipSuccess)
                return fail(c, PJB_ERR_NOMEM, "submit: cannot allocate %zu bytes of page-locked staging memory", want);
            c->stage_cap[si] = want;
        }
        if (!c->stage_ev[si]) HIP_TRY(c, hipEventCreateWithFlags(&c->stage_ev[si], hipEventDisableTiming));
        uint8_t *dev = (uint8_t *)slab_alloc(c, oc, total_b);
        if (!dev) return fail(c, PJB_ERR_NOMEM, "submit: out of device memory for a batch of %zu bytes", total_b);
        void *ptrs[14];
        for (int k = 0; k < 14; k++) {
            if (bytes[k] && src[k]) parallel_copy(c->stage[si] + offs[k], src[k], bytes[k]);
            ptrs[k] = dev + offs[k];
        }
        HIP_TRY(c, hipMemcpyAsync(dev, c->stage[si], total_b, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipEventRecord(c->stage_ev[si], c->stream));
        c->stage_busy[si] = true;
        d.pos = (const int32_t *)ptrs[0]; d.flag = (const uint16_t *)ptrs[1]; d.mapq = (const uint8_t *)ptrs[2];
        d.xs = (const uint8_t *)ptrs[3]; d.l_qseq = (const int32_t *)ptrs[4]; d.mtid = (const int32_t *)ptrs[5];
        d.mpos = (const int32_t *)ptrs[6]; d.cig_off = (const uint32_t *)ptrs[7]; d.cigar = (const uint32_t *)ptrs[8];
        d.seq_off = (const uint32_t *)ptrs[9]; d.seq4 = (const uint8_t *)ptrs[10];
        d.name_hash = c->extra ? (const u64 *)ptrs[11] : nullptr;
        if (two) d.seq2 = (const uint32_t *)ptrs[12], d.seq_exc = (const uint32_t *)ptrs[13];
    }
    oc.batches.push_back(d);
    if (!device) oc.on_main_stream = true;
    oc.last_known.push_back(device ? 0 : 1);
    oc.last_pos.push_back(device ? INT32_MIN : b->pos[b->n_reads - 1]);
    return PJB_OK;
}

int pjb_submit_batch(pjb_ctx *c, int32_t tid, const pjb_batch *b) { return add_batch(c, tid, b, false); }
int pjb_submit_batch_device(pjb_ctx *c, int32_t tid, const pjb_batch *b) { return add_batch(c, tid, b, true); }

// a contig without junctions still reports its counters through the row mirror
static void mirror_fold(pjb_ctx *c, const pjb_region_result &R, size_t new_rows) {
    c->mirror_rows += new_rows;
    c->mirror_acc[0] += (int64_t)R.spliced;
    c->mirror_acc[1] += (int64_t)R.unspliced;
    c->mirror_acc[2] += (int64_t)R.sum_len;
    c->mirror_acc[3] = std::min<int64_t>(c->mirror_acc[3], R.min_len);
    c->mirror_acc[4] = std::max<int64_t>(c->mirror_acc[4], R.max_len);
    int64_t *h = c->mirror_hdr;
    h[0] = (int64_t)c->mirror_rows;
    for (int k = 0; k < 5; k++) h[1 + k] = c->mirror_acc[k];
    h[6] = h[7] = 0;
}
static void mirror_reset(pjb_ctx *c) {
    c->mirror_rows = 0;
    c->mirror_acc[0] = c->mirror_acc[1] = c->mirror_acc[2] = 0;
    c->mirror_acc[3] = INT32_MAX;
    c->mirror_acc[4] = 0;
}
static int mirror_header_only(pjb_ctx *c, const pjb_region_result &R) {
    if (!c->mirror) return PJB_OK;
    aux_streams(c);
    mirror_fold(c, R, 0);
    HIP_TRY(c, hipMemcpyAsync(c->mirror, c->mirror_hdr, PJB_MIRROR_HEADER_BYTES, hipMemcpyHostToDevice, c->stream4));
    HIP_TRY(c, hipStreamSynchronize(c->stream4));
    return PJB_OK;
}

// --extra, per contig (calcExtraMetrics' per-target work, src/junction_builder.cc:293-312): the unspliced records'
// per-base depth, the junctions' flanking alignment counts, and the name codes phase 2 needs.  Runs after the
// contig's rows exist (b_rows, sidx, jid are still this contig's).
constexpr u32 X_ZCAP = 1u << 20;
static int extra_contig_dense(pjb_ctx *c, int32_t tid, std::vector<DevBatch> &batches, int64_t n_reads, u64 n_spliced, u32 P, u32 J,
                              const u32 *sidx, const u32 *jid_sorted, const u32 *pair_g, size_t row_base, bool codes_in_table) {
    hipStream_t st = c->stream;
    const int32_t L = c->ref_len[(size_t)tid];
    const size_t N = (size_t)n_reads;
    ExtraContig X;
    X.dense = true;
    X.codes_in_table = codes_in_table;
    X.tid = tid;
    X.len = L;
    X.row_base = row_base;
    X.n_rows = J;
    X.n_pairs = P;
    int rc;
    if ((rc = ensure(c, c->b_xtotal, 8))) return rc;
    if ((rc = ensure(c, c->x_pos, N * 4 + 16))) return rc;
    if ((rc = ensure(c, c->x_endx, N * 4 + 16))) return rc;
    if ((rc = ensure(c, c->x_q, N + 16))) return rc;
    if ((rc = ensure(c, c->x_prefq, (N + 1) * 4))) return rc;
    if ((rc = ensure(c, c->x_ce, ((size_t)L + 2) * 4))) return rc;
    if ((rc = ensure(c, c->x_zlist, (size_t)X_ZCAP * 4))) return rc;
    if ((rc = ensure(c, c->x_cnt, sizeof(ExtraCounters)))) return rc;
    struct Guard { // frees what this contig allocated unless it is handed over to the context
        ExtraContig *x;
        ~Guard() {
            if (!x) return;
            if (x->cover) (void)hipFree(x->cover);
            if (x->xr) (void)hipFree(x->xr);
            if (x->pair_code) (void)hipFree(x->pair_code);
            if (x->pair_row) (void)hipFree(x->pair_row);
            if (x->spl_codes) (void)hipFree(x->spl_codes);
        }
    } guard{&X};
    if (hipMalloc((void **)&X.cover, ((size_t)L + 2) * 4) != hipSuccess) return fail(c, PJB_ERR_NOMEM, "extra: depth array of target %d", tid);
    if (hipMalloc((void **)&X.spl_codes, std::max<size_t>((size_t)n_spliced, 1) * 8) != hipSuccess)
        return fail(c, PJB_ERR_NOMEM, "extra: name codes of target %d", tid);
    HIP_TRY(c, hipMemsetAsync(X.cover, 0, ((size_t)L + 2) * 4, st));
    HIP_TRY(c, hipMemsetAsync(c->x_ce.p, 0, ((size_t)L + 2) * 4, st));
    HIP_TRY(c, hipMemsetAsync((uint8_t *)c->x_q.p + N, 0, 1, st));
    ExtraCounters hc;
    memset(&hc, 0, sizeof hc);
    hc.hot_first = 0xffffffffu;
    HIP_TRY(c, hipMemcpyAsync(c->x_cnt.p, &hc, sizeof hc, hipMemcpyHostToDevice, st));
    ExtraCounters *d_cnt = (ExtraCounters *)c->x_cnt.p;
    int32_t *x_pos = (int32_t *)c->x_pos.p, *x_endx = (int32_t *)c->x_endx.p;
    uint8_t *x_q = (uint8_t *)c->x_q.p;
    u32 *prefq = (u32 *)c->x_prefq.p, *ce = (u32 *)c->x_ce.p;
    for (auto &b : batches)
        LAUNCH(c, "kx_classify", kx_classify, dim3((unsigned)((b.n + 255) / 256)), dim3(256), b, L, x_pos, x_endx, x_q, ce,
               (int32_t *)X.cover, (u32 *)c->x_zlist.p, X_ZCAP, d_cnt);
    {   // the spliced records' name codes, through the tile lists the contig's first kernels left in its slot
        CtlSlot &S = c->sl[c->cur_slot];
        u32 n_tiles = 0;
        for (auto &b : batches) n_tiles = std::max<u32>(n_tiles, b.tile_base + (u32)((b.n + K1_TILE - 1) / K1_TILE));
        if ((rc = ensure(c, c->x_tileoff, (size_t)n_tiles * 4 + 16))) return rc;
        LAUNCH(c, "kx_spliced_offsets", kx_spliced_offsets, dim3(1), dim3(1024), (const TileStats *)S.tile_stats.p, n_tiles, (u32 *)c->x_tileoff.p, d_cnt);
        for (auto &b : batches)
            LAUNCH(c, "kx_spliced_codes", kx_spliced_codes, dim3((unsigned)((b.n + K1_TILE - 1) / K1_TILE)), dim3(256), b, (const TileStats *)S.tile_stats.p,
                   (const u32 *)S.splidx.p, (const u32 *)c->x_tileoff.p, X.spl_codes);
    }
    if ((rc = run_scan(c, "kx_ends", ArrU32Fn{ce}, ExclusiveU32Sink{ce}, (u64)L + 2, (u64 *)c->b_xtotal.p))) return rc;
    if ((rc = run_scan(c, "kx_unspl", ArrU8Fn{x_q}, ExclusiveU32Sink{prefq}, (u64)N + 1, (u64 *)c->b_xtotal.p))) return rc;
    LAUNCH(c, "kx_cap_bound", kx_cap_bound, dim3((unsigned)((N + 255) / 256)), dim3(256), (const int32_t *)x_pos, (const uint8_t *)x_q,
           (const u32 *)prefq, (const u32 *)ce, (u32)N, L, (u32 *)nullptr, d_cnt);
    u32 n_unspl = 0;
    HIP_TRY(c, hipMemcpyAsync(&hc, d_cnt, sizeof hc, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(&n_unspl, prefq + N, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    if (hc.n_zero > X_ZCAP)
        return fail(c, PJB_ERR_ARG, "extra: target %d has %u mapped records without a reference span (limit %u)", tid, hc.n_zero, X_ZCAP);
    if (hc.max_buffered + 2 > PLP_MAXCNT) { // the pileup's record cap may bite: replay it over the hot span
        if ((rc = ensure(c, c->x_bound, N * 4 + 16))) return rc;
        if ((rc = ensure(c, c->x_de, ((size_t)L + 2) * 4))) return rc;
        if ((rc = ensure(c, c->x_dropped, N + 16))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->x_de.p, 0, ((size_t)L + 2) * 4, st));
        HIP_TRY(c, hipMemsetAsync(c->x_dropped.p, 0, N + 16, st));
        LAUNCH(c, "kx_cap_bound", kx_cap_bound, dim3((unsigned)((N + 255) / 256)), dim3(256), (const int32_t *)x_pos,
               (const uint8_t *)x_q, (const u32 *)prefq, (const u32 *)ce, (u32)N, L, (u32 *)c->x_bound.p, d_cnt);
        LAUNCH(c, "kx_cap_replay", kx_cap_replay, dim3(1), dim3(64), (const int32_t *)x_pos, (const int32_t *)x_endx, (const uint8_t *)x_q,
               (const u32 *)c->x_bound.p, (u32)N, L, (u32 *)c->x_de.p, (uint8_t *)c->x_dropped.p, d_cnt);
        for (auto &b : batches)
            LAUNCH(c, "kx_undo_dropped", kx_undo_dropped, dim3((unsigned)((b.n + 255) / 256)), dim3(256), b, L,
                   (const uint8_t *)c->x_dropped.p, (int32_t *)X.cover);
    }
    if ((rc = run_scan(c, "kx_depth", ArrI32Fn{(const int32_t *)X.cover}, InclusiveU32Sink{X.cover}, (u64)L + 1, (u64 *)c->b_xtotal.p)))
        return rc;
    X.has_unspliced = n_unspl > 0;
    X.n_spl = hc.n_spliced;
    if (J > 0) {
        if (hipMalloc((void **)&X.xr, (size_t)J * sizeof(ExtraRow)) != hipSuccess) return fail(c, PJB_ERR_NOMEM, "extra: rows of target %d", tid);
        HIP_TRY(c, hipMemsetAsync(X.xr, 0, (size_t)J * sizeof(ExtraRow), st));
        LAUNCH(c, "kx_flank", kx_flank, dim3((J + 255) / 256), dim3(256), (const pjb_junction_row *)c->sl[c->cur_slot].rows.p, J, (const int32_t *)x_pos,
               (u32)N, (const u32 *)prefq, (const u32 *)ce, L, (const u32 *)c->x_zlist.p, (const ExtraCounters *)d_cnt, X_ZCAP, X.xr);
        if (hipMalloc((void **)&X.pair_code, (size_t)P * 8) != hipSuccess || hipMalloc((void **)&X.pair_row, (size_t)P * 4) != hipSuccess)
            return fail(c, PJB_ERR_NOMEM, "extra: pair codes of target %d", tid);
        LAUNCH(c, "kx_pair_codes", kx_pair_codes, dim3((P + 255) / 256), dim3(256), sidx, jid_sorted, pair_g,
               (const DevBatch *)c->sl[c->cur_slot].batches.p, (int)batches.size(), P, (u32)row_base, X.pair_code, X.pair_row);
    }
    HIP_TRY(c, hipStreamSynchronize(st));
    if (c->ktime) ev_collect(c, MISC_POOL);
    c->xc.push_back(X);
    guard.x = nullptr;
    return PJB_OK;
}

#define XTRACE(what)                                                                                                              \
    do {                                                                                                                          \
        if (xtrace) {                                                                                                             \
            (void)hipStreamSynchronize(st);                                                                                       \
            const auto now_ = std::chrono::steady_clock::now();                                                                   \
            fprintf(stderr, "[xtrace] %-28s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now_ - xt0).count());     \
            xt0 = now_;                                                                                                           \
        }                                                                                                                         \
    } while (0)
// room in the name table for `add` more codes (load <= 1/2): a larger table takes over the old one's names
static int name_table_reserve(pjb_ctx *c, size_t add) {
    hipStream_t st = c->stream;
    const size_t need = (c->x_tab_n + add) + (c->x_tab_n + add) / 2 + 1; // load <= 2/3
    if (c->x_tab_n == 0 && c->x_tab_slots >= need) { // first codes of a file: wipe
        if (add) HIP_TRY(c, hipMemsetAsync(c->x_tab.p, 0xff, c->x_tab_slots * sizeof(NameSlot), st));
        return PJB_OK;
    }
    if (c->x_tab_slots >= need) return PJB_OK;
    if (need > 0xfffffff0ull) return fail(c, PJB_ERR_ARG, "extra: more than 2^31 spliced records");
    const size_t slots = std::min<size_t>(std::max<size_t>(2 * need + 16, 1024), 0xfffffff0ull); // (twice what is needed now: a file's targets arrive one by one)
    Buf nb;
    int rc = ensure(c, nb, slots * sizeof(NameSlot));
    if (rc) return rc;
    HIP_TRY(c, hipMemsetAsync(nb.p, 0xff, slots * sizeof(NameSlot), st));
    if (c->x_tab_n)
        LAUNCH(c, "kx_name_rehash", kx_name_rehash, dim3((unsigned)((c->x_tab_slots + 255) / 256)), dim3(256), (const NameSlot *)c->x_tab.p,
               (u32)c->x_tab_slots, (NameSlot *)nb.p, (u32)slots);
    if (c->x_tab.p) {
        HIP_TRY(c, hipStreamSynchronize(st));
        release(c->x_tab);
    }
    c->x_tab = nb;
    c->x_tab_slots = slots;
    return PJB_OK;
}
static int name_table_insert(pjb_ctx *c, const u64 *codes, u32 n) {
    if (!n) return PJB_OK;
    int rc = name_table_reserve(c, n);
    if (rc) return rc;
    LAUNCH(c, "kx_name_insert", kx_name_insert4, dim3((n + 1023) / 1024), dim3(256), codes, n, (NameSlot *)c->x_tab.p, (u32)c->x_tab_slots);
    c->x_tab_n += n;
    return PJB_OK;
}

// The same per-target work without an array of the target's length (pjb_extra.hip.h, "the sparse path"), in two parts.
// extra_pre needs the records only: queued on the service stream when the target's chain is queued, it runs beside the
// chains.  extra_contig needs the chain's rows and sorted pairs: queued when the chain is collected, beside the chains of
// the targets queued after this one; one wait at its end.  A target where the pileup's cap may bite goes through
// extra_contig_dense instead.
static int extra_pre(pjb_ctx *c, Flight &f) {
    if (f.x_pre || c->extra_dense_only || f.empty) return PJB_OK;
    hipStream_t st = c->stream;
    CtlSlot &S = c->sl[f.slot];
    const size_t N = (size_t)f.n_reads;
    int rc;
    f.x_gap_cap = (u32)std::min<size_t>(N / 16 + 1024, 0x7fffffffu);
    f.x_spos = (int32_t *)xarena_alloc(c, N * 4 + 16); // (compacted: the records with a span)
    f.x_send = (int32_t *)xarena_alloc(c, N * 4 + 16);
    f.x_gapoff = (u32 *)xarena_alloc(c, (N / 256 + 2) * 4);
    f.x_gaps = (Gap *)xarena_alloc(c, (size_t)f.x_gap_cap * sizeof(Gap));
    if (!f.x_spos || !f.x_send || !f.x_gapoff || !f.x_gaps)
        return fail(c, PJB_ERR_NOMEM, "extra: no device memory for what target %d keeps (%zu records)", f.tid, N);
    if ((rc = ensure(c, S.x_q, N + 16)) || (rc = ensure(c, S.x_spos, N * 4 + 16)) || (rc = ensure(c, S.x_send, N * 4 + 16)) ||
        (rc = ensure(c, S.x_gapoff, (N / 256 + 2) * 4)) || (rc = ensure(c, S.x_zlist, (size_t)X_ZCAP * 4)) ||
        (rc = ensure(c, S.x_scnt, sizeof(SparseCounters) + sizeof(ExtraCounters))))
        return rc;
    uint8_t *q = (uint8_t *)S.x_q.p;
    SparseCounters *d_cnt = (SparseCounters *)S.x_scnt.p;
    if (f.x_k1) HIP_TRY(c, hipStreamWaitEvent(st, S.ev_xk1, 0)); // (the chain's k1_count classified the records)
    else {
        HIP_TRY(c, hipMemsetAsync(q + N, 0, 1, st));
        HIP_TRY(c, hipMemsetAsync(d_cnt, 0, sizeof(SparseCounters) + sizeof(ExtraCounters), st));
        for (auto &b : f.batches)
            LAUNCH(c, "kx_classify_sparse", kx_classify_sparse, dim3((unsigned)((b.n + 255) / 256)), dim3(256), b, (int32_t *)S.x_spos.p, (int32_t *)S.x_send.p,
                   q, (u32 *)S.x_zlist.p, X_ZCAP, d_cnt);
    }
    if ((rc = run_scan(c, "kx_spans", SparseFn{q},
                       SparseSink{f.x_spos, f.x_send, (u32 *)S.x_gapoff.p, f.x_gapoff, (const int32_t *)S.x_spos.p, (const int32_t *)S.x_send.p, q}, (u64)N + 1,
                       &d_cnt->total)))
        return rc;
    for (auto &b : f.batches)
        if (b.n > 0)
        {
            const u32 nblk = (u32)((((u64)b.base + (u64)b.n + 255) >> 8) - (b.base >> 8));
            LAUNCH(c, "kx_gaps", kx_gaps, dim3(std::min<u32>(nblk, 2048)), dim3(256), b, (const uint8_t *)q, (u32)N, (const u32 *)S.x_gapoff.p, f.x_gaps,
                   f.x_gap_cap, d_cnt, nblk);
        }
    if (N >= PLP_MAXCNT)
        LAUNCH(c, "kx_cap_check", kx_cap_check, dim3((unsigned)((N + 255) / 256)), dim3(256), (const int32_t *)f.x_spos, d_cnt);
    f.x_pre = true;
    return PJB_OK;
}

static int extra_contig(pjb_ctx *c, Flight &f, int32_t tid, u64 n_spliced, u32 P, u32 J, size_t row_base) {
    std::vector<DevBatch> &batches = f.batches;
    if (c->extra_dense_only) return extra_contig_dense(c, tid, batches, f.n_reads, n_spliced, P, J, f.sidx, f.jid_sorted, f.pr.g, row_base, false);
    int rc;
    if ((rc = extra_pre(c, f))) return rc;
    hipStream_t st = c->stream;
    const int32_t L = c->ref_len[(size_t)tid];
    CtlSlot &S = c->sl[f.slot];
    ExtraContig X;
    X.tid = tid;
    X.len = L;
    X.row_base = row_base;
    X.n_rows = J;
    X.n_pairs = P;
    X.xr = J ? (ExtraRow *)xarena_alloc(c, (size_t)J * sizeof(ExtraRow)) : nullptr;
    X.pair_code = J ? (u64 *)xarena_alloc(c, (size_t)P * 8 + 16) : nullptr;
    X.pair_row = J ? (u32 *)xarena_alloc(c, (size_t)P * 4 + 16) : nullptr;
    if (J && (!X.xr || !X.pair_code || !X.pair_row)) return fail(c, PJB_ERR_NOMEM, "extra: no device memory for the pairs of target %d", tid);
    if ((rc = ensure(c, S.x_codes, std::max<size_t>((size_t)n_spliced, 1) * 8))) return rc;
    const bool xtrace = getenv("PJB_XTRACE") != nullptr;
    auto xt0 = std::chrono::steady_clock::now();
    XTRACE("post: pre-part done");
    SparseCounters *d_cnt = (SparseCounters *)S.x_scnt.p;
    ExtraCounters *d_xcnt = (ExtraCounters *)(d_cnt + 1);
    {   // the spliced records' name codes, through the tile lists the target's first kernels left in its slot -> the name table
        u32 n_tiles = 0;
        for (auto &b : batches) n_tiles = std::max<u32>(n_tiles, b.tile_base + (u32)((b.n + K1_TILE - 1) / K1_TILE));
        if ((rc = ensure(c, c->x_tileoff, (size_t)n_tiles * 4 + 16))) return rc;
        LAUNCH(c, "kx_spliced_offsets", kx_spliced_offsets, dim3(1), dim3(1024), (const TileStats *)S.tile_stats.p, n_tiles, (u32 *)c->x_tileoff.p, d_xcnt);
        for (auto &b : batches)
            LAUNCH(c, "kx_spliced_codes", kx_spliced_codes, dim3((unsigned)((b.n + K1_TILE - 1) / K1_TILE)), dim3(256), b, (const TileStats *)S.tile_stats.p,
                   (const u32 *)S.splidx.p, (const u32 *)c->x_tileoff.p, (u64 *)S.x_codes.p);
        XTRACE("post: codes");
        if ((rc = name_table_insert(c, (const u64 *)S.x_codes.p, (u32)n_spliced))) return rc;
        X.codes_in_table = true;
        XTRACE("post: insert");
    }
    if (J > 0) {
        HIP_TRY(c, hipMemsetAsync(X.xr, 0, (size_t)J * sizeof(ExtraRow), st));
        LAUNCH(c, "kx_flank_sparse", kx_flank_sparse, dim3((J + 255) / 256), dim3(256), (const pjb_junction_row *)S.rows.p, J,
               (const int32_t *)f.x_spos, (const int32_t *)f.x_send, L, (const u32 *)S.x_zlist.p, (const SparseCounters *)d_cnt, X_ZCAP, X.xr);
        LAUNCH(c, "kx_pair_codes", kx_pair_codes, dim3((P + 255) / 256), dim3(256), f.sidx, f.jid_sorted, f.pr.g, (const DevBatch *)S.batches.p,
               (int)batches.size(), P, (u32)row_base, X.pair_code, X.pair_row);
    }
    // one wait: the counters decide whether the sparse answer stands
    SparseCounters &hc = *(SparseCounters *)(S.pub + PUB_XCNT_AT);
    ExtraCounters &hx = *(ExtraCounters *)(S.pub + PUB_XCNT_AT + sizeof(SparseCounters));
    XTRACE("post: flank + pair codes");
    HIP_TRY(c, hipMemcpyAsync(&hc, d_cnt, sizeof(SparseCounters) + sizeof(ExtraCounters), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    XTRACE("post: counters");
    if (c->ktime) ev_collect(c, MISC_POOL);
    if (hc.n_zero > X_ZCAP)
        return fail(c, PJB_ERR_ARG, "extra: target %d has %u mapped records without a reference span (limit %u)", tid, hc.n_zero, X_ZCAP);
    if (hx.n_spliced != (u32)n_spliced)
        return fail(c, PJB_ERR_STATE, "extra: target %d: %u spliced records in the tile lists, the chain counted %llu", tid, hx.n_spliced, (unsigned long long)n_spliced);
    if (hc.need_dense) // the pileup's cap may bite (or the gap list is too small): the depth vector, as in round 2
        return extra_contig_dense(c, tid, batches, f.n_reads, n_spliced, P, J, f.sidx, f.jid_sorted, f.pr.g, row_base, true);
    X.has_unspliced = (u32)hc.total > 0;
    X.n_spl = hx.n_spliced;
    X.sparse = SparseDepth{f.x_spos, f.x_send, f.x_gaps, f.x_gapoff, (u32)hc.total, (u32)(hc.total >> 32), hc.max_span, hc.max_gap};
    c->xc.push_back(X);
    return PJB_OK;
}

// The device work of one contig, queued in one go.  The host does not learn a single count while the kernels run:
// buffers and grids are sized from LIMITS (pair_limit, junc_limit, the key format kf), the kernels read the actual
// counts from the control block in device memory (ContigStats) and stand still when a limit is exceeded.  Nothing
// here waits for the device: the last kernels (rows stream) write rows and control block into page-locked host memory
// and pjb_finish_contig_end waits for their event -- by which time the next contig may be queued behind this one.
static void wait_flight(pjb_ctx *c, Flight &f);

// the host row table is complete up to rows_n
static int rows_sync(pjb_ctx *c) {
    if (!c->rows_copy_pending) return PJB_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream4));
    c->rows_copy_pending = false;
    return PJB_OK;
}

// rows of the contigs collected so far plus the most the queued ones can add
static size_t rows_upper_bound(const pjb_ctx *c) {
    size_t n = c->rows_n;
    for (int k = 0; k < c->n_fl; k++)
        if (c->fl[k].queued) n += c->fl[k].lim.junc_limit;
    return n;
}

constexpr unsigned K6_BLOCKS = 128;
static int queue_contig(pjb_ctx *c, Flight &f) {
    std::vector<DevBatch> &batches = f.batches;
    CtlSlot &S = c->sl[f.slot];
    // the contig's chain runs on its slot's streams, beside the chain of the contig in the other slot; everything the
    // LAUNCH macro and run_scan do follows c->stream / c->scan_tiles until this function returns
    const hipStream_t service = c->stream, st = c->side_stream ? S.main : c->stream; // ("overlap" off: everything on one stream)
    struct ChainScope {
        pjb_ctx *c;
        hipStream_t service;
        ~ChainScope() {
            c->stream = service;
            c->scan_tiles = &c->b_scan_tiles;
        }
    } chain_scope{c, service};
    c->stream = st;
    c->scan_tiles = &S.scan_tiles;
    const u32 n_tiles = f.n_tiles;
    const ContigLimits &lim = f.lim;
    const int n_members = (int)f.tids.size();
    const bool group = n_members > 1;
    // the sequence the chain works on: the target itself, or the group's virtual sequence (every member at its offset)
    const int32_t ref_len = group ? (int32_t)f.vlen : c->ref_len[(size_t)f.tid];
    GroupTab GT;
    memset(&GT, 0, sizeof GT);
    GT.n = n_members;
    bool all_codes = true, any_x = false;
    for (int m = 0; m < n_members; m++) {
        const Contig &g = c->contigs[(size_t)f.tids[(size_t)m]];
        GT.voff[m] = f.voff[(size_t)m];
        GT.len[m] = (int32_t)g.len;
        GT.tid[m] = f.tids[(size_t)m];
        GT.d[m] = g.d;
        GT.codes[m] = g.codes;
        GT.codes2[m] = g.codes2;
        all_codes = all_codes && g.codes != nullptr;
        any_x = any_x || g.has_x;
    }
    const KeyFmt kf = lim.kf;
    const u32 PL = lim.pair_limit, JL = lim.junc_limit;
    const u32 SL = lim.sort_limit && lim.sort_limit < JL ? lim.sort_limit : JL; // (ids the sort's digits cover)
    int rc;
    // the place of this contig's rows: known here if nothing is queued ahead of it, else it follows on the device
    bool ahead = false;
    for (int k = 0; k < c->n_fl; k++) ahead |= c->fl[k].queued && &c->fl[k] != &f;
    const int64_t row_base = ahead ? -1 : (int64_t)c->rows_n, mirror_base = ahead ? -1 : (int64_t)c->mirror_rows;
    struct PoolScope { // LAUNCH brackets of this chain belong to the contig's slot
        pjb_ctx *c;
        ~PoolScope() { c->cur_pool = MISC_POOL; }
    } pool_scope{c};
    c->cur_pool = f.slot;
    c->cur_slot = f.slot;
    ev_drop(c, f.slot);
    if ((rc = ensure(c, S.batches, batches.size() * sizeof(DevBatch)))) return rc;
    if (batches.size() + 2 > S.batches_pinned_cap) { // (+ 2 descriptors' worth of room for a group's tile ranges: 33 words)
        if (S.batches_pinned) (void)hipHostFree(S.batches_pinned);
        S.batches_pinned = nullptr;
        S.batches_pinned_cap = 0;
        const size_t cap = std::max<size_t>(batches.size() * 2 + 2, 16);
        HIP_TRY(c, hipHostMalloc((void **)&S.batches_pinned, cap * sizeof(DevBatch), hipHostMallocDefault));
        S.batches_pinned_cap = cap;
    }
    memcpy(S.batches_pinned, batches.data(), batches.size() * sizeof(DevBatch));
    if ((rc = ensure(c, S.cstats, sizeof(ContigStats)))) return rc;
    if (!S.scan_parts.p) {
        if ((rc = ensure(c, S.scan_parts, sizeof(ScanPart) * K1S_BLOCKS))) return rc;
        HIP_TRY(c, hipMemset(S.scan_parts.p, 0, sizeof(ScanPart) * K1S_BLOCKS)); // (once per slot)
        HIP_TRY(c, hipStreamSynchronize(nullptr));
    }
    if ((rc = ensure(c, S.err, 8))) return rc;
    if (!c->b_cursor.p) {
        if ((rc = ensure(c, c->b_cursor, sizeof(RowCursor)))) return rc;
        HIP_TRY(c, hipMemset(c->b_cursor.p, 0, sizeof(RowCursor))); // (blocks_done)
        HIP_TRY(c, hipStreamSynchronize(nullptr));
    }
    if ((rc = ensure(c, S.tile_cnt, (size_t)n_tiles * 4))) return rc;
    if ((rc = ensure(c, S.tile_stats, (size_t)n_tiles * sizeof(TileStats)))) return rc;
    if ((rc = ensure(c, S.tile_soff, ((size_t)n_tiles + 1) * 4))) return rc;
    if ((rc = ensure(c, S.chunk_tile, ((size_t)n_tiles * (K1_TILE / 256) + 4) * 4))) return rc;
    if ((rc = ensure(c, S.splidx, (size_t)n_tiles * K1_TILE * 4))) return rc;
    if ((rc = ensure(c, S.splpoff, (size_t)n_tiles * K1_TILE * 4))) return rc;
    if ((rc = ensure(c, S.splrec, (size_t)n_tiles * K1_TILE * 16))) return rc;
    if (!S.members.p) {
        if ((rc = ensure(c, S.members, GROUP_MAX * sizeof(MemberStats) + GROUP_MAX * 4 + (GROUP_MAX + 1) * 4))) return rc;
        HIP_TRY(c, hipMemset(S.members.p, 0, S.members.cap)); // (member_junc: k6_rows_out (publish_chain) leaves it zeroed for the next chain)
        HIP_TRY(c, hipStreamSynchronize(nullptr));
    }
    MemberStats *d_members = (MemberStats *)S.members.p;
    u32 *d_member_junc = (u32 *)(d_members + GROUP_MAX);
    u32 *d_tile_lo = d_member_junc + GROUP_MAX;
    if ((rc = ensure(c, S.total, 8))) return rc;
    // ---- pair-sized buffers (one sort tile of slack: rs_scatter loads whole tiles unguarded)
    if ((rc = ensure(c, S.okey, ((size_t)PL + RS_TILE) * 8))) return rc; // the pairs' keys as emitted (BAM order): kept, the sort works on copies
    if ((rc = ensure(c, S.key[0], ((size_t)PL + RS_TILE) * 8))) return rc;
    if ((rc = ensure(c, S.key[1], ((size_t)PL + RS_TILE) * 8))) return rc;
    if ((rc = ensure(c, S.idx[0], ((size_t)PL + RS_TILE) * 4))) return rc;
    if ((rc = ensure(c, S.idx[1], ((size_t)PL + RS_TILE) * 4))) return rc;
    if ((rc = ensure(c, S.rec, ((size_t)PL + 1) * sizeof(PairRec)))) return rc;
    if ((rc = ensure(c, S.jid, (size_t)PL * 4 + 16))) return rc;
    if ((rc = ensure(c, S.jidbam, ((size_t)PL + RS_TILE) * 4))) return rc; // (the sort's first pass loads whole tiles)
    if (c->extra && (rc = ensure(c, S.g, (size_t)PL * 4 + 16))) return rc;
    if ((rc = ensure(c, S.seg, ((size_t)PL + 1) * 4))) return rc;
    if ((rc = ensure(c, S.runfirst, ((size_t)PL + 1) * 4))) return rc;
    if ((rc = ensure(c, S.runstart, ((size_t)PL + 1) * 4))) return rc;
    if ((rc = ensure(c, S.ent, (size_t)PL * 8 + 16))) return rc;
    const u32 pair_blocks = std::max<u32>(1, (PL + 255) / 256);
    // entries per sub-list (list_cap_forced: the test hook pjb_set_option("list_cap", n) -- a first attempt with a room that overflows)
    const u32 gen_cap = lim.list_cap ? lim.list_cap : (c->list_cap_forced ? c->list_cap_forced : gen_list_cap(PL));
    const u32 pack_nn = (u64)f.n_reads < (1ull << 28) ? 1u : 0u; // (EmitLists::pack_nn)
    if ((rc = ensure(c, S.genlist, (size_t)gen_cap * GEN_SHARDS * 8 * 3))) return rc; // (three lists: EmitLists)
    if ((rc = ensure(c, S.gencount, GEN_SHARDS * GEN_CNT_STRIDE * 4))) return rc; // a line per sub-list: reads, pairs
    // ---- junction-sized buffers
    const u32 slots_lim = JL + (PL + 63) / 64 + 1;
    if ((rc = ensure(c, S.frag, (size_t)slots_lim * F_WORDS * 4))) return rc;
    if ((rc = ensure(c, S.fragj, (size_t)slots_lim * 4))) return rc;
    if ((rc = ensure(c, S.acc, (size_t)JL * F_WORDS * 4 + 16))) return rc;
    if ((rc = ensure(c, S.ancl, (size_t)JL * 4 + 16))) return rc;
    if ((rc = ensure(c, S.ancr, (size_t)JL * 4 + 16))) return rc;
    if ((rc = ensure(c, S.jkey, (size_t)JL * 8 + 16))) return rc;
    if ((rc = ensure(c, S.rows, (size_t)JL * sizeof(pjb_junction_row) + 16))) return rc;
    // The row table lives twice, both grow-only: in HBM, where the rows stream appends each contig's rows (k6_rows_out),
    // and in page-locked host memory, filled by a DMA per contig once its row count is known (pjb_finish_contig_end).
    // (The kernel used to write the host table itself: 2 MB of PCIe stores per contig that slowed whatever ran beside
    // them -- the next contig's k1_count by 40 %.)
    size_t old = rows_upper_bound(c);
    if (old + JL > c->rows_cap) {
        for (int k = 0; k < c->n_fl; k++) // (the tables move: nothing may be writing to them)
            if (c->fl[k].queued && &c->fl[k] != &f) wait_flight(c, c->fl[k]);
        if ((rc = rows_sync(c))) return rc;
        old = std::min(old, c->rows_cap);
        const size_t ncap = std::max<size_t>((old + JL) * 3 / 2, 1024);
        pjb_junction_row *np = nullptr, *nd = nullptr;
        hipError_t e = hipHostMalloc((void **)&np, ncap * sizeof(pjb_junction_row), hipHostMallocPortable);
        if (e != hipSuccess) return fail(c, PJB_ERR_NOMEM, "hipHostMalloc(rows): %s", hipGetErrorString(e));
        e = hipMalloc((void **)&nd, ncap * sizeof(pjb_junction_row));
        if (e != hipSuccess) {
            (void)hipHostFree(np);
            return fail(c, PJB_ERR_NOMEM, "hipMalloc(row table): %s", hipGetErrorString(e));
        }
        if (old) {
            memcpy(np, c->rows_pinned, old * sizeof(pjb_junction_row));
            (void)hipMemcpy(nd, c->rows_table, old * sizeof(pjb_junction_row), hipMemcpyDeviceToDevice);
        }
        if (c->rows_pinned) (void)hipHostFree(c->rows_pinned);
        if (c->rows_table) (void)hipFree(c->rows_table);
        c->rows_pinned = np;
        c->rows_table = nd;
        c->rows_cap = ncap;
    }
    if (!S.pub) {
        HIP_TRY(c, hipHostMalloc((void **)&S.pub, PUB_BYTES, hipHostMallocMapped | hipHostMallocPortable));
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, S.pub, 0) != hipSuccess || !dp) return fail(c, PJB_ERR_HIP, "hipHostGetDevicePointer(control block) failed");
        S.pub_dev = (uint8_t *)dp;
    }

    // (records still being produced on the service stream -- host copies, BAM ingest -- come first)
    const hipStream_t front = st;
    {
        bool on_service = false;
        for (int32_t t : f.tids) {
            auto oit = c->open.find(t);
            on_service = on_service || (oit != c->open.end() && oit->second.on_main_stream);
        }
        if (st != service && on_service) {
            HIP_TRY(c, hipEventRecord(c->ev_front, service));
            HIP_TRY(c, hipStreamWaitEvent(st, c->ev_front, 0));
        }
    }
    HIP_TRY(c, hipMemcpyAsync(S.batches.p, S.batches_pinned, batches.size() * sizeof(DevBatch), hipMemcpyHostToDevice, front));
    if (group) { // (the tile ranges of the members; lives behind the batch descriptors in the page-locked staging block)
        u32 *h_lo = (u32 *)(S.batches_pinned + batches.size());
        for (int m = 0; m <= n_members; m++) h_lo[m] = f.tile_lo[(size_t)m];
        HIP_TRY(c, hipMemcpyAsync(d_tile_lo, h_lo, (size_t)(n_members + 1) * 4, hipMemcpyHostToDevice, front));
    }
    if (!S.at_rest) {
        HIP_TRY(c, hipMemsetAsync(S.err.p, 0xff, 8, front));
        HIP_TRY(c, hipMemsetAsync(S.gencount.p, 0, GEN_SHARDS * GEN_CNT_STRIDE * 4, front));
        HIP_TRY(c, hipMemsetAsync(d_member_junc, 0, GROUP_MAX * 4, front));
    }
    S.at_rest = false; // until k6_rows_out (publish_chain) is queued
    u64 *d_err = (u64 *)S.err.p;
    ContigStats *d_cs = (ContigStats *)S.cstats.p;
    const u32 *d_P = &d_cs->P, *d_J = &d_cs->J, *d_slots = &d_cs->n_slots;
    // stage boundaries are timed only under full instrumentation: an event between two kernels costs a ~6 us bubble
    const bool stage_events = c->ktime && c->ktime_only.empty();
#define STAGE_EVENT(k)                                            \
    do {                                                          \
        if (stage_events) HIP_TRY(c, hipEventRecord(S.ev[k], st));  \
    } while (0)
    HIP_TRY(c, hipEventRecord(S.ev[0], front));
    // K2d's bitmap and end slots (all-clear at rest): k1_emit sets the bits
    const size_t n_words = ((size_t)std::max(ref_len, 1) + 63) / 64;
    if (lim.dense) {
        const void *was[2] = {S.bitmap.p, S.ends.p};
        if ((rc = ensure(c, S.bitmap, n_words * 8 + 16))) return rc;
        if ((rc = ensure(c, S.wrank, n_words * 4 + 16))) return rc;
        const void *was_pc = S.pagecnt.p;
        if ((rc = ensure(c, S.pagecnt, ((n_words >> KD_PAGE_SHIFT) + 1) * 4 + 16))) return rc;
        if ((rc = ensure(c, S.pagerank, ((n_words >> KD_PAGE_SHIFT) + 1) * 4 + 16))) return rc;
        if ((rc = ensure(c, S.ends, (size_t)JL * DENSE_ENDS * 4 + 32))) return rc;
        if ((rc = ensure(c, S.firstid, (size_t)JL * 4 + 16))) return rc;
        if (!S.dense_at_rest || was[0] != S.bitmap.p || was[1] != S.ends.p || was_pc != S.pagecnt.p) { // (first use, new memory, or a chain that broke off)
            HIP_TRY(c, hipMemsetAsync(S.bitmap.p, 0, S.bitmap.cap, front));
            HIP_TRY(c, hipMemsetAsync(S.pagecnt.p, 0, S.pagecnt.cap, front));
            HIP_TRY(c, hipMemsetAsync(S.ends.p, 0xff, S.ends.cap, front));
        }
        S.dense_at_rest = false; // until kd_reset is queued
    }
    Pairs pr;
    pr.key = (u64 *)S.okey.p;
    pr.rec = (PairRec *)S.rec.p;
    pr.g = c->extra ? (u32 *)S.g.p : (u32 *)nullptr;
    f.pr = pr;
    u32 *d_gen_cnt = (u32 *)S.gencount.p;
    const bool fast_codes = all_codes && !any_x; // (else: no read is "simple", every pair takes k4b_generic's byte-wise walks)
    {
        // ---- K1a: count (a group's members: a tile whose alignments leave the member's own sequence is flagged); with
        // PJB_FLAG_EXTRA the first time also what the records span (a chain that is queued again leaves that alone: the
        // service stream may be reading it)
        const bool xk1 = c->extra && !c->extra_dense_only && !group && !f.x_k1 && !f.x_pre;
        XOut xo = {nullptr, nullptr, nullptr, nullptr, 0, nullptr};
        if (xk1) {
            const size_t N = (size_t)f.n_reads;
            if ((rc = ensure(c, S.x_q, N + 16)) || (rc = ensure(c, S.x_spos, N * 4 + 16)) || (rc = ensure(c, S.x_send, N * 4 + 16)) ||
                (rc = ensure(c, S.x_zlist, (size_t)X_ZCAP * 4)) || (rc = ensure(c, S.x_scnt, sizeof(SparseCounters) + sizeof(ExtraCounters))))
                return rc;
            HIP_TRY(c, hipMemsetAsync((uint8_t *)S.x_q.p + N, 0, 1, c->stream));
            HIP_TRY(c, hipMemsetAsync(S.x_scnt.p, 0, sizeof(SparseCounters) + sizeof(ExtraCounters), c->stream));
            xo = XOut{(int32_t *)S.x_spos.p, (int32_t *)S.x_send.p, (uint8_t *)S.x_q.p, (u32 *)S.x_zlist.p, X_ZCAP, (SparseCounters *)S.x_scnt.p};
        }
        // K1 of one chain fills the chip: the chains' K1 stages follow each other (this chain's waits for the last queued chain's),
        // and what comes behind a chain's K1 -- many small kernels -- runs beside the NEXT chain's K1 instead of beside its own twin
        if (c->k1_serial && c->last_k1_ev) HIP_TRY(c, hipStreamWaitEvent(st, c->last_k1_ev, 0));
        // (one launch over the chain's tiles: a block finds its batch from the tile index)
        if (xk1)
            LAUNCH(c, "k1_count", k1_count<true>, dim3(n_tiles), dim3(K1C_T), (const DevBatch *)S.batches.p, (int)batches.size(), (u32 *)S.tile_cnt.p, (TileStats *)S.tile_stats.p,
                   (u32 *)S.splidx.p, (u32 *)S.splpoff.p, (uint4 *)S.splrec.p, d_err, GT, 0, xo);
        else
            LAUNCH(c, "k1_count", k1_count<false>, dim3(n_tiles), dim3(K1C_T), (const DevBatch *)S.batches.p, (int)batches.size(), (u32 *)S.tile_cnt.p, (TileStats *)S.tile_stats.p,
                   (u32 *)S.splidx.p, (u32 *)S.splpoff.p, (uint4 *)S.splrec.p, d_err, GT, group ? 1 : 0, xo);
        if (xk1) {
            HIP_TRY(c, hipEventRecord(S.ev_xk1, c->stream));
            f.x_k1 = true;
        }
        LAUNCH(c, "k1_scan_tiles", k1_scan_tiles, dim3(c->k1s_blocks_forced ? (u32)c->k1s_blocks_forced : k1s_blocks(n_tiles)), dim3(K1S_THREADS), (u32 *)S.tile_cnt.p, (const TileStats *)S.tile_stats.p,
               n_tiles, d_cs, PL, kf, group ? INT32_MAX - 1 : ref_len, (const u64 *)nullptr, (u32 *)S.tile_soff.p, (u32 *)S.chunk_tile.p,
               (ScanPart *)S.scan_parts.p, ++S.scan_epoch);
        // ---- K1b: emit (coordinates in the group's virtual sequence): keys, the pairs' records -- complete for reads of the
        // simple shape --, K2d's candidate keys (free until the first scatter; they are used up before it), the list of
        // reads for k4b_generic
        EmitLists el;
        el.cand = lim.dense ? (u64 *)S.key[1].p : (u64 *)nullptr;
        el.bitmap = lim.dense ? (u64 *)S.bitmap.p : (u64 *)nullptr;
        el.page_cnt = lim.dense ? (u32 *)S.pagecnt.p : (u32 *)nullptr;
        el.cand_anc = (u64 *)S.ent.p; // (a pair-sized scratch buffer nothing else uses at this point)
        el.gen_list = (u64 *)S.genlist.p;
        el.gen_cnt = d_gen_cnt;
        el.gen_cap = gen_cap;
        el.pack_nn = pack_nn;
        // (one launch per chain -- per K1E_MAXB batches --: the blocks stride over the batches' trips of 256 spliced reads; a tile holds
        // ~300 spliced reads = 1.2 trips: a grid of half the tiles keeps two or three trips per block)
        for (size_t b0 = 0; b0 < batches.size(); b0 += K1E_MAXB) {
            const size_t nb = std::min<size_t>(K1E_MAXB, batches.size() - b0);
            u64 reads = 0;
            for (size_t bi = b0; bi < b0 + nb; bi++) reads += (u64)batches[bi].n;
            const u32 nt = (u32)std::min<u64>((reads + K1_TILE - 1) / K1_TILE + nb, 0x7fffffffu);
            // (tiles of reads per block: 2 / 4 / 8 / 16 = 9.17 / 8.96 / 8.94 / 9.05 ms a step, profiles/r05q_*; with round 6's 2-bit compare
            // 4 / 8 / 12 / 16 / 32 / 64 = 7.25 - 7.37 / 7.15 - 7.17 / 7.32 / 7.32 / 7.37 / 8.02: fewer, longer blocks make the kernel itself
            // faster still -- 970 us a chain in the step at 16 against 1 085 at 8 -- but the step no shorter: profiles/r06_k1_experiments.txt)
            static const u32 tiles_per_block = getenv("PJB_K1E_TILES") ? (u32)std::max(1, atoi(getenv("PJB_K1E_TILES"))) : 8u;
            const u32 grid = std::max<u32>(1, std::min<u32>(nt, std::max<u32>(1024, nt / tiles_per_block)));
            LAUNCH(c, "k1_emit", k1_emit, dim3(grid), dim3(256), (const DevBatch *)S.batches.p + b0, (int)nb, n_tiles, (const u32 *)S.tile_cnt.p,
                   (const u32 *)S.tile_soff.p, (const u32 *)S.chunk_tile.p, (const u32 *)S.splidx.p, (const u32 *)S.splpoff.p, (const uint4 *)S.splrec.p, pr, el, kf,
                   GT, fast_codes ? 1 : 0, (int)c->cfg.orientation, d_err, d_cs);
        }
        // (the next chain's K1 may start here: k1_generic -- a few reads walked by a few wavefronts, waiting for their loads -- runs beside
        // its k1_count, which is a stream)
        if (c->k1_serial && !getenv("PJB_K1_SERIAL_LATE")) {
            HIP_TRY(c, hipEventRecord(S.ev_k1, st));
            c->last_k1_ev = S.ev_k1;
        }
        // (per-member counters of a group: off the K1 stage -- beside the next chain's k1_count -- and before k5_finalize counts the
        // members' junctions into them)
        if (group)
            LAUNCH(c, "kg_member_stats", kg_member_stats, dim3((unsigned)n_members), dim3(256), (const u32 *)S.tile_cnt.p, (const TileStats *)S.tile_stats.p,
                   (const u32 *)d_tile_lo, n_members, d_members, n_tiles, (const ContigStats *)d_cs);
        // the reads k1_emit left: one launch over the chain's third list (the blocks stride over it)
        LAUNCH(c, "k1_generic", k1_generic, dim3(std::min<u32>(std::max<u32>(1, (u32)(((u64)gen_cap * GEN_SHARDS + K1E_T - 1) / K1E_T)), 1536u /* six blocks a CU; 512 .. 3072 measured: no difference */)), dim3(K1E_T),
               (const DevBatch *)S.batches.p, (int)batches.size(), (const u32 *)S.splidx.p, (const uint4 *)S.splrec.p, pr, el, kf, GT, fast_codes ? 1 : 0,
               (int)c->cfg.orientation, d_err, d_cs);
    }
    if (c->k1_serial && getenv("PJB_K1_SERIAL_LATE")) { // (experiment: behind k1_generic)
        HIP_TRY(c, hipEventRecord(S.ev_k1, st));
        c->last_k1_ev = S.ev_k1;
    }
    STAGE_EVENT(1);
    // k4b_generic: the pairs that need the generic walks, in BAM order, as soon as junction ids and anchors exist -- beside
    // the sort, on the side stream
    const u32 gen_grid = std::min<u32>(2 * (u32)(((u64)gen_cap * GEN_SHARDS + 255) / 256), 2560u); // (the blocks stride over the lists' entries)
    auto launch_k4b = [&]() -> int {
        LAUNCH(c, "k4b_generic", k4b_generic, dim3(gen_grid), dim3(256), (const u64 *)S.genlist.p, (const u32 *)d_gen_cnt, gen_cap, (const u64 *)pr.key,
               pr.rec, (const u32 *)S.jidbam.p, kf, (const DevBatch *)S.batches.p, (int)batches.size(), (const int32_t *)S.ancl.p, (const int32_t *)S.ancr.p,
               GT, any_x ? 1 : 0, any_x ? 0 : 1, d_err, (const ContigStats *)d_cs, pack_nn);
        return PJB_OK;
    };
    auto fork_k4b = [&]() -> int { // (the main stream has just produced jid_bam and the anchors)
        if (!c->side_stream) return launch_k4b();
        HIP_TRY(c, hipEventRecord(S.ev_fork, st));
        HIP_TRY(c, hipStreamWaitEvent(S.side, S.ev_fork, 0));
        c->stream = S.side;
        f.forked = true;
        const int rc2 = launch_k4b();
        c->stream = st;
        if (rc2) return rc2;
        HIP_TRY(c, hipEventRecord(S.ev_join, S.side));
        return PJB_OK;
    };

    // ---- K2d: ordered dense junction ids (the sort then works on 15-19 bits instead of 46-48)
    int sort_bits = kf.total_bits;
    // the sort's passes: as few as the widest digit allows, bits spread evenly (planned before kd_assign, which counts the first digit)
    const u32 rs_tiles = std::max<u32>(1, (PL + RS_TILE - 1) / RS_TILE);
    int n_pass = 1;
    std::vector<int> pass_bits;
    bool first_hist_done = false;
    auto plan_passes = [&]() -> int {
        n_pass = (sort_bits + c->radix_max_bits - 1) / c->radix_max_bits;
        if (n_pass < 1) n_pass = 1;
        pass_bits.assign((size_t)n_pass, sort_bits / n_pass);
        for (int p = 0; p < sort_bits % n_pass; p++) pass_bits[(size_t)p]++;
        const int dbits = pass_bits[0];
        int rc2;
        if ((rc2 = ensure(c, S.hist, (size_t)rs_tiles * (1u << dbits) * 4))) return rc2;
        if ((rc2 = ensure(c, S.hist_scan, (size_t)rs_tiles * (1u << dbits) * 4))) return rc2;
        if ((rc2 = ensure(c, S.bintotal, (size_t)4 << dbits))) return rc2;
        if ((rc2 = ensure(c, S.hist_part, (size_t)((rs_tiles + RSP_TILES - 1) / RSP_TILES) * (1u << dbits) * 4))) return rc2;
        return PJB_OK;
    };
    if (lim.dense) {
        const u32 cand_blocks = std::min<u32>(pair_blocks, 1024u); // (a few candidates per junction: these kernels stride)
        const u64 *okey = (const u64 *)pr.key;
        u64 *cand = (u64 *)S.key[1].p; // (k1_emit left the candidate keys here, and their starts' bits in the bitmap)
        // (ranks of the bitmap's words: a prefix sum over the pages' counts of starts, then the words of the pages that hold one)
        const u32 n_pages = (u32)((n_words + ((size_t)1 << KD_PAGE_SHIFT) - 1) >> KD_PAGE_SHIFT);
        if ((rc = run_scan(c, "kd_rank", ArrU32Fn{(const u32 *)S.pagecnt.p}, ExclusiveU32Sink{(u32 *)S.pagerank.p}, (u64)n_pages, (u64 *)S.total.p)))
            return rc;
        LAUNCH(c, "kd_rank_pages", kd_rank_pages, dim3((n_pages + 3) / 4), dim3(256), (const u64 *)S.bitmap.p, (const u32 *)S.pagecnt.p, (const u32 *)S.pagerank.p,
               (u32 *)S.wrank.p, n_pages, (u32)n_words);
        u32 *cand_rank = (u32 *)S.idx[1].p; // (free until the first scatter as well)
        LAUNCH(c, "kd_ends", kd_ends, dim3(cand_blocks), dim3(256), (const u64 *)cand, kf, (const u64 *)S.bitmap.p, (const u32 *)S.wrank.p, JL,
               (u32 *)S.ends.p, cand_rank, d_cs);
        if ((rc = run_scan(c, "kd_first", EndsCountFn{(const u32 *)S.ends.p}, FirstIdSink{(u32 *)S.firstid.p, (int32_t *)S.ancl.p, (int32_t *)S.ancr.p},
                           (u64)JL, (u64 *)S.total.p)))
            return rc;
        LAUNCH(c, "kd_table", kd_table, dim3(cand_blocks), dim3(256), (const u64 *)cand, (const u64 *)S.ent.p, (const u32 *)cand_rank, kf, JL,
               (const u32 *)S.ends.p, (const u32 *)S.firstid.p, (const u64 *)S.total.p, (u64 *)S.jkey.p, (int32_t *)S.ancl.p, (int32_t *)S.ancr.p, d_cs,
               (const u32 *)d_gen_cnt, gen_cap, SL);
        // (a block per tile of the sort: it leaves the tile's counts of the ids' first digit -- the first pass below has no rs_hist)
        sort_bits = std::max(1, bits_of((uint64_t)SL));
        if ((rc = plan_passes())) return rc;
        LAUNCH(c, "kd_assign", kd_assign, dim3(rs_tiles), dim3(256), okey, d_P, kf, (const u64 *)S.bitmap.p, (const u32 *)S.wrank.p,
               (const u32 *)S.ends.p, (const u32 *)S.firstid.p, JL, (const u64 *)S.total.p, (u32 *)S.jidbam.p, (u32 *)S.acc.p, (const u64 *)S.jkey.p,
               (const int32_t *)S.ancl.p, (const int32_t *)S.ancr.p, d_err, d_cs, pass_bits[0], (u32 *)S.hist.p);
        first_hist_done = true;
        if ((rc = fork_k4b())) return rc;
        LAUNCH(c, "kd_reset", kd_reset, dim3(cand_blocks), dim3(256), (const u64 *)cand, (const u32 *)cand_rank, kf, JL, (const ContigStats *)d_cs,
               (u64 *)S.bitmap.p, (u32 *)S.ends.p, (u32 *)S.pagecnt.p);
        S.dense_at_rest = true;
    } else if ((rc = plan_passes()))
        return rc;
    // ---- K2: radix sort (key, pair index)
    int cur = 0, shift = 0;
    // digit passes.  Dense ids are 32-bit keys: pass 0 reads them where kd_assign left them (BAM order; k4b_generic reads that array
    // beside the sort, so no pass writes to it), the ping-pong buffers are the first halves of the 64-bit key buffers.
    auto sort_pass = [&](auto key_tag, const void *kin_v, void *kout_v, const u32 *vin, u32 *vout, int bits) -> int {
        using K = decltype(key_tag);
        const K *kin = (const K *)kin_v;
        K *kout = (K *)kout_v;
        if (!(shift == 0 && first_hist_done)) // (dense ids: kd_assign counted the first digit)
            LAUNCH(c, "rs_hist", rs_hist<K>, dim3(rs_tiles), dim3(256), kin, d_P, shift, bits, (u32 *)S.hist.p, rs_tiles);
        {
            const u32 nb = 1u << bits, n_panels = (rs_tiles + RSP_TILES - 1) / RSP_TILES;
            LAUNCH(c, "rs_panel_sums", rs_panel_sums, dim3(n_panels, (nb + 255) / 256), dim3(256), (const u32 *)S.hist.p, rs_tiles, nb, (u32 *)S.hist_part.p);
            LAUNCH(c, "rs_panel_scan", rs_panel_scan, dim3(n_panels, (nb + 255) / 256), dim3(256), (const u32 *)S.hist.p, (const u32 *)S.hist_part.p, rs_tiles, nb,
                   (u32 *)S.hist_scan.p, (u32 *)S.bintotal.p);
        }
#define RS_SCATTER(B)                                                                                                                          \
    LAUNCH_LDS(c, "rs_scatter", (rs_scatter<B, K>), dim3(rs_tiles), dim3(256), rs_scatter_lds_bytes(bits, sizeof(K)), kin, vin, kout, vout, d_P, \
               shift, bits, (const u32 *)S.hist_scan.p, (const u32 *)S.bintotal.p, rs_tiles)
        switch (bits) { // the usual digit widths get an unrolled match loop
        case 9: RS_SCATTER(9); break;
        case 10: RS_SCATTER(10); break;
        case 11: RS_SCATTER(11); break;
        default: RS_SCATTER(0); break;
        }
#undef RS_SCATTER
        return PJB_OK;
    };
    for (int p = 0; p < n_pass; p++) {
        const int bits = pass_bits[(size_t)p];
        if (bits <= 0) break;
        const u32 *vin = p == 0 ? nullptr : (const u32 *)S.idx[cur].p;
        u32 *vout = (u32 *)S.idx[cur ^ 1].p;
        if (lim.dense) rc = sort_pass((u32)0, p == 0 ? S.jidbam.p : S.key[cur].p, S.key[cur ^ 1].p, vin, vout, bits);
        else rc = sort_pass((u64)0, p == 0 ? S.okey.p : S.key[cur].p, S.key[cur ^ 1].p, vin, vout, bits);
        if (rc) return rc;
        cur ^= 1;
        shift += bits;
    }
    f.n_pass = n_pass;
    const u32 *sidx = (const u32 *)S.idx[cur].p;
    f.sidx = sidx;
    const u32 *jid_sorted = lim.dense ? (const u32 *)S.key[cur].p : (const u32 *)S.jid.p; // junction id of every sorted pair
    f.jid_sorted = jid_sorted;
    STAGE_EVENT(2);
    if ((rc = ensure(c, S.entsum, (size_t)JL * 8 + 16))) return rc;
    const u32 n_slices_lim = (PL + 63) / 64 + 1;
    const hipStream_t tl = st;
    auto entropy_kernels = [&]() -> int {
        LAUNCH(c, "k5_entropy_sum", k5_entropy_sum, dim3(std::max<u32>(1, (JL + 15) / 16)), dim3(256), (const u32 *)S.seg.p, (const u32 *)S.runfirst.p,
               (const u32 *)S.runstart.p, d_J, (double *)S.entsum.p);
        return PJB_OK;
    };
    bool entropy_forked = false;
    auto fork_entropy = [&]() -> int { // the entropy kernels beside what follows on the main stream (small and latency-bound, both)
        if (!c->side_stream) return entropy_kernels();
        HIP_TRY(c, hipEventRecord(S.ev_fork2, st));
        HIP_TRY(c, hipStreamWaitEvent(S.side, S.ev_fork2, 0));
        c->stream = S.side;
        entropy_forked = true;
        const int rc2 = entropy_kernels();
        c->stream = st;
        if (rc2) return rc2;
        HIP_TRY(c, hipEventRecord(S.ev_join2, S.side));
        return PJB_OK;
    };
    if (lim.dense) {
        // ---- the usual chain: the sorted ids are the junction ids; K4 gathers the pairs -- one 32-byte record each -- folds them
        // to fragments and leaves the junction / run boundaries as bit masks; two small kernels turn the masks into
        // seg_off / run_first / run_start (no second pass over the pairs), then the entropy beside the fragment reduce
        if ((rc = ensure(c, S.masks, (size_t)n_slices_lim * 8 * 2 + (size_t)n_slices_lim * 4))) return rc;
        u64 *head_mask = (u64 *)S.masks.p, *run_mask = head_mask + n_slices_lim;
        u32 *run_base = (u32 *)(run_mask + n_slices_lim);
        STAGE_EVENT(3);
        STAGE_EVENT(4);
        if (f.forked) HIP_TRY(c, hipStreamWaitEvent(tl, S.ev_join, 0)); // k4b_generic's results (side stream) are needed from here on
#ifdef PJB_DEBUG_LAUNCH
        fprintf(stderr, "[k4_pairs] tid %d PL %u JL %u slots_lim %u n_slices_lim %u frag cap %zu fragj cap %zu masks cap %zu rec cap %zu idx cap %zu key cap %zu jkey cap %zu cur %d n_pass %d\n",
                f.tid, PL, JL, slots_lim, n_slices_lim, S.frag.cap, S.fragj.cap, S.masks.cap, S.rec.cap, S.idx[cur].cap, S.key[cur].cap, S.jkey.cap, cur, n_pass);
        {
            ContigStats hcs;
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(&hcs, d_cs, sizeof hcs, hipMemcpyDeviceToHost);
            u64 htot = 0;
            (void)hipMemcpy(&htot, S.total.p, 8, hipMemcpyDeviceToHost);
            fprintf(stderr, "[k4_pairs] cs: P %u n_pairs %llu J %u n_junc %u n_slots %u n_slices %u n_cand %u overflow %u R %u total %llu slot %d attempt %d\n", hcs.P,
                    (unsigned long long)hcs.n_pairs, hcs.J, hcs.n_junc, hcs.n_slots, hcs.n_slices, hcs.n_cand, hcs.overflow, hcs.R, (unsigned long long)htot, f.slot, f.attempt);
            // is the sort's output a permutation with ascending ids?  is its input what kd_assign wrote?
            std::vector<u32> hs(hcs.P), hj(hcs.P), hb(hcs.P);
            (void)hipMemcpy(hs.data(), sidx, (size_t)hcs.P * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hj.data(), jid_sorted, (size_t)hcs.P * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hb.data(), S.jidbam.p, (size_t)hcs.P * 4, hipMemcpyDeviceToHost);
            std::vector<char> seen(hcs.P, 0);
            size_t bad_idx = 0, dup = 0, bad_order = 0, bad_in = 0, mism = 0;
            long first_bad = -1;
            for (size_t i = 0; i < hcs.P; i++) {
                if (hb[i] >= hcs.J) bad_in++;
                if (hs[i] >= hcs.P) {
                    bad_idx++;
                    if (first_bad < 0) first_bad = (long)i;
                    continue;
                }
                if (seen[hs[i]]) dup++;
                seen[hs[i]] = 1;
                if (hj[i] != hb[hs[i]]) mism++;
                if (i && hj[i] < hj[i - 1]) bad_order++;
            }
            fprintf(stderr, "[k4_pairs] sort check: %zu indices out of range (first at %ld), %zu duplicates, %zu order breaks, %zu key/index mismatches, %zu input ids >= J\n", bad_idx,
                    first_bad, dup, bad_order, mism, bad_in);
        }
#endif
        LAUNCH(c, "k4_pairs", k4_pairs, dim3(pair_blocks), dim3(256), sidx, jid_sorted, (const PairRec *)pr.rec, (const u64 *)S.jkey.p, kf, d_P,
               (u32 *)S.frag.p, (int32_t *)S.fragj.p, head_mask, run_mask, (const ContigStats *)d_cs, d_err);
        if ((rc = run_scan(c, "k2_runs", Popc64Fn{(const u64 *)run_mask}, ExclusiveU32Sink{run_base}, (u64)n_slices_lim, (u64 *)S.total.p, &d_cs->n_slices)))
            return rc;
        LAUNCH(c, "k2_expand", k2_expand, dim3((pair_blocks + K2E_PER - 1) / K2E_PER), dim3(256), jid_sorted, (const u64 *)head_mask, (const u64 *)run_mask, (const u32 *)run_base,
               (const u64 *)S.total.p, (u32 *)S.seg.p, (u32 *)S.runfirst.p, (u32 *)S.runstart.p, d_cs);
        STAGE_EVENT(5);
        if ((rc = fork_entropy())) return rc;
    } else {
        // ---- a chain that sorted the full keys: junction ids and position runs from a scan over the sorted pairs (it fetches
        // every pair's read position), the junctions' keys, anchors, ids in BAM order -- then the generic pairs and K4
        const u64 *skey = (const u64 *)S.key[cur].p;
        HeadFn hf{skey, sidx, (const PairRec *)pr.rec};
        HeadSink hs{(u32 *)S.jid.p, (u32 *)S.seg.p, (u32 *)S.runfirst.p, (u32 *)S.runstart.p, skey, (u64 *)S.jkey.p, JL};
        if ((rc = run_scan(c, "k2_heads", hf, hs, (u64)PL, (u64 *)S.total.p, d_P))) return rc;
        LAUNCH(c, "k2_close", k2_close, dim3(1), dim3(1), (u64 *)S.total.p, (u32 *)S.seg.p, (u32 *)S.runfirst.p,
               (u32 *)S.runstart.p, d_cs, JL, (const u32 *)d_gen_cnt, gen_cap);
        STAGE_EVENT(3);
        if ((rc = fork_entropy())) return rc;
        LAUNCH(c, "kf_init", kf_init, dim3(std::max<u32>(1, std::min<u32>((JL * F_WORDS + 255) / 256, 4096))), dim3(256), (u32 *)S.acc.p, d_J, (int32_t *)S.ancl.p,
               (int32_t *)S.ancr.p);
        LAUNCH(c, "kf_anchors", kf_anchors, dim3(pair_blocks), dim3(256), sidx, (const u32 *)S.jid.p, (const PairRec *)pr.rec, d_P, (u32 *)S.jidbam.p,
               (int32_t *)S.ancl.p, (int32_t *)S.ancr.p);
        if ((rc = launch_k4b())) return rc;
        STAGE_EVENT(4);
        LAUNCH(c, "k4_pairs", k4_pairs, dim3(pair_blocks), dim3(256), sidx, jid_sorted, (const PairRec *)pr.rec, (const u64 *)S.jkey.p, kf, d_P,
               (u32 *)S.frag.p, (int32_t *)S.fragj.p, (u64 *)nullptr, (u64 *)nullptr, (const ContigStats *)d_cs, d_err);
        STAGE_EVENT(5);
    }

    // ---- K5: fragments -> junctions -> rows
    LAUNCH(c, "k5_frag_reduce", k5_frag_reduce, dim3((slots_lim + 4 * FRAG_SLOTS_PER_WAVE - 1) / (4 * FRAG_SLOTS_PER_WAVE)), dim3(256),
           (const u32 *)S.frag.p, (const int32_t *)S.fragj.p, d_slots, (u32 *)S.acc.p);
    if (entropy_forked) HIP_TRY(c, hipStreamWaitEvent(tl, S.ev_join2, 0));
    LAUNCH(c, "k5_finalize", k5_finalize, dim3(std::max<u32>(1, (JL + 255) / 256)), dim3(256), (const u64 *)S.jkey.p, (const u32 *)S.seg.p,
           (const u32 *)S.runfirst.p, (const u32 *)S.runstart.p, (const u32 *)S.acc.p,
           (const int32_t *)S.ancl.p, (const int32_t *)S.ancr.p, kf, GT, d_J, (const double *)S.entsum.p, (pjb_junction_row *)S.rows.p, d_err,
           d_member_junc);
    STAGE_EVENT(6);

    // ---- rows to the host (and into the caller's exchange slot), control block last: on the rows stream, so that the
    // next contig's first kernels need not wait for the PCIe writes
    u64 *mirror_table = nullptr;
    u32 mirror_room = 0; // rows the caller's slot can take (the kernel leaves the slot alone if the contig does not fit)
    if (c->mirror && c->mirror_cap >= PJB_MIRROR_HEADER_BYTES) {
        mirror_table = (u64 *)(c->mirror + PJB_MIRROR_HEADER_BYTES);
        mirror_room = (u32)std::min<size_t>((c->mirror_cap - PJB_MIRROR_HEADER_BYTES) / sizeof(pjb_junction_row), 0xffffffffu);
    }
    const hipStream_t rows_stream = c->side_stream ? c->stream3 : tl;
    if (rows_stream != tl) {
        HIP_TRY(c, hipEventRecord(S.ev_rows, tl));
        HIP_TRY(c, hipStreamWaitEvent(rows_stream, S.ev_rows, 0));
    }
    {
        struct StreamScope {
            pjb_ctx *c;
            hipStream_t main;
            ~StreamScope() { c->stream = main; }
        } scope{c, c->stream};
        c->stream = rows_stream; // LAUNCH (and its event bracket) follow c->stream
        LAUNCH(c, "k6_rows_out", k6_rows_out, dim3(K6_BLOCKS), dim3(256), (const u64 *)S.rows.p, (const ContigStats *)d_cs,
               (u64 *)c->rows_table, row_base, mirror_base, (RowCursor *)c->b_cursor.p, mirror_table, mirror_room, d_err, d_gen_cnt, S.pub_dev,
               (const MemberStats *)d_members, d_member_junc, n_members);
    }
    S.at_rest = true;
    HIP_TRY(c, hipEventRecord(S.ev[7], rows_stream));
    HIP_TRY(c, hipEventRecord(S.ev_done, rows_stream));
#undef STAGE_EVENT
    f.queued = true;
    return PJB_OK;
}

// blocks until the contig's rows and control block are on the host
static void wait_flight(pjb_ctx *c, Flight &f) {
    if (f.queued) (void)hipEventSynchronize(c->sl[f.slot].ev_done);
}

// The contigs queued behind fl[0] are taken back (after an overflow or an error of fl[0] their rows are in the wrong
// place): their device work is allowed to finish and is thrown away; their own pjb_finish_contig_end queues them again.
static void unqueue_followers(pjb_ctx *c) {
    for (int k = 1; k < c->n_fl; k++) {
        Flight &g = c->fl[k];
        if (!g.queued) continue;
        wait_flight(c, g);
        if (g.forked) (void)hipStreamSynchronize(c->sl[g.slot].side);
        g.queued = false;
        g.forked = false;
        ev_drop(c, g.slot);
    }
}

static void pop_flight(pjb_ctx *c) {
    if (c->n_fl <= 0) return;
    c->slot_busy[c->fl[0].slot] = false;
    for (int k = 1; k < c->n_fl; k++) c->fl[k - 1] = c->fl[k];
    c->n_fl--;
    c->fl[c->n_fl] = Flight();
}

// The flight's batch list, virtual offsets, tile ranges and limits from the open targets named in f.tids.
static void prepare_flight(pjb_ctx *c, Flight &f) {
    f.batches.clear();
    f.batch_member.clear();
    f.voff.assign(f.tids.size(), 0);
    f.tile_lo.assign(f.tids.size() + 1, 0);
    f.m_reads.assign(f.tids.size(), 0);
    f.n_tiles = 0;
    f.n_reads = 0;
    int64_t at = 0;
    bool genomes_ok = true;
    for (size_t m = 0; m < f.tids.size(); m++) {
        const int32_t tid = f.tids[m];
        const int32_t ref_len = c->ref_len[(size_t)tid];
        f.voff[m] = (int32_t)at;
        at += (((int64_t)std::max(ref_len, 1) + GROUP_GAP) + 63) & ~(int64_t)63;
        f.tile_lo[m] = f.n_tiles;
        const Contig &G = c->contigs[(size_t)tid];
        auto open_it = c->open.find(tid);
        if (open_it == c->open.end() || open_it->second.batches.empty()) continue;
        // without the target's genome only the counting stage may run: any pair then shows up as an overflow
        if (!G.present || G.len != ref_len) genomes_ok = false;
        OpenContig &oc = open_it->second;
        int32_t prev_pos = INT32_MIN;
        const int32_t *prev_ptr = nullptr;
        for (size_t k = 0; k < oc.batches.size(); k++) {
            DevBatch b = oc.batches[k];
            b.base = (uint32_t)f.n_reads; // read ordinals and tile numbers run through the whole group
            b.tile_base = f.n_tiles;
            f.n_tiles += (u32)((b.n + K1_TILE - 1) / K1_TILE);
            b.prev_pos = prev_pos;
            b.prev_pos_ptr = prev_ptr; // sortedness across batches: the last position of the previous batch, wherever it is known
            f.n_reads += b.n;
            f.m_reads[m] += b.n;
            if (oc.last_known[k]) {
                prev_pos = oc.last_pos[k];
                prev_ptr = nullptr;
            } else
                prev_ptr = b.pos + (b.n - 1);
            b.member = (int32_t)m;
            f.batches.push_back(b);
            f.batch_member.push_back((int)m);
        }
    }
    f.tile_lo[f.tids.size()] = f.n_tiles;
    f.vlen = f.tids.size() > 1 ? at : c->ref_len[(size_t)f.tid];
    f.empty = f.batches.empty();
    // ---- limits.  Pairs: a share of the reads (a chain with more N operations than that is repeated once with the
    // exact count).  Junctions: a share of the pair limit, at least twice what a chain of this context has had.  Key:
    // coordinates of the (virtual) sequence and the longest intron seen by this context so far.
    ContigLimits &lim = f.lim;
    const u64 guess = std::min<u64>(0xffffff00ull, (u64)f.n_reads * 5 / 8 + 4096);
    lim.pair_limit = (u32)std::max<u64>(guess, 4096);
    lim.junc_limit = std::max<u32>(std::max<u32>(lim.pair_limit / 32, 4096), 2 * c->junc_seen * (u32)std::max<size_t>(1, f.tids.size() > 1 ? 2 : 1));
    if (!genomes_ok) lim.pair_limit = 0;
    lim.kf.raw = 0;
    lim.kf.lbits = std::max(1, c->lbits_seen);
    lim.kf.total_bits = lim.kf.lbits + std::max(1, bits_of((uint64_t)std::max<int64_t>(f.vlen, 1)));
    lim.dense = c->dense_ids;
    // (twice what a chain of this context's targets has had, per member; a chain with more is repeated with digits for junc_limit)
    // (junctions per read: chains of one file have about the same, whatever their targets' number and size)
    lim.sort_limit = c->junc_per_read > 0 ? std::min<u32>(lim.junc_limit, std::max<u32>(c->sort_floor, (u32)std::min<double>(4.0e9, 2.0 * c->junc_per_read * (double)f.n_reads + 64.0))) : 0u;
}

// streams and events of a control slot, at its first use (a context that never queues eight chains never pays for them)
static int slot_init(pjb_ctx *c, int k) {
    CtlSlot &S = c->sl[k];
    if (S.ev_done) return PJB_OK;
    for (auto &ev : S.ev) HIP_TRY(c, hipEventCreate(&ev));
    HIP_TRY(c, hipEventCreateWithFlags(&S.ev_rows, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&S.ev_done, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&S.ev_k1, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&S.ev_xk1, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&S.ev_fork, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&S.ev_join, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&S.ev_fork2, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&S.ev_join2, hipEventDisableTiming));
    // (main before side, slot after slot: the order in which streams are created decides which hardware queue they share, and
    // every other order tried -- side first, all mains first, one to three streams created before -- cost the configs[2] step
    // 5 - 12 %: profiles/r03z_stream_order.txt)
    HIP_TRY(c, hipStreamCreateWithFlags(&S.main, hipStreamNonBlocking));
    HIP_TRY(c, hipStreamCreateWithFlags(&S.side, hipStreamNonBlocking));
    return PJB_OK;
}

// the chain and, with PJB_FLAG_EXTRA, the part of the extra metrics that needs the records only (service stream, beside the chain)
static int queue_chain(pjb_ctx *c, Flight &f) {
    int rc = queue_contig(c, f);
    if (!rc && c->extra) rc = extra_pre(c, f);
    return rc;
}

static int begin_flight(pjb_ctx *c, const int32_t *tids, int32_t n, const char *who) {
    if (!c) return PJB_ERR_ARG;
    if (!tids || n < 1 || n > GROUP_MAX) return fail(c, PJB_ERR_ARG, "%s: 1 to %d targets", who, GROUP_MAX);
    aux_streams(c);
    for (int32_t k = 0; k < n; k++) {
        if (tids[k] < 0 || (size_t)tids[k] >= c->ref_len.size()) return fail(c, PJB_ERR_ARG, "%s: bad tid %d", who, tids[k]);
        for (int32_t q = 0; q < k; q++)
            if (tids[q] == tids[k]) return fail(c, PJB_ERR_ARG, "%s: target %d named twice", who, tids[k]);
    }
    if (c->n_fl >= PJB_MAX_QUEUED)
        return fail(c, PJB_ERR_STATE, "%s: %d chains are queued already (oldest: target %d); collect one first", who, c->n_fl, c->fl[0].tid);
    for (int k = 0; k < c->n_fl; k++)
        for (int32_t t : c->fl[k].tids)
            for (int32_t q = 0; q < n; q++)
                if (t == tids[q]) return fail(c, PJB_ERR_STATE, "%s: target %d is queued already", who, t);
    if (n > 1) {
        // what a group needs (a caller that gets PJB_ERR_ARG here finishes the targets one by one)
        if (c->extra) return fail(c, PJB_ERR_ARG, "%s: PJB_FLAG_EXTRA contexts finish one target at a time", who);
        int64_t vlen = 0;
        for (int32_t k = 0; k < n; k++) {
            const Contig &G = c->contigs[(size_t)tids[k]];
            vlen += (((int64_t)std::max(c->ref_len[(size_t)tids[k]], 1) + GROUP_GAP) + 63) & ~(int64_t)63;
            auto it = c->open.find(tids[k]);
            const bool has_reads = it != c->open.end() && !it->second.batches.empty();
            if (has_reads && (!G.present || G.len != c->ref_len[(size_t)tids[k]]))
                return fail(c, PJB_ERR_ARG, "%s: the genome of target %d has not been uploaded", who, tids[k]);
            if (has_reads && (!G.codes || G.has_x))
                return fail(c, PJB_ERR_ARG, "%s: target %d has characters outside the 16-letter alphabet: finish it alone", who, tids[k]);
        }
        if (vlen >= (int64_t)INT32_MAX - (1 << 20)) return fail(c, PJB_ERR_ARG, "%s: the group's targets add up to %lld bases (limit 2^31)", who, (long long)vlen);
    }
    c->cur_tid = tids[0];
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    Flight &f = c->fl[c->n_fl];
    int slot = 0;
    while (slot < PJB_MAX_QUEUED && c->slot_busy[slot]) slot++;
    if (slot >= PJB_MAX_QUEUED) return fail(c, PJB_ERR_STATE, "%s: no free control slot", who);
    {
        const int rc = slot_init(c, slot);
        if (rc) return rc;
    }
    f = Flight();
    f.slot = slot;
    f.tid = tids[0];
    f.tids.assign(tids, tids + n);
    c->slot_busy[slot] = true;
    prepare_flight(c, f);
    c->n_fl++;
    if (f.empty) return PJB_OK;
    // a chain that was taken back goes first (rows are in queue order): the end of the oldest one queues them all
    for (int k = 0; k + 1 < c->n_fl; k++)
        if (!c->fl[k].queued && !c->fl[k].empty) return PJB_OK;
    const int rc = queue_chain(c, f);
    if (rc) { // nothing of this chain stays behind
        (void)hipDeviceSynchronize();
        c->n_fl--;
        c->slot_busy[c->fl[c->n_fl].slot] = false;
        std::vector<int32_t> members = c->fl[c->n_fl].tids;
        c->fl[c->n_fl] = Flight();
        for (int32_t t : members) close_contig(c, t);
    }
    return rc;
}

int pjb_finish_contig_begin(pjb_ctx *c, int32_t tid) { return begin_flight(c, &tid, 1, "finish"); }
int pjb_finish_group_begin(pjb_ctx *c, const int32_t *tids, int32_t n_tids) { return begin_flight(c, tids, n_tids, "finish_group"); }

// Collects the chain in fl[0] -- waits for it, repeats it when a limit it was queued with turned out too small -- and
// fills one result per member.  A group whose members turn out to hold alignments outside their own sequence is taken
// apart: *redo_single is set and nothing is committed.
static int collect_flight(pjb_ctx *c, pjb_region_result *res, bool *redo_single) {
    Flight &f = c->fl[0];
    const int n_members = (int)f.tids.size();
    const bool group = n_members > 1;
    CtlSlot &S = c->sl[f.slot];
    ContigLimits &lim = f.lim;
    const int32_t tid = f.tid;
    const int32_t ref_len = group ? (int32_t)f.vlen : c->ref_len[(size_t)tid];
    int rc;
    ContigStats cs;
    u64 herr = ~0ull;
    int64_t repeats = 0, repeat_reasons = 0;
    for (;; f.attempt++) {
        if (!f.queued && (rc = queue_chain(c, f))) return rc;
        wait_flight(c, f);
        memcpy(&cs, S.pub, sizeof cs);
        memcpy(&herr, S.pub + PUB_ERR_AT, 8);
        if ((rc = check_device_error(c, herr))) return rc;
        for (int m = 0; m < n_members && cs.n_pairs > 0; m++) {
            const Contig &G = c->contigs[(size_t)f.tids[(size_t)m]];
            if (f.m_reads[(size_t)m] == 0) continue;
            if (!G.present) return fail(c, PJB_ERR_STATE, "finish: genome of target %d was not uploaded", f.tids[(size_t)m]);
            if (G.len != c->ref_len[(size_t)f.tids[(size_t)m]])
                return fail(c, PJB_ERR_ARG, "finish: genome of target %d has %lld bases, header says %d", f.tids[(size_t)m], (long long)G.len,
                            c->ref_len[(size_t)f.tids[(size_t)m]]);
        }
        if (!cs.overflow) break;
        // (a repeat for the read lists' room alone has a budget of its own: it must not use up the attempts the other limits may need)
        if (cs.overflow == OVF_LISTS && f.list_attempt < 3) {
            f.list_attempt++;
            f.attempt--;
        } else if (f.attempt >= 3)
            return fail(c, PJB_ERR_STATE, "finish: limits of target %d did not settle (overflow bits %u)", tid, cs.overflow);
        // a limit was too small: the control block says by how much; everything is queued again (and so is the chain
        // queued behind this one: its rows went where this one's belong)
        repeats++;
        repeat_reasons |= (int64_t)cs.overflow;
        unqueue_followers(c);
        if (f.forked) (void)hipStreamSynchronize(c->sl[f.slot].side);
        f.queued = f.forked = false;
        if (cs.overflow & OVF_PAIRS) {
            if (cs.n_pairs >= 0xfffffff0ull)
                return fail(c, PJB_ERR_ARG, group ? "finish_group: more than 2^32 spliced pairs in one group: finish its targets in smaller groups"
                                                  : "finish: more than 2^32 spliced pairs on one target are not supported");
            lim.pair_limit = (u32)cs.n_pairs + 64;
            lim.junc_limit = std::max<u32>(lim.junc_limit, lim.pair_limit / 8);
        }
        if (cs.overflow & OVF_KEYFMT) {
            const bool weird = cs.min_pos < 0 || cs.max_end > ref_len || cs.max_end < 0;
            if (weird && group) { // (k1_count reports INT32_MAX for a member whose alignments leave it)
                *redo_single = true;
                return PJB_OK;
            }
            if (weird) {
                lim.kf.raw = 1;
                lim.kf.lbits = 32;
                lim.kf.total_bits = 64;
                lim.dense = false; // the bitmap of intron starts needs coordinates inside the contig
            } else {
                lim.kf.lbits = std::max(1, bits_of((uint64_t)cs.max_nlen));
                lim.kf.total_bits = lim.kf.lbits + std::max(1, bits_of((uint64_t)std::max(ref_len, 1)));
            }
        }
        const bool sort_only = (cs.overflow & OVF_JUNC) && lim.sort_limit && cs.n_junc <= lim.junc_limit; // (the buffers were large enough)
        if (cs.overflow & OVF_JUNC) lim.sort_limit = 0;
        if ((cs.overflow & OVF_JUNC) && !sort_only) lim.junc_limit = std::max<u32>(cs.n_junc + 64, (cs.overflow & OVF_DENSE) || !lim.dense ? 0u : lim.junc_limit * 4);
        if (cs.overflow & OVF_DENSE) lim.dense = false; // a donor with more alternative acceptors than K2d keeps: sort the full keys
        if (cs.overflow & OVF_LISTS) lim.list_cap = std::max(gen_list_cap(lim.pair_limit), (cs.list_need + cs.list_need / 4 + 511u) & ~255u); // (k1_generic's entries depend on the appends' order: some slack)
    }
    if (!lim.kf.raw) c->lbits_seen = std::max(c->lbits_seen, std::max(1, bits_of((uint64_t)cs.max_nlen)));
    const u32 P = cs.P, J = cs.J;
    // ---- one result per member
    pjb_region_result Rsum;
    memset(&Rsum, 0, sizeof Rsum);
    Rsum.min_len = INT32_MAX;
    if (!group) {
        pjb_region_result &R = res[0];
        R.spliced = cs.spliced;
        R.unspliced = cs.unspliced;
        R.sum_len = cs.sum_len;
        R.min_len = cs.min_len;
        R.max_len = cs.max_len;
        R.n_reads = f.n_reads;
        R.n_pairs = (int64_t)cs.n_pairs;
        R.n_junctions = J;
        Rsum = R;
    } else {
        const MemberStats *ms = (const MemberStats *)(S.pub + PUB_MEMBERS_AT);
        u64 pairs = 0, juncs = 0;
        for (int m = 0; m < n_members; m++) {
            pjb_region_result &R = res[m];
            memset(&R, 0, sizeof R);
            R.min_len = INT32_MAX;
            if (f.m_reads[(size_t)m] == 0) continue;
            R.spliced = ms[m].spliced;
            R.unspliced = ms[m].unspliced;
            R.sum_len = ms[m].sum_len;
            R.min_len = ms[m].min_len;
            R.max_len = ms[m].max_len;
            R.n_reads = f.m_reads[(size_t)m];
            R.n_pairs = (int64_t)ms[m].n_pairs;
            R.n_junctions = ms[m].n_junc;
            pairs += ms[m].n_pairs;
            juncs += ms[m].n_junc;
        }
        if (pairs != cs.n_pairs || juncs != J)
            return fail(c, PJB_ERR_STATE, "finish_group: members hold %llu pairs / %llu junctions, the chain %llu / %u", (unsigned long long)pairs,
                        (unsigned long long)juncs, (unsigned long long)cs.n_pairs, J);
        Rsum.spliced = cs.spliced;
        Rsum.unspliced = cs.unspliced;
        Rsum.sum_len = cs.sum_len;
        Rsum.min_len = cs.min_len;
        Rsum.max_len = cs.max_len;
    }
    c->junc_seen = std::max(c->junc_seen, group ? J / (u32)n_members : J);
    if (f.n_reads > 0) c->junc_per_read = std::max(c->junc_per_read, (double)J / (double)f.n_reads);
    c->timing.sort_passes = f.n_pass;
    c->timing.repeats = repeats;
    c->timing.repeat_reasons = repeat_reasons;
    c->timing.generic_pairs = 0;
    c->timing.generic_reads = 0;
    c->timing.checked_reads = *(const u32 *)(S.pub + PUB_CHECKED_AT);
    for (u32 k = 0; k < GEN_SHARDS; k++) {
        c->timing.generic_pairs += ((const u32 *)(S.pub + PUB_GEN_AT))[k];
        c->timing.generic_reads += ((const u32 *)(S.pub + PUB_GREADS_AT))[k];
    }
    c->timing.position_runs = cs.R;
    c->timing.candidates = lim.dense ? cs.n_cand : 0;
    const size_t old = c->rows_n;
    {
        u32 at[2];
        memcpy(at, S.pub + PUB_BASE_AT, 8);
        if (at[0] != (u32)old || (c->mirror && at[1] != (u32)c->mirror_rows))
            return fail(c, PJB_ERR_STATE, "finish: rows of target %d went to %u (exchange slot %u), expected %zu (%zu)", tid, at[0], at[1], old, c->mirror_rows);
    }
    if (c->mirror) { // the rows are in the exchange slot already (k6_rows_out); the header follows, covered by a small wait
        const size_t need = PJB_MIRROR_HEADER_BYTES + (c->mirror_rows + J) * sizeof(pjb_junction_row);
        if (need > c->mirror_cap) return fail(c, PJB_ERR_ARG, "finish: %zu rows do not fit the row mirror (%zu bytes)", c->mirror_rows + J, c->mirror_cap);
        mirror_fold(c, Rsum, J);
        HIP_TRY(c, hipMemcpyAsync(c->mirror, c->mirror_hdr, PJB_MIRROR_HEADER_BYTES, hipMemcpyHostToDevice, c->stream4));
        HIP_TRY(c, hipStreamSynchronize(c->stream4));
    }
    if (J) { // the chain's rows: HBM table -> host table, by DMA, behind whatever the caller does next
        HIP_TRY(c, hipMemcpyAsync(c->rows_pinned + old, c->rows_table + old, (size_t)J * sizeof(pjb_junction_row), hipMemcpyDeviceToHost, c->stream4));
        c->rows_copy_pending = true;
    }
    c->cur_slot = f.slot;
    if (c->extra && (rc = extra_contig(c, f, tid, cs.spliced, P, J, old))) return rc;
    c->rows_n = old + J;
    c->last_rows_n = J;
    c->last_slot = f.slot;
    if (c->ktime) ev_collect(c, f.slot);
    if (c->ktime && c->ktime_only.empty())
        for (int k = 0; k < 7; k++) (void)hipEventElapsedTime(&c->timing.stage_ms[k], S.ev[k], S.ev[k + 1]);
    (void)hipEventElapsedTime(&c->timing.total_ms, S.ev[0], S.ev[7]);
    return PJB_OK;
}

static int end_flight(pjb_ctx *c, const int32_t *tids, int32_t n, pjb_region_result *res, const char *who) {
    if (!c) return PJB_ERR_ARG;
    if (!tids || n < 1) return fail(c, PJB_ERR_ARG, "%s: no targets", who);
    bool same = c->n_fl > 0 && (int32_t)c->fl[0].tids.size() == n;
    for (int32_t k = 0; same && k < n; k++) same = c->fl[0].tids[(size_t)k] == tids[k];
    if (!same)
        return fail(c, PJB_ERR_STATE, "%s: target %d (and the %d named with it) is not the oldest queued chain (begin first; collect in the same order, "
                                      "with the same targets)", who, tids[0], n - 1);
    c->cur_tid = tids[0];
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    struct Closer { // also on every error path: the side stream (k4b_generic) may still read the contig's batches
        pjb_ctx *c;
        bool ok = false;
        ~Closer() {
            if (!ok) { // whatever is queued is in an unknown place now
                (void)hipDeviceSynchronize();
                unqueue_followers(c);
            }
            const std::vector<int32_t> members = c->fl[0].tids;
            pop_flight(c);
            for (int32_t t : members) close_contig(c, t);
        }
    } closer{c};
    std::vector<pjb_region_result> tmp((size_t)n);
    for (auto &R : tmp) {
        memset(&R, 0, sizeof R);
        R.min_len = INT32_MAX;
    }
    memset(&c->timing, 0, sizeof c->timing);
    c->last_rows_n = 0;
    if (res) memcpy(res, tmp.data(), tmp.size() * sizeof(pjb_region_result));
    int rc;
    if (c->fl[0].empty) {
        rc = mirror_header_only(c, tmp[0]);
        closer.ok = rc == PJB_OK;
        return rc;
    }
    bool redo_single = false;
    if ((rc = collect_flight(c, tmp.data(), &redo_single))) return rc;
    if (redo_single) {
        // a member holds alignments outside its own sequence: the members go through one by one, in this chain's slot
        const Flight whole = c->fl[0];
        size_t rows_total = 0;
        for (int32_t m = 0; m < n; m++) {
            Flight one = Flight();
            one.slot = whole.slot;
            one.tid = whole.tids[(size_t)m];
            one.tids.assign(1, one.tid);
            prepare_flight(c, one);
            c->fl[0] = one;
            if (one.empty) continue;
            bool again = false;
            rc = collect_flight(c, &tmp[(size_t)m], &again);
            if (rc) {
                c->fl[0] = whole; // (the closer releases every member)
                return rc;
            }
            rows_total += c->last_rows_n;
        }
        c->fl[0] = whole;
        c->fl[0].queued = c->fl[0].forked = false;
        c->last_rows_n = rows_total; // (pjb_collect_device covers the last member only: a caller of groups uses pjb_collect)
    }
    if (res) memcpy(res, tmp.data(), tmp.size() * sizeof(pjb_region_result));
    closer.ok = true;
    // followers that were taken back (or waited for this one) are queued now, in order, the first one's place known
    c->fl[0].queued = false; // (fl[0] is still this chain: its rows are collected, it is not "ahead" of anything)
    for (int k = 1; k < c->n_fl; k++) {
        Flight &g = c->fl[k];
        if (g.queued || g.empty) continue;
        // a follower that cannot be queued here is queued again -- and reports its error -- by its own _end; THIS
        // chain is collected and its rows are in the table
        if (queue_chain(c, g)) break;
    }
    return PJB_OK;
}

int pjb_finish_ready(pjb_ctx *c) {
    if (!c || c->n_fl <= 0) return 1;
    const Flight &f = c->fl[0];
    if (f.empty || !f.queued) return 1;
    const bool done = hipEventQuery(c->sl[f.slot].ev_done) == hipSuccess;
    (void)hipGetLastError(); // (hipErrorNotReady is not an error here)
    return done ? 1 : 0;
}

int pjb_finish_contig_end(pjb_ctx *c, int32_t tid, pjb_region_result *res) { return end_flight(c, &tid, 1, res, "finish"); }
int pjb_finish_group_end(pjb_ctx *c, const int32_t *tids, int32_t n_tids, pjb_region_result *results) {
    return end_flight(c, tids, n_tids, results, "finish_group");
}

int pjb_finish_contig(pjb_ctx *c, int32_t tid, pjb_region_result *res) {
    const int rc = pjb_finish_contig_begin(c, tid);
    if (rc) return rc;
    return pjb_finish_contig_end(c, tid, res);
}

int pjb_collect(pjb_ctx *c, const pjb_junction_row **rows, int64_t *n) {
    if (!c || !rows || !n) return PJB_ERR_ARG;
    const int rc = rows_sync(c);
    if (rc) return rc;
    *rows = c->rows_pinned;
    *n = (int64_t)c->rows_n;
    return PJB_OK;
}

int pjb_collect_device(pjb_ctx *c, const pjb_junction_row **rows, int64_t *n) {
    if (!c || !rows || !n) return PJB_ERR_ARG;
    *rows = (const pjb_junction_row *)c->sl[c->last_slot].rows.p;
    *n = (int64_t)c->last_rows_n;
    return PJB_OK;
}

int pjb_set_row_mirror(pjb_ctx *c, void *device_buffer, int64_t cap_bytes) {
    if (!c) return PJB_ERR_ARG;
    if (device_buffer && cap_bytes < PJB_MIRROR_HEADER_BYTES) return fail(c, PJB_ERR_ARG, "set_row_mirror: buffer smaller than its header");
    if (c->n_fl) return fail(c, PJB_ERR_STATE, "set_row_mirror: target %d is still queued", c->fl[0].tid);
    c->mirror = (uint8_t *)device_buffer;
    c->mirror_cap = device_buffer ? (size_t)cap_bytes : 0;
    mirror_reset(c);
    if (c->mirror && !c->mirror_hdr) {
        HIP_TRY(c, hipSetDevice(c->cfg.device));
        HIP_TRY(c, hipHostMalloc((void **)&c->mirror_hdr, PJB_MIRROR_HEADER_BYTES, hipHostMallocDefault));
    }
    return PJB_OK;
}

int pjb_plan_groups(const int32_t *ref_len, const int32_t *tids, int32_t n_tids, int64_t max_bases, int32_t *group_of) {
    if (n_tids < 0 || (n_tids > 0 && (!ref_len || !tids || !group_of))) return PJB_ERR_ARG;
    if (max_bases <= 0) max_bases = (int64_t)1 << 30;
    auto span = [&](int32_t k) { return (((int64_t)std::max(ref_len[tids[k]], 1) + GROUP_GAP) + 63) & ~(int64_t)63; };
    auto plan = [&](int64_t cap) {
        int32_t g = 0, members = 0;
        int64_t tot = 0;
        for (int32_t k = 0; k < n_tids; k++) {
            const int64_t s = span(k);
            if (members && (tot + s > cap || members >= GROUP_MAX)) g++, members = 0, tot = 0;
            group_of[k] = g;
            members++;
            tot += s;
        }
        return n_tids ? g + 1 : 0;
    };
    for (int32_t k = 0; k < n_tids; k++)
        if (tids[k] < 0) return PJB_ERR_ARG;
    int32_t n = plan(max_bases);
    if (n == 1 && n_tids > 1) { // ONE chain of more than 0.6 Gb: two, so that the first one's tail has the second one's K1 stage beside it
        int64_t total = 0;
        for (int32_t k = 0; k < n_tids; k++) total += span(k);
        if (total > 600000000) n = plan((int64_t)((double)total * 0.55));
    }
    return n;
}

int pjb_merge_rows(const void *gathered, int32_t n_ranks, int64_t slot_stride_bytes, pjb_junction_row *rows_out, int64_t cap_rows, int64_t *n_rows,
                   pjb_region_result *totals) {
    if (n_rows) *n_rows = 0;
    if (!gathered || n_ranks < 1 || slot_stride_bytes < PJB_MIRROR_HEADER_BYTES || !n_rows || !totals || cap_rows < 0 || (cap_rows > 0 && !rows_out))
        return PJB_ERR_ARG;
    struct Run { // rows of one target in one rank's slot
        int32_t refid;
        const pjb_junction_row *first;
        int64_t n;
    };
    std::vector<Run> runs;
    pjb_region_result T;
    memset(&T, 0, sizeof T);
    T.min_len = INT32_MAX;
    int64_t total = 0;
    for (int32_t r = 0; r < n_ranks; r++) {
        const uint8_t *slot = (const uint8_t *)gathered + (size_t)r * (size_t)slot_stride_bytes;
        int64_t h[6];
        memcpy(h, slot, sizeof h); // n_rows, spliced, unspliced, sum_len, min_len, max_len
        if (h[0] < 0 || h[0] > (slot_stride_bytes - PJB_MIRROR_HEADER_BYTES) / (int64_t)sizeof(pjb_junction_row)) return PJB_ERR_ARG;
        T.spliced += (uint64_t)h[1];
        T.unspliced += (uint64_t)h[2];
        T.sum_len += (uint64_t)h[3];
        T.min_len = std::min<int32_t>(T.min_len, (int32_t)std::min<int64_t>(h[4], INT32_MAX));
        T.max_len = std::max<int32_t>(T.max_len, (int32_t)h[5]);
        const pjb_junction_row *rows = (const pjb_junction_row *)(slot + PJB_MIRROR_HEADER_BYTES);
        for (int64_t i = 0; i < h[0];) {
            int64_t j = i + 1;
            while (j < h[0] && rows[j].refid == rows[i].refid) j++;
            runs.push_back(Run{rows[i].refid, rows + i, j - i});
            i = j;
        }
        total += h[0];
    }
    T.n_reads = (int64_t)(T.spliced + T.unspliced);
    T.n_junctions = total;
    *totals = T;
    *n_rows = total;
    if (total > cap_rows) return PJB_ERR_ARG;
    std::stable_sort(runs.begin(), runs.end(), [](const Run &a, const Run &b) { return a.refid < b.refid; });
    pjb_junction_row *out = rows_out;
    for (const Run &u : runs) {
        memcpy(out, u.first, (size_t)u.n * sizeof(pjb_junction_row));
        out += u.n;
    }
    return PJB_OK;
}

int pjb_clear_rows(pjb_ctx *c) {
    if (!c) return PJB_ERR_ARG;
    if (c->n_fl) return fail(c, PJB_ERR_STATE, "clear_rows: target %d is still queued", c->fl[0].tid);
    (void)rows_sync(c);
    c->rows_n = 0;
    mirror_reset(c);
    if (c->extra) {
        (void)hipSetDevice(c->cfg.device);
        (void)hipStreamSynchronize(c->stream);
        extra_clear(c);
    }
    return PJB_OK;
}

int pjb_extra_finish(pjb_ctx *c, const pjb_extra_row **rows_out, int64_t *n_out) {
    if (!c || !rows_out || !n_out) return PJB_ERR_ARG;
    const bool xtrace = getenv("PJB_XTRACE") != nullptr;
    auto xt0 = std::chrono::steady_clock::now();
    if (!c->extra) return fail(c, PJB_ERR_STATE, "pjb_extra_finish: the context was not created with PJB_FLAG_EXTRA");
    if (!c->open.empty()) return fail(c, PJB_ERR_STATE, "pjb_extra_finish: target %d is still open", c->open.begin()->first);
    if (c->n_fl) return fail(c, PJB_ERR_STATE, "pjb_extra_finish: target %d is still queued", c->fl[0].tid);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    int rc;
    const size_t Jall = c->rows_n;
    *rows_out = c->xrows_pinned;
    *n_out = (int64_t)Jall;
    if (Jall == 0) return PJB_OK;
    if (Jall > c->xrows_pinned_cap) {
        if (c->xrows_pinned) (void)hipHostFree(c->xrows_pinned);
        c->xrows_pinned = nullptr;
        c->xrows_pinned_cap = 0;
        const size_t cap = Jall + Jall / 4 + 1024;
        HIP_TRY(c, hipHostMalloc((void **)&c->xrows_pinned, cap * sizeof(pjb_extra_row), hipHostMallocDefault));
        c->xrows_pinned_cap = cap;
    }
    *rows_out = c->xrows_pinned;
    // every target's flanking counts into one table, parallel to the row table in HBM (which the kernels below read)
    if ((rc = ensure(c, c->x_xrall, Jall * (sizeof(ExtraRow) + sizeof(pjb_extra_row))))) return rc;
    ExtraRow *xr = (ExtraRow *)c->x_xrall.p;
    pjb_extra_row *xout = (pjb_extra_row *)(xr + Jall);
    HIP_TRY(c, hipMemsetAsync(xr, 0, Jall * sizeof(ExtraRow), st));
    for (auto &x : c->xc)
        if (x.n_rows) HIP_TRY(c, hipMemcpyAsync(xr + x.row_base, x.xr, x.n_rows * sizeof(ExtraRow), hipMemcpyDeviceToDevice, st));
    XTRACE("finish: memset + copies");
    // ---- splicedAlignmentMap over every spliced record of the file (src/junction_builder.cc:168-176): the targets' codes went
    // into the table as the targets were collected (a target of the dense path: now)
    for (auto &x : c->xc)
        if (!x.codes_in_table) {
            if ((rc = name_table_insert(c, (const u64 *)x.spl_codes, x.n_spl))) return rc;
            x.codes_in_table = true;
        }
    if (c->x_tab_n)
        for (auto &x : c->xc)
            if (x.n_pairs && x.n_rows)
                LAUNCH(c, "kx_name_sum", kx_name_sum, dim3((x.n_pairs + 1023) / 1024), dim3(256), (const u64 *)x.pair_code, (const u32 *)x.pair_row,
                       x.n_pairs, (const NameSlot *)c->x_tab.p, (u32)c->x_tab_slots, xr);
    // ---- JunctionSystem::calcCoverage (lib/src/junction_system.cc:231-242).  DepthParser::loadNextBatch
    // (lib/src/depth_parser.cc:112-164) returns the vector of the target it started in, but by then `last`
    // names the target the pileup has moved on to, and getCurrentRefIndex() selects THAT target's junctions:
    // every batch is applied to the junctions of the next target that has unspliced records; only the final
    // batch (the pileup ended inside it) meets its own junctions, after they were first given the previous
    // target's.  Targets without unspliced records never appear.
    XTRACE("finish: name sums");
    std::vector<const ExtraContig *> T;
    for (auto &x : c->xc)
        if (x.has_unspliced) T.push_back(&x);
    std::sort(T.begin(), T.end(), [](const ExtraContig *a, const ExtraContig *b) { return a->tid < b->tid; });
    const pjb_junction_row *rows = c->rows_table;
    for (size_t k = 0; k < T.size(); k++) {
        const ExtraContig &x = *T[k];
        if (!x.n_rows) continue;
        const ExtraContig *src = (k + 1 == T.size()) ? &x : (k > 0 ? T[k - 1] : nullptr);
        if (!src) continue; // the first target's junctions are never visited (unless it is also the last)
        if (src->dense)
            LAUNCH(c, "kx_coverage", kx_coverage, dim3((unsigned)((x.n_rows + 255) / 256)), dim3(256), rows, (u32)x.row_base, (u32)x.n_rows,
                   (const u32 *)src->cover, src->len, xr);
        else
            LAUNCH(c, "kx_coverage_sparse", kx_coverage_sparse, dim3((unsigned)((x.n_rows + 255) / 256)), dim3(256), rows, (u32)x.row_base,
                   (u32)x.n_rows, src->sparse, src->len, xr);
    }
    XTRACE("finish: coverage");
    LAUNCH(c, "kx_rows_out", kx_rows_out, dim3((unsigned)((Jall + 255) / 256)), dim3(256), rows, (const ExtraRow *)xr, (u32)Jall, xout);
    XTRACE("finish: rows_out");
    HIP_TRY(c, hipMemcpyAsync(c->xrows_pinned, xout, Jall * sizeof(pjb_extra_row), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    XTRACE("finish: D2H");
    if (c->ktime) ev_collect(c, MISC_POOL);
    return PJB_OK;
}

int pjb_filter_set_junctions(pjb_ctx *c, int32_t tid, const uint64_t *sorted_keys, int64_t n_keys) {
    if (!c) return PJB_ERR_ARG;
    if (tid < 0 || n_keys < 0 || n_keys > 0xfffffff0ll || (n_keys > 0 && !sorted_keys))
        return fail(c, PJB_ERR_ARG, "pjb_filter_set_junctions: bad arguments (tid %d)", tid);
    for (int64_t i = 1; i < n_keys; i++)
        if (sorted_keys[i - 1] >= sorted_keys[i]) return fail(c, PJB_ERR_ARG, "pjb_filter_set_junctions: keys must be strictly ascending");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    auto it = c->filter_keys.find(tid);
    if (it != c->filter_keys.end()) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (it->second.first) (void)hipFree(it->second.first);
        c->filter_keys.erase(it);
    }
    u64 *d = nullptr;
    if (n_keys) {
        if (hipMalloc((void **)&d, (size_t)n_keys * 8) != hipSuccess) return fail(c, PJB_ERR_NOMEM, "pjb_filter_set_junctions: %lld keys", (long long)n_keys);
        hipError_t e = hipMemcpy(d, sorted_keys, (size_t)n_keys * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(d);
            return fail(c, PJB_ERR_HIP, "pjb_filter_set_junctions: %s", hipGetErrorString(e));
        }
    }
    c->filter_keys[tid] = std::make_pair(d, (u32)n_keys);
    return PJB_OK;
}

int pjb_filter_batch(pjb_ctx *c, int32_t tid, const pjb_batch *b, int32_t clip_mode, uint8_t *codes_out) {
    if (!c) return PJB_ERR_ARG;
    if (!b || b->n_reads < 0 || (b->n_reads > 0 && (!b->pos || !b->cig_off || !b->cigar || !codes_out)))
        return fail(c, PJB_ERR_ARG, "pjb_filter_batch: bad batch");
    if (clip_mode < PJB_CLIP_HARD || clip_mode > PJB_CLIP_COMPLETE) return fail(c, PJB_ERR_ARG, "pjb_filter_batch: bad clip mode %d", clip_mode);
    if (b->n_reads == 0) return PJB_OK;
    if (b->n_reads > 0xfffffff0ll) return fail(c, PJB_ERR_ARG, "pjb_filter_batch: batch too large");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    const size_t n = (size_t)b->n_reads, n_ops = b->cig_off[n];
    int rc;
    if ((rc = ensure(c, c->f_pos, n * 4))) return rc;
    if ((rc = ensure(c, c->f_cigoff, (n + 1) * 4))) return rc;
    if ((rc = ensure(c, c->f_cigar, n_ops * 4 + 16))) return rc;
    if ((rc = ensure(c, c->f_codes, n + 16))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->f_pos.p, b->pos, n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->f_cigoff.p, b->cig_off, (n + 1) * 4, hipMemcpyHostToDevice, st));
    if (n_ops) HIP_TRY(c, hipMemcpyAsync(c->f_cigar.p, b->cigar, n_ops * 4, hipMemcpyHostToDevice, st));
    const u64 *keys = nullptr;
    u32 n_keys = 0;
    auto it = c->filter_keys.find(tid);
    if (it != c->filter_keys.end()) {
        keys = it->second.first;
        n_keys = it->second.second;
    }
    LAUNCH(c, "kf_filter", kf_filter, dim3((unsigned)((n + 255) / 256)), dim3(256), (const int32_t *)c->f_pos.p, (const u32 *)c->f_cigoff.p,
           (const u32 *)c->f_cigar.p, (u32)n, keys, n_keys, (int)clip_mode, (uint8_t *)c->f_codes.p);
    HIP_TRY(c, hipMemcpyAsync(codes_out, c->f_codes.p, n, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    if (c->ktime) ev_collect(c, MISC_POOL);
    return PJB_OK;
}

int pjb_filt_features(pjb_ctx *c, const pjb_junction_row *rows, int64_t n_rows, double mean_read_length, uint32_t l95,
                      const pjb_markov_models *models, double *features_out) {
    if (!c) return PJB_ERR_ARG;
    if (n_rows < 0 || (n_rows > 0 && (!rows || !features_out)) || !models || n_rows > 0xfffffff0ll)
        return fail(c, PJB_ERR_ARG, "pjb_filt_features: bad arguments");
    if (n_rows == 0) return PJB_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    const size_t n = (size_t)n_rows;
    int rc;
    if ((rc = ensure(c, c->g_rows, n * sizeof(pjb_junction_row)))) return rc;
    if ((rc = ensure(c, c->g_models, ((size_t)6 * PJB_KMER_TABLE + 2 * PJB_PW_LEN * 5) * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->g_refs, std::max<size_t>(c->contigs.size(), 1) * sizeof(GenomeRef)))) return rc;
    if ((rc = ensure(c, c->g_out, n * PJB_N_FEATURES * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->g_bad, sizeof(int)))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->g_rows.p, rows, n * sizeof(pjb_junction_row), hipMemcpyHostToDevice, st));
    DevModels M;
    memset(&M, 0, sizeof M);
    double *dm = (double *)c->g_models.p;
    const double *src[8] = {models->exon, models->intron, models->donor_t, models->donor_f, models->acceptor_t, models->acceptor_f,
                            models->donor_pw, models->acceptor_pw};
    const double **dst[8] = {&M.exon, &M.intron, &M.don_t, &M.don_f, &M.acc_t, &M.acc_f, &M.don_pw, &M.acc_pw};
    size_t at = 0;
    for (int k = 0; k < 8; k++) {
        const size_t cnt = k < 6 ? (size_t)PJB_KMER_TABLE : (size_t)PJB_PW_LEN * 5;
        if (src[k]) {
            HIP_TRY(c, hipMemcpyAsync(dm + at, src[k], cnt * sizeof(double), hipMemcpyHostToDevice, st));
            *dst[k] = dm + at;
        }
        at += cnt;
    }
    M.exon_size = models->exon ? models->exon_size : 0;
    M.intron_size = models->intron ? models->intron_size : 0;
    M.don_pw_size = models->donor_pw ? models->donor_pw_size : 0;
    M.acc_pw_size = models->acceptor_pw ? models->acceptor_pw_size : 0;
    std::vector<GenomeRef> refs(std::max<size_t>(c->contigs.size(), 1));
    for (size_t t = 0; t < c->contigs.size(); t++) {
        refs[t].d = c->contigs[t].present ? c->contigs[t].d : nullptr;
        refs[t].len = (int32_t)c->contigs[t].len;
    }
    HIP_TRY(c, hipMemcpyAsync(c->g_refs.p, refs.data(), refs.size() * sizeof(GenomeRef), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemsetAsync(c->g_bad.p, 0, sizeof(int), st));
    LAUNCH(c, "kg_features", kg_features, dim3((unsigned)((n + 255) / 256)), dim3(256), (const pjb_junction_row *)c->g_rows.p, (u32)n,
           (const GenomeRef *)c->g_refs.p, (int)c->contigs.size(), M, mean_read_length, (u32)l95, (double *)c->g_out.p, (int *)c->g_bad.p);
    int bad = 0;
    HIP_TRY(c, hipMemcpyAsync(features_out, c->g_out.p, n * PJB_N_FEATURES * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(&bad, c->g_bad.p, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    if (c->ktime) ev_collect(c, MISC_POOL);
    if (bad) return fail(c, PJB_ERR_STATE, "pjb_filt_features: a junction lies on a target whose genome was not uploaded");
    return PJB_OK;
}

int pjb_get_kernel_timing(const pjb_ctx *c, pjb_kernel_time *out, int32_t cap, int32_t *n) {
    if (!c || !n) return PJB_ERR_ARG;
    *n = (int32_t)c->knames.size();
    for (int32_t i = 0; out && i < cap && i < *n; i++) {
        memset(&out[i], 0, sizeof out[i]);
        strncpy(out[i].name, c->knames[(size_t)i].c_str(), sizeof(out[i].name) - 1);
        out[i].launches = c->kcount[(size_t)i];
        out[i].total_ms = c->kms[(size_t)i];
    }
    return PJB_OK;
}

int pjb_select_timed_kernels(pjb_ctx *c, const char *comma_separated_names) {
    if (!c) return PJB_ERR_ARG;
    c->ktime_only.clear();
    std::string s = comma_separated_names ? comma_separated_names : "";
    size_t a = 0;
    while (a < s.size()) {
        size_t b = s.find(',', a);
        if (b == std::string::npos) b = s.size();
        if (b > a) c->ktime_only.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    return PJB_OK;
}

int pjb_reset_kernel_timing(pjb_ctx *c) {
    if (!c) return PJB_ERR_ARG;
    std::fill(c->kcount.begin(), c->kcount.end(), 0);
    std::fill(c->kms.begin(), c->kms.end(), 0.0);
    return PJB_OK;
}

int pjb_set_option(pjb_ctx *c, const char *name, int64_t value) {
    if (!c || !name) return PJB_ERR_ARG;
    if (c->n_fl) return fail(c, PJB_ERR_STATE, "set_option: target %d is still queued", c->fl[0].tid);
    const std::string n = name;
    if (n == "overlap") c->side_stream = value != 0;
    else if (n == "dense_ids") c->dense_ids = value != 0;
    else if (n == "extra_dense") c->extra_dense_only = value != 0;
    else if (n == "sort_floor") c->sort_floor = (u32)std::max<int64_t>(1, std::min<int64_t>(value, 1 << 30));
    else if (n == "list_cap") c->list_cap_forced = (u32)std::max<int64_t>(0, std::min<int64_t>(value, 1 << 30));
    else return fail(c, PJB_ERR_ARG, "set_option: unknown option '%s'", name);
    return PJB_OK;
}

int pjb_get_timing(const pjb_ctx *c, pjb_timing *out) {
    if (!c || !out) return PJB_ERR_ARG;
    if (c->cfg.abi_version >= 4) *out = c->timing;
    else memcpy(out, &c->timing, offsetof(pjb_timing, repeats)); // (ABI 3's pjb_timing ended at checked_reads)
    return PJB_OK;
}

} // extern "C"

// ---- device-side ingest ---------------------------------------------------------------------------
namespace {
const char *inf_text(int code) {
    switch (code) {
    case INF_ERR_BTYPE: return "reserved DEFLATE block type";
    case INF_ERR_STORED: return "stored block length check failed";
    case INF_ERR_CODELENS: return "invalid code length set";
    case INF_ERR_CODE: return "invalid Huffman code";
    case INF_ERR_DIST: return "match distance before the start of the block";
    case INF_ERR_OVERRUN: return "block inflates or reads past its declared size";
    case INF_ERR_SIZE: return "block inflates to fewer bytes than its ISIZE";
    default: return "bad block";
    }
}

// hop over the BGZF block headers (bgzf.c:348-356 check_header, BSIZE from the BC extra subfield)
int scan_bgzf(pjb_ctx *c, const uint8_t *comp, int64_t n, std::vector<InfBlock> &blocks, int64_t &total_out) {
    int64_t off = 0;
    total_out = 0;
    while (off < n) {
        if (off + 18 > n) return fail(c, PJB_ERR_BGZF, "truncated BGZF block header at byte %lld", (long long)off);
        const uint8_t *h = comp + off;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4))
            return fail(c, PJB_ERR_BGZF, "not a BGZF block header at byte %lld", (long long)off);
        const uint32_t xlen = h[10] | (uint32_t)h[11] << 8;
        if (off + 12 + xlen > n) return fail(c, PJB_ERR_BGZF, "truncated BGZF extra field at byte %lld", (long long)off);
        int64_t bsize = -1;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t *f = h + 12 + x;
            const uint32_t slen = f[2] | (uint32_t)f[3] << 8;
            if (f[0] == 'B' && f[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = (int64_t)(f[4] | (uint32_t)f[5] << 8) + 1;
            x += 4 + slen;
        }
        if (bsize < 0) return fail(c, PJB_ERR_BGZF, "BGZF block at byte %lld has no BC field", (long long)off);
        if (bsize < (int64_t)xlen + 20 || off + bsize > n)
            return fail(c, PJB_ERR_BGZF, "BGZF block at byte %lld has an impossible size %lld", (long long)off, (long long)bsize);
        const uint8_t *foot = comp + off + bsize - 8;
        const uint32_t isize = foot[4] | (uint32_t)foot[5] << 8 | (uint32_t)foot[6] << 16 | (uint32_t)foot[7] << 24;
        if (isize > 65536u) return fail(c, PJB_ERR_BGZF, "BGZF block at byte %lld declares %u inflated bytes", (long long)off, isize);
        InfBlock b;
        b.in_off = (iu64)(off + 12 + xlen);
        b.in_len = (iu32)(bsize - xlen - 20);
        b.out_off = (iu64)total_out;
        b.out_len = isize;
        blocks.push_back(b);
        total_out += isize;
        off += bsize;
    }
    return PJB_OK;
}

// pageable host memory -> device through the two page-locked staging buffers: a few threads memcpy a
// piece into one buffer while the DMA engine drains the other
int upload_staged(pjb_ctx *c, void *dst, const uint8_t *src, size_t bytes) {
    const size_t PIECE = (size_t)64 << 20;
    for (size_t off = 0; off < bytes; off += PIECE) {
        const size_t nb = std::min(PIECE, bytes - off);
        const unsigned si = c->stage_next++ & 1u;
        if (c->stage_busy[si]) {
            HIP_TRY(c, hipEventSynchronize(c->stage_ev[si]));
            c->stage_busy[si] = false;
        }
        if (c->stage_cap[si] < nb) {
            if (c->stage[si]) (void)hipHostFree(c->stage[si]);
            c->stage[si] = nullptr;
            c->stage_cap[si] = 0;
            if (hipHostMalloc((void **)&c->stage[si], PIECE, hipHostMallocDefault) != hipSuccess)
                return fail(c, PJB_ERR_NOMEM, "cannot allocate %zu bytes of page-locked staging memory", PIECE);
            c->stage_cap[si] = PIECE;
        }
        if (!c->stage_ev[si]) HIP_TRY(c, hipEventCreateWithFlags(&c->stage_ev[si], hipEventDisableTiming));
        parallel_copy(c->stage[si], src + off, nb);
        HIP_TRY(c, hipMemcpyAsync((uint8_t *)dst + off, c->stage[si], nb, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipEventRecord(c->stage_ev[si], c->stream));
        c->stage_busy[si] = true;
    }
    return PJB_OK;
}


// comp already on the device (padded); blocks on the host
// the status words of a finished bgzf_inflate (d_status[nb] = "some block failed")
int inflate_status(pjb_ctx *c, const std::vector<InfBlock> &blocks, const int *d_status) {
    const size_t nb = blocks.size();
    int any = 0;
    HIP_TRY(c, hipMemcpy(&any, d_status + nb, 4, hipMemcpyDeviceToHost));
    if (!any) return PJB_OK;
    std::vector<int> status(nb);
    HIP_TRY(c, hipMemcpy(status.data(), d_status, nb * 4, hipMemcpyDeviceToHost));
    for (size_t b = 0; b < nb; b++)
        if (status[b])
            return fail(c, PJB_ERR_BGZF, "BGZF block %zu (payload at byte %llu): %s", b, (unsigned long long)blocks[b].in_off, inf_text(status[b]));
    return fail(c, PJB_ERR_BGZF, "BGZF inflate failed");
}

int inflate_on_device(pjb_ctx *c, const uint8_t *d_comp, const std::vector<InfBlock> &blocks, uint8_t *d_out) {
    int rc;
    const size_t nb = blocks.size();
    if (nb == 0) return PJB_OK;
    if ((rc = ensure(c, c->b_inf_blocks, nb * sizeof(InfBlock)))) return rc;
    if ((rc = ensure(c, c->b_inf_status, nb * 4 + 16))) return rc;
    // one launch: as many lanes as the chip holds at once (two 64-lane workgroups per CU: the tables' LDS), each taking
    // block after block from a counter
    size_t lanes = std::min<size_t>((nb + 63) / 64 * 64, (size_t)c->inflate_lanes);
    if (const char *e = getenv("PJB_INF_BLOCKS_PER_LAUNCH")) lanes = std::min<size_t>((nb + 63) / 64 * 64, (size_t)std::max(64, atoi(e)) / 64 * 64); // tests: few lanes, long lists
    if ((rc = ensure(c, c->b_inf_scratch, lanes * INF_SCRATCH_PER_LANE))) return rc;
    hipStream_t st = c->stream;
    HIP_TRY(c, hipMemcpyAsync(c->b_inf_blocks.p, blocks.data(), nb * sizeof(InfBlock), hipMemcpyHostToDevice, st));
    int *d_status = (int *)c->b_inf_status.p;
    int *d_any = d_status + nb;
    iu32 *d_next = (iu32 *)(d_any + 1);
    const iu32 ctl[2] = {0u, (iu32)lanes};
    HIP_TRY(c, hipMemcpyAsync(d_any, ctl, 8, hipMemcpyHostToDevice, st));
    {
        // decode (lane per block: literals in place, a token + a bitmap bit per match), then the copies (wave per block)
        if ((rc = ensure(c, c->b_inf_bitmap, nb * INF_BITMAP_WORDS * 8))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->b_inf_bitmap.p, 0, nb * INF_BITMAP_WORDS * 8, st));
        LAUNCH_LDS(c, "bgzf_decode", bgzf_decode, dim3((unsigned)(lanes / 64)), dim3(64), I3_LDS_BYTES, d_comp, (const InfBlock *)c->b_inf_blocks.p, (iu32)nb,
                   d_out, (uint8_t *)c->b_inf_scratch.p, d_status, d_any, d_next, (iu64 *)c->b_inf_bitmap.p, 8);
        LAUNCH(c, "bgzf_resolve", bgzf_resolve, dim3((unsigned)((nb + 3) / 4)), dim3(256), (const InfBlock *)c->b_inf_blocks.p, (iu32)nb, d_out,
               (const iu64 *)c->b_inf_bitmap.p, (const int *)d_status);
    }
    HIP_TRY(c, hipStreamSynchronize(st));
    if (c->ktime) ev_collect(c, MISC_POOL);
    return inflate_status(c, blocks, d_status);
}
} // namespace

extern "C" int pjb_inflate_bgzf(pjb_ctx *c, const uint8_t *comp, int64_t comp_bytes, uint8_t *out, int64_t out_cap,
                                int64_t *out_bytes) {
    if (!c || !out_bytes || comp_bytes < 0 || (comp_bytes && !comp)) return fail(c, PJB_ERR_ARG, "inflate_bgzf: bad arguments");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    std::vector<InfBlock> blocks;
    int64_t total = 0;
    int rc = scan_bgzf(c, comp, comp_bytes, blocks, total);
    if (rc) return rc;
    *out_bytes = total;
    if (total > out_cap) return fail(c, PJB_ERR_ARG, "inflate_bgzf: output needs %lld bytes, capacity is %lld", (long long)total, (long long)out_cap);
    if (total == 0) return PJB_OK;
    if (!out) return fail(c, PJB_ERR_ARG, "inflate_bgzf: no output buffer");
    if ((rc = ensure(c, c->b_inf_comp, (size_t)comp_bytes + INF_PAD))) return rc;
    if ((rc = ensure(c, c->b_inf_out, (size_t)total + 64))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->b_inf_comp.p, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync((uint8_t *)c->b_inf_comp.p + comp_bytes, 0, INF_PAD, c->stream));
    if ((rc = inflate_on_device(c, (const uint8_t *)c->b_inf_comp.p, blocks, (uint8_t *)c->b_inf_out.p))) return rc;
    HIP_TRY(c, hipMemcpy(out, c->b_inf_out.p, (size_t)total, hipMemcpyDeviceToHost));
    return PJB_OK;
}

// BGZF deflate on the device (pjb_deflate.hip.h): at most DFL_LAUNCH_BLOCKS blocks per launch (1 GB of symbol scratch)
constexpr int64_t DFL_LAUNCH_BLOCKS = 4096;
extern "C" int pjb_deflate_bgzf(pjb_ctx *c, const uint8_t *in, int64_t n_bytes, int32_t block_bytes, uint8_t *out, int64_t out_cap, int64_t *out_bytes,
                                uint32_t *member_size) {
    if (!c || !out_bytes || n_bytes < 0 || (n_bytes && (!in || !out))) return fail(c, PJB_ERR_ARG, "deflate_bgzf: bad arguments");
    if (block_bytes < 4 || block_bytes > (int32_t)DFL_IN_MAX || (block_bytes & 3))
        return fail(c, PJB_ERR_ARG, "deflate_bgzf: block_bytes must be a multiple of 4 between 4 and %u", DFL_IN_MAX);
    *out_bytes = 0;
    if (n_bytes == 0) return PJB_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t st = c->stream;
    const int64_t n_blocks = (n_bytes + block_bytes - 1) / block_bytes;
    int rc;
    std::vector<u32> sizes;
    std::vector<iu64> offs;
    int64_t written = 0;
    for (int64_t b0 = 0; b0 < n_blocks; b0 += DFL_LAUNCH_BLOCKS) {
        const int64_t nb = std::min<int64_t>(DFL_LAUNCH_BLOCKS, n_blocks - b0);
        const int64_t in_off = b0 * block_bytes, in_len = std::min<int64_t>(n_bytes - in_off, nb * block_bytes);
        if ((rc = ensure(c, c->b_dfl_in, (size_t)in_len + 64)) || (rc = ensure(c, c->b_dfl_sym, (size_t)nb * DFL_SYM_STRIDE * 4)) ||
            (rc = ensure(c, c->b_dfl_slots, (size_t)nb * DFL_SLOT)) || (rc = ensure(c, c->b_dfl_size, (size_t)nb * 4)) ||
            (rc = ensure(c, c->b_dfl_off, (size_t)nb * 8)))
            return rc;
        HIP_TRY(c, hipMemcpyAsync(c->b_dfl_in.p, in + in_off, (size_t)in_len, hipMemcpyHostToDevice, st));
        LAUNCH(c, "bgzf_deflate", bgzf_deflate, dim3((unsigned)nb), dim3(64), (const uint8_t *)c->b_dfl_in.p, (iu64)in_len, (u32)block_bytes, (u32)nb,
               (u32 *)c->b_dfl_sym.p, (uint8_t *)c->b_dfl_slots.p, (u32 *)c->b_dfl_size.p);
        sizes.resize((size_t)nb);
        HIP_TRY(c, hipMemcpyAsync(sizes.data(), c->b_dfl_size.p, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        offs.resize((size_t)nb);
        iu64 total = 0;
        for (int64_t k = 0; k < nb; k++) {
            if (sizes[(size_t)k] < 26 || sizes[(size_t)k] > 65536) return fail(c, PJB_ERR_STATE, "deflate_bgzf: block %lld came out with %u bytes", (long long)(b0 + k), sizes[(size_t)k]);
            offs[(size_t)k] = total;
            total += sizes[(size_t)k];
            if (member_size) member_size[b0 + k] = sizes[(size_t)k];
        }
        if (written + (int64_t)total > out_cap)
            return fail(c, PJB_ERR_ARG, "deflate_bgzf: the output needs more than %lld bytes", (long long)out_cap);
        if ((rc = ensure(c, c->b_dfl_packed, (size_t)total + 64))) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->b_dfl_off.p, offs.data(), (size_t)nb * 8, hipMemcpyHostToDevice, st));
        LAUNCH(c, "bgzf_pack", bgzf_pack, dim3((unsigned)nb), dim3(256), (const uint8_t *)c->b_dfl_slots.p, (const u32 *)c->b_dfl_size.p,
               (const iu64 *)c->b_dfl_off.p, (u32)nb, (uint8_t *)c->b_dfl_packed.p);
        HIP_TRY(c, hipMemcpyAsync(out + written, c->b_dfl_packed.p, (size_t)total, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        written += (int64_t)total;
    }
    *out_bytes = written;
    if (c->ktime) ev_collect(c, MISC_POOL);
    return PJB_OK;
}

static int ingest_parse(pjb_ctx *c, int32_t tid, OpenContig &oc, const uint8_t *d_out, size_t n_blocks, int64_t comp_bytes, int64_t total,
                        int32_t first_uoffset, int64_t *n_records, double t_scan, double t_up, double t_inf);

// the part of pjb_submit_bam behind the upload: `d_comp` holds the target's BGZF bytes (padded), `blocks` their layout
static int ingest_staged(pjb_ctx *c, int32_t tid, OpenContig &oc, const uint8_t *d_comp, const std::vector<InfBlock> &blocks, int64_t comp_bytes,
                         int64_t total, int32_t first_uoffset, int64_t *n_records, double t_scan, double t_up) {
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    int rc;
    if ((rc = ensure(c, c->b_inf_out, (size_t)total + 64))) return rc;
    HIP_TRY(c, hipMemsetAsync((uint8_t *)c->b_inf_out.p + total, 0, 64, c->stream));
    if ((rc = inflate_on_device(c, d_comp, blocks, (uint8_t *)c->b_inf_out.p))) return rc;
    return ingest_parse(c, tid, oc, (const uint8_t *)c->b_inf_out.p, blocks.size(), comp_bytes, total, first_uoffset, n_records, t_scan, t_up, now() - t0);
}

// the inflated bytes of one target's region (d_out, `total` of them followed by 64 zero bytes) -> the SoA batch of the target
static int ingest_parse(pjb_ctx *c, int32_t tid, OpenContig &oc, const uint8_t *d_out, size_t n_blocks, int64_t comp_bytes, int64_t total,
                        int32_t first_uoffset, int64_t *n_records, double t_scan, double t_up, double t_inf) {
    const bool prof = getenv("PJB_PROFILE_HOST") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now(), t_walk;
    hipStream_t st = c->stream;
    int rc;
    // ---- record boundaries
    BamRegion R;
    R.U = d_out;
    R.total = (iu64)total;
    R.first = (iu64)first_uoffset;
    R.tid = tid;
    R.ref_len = c->ref_len[(size_t)tid];
    R.n_ref = (int32_t)c->ref_len.size();
    const uint32_t n_seg = (uint32_t)(((iu64)total + BAM_SEG - 1) / BAM_SEG);
    // seg_start u64 | seg_base u64 | land u64 | seg_n u32
    if ((rc = ensure(c, c->b_bam_seg, (size_t)n_seg * 28 + 64))) return rc;
    if ((rc = ensure(c, c->b_bam_ctl, 64))) return rc;
    iu64 *seg_start = (iu64 *)c->b_bam_seg.p;
    iu64 *seg_base = seg_start + n_seg;
    iu64 *land = seg_base + n_seg;
    iu32 *seg_n = (iu32 *)(land + n_seg);
    iu32 *ctl = (iu32 *)c->b_bam_ctl.p; // [0..2] end / mismatch / bad segment, [4..5] u64 total of a scan
    iu64 *d_total = (iu64 *)(ctl + 4);
    HIP_TRY(c, hipMemsetAsync(ctl, 0xff, 16, st));
    BamWalkOut O;
    O.seg_n = seg_n;
    O.land = land;
    O.rec_off = nullptr;
    O.seg_base = seg_base;
    O.ctl = ctl;
    LAUNCH(c, "bam_find_starts", bam_find_starts, dim3(n_seg), dim3(64), R, n_seg, seg_start);
    if (const char *e = getenv("PJB_TEST_FALSE_START")) { // test hook: damage the guessed start of one segment
        const uint32_t k = (uint32_t)atoi(e);
        if (k > 0 && k < n_seg) {
            iu64 v = 0;
            HIP_TRY(c, hipMemcpyAsync(&v, seg_start + k, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipStreamSynchronize(st));
            if (v != BAM_NONE) {
                v += 1;
                HIP_TRY(c, hipMemcpyAsync(seg_start + k, &v, 8, hipMemcpyHostToDevice, st));
                HIP_TRY(c, hipStreamSynchronize(st));
            }
        }
    }
    uint32_t h_ctl[8];
    uint32_t end_seg = 0xffffffffu;
    for (int attempt = 0;; attempt++) {
        HIP_TRY(c, hipMemsetAsync(ctl, 0xff, 16, st));
        HIP_TRY(c, hipMemsetAsync(ctl + 6, 0xff, 4, st));
        LAUNCH(c, "bam_walk_count", bam_walk<false>, dim3((n_seg + 255) / 256), dim3(256), R, n_seg, (const iu64 *)seg_start, O);
        HIP_TRY(c, hipMemcpyAsync(h_ctl, ctl, 32, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        end_seg = h_ctl[0];
        if (h_ctl[2] != 0xffffffffu && h_ctl[2] <= end_seg && (h_ctl[1] == 0xffffffffu || h_ctl[2] <= h_ctl[1]))
            return fail(c, PJB_ERR_BGZF, "Invalid BAM record layout on target %d (inflated offset %llu..)", tid,
                        (unsigned long long)h_ctl[2] * BAM_SEG);
        if (h_ctl[1] == 0xffffffffu || h_ctl[1] > end_seg) break; // every walk landed on the next start
        // a guessed start was not a record boundary: replace it by the boundary the verified walk reached, walk again
        if (attempt >= 16)
            return fail(c, PJB_ERR_BGZF, "BAM record chain of target %d is inconsistent near inflated offset %llu", tid,
                        (unsigned long long)h_ctl[1] * BAM_SEG);
        LAUNCH(c, "bam_repair_start", bam_repair_start, dim3(1), dim3(1), seg_start, n_seg, h_ctl[1], (const iu64 *)land, (iu64)total, ctl);
        HIP_TRY(c, hipMemcpyAsync(h_ctl, ctl, 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if (h_ctl[3] != 0xffffffffu)
            return fail(c, PJB_ERR_BGZF, "Invalid BAM record on target %d (inflated offset %llu..)", tid, (unsigned long long)h_ctl[3] * BAM_SEG);
    }
    // The data ends inside a record of this target and no record of another target (or past the target's end) was seen:
    // the bytes handed over stop short of the target's last alignment (a stale index, a truncated file).  The reference
    // fails on a truncated file too (bgzf_read / bam_read1); dropping the tail silently would change counts.
    if (end_seg == 0xffffffffu && h_ctl[6] != 0xffffffffu)
        return fail(c, PJB_ERR_BGZF, "the data for target %d ends inside an alignment record (inflated offset %llu..): truncated "
                                     "file, or the index's span for the target is too short", tid, (unsigned long long)h_ctl[6] * BAM_SEG);
    LAUNCH(c, "bam_trim_segments", bam_trim_segments, dim3((n_seg + 255) / 256), dim3(256), seg_n, n_seg, (const iu32 *)ctl);
    if ((rc = run_scan(c, "bam_seg", SegCountFn{seg_n}, SegBaseSink{seg_base}, n_seg, d_total))) return rc;
    HIP_TRY(c, hipMemcpyAsync(h_ctl, ctl, 24, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    t_walk = now() - t0;
    t0 = now();
    iu64 n64;
    memcpy(&n64, &h_ctl[4], 8);
    if (n64 == 0) return PJB_OK;
    if (n64 >= 0xffffff00ull) return fail(c, PJB_ERR_ARG, "submit_bam: more than 2^32 alignments on one target are not supported");
    const size_t n = (size_t)n64;
    if ((rc = ensure(c, c->b_bam_rec, n * 8))) return rc;
    O.rec_off = (iu64 *)c->b_bam_rec.p;
    LAUNCH(c, "bam_walk_fill", bam_walk<true>, dim3((n_seg + 255) / 256), dim3(256), R, n_seg, (const iu64 *)seg_start, O);

    // ---- SoA arrays in the target's slabs (same packing as a host-submitted batch)
    // (150-base paired-end records: fields + operations + 4- and 2-bit bases of the spliced third are 0.28 of the inflated bytes)
    if (oc.slabs.empty()) oc.slab_hint = (((size_t)total / 100 * 32) + ((size_t)32 << 20)) & ~(((size_t)1 << 20) - 1);
    const size_t fixed[8] = {n * 4, n * 2, n, n, n * 4, n * 4, n * 4, (n + 1) * 4}; // pos flag mapq xs l_qseq mtid mpos cig_off
    size_t offs[9], tot_b = 0;
    for (int k = 0; k < 8; k++) {
        offs[k] = tot_b;
        tot_b += (std::max<size_t>(fixed[k], 16) + 255) & ~(size_t)255;
    }
    offs[8] = tot_b; // seq_off
    tot_b += (((n + 1) * 4) + 255) & ~(size_t)255;
    uint8_t *dev = (uint8_t *)slab_alloc(c, oc, tot_b);
    if (!dev) return fail(c, PJB_ERR_NOMEM, "submit_bam: out of device memory for %zu alignments", n);
    BamSoA B;
    B.pos = (int32_t *)(dev + offs[0]);
    B.flag = (uint16_t *)(dev + offs[1]);
    B.mapq = dev + offs[2];
    B.xs = dev + offs[3];
    B.l_qseq = (int32_t *)(dev + offs[4]);
    B.mtid = (int32_t *)(dev + offs[5]);
    B.mpos = (int32_t *)(dev + offs[6]);
    B.cig_off = (iu32 *)(dev + offs[7]);
    B.seq_off = (iu32 *)(dev + offs[8]);
    B.cigar = nullptr;
    B.seq4 = nullptr;
    B.name_hash = nullptr;
    B.seq2 = nullptr;
    B.seq_exc = nullptr;
    if (c->extra) {
        B.name_hash = (iu64 *)slab_alloc(c, oc, n * 8 + 16);
        if (!B.name_hash) return fail(c, PJB_ERR_NOMEM, "submit_bam: out of device memory for name codes");
    }
    if ((rc = run_scan(c, "bam_sizes", BamSizesFn{R.U, (const iu64 *)c->b_bam_rec.p}, BamOffsetsSink{B.cig_off, B.seq_off}, n, d_total)))
        return rc;
    iu64 tot = 0;
    HIP_TRY(c, hipMemcpyAsync(&tot, d_total, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    const uint32_t n_ops = (uint32_t)(tot >> 32), n_words = (uint32_t)tot;
    // (a carry out of the low half would mean 2^32 sequence words: 16 GB of bases on one target)
    const uint32_t tails[2] = {n_ops, n_words};
    HIP_TRY(c, hipMemcpyAsync(B.cig_off + n, &tails[0], 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(B.seq_off + n, &tails[1], 4, hipMemcpyHostToDevice, st));
    B.cigar = (iu32 *)slab_alloc(c, oc, (size_t)n_ops * 4 + 16);
    B.seq4 = (uint8_t *)slab_alloc(c, oc, (size_t)n_words * 4 + 16);
    if (!B.cigar || !B.seq4) return fail(c, PJB_ERR_NOMEM, "submit_bam: out of device memory for CIGARs / bases");
    static const bool no_seq2 = getenv("PJB_NO_SEQ2") && atoi(getenv("PJB_NO_SEQ2")) != 0;
    if (!no_seq2) { // the bases in 2 bits as well (what pjb_batch.seq2 / .seq_exc hold): written where the 4-bit bases are
        B.seq2 = (unsigned short *)slab_alloc(c, oc, ((size_t)n_words + 2) * 2 + 16);
        B.seq_exc = (iu32 *)slab_alloc(c, oc, ((n + 31) / 32) * 4 + 16);
        if (!B.seq2 || !B.seq_exc) return fail(c, PJB_ERR_NOMEM, "submit_bam: out of device memory for the 2-bit bases");
    }
    LAUNCH(c, "bam_transcode", bam_transcode, dim3((unsigned)((n + 255) / 256)), dim3(256), R.U, (const iu64 *)c->b_bam_rec.p, (iu64)n, B);
    HIP_TRY(c, hipStreamSynchronize(st)); // `tails` is on this stack frame
    if (c->ktime) ev_collect(c, MISC_POOL);
    if (prof)
        fprintf(stderr, "[host profile] submit_bam tid %d: %zu blocks, %.1f MB -> %.1f MB, %zu records: header scan %.3f, upload %.3f, inflate %.3f, "
                        "boundaries %.3f, fill+sizes+transcode %.3f s\n",
                tid, n_blocks, comp_bytes / 1e6, total / 1e6, n, t_scan, t_up, t_inf, t_walk, now() - t0);
    DevBatch d;
    memset(&d, 0, sizeof d);
    d.n = (int64_t)n;
    d.base = 0;
    d.pos = B.pos; d.flag = B.flag; d.mapq = B.mapq; d.xs = B.xs; d.l_qseq = B.l_qseq; d.mtid = B.mtid; d.mpos = B.mpos;
    d.cig_off = B.cig_off; d.cigar = B.cigar; d.seq_off = B.seq_off; d.seq4 = B.seq4;
    d.name_hash = (const u64 *)B.name_hash;
    d.seq2 = (const uint32_t *)B.seq2;
    d.seq_exc = B.seq_exc;
    oc.on_main_stream = true;
    oc.batches.push_back(d);
    oc.last_known.push_back(0);
    oc.last_pos.push_back(INT32_MIN);
    if (n_records) *n_records = (int64_t)n;
    return PJB_OK;
}

extern "C" int pjb_submit_bam(pjb_ctx *c, int32_t tid, const uint8_t *comp, int64_t comp_bytes, int32_t first_uoffset,
                              int64_t *n_records) {
    if (!c) return PJB_ERR_ARG;
    if (n_records) *n_records = 0;
    if (comp_bytes < 0 || (comp_bytes && !comp) || first_uoffset < 0) return fail(c, PJB_ERR_ARG, "submit_bam: bad arguments");
    if (tid < 0 || (size_t)tid >= c->ref_len.size()) return fail(c, PJB_ERR_ARG, "submit_bam: bad tid %d", tid);
    c->cur_tid = tid;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    OpenContig &oc = c->open[tid];
    if (!oc.batches.empty()) return fail(c, PJB_ERR_STATE, "submit_bam: target %d already has batches (one call per target)", tid);
    const bool prof = getenv("PJB_PROFILE_HOST") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now(), t_scan, t_up;
    std::vector<InfBlock> blocks;
    int64_t total = 0;
    int rc = scan_bgzf(c, comp, comp_bytes, blocks, total);
    if (rc) return rc;
    if (total == 0 || (int64_t)first_uoffset >= total) return PJB_OK;
    t_scan = now() - t0;
    t0 = now();
    hipStream_t st = c->stream;
    if ((rc = ensure(c, c->b_inf_comp, (size_t)comp_bytes + INF_PAD))) return rc;
    {
        // page-locked input (pjb_host_alloc): one DMA, no staging copy
        hipPointerAttribute_t at;
        const bool pinned = hipPointerGetAttributes(&at, comp) == hipSuccess && at.type == hipMemoryTypeHost;
        if (!pinned) (void)hipGetLastError();
        if (pinned) HIP_TRY(c, hipMemcpyAsync(c->b_inf_comp.p, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, st));
        else if ((rc = upload_staged(c, c->b_inf_comp.p, comp, (size_t)comp_bytes))) return rc;
    }
    HIP_TRY(c, hipMemsetAsync((uint8_t *)c->b_inf_comp.p + comp_bytes, 0, INF_PAD, st));
    if (prof) (void)hipStreamSynchronize(st);
    t_up = now() - t0;
    return ingest_staged(c, tid, oc, (const uint8_t *)c->b_inf_comp.p, blocks, comp_bytes, total, first_uoffset, n_records, t_scan, t_up);
}

// ---- the same in pieces -----------------------------------------------------------------------------------------
// (pjb_bam_begin / pjb_bam_piece / pjb_bam_pieces_done / pjb_bam_end, see the header)
struct BamStage {
    Buf dev;                 // the target's BGZF bytes on the device (from the context's pool)
    int64_t total = 0, got = 0;
    std::vector<InfBlock> blocks;
    int64_t total_out = 0;
    int64_t next = 0;        // file-relative offset of the next block header to look at
    uint8_t keep[65536 + 64]; // bytes [keep_at, got) of what arrived, for a block whose header or footer straddles two pieces
    int64_t keep_at = 0, keep_n = 0;
    double t_scan = 0, t_up = 0;
    // the inflate launched at the last piece (launched: ev_inf follows the kernel on its stream)
    bool launched = false;
    iu32 ctl[2] = {0, 0}; // { "some block failed", lanes }: copied to the device asynchronously, so it lives here and not on a stack
    Buf out, d_blocks, d_status, d_scratch, d_bitmap;
    hipEvent_t ev_inf = nullptr, ev_last = nullptr;
};

// a buffer of at least `bytes` from a pool (the smallest that fits), else a new one
static int pool_take(pjb_ctx *c, std::vector<Buf> &pool, Buf &b, size_t bytes) {
    int best = -1;
    for (size_t k = 0; k < pool.size(); k++)
        if (pool[k].cap >= bytes && (best < 0 || pool[k].cap < pool[(size_t)best].cap)) best = (int)k;
    if (best >= 0) {
        b = pool[(size_t)best];
        pool.erase(pool.begin() + best);
        return PJB_OK;
    }
    return ensure(c, b, bytes);
}
static void pool_give(std::vector<Buf> &pool, Buf &b) {
    if (b.p) pool.push_back(b);
    b.p = nullptr;
    b.cap = 0;
}
static void stage_release(pjb_ctx *c, BamStage &st) { // (after the work that uses the buffers has completed)
    pool_give(c->stage_pool, st.dev);
    pool_give(c->out_pool, st.out);
    pool_give(c->misc_pool, st.d_blocks);
    pool_give(c->misc_pool, st.d_status);
    pool_give(c->misc_pool, st.d_scratch);
    pool_give(c->out_pool, st.d_bitmap); // (output-sized: an eighth of the inflated bytes)
    if (st.ev_inf) (void)hipEventDestroy(st.ev_inf);
    if (st.ev_last) (void)hipEventDestroy(st.ev_last);
    st.ev_inf = st.ev_last = nullptr;
}

// every byte of the target has been queued for copying and every block header seen: inflate on a stream of its own, behind
// the last copy.  Nothing here waits; a failure just leaves the inflate to pjb_bam_end.
static void inflate_early(pjb_ctx *c, BamStage &st) {
    const size_t nb = st.blocks.size();
    if (st.launched || nb == 0 || st.total_out <= 0) return;
    size_t lanes = std::min<size_t>((nb + 63) / 64 * 64, (size_t)c->inflate_lanes);
    if (const char *e = getenv("PJB_INF_BLOCKS_PER_LAUNCH")) lanes = std::min<size_t>((nb + 63) / 64 * 64, (size_t)std::max(64, atoi(e)) / 64 * 64);
    if (pool_take(c, c->out_pool, st.out, (size_t)st.total_out + 64) || pool_take(c, c->misc_pool, st.d_blocks, nb * sizeof(InfBlock)) ||
        pool_take(c, c->misc_pool, st.d_status, nb * 4 + 16) || pool_take(c, c->misc_pool, st.d_scratch, lanes * INF_SCRATCH_PER_LANE) ||
        pool_take(c, c->out_pool, st.d_bitmap, nb * INF_BITMAP_WORDS * 8)) {
        std::lock_guard<std::mutex> lk(c->err_mu); // (a failure here just leaves the inflate to pjb_bam_end)
        c->err.clear();
        return;
    }
    hipStream_t &is = c->inf_streams[c->inf_next++ & 3u];
    if (!is) {
        // The inflate streams have the lowest priority: a launch holds every LDS byte of the chip for ~50 ms, and the short
        // kernels beside it -- record parsing, genome uploads, the junc chains of the targets before it -- are what the one
        // host thread that serves all targets waits for (end to end 2.73 -> 2.44 s)
        int lo = 0, hi = 0; // (least, greatest priority)
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&is, hipStreamNonBlocking, lo) != hipSuccess) return;
    }
    if (hipEventCreateWithFlags(&st.ev_last, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&st.ev_inf, hipEventDisableTiming) != hipSuccess) return;
    int *d_status = (int *)st.d_status.p;
    int *d_any = d_status + nb;
    iu32 *d_next = (iu32 *)(d_any + 1);
    st.ctl[0] = 0u;
    st.ctl[1] = (iu32)lanes;
    bool ok = hipEventRecord(st.ev_last, c->stream_up) == hipSuccess && hipStreamWaitEvent(is, st.ev_last, 0) == hipSuccess &&
              hipMemsetAsync((uint8_t *)st.out.p + st.total_out, 0, 64, is) == hipSuccess &&
              hipMemcpyAsync(st.d_blocks.p, st.blocks.data(), nb * sizeof(InfBlock), hipMemcpyHostToDevice, is) == hipSuccess &&
              hipMemcpyAsync(d_any, st.ctl, 8, hipMemcpyHostToDevice, is) == hipSuccess;
    if (ok) {
        ok = hipMemsetAsync(st.d_bitmap.p, 0, nb * INF_BITMAP_WORDS * 8, is) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(bgzf_decode, dim3((unsigned)(lanes / 64)), dim3(64), I3_LDS_BYTES, is, (const uint8_t *)st.dev.p, (const InfBlock *)st.d_blocks.p,
                               (iu32)nb, (uint8_t *)st.out.p, (uint8_t *)st.d_scratch.p, d_status, d_any, d_next, (iu64 *)st.d_bitmap.p, 8);
            hipLaunchKernelGGL(bgzf_resolve, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, is, (const InfBlock *)st.d_blocks.p, (iu32)nb, (uint8_t *)st.out.p,
                               (const iu64 *)st.d_bitmap.p, (const int *)d_status);
            ok = hipGetLastError() == hipSuccess && hipEventRecord(st.ev_inf, is) == hipSuccess;
        }
    }
    if (!ok) { // whatever was queued must be over before the buffers are used again
        (void)hipStreamSynchronize(is);
        (void)hipGetLastError();
        return;
    }
    st.launched = true;
}

// block headers that are complete with the bytes received so far (the last `avail` bytes of the stream are at `p`, the
// first of them is byte `p_at` of the target's bytes); leaves st.next at the first block it cannot finish yet
static int stage_scan(pjb_ctx *c, BamStage &st, const uint8_t *p, int64_t p_at, int64_t avail) {
    auto byte_at = [&](int64_t off) -> int { // a byte of the stream that is still in reach (this piece or the kept tail)
        if (off >= p_at && off < p_at + avail) return p[off - p_at];
        if (off >= st.keep_at && off < st.keep_at + st.keep_n) return st.keep[off - st.keep_at];
        return -1;
    };
    const int64_t end = p_at + avail;
    while (st.next < st.total) {
        const int64_t off = st.next;
        if (off + 18 > end) break;
        uint8_t h[18];
        for (int k = 0; k < 18; k++) {
            const int v = byte_at(off + k);
            if (v < 0) return fail(c, PJB_ERR_STATE, "bam_piece: internal: header byte %lld out of reach", (long long)(off + k));
            h[k] = (uint8_t)v;
        }
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return fail(c, PJB_ERR_BGZF, "not a BGZF block header at byte %lld", (long long)off);
        const uint32_t xlen = h[10] | (uint32_t)h[11] << 8;
        if (off + 12 + xlen > end) break;
        int64_t bsize = -1;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            int f[6];
            for (int k = 0; k < 6; k++) f[k] = x + (uint32_t)k < xlen ? byte_at(off + 12 + x + k) : 0;
            if (f[0] < 0 || f[1] < 0 || f[2] < 0 || f[3] < 0) return fail(c, PJB_ERR_STATE, "bam_piece: internal: extra field out of reach");
            const uint32_t slen = (uint32_t)f[2] | (uint32_t)f[3] << 8;
            if (f[0] == 'B' && f[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = (int64_t)((uint32_t)f[4] | (uint32_t)f[5] << 8) + 1;
            x += 4 + slen;
        }
        if (bsize < 0) return fail(c, PJB_ERR_BGZF, "BGZF block at byte %lld has no BC field", (long long)off);
        if (bsize < (int64_t)xlen + 20 || off + bsize > st.total)
            return fail(c, PJB_ERR_BGZF, "BGZF block at byte %lld has an impossible size %lld", (long long)off, (long long)bsize);
        if (off + bsize > end) break; // its footer has not arrived
        uint32_t isize = 0;
        for (int k = 0; k < 4; k++) {
            const int v = byte_at(off + bsize - 4 + k);
            if (v < 0) return fail(c, PJB_ERR_STATE, "bam_piece: internal: footer byte out of reach");
            isize |= (uint32_t)v << (8 * k);
        }
        if (isize > 65536u) return fail(c, PJB_ERR_BGZF, "BGZF block at byte %lld declares %u inflated bytes", (long long)off, isize);
        InfBlock b;
        b.in_off = (iu64)(off + 12 + xlen);
        b.in_len = (iu32)(bsize - xlen - 20);
        b.out_off = (iu64)st.total_out;
        b.out_len = isize;
        st.blocks.push_back(b);
        st.total_out += isize;
        st.next = off + bsize;
    }
    return PJB_OK;
}

static void bam_stage_clear(pjb_ctx *c) {
    for (auto &is : c->inf_streams)
        if (is) (void)hipStreamSynchronize(is);
    for (auto &kv : c->bam_stage) {
        stage_release(c, *kv.second);
        delete kv.second;
    }
    c->bam_stage.clear();
    for (auto *pool : {&c->stage_pool, &c->out_pool, &c->misc_pool}) {
        for (auto &b : *pool) release(b);
        pool->clear();
    }
    for (auto &is : c->inf_streams)
        if (is) (void)hipStreamDestroy(is);
    for (auto &ev : c->up_events)
        if (ev) (void)hipEventDestroy(ev);
    if (c->ev_up) (void)hipEventDestroy(c->ev_up);
    if (c->stream_up) (void)hipStreamDestroy(c->stream_up);
}

extern "C" int pjb_bam_begin(pjb_ctx *c, int32_t tid, int64_t total_bytes) {
    if (!c) return PJB_ERR_ARG;
    if (tid < 0 || (size_t)tid >= c->ref_len.size() || total_bytes <= 0) return fail(c, PJB_ERR_ARG, "bam_begin: bad arguments (tid %d)", tid);
    std::lock_guard<std::mutex> lk(c->bam_mu);
    if (c->bam_stage.count(tid)) return fail(c, PJB_ERR_STATE, "bam_begin: target %d is being staged already", tid);
    // (that the target has no batches yet is checked by pjb_bam_end, on the thread that owns the open targets)
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    std::unique_ptr<BamStage> st(new (std::nothrow) BamStage());
    if (!st) return fail(c, PJB_ERR_NOMEM, "bam_begin: out of host memory");
    // device buffer: the smallest free one of the pool that fits, else a new one
    const size_t need = (size_t)total_bytes + INF_PAD;
    int best = -1;
    for (size_t k = 0; k < c->stage_pool.size(); k++)
        if (c->stage_pool[k].cap >= need && (best < 0 || c->stage_pool[k].cap < c->stage_pool[(size_t)best].cap)) best = (int)k;
    if (best >= 0) {
        st->dev = c->stage_pool[(size_t)best];
        c->stage_pool.erase(c->stage_pool.begin() + best);
    } else {
        int rc = ensure(c, st->dev, need);
        if (rc) return rc;
    }
    st->total = total_bytes;
    hipError_t he = hipSuccess;
    if (!c->stream_up) he = hipStreamCreateWithFlags(&c->stream_up, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipMemsetAsync((uint8_t *)st->dev.p + total_bytes, 0, INF_PAD, c->stream_up);
    if (he != hipSuccess) {
        pool_give(c->stage_pool, st->dev);
        return fail(c, PJB_ERR_HIP, "bam_begin: %s", hipGetErrorString(he));
    }
    c->bam_stage[tid] = st.release();
    return PJB_OK;
}

static int bam_piece_body(pjb_ctx *c, int32_t tid, BamStage &st, const uint8_t *piece, int64_t bytes, int64_t *ticket);

extern "C" int pjb_bam_piece(pjb_ctx *c, int32_t tid, const uint8_t *piece, int64_t bytes, int64_t *ticket) {
    if (!c || !piece || bytes <= 0) return fail(c, PJB_ERR_ARG, "bam_piece: bad arguments");
    std::lock_guard<std::mutex> lk(c->bam_mu);
    auto it = c->bam_stage.find(tid);
    if (it == c->bam_stage.end()) return fail(c, PJB_ERR_STATE, "bam_piece: target %d was not begun (pjb_bam_begin)", tid);
    const int rc = bam_piece_body(c, tid, *it->second, piece, bytes, ticket);
    if (rc) { // the target's staging is dropped: it has to be begun again
        (void)hipStreamSynchronize(c->stream_up);
        for (auto &is : c->inf_streams)
            if (is) (void)hipStreamSynchronize(is);
        stage_release(c, *it->second);
        delete it->second;
        c->bam_stage.erase(it);
    }
    return rc;
}

static int bam_piece_body(pjb_ctx *c, int32_t tid, BamStage &st, const uint8_t *piece, int64_t bytes, int64_t *ticket) {
    if (st.got + bytes > st.total) return fail(c, PJB_ERR_ARG, "bam_piece: target %d: more bytes than announced", tid);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    // the copy first (asynchronous, on the upload stream), the header hop meanwhile
    HIP_TRY(c, hipMemcpyAsync((uint8_t *)st.dev.p + st.got, piece, (size_t)bytes, hipMemcpyHostToDevice, c->stream_up));
    const int64_t tk = ++c->up_ticket;
    hipEvent_t &ev = c->up_events[(size_t)(tk % (int64_t)PJB_UP_EVENTS)];
    if (!ev) HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    else if (tk - c->up_done >= (int64_t)PJB_UP_EVENTS) { // the ring is full: its oldest copy must have completed
        HIP_TRY(c, hipEventSynchronize(ev));
        c->up_done = std::max<int64_t>(c->up_done, tk - (int64_t)PJB_UP_EVENTS);
    }
    HIP_TRY(c, hipEventRecord(ev, c->stream_up));
    st.t_up += now() - t0;
    t0 = now();
    int rc = stage_scan(c, st, piece, st.got, bytes);
    if (rc) return rc;
    st.got += bytes;
    // keep what the next piece's first block may still need: everything from st.next on, if it is short (a block is
    // at most 64 KB), else nothing (the block then starts in a later piece)
    if (st.next < st.got) {
        const int64_t from = st.next;
        const int64_t n = st.got - from;
        if (n > (int64_t)sizeof st.keep) return fail(c, PJB_ERR_BGZF, "BGZF block at byte %lld is longer than 64 KB", (long long)from);
        uint8_t tmp[sizeof st.keep];
        for (int64_t k = 0; k < n; k++) {
            const int64_t off = from + k;
            tmp[k] = off >= st.got - bytes ? piece[off - (st.got - bytes)] : st.keep[off - st.keep_at];
        }
        memcpy(st.keep, tmp, (size_t)n);
        st.keep_at = from;
        st.keep_n = n;
    } else
        st.keep_n = 0;
    st.t_scan += now() - t0;
    if (ticket) *ticket = tk;
    if (st.got == st.total && st.next == st.total) inflate_early(c, st);
    return PJB_OK;
}

extern "C" int pjb_bam_inflate_done(pjb_ctx *c, int32_t tid) {
    if (!c) return 1;
    std::lock_guard<std::mutex> lk(c->bam_mu);
    auto it = c->bam_stage.find(tid);
    if (it == c->bam_stage.end() || !it->second->launched) return 1; // (nothing in flight: pjb_bam_end does all the work)
    const bool done = hipEventQuery(it->second->ev_inf) == hipSuccess;
    (void)hipGetLastError();
    return done ? 1 : 0;
}

extern "C" int pjb_bam_pieces_done(pjb_ctx *c, int64_t *completed_ticket) {
    if (!c || !completed_ticket) return PJB_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->bam_mu);
    while (c->up_done < c->up_ticket) {
        hipEvent_t ev = c->up_events[(size_t)((c->up_done + 1) % (int64_t)PJB_UP_EVENTS)];
        if (!ev || hipEventQuery(ev) != hipSuccess) break;
        c->up_done++;
    }
    (void)hipGetLastError(); // (hipErrorNotReady is not an error here)
    *completed_ticket = c->up_done;
    return PJB_OK;
}

extern "C" int pjb_bam_end(pjb_ctx *c, int32_t tid, int32_t first_uoffset, int64_t *n_records) {
    if (!c) return PJB_ERR_ARG;
    if (n_records) *n_records = 0;
    std::unique_ptr<BamStage> st;
    {
        std::lock_guard<std::mutex> lk(c->bam_mu);
        auto it = c->bam_stage.find(tid);
        if (it == c->bam_stage.end()) return fail(c, PJB_ERR_STATE, "bam_end: target %d was not begun (pjb_bam_begin)", tid);
        st.reset(it->second);
        c->bam_stage.erase(it);
    }
    struct Return { // the device buffer goes back to the pool whatever happens (after the work that reads it)
        pjb_ctx *c;
        BamStage *st;
        ~Return() {
            (void)hipStreamSynchronize(c->stream);
            if (st->launched) (void)hipEventSynchronize(st->ev_inf); // (the inflate waited for the target's last copy)
            else if (c->stream_up) (void)hipStreamSynchronize(c->stream_up); // early returns: the copies may still read the caller's buffers
            std::lock_guard<std::mutex> lk(c->bam_mu);
            stage_release(c, *st);
        }
    } ret{c, st.get()};
    c->cur_tid = tid;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (first_uoffset < 0) return fail(c, PJB_ERR_ARG, "bam_end: bad first_uoffset");
    if (st->got != st->total) return fail(c, PJB_ERR_ARG, "bam_end: target %d: %lld of %lld bytes arrived", tid, (long long)st->got, (long long)st->total);
    if (st->next != st->total) return fail(c, PJB_ERR_BGZF, "truncated BGZF block at byte %lld", (long long)st->next);
    OpenContig &oc = c->open[tid];
    if (!oc.batches.empty()) return fail(c, PJB_ERR_STATE, "bam_end: target %d already has batches (one call per target)", tid);
    if (st->total_out == 0 || (int64_t)first_uoffset >= st->total_out) return PJB_OK;
    if (st->launched) { // the inflate started with the last piece: wait for it, look at its status words, go on with the records
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = now();
        HIP_TRY(c, hipEventSynchronize(st->ev_inf));
        int rc = inflate_status(c, st->blocks, (const int *)st->d_status.p);
        if (rc) return rc;
        return ingest_parse(c, tid, oc, (const uint8_t *)st->out.p, st->blocks.size(), st->total, st->total_out, first_uoffset, n_records, st->t_scan, st->t_up,
                            now() - t0);
    }
    // the service stream picks up behind the last copy
    if (!c->ev_up) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->ev_up, c->stream_up));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_up, 0));
    return ingest_staged(c, tid, oc, (const uint8_t *)st->dev.p, st->blocks, st->total, st->total_out, first_uoffset, n_records, st->t_scan, st->t_up);
}
