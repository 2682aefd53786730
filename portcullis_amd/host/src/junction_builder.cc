// JunctionBuilder: orchestration of the junc stage on top of the device path.
// Flow and console output follow src/junction_builder.cc:84-291 of the reference.
#include <atomic>
#include <set>
#include <portcullis/junction_builder.hpp>
#include <portcullis/bam/bam_writer.hpp>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <malloc.h>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <sys/stat.h>
#include <thread>

#include "../../../include/portcullis_amd.h"

namespace portcullis {

using bam::BamReader;
using bam::GenomeMapper;
using std::cerr;
using std::cout;
using std::endl;

static bool pathExists(const std::string& p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0;
}

static bool makeDirs(const std::string& p) {
    if (p.empty() || pathExists(p)) return true;
    const size_t slash = p.find_last_of('/');
    if (slash != std::string::npos && slash > 0 && !makeDirs(p.substr(0, slash))) return false;
    return mkdir(p.c_str(), 0777) == 0 || pathExists(p);
}

namespace {
struct WallTimer {  // prints like boost::timer::auto_cpu_timer(1, " = Wall time taken: %ws\n\n")
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double elapsed() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
    ~WallTimer() {
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::ios::fmtflags f(cout.flags());
        cout << " = Wall time taken: " << std::fixed << std::setprecision(1) << s << "s" << endl << endl;
        cout.flags(f);
    }
};
}  // namespace

JunctionBuilder::JunctionBuilder(const std::string& prepDir, const std::string& output) {
    prepData = PreparedFiles(prepDir);
    if (output.empty()) {
        outputDir = ".";
        outputPrefix = "portcullis";
    } else {
        const size_t slash = output.find_last_of('/');
        outputDir = slash == std::string::npos ? "" : output.substr(0, slash);
        outputPrefix = slash == std::string::npos ? output : output.substr(slash + 1);
        if (slash == 0) outputDir = "/";
    }
    if (const char* e = getenv("PORTCULLIS_GPUS")) devices = atoi(e);
    if (const char* e = getenv("PJB_TEST_BATCH")) setBatchRecords((size_t)atol(e));
    if (const char* e = getenv("PORTCULLIS_INGEST")) setDeviceIngest(std::string(e) == "device");
}

namespace {
struct HostProfile {  // PJB_PROFILE_HOST=1: where the host side of findJuncs spends its time
    bool on = getenv("PJB_PROFILE_HOST") != nullptr;
    double t0 = now();  // (static initialisation: about when the process starts)
    void mark(const char* what) {
        if (!on) return;
        std::lock_guard<std::mutex> lk(mu);
        std::cerr << "[host profile] t=" << (now() - t0) << " s: " << what << std::endl;
    }
    std::mutex mu;
    double genome = 0, submit = 0, finish = 0, total = 0;
    // PJB_PROFILE_HOST=2: every command of the device threads and every step of the workers with its start and end
    struct Event {
        double a, b;
        std::string what;
    };
    bool events_on = on && atoi(getenv("PJB_PROFILE_HOST")) >= 2;
    std::vector<Event> events;
    void event(double a, double b, const std::string& what) {
        if (!events_on) return;
        std::lock_guard<std::mutex> lk(mu);
        events.push_back({a - t0, b - t0, what});
    }
    void dumpEvents() {
        if (!events_on) return;
        std::lock_guard<std::mutex> lk(mu);
        std::sort(events.begin(), events.end(), [](const Event& x, const Event& y) { return x.a < y.a; });
        for (auto& e : events) {
            char line[256];
            snprintf(line, sizeof line, "[host event] %8.4f %8.4f %7.1f ms  %s", e.a, e.b, (e.b - e.a) * 1e3, e.what.c_str());
            std::cerr << line << "\n";
        }
    }
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};
HostProfile g_prof;
}  // namespace

void JunctionBuilder::process() {
    const double t_p0 = HostProfile::now();
    // many decode threads allocate and free multi-megabyte arrays: keep them on the heap instead of
    // one mmap/munmap pair each (munmap broadcasts TLB shootdowns to every core running a thread)
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, -1);
    // the HIP runtime takes ~0.2 s to come up: start it now, beside the header / index reads
    deviceCount = std::async(std::launch::async, [] { return pjb_device_count(); }).share();
    const std::string outDir = outputDir.empty() ? "." : outputDir;
    if (!pathExists(outDir) && !makeDirs(outDir))
        throw JunctionBuilderException("Could not create output directory at: " + outDir);
    if (!pathExists(prepData.getSortedBamFilePath()))
        throw JunctionBuilderException("Could not find prepared BAM file at: " + prepData.getSortedBamFilePath());
    try {
        prepData.valid(useCsi);
    } catch (const PrepareException& e) {
        throw JunctionBuilderException(std::string("Prepared data is not complete: ") + prepData.getPrepDir() + " (" + e.what() + ")");
    }
    BamReader reader(prepData.getSortedBamFilePath());
    reader.open(useCsi);
    refs = reader.createRefList();
    refMap = reader.createRefMap(*refs);
    reader.close();
    junctionSystem = JunctionSystem();
    junctionSystem.setRefs(refs);
    if (hostThreads == 0) hostThreads = threads;  // total decode threads survive the per-target cap below
    if (refs->size() < threads) {
        cerr << "Warning: User requested " << threads << " threads but there are only " << refs->size()
             << " target sequences to process.  Setting number of threads to " << refs->size() << "." << endl << endl;
        threads = (uint16_t)refs->size();
    }
    // (the reference forces --separate when --extra is given because calcExtraMetrics re-reads the split files,
    // src/junction_builder.cc:113-117; here --extra works on the records in device memory and writes no files)
    cout << "Settings:" << endl
         << std::boolalpha << " - BAM Strandedness: " << bam::strandednessToString(strandSpecific) << endl
         << " - BAM Read Orientation: " << bam::orientationToString(orientation) << endl
         << " - BAM Indexing mode: " << (useCsi ? "CSI" : "BAI") << endl
         << " - Threads: " << threads << endl
         << " - Separate BAMs: " << separate << endl
         << endl;
    cout << reader.bamDetails() << endl;
    const double t_p1 = HostProfile::now();
    if (separate) separateBams();
    findJunctions();
    const double t_p2 = HostProfile::now();
    cout << "Saving junctions: " << endl;
    {
        WallTimer t;
        junctionSystem.saveAll(outDir + "/" + outputPrefix, source, false, outputExonGFF, outputIntronGFF);
    }
    g_prof.mark("outputs written");
    g_prof.dumpEvents();
    if (g_prof.on)
        cerr << "[host profile] process: header+index " << (t_p1 - t_p0) << " s, findJunctions " << (t_p2 - t_p1) << " s, saveAll "
             << (HostProfile::now() - t_p2) << " s" << endl;
    std::pair<bam::Orientation, bam::Strandedness> actual = junctionSystem.determineStrandedness(true);
    cout << "Determined sequence orientation to be: " << bam::orientationToLongString(actual.first) << endl;
    cout << "Determined RNAseq strandedness to be: " << bam::strandednessToLongString(actual.second) << endl << endl;
    if (strandSpecific != bam::Strandedness::UNKNOWN && strandSpecific != actual.second)
        cerr << "Warning!  User input and portcullis disagree about the strandedness of the dataset" << endl << endl;
}



// src/junction_builder.cc:152-226: one pass over the whole prepared BAM (unplaced records included); a record with an N
// operation goes to <prefix>.spliced.bam, another mapped one to <prefix>.unspliced.bam, the rest to <prefix>.unmapped.bam.
// The reference then shells out to `samtools index` for the first two; BamWriter writes the .bai itself.
void JunctionBuilder::separateBams() {
    WallTimer timer;
    uint64_t splicedCount = 0, unsplicedCount = 0, unmappedCount = 0;
    bam::BamReader reader(prepData.getSortedBamFilePath());
    reader.open(useCsi);
    const int wt = std::max(1, (int)threads);
    bam::BamWriter unsplicedWriter(getUnsplicedBamFile(), wt), splicedWriter(getSplicedBamFile(), wt), unmappedWriter(getUnmappedBamFile(), wt);
    unmappedWriter.setWriteIndex(false);
    cout << "Splitting BAM:" << endl;
    cout << " - Saving unspliced alignments to: " << getUnsplicedBamFile() << endl;
    unsplicedWriter.open(reader.getHeaderText(), reader.getTargets());
    cout << " - Saving spliced alignments to: " << getSplicedBamFile() << endl;
    splicedWriter.open(reader.getHeaderText(), reader.getTargets());
    cout << " - Saving unmapped reads to: " << getUnmappedBamFile() << endl;
    unmappedWriter.open(reader.getHeaderText(), reader.getTargets());
    cout << " - Processing BAM ...";
    cout.flush();
    std::vector<uint8_t> rec;
    reader.rewind();
    while (reader.nextRecord(rec)) {
        const uint8_t* r = rec.data();
        const uint32_t l_name = r[12], n_cig = (uint32_t)r[16] | ((uint32_t)r[17] << 8);
        const uint32_t flag = (uint32_t)r[18] | ((uint32_t)r[19] << 8);
        if (36ull + l_name + 4ull * n_cig > rec.size()) throw JunctionBuilderException("Invalid BAM record layout");
        bool spliced = false;  // BamAlignment::isSplicedRead, lib/src/bam_alignment.cc:294-301
        for (uint32_t k = 0; k < n_cig && !spliced; k++) spliced = (r[36 + l_name + 4 * k] & 15u) == 3u;
        if (spliced) {
            splicedWriter.write(r, rec.size());
            splicedCount++;
        } else if (!(flag & 0x4u)) {
            unsplicedWriter.write(r, rec.size());
            unsplicedCount++;
        } else {
            unmappedWriter.write(r, rec.size());
            unmappedCount++;
        }
    }
    cout << " done." << endl;
    cout << " - Found " << splicedCount << " spliced alignments." << endl;
    cout << " - Found " << unsplicedCount << " unspliced alignments." << endl;
    cout << " - Found " << unmappedCount << " unmapped reads." << endl;
    reader.close();
    cout << " - Indexing unspliced alignments ... ";
    unsplicedWriter.close();
    cout << "done." << endl << " - Indexing spliced alignments ... ";
    splicedWriter.close();
    unmappedWriter.close();
    cout << "done." << endl;
}

// ---------------------------------------------------------------------------------------------
// DeviceThread: the one thread that talks to a GPU.  It owns the pjb context (created here, so HIP
// start-up overlaps the first BGZF blocks) and executes commands from the decode workers in order:
// genome uploads, batches (of several contigs at once, interleaved) and contig finishes.  Keeping a
// single context per GPU avoids the runtime-lock contention of one context per worker.
// ---------------------------------------------------------------------------------------------
// A few page-locked buffers for the file bytes of large inputs (device ingest): a worker preads straight into one,
// the device thread DMAs from it without the staging copy and hands it back.  Allocated on first use and kept.
class PinnedPool {
    struct Buf {
        uint8_t* p = nullptr;
        size_t cap = 0;
        bool busy = false;
    };
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Buf> bufs;

    size_t pieceBytes = 0;  // > 0: every buffer has exactly this size (the ring of pieces of the streaming ingest)

    // Targets stream their pieces a few at a time, in the order they asked: with every worker's target on its way at once
    // all of them arrive at about the same time -- late -- and the device has nothing to inflate until then (measured:
    // first bgzf_inflate 1.0 s after the contexts were ready, PCIe idle before and after a burst).  A target that has the
    // ring to itself and three others is complete after a few hundred milliseconds and inflates while the next ones cross.
    // (One gate per device thread: a target's pieces, its genome and its kernels all go through the context of the worker
    // that took it, so the targets in transfer must be spread over the contexts.)
    struct Gate {
        std::mutex mu;
        std::condition_variable cv;
        std::vector<int> waiting;  // ranks of the targets that wait (the best rank = the smallest goes first)
        int active = 0;
        uint64_t used = 0;  // slots in use (bit k)
    };
    Gate gates[16];
    int gatePermits = 1 << 30;

public:
    explicit PinnedPool(size_t n, size_t piece = 0) : bufs(n), pieceBytes(piece) {}
    size_t piece() const { return pieceBytes; }
    int readThreads = 1;  // threads a target in transfer reads its pieces with
    void setTransferSlots(int perLane) { gatePermits = std::max(1, perLane); }
    // `rank`: the target's place in the order the run wants its targets to cross (the workers all arrive here in the same
    // instant, when the context is ready: who gets the lock first must not decide that chr1 crosses sixth)
    // returns the slot taken (0 .. slots - 1): the slots differ in how their target's bytes travel
    int enterTransfer(int lane, int rank) {
        Gate& g = gates[lane & 15];
        std::unique_lock<std::mutex> lk(g.mu);
        g.waiting.push_back(rank);
        g.cv.wait(lk, [&] { return g.active < gatePermits && *std::min_element(g.waiting.begin(), g.waiting.end()) == rank; });
        g.waiting.erase(std::find(g.waiting.begin(), g.waiting.end(), rank));
        g.active++;
        int slot = 0;
        while (slot < 63 && (g.used >> slot) & 1) slot++;
        g.used |= 1ull << slot;
        g.cv.notify_all();
        return slot;
    }
    void leaveTransfer(int lane, int slot) {
        Gate& g = gates[lane & 15];
        std::lock_guard<std::mutex> lk(g.mu);
        g.active--;
        g.used &= ~(1ull << slot);
        g.cv.notify_all();
    }
    // PORTCULLIS_REGISTER_SLOT=k: the target in slot k sends pieces of the file's own mapping (page-locked for the copy)
    int registerSlot = -1;
    ~PinnedPool() {
        for (auto& b : bufs) pjb_host_free(b.p);
    }
    // a free buffer of the ring, or nullptr at once (ring pieces only: every buffer has the ring's size once allocated)
    uint8_t* tryAcquire(size_t bytes) { return take(bytes, false); }
    uint8_t* acquire(size_t bytes) { return take(bytes, true); }

private:
    // The buffer is picked and marked busy under one lock (a second caller can never be sent to sleep for a buffer the
    // first one saw); a first-use or growing allocation happens with the slot marked busy and its pointer and size are
    // published under the lock again (release() compares pointers of every slot).
    uint8_t* take(size_t bytes, bool wait) {
        Buf* mine = nullptr;
        {
            std::unique_lock<std::mutex> lk(mu);
            auto anyFree = [&] {
                for (auto& b : bufs)
                    if (!b.busy) return true;
                return false;
            };
            if (!anyFree()) {
                if (!wait) return nullptr;
                cv.wait(lk, anyFree);
            }
            for (auto& b : bufs)  // prefer one that is large enough already
                if (!b.busy && b.cap >= bytes) mine = &b;
            if (!mine)
                for (auto& b : bufs)
                    if (!b.busy) mine = &b;
            mine->busy = true;
            if (mine->cap >= bytes) return mine->p;
        }
        uint8_t* old = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu);
            old = mine->p;
            mine->p = nullptr;
            mine->cap = 0;
        }
        pjb_host_free(old);
        const size_t cap = pieceBytes ? bytes : bytes + bytes / 8;  // (the ring's pieces never grow; page-locking costs ~0.1 s per GB, twice: to get and to give back)
        uint8_t* np = (uint8_t*)pjb_host_alloc(cap);
        std::lock_guard<std::mutex> lk(mu);
        if (!np) {
            mine->busy = false;
            cv.notify_all();
            return nullptr;
        }
        mine->p = np;
        mine->cap = cap;
        return np;
    }

public:
    void release(uint8_t* p, void* which = nullptr) {
        std::lock_guard<std::mutex> lk(mu);
        for (auto& b : bufs)
            if ((p && b.p == p) || (which && &b == which)) b.busy = false;
        cv.notify_all();
    }
};

struct ContigDone {
    pjb_region_result rr;
    std::vector<pjb_junction_row> rows;
    size_t rowBase = 0;  // --extra: index of rows[0] in the context's row table (pjb_extra_finish's order)
};

// a target between "asked to be finished" and "its chain has been collected"
struct DeferredTarget {
    std::promise<void> seen;
    std::promise<ContigDone> done;
    std::future<ContigDone> fut;
    std::string decodeError, genomeError, name;
    bool any = false;
    double t_begin = 0, t_decoded = 0, t_blocked = 0, t_genome = 0;
};

class DeviceThread {
public:
    struct Cmd {
        enum Kind { GENOME, BATCH, BAM, FINISH, EXTRA, STOP, BAMBEGIN, BAMPIECE, BAMEND, FLUSH } kind = STOP;
        int32_t tid = -1;
        std::string genome;
        // GENOME with the record's bytes as they are in the FASTA file (page-locked, from rawPool; the device takes the line
        // terminators out): rawBytes > 0
        uint8_t* raw = nullptr;
        size_t rawBytes = 0;
        int32_t lineBases = 0, lineWidth = 0;
        int64_t genomeLen = 0;
        PinnedPool* rawPool = nullptr;
        std::promise<bool>* rawDone = nullptr;  // false: the record is not laid out as its index says (the worker sends the filtered bases)
        bam::ReadBatch batch;
        std::vector<bam::ReadBatch>* spare = nullptr;  // where the batch storage goes back to
        std::mutex* spareMu = nullptr;
        std::promise<ContigDone>* done = nullptr;
        std::promise<void>* seen = nullptr;  // FINISH: fulfilled when the device thread takes the command (everything the worker queued before it -- batches that point at the worker's stack -- has been served)
        // BAM: the target's file bytes for the device-side ingest (freed by the device thread)
        uint8_t* bamBytes = nullptr;
        size_t bamSize = 0;
        uint32_t bamFirst = 0;
        PinnedPool* bamPool = nullptr;  // where bamBytes goes back to (nullptr: bigFree)
        std::promise<int64_t>* bamDone = nullptr;
        std::promise<std::vector<pjb_extra_row>>* extraDone = nullptr;  // EXTRA: calcExtraMetrics for every row so far
    };

    // blocks until this thread's context exists (or failed): page-locking the file pieces and creating contexts at the same
    // time fight over the runtime's locks (contexts ready at 0.65 s instead of 0.4 s)
    void waitReady() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return ready; });
    }
    // the context, for the calls that may come from other threads (pjb_bam_begin / _piece / _pieces_done); nullptr if its
    // creation failed (the commands then report why)
    pjb_ctx* context() {
        waitReady();
        return sharedCtx;
    }

private:
    bool ready = false;
    pjb_ctx* sharedCtx = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Cmd> q;
    size_t cap = 6;  // (commands waiting for the device thread; workers block when it is full)
    std::map<int32_t, std::string> failed;  // contig -> first error
    std::string fatal;                      // context creation failed

    // The chain plan (pjb_plan_groups over the targets this thread will be asked to finish, in index order): a FINISH of a target that
    // belongs to a group of several waits here until the group's last member has been asked for, then the group is queued as ONE
    // kernel chain (pjb_finish_group_begin) -- three chains for a human genome instead of twenty-five, which is what bench.py measures.
    // Empty: every target is a chain of its own (several contexts share the targets, --extra, PORTCULLIS_CHAIN_PLAN=targets).
    std::vector<std::vector<int32_t>> plan;
    void run(int device, bam::Orientation orientation, bam::Strandedness strandedness, std::vector<int32_t> lens,
             std::shared_future<int> deviceCount, bool extra) {
        pjb_ctx* ctx = nullptr;
        try {
            if (deviceCount.get() <= 0)
                throw JunctionBuilderException("No MI355X (HIP device) is visible: the junc hot path runs on the GPU and has no CPU fallback");
            pjb_config cfg;
            memset(&cfg, 0, sizeof cfg);
            cfg.abi_version = PJB_ABI_VERSION;
            // PORTCULLIS_DEVICES_SHARE_GPU=1: every device thread of a --devices N run uses GPU 0 (a one-GPU box walks through
            // the N-device code path: worker -> device thread assignment, N contexts, the host merge; never a measurement)
            cfg.device = getenv("PORTCULLIS_DEVICES_SHARE_GPU") ? 0 : device;
            cfg.orientation = (int32_t)orientation;
            cfg.strandedness = (int32_t)strandedness;
            if (extra) cfg.flags |= PJB_FLAG_EXTRA;
            if (pjb_create(&ctx, &cfg) != PJB_OK) throw JunctionBuilderException(std::string("pjb_create: ") + pjb_last_error(nullptr));
            if (pjb_set_refs(ctx, (int32_t)lens.size(), lens.data()) != PJB_OK)
                throw JunctionBuilderException(std::string("pjb_set_refs: ") + pjb_last_error(ctx));
        } catch (const std::exception& e) {
            fatal = e.what();
        }
        // Targets are QUEUED on the device (pjb_finish_contig_begin) and collected later (_end): the kernel chains of up
        // to kQueued targets run side by side on the GPU, the device never waits for this thread between targets, and
        // the next target's upload / ingest overlaps the queued chains.  The rows of every target stay in the context's
        // table (rows arrive in queue order; rowsSoFar marks where the next target's begin).  --extra queues the same
        // way (a target's extra metrics are queued when its chain is collected).
        size_t kQueued = 3;  // (the library creates the streams of four control slots up front; deeper ones on a busy device cost seconds)
        struct Pending {
            int32_t tid;
            std::promise<ContigDone>* done; // (the worker thread that owns it waits on its future)
            std::vector<int32_t> tids;      // a group chain: its members, in the order they were named to pjb_finish_group_begin
            std::vector<std::promise<ContigDone>*> dones;
        };
        std::deque<Pending> pending;
        size_t rowsSoFar = 0;
        const bool printPlan = getenv("PJB_PRINT_CHAIN_PLAN") != nullptr;
        auto collectOldest = [&]() {
            Pending p = std::move(pending.front());
            pending.pop_front();
            std::string err;
            const pjb_junction_row* rows = nullptr;
            int64_t n = 0;
            if (!p.tids.empty()) { // a group: one result per member, the rows member after member in the order of `tids`
                std::vector<pjb_region_result> rr(p.tids.size());
                for (auto& r : rr) memset(&r, 0, sizeof r);
                if (pjb_finish_group_end(ctx, p.tids.data(), (int32_t)p.tids.size(), rr.data()) != PJB_OK) err = std::string("pjb_finish_group: ") + pjb_last_error(ctx);
                else if (pjb_collect(ctx, &rows, &n) != PJB_OK) err = std::string("pjb_collect: ") + pjb_last_error(ctx);
                size_t at = rowsSoFar;
                for (size_t m = 0; m < p.tids.size(); m++) {
                    (void)pjb_release_contig(ctx, p.tids[m]);
                    if (!err.empty()) {
                        p.dones[m]->set_exception(std::make_exception_ptr(JunctionBuilderException(err)));
                        continue;
                    }
                    ContigDone d;
                    d.rr = rr[m];
                    size_t e = at;
                    while (e < (size_t)n && rows[e].refid == p.tids[m]) e++;
                    d.rows.assign(rows + at, rows + e);
                    d.rowBase = at;
                    at = e;
                    p.dones[m]->set_value(std::move(d));
                }
                if (err.empty()) rowsSoFar = (size_t)n;
                g_prof.mark(("chain collected: group from target " + std::to_string(p.tids[0])).c_str());
                return;
            }
            ContigDone d;
            memset(&d.rr, 0, sizeof d.rr);
            if (pjb_finish_contig_end(ctx, p.tid, &d.rr) != PJB_OK) err = std::string("pjb_finish_contig: ") + pjb_last_error(ctx);
            else if (pjb_collect(ctx, &rows, &n) != PJB_OK) err = std::string("pjb_collect: ") + pjb_last_error(ctx);
            else {
                d.rows.assign(rows + rowsSoFar, rows + n);
                d.rowBase = rowsSoFar;
                rowsSoFar = (size_t)n;
            }
            (void)pjb_release_contig(ctx, p.tid);
            if (err.empty()) p.done->set_value(std::move(d));
            else p.done->set_exception(std::make_exception_ptr(JunctionBuilderException(err)));
        };
        // ---- the plan's book-keeping: group of a target, the members that have been asked for so far
        std::map<int32_t, size_t> groupOf;
        for (size_t g = 0; g < plan.size(); g++)
            for (int32_t t : plan[g]) groupOf[t] = g;
        struct Waiting {
            std::vector<std::pair<int32_t, std::promise<ContigDone>*>> got;
            bool single = false; // a member failed, or the library said "not as a group": the members go one by one
        };
        std::vector<Waiting> waiting(plan.size());
        if (printPlan && !plan.empty()) {
            std::string txt;
            for (auto& g : plan) {
                txt += txt.empty() ? "" : " | ";
                for (size_t k = 0; k < g.size(); k++) txt += (k ? "," : "") + std::to_string(g[k]);
            }
            cerr << "[chain plan] " << plan.size() << " chains: " << txt << endl;
        }
        auto beginSingle = [&](int32_t tid, std::promise<ContigDone>* done) {
            while (pending.size() >= kQueued) collectOldest();
            if (pjb_finish_contig_begin(ctx, tid) != PJB_OK) {
                const std::string err = std::string("pjb_finish_contig: ") + pjb_last_error(ctx);
                (void)pjb_release_contig(ctx, tid);
                done->set_exception(std::make_exception_ptr(JunctionBuilderException(err)));
            } else {
                if (printPlan) cerr << "[chain] target " << tid << endl;
                g_prof.mark(("chain queued: target " + std::to_string(tid)).c_str());
                pending.push_back(Pending{tid, done, {}, {}});
            }
        };
        auto beginGroup = [&](Waiting& w) {
            std::sort(w.got.begin(), w.got.end());
            std::vector<int32_t> tids;
            std::vector<std::promise<ContigDone>*> dones;
            for (auto& x : w.got) tids.push_back(x.first), dones.push_back(x.second);
            w.got.clear();
            if (tids.empty()) return;
            while (pending.size() >= kQueued) collectOldest();
            if (tids.size() > 1 && !w.single && pjb_finish_group_begin(ctx, tids.data(), (int32_t)tids.size()) == PJB_OK) {
                if (printPlan) {
                    std::string txt;
                    for (size_t k = 0; k < tids.size(); k++) txt += (k ? "," : "") + std::to_string(tids[k]);
                    cerr << "[chain] group " << txt << endl;
                }
                g_prof.mark(("chain queued: group of " + std::to_string(tids.size()) + " from target " + std::to_string(tids[0])).c_str());
                pending.push_back(Pending{tids[0], nullptr, tids, dones});
                return;
            }
            // (PJB_ERR_ARG: "not as a group" -- a target with characters outside the nucleotide alphabet, ...: one by one)
            for (size_t k = 0; k < tids.size(); k++) beginSingle(tids[k], dones[k]);
        };
        g_prof.mark("device thread: context ready");
        sharedCtx = fatal.empty() ? ctx : nullptr;
        static std::atomic<int> profIds{0};
        const int profId = profIds++;
        {
            std::lock_guard<std::mutex> lk(mu);
            ready = true;
            cv.notify_all();
        }
        double tKind[16] = {0}, tIdle = 0, tCollect = 0;  // PJB_PROFILE_HOST: where this thread's time goes
        // pieces of file bytes whose copy to the device is in flight: (ticket, buffer, pool); released when pjb_bam_pieces_done
        // says the copy has left the buffer
        struct InFlight {
            int64_t ticket;
            uint8_t* buf;
            PinnedPool* pool;
        };
        std::deque<InFlight> inflight;
        auto releaseDone = [&](bool all) {
            if (inflight.empty() || !ctx) return;
            int64_t done = 0;
            if (pjb_bam_pieces_done(ctx, &done) != PJB_OK) done = all ? INT64_MAX : 0;
            while (!inflight.empty() && (all || inflight.front().ticket <= done)) {
                inflight.front().pool->release(inflight.front().buf);
                inflight.pop_front();
            }
        };
        const double tStart = HostProfile::now();
        // BAMEND commands whose inflate (started by the target's last piece) is still running: the thread serves other
        // targets meanwhile -- pieces, whose last one starts the next inflate beside this one -- instead of waiting
        std::deque<Cmd> deferred;
        std::set<int32_t> ended;  // targets whose records are complete (BAMEND / BAM taken from the queue)
        auto readyDeferred = [&]() -> int {
            for (size_t k = 0; k < deferred.size(); k++)
                if (!ctx || pjb_bam_inflate_done(ctx, deferred[k].tid)) return (int)k;
            return -1;
        };
        for (;;) {
            Cmd c;
            bool have = false;
            {
                const int k = readyDeferred();
                if (k >= 0) {
                    c = std::move(deferred[(size_t)k]);
                    deferred.erase(deferred.begin() + k);
                    have = true;
                }
            }
            // chains that have completed are collected at once, without ever waiting for one that has not: a chain queued beside
            // the inflates of the next targets can take 100 ms and more (their resident workgroups hold the CUs' LDS), and a
            // thread that sat in pjb_finish_contig_end for that long held up every other target's commands -- and, through the
            // bounded command queue, the workers that read the file (the input stood still whenever this thread waited)
            while (!pending.empty() && ctx && pjb_finish_ready(ctx)) {
                const double t0 = HostProfile::now();
                const int ptid = pending.front().tid;
                collectOldest();
                tCollect += HostProfile::now() - t0;
                g_prof.event(t0, HostProfile::now(), "dev" + std::to_string(profId) + " collect tid " + std::to_string(ptid));
            }
            if (!have) {
                std::unique_lock<std::mutex> lk(mu);
                const double t0 = HostProfile::now();
                bool again = false;
                while (q.empty()) {
                    if (inflight.empty() && deferred.empty() && pending.empty()) cv.wait(lk, [&] { return !q.empty(); });
                    else {  // a worker may be waiting for one of the pieces in flight: keep handing them back; queued chains complete
                        cv.wait_for(lk, std::chrono::microseconds(200), [&] { return !q.empty(); });
                        lk.unlock();
                        releaseDone(false);
                        const bool ready = readyDeferred() >= 0 || (!pending.empty() && ctx && pjb_finish_ready(ctx));
                        lk.lock();
                        if (ready && q.empty()) {
                            again = true;
                            break;
                        }
                    }
                }
                tIdle += HostProfile::now() - t0;
                if (again) continue;
                // a genome is needed when its target is finished, the file pieces are needed now: uploads of genomes whose
                // target's records are not complete yet let every other command pass (each is 30-70 ms of allocations and
                // a synchronisation, and pieces stuck behind them left PCIe idle at the start of a run)
                size_t pick = 0;
                // (only the uploads a worker waits for before it asks for the finish: the others must keep their place)
                if (q.front().kind == Cmd::GENOME && q.front().rawDone && !ended.count(q.front().tid)) {
                    size_t urgent = q.size(), other = q.size();
                    for (size_t k = 0; k < q.size(); k++) {
                        if (q[k].kind == Cmd::GENOME && q[k].rawDone) {
                            if (urgent == q.size() && ended.count(q[k].tid)) urgent = k;
                        } else if (other == q.size())
                            other = k;
                    }
                    pick = urgent < q.size() ? urgent : other < q.size() ? other : 0;
                }
                c = std::move(q[pick]);
                q.erase(q.begin() + (long)pick);
                if (c.kind == Cmd::BAMEND || c.kind == Cmd::BAM) ended.insert(c.tid);
                cv.notify_all();
                if (c.kind == Cmd::BAMEND && ctx) {
                    // (asked without the queue's lock: pjb_bam_inflate_done takes the context's staging lock, which a worker
                    // holds for the length of a pjb_bam_piece -- push() must not wait for that)
                    lk.unlock();
                    const bool inflated = pjb_bam_inflate_done(ctx, c.tid) != 0;
                    lk.lock();
                    if (!inflated) {
                        deferred.push_back(std::move(c));
                        continue;
                    }
                }
                if (c.kind == Cmd::STOP && !deferred.empty()) { // (cannot happen: a worker waits for its BAMEND; keep the order anyway)
                    q.push_back(std::move(c));
                    continue;
                }
            }
            struct KindTimer {
                double* slot;
                int kind, tid, dev;
                double t0 = HostProfile::now();
                ~KindTimer() {
                    const double t1 = HostProfile::now();
                    *slot += t1 - t0;
                    static const char* names[] = {"GENOME", "BATCH", "BAM", "FINISH", "EXTRA", "STOP", "BAMBEGIN", "BAMPIECE", "BAMEND", "FLUSH"};
                    if (g_prof.events_on) g_prof.event(t0, t1, std::string("dev") + std::to_string(dev) + " " + names[kind] + " tid " + std::to_string(tid));
                }
            } kindTimer{&tKind[(int)c.kind & 15], (int)c.kind, (int)c.tid, profId};
            releaseDone(false);
            if (c.kind == Cmd::STOP) {
                for (auto& w : waiting) beginGroup(w);
                while (!pending.empty()) collectOldest();
                releaseDone(true);
                if (g_prof.on) {
                    std::lock_guard<std::mutex> lk(g_prof.mu);
                    cerr << "[host profile] device thread: alive " << (HostProfile::now() - tStart) << " s: idle " << tIdle << ", collect " << tCollect
                         << ", GENOME " << tKind[(int)Cmd::GENOME] << ", BATCH " << tKind[(int)Cmd::BATCH] << ", BAM " << tKind[(int)Cmd::BAM]
                         << ", BAMPIECE " << tKind[(int)Cmd::BAMPIECE] << ", BAMEND " << tKind[(int)Cmd::BAMEND] << ", FINISH "
                         << tKind[(int)Cmd::FINISH] << ", EXTRA " << tKind[(int)Cmd::EXTRA] << endl;
                }
                break;
            }
            std::string err = fatal;
            if (err.empty() && failed.count(c.tid)) err = failed[c.tid];
            if (c.kind == Cmd::GENOME && c.rawDone) {
                int ok = 0;
                if (err.empty() && pjb_upload_contig_fasta(ctx, c.tid, c.raw, (int64_t)c.rawBytes, c.lineBases, c.lineWidth, c.genomeLen, &ok) != PJB_OK) {
                    failed[c.tid] = std::string("pjb_upload_contig_fasta: ") + pjb_last_error(ctx);
                    ok = 1;  // (an error, not a malformed record: no second attempt)
                }
                if (c.rawPool) c.rawPool->release(c.raw);
                c.rawDone->set_value(ok != 0 || !err.empty());
            } else if (c.kind == Cmd::GENOME) {
                if (err.empty() && pjb_upload_contig(ctx, c.tid, (const uint8_t*)c.genome.data(), (int64_t)c.genome.size()) != PJB_OK)
                    failed[c.tid] = std::string("pjb_upload_contig: ") + pjb_last_error(ctx);
            } else if (c.kind == Cmd::BATCH) {
                if (err.empty()) {
                    pjb_batch pb;
                    c.batch.view(pb);
                    if (pjb_submit_batch(ctx, c.tid, &pb) != PJB_OK) failed[c.tid] = std::string("pjb_submit_batch: ") + pjb_last_error(ctx);
                }
                if (c.spare) {
                    std::lock_guard<std::mutex> lk(*c.spareMu);
                    if (c.spare->size() < 4) c.spare->emplace_back(std::move(c.batch));
                }
            } else if (c.kind == Cmd::BAM) {
                int64_t n = 0;
                if (err.empty() && pjb_submit_bam(ctx, c.tid, c.bamBytes, (int64_t)c.bamSize, (int32_t)c.bamFirst, &n) != PJB_OK)
                    failed[c.tid] = std::string("pjb_submit_bam: ") + pjb_last_error(ctx) +
                                    " (--ingest host decodes the file on the host threads and streams batches instead)";
                if (c.bamPool) c.bamPool->release(c.bamBytes);
                else bam::bigFree(c.bamBytes);
                c.bamDone->set_value(n);
            } else if (c.kind == Cmd::BAMBEGIN) {
                if (err.empty() && pjb_bam_begin(ctx, c.tid, (int64_t)c.bamSize) != PJB_OK)
                    failed[c.tid] = std::string("pjb_bam_begin: ") + pjb_last_error(ctx);
            } else if (c.kind == Cmd::BAMPIECE) {
                int64_t ticket = 0;
                bool queued = false;
                if (err.empty()) {
                    if (pjb_bam_piece(ctx, c.tid, c.bamBytes, (int64_t)c.bamSize, &ticket) != PJB_OK)
                        failed[c.tid] = std::string("pjb_bam_piece: ") + pjb_last_error(ctx) +
                                        " (--ingest host decodes the file on the host threads and streams batches instead)";
                    else
                        queued = true;
                }
                if (queued) inflight.push_back(InFlight{ticket, c.bamBytes, c.bamPool});
                else c.bamPool->release(c.bamBytes);  // (a failing piece call has waited for the upload stream)
            } else if (c.kind == Cmd::BAMEND) {
                int64_t n = 0;
                if (!err.empty() && ctx) (void)pjb_bam_end(ctx, c.tid, (int32_t)c.bamFirst, nullptr);  // (drops what was staged)
                if (err.empty() && pjb_bam_end(ctx, c.tid, (int32_t)c.bamFirst, &n) != PJB_OK)
                    failed[c.tid] = std::string("pjb_bam_end: ") + pjb_last_error(ctx) +
                                    " (--ingest host decodes the file on the host threads and streams batches instead)";
                releaseDone(false);
                c.bamDone->set_value(n);
            } else if (c.kind == Cmd::FLUSH) { // (no more targets will come: what still waits for the rest of its group goes now)
                for (auto& w : waiting) beginGroup(w);
                while (!pending.empty()) collectOldest();
            } else if (c.kind == Cmd::FINISH) {
                if (c.seen) c.seen->set_value();
                auto git = groupOf.find(c.tid);
                if (err.empty() && git != groupOf.end() && plan[git->second].size() > 1 && !waiting[git->second].single) {
                    Waiting& w = waiting[git->second];
                    w.got.emplace_back(c.tid, c.done);
                    if (w.got.size() == plan[git->second].size()) beginGroup(w);
                } else if (err.empty()) {
                    beginSingle(c.tid, c.done);
                } else {
                    if (git != groupOf.end()) { // the group is not complete any more: its members go one by one
                        Waiting& w = waiting[git->second];
                        w.single = true;
                        beginGroup(w);
                    }
                    if (ctx) {
                        while (!pending.empty()) collectOldest();
                        pjb_region_result dummy;
                        (void)pjb_finish_contig(ctx, c.tid, &dummy); // drop whatever was submitted
                        const pjb_junction_row* rows = nullptr;
                        int64_t n = 0;
                        if (pjb_collect(ctx, &rows, &n) == PJB_OK) rowsSoFar = (size_t)n; // (rows of a dropped target are skipped)
                        (void)pjb_release_contig(ctx, c.tid);
                    }
                    c.done->set_exception(std::make_exception_ptr(JunctionBuilderException(err)));
                }
                failed.erase(c.tid);
            } else if (c.kind == Cmd::EXTRA) {
                while (!pending.empty()) collectOldest();
                const pjb_extra_row* xr = nullptr;
                int64_t n = 0;
                if (err.empty() && pjb_extra_finish(ctx, &xr, &n) != PJB_OK) err = std::string("pjb_extra_finish: ") + pjb_last_error(ctx);
                if (err.empty()) c.extraDone->set_value(std::vector<pjb_extra_row>(xr, xr + n));
                else c.extraDone->set_exception(std::make_exception_ptr(JunctionBuilderException(err)));
            }
        }
        if (ctx) pjb_destroy(ctx);
    }

public:
    DeviceThread(int device, bam::Orientation o, bam::Strandedness s, const std::vector<int32_t>& lens, std::shared_future<int> dc,
                 bool extra = false, std::vector<std::vector<int32_t>> chainPlan = {}) {
        plan = std::move(chainPlan);
        th = std::thread([=] { run(device, o, s, lens, dc, extra); });
    }
    bool grouped() const { return !plan.empty(); }
    ~DeviceThread() {
        Cmd c;
        c.kind = Cmd::STOP;
        push(std::move(c));
        th.join();
    }
    int lane = 0;  // index among the device threads (the transfer gate of this context)
    void push(Cmd&& c) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return q.size() < cap; });
        q.emplace_back(std::move(c));
        cv.notify_all();
    }
};

// One target sequence: the decode thread (with its inner pool) pushes batches straight to the device
// thread, this thread reads the genome meanwhile, then asks for the contig to be finished.
void JunctionBuilder::findJuncs(DeviceThread& device, BamReader& reader, GenomeMapper& gmap, int32_t seq) {
    RegionResult& res = results[(size_t)seq];
    if (!reader.hasAlignments(seq)) return;  // nothing placed on this target: counters keep their neutral values
    const double t_begin = HostProfile::now();
    const std::string name = refs->at((size_t)seq)->name;
    std::vector<bam::ReadBatch> spare;
    std::mutex spareMu;
    std::string decodeError;
    bool any = false;
    double t_blocked = 0;
    // raised by the decoder thread when the target's file pieces are all on their way (or it has stopped): the genome is
    // needed when the target is finished, the pieces are needed now -- until then this target's genome leaves the cores, the
    // page-locking calls and PCIe to the pieces (PORTCULLIS_GENOME_EARLY=1: both at once, as before)
    std::promise<void> piecesGone;
    std::shared_future<void> piecesGoneF = piecesGone.get_future().share();
    bool piecesGoneSet = false;
    auto raisePiecesGone = [&] {
        if (!piecesGoneSet) {
            piecesGoneSet = true;
            piecesGone.set_value();
        }
    };
    const bool genomeLate = deviceIngest && pinnedPool;
    std::thread decoder([&] {
        struct Raise {
            std::function<void()> f;
            ~Raise() { f(); }
        } raiseAtExit{raisePiecesGone};
        auto send = [&](bam::ReadBatch& b) {
            const double tb0 = HostProfile::now();
            DeviceThread::Cmd c;
            c.kind = DeviceThread::Cmd::BATCH;
            c.tid = seq;
            std::swap(c.batch, b);
            c.spare = &spare;
            c.spareMu = &spareMu;
            any = true;
            device.push(std::move(c));
            t_blocked += HostProfile::now() - tb0;
            std::lock_guard<std::mutex> lk(spareMu);
            if (!spare.empty()) {
                std::swap(b, spare.back());
                spare.pop_back();
            }
        };
        try {
            if (deviceIngest) {
                // the device inflates and parses: this thread only moves the target's file bytes
                size_t nb = 0;
                uint32_t firstU = 0;
                uint64_t fileOff = 0;
                uint8_t* bytes = nullptr;
                PinnedPool* pool = nullptr;
                if (reader.regionSpan(seq, fileOff, nb, firstU) && pinnedPool && pinnedPool->piece() > 0 && nb >= pieceMinTarget) {
                    // large target of a large file: the bytes go to the device in pieces through a small ring of
                    // page-locked buffers (page-locking a buffer per target costs ~0.15 s per GB); the device thread
                    // copies a piece while this thread reads the next one
                    const size_t piece = pinnedPool->piece();
                    const double tg0 = HostProfile::now();
                    const int slot = pinnedPool->enterTransfer(device.lane, transferRank(seq));
                    g_prof.event(tg0, HostProfile::now(), "worker gate wait tid " + std::to_string(seq));
                    struct Leave {
                        PinnedPool* p;
                        int lane, slot;
                        ~Leave() {
                            if (p) p->leaveTransfer(lane, slot);
                        }
                        void now() {
                            if (p) p->leaveTransfer(lane, slot);
                            p = nullptr;
                        }
                    } leave{pinnedPool.get(), device.lane, slot};
                    // One slot's target crosses straight out of the page cache: pieces of the file's mapping are page-locked
                    // for their copy (2.5 ms per 64 MB, one thread; registration does not scale over threads) while the other
                    // slot's readers copy theirs into the ring -- two different resources.
                    size_t mapBytes = 0;
                    const uint8_t* fileMap = slot == pinnedPool->registerSlot ? bam::BamReader::mapFile(prepData.getSortedBamFilePath(), mapBytes) : nullptr;
                    const int readThreads = fileMap || pinnedPool->registerSlot < 0 ? pinnedPool->readThreads : pinnedPool->readThreads * 2;
                    // This thread hands the pieces to the device itself (pjb_bam_begin / _piece are safe beside the device
                    // thread's calls): queued behind genome uploads, finishes and record parsing on the device thread the
                    // copies started late and PCIe idled between them.
                    pjb_ctx* dctx = directPieces ? device.context() : nullptr;
                    std::string readError;
                    struct Mine {
                        int64_t ticket;
                        uint8_t* buf;  // a ring buffer, or
                        void* reg;     // a registered range of the file's mapping
                        size_t regBytes;
                    };
                    std::deque<Mine> mine;  // pieces of this target whose copy may still read the buffer
                    auto releaseDone = [&](bool all) {
                        int64_t done = 0;
                        if (mine.empty()) return;
                        if (!dctx || pjb_bam_pieces_done(dctx, &done) != PJB_OK) done = all ? INT64_MAX : 0;
                        for (int spin = 0; all && dctx && done < mine.back().ticket && spin < 20000; spin++) {  // (at most 2 s: the copies of a failed target)
                            std::this_thread::sleep_for(std::chrono::microseconds(100));
                            if (pjb_bam_pieces_done(dctx, &done) != PJB_OK) break;
                        }
                        if (all) done = INT64_MAX;
                        while (!mine.empty() && mine.front().ticket <= done) {
                            if (mine.front().reg) (void)pjb_host_unregister(mine.front().reg);
                            else pinnedPool->release(mine.front().buf);
                            mine.pop_front();
                        }
                    };
                    if (dctx) {
                        const double tb0 = HostProfile::now();
                        if (pjb_bam_begin(dctx, seq, (int64_t)nb) != PJB_OK) readError = std::string("pjb_bam_begin: ") + pjb_last_error(dctx);
                        g_prof.event(tb0, HostProfile::now(), "worker bam_begin tid " + std::to_string(seq));
                    } else {
                        DeviceThread::Cmd b0;
                        b0.kind = DeviceThread::Cmd::BAMBEGIN;
                        b0.tid = seq;
                        b0.bamSize = nb;
                        device.push(std::move(b0));
                    }
                    // (with the mapping: pieces end on page boundaries of the FILE, so that no two pieces share a page)
                    const size_t firstPiece = fileMap ? piece - (size_t)(fileOff & 4095) : piece;
                    for (size_t off = 0, step = firstPiece; off < nb && readError.empty(); off += step, step = piece) {
                        const size_t n = std::min(step, nb - off);
                        const double ta0 = HostProfile::now();
                        if (fileMap && dctx && fileOff + off + n <= mapBytes) {
                            // [a, e): the pages that hold the piece (the first and the last page of a target may still be locked for
                            // a neighbouring target's copy: then the piece is read like any other)
                            const uintptr_t lo = (uintptr_t)(fileMap + fileOff + off), hi = lo + n;
                            const uintptr_t a = lo & ~(uintptr_t)4095, e = (hi + 4095) & ~(uintptr_t)4095;
                            while (mine.size() >= 6) {  // (at most six pieces' pages locked at a time)
                                releaseDone(false);
                                if (mine.size() >= 6) std::this_thread::sleep_for(std::chrono::microseconds(100));
                            }
                            bool sent = false;
                            if (pjb_host_register((void*)a, (size_t)(e - a)) == PJB_OK) {
                                int64_t ticket = 0;
                                if (pjb_bam_piece(dctx, seq, (const uint8_t*)lo, (int64_t)n, &ticket) != PJB_OK) {
                                    readError = std::string("pjb_bam_piece: ") + pjb_last_error(dctx);
                                    releaseDone(true);
                                    (void)pjb_host_unregister((void*)a);
                                    break;
                                }
                                mine.push_back(Mine{ticket, nullptr, (void*)a, (size_t)(e - a)});
                                g_prof.event(ta0, HostProfile::now(), "worker map piece tid " + std::to_string(seq) + " " + std::to_string(n >> 20) + " MB");
                                releaseDone(false);
                                sent = true;
                            }
                            if (sent) continue;
                            // (a piece that cannot be registered -- its first page still locked for a copy in flight -- is read)
                        }
                        uint8_t* buf = nullptr;
                        if (dctx) {
                            while (!(buf = pinnedPool->tryAcquire(piece))) {  // (this thread's finished copies may be what the ring waits for)
                                releaseDone(false);
                                std::this_thread::sleep_for(std::chrono::microseconds(100));
                            }
                        } else
                            buf = pinnedPool->acquire(piece);
                        const double ta1 = HostProfile::now();
                        if (ta1 - ta0 > 1e-3) g_prof.event(ta0, ta1, "worker ring wait tid " + std::to_string(seq));
                        if (!buf) {
                            readError = "out of page-locked memory for the file pieces";
                            break;
                        }
                        try {
                            reader.readSpan(fileOff + off, n, buf, readThreads);
                            g_prof.event(ta1, HostProfile::now(), "worker read piece tid " + std::to_string(seq) + " " + std::to_string(n >> 20) + " MB");
                        } catch (const std::exception& e) {
                            pinnedPool->release(buf);
                            readError = e.what();
                            break;
                        }
                        if (dctx) {
                            int64_t ticket = 0;
                            const double tp0 = HostProfile::now();
                            const int prc = pjb_bam_piece(dctx, seq, buf, (int64_t)n, &ticket);
                            if (HostProfile::now() - tp0 > 1e-3) g_prof.event(tp0, HostProfile::now(), "worker bam_piece call tid " + std::to_string(seq));
                            if (prc != PJB_OK) {
                                readError = std::string("pjb_bam_piece: ") + pjb_last_error(dctx);
                                releaseDone(true);  // (a failing piece call has waited for the upload stream)
                                pinnedPool->release(buf);
                                break;
                            }
                            mine.push_back(Mine{ticket, buf, nullptr, 0});
                            releaseDone(false);
                            continue;
                        }
                        DeviceThread::Cmd c;
                        c.kind = DeviceThread::Cmd::BAMPIECE;
                        c.tid = seq;
                        c.bamBytes = buf;
                        c.bamSize = n;
                        c.bamPool = pinnedPool.get();
                        device.push(std::move(c));
                    }
                    std::promise<int64_t> got;
                    std::future<int64_t> f = got.get_future();
                    DeviceThread::Cmd c;
                    c.kind = DeviceThread::Cmd::BAMEND;  // (after a read error: fails with "n of m bytes arrived" and drops the staging)
                    c.tid = seq;
                    c.bamFirst = firstU;
                    c.bamDone = &got;
                    const double tq0 = HostProfile::now();
                    device.push(std::move(c));
                    if (HostProfile::now() - tq0 > 1e-3) g_prof.event(tq0, HostProfile::now(), "worker BAMEND push tid " + std::to_string(seq));
                    leave.now();  // the next target's pieces cross while this one is inflated and parsed
                    raisePiecesGone();
                    // (The push waits while the device thread's queue is full -- 0.4 s over a run, up to 0.13 s at a time -- and the slot
                    // stays taken meanwhile.  Releasing the slot before the push and a queue of 32 were blamed in round 3 for runs that
                    // stood still for a second or two (profiles/r03v_e2e_scheduling_ab.txt); those were runs right behind another
                    // process (profiles/r06_e2e_pause.txt).  Measured again with a pause before every run, neither changes the wall:
                    // medians 1.82 - 1.88 s for all four combinations, profiles/r06_e2e_slot_turnover.txt.  Left as it was.)
                    // (the ring gets this target's buffers back as their copies complete, not when its records are parsed)
                    while (f.wait_for(std::chrono::microseconds(200)) != std::future_status::ready) releaseDone(false);
                    any = f.get() > 0;
                    releaseDone(true);  // (pjb_bam_end has waited for the copies)
                    if (!readError.empty()) throw bam::BamException(readError);
                    nb = 0;  // (handled)
                }
                if (nb && reader.regionSpan(seq, fileOff, nb, firstU)) {
                    if (!bytes) bytes = (uint8_t*)bam::bigAlloc(nb + 64);
                    try {
                        reader.readSpan(fileOff, nb, bytes, innerThreads);
                    } catch (...) {
                        if (pool) pool->release(bytes);
                        else bam::bigFree(bytes);
                        throw;
                    }
                }
                if (bytes) {
                    std::promise<int64_t> got;
                    std::future<int64_t> f = got.get_future();
                    DeviceThread::Cmd c;
                    c.kind = DeviceThread::Cmd::BAM;
                    c.tid = seq;
                    c.bamBytes = bytes;
                    c.bamSize = nb;
                    c.bamFirst = firstU;
                    c.bamPool = pool;
                    c.bamDone = &got;
                    device.push(std::move(c));
                    any = f.get() > 0;
                }
            } else if (innerThreads > 1) {
                reader.decodeRegionParallel(seq, innerThreads, batchRecords, send);
            } else {
                reader.setRegion(seq);
                bam::ReadBatch b;
                while (true) {
                    b.clear();
                    b.reserve(batchRecords);
                    if (!reader.nextBatch(b, batchRecords)) break;
                    send(b);
                }
            }
        } catch (const std::exception& e) {
            decodeError = e.what();
        }
    });
    std::string genomeError;
    double t_genome = 0;
    try {
        if (genomeLate) piecesGoneF.wait();
        const double t0 = HostProfile::now();
        // large runs: the record's bytes go to the device as they are in the file (a pread into a page-locked buffer; the
        // device takes the line terminators out) -- parsing 3 GB of FASTA on the host was 6 core-seconds at the very moment
        // the file pieces of the first targets want the cores
        bool uploaded = false;
        bam::GenomeMapper::RawSpan span;
        if (genomePool && gmap.rawSpan(name, span) && span.length == refs->at((size_t)seq)->length && span.bytes > 0) {
            uint8_t* buf = genomePool->acquire(span.bytes);
            if (buf) {
                const double t1 = HostProfile::now();
                bool whole = false;
                try {
                    whole = gmap.readRaw(span, buf, std::max(innerThreads, 4));
                } catch (...) {
                    genomePool->release(buf);
                    throw;
                }
                g_prof.event(t0, t1, "worker genome buffer wait tid " + std::to_string(seq));
                g_prof.event(t1, HostProfile::now(), "worker genome raw read tid " + std::to_string(seq));
                if (!whole) {  // the file ends before the span the index describes: the record is not laid out that way
                    genomePool->release(buf);
                } else {
                    std::promise<bool> ok;
                    std::future<bool> f = ok.get_future();
                    DeviceThread::Cmd c;
                    c.kind = DeviceThread::Cmd::GENOME;
                    c.tid = seq;
                    c.raw = buf;
                    c.rawBytes = span.bytes;
                    c.lineBases = span.lineBases;
                    c.lineWidth = span.lineWidth;
                    c.genomeLen = span.length;
                    c.rawPool = genomePool.get();
                    c.rawDone = &ok;
                    device.push(std::move(c));
                    uploaded = f.get();
                    t_genome = HostProfile::now() - t0;
                }
            }
        }
        if (!uploaded) {
            std::string contig = gmap.fetchContig(name);
            t_genome = HostProfile::now() - t0;
            g_prof.event(t0, t0 + t_genome, "worker genome read tid " + std::to_string(seq));
            if ((int64_t)contig.size() != refs->at((size_t)seq)->length)
                throw JunctionBuilderException("Genome sequence " + name + " has " + std::to_string(contig.size()) +
                                               " bases but the BAM header says " + std::to_string(refs->at((size_t)seq)->length));
            DeviceThread::Cmd c;
            c.kind = DeviceThread::Cmd::GENOME;
            c.tid = seq;
            c.genome = std::move(contig);
            device.push(std::move(c));
        }
    } catch (const std::exception& e) {
        genomeError = e.what();
    }
    decoder.join();
    const double t_decoded = HostProfile::now();
    // always close the contig on the device, also after a host-side error
    auto dt = std::make_shared<DeferredTarget>();
    dt->fut = dt->done.get_future();
    dt->decodeError = decodeError;
    dt->genomeError = genomeError;
    dt->any = any;
    dt->name = name;
    dt->t_begin = t_begin;
    dt->t_decoded = t_decoded;
    dt->t_blocked = t_blocked;
    dt->t_genome = t_genome;
    {
        DeviceThread::Cmd c;
        c.kind = DeviceThread::Cmd::FINISH;
        c.tid = seq;
        c.done = &dt->done;
        if (device.grouped()) c.seen = &dt->seen;
        device.push(std::move(c));
    }
    if (device.grouped()) {
        dt->seen.get_future().wait();  // (the batches queued before the FINISH name this frame's buffers)
        // the target's chain is queued when the last member of its group has been asked for: this worker goes on to its next target
        // (waiting here, a worker would hold the thread the group's other members need) and findJunctions takes the result later
        std::lock_guard<std::mutex> lk(deferredMu);
        deferredTargets[(size_t)seq] = dt;
        return;
    }
    completeTarget(seq, *dt);
}

// what a worker does once its target's chain has been collected (or failed)
void JunctionBuilder::completeTarget(int32_t seq, DeferredTarget& dt) {
    RegionResult& res = results[(size_t)seq];
    ContigDone d;
    std::string finishError;
    try {
        d = dt.fut.get();
    } catch (const std::exception& e) {
        finishError = e.what();
    }
    if (!dt.decodeError.empty()) throw JunctionBuilderException(dt.decodeError);
    if (!dt.genomeError.empty()) throw JunctionBuilderException(dt.genomeError);
    if (!finishError.empty()) throw JunctionBuilderException(finishError);
    if (!dt.any) return;
    res.js.appendRows(d.rows.data(), d.rows.size());
    res.rowBase = d.rowBase;
    res.splicedCount = d.rr.spliced;
    res.unsplicedCount = d.rr.unspliced;
    res.sumQueryLengths = d.rr.sum_len;
    res.minQueryLength = d.rr.min_len;
    res.maxQueryLength = d.rr.max_len;
    if (g_prof.on) {
        const double t_end = HostProfile::now();
        std::lock_guard<std::mutex> lk(g_prof.mu);
        cerr << "[host profile] " << dt.name << ": total " << (t_end - dt.t_begin) << " s = decode (incl. queueing) " << (dt.t_decoded - dt.t_begin)
             << " (of which blocked on the device queue " << dt.t_blocked << ") + finish/rows " << (t_end - dt.t_decoded)
             << "; genome read " << dt.t_genome << endl;
    }
}

void JunctionBuilder::findJunctions() {
    WallTimer timer;
    results.clear();
    results.resize(refs->size());
    if (!deviceCount.valid()) deviceCount = std::async(std::launch::async, [] { return pjb_device_count(); }).share();
    // `threads` host threads in total: one worker per target sequence in flight, the rest decode
    // inside the targets (a single big contig still uses every thread)
    int withReads = 0;
    {
        BamReader probe(prepData.getSortedBamFilePath());
        probe.open(useCsi);
        for (size_t i = 0; i < refs->size(); i++) withReads += probe.hasAlignments((int32_t)i) ? 1 : 0;
    }
    const int total = std::max<int>(1, hostThreads > 0 ? hostThreads : threads);
    const int nthreads = std::max(1, std::min(total, std::max(1, withReads)));
    innerThreads = std::max(1, total / nthreads);
    cout << "Creating " << nthreads << " threads, each with BAM and genome indicies loaded ...";
    cout.flush();
    std::vector<int32_t> order;  // longest targets first: better balance across workers
    for (size_t i = 0; i < refs->size(); i++) {
        results[i].js.setRefs(refs);
        results[i].name = refs->at(i)->name;
        order.push_back((int32_t)i);
    }
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return refs->at((size_t)a)->length > refs->at((size_t)b)->length; });
    // ... except that a smaller one goes first: the device has nothing to do until a target's last byte has crossed, and the
    // largest target takes longest to cross (first inflate 0.23 s after the contexts were ready; the smallest ones still end
    // the run: a short tail)
    if (order.size() >= 5) {
        const size_t k = order.size() * 4 / 5;
        const int32_t small = order[k];
        order.erase(order.begin() + (long)k);
        order.insert(order.begin(), small);
    }
    transferRanks.assign(refs->size(), 1 << 30);
    for (size_t k = 0; k < order.size(); k++) transferRanks[(size_t)order[k]] = (int)k;
    std::mutex mu;
    size_t nextTask = 0;
    std::string firstError;
    std::vector<int32_t> lens;
    for (auto& r : *refs) lens.push_back(r->length);
    // large inputs: page-locked buffers for the file bytes (allocating them costs ~0.15 s per GB once, so small runs
    // keep the pageable path whose staging copy is cheaper than that)
    pinnedPool.reset();
    genomePool.reset();
    directPieces = true;
    int transferSlots = 0;
    if (deviceIngest) {
        struct stat bst;
        uint64_t minFile = 8ull << 30;
        pieceMinTarget = (size_t)64 << 20;
        size_t nbuf = 12, pieceBytes = (size_t)64 << 20;  // 0.75 GB of page-locked memory in all (64 MB pieces measured best of 32 / 64 / 128)
        if (const char* e = getenv("PORTCULLIS_PINNED_BUFFERS")) nbuf = (size_t)std::max(2, atoi(e));
        if (const char* e = getenv("PORTCULLIS_PIECE_MB")) pieceBytes = (size_t)std::max(1, atoi(e)) << 20;
        if (const char* e = getenv("PORTCULLIS_PIECE_BYTES")) {  // (tests: small files in small pieces)
            pieceBytes = (size_t)std::max(64, atoi(e));
            minFile = 0;
            pieceMinTarget = 0;
        }
        if (stat(prepData.getSortedBamFilePath().c_str(), &bst) == 0 && (uint64_t)bst.st_size >= minFile) {
            pinnedPool.reset(new PinnedPool(nbuf, pieceBytes));
            genomePool.reset(new PinnedPool(3));  // (the FASTA records' bytes go up as they are: pjb_upload_contig_fasta)
            // Targets whose pieces are on their way at once, over all contexts (PORTCULLIS_TRANSFER_SLOTS; 0: no limit), each read by
            // PORTCULLIS_READ_THREADS threads.  FEW readers: four threads pread 31.6 GB/s out of the page cache into page-locked
            // buffers, eight 25.0, fifteen 24.8 (profiles/r03ap_register_probe.txt), and the run with 2 x 2 readers takes 2.03 s
            // where 3 x 5 took 2.3 - 2.6 (profiles/r03aq_e2e_readers.txt).
            // (Round 6, every run behind a 3 s pause -- the stalls that made three slots look unstable were the process before the run,
            // profiles/r06_e2e_pause.txt --: three targets in transfer against two, medians of 8 / 8 / 14 runs in three calls: 1.61 / 1.63 /
            // 1.71 s against 1.68 / 1.64 / 1.68 s, and 1.69 s against 1.63 - 1.65 in the bench's own leg; four: as three.  A wash with the
            // wider spread on three's side: two stays.  profiles/r06_e2e_retune*.txt)
            transferSlots = 2;
            if (const char* e = getenv("PORTCULLIS_TRANSFER_SLOTS")) transferSlots = atoi(e);
        }
    }
    // one device thread per GPU in use; decode workers are assigned round robin
    const int ndevWanted = devices > 0 ? devices : 0;
    std::vector<std::unique_ptr<DeviceThread>> deviceThreads;
    auto deviceFor = [&](int w) -> DeviceThread& {
        std::lock_guard<std::mutex> lk(mu);
        if (deviceThreads.empty()) {
            // the device count is only known once HIP is up; until then assume one GPU per requested device
            int nd = 1;
            if (ndevWanted > 1 || devices == 0) {
                int visible = deviceCount.get();
                if (getenv("PORTCULLIS_DEVICES_SHARE_GPU") && visible > 0 && ndevWanted > 0) visible = ndevWanted;
                nd = std::max(1, std::min(ndevWanted > 0 ? ndevWanted : visible, std::min(visible, nthreads)));
            }
            // two contexts (device threads, streams) per GPU: while one target's kernels run, the other target's
            // file bytes and genome cross PCIe
            // (With the file pieces streaming through one context -- whose inflates run on their own streams beside
            // everything else -- a second context only adds contention for the runtime's locks: 2.9-3.0 s against 3.2 s.)
            int per = pinnedPool ? 1 : 2;
            if (const char* e = getenv("PORTCULLIS_CTX_PER_GPU")) per = std::max(1, atoi(e));
            per = std::max(1, std::min(per, nthreads / nd));
            if (extra) nd = per = 1;  // the name multiplicities and the depth hand-over between targets are file-wide: one context
            // The chain plan.  ONE context serves every target (the default for large inputs): the targets that hold alignments, in index
            // order, are finished in the groups pjb_plan_groups makes of them -- what bench.py's step does.  Several contexts take their
            // targets as the workers come (no telling which context a target goes to), and --extra contexts do not take groups: a
            // chain per target.  PORTCULLIS_CHAIN_PLAN=targets | groups overrides (groups: only with one context); PORTCULLIS_GROUP_BASES
            // sets the bases of a group (tests: small genomes in several groups).
            std::vector<std::vector<int32_t>> chainPlan;
            const char* planEnv = getenv("PORTCULLIS_CHAIN_PLAN");
            const bool wantGroups = planEnv ? std::string(planEnv) == "groups" : (pinnedPool != nullptr);
            if (wantGroups && nd * per == 1 && !extra) {
                std::vector<int32_t> with;
                {
                    BamReader probe(prepData.getSortedBamFilePath());
                    probe.open(useCsi);
                    for (size_t i = 0; i < refs->size(); i++)
                        if (probe.hasAlignments((int32_t)i)) with.push_back((int32_t)i);
                }
                std::vector<int32_t> groupOf(with.size(), 0);
                int64_t gb = 0;
                if (const char* e = getenv("PORTCULLIS_GROUP_BASES")) gb = atoll(e);
                const int ng = pjb_plan_groups(lens.data(), with.data(), (int32_t)with.size(), gb, groupOf.data());
                if (ng > 0) {
                    chainPlan.resize((size_t)ng);
                    for (size_t k = 0; k < with.size(); k++) chainPlan[(size_t)groupOf[k]].push_back(with[k]);
                }
            }
            for (int k = 0; k < per; k++)
                for (int d = 0; d < nd; d++)
                    deviceThreads.emplace_back(new DeviceThread(d, orientation, strandSpecific, lens, deviceCount, extra, chainPlan));
            for (size_t k = 0; k < deviceThreads.size(); k++) deviceThreads[k]->lane = (int)k;
            if (pinnedPool && transferSlots > 0) {
                const int perLane = std::max(1, transferSlots / (int)deviceThreads.size());
                pinnedPool->setTransferSlots(perLane);
                pinnedPool->readThreads = std::max(1, std::min(2, total / (perLane * (int)deviceThreads.size())));
                if (const char* e = getenv("PORTCULLIS_READ_THREADS")) pinnedPool->readThreads = std::max(1, atoi(e));  // (threads per target in transfer)
                if (const char* e = getenv("PORTCULLIS_REGISTER_SLOT")) pinnedPool->registerSlot = atoi(e);
            }
        }
        return *deviceThreads[(size_t)w % deviceThreads.size()];
    };
    auto worker = [&](int w) {
        try {
            GenomeMapper gmap(prepData.getGenomeFilePath());
            gmap.loadFastaIndex();
            BamReader reader(prepData.getSortedBamFilePath());
            reader.open(useCsi);
            reader.setNameHashes(extra);
            DeviceThread& dev = deviceFor(w);
            if (pinnedPool) dev.waitReady();
            while (true) {
                int32_t tid;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (nextTask >= order.size() || !firstError.empty()) break;
                    tid = order[nextTask++];
                }
                findJuncs(dev, reader, gmap, tid);
            }
        } catch (const std::exception& e) {
            std::lock_guard<std::mutex> lk(mu);
            if (firstError.empty()) firstError = e.what();
        }
    };
    const double t_workers0 = HostProfile::now();
    g_prof.mark("workers start");
    cout << " done." << endl;
    cout << "Finding junctions and calculating basic metrics:" << endl;
    cout << " - Queueing " << refs->size() << " target sequences for processing in the thread pool" << endl;
    cout << " - Processing: " << endl;
    std::vector<std::thread> pool;
    // (PORTCULLIS_WORKER_PER_TARGET=1, an experiment: a worker belongs to its target until the target's rows are back and
    // mostly waits; one worker per target keeps the file moving while every other worker waits for the device.  Not the
    // default: see the note at the transfer gate.)
    const int nworkers = nthreads;
    deferredTargets.assign(refs->size(), nullptr);
    for (int w = 0; w < nworkers; w++) pool.emplace_back(worker, w);
    for (auto& t : pool) t.join();
    // group chains: every target has been asked for -- what still waits for the rest of its group goes now, then the results are taken
    for (auto& dth : deviceThreads)
        if (dth->grouped()) {
            DeviceThread::Cmd c;
            c.kind = DeviceThread::Cmd::FLUSH;
            dth->push(std::move(c));
        }
    {
        // A few threads take the targets in index order: a target's rows become Junction objects as soon as its chain has been collected
        // -- the first groups' while the last group's chain still runs -- instead of all 250 000 behind the last chain on this thread
        // (50 ms of the run).  The first error in index order is the one reported, as before.
        std::vector<std::string> errs(deferredTargets.size());
        std::atomic<size_t> next{0};
        auto work = [&] {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= deferredTargets.size()) break;
                if (!deferredTargets[i]) continue;
                try {
                    completeTarget((int32_t)i, *deferredTargets[i]);
                } catch (const std::exception& e) {
                    errs[i] = e.what();
                    if (errs[i].empty()) errs[i] = "unknown error";
                }
                deferredTargets[i].reset();
            }
        };
        size_t waiting = 0;
        for (auto& d : deferredTargets) waiting += d ? 1 : 0;
        const size_t nt = std::min<size_t>(waiting, 16);
        if (nt <= 1) {
            work();
        } else {
            std::vector<std::thread> th;
            for (size_t t = 0; t < nt; t++) th.emplace_back(work);
            for (auto& x : th) x.join();
        }
        for (auto& e : errs)
            if (!e.empty() && firstError.empty()) firstError = e;
    }
    if (extra && firstError.empty() && !deviceThreads.empty()) {
        // calcExtraMetrics (src/junction_builder.cc:293-312): multiple mapping score, flanking alignments, coverage
        cout << "Calculating extra junction metrics:" << endl;
        try {
            std::promise<std::vector<pjb_extra_row>> got;
            std::future<std::vector<pjb_extra_row>> f = got.get_future();
            DeviceThread::Cmd c;
            c.kind = DeviceThread::Cmd::EXTRA;
            c.extraDone = &got;
            deviceThreads[0]->push(std::move(c));
            const std::vector<pjb_extra_row> xr = f.get();
            for (auto& res : results) {
                const JunctionList& jl = res.js.getJunctions();
                for (size_t k = 0; k < jl.size(); k++) {
                    const pjb_extra_row& x = xr.at(res.rowBase + k);
                    jl[k]->setMultipleMappingScore(x.mm_score);
                    jl[k]->setCoverage(x.coverage);
                    jl[k]->setNbUpstreamFlankingAlignments(x.up_aln);
                    jl[k]->setNbDownstreamFlankingAlignments(x.down_aln);
                }
            }
        } catch (const std::exception& e) {
            firstError = e.what();
        }
    }
    // Everything the device produced is on the host.  Taking the contexts down (every device buffer), and the page-locked
    // rings with them, is 0.1-0.3 s of runtime and driver work that nothing waits for: it runs beside the merge and the
    // writers instead of before them (PORTCULLIS_SYNC_TEARDOWN=1: as before).
    // (a process that will leave through exit handlers must not have a thread inside the runtime by then)
    if (!backgroundTeardown || getenv("PJB_NORMAL_EXIT") || !firstError.empty()) {
        deviceThreads.clear();  // joins the device threads (destroys the contexts)
    } else {
        auto dts = std::make_shared<std::vector<std::unique_ptr<DeviceThread>>>(std::move(deviceThreads));
        auto pp = std::move(pinnedPool);
        auto gp = std::move(genomePool);
        deviceThreads.clear();
        // (the page-locked rings first: giving 1.6 GB of them back takes ~0.15 s here or in the kernel when the process leaves, a context's
        // device memory a third of that -- tools/debug/exit_probe.cc; PORTCULLIS_TEARDOWN_CONTEXTS_FIRST=1: the order until round 6.
        // Giving the rings back EARLIER -- behind the last target's bytes, beside the last chains -- was measured twice and made the run
        // longer both times: the unregister calls hold up the chains' launches, profiles/r06_e2e_pools.txt, r06_e2e_early_free.txt)
        const bool contextsFirst = getenv("PORTCULLIS_TEARDOWN_CONTEXTS_FIRST") != nullptr;
        // (moved into the threads: the last owner frees, and that must not be this thread)
        if (contextsFirst || getenv("PORTCULLIS_TEARDOWN_ONE_THREAD")) {
            std::thread([dts = std::move(dts), pp = std::move(pp), gp = std::move(gp), contextsFirst]() mutable {
                if (contextsFirst) dts->clear();
                pp.reset();
                gp.reset();
                dts->clear();
            }).detach();
        } else {  // the rings on one thread, the contexts on another (1.67 against 1.70 s on one thread, profiles/r06_e2e_host_tail2.txt)
            std::thread([pp = std::move(pp), gp = std::move(gp)]() mutable {
                pp.reset();
                gp.reset();
            }).detach();
            std::thread([dts = std::move(dts)]() mutable { dts->clear(); }).detach();
        }
    }
    if (!firstError.empty()) throw JunctionBuilderException(firstError);
    const double t_workers1 = HostProfile::now();
    g_prof.mark("workers and device threads done");
    cout << " - All threads completed." << endl << " - Combining results from threads." << endl << endl;
    uint64_t unsplicedCount = 0, splicedCount = 0, sumQueryLengths = 0;
    int32_t minQueryLength = INT32_MAX, maxQueryLength = 0;
    cout << std::left << std::setw(12) << "Sequence"
         << "\t" << std::right << std::setw(12) << "unspliced"
         << "\t" << std::right << std::setw(12) << "spliced"
         << "\t" << std::right << std::setw(12) << "total" << endl;
    {
        size_t total = 0;
        for (auto& res : results) total += res.js.getJunctions().size();
        junctionSystem.reserve(total);
    }
    for (auto& res : results) {
        junctionSystem.absorb(res.js);  // (the per-target systems keep their lists -- --extra walks them -- not their maps)
        unsplicedCount += res.unsplicedCount;
        splicedCount += res.splicedCount;
        sumQueryLengths += res.sumQueryLengths;
        minQueryLength = std::min(minQueryLength, res.minQueryLength);
        maxQueryLength = std::max(maxQueryLength, res.maxQueryLength);
        cout << std::left << std::setw(12) << res.name << "\t" << std::right << std::setw(12) << res.unsplicedCount << "\t"
             << std::right << std::setw(12) << res.splicedCount << "\t" << std::right << std::setw(12)
             << res.splicedCount + res.unsplicedCount << endl;
    }
    cout << endl << "Sorting and reindexing merged junctions...";
    cout.flush();
    junctionSystem.sort();
    junctionSystem.index();
    cout << " done." << endl << endl;
    const uint64_t totalAlignments = splicedCount + unsplicedCount;
    const double meanQueryLength = (double)sumQueryLengths / (double)totalAlignments;
    junctionSystem.setQueryLengthStats(minQueryLength, meanQueryLength, maxQueryLength);
    cout << "Final stats:" << endl
         << " - Processed " << totalAlignments << " alignments." << endl
         << " - Alignment query length statistics: min: " << minQueryLength << "; mean: " << meanQueryLength
         << "; max: " << maxQueryLength << ";" << endl
         << " - Found " << junctionSystem.size() << " junctions from " << splicedCount << " spliced alignments." << endl
         << " - Found " << unsplicedCount << " unspliced alignments." << endl;
    const double t_merge1 = HostProfile::now();
    if (junctionSystem.size() > 1) {
        cout << " - Calculating junctions stats that require comparisons with other junctions...";
        cout.flush();
        junctionSystem.calcJunctionStats();
        cout << " done." << endl;
    }
    if (g_prof.on)
        cerr << "[host profile] workers " << (t_workers1 - t_workers0) << " s, merge+sort+index " << (t_merge1 - t_workers1)
             << " s, calcJunctionStats " << (HostProfile::now() - t_merge1) << " s" << endl;
}

// command line of `portcullis junc` (src/junction_builder.cc:359-454); a small hand-rolled parser
// replaces boost::program_options.
int JunctionBuilder::main(int argc, char* argv[]) {
    std::string prepDir, output = DEFAULT_JUNC_OUTPUT, source = DEFAULT_JUNC_SOURCE, ori = "UNKNOWN", strand = "UNKNOWN";
    int threads = DEFAULT_JUNC_THREADS, devices = 0;
    size_t batch = 0;
    std::string ingest;
    bool extra = false, separate = false, useCsi = false, exonGff = false, intronGff = false, verbose = false, help = false;
    auto need = [&](int& i) -> std::string {
        if (i + 1 >= argc) throw JunctionBuilderException(std::string("Missing value for option ") + argv[i]);
        return argv[++i];
    };
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-o" || a == "--output") output = need(i);
        else if (a == "-t" || a == "--threads") threads = atoi(need(i).c_str());
        else if (a == "--orientation") ori = need(i);
        else if (a == "--strandedness") strand = need(i);
        else if (a == "--source") source = need(i);
        else if (a == "--devices") devices = atoi(need(i).c_str());
        else if (a == "--batch") batch = (size_t)atol(need(i).c_str());
        else if (a == "--ingest") ingest = need(i);
        else if (a == "--separate") separate = true;
        else if (a == "--extra") extra = true;
        else if (a == "-c" || a == "--use_csi") useCsi = true;
        else if (a == "--exon_gff") exonGff = true;
        else if (a == "--intron_gff") intronGff = true;
        else if (a == "-v" || a == "--verbose") verbose = true;
        else if (a == "--help" || a == "-h") help = true;
        else if (!a.empty() && a[0] == '-') throw JunctionBuilderException("Unknown option: " + a);
        else prepDir = a;
    }
    if (help || prepDir.empty()) {
        cout << title() << endl << endl << description() << endl << endl << "Usage: " << usage() << endl
             << "  -o, --output <prefix>      Output prefix for files generated by this program (default " << DEFAULT_JUNC_OUTPUT << ")" << endl
             << "  -t, --threads <n>          Host decode threads; one target sequence per thread at a time" << endl
             << "      --orientation <o>      SE, FR, RF, FF or UNKNOWN" << endl
             << "      --strandedness <s>     unstranded, firststrand, secondstrand or UNKNOWN" << endl
             << "      --source <name>        Source column of the BED/GFF output (default portcullis)" << endl
             << "      --exon_gff             Also write <prefix>.junctions.exon.gff3" << endl
             << "      --intron_gff           Also write <prefix>.junctions.intron.gff3" << endl
             << "  -c, --use_csi              Use the CSI index of the prepared BAM instead of the BAI" << endl
             << "      --devices <n>          Number of GPUs to use (default: all visible)" << endl
             << "      --ingest <device|host> Where BGZF inflate and BAM record parsing run (default device; env PORTCULLIS_INGEST)" << endl
             << "  -v, --verbose" << endl;
        return help ? 0 : 1;
    }
    WallTimer timer;
    cout << "Running portcullis in junction builder mode" << endl << "------------------------------------------" << endl << endl;
    // deliberately never destroyed: the program ends right after process() and freeing every junction object
    // one by one would only delay that
    JunctionBuilder& jb = *new JunctionBuilder(prepDir, output);
    jb.setThreads((uint16_t)std::max(1, threads));
    jb.setExtra(extra);
    jb.setSeparate(separate);
    jb.setSource(source);
    jb.setUseCsi(useCsi);
    jb.setOutputExonGFF(exonGff);
    jb.setOutputIntronGFF(intronGff);
    jb.setVerbose(verbose);
    jb.setOrientation(bam::orientationFromString(ori));
    jb.setStrandSpecific(bam::strandednessFromString(strand));
    if (devices > 0) jb.setDevices(devices);
    if (batch > 0) jb.setBatchRecords(batch);
    if (!ingest.empty()) {
        if (ingest != "host" && ingest != "device") throw JunctionBuilderException("--ingest takes host or device");
        jb.setDeviceIngest(ingest == "device");
    }
    jb.backgroundTeardown = true;  // (this program leaves through _exit)
    jb.process();
    if (getenv("PJB_PROFILE_HOST")) cerr << "[host profile] main: " << timer.elapsed() << " s until process() returned" << endl;
    return 0;
}

}  // namespace portcullis
