// BamFilter: the `bamfilt` stage.  Flow and console output follow src/bam_filter.cc:152-247 of the reference; the
// decision per alignment is made on the device (pjb_filter_batch), one batch of (pos, CIGAR) per run of records of a
// target.  Kept records are written byte for byte as they were read (like the reference: BamWriter::write emits the
// untouched bam1_t, lib/src/bam_writer.cc:58-60, also for "clipped" multiply spliced reads).
#include <portcullis/bam_filter.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <map>
#include <future>
#include <memory>
#include <sys/stat.h>

#include <portcullis/bam/bam_reader.hpp>
#include <portcullis/bam/bam_writer.hpp>
#include <portcullis/bam/phase_pool.hpp>
#include <portcullis/junction_system.hpp>

#include "../../../include/portcullis_amd.h"

namespace portcullis {

using std::cerr;
using std::cout;
using std::endl;

ClipMode clipFromString(const std::string& cm) {
    std::string u = cm;
    for (auto& ch : u) ch = (char)toupper((unsigned char)ch);
    if (u == "HARD") return ClipMode::HARD;
    if (u == "SOFT") return ClipMode::SOFT;
    if (u == "COMPLETE") return ClipMode::COMPLETE;
    throw BamFilterException("Unrecognised clip mode: " + cm);
}

static bool exists(const std::string& p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0;
}

BamFilter::BamFilter(const std::string& jf, const std::string& bf, const std::string& ob) : junctionFile(jf), bamFile(bf), outputBam(ob) {
    // src/bam_filter.cc:49-66
    if (!exists(junctionFile)) throw BamFilterException("Could not find junction file at: " + junctionFile);
    if (!exists(bamFile)) throw BamFilterException("Could not find BAM file at: " + bamFile);
}

static std::chrono::steady_clock::time_point g_filterEnded;  // (PORTCULLIS_PROFILE: how long the release of filter()'s locals takes)

void BamFilter::filter() {
    const bool profTop = getenv("PORTCULLIS_PROFILE") != nullptr;
    const auto tTop = std::chrono::steady_clock::now();
    struct ScopeMark {  // (PORTCULLIS_PROFILE: when the locals declared before this one start to be released)
        const char* what;
        bool on;
        std::chrono::steady_clock::time_point t0;
        ~ScopeMark() {
            if (on) fprintf(stderr, "[bamfilt profile] t=%.3f s: releasing what was declared before: %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), what);
        }
    };
    auto mark = [&](const char* what) {
        if (profTop) fprintf(stderr, "[bamfilt profile] t=%.3f s: %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - tTop).count(), what);
    };
    // (the device context comes up -- 0.2 s of runtime start -- while the junctions are read)
    pjb_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = PJB_ABI_VERSION;
    cfg.device = device;
    cfg.orientation = PJB_OR_UNKNOWN;
    cfg.strandedness = PJB_SS_UNKNOWN;
    cfg.flags = PJB_FLAG_NO_CHAINS;  // (decisions and BGZF blocks only)
    std::future<pjb_ctx*> ctxComing = std::async(std::launch::async, [cfg]() -> pjb_ctx* {
        pjb_ctx* c1 = nullptr;
        if (pjb_create(&c1, &cfg) != PJB_OK) throw BamFilterException(std::string("pjb_create: ") + pjb_last_error(nullptr));
        return c1;
    });
    struct CtxGuard {  // (an exception before the context is taken over must still wait for it and release it)
        std::future<pjb_ctx*>* f;
        ~CtxGuard() {
            try {
                if (f->valid()) pjb_destroy(f->get());
            } catch (...) {
            }
        }
    } ctxGuard{&ctxComing};
    ScopeMark sm_ctxGuard{"ctxGuard", profTop, tTop};
    cout << "Loading junctions from: " << junctionFile << endl;
    // (the junctions and the reader's index are thousands of small allocations: in the command's forked child they are not
    // taken apart one by one before the process ends -- see `leaves` below)
    std::unique_ptr<JunctionSystem> jsHold(new JunctionSystem(junctionFile));
    JunctionSystem& js = *jsHold;
    cout << " - Found " << js.size() << " junctions" << endl << endl;
    mark("junctions loaded");
    std::unique_ptr<bam::BamReader> readerHold(new bam::BamReader(bamFile));
    bam::BamReader& reader = *readerHold;
    ScopeMark sm_reader{"reader", profTop, tTop};
    reader.open(useCsi);
    mark("BAM header and index read");
    std::shared_ptr<bam::RefSeqPtrList> refs = reader.createRefList();
    js.setRefs(refs);
    {
        const size_t slash = outputBam.find_last_of('/');
        const std::string outDir = slash == std::string::npos ? "." : outputBam.substr(0, slash);
        if (!outDir.empty() && !exists(outDir) && mkdir(outDir.c_str(), 0777) != 0 && !exists(outDir))
            throw BamFilterException("Could not create output directory at: " + outDir);
    }
    // ---- the filter's junction set, per target, as sorted keys on the device
    // (the context takes 0.2 s to come up: the junction keys follow it on a thread of their own, and the first piece of the
    // file is read and inflated meanwhile -- whoever needs the context asks ctxReady)
    std::shared_future<pjb_ctx*> ctxReady = std::async(std::launch::async, [&js, &ctxComing, &mark]() -> pjb_ctx* {
        pjb_ctx* c1 = ctxComing.get();
        mark("device context is up");
        std::map<int32_t, std::vector<uint64_t>> keys;
        for (const JunctionPtr& j : js.getJunctions()) {
            const Intron& in = *j->getIntron();
            keys[in.ref.index].push_back(((uint64_t)(uint32_t)in.start << 32) | (uint64_t)(uint32_t)in.end);
        }
        for (auto& kv : keys) {
            std::sort(kv.second.begin(), kv.second.end());
            kv.second.erase(std::unique(kv.second.begin(), kv.second.end()), kv.second.end());
            if (pjb_filter_set_junctions(c1, kv.first, kv.second.data(), (int64_t)kv.second.size()) != PJB_OK) {
                const std::string msg = std::string("pjb_filter_set_junctions: ") + pjb_last_error(c1);
                pjb_destroy(c1);
                throw BamFilterException(msg);
            }
        }
        mark("junction keys on the device");
        return c1;
    }).share();
    // In the forked child of the command (main.cc) the contexts and the page-locked buffers are left to the end of the
    // process, which follows the report to the waiting side: taking them down here is 80 ms of the command's 0.7 s.
    const bool leaves = getenv("PORTCULLIS_CHILD_LEAVES") != nullptr;
    struct Closer {
        std::shared_future<pjb_ctx*> f;
        bool leaves;
        ~Closer() {
            try {
                pjb_ctx* c1 = f.get();
                if (!leaves) pjb_destroy(c1);
            } catch (...) {
            }
        }
    } closer{ctxReady, leaves};
    ScopeMark sm_closer{"closer", profTop, tTop};
    mark("context and junction keys on their way");
    cout << " - Processing alignments from: " << bamFile << endl;
    // BGZF blocks are compressed on the device (pjb_deflate_bgzf); PORTCULLIS_HOST_DEFLATE=1: by zlib on the workers.
    // The filtered file's blocks go through a context of their own: its writer compresses on a thread of its own while
    // this thread asks the first context for decisions.
    const bool hostDeflate = getenv("PORTCULLIS_HOST_DEFLATE") != nullptr;
    std::shared_future<pjb_ctx*> deflateCtx = std::async(std::launch::async, [cfg, hostDeflate]() -> pjb_ctx* {
        pjb_ctx* c2 = nullptr;
        if (!hostDeflate && pjb_create(&c2, &cfg) != PJB_OK) throw BamFilterException(std::string("pjb_create: ") + pjb_last_error(nullptr));
        return c2;
    }).share();
    struct Closer2 {
        std::shared_future<pjb_ctx*> f;
        bool leaves;
        ~Closer2() {
            try {
                pjb_ctx* c2 = f.get();
                if (!leaves) pjb_destroy(c2);
            } catch (...) {
            }
        }
    } closer2{deflateCtx, leaves};
    ScopeMark sm_closer2{"closer2", profTop, tTop};
    std::unique_ptr<bam::BamWriter> writerHold(new bam::BamWriter(outputBam, threads));
    bam::BamWriter& writer = *writerHold;
    ScopeMark sm_writer{"writer", profTop, tTop};
    auto deflateOn = [](pjb_ctx* ctx, const uint8_t* in, size_t n, size_t block, bam::ByteBuf& out, std::vector<uint32_t>& sizes) -> bool {
        const size_t nblk = (n + block - 1) / block;
        // (room for a quarter more: a piece with more blocks than any before it would move the buffer, and a page-locked
        // buffer of 200 MB takes 30 - 40 ms to make -- as long as compressing the piece)
        if (out.capacity() < nblk * 65536) out.reserve(nblk * 65536 + nblk * 16384 + (1u << 20));
        out.resize(nblk * 65536);
        sizes.resize(nblk);
        int64_t got = 0;
        if (pjb_deflate_bgzf(ctx, in, (int64_t)n, (int32_t)block, out.data(), (int64_t)out.size(), &got, sizes.data()) != PJB_OK)
            throw BamFilterException(std::string("pjb_deflate_bgzf: ") + pjb_last_error(ctx));
        out.resize((size_t)got);
        return true;
    };
    auto deviceDeflate = [deflateCtx, deflateOn](const uint8_t* in, size_t n, size_t block, bam::ByteBuf& out, std::vector<uint32_t>& sizes) -> bool {
        return deflateOn(deflateCtx.get(), in, n, block, out, sizes);
    };
    auto deviceDeflateHere = [ctxReady, deflateOn](const uint8_t* in, size_t n, size_t block, bam::ByteBuf& out, std::vector<uint32_t>& sizes) -> bool {
        return deflateOn(ctxReady.get(), in, n, block, out, sizes);  // (the two small files of --save_msrs: on this thread, this thread's context)
    };
    if (!hostDeflate) {
        writer.setBlockCompressor(deviceDeflate);
        writer.setAsyncFlush(true);
        if (const char* e = getenv("PORTCULLIS_FLUSH_BLOCKS")) writer.setFlushBlocks((size_t)std::max(1, atoi(e)));  // (tests: many hand-overs in a small file)
        if (!getenv("PORTCULLIS_PAGEABLE_BUFFERS")) {  // (page-locked: the device reads and writes the reader's and the writers' buffers by DMA)
            bam::BufferHooks hooks;
            // (small buffers stay on the heap: page-locking means waiting for the runtime to come up, and the output's header
            // is written while it does)
            hooks.alloc = [](size_t n) -> void* { return n >= ((size_t)8 << 20) ? pjb_host_alloc(n) : nullptr; };
            hooks.release = leaves ? +[](void*) {} : pjb_host_free;
            bam::setBufferHooks(hooks);
        }
    }
    struct HooksOff {
        ~HooksOff() { bam::setBufferHooks(bam::BufferHooks()); }
    } hooksOff;
    ScopeMark sm_hooksOff{"hooksOff", profTop, tTop};
    // PORTCULLIS_DEVICE_INFLATE=1: the input's blocks are inflated on the device too (pjb_inflate_bgzf, this thread's context)
    // instead of by zlib on the workers.  Not the default: bgzf_decode is a lane per block, a piece of 256 MB has 4 k blocks of
    // the 60 k the chip holds at once, and a lane needs 40 ms for its block -- 0.52 s for the configs[1] file against zlib's
    // 0.45 s on 16 threads (profiles/r03bn_bamfilt_device_inflate.txt).
    const bool deviceInflate = getenv("PORTCULLIS_DEVICE_INFLATE") != nullptr;
    if (deviceInflate)
        reader.setBlockInflater([ctxReady](const uint8_t* comp, size_t n, uint8_t* out, size_t outBytes) -> bool {
            pjb_ctx* ctx = ctxReady.get();
            int64_t got = 0;
            if (pjb_inflate_bgzf(ctx, comp, (int64_t)n, out, (int64_t)outBytes, &got) != PJB_OK)
                throw BamFilterException(std::string("pjb_inflate_bgzf: ") + pjb_last_error(ctx));
            if ((size_t)got != outBytes) throw BamFilterException("pjb_inflate_bgzf: the blocks' sizes do not add up");
            return true;
        });
    writer.open(reader.getHeaderText(), reader.getTargets());
    cout << " - Saving filtered alignments to: " << outputBam << endl;
    mark("output opened, header written");
    std::unique_ptr<bam::BamWriter> mod, unmod;
    if (saveMSRs) {
        mod.reset(new bam::BamWriter(outputBam + ".mod.bam", threads));
        unmod.reset(new bam::BamWriter(outputBam + ".unmod.bam", threads));
        mod->setWriteIndex(false);
        unmod->setWriteIndex(false);
        if (!hostDeflate) {
            mod->setBlockCompressor(deviceDeflateHere);
            unmod->setBlockCompressor(deviceDeflateHere);
        }
        mod->open(reader.getHeaderText(), reader.getTargets());
        unmod->open(reader.getHeaderText(), reader.getTargets());
        cout << " - Saving modified MSRs to: " << outputBam << ".mod.bam" << endl;
        cout << " - Saving unmodified MSRs to: " << outputBam << ".unmod.bam" << endl;
    }
    nbReadsIn = nbReadsOut = nbReadsModifiedOut = 0;
    // ---- the file in pieces of 256 MB of records (src/bam_filter.cc:190-225 visits them one by one).  Everything per
    // record is done by `threads` workers at once: BGZF inflate and finding the records (BamReader::scanRecordsParallel),
    // pulling out (target, pos, CIGAR), gathering the kept records and compressing them (BamWriter::writeRecords); the
    // decision comes from the device, one batch per run of records of a target.
    const int32_t mode = clipMode == ClipMode::HARD ? PJB_CLIP_HARD : clipMode == ClipMode::SOFT ? PJB_CLIP_SOFT : PJB_CLIP_COMPLETE;
    std::unique_ptr<bam::PhasePool> workersHold(new bam::PhasePool(threads > 1 ? threads : 0));
    bam::PhasePool& workers = *workersHold;
    ScopeMark sm_workers{"workers", profTop, tTop};
    std::vector<int32_t> tids, pos;
    std::vector<uint32_t> cigOff, cigar, runOff;
    std::vector<uint8_t> codes;
    ScopeMark sm_vectors{"vectors", profTop, tTop};
    auto le32 = [](const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); };
    const bool prof = getenv("PORTCULLIS_PROFILE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tExtract = 0, tDevice = 0, tWrite = 0, tSink0 = now(), tScan = 0;
    // The scan runs two pieces ahead of the decisions (PORTCULLIS_SCAN_AHEAD; 0: in turns) on `threads` workers of its own:
    // the first pieces are read and inflated while the device context is still coming up, the later ones while the
    // workers here extract and gather.
    mark("workers started");
    const int scanAhead = deviceInflate ? 0 /* (one context, one thread) */
                          : getenv("PORTCULLIS_SCAN_AHEAD") ? std::max(0, atoi(getenv("PORTCULLIS_SCAN_AHEAD"))) : 2;
    reader.scanRecordsParallel(threads, (size_t)256 << 20, [&](const bam::BamReader::FileChunk& fc) {
        double t0 = now();
        tScan += t0 - tSink0;
        const double tScanLast = t0 - tSink0;
        const size_t ns = fc.slices.size(), n = fc.records;
        std::vector<size_t> base(ns + 1, 0), opBase(ns + 1, 0);
        for (size_t s = 0; s < ns; s++) base[s + 1] = base[s] + fc.slices[s]->size();
        tids.resize(n);
        pos.resize(n);
        cigOff.resize(n + 1);
        codes.assign(n, 1);
        workers.run(ns, [&](size_t s) {  // (the reader checked that every record's name and CIGAR lie inside it)
            size_t ops = 0, i = base[s];
            for (uint64_t off : *fc.slices[s]) {
                const uint8_t* r = fc.data + off;
                tids[i] = (int32_t)le32(r + 4);
                pos[i] = (int32_t)le32(r + 8);
                ops += (uint32_t)r[16] | ((uint32_t)r[17] << 8);
                i++;
            }
            opBase[s + 1] = ops;
        });
        for (size_t s = 0; s < ns; s++) opBase[s + 1] += opBase[s];
        if (opBase[ns] > 0xfffffff0ull) throw BamFilterException("Too many CIGAR operations in one piece of the file");
        cigar.resize(opBase[ns] + 1);
        cigOff[n] = (uint32_t)opBase[ns];
        workers.run(ns, [&](size_t s) {
            size_t o = opBase[s], i = base[s];
            for (uint64_t off : *fc.slices[s]) {
                const uint8_t* r = fc.data + off;
                const uint32_t l_name = r[12], n_cig = (uint32_t)r[16] | ((uint32_t)r[17] << 8);
                cigOff[i++] = (uint32_t)o;
                memcpy(&cigar[o], r + 36 + l_name, 4ull * n_cig);  // (little-endian host, like everything around the C ABI)
                o += n_cig;
            }
        });
        tExtract += now() - t0;
        t0 = now();
        // runs of one target: the device answers for each
        for (size_t a = 0; a < n;) {
            size_t b = a + 1;
            while (b < n && tids[b] == tids[a]) b++;
            if (tids[a] >= 0) {
                const uint32_t o0 = cigOff[a];
                const uint32_t* co = cigOff.data() + a;
                if (o0) {  // the batch's offsets start at 0
                    runOff.resize(b - a + 1);
                    for (size_t k = 0; k <= b - a; k++) runOff[k] = cigOff[a + k] - o0;
                    co = runOff.data();
                }
                pjb_batch pb;
                memset(&pb, 0, sizeof pb);
                pb.n_reads = (int64_t)(b - a);
                pb.pos = pos.data() + a;
                pb.cig_off = co;
                pb.cigar = cigar.data() + o0;
                pjb_ctx* ctx = ctxReady.get();
                if (pjb_filter_batch(ctx, tids[a], &pb, mode, codes.data() + a) != PJB_OK)
                    throw BamFilterException(std::string("pjb_filter_batch: ") + pjb_last_error(ctx));
            }
            a = b;
        }
        size_t out = 0, modified = 0;
        for (size_t i = 0; i < n; i++) {
            out += codes[i] != 0;
            modified += codes[i] == 3;
        }
        nbReadsIn += n;
        nbReadsOut += out;
        nbReadsModifiedOut += modified;
        tDevice += now() - t0;
        t0 = now();
        writer.writeRecords(fc.data, fc.slices, codes.data(), 0, workers);
        if (saveMSRs && modified) {
            mod->writeRecords(fc.data, fc.slices, codes.data(), 3, workers);
            unmod->writeRecords(fc.data, fc.slices, codes.data(), 3, workers);
        }
        tWrite += now() - t0;
        if (prof && getenv("PORTCULLIS_PROFILE_PIECES"))
            fprintf(stderr, "[piece] %zu records: sink %.3f .. %.3f (writer part from %.3f)\n", n, fmod(tSink0 + (tScanLast), 1000.0), fmod(now(), 1000.0), fmod(t0, 1000.0));
        tSink0 = now();
    }, scanAhead);
    if (prof)
        fprintf(stderr, "[bamfilt profile] read + inflate + find records (or waiting for them) %.3f s, extract %.3f s, device decisions %.3f s, write %.3f s\n", tScan, tExtract,
                tDevice, tWrite);
    mark("last piece handed to the writer");
    reader.close();
    writer.close();
    mark("output closed (last blocks, EOF block, .bai)");
    if (saveMSRs) {
        mod->close();
        unmod->close();
    }
    cout << "done." << endl;
    const uint32_t diff = (uint32_t)(nbReadsIn - nbReadsOut);
    cout << "Filtered out " << diff << " alignments.  In: " << nbReadsIn << "; Out: " << nbReadsOut << " (Modified: " << nbReadsModifiedOut
         << ");" << endl
         << endl;
    cout << "Indexing:" << endl << " - filtered alignments ... done." << endl;  // the .bai was written by BamWriter::close
    if (leaves) {
        (void)jsHold.release();
        (void)readerHold.release();
        (void)writerHold.release();  // (closed above)
        (void)workersHold.release();
    }
    mark(leaves ? "filter() ends (the command's child: what it holds is left to the end of the process)" : "filter() ends (what it holds is released next)");
    if (profTop) g_filterEnded = std::chrono::steady_clock::now();
}

std::string BamFilter::helpMessage() {
    return "Portcullis BAM Filter Mode Help.\n\n"
           "Removes alignments associated with bad junctions from BAM file\n\n"
           "Usage: portcullis_amd bamfilt [options] <junction-file> <bam-file>\n\n"
           "Options:\n"
           "  -o [ --output ] arg (=filtered.bam)  Output BAM file generated by this program.\n"
           "  -c [ --clip_mode ] arg (=HARD)       How to clip reads associated with bad junctions: HARD, SOFT or COMPLETE\n"
           "  -m [ --save_msrs ]                   Whether or not to output modified MSRs to a separate file.\n"
           "  --use_csi                            Whether to use CSI indexing rather than BAI indexing (input side).\n"
           "  -t [ --threads ] arg (=1)            Threads that compress the output.\n"
           "  -v [ --verbose ]                     Print extra information\n"
           "  --help                               Produce help message\n";
}

int BamFilter::main(int argc, char* argv[]) {
    std::string junctionFile, bamFile, outputBam = "filtered.bam", clip = "HARD";
    bool saveMSRs = false, useCsi = false, verbose = false, help = false;
    int threads = 1;
    std::vector<std::string> positional;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto need = [&](const char* name) -> std::string {
            if (i + 1 >= argc) throw BamFilterException(std::string("Option ") + name + " needs a value");
            return argv[++i];
        };
        if (a == "-o" || a == "--output") outputBam = need("--output");
        else if (a.rfind("--output=", 0) == 0) outputBam = a.substr(9);
        else if (a == "-c" || a == "--clip_mode") clip = need("--clip_mode");
        else if (a.rfind("--clip_mode=", 0) == 0) clip = a.substr(12);
        else if (a == "-m" || a == "--save_msrs") saveMSRs = true;
        else if (a == "--use_csi") useCsi = true;
        else if (a == "-t" || a == "--threads") threads = std::stoi(need("--threads"));
        else if (a == "-v" || a == "--verbose") verbose = true;
        else if (a == "--help") help = true;
        else if (!a.empty() && a[0] == '-') throw BamFilterException("Unknown option: " + a);
        else positional.push_back(a);
    }
    if (help || argc <= 1 || positional.size() < 2) {
        cout << helpMessage() << endl;
        return 1;
    }
    junctionFile = positional[0];
    bamFile = positional[1];
    const auto t0 = std::chrono::steady_clock::now();
    cout << "Running portcullis in BAM filter mode" << endl << "-------------------------------------" << endl << endl;
    BamFilter filter(junctionFile, bamFile, outputBam);
    filter.setClipMode(clipFromString(clip));
    filter.setSaveMSRs(saveMSRs);
    filter.setUseCsi(useCsi);
    filter.setVerbose(verbose);
    filter.setThreads(threads);
    filter.filter();
    if (getenv("PORTCULLIS_PROFILE"))
        fprintf(stderr, "[bamfilt profile] filter() returned %.3f s after its last statement\n",
                std::chrono::duration<double>(std::chrono::steady_clock::now() - g_filterEnded).count());
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::ios::fmtflags f(cout.flags());
    cout << endl << "Portcullis BAM filter completed." << endl << "Total runtime: " << std::fixed << std::setprecision(1) << s << "s" << endl << endl;
    cout.flags(f);
    return 0;
}

}  // namespace portcullis
