#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X `junc` hot path on BASELINE.json's headline workload.

Workload (every N): BASELINE.json configs[2] -- 200 M synthetic paired-end 150-bp reads over 25 contigs with
GRCh38 lengths, ~250 k junctions, orientation FR (`portcullis_amd.synth.c3_contig_configs`).

A "step" is one pass of the hot path over every contig's alignment records, already resident in HBM: per contig
pjb_submit_batch_device + pjb_finish_contig (CIGAR scan/emit, sort/group, anchors, per-pair match statistics,
junction reduce, rows to the host), then the merged row table.

  N = 1   configs[2]: one GPU does all 25 contigs.
  N > 1   configs[3]: the SAME 25 contigs sharded over the ranks by read count (longest-processing-time,
          `distributed.shard_contigs`; the reference shards the same way over threads,
          src/junction_builder.cc:241-245); the path's only exchange is one RCCL all-gather per step of every
          rank's rows + read-length counters; rank 0 holds the merged table and, after the timed region, checks
          it byte for byte against the table one GPU produces alone.  `value` = reads of the whole set / time
          (strong scaling: the total work is fixed as N grows).

    python bench.py --gpus N --steps K --warmup W      (N > 1 without a launcher: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One JSON line on stdout.  Beside the contract's fields it carries `roofline` (dominant kernel, HIP events inside
the timed region), `cpu_baseline` (the oracle over the WHOLE workload on the box's host cores, one thread per
contig like the reference; every device row compared with it) and `e2e` (the same alignments as a BGZF BAM on
disk -> `portcullis_amd junc` -> .tab, compared with the oracle's .tab).
"""
import argparse
import hashlib
import json
import os
import shutil
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (torch's caching allocator keeps blocks above 256 MB whole: the generator's multi-GB temporaries then go back to the driver with
# empty_cache() instead of being pinned by small tensors carved out of them -- the library allocates with hipMalloc, not through torch)
for _v in ("PYTORCH_HIP_ALLOC_CONF", "PYTORCH_CUDA_ALLOC_CONF"):
    os.environ.setdefault(_v, "max_split_size_mb:256")
PEAK_HBM_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes(name, N, C, S, Cs, P, J, L, Pg=0, Rg=0, R=0, cand=0, Rv=0, sort_u32=True, G=0, sort_ids=None, two_bit=True):
    """Algorithmic HBM bytes of ONE launch of kernel `name` over one chain (each byte counted once per logical
    pass, caches ignored; DESIGN.md section 4 derives these).  N reads, C cigar ops, S spliced reads, Cs cigar ops
    of spliced reads, P pairs, J junctions, L read length, Pg pairs / Rg reads that take the generic walks, R
    position runs, cand candidate keys, Rv reads whose closed form k4b_generic only checks, sort_u32: the sort ran on
    dense 32-bit ids (else on the 64-bit keys), G bases of the chain's targets, two_bit: the batches carry their bases in 2 bits as well
    (pjb_batch.seq2, ABI 4) and k1_emit compares those -- a base of read and genome then costs 2 + 2 bits, not 4 + 4."""
    frags = P / 64.0 + J
    ops_s = Cs / max(S, 1)  # cigar ops of a spliced read
    # the sort's digit tables: one count per digit value and tile of 4096 pairs.  Dense ids: digits planned for twice the junctions
    # the context's chains have had, at least 2^16 (pjb_api.hip: prepare_flight), in passes of at most 11 bits
    rs_tiles = P / 4096.0
    if sort_u32:
        id_bits = int(sort_ids if sort_ids else max(1 << 16, 2 * max(int(J), 1))).bit_length()  # (sort_ids: what the library planned for)
        n_pass = -(-id_bits // 11)
        rs_nb = 1 << -(-id_bits // n_pass)
    else:
        rs_nb = 1 << 11
    rs_table = rs_tiles * rs_nb * 4
    table = {
        # pos, cig_off, l_qseq, seq_off (4 each), xs (1) + every cigar op; 24 B written per spliced read (index, pair offset, and the
        # 16-B record of what this pass holds of the read: operations index / count, position, bases offset, l_qseq)
        "k1_count": N * 17 + C * 4 + S * 24,
        # spliced reads only: the list's 24 B, mtid, mpos (4 each), flag (2), mapq, xs (1 each), the ops once; per pair 8 B key +
        # 32 B record written; pairs finished in closed form: L/2 B of packed bases + L/2 B of genome codes -- L/4 + L/4 in 2 bits, plus
        # the read's bit of seq_exc and the genome's exception bitmap (a bit per 64 bases) --; 8 B list entry per read on k4b_generic's
        # lists; candidate key + anchors (16)
        "k1_emit": S * 36 + Cs * 4 + P * 40 + (P - Pg) * L * (0.5 if two_bit else 1.0) + (S / 8.0 + G / 512.0 if two_bit else 0.0) + Rg * 8 + cand * 8,
        # the reads k1_emit leaves to a walk of their operations (those on k4b_generic's first list and the multi-intron reads beyond
        # two introns, about as many again): list entry (8), list record (16), five gathers (12), operations, bases + codes, the pairs
        "k1_generic": (Rg + Rv / 2.0) * (36 + 4 * ops_s + L) + (Pg + Rv) * 40,  # (about half of the checked reads come from here)
        # K2d.  kd_assign: the pair's key read (8), its junction id written (4); the accumulators' rest state (192 B per junction);
        # the tiles' counts of the ids' first digit (the sort's first pass has no rs_hist)
        "kd_assign": P * 12 + J * 192 + rs_table,
        # candidates: key (8), anchors (8), rank (4); bitmap words / end slots they touch; kd_table writes key + anchors per junction
        "kd_mark": cand * 16, "kd_ends": cand * 12 + cand * 4, "kd_table": cand * 20 + J * 16, "kd_reset": cand * 12 + cand * 40,
        # prefix popcount over the start bitmap (a bit per base of the chain's targets): words read twice, a 4-B rank per word
        # written; the scan over the start ranks' end slots (32 B per start, J starts at most) gives the first ids
        # (round 5: the prefix sum runs over the counts of starts per PAGE of the bitmap -- 64 words = 4 096 bases --, and kd_rank_pages
        # reads the words and writes the ranks of the pages that hold a start only: at most J of them)
        "kd_rank_reduce": G / 4096.0 * 4, "kd_rank_apply": G / 4096.0 * 8, "kd_rank_tiles": G / 4096.0 / 2048 * 16 + 64,
        "kd_rank_pages": G / 4096.0 * 8 + min(J, G / 4096.0) * (512 + 256), "kd_first_reduce": J * 32.0, "kd_first_apply": J * 44.0,
        # the sort: key in (4 or 8), index in (4, but for the first pass), both out
        # ... and the tiles' digit counts (rs_table: written by rs_hist, read by rs_panel_sums, read and written as offsets by
        # rs_panel_scan, read by rs_scatter -- round 4's PMC pass: a third of the sort's traffic, not in the formulas then)
        "rs_hist": P * (4 if sort_u32 else 8) + rs_table,
        "rs_scatter": P * (16 if sort_u32 else 24) + rs_table,
        # (a panel = 64 tiles; rs_panel_scan lets every panel add up the sums of the panels before it: panels^2 / 2 rows of rs_nb counts)
        "rs_panel_sums": rs_table + rs_table / 64.0, "rs_panel_scan": 2 * rs_table + (rs_tiles / 64.0) ** 2 / 2.0 * rs_nb * 4,
        # sorted key (8) + index (4) + the position word of the record behind it (4); the apply pass writes the junction id
        # (chains that sort the full keys) the scan's closing kernel; rest state of accumulators and anchors; anchors from the sorted
        # pairs (index, id, lStart / rEnd half of the record) and the BAM-order ids
        "k2_close": 64.0, "kf_init": J * 200, "kf_anchors": P * 28 + J * 8,
        "k2_heads_reduce": P * 16,
        "k2_heads_apply": P * 16 + P * 4 + (J + J + R) * 4,
        # generic reads: list entry (8), cig_off (8), ops, pos/aend half of the first record (16), l_qseq, seq_off (12), L/2 B of
        # bases; per generic pair: id (4), key (8), anchors (8), genome codes of its window (~L/2), result (8)
        "k4b_generic": Rg * (44 + 4 * ops_s + L / 2) + Pg * (28 + L / 2) + Rv * (8 + 16 + 2 * 28),
        # dense chains: a popcount scan over the slices' two mask words, then seg_off / run_first per junction, run_start per run, from
        # the sorted ids (4 B per pair) and the masks
        "k2_runs_reduce": P / 64.0 * 8, "k2_runs_apply": P / 64.0 * 12,
        "k2_expand": P * 4 + P / 64.0 * 20 + (2 * J + R) * 4,
        "k1_scan_tiles": N / 1024.0 * 48, "kg_member_stats": N / 1024.0 * 36,
        # sorted index + id (8), ONE 32-B record, the junction's key (J entries, cached); one 192-B fragment record + its id
        "k4_pairs": P * 40 + frags * 196 + P / 64.0 * 16,
        "k5_frag_reduce": frags * 196 + J * 164,
        "k5_finalize": J * (192 + 8 + 8 + 8 + 48 + 200),
        # per run: its start (4; the next run's start is the neighbour's); per junction seg_off, run_first, the 8-B sum
        "k5_entropy_sum": R * 4 + J * 20,
        "k6_rows_out": J * 200 * 2 + 4096,
    }
    return table.get(name)


def survey_bytes(N, C, S, Cs, P, J, L):
    """SURVEY.md section 8(d)'s formula for the whole path (what a step is priced at whatever the implementation moves):
    N (18 + 4 c) + P (24 + 128 + 16 + 36 + 1.5 A + 4 c_s + 16 + 32) + 264 J with A = mean anchor span per pair = L."""
    c_bar = C / max(N, 1)
    cs_bar = Cs / max(S, 1)
    return N * (18 + 4 * c_bar) + P * (24 + 128 + 16 + 36 + 1.5 * L + 4 * cs_bar + 16 + 32) + J * 264


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes.  Nothing in this
    process has touched the GPU (torch is not even imported yet); the children inherit a clean state."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


def contig_stats(batch):
    """Counts the byte formulas need (cigar ops of spliced reads)."""
    import torch
    cig_off = batch["cig_off"].to(torch.int64)
    n_ops = cig_off[1:] - cig_off[:-1]
    seq_off = batch["seq_off"].to(torch.int64)
    spl = (seq_off[1:] - seq_off[:-1]) > 0
    return int(n_ops[spl].sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=int(os.environ.get("PJB_BENCH_READS", 200_000_000)),
                    help="total reads of the 25-contig set (200 M = BASELINE configs[2]; smaller values are for tests)")
    ap.add_argument("--junctions", type=int, default=int(os.environ.get("PJB_BENCH_JUNCTIONS", 250_000)))
    ap.add_argument("--queue", type=int, default=int(os.environ.get("PJB_BENCH_QUEUE", 3)),
                    help="contigs queued at once (pjb_finish_contig_begin / _end; at most PJB_MAX_QUEUED = 8)")
    ap.add_argument("--group-bases", type=int, default=int(os.environ.get("PJB_BENCH_GROUP_BASES", 1 << 30)),
                    help="targets are finished in groups (pjb_finish_group_begin: ONE kernel chain over several targets) of consecutive "
                         "targets adding up to at most this many bases -- GRCh38: three chains of up to 1 Gb (round 4, measured: 16 / 7 / 5 / 3 / 2 chains = 11.8 / 11.1 / 10.9 / 10.6 / 11.2 ms a step -- profiles/r04d_grouping_sweep.txt; round 3's longer chains favoured 7); 0: one chain per target")
    ap.add_argument("--config", default="c3", choices=["c3", "c5"],
                    help="c3: BASELINE configs[2] (200 M reads; the default and the driver's line).  c5: BASELINE configs[4] WHOLE on one GPU -- "
                         "1 B paired-end reads, 300 k junctions, Zipf depth, strandedness=firststrand, 73 GB of records resident in HBM; no BAM "
                         "leg (the file would be 170 GB), the CPU baseline runs on three of the 25 targets")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", default=os.environ.get("PJB_BENCH_E2E", "1") == "0")
    ap.add_argument("--e2e-workdir", default=os.environ.get("PJB_BENCH_WORKDIR", "/tmp/pjb_bench_e2e"))
    ap.add_argument("--no-verify", action="store_true", help="N > 1: skip rank 0's single-GPU re-run of the whole set")
    ap.add_argument("--no-back-to-back", action="store_true", help="skip the secondary measurement of passes queued back to back")
    args = ap.parse_args()
    if args.config == "c5":
        args.reads, args.junctions, args.no_e2e = 1_000_000_000, 300_000, True
        args.queue = min(args.queue, 2)  # (a 335 M-read chain's scratch is ~45 GB: two of the three chains in flight beside the 73 GB of records)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)

    # exactly one line on stdout: everything the libraries print (RCCL banners, ...) goes to stderr
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    from portcullis_amd import distributed as pd
    from portcullis_amd import ffi, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # PJB_BENCH_SHARE_GPU=1: every rank uses GPU 0 and the exchange runs over gloo (a one-GPU box can then walk
    # through the N > 1 code path; never a measurement)
    share = os.environ.get("PJB_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # PJB_BENCH_FORCE_EXCHANGE=1 runs the N > 1 exchange code with a one-rank group (single-GPU check of that path)
    force_x = world == 1 and os.environ.get("PJB_BENCH_FORCE_EXCHANGE") == "1"
    multi = world > 1 or force_x
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        elif force_x:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev)

    # ------------------------------------------------------------------ workload: the 25-contig set, sharded by contig
    cfgs = synth.c3_contig_configs(args.reads, args.junctions)
    lens = [c.contig_len for c in cfgs]
    shards = pd.shard_contigs([c.n_reads for c in cfgs], world)
    mine = shards[rank]
    if world == 1 and os.environ.get("PJB_BENCH_AS_RANK"):  # (experiment: "r/N": one GPU with the targets rank r of N would get; never a measurement of N GPUs)
        r_, n_ = (int(x) for x in os.environ["PJB_BENCH_AS_RANK"].split("/"))
        mine = pd.shard_contigs([c.n_reads for c in cfgs], n_)[r_]
    L = cfgs[0].read_len
    ORI = "FR"

    ctx = ffi.Context(device=dev_index, orientation=ORI, flags=ffi.FLAG_KERNEL_TIMING, strandedness=1 if args.config == "c5" else 3)
    ctx.set_refs(lens)
    t_gen = time.time()
    contigs = {}  # tid -> dict(batch, n, P, C, S, Cs)

    def load_contig(tid, into_ctx):
        d = synth.generate(cfgs[tid], device=dev, tid=tid)
        into_ctx.upload_contig_device(tid, d["genome"])
        return dict(batch=d["batch"], genome=d["genome"], n=d["n_reads"], P=d["n_pairs"], C=d["n_cigar_ops"],
                    S=d["n_spliced"], Cs=contig_stats(d["batch"]))

    for tid in mine:
        contigs[tid] = load_contig(tid, ctx)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()  # the generator's temporaries go back to the driver (the library allocates with hipMalloc, not through torch)
    t_gen = time.time() - t_gen
    hbm_gb = torch.cuda.memory_allocated() / 1e9
    N_mine = sum(c["n"] for c in contigs.values())
    P_mine = sum(c["P"] for c in contigs.values())
    N_total = sum(c.n_reads for c in cfgs)

    state = {}
    xchg = None
    row_bytes = ffi.ROW_DTYPE.itemsize

    # the chains of a step: groups of consecutive targets (one kernel chain each), or every target alone
    if args.group_bases > 0:
        gb = args.group_bases  # (a share that would be ONE chain of more than 0.6 Gb -- 3 or 4 ranks -- goes as two: the library's rule, pjb_plan_groups)
        chains = ffi.plan_groups(lens, sorted(mine), gb)
    else:
        chains = [[t] for t in mine]
    chains.sort(key=lambda g: -sum(contigs[t]["n"] for t in g))  # largest first: the queue drains at the end of a step on the small ones
    if os.environ.get("PJB_BENCH_CHAINS"):  # (experiment: "0-3;4-9;10-24": the chains and their order as given)
        chains = [list(range(int(a.split("-")[0]), int(a.split("-")[-1]) + 1)) for a in os.environ["PJB_BENCH_CHAINS"].split(";")]
        chains = [[t for t in g if t in mine] for g in chains]
        assert sorted(t for g in chains for t in g) == sorted(mine)

    def step():
        ctx.clear_rows()
        if xchg is not None:
            ctx.set_row_mirror(*xchg.slot_for_next_finish())  # every finish appends header + rows to the exchange slot
        regs = {}
        queued = []  # several chains queued: they run side by side, the device never waits for the host

        def collect_oldest():
            g = queued.pop(0)
            if len(g) == 1:
                regs[g[0]] = ctx.finish_contig_end(g[0])
            else:
                regs.update(ctx.finish_group_end(g))
            if state.get("want_timing"):
                state.setdefault("per_chain", {})[tuple(g)] = ctx.timing()

        for g in chains:
            tq0 = time.perf_counter()
            for tid in g:
                c = contigs[tid]
                ctx.submit_batch_device(tid, c["batch"], c["n"])
            if len(g) == 1:
                ctx.finish_contig_begin(g[0])
            else:
                ctx.finish_group_begin(g)
            state["host_queue_s"] = state.get("host_queue_s", 0.0) + time.perf_counter() - tq0  # (the host's share: PJB_BENCH_HOST_SPLIT)
            queued.append(g)
            if len(queued) >= args.queue:
                collect_oldest()
        while queued:
            collect_oldest()
        if not mine:
            ctx.finish_contig(0)  # a rank without contigs still publishes an (empty) header
        rows = ctx.collect(copy=False)  # view of the pinned row table
        if xchg is not None:
            # the path's only exchange: the merge of the per-rank junction tables and read-length counters (they
            # ride in the slot's header): ONE all-gather per step over RCCL / xGMI straight from HBM, asynchronous
            # -- it overlaps the next step's kernels.  Rank 0 copies the merged table to its host once, at the end
            # of the timed region (a job merges once)
            xchg.launch()
        state["regs"] = regs
        state["rows"] = rows

    for _ in range(max(args.warmup, 1 if multi else 0)):
        step()
    if multi:
        # slot size of the row exchange: the largest table any rank produced in the warm-up, with headroom
        jmax = torch.tensor([len(state["rows"])], dtype=torch.int64, device="cpu" if share else dev)
        dist.all_reduce(jmax, op=dist.ReduceOp.MAX)
        xchg = pd.MirrorExchange(row_bytes, int(jmax.item()) * 5 // 4 + 64, dev)
        step()  # one untimed step with the exchange (buffers, communicator warm-up)
        xchg.finish()
    # per-kernel table from a few fully instrumented steps (outside the timed region), one kernel at a time on one
    # stream: in the timed region kernels of two contigs and of the side streams overlap, and a bracketed duration
    # then includes whatever ran beside the kernel ...
    ctx.reset_kernel_timing()
    n_prof = 2
    state["want_timing"] = True  # sort passes / generic pairs per contig, for the byte formulas
    ctx.set_option("overlap", 0)
    for _ in range(n_prof):
        step()
    ctx.set_option("overlap", 1)
    state["want_timing"] = False
    if xchg is not None:
        xchg.finish()
    kt_all = ctx.kernel_timing()
    dominant = max(kt_all.items(), key=lambda kv: kv[1][1])[0] if kt_all else None
    if multi:  # every rank brackets the same kernel (rank 0's choice)
        names = sorted(kt_all)
        pick = torch.tensor([names.index(dominant) if dominant else 0], dtype=torch.int64, device="cpu" if share else dev)
        dist.broadcast(pick, 0)
        dominant = names[int(pick.item())] if names else None
    # ... and only the dominant kernel keeps its HIP-event bracket inside the timed region
    ctx.select_timed_kernels([dominant] if dominant else [])
    ctx.reset_kernel_timing()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    state["host_queue_s"] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_queue_ms = state["host_queue_s"] * 1e3 / max(args.steps, 1)  # the calling thread inside submit + begin, per step
    merged = xchg.finish() if xchg is not None else None  # the last exchange completes inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    regs, rows = state["regs"], state["rows"].copy()
    # ---- NOT the headline: the same passes back to back -- a pass's first chains are queued while the pass before it still finishes its
    # last one (never more than --queue chains in flight, every chain collected before its targets are submitted again).  A step of the
    # timed region above ends with its last chain's tail (junction ids, sort, reductions, rows over PCIe) alone on the chip; a queue
    # of samples hides it behind the next sample's K1 stage.  Rows of all passes stay in the table: each pass's must equal the step's.
    back_to_back = None
    if not multi and not os.environ.get("PJB_BENCH_ABLATION") and not args.no_back_to_back and args.steps > 1:  # (no row exchange: its slot holds one pass)
        def passes_back_to_back(k):
            ctx.clear_rows()
            inflight = []

            def collect_first():
                g = inflight.pop(0)
                if len(g) == 1:
                    ctx.finish_contig_end(g[0])
                else:
                    ctx.finish_group_end(g)

            for _ in range(k):
                for g in chains:
                    for tid in g:
                        ctx.submit_batch_device(tid, contigs[tid]["batch"], contigs[tid]["n"])
                    if len(g) == 1:
                        ctx.finish_contig_begin(g[0])
                    else:
                        ctx.finish_group_begin(g)
                    inflight.append(g)
                    if len(inflight) >= args.queue:
                        collect_first()
            while inflight:
                collect_first()
            return ctx.collect(copy=False)

        kb = min(args.steps, 10)
        state["kt_timed"] = ctx.kernel_timing()  # (the per-kernel table below is the timed region's, not these passes')
        passes_back_to_back(kb)  # (the row table grows to kb passes' rows once)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        all_rows = passes_back_to_back(kb)
        torch.cuda.synchronize()
        tb = time.perf_counter() - tb
        same = len(all_rows) == kb * len(rows) and all(all_rows[i * len(rows):(i + 1) * len(rows)].tobytes() == rows.tobytes() for i in range(kb))
        assert same, "back-to-back passes: rows differ from the step's"
        back_to_back = dict(ms_per_pass=round(tb / kb * 1e3, 3), passes=kb, reads_per_sec=N_mine * kb / tb, rows_equal_the_steps=bool(same),
                            note="not the headline: passes queued back to back (the next pass's K1 stage runs beside this pass's last chain's tail); "
                                 "every pass's rows on the host and equal to the step's")
        ctx.clear_rows()
    if not os.environ.get("PJB_BENCH_ABLATION"):  # (kernel ablation builds, tools/build_variants.sh: timings only, results are wrong)
        assert sum(r["n_reads"] for r in regs.values()) == N_mine and sum(r["n_pairs"] for r in regs.values()) == P_mine
        assert int(rows["nb_raw"].astype(np.int64).sum()) == P_mine  # every N op lands in exactly one junction row
    J_mine = len(rows)

    n_ranks_seen = dist.get_world_size() if multi else 1
    merged_rows = rows
    if multi and rank == 0:
        # merged table on rank 0: every rank's rows in rank order -> contig order, counters folded
        mt = merged.view(ffi.ROW_DTYPE)
        assert len(mt) == sum(xchg.counts) and xchg.counts[0] == len(rows)
        assert mt[: len(rows)].tobytes() == rows.tobytes()
        # the merge itself is the library's (pjb_merge_rows, the receive side of the C ABI: what a C++ caller runs behind its all-gather);
        # the numpy restatement checks it
        merged_rows, totals = ffi.merge_rows(xchg.host.numpy(), n_ranks_seen, xchg.slot)
        chk_rows, chk_tot = pd.merge_rank_tables(merged, ffi.ROW_DTYPE, xchg.regions)
        assert merged_rows.tobytes() == chk_rows.tobytes() and all(totals[k] == v for k, v in chk_tot.items())
        assert totals["spliced"] + totals["unspliced"] == N_total, (totals, N_total)
        assert totals["sum_len"] == N_total * L
    J_total = len(merged_rows) if rank == 0 else 0

    # ---- N > 1: the merged table must equal what ONE GPU produces for the whole set
    verify = None
    if world > 1 and not args.no_verify:
        if rank == 0:
            t_v = time.time()
            ctx1 = ffi.Context(device=dev_index, orientation=ORI)
            ctx1.set_refs(lens)
            ctx1.clear_rows()
            for tid in range(len(cfgs)):
                c = contigs.get(tid) or load_contig(tid, ctx1)
                if tid in contigs:
                    ctx1.upload_contig_device(tid, c["genome"])
                ctx1.submit_batch_device(tid, c["batch"], c["n"])
                ctx1.finish_contig(tid)
                ctx1.release_contig(tid)
                del c
            single = ctx1.collect()
            ctx1.close()
            same = single.tobytes() == merged_rows.tobytes()
            verify = dict(merged_equals_single_gpu_table=bool(same), rows=len(single), md5=hashlib.md5(single.tobytes()).hexdigest(),
                          seconds=round(time.time() - t_v, 1))
            assert same, "merged multi-GPU table differs from the single-GPU table"
        dist.barrier()

    # ---- per-kernel device time over the timed region (HIP events on the context's stream)
    result = None
    if rank == 0:
        kt_timed = state.get("kt_timed") or ctx.kernel_timing()
        kt = {k: (v[0] / n_prof * args.steps, v[1] / n_prof * args.steps) for k, v in kt_all.items()}
        dom_serial_ms = kt[dominant][1] / kt[dominant][0] if dominant and kt[dominant][0] else None
        if dominant:
            kt[dominant] = kt_timed[dominant]  # measured live over the timed region (beside whatever overlapped it)
        per = state.get("per_chain", {})
        kern = []
        # the batches carry their bases in 2 bits as well (synth; PJB_FFI_SEQ2=0 submits them without: A/B runs of the 4-bit compare)
        two_bit = os.environ.get("PJB_FFI_SEQ2", "1") != "0" and os.environ.get("PJB_NO_SEQ2", "0") == "0"
        # (what the library plans the sort's digits from: the most junctions per read a chain of this context has had -- pjb_api.hip: junc_per_read)
        junc_per_read = max([sum(int(regs[t]["n_junctions"]) for t in g) / max(sum(contigs[t]["n"] for t in g), 1) for g in chains] + [0.0])
        for name, (launches, ms) in kt.items():
            if launches == 0:
                continue
            # algorithmic bytes of this kernel over one step = sum over this rank's chains (x sort passes); the formulas are
            # linear in the counts, so a chain's bytes are those of its targets' summed counts
            tot_b = 0.0
            known = True
            for g in chains:
                cs_ = [contigs[t] for t in g]
                Jc = sum(int(regs[t]["n_junctions"]) for t in g)
                tm = per.get(tuple(g), {})
                b = algorithmic_bytes(name, sum(c["n"] for c in cs_), sum(c["C"] for c in cs_), sum(c["S"] for c in cs_),
                                      sum(c["Cs"] for c in cs_), sum(c["P"] for c in cs_), Jc, L, int(tm.get("generic_pairs", 0)),
                                      int(tm.get("generic_reads", 0)), int(tm.get("position_runs", 0)), int(tm.get("candidates", 0)),
                                      int(tm.get("checked_reads", 0)), int(tm.get("candidates", 0)) > 0,
                                      sum(int(c["genome"].numel()) for c in cs_),
                                      sort_ids=max(1 << 16, int(2.0 * junc_per_read * sum(c["n"] for c in cs_) + 64.0)), two_bit=two_bit)
                if b is None:
                    known = False
                    break
                mult = int(tm.get("sort_passes", 1)) if name in ("rs_hist", "rs_scatter", "rs_panel_sums", "rs_panel_scan") else 1
                if name == "rs_hist" and int(tm.get("candidates", 0)) > 0:
                    mult -= 1  # (dense ids: kd_assign counts the first digit)
                tot_b += b * mult
            per_step = launches / args.steps
            kern.append(dict(name=name, launches=launches, avg_ms=ms / launches, total_ms=ms,
                             alg_bytes=(tot_b / per_step) if known and per_step else None))
        for k in kern:
            k["gbps"] = (k["alg_bytes"] / (k["avg_ms"] * 1e-3) / 1e9) if k["alg_bytes"] else None
        kern.sort(key=lambda k: -k["total_ms"])
        dom = next((k for k in kern if k["name"] == dominant), kern[0])
        roofline = dict(bound="hbm", kernel=dom["name"], achieved=round(dom["gbps"], 1) if dom["gbps"] else None,
                        peak=PEAK_HBM_GBPS, unit="GB/s", frac=round(dom["gbps"] / PEAK_HBM_GBPS, 4) if dom["gbps"] else None,
                        traffic=None, avg_kernel_ms=round(dom["avg_ms"], 5),
                        alg_bytes_per_launch=int(dom["alg_bytes"]) if dom["alg_bytes"] else None,
                        launches_per_step=dom["launches"] / args.steps,
                        note="achieved = algorithmic bytes per launch (DESIGN.md section 4, averaged over the step's launches: "
                             "contigs differ in size) / average launch duration, HIP events on the kernel's stream inside the timed "
                             "region, where kernels of two contigs and of the side streams run beside it; *_alone: the same kernel "
                             "in the instrumented steps before the timed region, one kernel at a time (overlap off)")
        if two_bit and dom["name"] == "k1_emit" and dom["alg_bytes"]:
            # the same launches priced as rounds 1 - 5 priced them (bases at 4 + 4 bits): NOT this kernel's bytes any more -- kept so
            # that the fraction can be read against the earlier rounds' lines
            b4 = sum(algorithmic_bytes("k1_emit", sum(contigs[t]["n"] for t in g), sum(contigs[t]["C"] for t in g), sum(contigs[t]["S"] for t in g),
                                       sum(contigs[t]["Cs"] for t in g), sum(contigs[t]["P"] for t in g), 0, L, int(per.get(tuple(g), {}).get("generic_pairs", 0)),
                                       int(per.get(tuple(g), {}).get("generic_reads", 0)), 0, int(per.get(tuple(g), {}).get("candidates", 0)), two_bit=False)
                     for g in chains) / max(dom["launches"] / args.steps, 1)
            roofline["frac_at_4bit_prices"] = round(b4 / (dom["avg_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4)
            roofline["frac_at_4bit_prices_note"] = ("the same launches with every compared base priced at 4 + 4 bits as in rounds 1 - 5 (the kernel reads 2 + 2): "
                                                    "for comparison across rounds only")
        if dom_serial_ms and dom["alg_bytes"]:
            roofline["avg_kernel_ms_alone"] = round(dom_serial_ms, 5)
            roofline["achieved_alone"] = round(dom["alg_bytes"] / (dom_serial_ms * 1e-3) / 1e9, 1)
            roofline["frac_alone"] = round(roofline["achieved_alone"] / PEAK_HBM_GBPS, 4)
        kernel_ms_per_step = sum(k["total_ms"] for k in kern) / args.steps
        # the whole step against the roofline: algorithmic bytes of all kernels of a step over the step's wall time (the
        # per-kernel fraction above says little once several chains share the chip)
        step_bytes = sum((k["alg_bytes"] or 0) * k["launches"] / args.steps for k in kern)
        roofline["step_alg_bytes"] = int(step_bytes)
        roofline["step_achieved"] = round(step_bytes / (elapsed / args.steps) / 1e9, 1)
        roofline["step_frac"] = round(roofline["step_achieved"] / PEAK_HBM_GBPS, 4)
        # the same step priced by SURVEY.md section 8(d)'s formula (independent of what this implementation moves)
        sv = sum(survey_bytes(c["n"], c["C"], c["S"], c["Cs"], c["P"], int(regs[t]["n_junctions"]), L) for t, c in contigs.items())
        roofline["step_alg_bytes_survey"] = int(sv)
        roofline["step_frac_survey"] = round(sv / (elapsed / args.steps) / 1e9 / PEAK_HBM_GBPS, 4)
        roofline["kernels_without_formula"] = sorted(k["name"] for k in kern if not k["alg_bytes"])
        # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes of
        # this same command: tools/pmc_traffic.sh -> tools/summarize_pmc.py).  Counters cannot be read from inside
        # the process; the committed measurement for this workload is attached with the commit it was taken at.
        # A committed figure describes the kernels it was measured on: it carries the hash of portcullis_amd/csrc/ at that time
        # (tools/csrc_hash.py) and is dropped when the working tree's kernels differ.
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
        here_hash = csrc_hash()
        if world == 1 and args.reads == 200_000_000 and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("_workload") == "C3":
                    tr = tj.get("pjb::" + dom["name"])
                    if tr and tj.get("_csrc_hash") == here_hash:
                        roofline["traffic"] = tr["hbm_bytes_per_launch"]
                        roofline["traffic_source"] = ("profiles/pmc_traffic_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                      f"passes of this command at commit {tj.get('_commit', '?')}, csrc hash {here_hash})")
                    elif tr:
                        roofline["traffic_source"] = (f"none: profiles/pmc_traffic_latest.json was measured on other kernels (csrc hash "
                                                      f"{tj.get('_csrc_hash', 'unrecorded')}, this tree {here_hash})")
            except Exception:
                pass
        # the third reading of the same kernel: algorithmic bytes over the rocprofv3 --kernel-trace --stats average of the committed
        # summary (profiles/rocprof_latest.json, tools/profile_round.sh), if it was taken on these kernels
        rpath = os.path.join(ROOT, "profiles", "rocprof_latest.json")
        if world == 1 and args.reads == 200_000_000 and os.path.exists(rpath) and dom["alg_bytes"]:
            try:
                rj = json.load(open(rpath))
                avg_ns = rj.get("kernels", {}).get("pjb::" + dom["name"], {}).get("avg_ns")
                if rj.get("_workload") == "C3" and rj.get("_csrc_hash") == here_hash and avg_ns:
                    roofline["avg_kernel_ms_rocprof"] = round(avg_ns * 1e-6, 5)
                    roofline["frac_rocprof"] = round(dom["alg_bytes"] / (avg_ns * 1e-9) / 1e9 / PEAK_HBM_GBPS, 4)
                    roofline["rocprof_source"] = f"profiles/rocprof_latest.json ({rj.get('_source', '?')}, commit {rj.get('_commit', '?')})"
            except Exception:
                pass

        cpu = None
        oracle_tab_md5 = None
        if world == 1 and not args.no_cpu_baseline:
            sample = None if args.config != "c5" else {t: contigs[t] for t in (0, 12, 24) if t in contigs}
            cpu, oracle_tab_md5, oracle_tab_len = cpu_baseline(sample or contigs, cfgs, rows, regs, ORI, synth, partial=sample is not None)
        e2e = None
        if world == 1 and not args.no_e2e:
            try:
                e2e = e2e_leg(contigs, cfgs, args.e2e_workdir, ORI, oracle_tab_md5, args.reads, args.junctions)
            except Exception as ex:  # the e2e leg must never cost the bench line
                e2e = {"error": f"{type(ex).__name__}: {ex}"[:500]}
            if os.environ.get("PJB_BENCH_DROP_WORKDIR"):  # (kept by default: the next run on this box finds the prepared BAM)
                shutil.rmtree(args.e2e_workdir, ignore_errors=True)

        tot_bytes = sum((k["alg_bytes"] or 0) * k["launches"] / args.steps for k in kern)
        cfg_name = "configs[4] (whole, on one GPU)" if args.config == "c5" else "configs[2]" if world == 1 else "configs[3]"
        result = {
            "metric": "junc_reads_per_sec",
            "value": N_total * args.steps / elapsed,
            "unit": "reads/s",
            "n_gpus": n_ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE {cfg_name}: synthetic {N_total} paired-end {L}-bp reads, {len(cfgs)} contigs of "
                                   f"GRCh38 lengths ({sum(lens)} bp), {J_total} junctions, orientation {ORI}"
                                   + ("" if world == 1 else f", contigs sharded over {world} GPUs by read count (LPT), "
                                      "one RCCL all-gather of rows + counters per step"),
                       "reads_total": N_total, "junctions_total": J_total, "contigs": len(cfgs),
                       "contigs_per_rank": [len(s) for s in shards], "reads_rank0": N_mine, "pairs_rank0": P_mine,
                       # the deal of targets to ranks (longest-processing-time by read count, distributed.shard_contigs): what an N-GPU run of
                       # this workload can be at best -- step time ~ max shard, so speed-up <= N / max_over_mean -- for 1, 2, 4 and 8 ranks
                       "lpt_balance": lpt_balance([c.n_reads for c in cfgs], world, pd),
                       "lpt_balance_by_ranks": {str(k): lpt_balance([c.n_reads for c in cfgs], k, pd)["max_over_mean_reads"] for k in (1, 2, 4, 8)},
                       "sharding": "by contig", "input": "device-resident SoA records (pjb_submit_batch_device)" + (", bases in 4 and in 2 bits (pjb_batch.seq2 / .seq_exc, ABI 4)" if two_bit else ", bases in 4 bits only"),
                       "chains": [len(g) for g in chains],
                       "queue": (f"{len(chains)} kernel chains per step, each over a group of consecutive targets (pjb_finish_group_begin / _end), "
                                 if args.group_bases > 0 else "one kernel chain per target (pjb_finish_contig_begin / _end), ")
                                + f"{args.queue} chains queued at once, side by side on the device",
                       "hbm_resident_gb_rank0": round(hbm_gb, 2)},
            "junctions_per_sec": J_total * args.steps / elapsed,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "e2e": e2e,
            "multi_gpu_check": verify,
            "device_kernel_ms_per_step": round(kernel_ms_per_step, 4),
            "launches_per_step": round(sum(k["launches"] for k in kern) / args.steps, 1),
            # sum of the kernels' own durations (one at a time) over the step's wall time: > 1 = what the streams overlap
            "overlap_factor": round(kernel_ms_per_step / (elapsed / args.steps * 1e3), 4),
            "back_to_back": back_to_back,
            "host_queue_ms_per_step": round(host_queue_ms, 3),  # the calling thread inside pjb_submit_batch_device + pjb_finish_*_begin
            # a chain's stages on their own (the instrumented steps: one chain at a time, one kernel at a time, gaps included): what the
            # LAST chain of a step costs behind its K1 stage, where no other chain runs beside it
            "chain_stages_ms_alone": [dict(targets=len(g), **{k: round(float(v), 3) for k, v in state["per_chain"][tuple(g)]["stage_ms"].items()})
                                      for g in chains if tuple(g) in state.get("per_chain", {})],
            "pipeline_gbps": round(tot_bytes / (kernel_ms_per_step * 1e-3) / 1e9, 1) if kernel_ms_per_step else None,
            "kernels": [dict(name=k["name"], launches_per_step=k["launches"] / args.steps, avg_ms=round(k["avg_ms"], 5),
                             ms_per_step=round(k["total_ms"] / args.steps, 4),
                             gbps=round(k["gbps"], 1) if k["gbps"] else None) for k in kern],
            "kernels_note": "per-launch durations from the instrumented steps before the timed region (one kernel at a time); "
                            "the roofline kernel's row is the one measured inside the timed region",
            "datagen_s": round(t_gen, 2),
        }
        # a line from an ablation build or another library build says so: its results were not checked
        if world == 1 and os.environ.get("PJB_BENCH_AS_RANK"):
            # (experiment: this GPU held the targets ONE rank of an N-rank run would get -- no exchange, never a measurement of N GPUs;
            # `value` counts the share's own reads, not the set's)
            result["as_rank"] = os.environ["PJB_BENCH_AS_RANK"]
            result["value"] = N_mine * args.steps / elapsed
            result["config"]["targets_rank"] = sorted(mine)
        if os.environ.get("PJB_BENCH_ABLATION") or os.environ.get("PJB_LIB_PATH"):
            result["ablation"] = bool(os.environ.get("PJB_BENCH_ABLATION"))
            result["lib_path"] = os.environ.get("PJB_LIB_PATH")
            result["checked"] = not os.environ.get("PJB_BENCH_ABLATION")
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    ctx.set_row_mirror(0, 0)
    ctx.close()
    if multi:
        dist.destroy_process_group()


def csrc_hash():
    """sha256 (16 hex digits) over the kernel sources: what a committed PMC / rocprof figure must match to be quoted."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from csrc_hash import csrc_hash as h
    return h(ROOT)


def lpt_balance(reads, ranks, pd):
    sh = pd.shard_contigs(reads, ranks)
    per = [sum(reads[t] for t in s_) for s_ in sh]
    mean = sum(per) / max(len(per), 1)
    return {"ranks": ranks, "reads_per_rank": per, "max_over_mean_reads": round(max(per) / mean, 4) if mean else None}


def host_cores():
    """Cores this process may really use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def cpu_baseline(contigs, cfgs, dev_rows, dev_regs, orientation, synth, partial=False):
    """The CPU oracle (oracle/portcullis_oracle.c, kind "port") over the WHOLE workload on the box's host cores:
    one thread per contig, longest first, exactly the parallelism the reference has (one thread per target,
    src/junction_builder.cc:241-245).  The sample is bounded by construction (about 25 s of single-core work for
    the 200 M-read set).  Every device row of the timed run is compared with the oracle's."""
    import concurrent.futures as cf

    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as orc
    from parity import assert_rows_equal, region_equal

    cores = min(host_cores(), len(contigs))
    order = sorted(contigs, key=lambda t: -contigs[t]["n"])
    lens = [c.contig_len for c in cfgs]

    def run(tid):
        c = contigs[tid]
        hb = synth.batch_to_numpy(c["batch"], 0, c["n"])
        g = c["genome"].cpu().numpy().tobytes()
        t = time.perf_counter()
        orows, oreg = orc.find_juncs(tid, lens[tid], g, hb, orientation)  # the C call releases the GIL
        return tid, orows, oreg, time.perf_counter() - t

    orc.lib()
    t0 = time.perf_counter()
    out = {}
    with cf.ThreadPoolExecutor(max_workers=cores) as ex:
        for tid, orows, oreg, dt in ex.map(run, order):
            out[tid] = (orows, oreg, dt)
    wall = time.perf_counter() - t0
    cpu_s = sum(v[2] for v in out.values())
    worst = 0.0
    n_reads = 0
    for tid in sorted(out):
        orows, oreg, _ = out[tid]
        region_equal(dev_regs[tid], oreg)
        worst = max(worst, assert_rows_equal(dev_rows[dev_rows["refid"] == tid], orows))
        n_reads += contigs[tid]["n"]
    # the oracle's .tab for the whole set (merge, global mean read length, calcJunctionStats, writer)
    allrows = np.concatenate([out[t][0] for t in sorted(out)])
    tot = sum(out[t][1]["spliced"] + out[t][1]["unspliced"] for t in out)
    mean = sum(out[t][1]["sum_len"] for t in out) / tot
    allrows = orc.finalize(allrows, mean)
    tab = orc.write_tab(allrows, list(synth.GRCH38_NAMES[: len(cfgs)]), lens)
    longest = max(v[2] for v in out.values())
    return ({"value": n_reads / wall, "unit": "reads/s", "cores": cores, "kind": "port",
             "sample": ("a sample of the workload: " if partial else "the whole workload: ") + f"{n_reads} records of {len(out)} contigs ({len(allrows)} junctions), "
                       f"oracle/portcullis_oracle.c on pre-decoded records, one thread per contig on {cores} cores: "
                       f"{wall:.1f} s wall including the records' copy out of HBM, {cpu_s:.1f} s of oracle CPU time, longest contig {longest:.1f} s; "
                       f"every device row of the timed run equals the oracle's (integers bit-exact, max |entropy diff| {worst:.2g})",
             "single_core_reads_per_sec": n_reads / cpu_s,
             "junctions_per_sec": len(allrows) / wall},
            None if partial else hashlib.md5(tab).hexdigest(), len(tab))


def wait_until_gone(exe, timeout=20.0):
    """Seconds until no process named like `exe` is left in the process table (the forked child of the last `portcullis_amd`
    command leaving -- it stays there until the driver has taken its device memory back): looks, never signals."""
    name = os.path.basename(exe)[:15]  # (what /proc/<pid>/comm holds)
    t0 = time.time()
    while time.time() - t0 < timeout:
        alive = False
        for pid in os.listdir("/proc"):
            if not pid.isdigit():
                continue
            try:
                with open(os.path.join("/proc", pid, "comm")) as f:
                    if f.read().strip() == name:
                        alive = True
                        break
            except OSError:
                pass
        if not alive:
            break
        time.sleep(0.02)
    return time.time() - t0


def warm_file(path, threads, passes=2):
    """Reads the file `passes` times with `threads` readers (os.pread releases the GIL)."""
    import concurrent.futures as cf

    size = os.path.getsize(path)
    chunk = 64 << 20
    fd = os.open(path, os.O_RDONLY)
    try:
        def rd(off):
            return len(os.pread(fd, chunk, off))
        for _ in range(max(0, passes)):
            with cf.ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
                sum(ex.map(rd, range(0, size, chunk)))
    finally:
        os.close(fd)


class StalePreparedDir(RuntimeError):
    pass


def e2e_leg(contigs, cfgs, workdir, orientation, oracle_tab_md5, reads_arg, junctions_arg):
    """e2e_leg_once, and once more from scratch if a prepared directory kept from an earlier run turns out stale (its .tab differs
    from the oracle's although the manifest matched)."""
    try:
        return e2e_leg_once(contigs, cfgs, workdir, orientation, oracle_tab_md5, reads_arg, junctions_arg)
    except StalePreparedDir:
        shutil.rmtree(workdir, ignore_errors=True)
        return e2e_leg_once(contigs, cfgs, workdir, orientation, oracle_tab_md5, reads_arg, junctions_arg)


def generator_hash():
    """What the prepared files were made by: the synthetic generator and the BAM writer (a change in either makes a kept directory stale)."""
    h = hashlib.sha256()
    for rel in ("portcullis_amd/synth.py", "tools/soa2bam.cc"):
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()[:16]


def e2e_leg_once(contigs, cfgs, workdir, orientation, oracle_tab_md5, reads_arg, junctions_arg):
    """End to end on the same alignments: BGZF BAM + FASTA on disk (a Portcullis prep directory written by
    tools/soa2bam from the records in HBM) -> `portcullis_amd junc` (file bytes -> pjb_submit_bam: inflate, record
    parse and the junc pipeline on the device; merge, calcJunctionStats and the writers on the host) -> .tab,
    whose md5 must equal the oracle's .tab for the whole workload -- after EVERY run.  `wall_s` is the MEDIAN of the
    runs of the command as it is by default: ONE process, timed until it is gone (all runs are in `runs_s`).  The opt-in
    early return (PORTCULLIS_EARLY_RETURN=1, host/src/main.cc) is timed beside it: until the outputs are closed and until
    the process tree is gone.  `speedup_vs_cpu` uses the slowest of the three.  `cpu`: the same prepared directory -> .tab on the host cores alone
    (oracle/orc_bam2tab: zlib inflate + record parse + the oracle port, one thread per target like the reference)."""
    from portcullis_amd import synth

    t_all = time.time()
    prep = os.path.join(workdir, "prep")
    bam = os.path.join(prep, "portcullis.sorted.alignments.bam")
    # the prepared directory is kept between runs on the same box (writing the 33 GB BAM is 80 s of zlib on 16 cores):
    # a manifest names the workload it was made from
    manifest = dict(reads=reads_arg, junctions=junctions_arg, contigs=len(cfgs), read_len=cfgs[0].read_len,
                    n=[int(contigs[t]["n"]) for t in sorted(contigs)], P=[int(contigs[t]["P"]) for t in sorted(contigs)],
                    seeds=[int(getattr(c, "seed", 0)) for c in cfgs], generator=generator_hash())
    mpath = os.path.join(workdir, "manifest.json")
    cached = False
    try:
        cached = json.load(open(mpath)) == manifest and os.path.exists(bam) and os.path.exists(bam + ".bai")
    except Exception:
        cached = False
    cores = host_cores()
    t_dump = t_bam = t_sync = 0.0
    if not cached:
        shutil.rmtree(workdir, ignore_errors=True)
        os.makedirs(prep)
        ext = dict(pos="i32", flag="u16", mapq="u8", xs="u8", l_qseq="i32", mtid="i32", mpos="i32", cig_off="u32",
                   cigar="u32", seq_off="u32", seq4="u8")
        dirs = []
        t0 = time.time()
        for tid in sorted(contigs):
            d = os.path.join(workdir, f"contig{tid}")
            os.makedirs(d)
            open(os.path.join(d, "name.txt"), "w").write(synth.GRCH38_NAMES[tid])
            contigs[tid]["genome"].cpu().numpy().tofile(os.path.join(d, "genome.u8"))
            for k, e in ext.items():
                contigs[tid]["batch"][k].cpu().numpy().tofile(os.path.join(d, f"{k}.{e}"))
            dirs.append(d)
        t_dump = time.time() - t0
        exe = os.path.join(ROOT, "tools", "soa2bam")
        if not os.path.exists(exe):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tools", "soa2bam.cc"), "-lz", "-lpthread"])
        t0 = time.time()
        subprocess.check_call([exe, prep, str(cores)] + dirs, stdout=subprocess.DEVNULL)
        t_bam = time.time() - t0
        for d in dirs:
            shutil.rmtree(d, ignore_errors=True)
        # the BAM has just been written: until its pages have gone to the disk, reading them back -- from the page cache -- runs
        # at a tenth of the speed (measured: pread of cached-but-dirty pages 12 GB/s with 16 threads, 100 GB/s once clean) and
        # the first run measures the writeback, not the program.  A prepared BAM that a pipeline hands to `junc` is clean.
        t0 = time.time()
        os.sync()
        t_sync = time.time() - t0
        json.dump(manifest, open(mpath, "w"))
    bam_bytes = os.path.getsize(bam)
    cli = os.path.join(ROOT, "portcullis_amd", "host", "portcullis_amd")
    n_reads = sum(c["n"] for c in contigs.values())
    walls, md5s = [], []
    out = os.path.join(workdir, "out", "pc")
    n_tab = 0
    # "Page cache warm" made true before anything is timed.  The runs right after the BAM was written were the outliers of
    # every bench line so far (driver, round 2: 3.49 / 4.16 / 2.47 s; this round: 2.50 / 3.69 / 2.38 / 2.47 / 2.43, 4.66 / 3.85 /
    # 2.36 / 3.58 / 2.26 -- and 2.0-2.3 s five times in a row a minute later, same files, same binary): pages that were
    # written, synced and read once or twice are still moving between the kernel's inactive and active lists, and a 33 GB
    # pread pays for it.  Two plain passes over the file and one untimed run of the program settle that; a prepared BAM that
    # a pipeline has just sorted and indexed is in the same state.
    t0 = time.time()
    warm_file(bam, cores, passes=int(os.environ.get("PJB_BENCH_E2E_WARM_PASSES", 2)))
    t_warm = time.time() - t0
    warmup_run_s = None
    waits = []
    if not os.environ.get("PJB_BENCH_E2E_NO_WARMUP_RUN"):
        t = time.time()
        p = subprocess.run([cli, "junc", "-t", str(cores), "--orientation", orientation, "-o", out, prep], capture_output=True, text=True)
        warmup_run_s = round(time.time() - t, 3)
        if p.returncode != 0:
            raise RuntimeError("portcullis_amd junc failed: " + (p.stderr or p.stdout)[-400:])
    # A run that starts right behind another one is not a run of the program alone: the driver is still taking the earlier process's
    # 100+ GB of device memory apart, and somewhere in the next run every copy to the device stands still for 1.5 - 2 s (at start-up, or
    # with the first target half way across).  profiles/r06_e2e_pause.txt: 18 runs back to back, median 2.68 s, ten of them over 2.4 s;
    # 18 runs with 3 s of nothing before each, median 1.85 s, none over 2.2 s.  So: a pause before every timed run (not timed).
    settle_s = float(os.environ.get("PJB_BENCH_E2E_SETTLE_S", 3.0))
    for rep in range(max(1, int(os.environ.get("PJB_BENCH_E2E_REPS", 5)))):
        env = dict(os.environ)
        if os.environ.get("PJB_BENCH_E2E_SWEEP"):  # (experiment: "VAR=a,b,c": repeat k runs with VAR = the k-th value)
            var, vals = os.environ["PJB_BENCH_E2E_SWEEP"].split("=")
            vals = vals.split(",")
            env[var] = vals[rep % len(vals)]
        try:
            os.remove(out + ".junctions.tab")
        except OSError:
            pass
        quiet_s = wait_until_gone(cli)  # (nothing of an earlier run is left in the process table)
        time.sleep(settle_s)
        t = time.time()
        p = subprocess.run([cli, "junc", "-t", str(cores), "--orientation", orientation, "-o", out, prep],
                           capture_output=True, text=True, env=env)
        walls.append(time.time() - t)
        waits.append(round(quiet_s, 3))
        if p.returncode != 0:
            raise RuntimeError("portcullis_amd junc failed: " + (p.stderr or p.stdout)[-400:])
        tab = open(out + ".junctions.tab", "rb").read()
        n_tab = tab.count(b"\n") - 2
        md5s.append(hashlib.md5(tab).hexdigest())
        if oracle_tab_md5 and md5s[-1] != oracle_tab_md5:
            if cached:  # a kept directory that no longer matches what the generator makes: once more from scratch
                try:
                    os.remove(mpath)
                except OSError:
                    pass
                raise StalePreparedDir()
            raise RuntimeError(f"e2e run {rep}: .tab md5 {md5s[-1]} differs from the oracle's {oracle_tab_md5}")
    # the opt-in early return: the command comes back when the outputs are closed, the child that did the work is gone later
    early_closed, early_tree = [], []
    for rep in range(max(0, int(os.environ.get("PJB_BENCH_E2E_EARLY_REPS", 3)))):
        wait_until_gone(cli)
        time.sleep(settle_s)
        try:
            os.remove(out + ".junctions.tab")
        except OSError:
            pass
        t = time.time()
        p = subprocess.run([cli, "junc", "-t", str(cores), "--orientation", orientation, "-o", out, prep], capture_output=True, text=True,
                           env=dict(os.environ, PORTCULLIS_EARLY_RETURN="1"))
        early_closed.append(time.time() - t)
        wait_until_gone(cli)
        early_tree.append(time.time() - t)
        if p.returncode != 0:
            raise RuntimeError("portcullis_amd junc (early return) failed: " + (p.stderr or p.stdout)[-400:])
        m = hashlib.md5(open(out + ".junctions.tab", "rb").read()).hexdigest()
        if oracle_tab_md5 and m != oracle_tab_md5:
            raise RuntimeError(f"e2e early-return run {rep}: .tab md5 {m} differs from the oracle's {oracle_tab_md5}")
    if os.environ.get("PJB_BENCH_E2E_PROFILE"):  # one more run with the host-side timers on; their report goes to a file
        env = dict(os.environ, PJB_PROFILE_HOST="1")
        p = subprocess.run([cli, "junc", "-t", str(cores), "--orientation", orientation, "-o", out, prep], capture_output=True, text=True, env=env)
        with open(os.environ["PJB_BENCH_E2E_PROFILE"], "w") as f:
            f.write(p.stderr + "\n---- stdout ----\n" + p.stdout)
    med = sorted(walls)[len(walls) // 2] if len(walls) % 2 else sum(sorted(walls)[len(walls) // 2 - 1: len(walls) // 2 + 1]) / 2
    def median(v):
        v = sorted(v)
        return (v[len(v) // 2] if len(v) % 2 else (v[len(v) // 2 - 1] + v[len(v) // 2]) / 2) if v else None

    slowest = max([med] + [x for x in (median(early_closed), median(early_tree)) if x])
    res = {"wall_s": round(med, 3), "wall_is": f"median of {len(walls)} runs of the command as it is by default: one process, timed until it is gone; "
                                               f"{settle_s:g} s of nothing before each run (the process before it has left the device by then)",
           "pause_before_each_run_s": settle_s,
           "runs_s": [round(w, 3) for w in walls],
           "early_return": {"what": "PORTCULLIS_EARLY_RETURN=1: a child forked before the GPU is touched does the work, the command returns when "
                                    "the outputs are closed; the child's device memory goes back to the driver after that",
                            "wall_s_outputs_closed": round(median(early_closed), 3) if early_closed else None,
                            "wall_s_process_tree": round(median(early_tree), 3) if early_tree else None,
                            "runs_outputs_closed_s": [round(w, 3) for w in early_closed], "runs_process_tree_s": [round(w, 3) for w in early_tree]},
           "wall_s_slowest_of_the_three": round(slowest, 3),
           "min_s": round(min(walls), 3), "max_s": round(max(walls), 3), "reads_per_sec": n_reads / med,
           "reads": n_reads, "bam_gb": round(bam_bytes / 1e9, 2), "host_cores": cores,
           "junctions": n_tab, "tab_md5": md5s[-1], "oracle_tab_md5": oracle_tab_md5,
           "tab_identical_to_oracle": (all(m == oracle_tab_md5 for m in md5s)) if oracle_tab_md5 else None,
           "tab_md5_checked_runs": len(md5s) if oracle_tab_md5 else 0,
           "path": "BGZF BAM bytes on disk (page cache warm) -> portcullis_amd junc (device ingest: pjb_submit_bam) -> .junctions.tab/.bed",
           "warmup_run_s": warmup_run_s,
           "prep_s": {"dump_soa": round(t_dump, 1), "soa2bam": round(t_bam, 1), "sync": round(t_sync, 1), "warm_passes": round(t_warm, 1), "cached": cached}}
    # ---- the CPU neighbour: same files, host cores only
    if not os.environ.get("PJB_BENCH_NO_E2E_CPU"):
        try:
            exe = os.path.join(ROOT, "oracle", "orc_bam2tab")
            if not os.path.exists(exe):
                subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "orc_bam2tab"], stdout=subprocess.DEVNULL)
            cpu_tab = os.path.join(workdir, "out", "cpu.junctions.tab")
            t = time.time()
            p = subprocess.run([exe, prep, cpu_tab, str(cores), orientation], capture_output=True, text=True)
            wall = time.time() - t
            if p.returncode != 0:
                raise RuntimeError("orc_bam2tab failed: " + (p.stderr or p.stdout)[-300:])
            info = json.loads(p.stdout.strip().split("\n")[-1])
            cmd5 = hashlib.md5(open(cpu_tab, "rb").read()).hexdigest()
            res["cpu"] = {"wall_s": round(wall, 2), "reads_per_sec": n_reads / wall, "cores": cores, "threads": info["threads"],
                          "kind": "port", "decode_cpu_s": info["decode_cpu_s"], "oracle_cpu_s": info["oracle_cpu_s"],
                          "longest_target_s": info["longest_target_s"], "tab_md5": cmd5,
                          "tab_identical_to_device": cmd5 == md5s[-1],
                          "what": "oracle/orc_bam2tab on the same prepared directory: per target one thread (the reference's "
                                  "model, src/junction_builder.cc:241-245) seeks through the .bai, inflates block after block with "
                                  "zlib, parses the records and runs the oracle port of findJuncs; merge, calcJunctionStats and "
                                  ".tab writer.  The port is several times faster per thread than the reference binary, so "
                                  "this is a lower bound on the reference's wall clock on these cores"}
            res["speedup_vs_cpu"] = round(wall / med, 1)  # (against the command as it is by default: one process, timed until it is gone)
            res["speedup_vs_cpu_slowest"] = round(wall / slowest, 1)  # (against the slowest of: one process, early return until closed / until its child is gone)
        except Exception as ex:
            res["cpu"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    res["leg_s"] = round(time.time() - t_all, 1)
    return res


if __name__ == "__main__":
    main()
