"""Parity at BASELINE.json's full single-GPU size (configs[1]: 10 M reads, 1 contig, ~50 k
junctions): the oracle finishes this workload in about a second, so the device rows are compared
with it directly, plus the size-independent properties (conservation of pairs, determinism,
batch-split invariance)."""
import hashlib
import os

import numpy as np
import pytest

from parity import assert_rows_equal, region_equal, sort_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2():
    import torch
    from portcullis_amd import ffi, synth

    assert ffi.device_count() >= 1
    cfg = synth.CONFIGS["C2"]
    data = synth.generate(cfg, device="cuda")
    torch.cuda.synchronize()
    ctx = ffi.Context(0, "UNKNOWN")
    ctx.set_refs([cfg.contig_len])
    ctx.upload_contig_device(0, data["genome"])
    ctx.clear_rows()
    ctx.submit_batch_device(0, data["batch"], data["n_reads"])
    reg = ctx.finish_contig(0)
    rows = ctx.collect()
    yield cfg, data, ctx, reg, rows
    ctx.close()


def test_fullsize_properties(c2):
    cfg, data, ctx, reg, rows = c2
    assert reg["n_reads"] == cfg.n_reads and reg["n_pairs"] == data["n_pairs"]
    assert reg["spliced"] == data["n_spliced"] and reg["spliced"] + reg["unspliced"] == cfg.n_reads
    assert reg["sum_len"] == cfg.n_reads * cfg.read_len
    assert int(rows["nb_raw"].sum()) == data["n_pairs"]                      # every N op lands in exactly one junction
    assert (rows["r1pos"] + rows["r1neg"] + rows["r2pos"] + rows["r2neg"] == rows["nb_raw"]).all()
    assert (rows["nb_dist"] <= rows["nb_raw"]).all() and (rows["nb_dist"] >= 1).all()
    assert (rows["left"] <= rows["start"]).all() and (rows["end"] <= rows["right"]).all()
    assert (rows["jad"][:, :-1] >= rows["jad"][:, 1:]).all()                 # JAD is non-increasing
    assert (rows["jad"][:, 0] <= rows["nb_raw"]).all()
    assert (rows["entropy"] >= 0).all() and (rows["entropy"] <= np.log2(np.maximum(rows["nb_raw"], 2)) + 1e-9).all()
    key = rows["start"].astype(np.int64) << 32 | rows["end"].astype(np.int64)
    assert (np.diff(key) > 0).all()                                          # one row per intron, sorted
    assert rows["canonical"].max() <= 2 and len(rows) > 45_000


def test_fullsize_deterministic_and_split_invariant(c2):
    cfg, data, ctx, reg, rows = c2
    from portcullis_amd import synth

    ctx.clear_rows()
    ctx.submit_batch_device(0, data["batch"], data["n_reads"])
    ctx.finish_contig(0)
    again = ctx.collect()
    assert hashlib.md5(again.tobytes()).hexdigest() == hashlib.md5(rows.tobytes()).hexdigest()
    # the same records submitted as three host batches
    n = cfg.n_reads
    ctx.clear_rows()
    for lo, hi in ((0, n // 3), (n // 3, n // 3 + 1_000_001), (n // 3 + 1_000_001, n)):
        ctx.submit_batch(0, synth.batch_to_numpy(data["batch"], lo, hi))
    ctx.finish_contig(0)
    split = ctx.collect()
    assert hashlib.md5(split.tobytes()).hexdigest() == hashlib.md5(rows.tobytes()).hexdigest()


def test_fullsize_matches_oracle(c2):
    cfg, data, ctx, reg, rows = c2
    from oracle import oracle as orc
    from portcullis_amd import synth

    hb = synth.batch_to_numpy(data["batch"])
    orows, oreg = orc.find_juncs(0, cfg.contig_len, data["genome"].cpu().numpy().tobytes(), hb, "UNKNOWN")
    region_equal(reg, oreg)
    assert_rows_equal(rows, orows)


@pytest.mark.parametrize("env", [{"PJB_RADIX_BITS": "6"}, {"PJB_RADIX_BITS": "10"}, {"PJB_RADIX_BITS": "12"},
                                 {"PJB_K1S_BLOCKS": "1"}, {"PJB_K1S_BLOCKS": "3"}, {"PJB_K1S_BLOCKS": "64"}])
def test_sort_variants_agree(c2, monkeypatch, env):
    """The radix sort's digit width (unrolled 9/10/11-bit and generic match loops, 12-bit digits with
    more than 64 KB of LDS) must not change a single byte of the row table.  Neither must the number of blocks
    k1_scan_tiles scans the 9 766 tiles with: 1 (three rounds in one block, nobody to wait for), 3 (ranges of several
    rounds behind predecessors), 64 (more blocks than the default's ten: short ranges)."""
    cfg, data, ctx0, reg, rows = c2
    from portcullis_amd import ffi

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    with ffi.Context(0, "UNKNOWN") as ctx:
        ctx.set_refs([cfg.contig_len])
        ctx.upload_contig_device(0, data["genome"])
        for _ in range(2):
            ctx.clear_rows()
            ctx.submit_batch_device(0, data["batch"], data["n_reads"])
            reg2 = ctx.finish_contig(0)
            again = ctx.collect()
            assert reg2 == reg
            assert hashlib.md5(again.tobytes()).hexdigest() == hashlib.md5(rows.tobytes()).hexdigest()


def _run_bench(extra_args, env_extra, timeout=900):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **env_extra)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--reads", "4000000", "--junctions", "5000", "--steps", "3",
                        "--warmup", "1"] + extra_args, capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    return json.loads([l for l in p.stdout.split("\n") if l.startswith("{")][-1])


def test_bench_exchange_path_single_rank():
    """bench.py's N > 1 step (device-side all-gather of the row mirror over RCCL, asynchronous, counters in the slot
    header) with a one-rank process group: the CUDA-specific parts of MirrorExchange on real hardware."""
    d = _run_bench(["--no-cpu-baseline", "--no-e2e"], {"PJB_BENCH_FORCE_EXCHANGE": "1"})
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["contigs"] == 25


def test_bench_two_ranks_share_one_gpu():
    """`bench.py --gpus 2` starts its two ranks itself; they share GPU 0 and exchange over gloo
    (PJB_BENCH_SHARE_GPU=1: a debug mode for one-GPU boxes).  The contig set is sharded by read count, every rank's
    mirror slot accumulates its contigs' rows, and rank 0 checks the merged table byte for byte against the table a
    single context produces for all 25 contigs (BASELINE configs[3] in miniature)."""
    d = _run_bench(["--gpus", "2"], {"PJB_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["multi_gpu_check"]["merged_equals_single_gpu_table"] is True
    assert sorted(d["config"]["contigs_per_rank"]) == [12, 13] and d["config"]["junctions_total"] == d["multi_gpu_check"]["rows"]


def test_bench_eight_ranks_share_one_gpu():
    """BASELINE configs[3] / [4] in miniature: `bench.py --gpus 8`, eight ranks on one GPU (gloo exchange), the 25 targets dealt out
    by read count; rank 0's merged table equals the single-context table byte for byte, every rank got targets, and the line
    carries the balance of the deal (max shard / mean shard) that a real 8-GPU run can be held against."""
    d = _run_bench(["--gpus", "8"], {"PJB_BENCH_SHARE_GPU": "1"}, timeout=1500)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong"
    assert d["multi_gpu_check"]["merged_equals_single_gpu_table"] is True
    per = d["config"]["contigs_per_rank"]
    assert len(per) == 8 and sum(per) == 25 and min(per) >= 1
    bal = d["config"]["lpt_balance"]
    assert bal["ranks"] == 8 and 1.0 <= bal["max_over_mean_reads"] < 1.25 and len(bal["reads_per_rank"]) == 8


def test_bench_line_small_workload(tmp_path):
    """The whole bench line (roofline, cpu_baseline over every contig with parity, e2e through BAM bytes) on a
    scaled-down 25-contig set."""
    d = _run_bench(["--e2e-workdir", str(tmp_path / "e2e")], {})
    assert d["roofline"]["frac"] > 0 and d["roofline"]["alg_bytes_per_launch"] > 0
    assert d["cpu_baseline"]["value"] > 0 and "every device row" in d["cpu_baseline"]["sample"]
    assert "error" not in d["e2e"], d["e2e"]
    assert d["e2e"]["tab_identical_to_oracle"] is True and d["e2e"]["reads"] == d["config"]["reads_total"]
    er = d["e2e"]["early_return"]  # the opt-in early return is timed beside the default (one process): closed before the tree is gone
    assert er["wall_s_outputs_closed"] <= er["wall_s_process_tree"]


# ---- BASELINE configs[2] at full size: 200 M paired-end reads over 25 GRCh38-sized contigs, one context
@pytest.fixture(scope="module")
def c3_full():
    import torch
    from portcullis_amd import ffi, synth

    cfgs = synth.c3_contig_configs()
    ctx = ffi.Context(0, "FR")
    ctx.set_refs([c.contig_len for c in cfgs])
    data = []
    for tid, c in enumerate(cfgs):
        d = synth.generate(c, device="cuda", tid=tid)
        ctx.upload_contig_device(tid, d["genome"])
        data.append(d)
    torch.cuda.synchronize()

    def run():
        ctx.clear_rows()
        regs = []
        for tid, d in enumerate(data):
            ctx.submit_batch_device(tid, d["batch"], d["n_reads"])
            regs.append(ctx.finish_contig(tid))
        return ctx.collect(), regs

    def run_groups(queue=3, group_bases=1 << 30):
        """the bench's step: chains over groups of consecutive targets (pjb_finish_group_begin / _end), `queue` of them in flight"""
        ctx.clear_rows()
        chains = ffi.plan_groups([c.contig_len for c in cfgs], list(range(len(cfgs))), group_bases)
        regs, queued = {}, []

        def collect():
            g = queued.pop(0)
            if len(g) == 1:
                regs[g[0]] = ctx.finish_contig_end(g[0])
            else:
                regs.update(ctx.finish_group_end(g))

        for g in chains:
            for tid in g:
                ctx.submit_batch_device(tid, data[tid]["batch"], data[tid]["n_reads"])
            if len(g) == 1:
                ctx.finish_contig_begin(g[0])
            else:
                ctx.finish_group_begin(g)
            queued.append(g)
            if len(queued) >= queue:
                collect()
        while queued:
            collect()
        return ctx.collect(), [regs[t] for t in range(len(cfgs))], chains

    rows, regs = run()
    yield cfgs, data, rows, regs, run, run_groups
    ctx.close()


def test_c3_fullsize_properties_and_oracle(c3_full):
    from oracle import oracle as orc
    from portcullis_amd import synth

    cfgs, data, rows, regs, run, run_groups = c3_full
    n_reads = sum(c.n_reads for c in cfgs)
    n_pairs = sum(d["n_pairs"] for d in data)
    assert n_reads >= 199_999_000 and len(cfgs) == 25
    assert sum(r["n_reads"] for r in regs) == n_reads and sum(r["n_pairs"] for r in regs) == n_pairs
    assert sum(r["spliced"] + r["unspliced"] for r in regs) == n_reads
    assert int(rows["nb_raw"].astype(np.int64).sum()) == n_pairs            # conservation: every N op lands in one junction
    assert (rows["r1pos"] + rows["r1neg"] + rows["r2pos"] + rows["r2neg"] == rows["nb_raw"]).all()
    assert (rows["nb_ppp"] <= rows["nb_bpp"] + rows["nb_raw"]).all() and (rows["nb_rel"] <= rows["nb_um"]).all()
    key = (rows["refid"].astype(np.int64) << 48) | (rows["start"].astype(np.int64) << 20)
    assert (np.diff(key) >= 0).all()                                          # contig-major, start-sorted
    assert len(np.unique(rows[["refid", "start", "end"]])) == len(rows)       # one row per intron
    again, _ = run()
    assert hashlib.md5(again.tobytes()).hexdigest() == hashlib.md5(rows.tobytes()).hexdigest()   # deterministic
    worst = 0.0
    for tid in (0, 1, 24):                                                     # chr1, chr2 (16 M reads each) and chrM vs the oracle
        hb = synth.batch_to_numpy(data[tid]["batch"], 0, data[tid]["n_reads"])
        orows, oreg = orc.find_juncs(tid, cfgs[tid].contig_len, data[tid]["genome"].cpu().numpy().tobytes(), hb, "FR")
        region_equal(regs[tid], oreg)
        worst = max(worst, assert_rows_equal(rows[rows["refid"] == tid], orows))
    assert worst <= 1e-6


def test_c3_fullsize_group_chains_equal_per_target_chains(c3_full):
    """The bench's step at full size -- three chains over groups of consecutive targets (1 Gb each), all in flight -- must give, byte for
    byte, the row table of the per-target chains (which the test above holds against the oracle), and every target's counters."""
    cfgs, data, rows, regs, run, run_groups = c3_full
    grows, gregs, chains = run_groups()
    assert len(chains) == 3 and sorted(t for g in chains for t in g) == list(range(25))
    assert hashlib.md5(grows.tobytes()).hexdigest() == hashlib.md5(rows.tobytes()).hexdigest()
    for a, b in zip(gregs, regs):
        region_equal(a, b)
        assert a["n_pairs"] == b["n_pairs"] and a["n_junctions"] == b["n_junctions"]
    g2, _, _ = run_groups(queue=1, group_bases=1 << 29)   # other group sizes (seven chains), one chain at a time: the same table
    assert hashlib.md5(g2.tobytes()).hexdigest() == hashlib.md5(rows.tobytes()).hexdigest()


# ---- BASELINE configs[4], one rank's share: 125 M reads of a 1 B-read run on one 248 Mb contig, 300 k junctions,
# Zipf depth (the deepest junction holds millions of alignments); strandedness=firststrand does not change junc output
def test_c5_rank_share_properties_and_prefix_oracle():
    import torch
    from oracle import oracle as orc
    from portcullis_amd import ffi, synth

    cfg = synth.SynthConfig("C5-share", 248_956_422, 125_000_000, 300_000, 100, seed=5_000_001)
    d = synth.generate(cfg, device="cuda")
    torch.cuda.synchronize()
    with ffi.Context(0, "UNKNOWN", strandedness=1) as ctx:      # PJB_SS_FIRSTSTRAND: accepted, no effect
        ctx.set_refs([cfg.contig_len])
        ctx.upload_contig_device(0, d["genome"])

        def run(n):
            ctx.clear_rows()
            b = d["batch"] if n == d["n_reads"] else {k: v for k, v in d["batch"].items()}
            ctx.submit_batch_device(0, b, n)
            return ctx.finish_contig(0), ctx.collect()

        reg, rows = run(d["n_reads"])
        assert reg["n_reads"] == cfg.n_reads and reg["n_pairs"] == d["n_pairs"]
        assert int(rows["nb_raw"].astype(np.int64).sum()) == d["n_pairs"]
        assert int(rows["nb_raw"].max()) > 1_000_000                           # a junction with > 10^6 supporting alignments
        deep = rows[np.argmax(rows["nb_raw"])]
        assert deep["jad"][0] <= deep["nb_raw"] and deep["nb_dist"] >= 1 and 0 <= deep["entropy"] <= np.log2(deep["nb_raw"]) + 1e-9
        key = rows["start"].astype(np.int64) << 32 | rows["end"].astype(np.int64)
        assert (np.diff(key) > 0).all() and len(rows) > 250_000
        reg2, rows2 = run(d["n_reads"])
        assert hashlib.md5(rows2.tobytes()).hexdigest() == hashlib.md5(rows.tobytes()).hexdigest()
    # a 3 M-read prefix of the same records against the oracle (cig_off / seq_off of a prefix are a valid batch)
    n = 3_000_000
    hb = synth.batch_to_numpy(d["batch"], 0, n)
    genome = d["genome"].cpu().numpy().tobytes()
    orows, oreg = orc.find_juncs(0, cfg.contig_len, genome, hb, "UNKNOWN")
    with ffi.Context(0, "UNKNOWN", strandedness=1) as ctx:
        ctx.set_refs([cfg.contig_len])
        ctx.upload_contig_device(0, d["genome"])
        ctx.clear_rows()
        ctx.submit_batch(0, hb)
        dreg = ctx.finish_contig(0)
        region_equal(dreg, oreg)
        assert assert_rows_equal(ctx.collect(), orows) <= 1e-6
