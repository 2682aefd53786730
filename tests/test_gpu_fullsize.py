"""Parity at BASELINE.json's full single-GPU size (configs[1]: 10 M reads, 1 contig, ~50 k
junctions): the oracle finishes this workload in about a second, so the device rows are compared
with it directly, plus the size-independent properties (conservation of pairs, determinism,
batch-split invariance)."""
import hashlib

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
    orows, oreg = orc.find_juncs(0, cfg.contig_len, data["genome"].cpu().numpy().tobytes(), hb.to_oracle(), "UNKNOWN")
    region_equal(reg, oreg)
    assert_rows_equal(rows, orows)


@pytest.mark.parametrize("env", [{"PJB_RADIX_BITS": "6"}, {"PJB_RADIX_BITS": "10"}, {"PJB_RADIX_BITS": "12"}])
def test_sort_variants_agree(c2, monkeypatch, env):
    """The radix sort's digit width (unrolled 9/10/11-bit and generic match loops, 12-bit digits with
    more than 64 KB of LDS) must not change a single byte of the row table."""
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


def test_bench_line_small_workload(tmp_path):
    """The whole bench line (roofline, cpu_baseline over every contig with parity, e2e through BAM bytes) on a
    scaled-down 25-contig set."""
    d = _run_bench(["--e2e-workdir", str(tmp_path / "e2e")], {})
    assert d["roofline"]["frac"] > 0 and d["roofline"]["alg_bytes_per_launch"] > 0
    assert d["cpu_baseline"]["value"] > 0 and "every device row" in d["cpu_baseline"]["sample"]
    assert d["e2e"]["tab_identical_to_oracle"] is True and d["e2e"]["reads"] == d["config"]["reads_total"]
