#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X `junc` hot path on BASELINE.json's synthetic workload.

A "step" is one pass of the hot path over one contig's alignment records that are
already resident in HBM: pjb_submit_batch_device + pjb_finish_contig (CIGAR scan/emit,
sort/group, anchors, per-pair match statistics, junction reduce) + the junction rows
copied back to the host.  At N > 1 every rank owns one contig of the same size (the path
shards by reference contig, src/junction_builder.cc:241-245) and the only exchange is the
RCCL all-gather that merges the per-rank junction tables.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def algorithmic_bytes(name, N, C, S, Cs, P, J, L, passes, Pg=0):
    """Algorithmic HBM bytes of ONE launch of kernel `name` (each byte counted once per logical
    pass, caches ignored; DESIGN.md section 4 derives these).  N reads, C cigar ops, S spliced
    reads, Cs cigar ops of spliced reads, P pairs, J junctions, L read length."""
    frags = P / 64.0 + J
    table = {
        # pos, cig_off, l_qseq, xs + every cigar op; 8 B written per spliced read (compacted index + pair offset)
        "k1_count": N * 13 + C * 4 + S * 8,
        # spliced reads only: compacted slot (8), cig_off (8), pos/flag/mapq/xs (8), ops fetched once; 36 B written per pair
        "k1_emit": S * 24 + Cs * 4 + P * 36,
        "rs_hist": P * 8,
        "rs_scatter": P * 24,
        "k2_heads_reduce": P * 20,
        "k2_heads_apply": P * 20 + P * 4 + (J + J + P / 8) * 4,
        "k3_anchors_frag": P * 16 + frags * 12,
        # simple pairs: meta, read ordinal, key, pos, rend, seq_off (8), packed read bases L/2, 4-bit genome codes L/2,
        # 8-byte result; other pairs only read their meta word
        "k4a_simple": (P - Pg) * (4 + 4 + 8 + 4 + 4 + 8 + L + 8) + Pg * 4,
        # generic pairs: list entry, idx, jid, key, ordinal, cig_off (8), ops, pos/aend, l_qseq, seq_off (8), anchors (8),
        # read bases + genome codes (L), result
        "k4b_generic": Pg * (4 + 4 + 4 + 8 + 4 + 8 + 4 * (Cs / max(S, 1)) + 8 + 4 + 8 + 8 + L + 8),
        # key, idx, jid, pos, aend, result, meta, previous pair (16), lstart, rend, updown; one 192-B fragment record
        "k4_pairs": P * (8 + 4 + 4 + 4 + 4 + 8 + 4 + 16 + 4 + 4 + 4) + frags * 196,
        "k5_frag_reduce": frags * 196 + J * 164,
        "k5_finalize": J * (192 + 24 + 48 + 200),
        "k5_entropy_terms": P * 0 + J * 8,
    }
    return table.get(name)


def main():
    # exactly one line on stdout: everything the libraries print (RCCL banners, ...) goes to stderr
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=os.environ.get("PJB_BENCH_CONFIG", "C2"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # PJB_BENCH_FORCE_EXCHANGE=1 runs the N > 1 exchange code with a one-rank group (single-GPU check of that path)
    force_x = world == 1 and os.environ.get("PJB_BENCH_FORCE_EXCHANGE") == "1"
    if world > 1 or force_x:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_x:
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev)
    multi = world > 1 or force_x

    cfg = synth.CONFIGS[args.config]
    t_gen = time.time()
    data = synth.generate(cfg, device=dev, seed=cfg.seed + rank)
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen
    batch, genome = data["batch"], data["genome"]
    N, P, C, S = data["n_reads"], data["n_pairs"], data["n_cigar_ops"], data["n_spliced"]
    L = cfg.read_len

    ctx = ffi.Context(device=local_rank, orientation="UNKNOWN", flags=ffi.FLAG_KERNEL_TIMING)
    ctx.set_refs([cfg.contig_len])
    ctx.upload_contig_device(0, genome)

    state = {}
    xchg = None
    row_bytes = ffi.ROW_DTYPE.itemsize

    def step():
        ctx.clear_rows()
        ctx.submit_batch_device(0, batch, N)
        if xchg is not None:
            ctx.set_row_mirror(*xchg.slot_for_next_finish())  # finish_contig leaves header + rows in the exchange slot
        reg = ctx.finish_contig(0)
        rows = ctx.collect(copy=False)  # view of the pinned row table
        if xchg is not None:
            # the path's only exchange: the merge of the per-rank junction tables and read-length counters (they ride in
            # the slot's header): all-gather over RCCL / xGMI straight from HBM, asynchronous -- it overlaps the next
            # contig's kernels.  Every rank's own rows are on its host after finish_contig; rank 0 copies the merged
            # table to its host once, at the end of the timed region (a job merges once, not once per contig)
            xchg.launch()
        state["reg"] = reg
        state["rows"] = rows

    for _ in range(args.warmup):
        step()
    if multi:
        if not state:
            step()
        # slot size of the row exchange: the largest table any rank produced in the warm-up, with headroom
        jmax = torch.tensor([int(state["reg"]["n_junctions"])], device=dev, dtype=torch.int64)
        dist.all_reduce(jmax, op=dist.ReduceOp.MAX)
        xchg = pd.MirrorExchange(row_bytes, int(jmax.item()) * 5 // 4 + 64, dev)
        step()  # one untimed step with the exchange (buffers, communicator warm-up)
        xchg.finish()
    # per-kernel table from a few fully instrumented steps (outside the timed region) ...
    ctx.reset_kernel_timing()
    n_prof = 3
    for _ in range(n_prof):
        step()
    kt_all = ctx.kernel_timing()
    dominant = max(kt_all.items(), key=lambda kv: kv[1][1])[0]
    # ... and only the dominant kernel keeps its HIP-event bracket inside the timed region
    ctx.select_timed_kernels([dominant])
    ctx.reset_kernel_timing()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    merged = xchg.finish() if xchg is not None else None  # the last exchange completes inside the timed region
    if xchg is not None and rank == 0:
        tot = xchg.regions  # per-rank read-length counters of the last exchange -> the global ones
        state["totals"] = dict(spliced=sum(r["spliced"] for r in tot), unspliced=sum(r["unspliced"] for r in tot),
                               sum_len=sum(r["sum_len"] for r in tot), min_len=min(r["min_len"] for r in tot),
                               max_len=max(r["max_len"] for r in tot))
        assert state["totals"]["spliced"] + state["totals"]["unspliced"] >= N
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    reg, rows = state["reg"], state["rows"].copy()
    if xchg is not None and rank == 0:
        # merged table on rank 0: every rank's rows, in rank order; this rank's part must be its own table
        mt = merged.view(ffi.ROW_DTYPE)
        assert len(mt) == sum(xchg.counts) and xchg.counts[0] == len(rows)
        assert mt[: len(rows)].tobytes() == rows.tobytes()
    J = int(reg["n_junctions"])
    assert reg["n_pairs"] == P and reg["n_reads"] == N
    # size-independent sanity (full-size parity properties are in tests/test_gpu_fullsize.py)
    assert int(rows["nb_raw"].sum()) == P

    totals = torch.tensor([N, J], device=dev, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(totals)
    reads_total, junc_total = int(totals[0]), int(totals[1])

    # ---- per-kernel device time over the timed region (HIP events on the context's stream)
    kt_timed = ctx.kernel_timing()
    kt = {k: (v[0] / n_prof * args.steps, v[1] / n_prof * args.steps) for k, v in kt_all.items()}
    kt[dominant] = kt_timed[dominant]  # measured live over the timed region
    timing = ctx.timing()
    sort_passes = int(timing["sort_passes"])
    cs_ops = None
    if rank == 0:
        # cigar ops of spliced reads (for the byte formulas)
        cig_off = batch["cig_off"].to(torch.int64)
        n_ops = cig_off[1:] - cig_off[:-1]
        seq_off = batch["seq_off"].to(torch.int64)
        spl = (seq_off[1:] - seq_off[:-1]) > 0
        cs_ops = int(n_ops[spl].sum())
    result = None
    if rank == 0:
        kern = []
        for name, (launches, ms) in kt.items():
            if launches == 0:
                continue
            avg = ms / launches
            b = algorithmic_bytes(name, N, C, S, cs_ops, P, J, L, sort_passes, int(timing.get("generic_pairs", 0)))
            kern.append(dict(name=name, launches=launches, avg_ms=avg, total_ms=ms,
                             alg_bytes=b, gbps=(b / (avg * 1e-3) / 1e9) if b else None))
        kern.sort(key=lambda k: -k["total_ms"])
        dom = kern[0]
        peak = 8000.0
        roofline = dict(bound="hbm", kernel=dom["name"], achieved=round(dom["gbps"], 1) if dom["gbps"] else None,
                        peak=peak, unit="GB/s", frac=round(dom["gbps"] / peak, 4) if dom["gbps"] else None,
                        traffic=None, avg_kernel_ms=round(dom["avg_ms"], 4),
                        alg_bytes_per_launch=int(dom["alg_bytes"]) if dom["alg_bytes"] else None)
        kernel_ms_per_step = sum(k["total_ms"] for k in kern) / args.steps
        # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc
        # passes of this same command: tools/pmc_traffic.sh -> tools/summarize_pmc.py); counters cannot be
        # read from inside the process, so the last committed measurement for this workload is attached
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
        if args.config == "C2" and os.path.exists(tpath):
            try:
                tr = json.load(open(tpath)).get("pjb::" + dom["name"])
                if tr:
                    roofline["traffic"] = tr["hbm_bytes_per_launch"]
                    roofline["traffic_source"] = "profiles/pmc_traffic_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
            except Exception:
                pass

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(data, cfg, ctx, ffi, synth)

        result = {
            "metric": "junc_reads_per_sec",
            "value": reads_total * args.steps / elapsed,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1] ({cfg.name}): synthetic {cfg.n_reads} single-end {L}-bp reads, "
                                   f"1 contig of {cfg.contig_len} bp per GPU, {J} junctions, {P} spliced pairs",
                       "reads_per_gpu": N, "pairs_per_gpu": P, "junctions_per_gpu": J, "sharding": "by contig",
                       "input": "device-resident SoA records (pjb_submit_batch_device)"},
            "junctions_per_sec": junc_total * args.steps / elapsed,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "device_kernel_ms_per_step": round(kernel_ms_per_step, 4),
            "pipeline_gbps": None,
            "sort_passes": sort_passes,
            "generic_pairs": int(timing.get("generic_pairs", 0)),
            "kernels": [dict(name=k["name"], launches_per_step=k["launches"] / args.steps, avg_ms=round(k["avg_ms"], 4),
                             gbps=round(k["gbps"], 1) if k["gbps"] else None) for k in kern],
            "datagen_s": round(t_gen, 2),
            "finish_contig_event_ms": round(timing["total_ms"], 4),
        }
        # pipeline_gbps: sum over kernels of (bytes per launch x launches per step) / device kernel time per step
        tot_bytes = sum((k["alg_bytes"] or 0) * k["launches"] / args.steps for k in kern)
        result["pipeline_gbps"] = round(tot_bytes / (kernel_ms_per_step * 1e-3) / 1e9, 1)
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    ctx.set_row_mirror(0, 0)
    ctx.close()
    if multi:
        dist.destroy_process_group()


def cpu_baseline(data, cfg, ctx, ffi, synth):
    """Time the CPU oracle (single thread, kind "port") on a bounded prefix of the same records
    and check the device rows for that prefix against it."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as orc
    from parity import assert_rows_equal, region_equal

    N = data["n_reads"]
    genome_host = data["genome"].cpu().numpy().tobytes()
    # pilot on 1M reads to size the sample for ~15 s of CPU work
    pilot = min(N, 1_000_000)
    hb = synth.batch_to_numpy(data["batch"], 0, pilot)
    t = time.perf_counter()
    orows, oreg = orc.find_juncs(0, cfg.contig_len, genome_host, hb.to_oracle(), "UNKNOWN")
    dt = time.perf_counter() - t
    rate = pilot / dt
    M = int(min(N, max(pilot, rate * 15.0)))
    reps = 1
    if M > pilot:
        hb = synth.batch_to_numpy(data["batch"], 0, M)
        ob = hb.to_oracle()
        # the whole workload fits the budget several times over: repeat it (about 10 s of CPU work) and take the median
        reps = int(max(1, min(12, round(10.0 / max(M / rate, 1e-3)))))
        times = []
        for _ in range(reps):
            t = time.perf_counter()
            orows, oreg = orc.find_juncs(0, cfg.contig_len, genome_host, ob, "UNKNOWN")
            times.append(time.perf_counter() - t)
        times.sort()
        dt = times[len(times) // 2]
    # parity of the device path on exactly this sample
    ctx.clear_rows()
    ctx.submit_batch(0, hb)
    dreg = ctx.finish_contig(0)
    drows = ctx.collect()
    region_equal(dreg, oreg)
    max_ent = assert_rows_equal(drows, orows)
    return {"value": M / dt, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": f"first {M} of {N} records of the same workload ({len(orows)} junctions), oracle/portcullis_oracle.c, "
                      f"median of {reps} runs of {dt:.2f} s; device rows for the sample match the oracle "
                      f"(max |entropy diff| {max_ent:.2g})",
            "junctions_per_sec": len(orows) / dt}


if __name__ == "__main__":
    main()
