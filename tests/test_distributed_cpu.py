"""world_size-2 gloo tests of the N>1 plumbing (no GPU): contig sharding, the read-length
all-reduce and the junction-table all-gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from portcullis_amd import distributed as pd
from portcullis_amd.ffi import ROW_DTYPE


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rows(rank, n):
    r = np.zeros(n, dtype=ROW_DTYPE)
    r["refid"] = rank
    r["start"] = np.arange(n) * 10 + rank
    r["end"] = r["start"] + 5
    r["nb_raw"] = rank * 1000 + np.arange(n)
    r["entropy"] = 0.25 * (rank + 1)
    return r


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    n = [3, 0][rank] if world == 2 else rank + 1
    merged = pd.allgather_rows(_rows(rank, n), dev)
    reg = dict(spliced=10 * (rank + 1), unspliced=5, sum_len=1000 * (rank + 1), min_len=50 + rank, max_len=100 + rank)
    tot = pd.allreduce_region(reg, dev)
    q.put((rank, merged.tobytes(), tot))
    dist.destroy_process_group()


def test_allgather_and_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect = np.concatenate([_rows(0, 3), _rows(1, 0)])
    for rank, blob, tot in got:
        merged = np.frombuffer(blob, dtype=ROW_DTYPE)
        assert merged.tobytes() == expect.tobytes()
        assert tot == dict(spliced=30, unspliced=10, sum_len=3000, min_len=50, max_len=101)


def test_shard_contigs_lpt():
    w = [248, 242, 198, 190, 181, 170, 159, 145, 138, 133, 135, 133, 114, 107, 101, 90, 83, 80, 58, 64, 46, 50, 156, 57, 1]
    parts = pd.shard_contigs(w, 8)
    assert sorted(i for p in parts for i in p) == list(range(len(w)))
    loads = [sum(w[i] for i in p) for p in parts]
    assert max(loads) <= 1.15 * (sum(w) / 8)
    assert pd.shard_contigs([5], 4)[0] == [0]
    assert pd.shard_contigs([], 2) == [[], []]


def _worker_mirror(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = pd.MirrorExchange(ROW_DTYPE.itemsize, cap_rows=8, device=torch.device("cpu"))
    outs = []
    for step, n in enumerate([[3, 0], [8, 5], [1, 2]]):
        ptr, nbytes = x.slot_for_next_finish()
        slot = x.send[x.k]
        assert slot.data_ptr() == ptr and nbytes == 64 + 8 * ROW_DTYPE.itemsize
        # what pjb_finish_contig does with the mirror: header + rows
        rows = _rows(rank + 10 * step, n[rank])
        hdr = np.array([n[rank], 7 + rank, 1, 100 * (step + 1), 40 + rank, 90 + step, 0, 0], dtype=np.int64)
        slot[:64] = torch.from_numpy(hdr.view(np.uint8).copy())
        if n[rank]:
            slot[64:64 + rows.nbytes] = torch.from_numpy(rows.view(np.uint8).copy())
        x.launch()
    merged = x.finish()
    q.put((rank, None if merged is None else merged.tobytes(), x.counts, x.regions))
    dist.destroy_process_group()


def test_mirror_exchange_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_mirror, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, blob, counts, regions in got:
        if rank == 0:
            assert blob == np.concatenate([_rows(20, 1), _rows(21, 2)]).tobytes()
            assert counts == [1, 2]
            assert regions[1] == dict(spliced=8, unspliced=1, sum_len=300, min_len=41, max_len=92)
        else:
            assert blob is None


# ---- configs[3]: the same contig set sharded over ranks, one exchange per step, merged table == single-GPU table
_WEIGHTS = [2490, 2420, 1980, 1900, 1810, 1700, 1590, 1450, 1380, 1330, 1350, 1330, 1140, 1070, 1010, 900, 830, 800, 580,
            640, 460, 500, 1560, 570, 1]


def _contig_rows(tid):
    n = (_WEIGHTS[tid] % 7) + (0 if tid == 24 else 1)  # the last contig has no junction at all
    r = np.zeros(n, dtype=ROW_DTYPE)
    r["refid"] = tid
    r["start"] = 100 * np.arange(n) + tid
    r["end"] = r["start"] + 50
    r["nb_raw"] = 1 + np.arange(n) + tid
    return r


def _contig_region(tid):
    w = _WEIGHTS[tid]
    return dict(spliced=w // 3, unspliced=w - w // 3, sum_len=150 * w, min_len=150 - (tid % 3), max_len=150 + (tid % 5))


def _finish_into_slot(slot, tid, state):
    """What pjb_finish_contig does with the row mirror: rows appended, header = row total + folded counters."""
    rows, reg = _contig_rows(tid), _contig_region(tid)
    at = 64 + state["rows"] * ROW_DTYPE.itemsize
    if len(rows):
        slot[at:at + rows.nbytes] = torch.from_numpy(rows.view(np.uint8).copy())
    state["rows"] += len(rows)
    state["acc"] = [state["acc"][0] + reg["spliced"], state["acc"][1] + reg["unspliced"], state["acc"][2] + reg["sum_len"],
                    min(state["acc"][3], reg["min_len"]), max(state["acc"][4], reg["max_len"])]
    hdr = np.array([state["rows"]] + state["acc"] + [0, 0], dtype=np.int64)
    slot[:64] = torch.from_numpy(hdr.view(np.uint8).copy())


def _worker_sharded(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = pd.shard_contigs(_WEIGHTS, world)[rank]
    cap = sum(len(_contig_rows(t)) for t in range(len(_WEIGHTS)))
    x = pd.MirrorExchange(ROW_DTYPE.itemsize, cap_rows=cap, device=torch.device("cpu"))
    for _step in range(3):
        x.slot_for_next_finish()
        slot = x.send[x.k]
        st = dict(rows=0, acc=[0, 0, 0, 2**31 - 1, 0])  # pjb_clear_rows / pjb_set_row_mirror reset the accumulation
        for tid in mine:
            _finish_into_slot(slot, tid, st)
        x.launch()
    merged = x.finish()
    out = None
    if rank == 0:
        rows, totals = pd.merge_rank_tables(merged, ROW_DTYPE, x.regions)
        # the same merge through the C ABI's receive side (pjb_merge_rows: what a C++ caller runs behind its ncclAllGather)
        from portcullis_amd import ffi
        nrows, ntot = ffi.merge_rows(x.host.numpy(), world, x.slot)
        assert nrows.tobytes() == rows.tobytes() and {k: ntot[k] for k in totals} == totals
        assert ntot["n_junctions"] == len(rows) and ntot["n_reads"] == totals["spliced"] + totals["unspliced"]
        out = (rows.tobytes(), totals, x.counts)
    q.put((rank, mine, out))
    dist.destroy_process_group()


def test_sharded_contig_set_merges_to_single_table_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(2)), key=lambda g: g[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, mine0, out0), (_, mine1, out1) = got
    assert sorted(mine0 + mine1) == list(range(len(_WEIGHTS))) and not set(mine0) & set(mine1)
    load = [sum(_WEIGHTS[t] for t in m) for m in (mine0, mine1)]
    assert abs(load[0] - load[1]) <= 0.02 * sum(_WEIGHTS)  # longest-processing-time balance
    blob, totals, counts = out0
    single = np.concatenate([_contig_rows(t) for t in range(len(_WEIGHTS))])  # what one GPU produces: contigs in order
    assert blob == single.tobytes()
    assert counts == [sum(len(_contig_rows(t)) for t in m) for m in (mine0, mine1)]
    regs = [_contig_region(t) for t in range(len(_WEIGHTS))]
    assert totals == dict(spliced=sum(r["spliced"] for r in regs), unspliced=sum(r["unspliced"] for r in regs),
                          sum_len=sum(r["sum_len"] for r in regs), min_len=min(r["min_len"] for r in regs),
                          max_len=max(r["max_len"] for r in regs))
    assert out1 is None


def test_merge_rows_edge_cases():
    """pjb_merge_rows alone (host arithmetic of the library, no device): empty ranks, a target's rows staying in their order, interleaved
    targets across ranks, malformed headers."""
    from portcullis_amd import ffi

    stride = 64 + 6 * ROW_DTYPE.itemsize

    def slot(rows, reg):
        b = np.zeros(stride, dtype=np.uint8)
        b[:48] = np.array([len(rows), reg[0], reg[1], reg[2], reg[3], reg[4]], dtype=np.int64).view(np.uint8)
        if len(rows):
            b[64:64 + rows.nbytes] = rows.view(np.uint8)
        return b

    def rows_of(tids):
        return np.concatenate([_contig_rows(t)[:2] for t in tids]) if tids else np.zeros(0, dtype=ROW_DTYPE)

    a, b, c = rows_of([5, 2, 9]), rows_of([]), rows_of([7, 0, 3])  # ranks finish their targets in any order
    g = np.concatenate([slot(a, (10, 1, 1500, 149, 151)), slot(b, (0, 0, 0, 2**31 - 1, 0)), slot(c, (4, 2, 900, 150, 153))])
    rows, tot = ffi.merge_rows(g, 3, stride)
    want = np.concatenate([_contig_rows(t)[:2] for t in (0, 2, 3, 5, 7, 9)])
    assert rows.tobytes() == want.tobytes()
    assert (tot["spliced"], tot["unspliced"], tot["sum_len"], tot["min_len"], tot["max_len"], tot["n_reads"], tot["n_junctions"]) == (14, 3, 2400, 149, 153, 17, len(want))
    bad = g.copy()
    bad[:8] = np.array([7], dtype=np.int64).view(np.uint8)  # more rows than a slot holds
    with pytest.raises(ffi.PjbError):
        ffi.merge_rows(bad, 3, stride)
    bad[:8] = np.array([-1], dtype=np.int64).view(np.uint8)
    with pytest.raises(ffi.PjbError):
        ffi.merge_rows(bad, 3, stride)
