"""world_size-2 gloo tests of the N>1 plumbing (no GPU): contig sharding, the read-length
all-reduce and the junction-table all-gather."""
import os
import socket

import numpy as np
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
