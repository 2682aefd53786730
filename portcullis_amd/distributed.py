"""Multi-GPU plumbing for the junc path: one process per GPU (torch.distributed; backend "nccl"
is RCCL on ROCm, "gloo" in the CPU tests).

The path shards by reference contig exactly as the reference's thread pool does
(src/junction_builder.cc:241-245): every metric of a junction depends only on the alignments of
its own contig, so ranks work independently and meet twice, both tiny:
  * all-reduce of the read-length counters (sum, count, min, max) -- the global mean read length
    feeds `mean_readlen` and `pfp` (src/junction_builder.cc:276-278, junction_system.cc:311-318)
  * all-gather of the per-rank junction rows -- the merge of JunctionSystem::append
    (src/junction_builder.cc:258-269); rows are PODs of ffi.ROW_DTYPE
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_contigs(weights, world_size):
    """Longest-processing-time partition of contigs over ranks.  weights: per-contig cost estimate
    (alignment count from the index, else contig length).  Returns list of contig-id lists."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    load = [0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += weights[i]
    for l in out:
        l.sort()
    return out


def allreduce_region(region, device, group=None):
    """Sum / min / max of the RegionResult counters over ranks (dict in, dict out)."""
    s = torch.tensor([region["spliced"], region["unspliced"], region["sum_len"]], dtype=torch.int64, device=device)
    mn = torch.tensor([region["min_len"]], dtype=torch.int64, device=device)
    mx = torch.tensor([region["max_len"]], dtype=torch.int64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    return dict(spliced=int(s[0]), unspliced=int(s[1]), sum_len=int(s[2]), min_len=int(mn[0]), max_len=int(mx[0]))


def allgather_rows(rows, device, group=None):
    """All-gather variable-length row tables (numpy structured array) -> one table on every rank,
    ordered by rank.  Rows travel as raw bytes padded to the longest table."""
    world = dist.get_world_size(group)
    itemsize = rows.dtype.itemsize
    n = torch.tensor([len(rows)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c) for c in counts]
    nmax = max(max(counts), 1)
    buf = torch.zeros((nmax, itemsize), dtype=torch.uint8, device=device)
    if len(rows):
        host = torch.from_numpy(np.ascontiguousarray(rows).view(np.uint8).reshape(len(rows), itemsize).copy())
        buf[: len(rows)] = host.to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    parts = [o[:c].cpu().numpy().reshape(-1).view(rows.dtype) for o, c in zip(out, counts) if c]
    return np.concatenate(parts) if parts else np.zeros(0, dtype=rows.dtype)
