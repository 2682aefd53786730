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


def merge_rank_tables(merged_u8, row_dtype, regions):
    """What rank 0 does with the gathered slots: the ranks' rows (rank order, each rank's contigs in its finish
    order) become ONE table in contig order -- every contig's rows are contiguous and already (start, end)-sorted, so
    a stable sort on refid is JunctionSystem::sort (lib/src/junction_system.cc:322-330) -- and the per-rank
    read-length counters fold into the global ones (src/junction_builder.cc:258-278).
    Returns (rows, totals dict)."""
    rows = np.ascontiguousarray(merged_u8).view(row_dtype)
    rows = rows[np.argsort(rows["refid"], kind="stable")]
    totals = dict(spliced=sum(r["spliced"] for r in regions), unspliced=sum(r["unspliced"] for r in regions),
                  sum_len=sum(r["sum_len"] for r in regions), min_len=min(r["min_len"] for r in regions),
                  max_len=max(r["max_len"] for r in regions))
    return rows, totals


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


class MirrorExchange:
    """All-gather of the per-rank junction tables, device to device, without a copy or a synchronisation of its own on the
    sending side: the send slots are handed to
    the library (pjb_set_row_mirror), whose pjb_finish_contig leaves header + rows in them, so that after
    finish_contig returns launch() only starts the asynchronous all-gather.

    Two send slots alternate: the collective of contig k reads slot k % 2 while finish_contig of contig k + 1 fills
    the other one; launch() first makes sure the previous collective is complete (it had a whole contig's time), so
    the slot handed out next is free again.  Header: int64 n_rows, spliced, unspliced, sum_len, min_len, max_len.
    """
    HDR = 64

    def __init__(self, row_bytes, cap_rows, device, group=None, root=0):
        self.group, self.device, self.row_bytes = group, torch.device(device), row_bytes
        self.world, self.rank, self.root = dist.get_world_size(group), dist.get_rank(group), root
        self.cuda = self.device.type == "cuda"
        self.nccl = dist.get_backend(group) == "nccl"
        self.cap = int(cap_rows)
        self.slot = self.HDR + self.cap * row_bytes
        self.send = [torch.zeros(self.slot, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.recv = torch.zeros(self.world * self.slot, dtype=torch.uint8, device=self.device)
        self.host = torch.zeros(self.world * self.slot, dtype=torch.uint8, pin_memory=self.cuda) if self.rank == root else None
        self.k = 0
        self.work = None
        self.counts = None
        self.regions = None

    def slot_for_next_finish(self):
        """(device pointer, bytes) of the slot the next finish_contig must fill."""
        return self.send[self.k].data_ptr(), self.slot

    def _drain(self):
        if self.work is not None:
            self.work.wait()
            if self.cuda:
                torch.cuda.current_stream(self.device).synchronize()  # host-side: the other slot may be rewritten now
            self.work = None

    def launch(self):
        """The slot handed out last is filled (finish_contig has returned): start its all-gather."""
        self._drain()
        src = self.send[self.k]
        if self.nccl:
            self.work = dist.all_gather_into_tensor(self.recv, src, group=self.group, async_op=True)
        elif self.cuda:
            # gloo cannot all-gather device tensors: staged through the host (debug runs of several ranks on one GPU)
            parts = [torch.empty(self.slot, dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(parts, src.cpu(), group=self.group)
            self.recv.copy_(torch.cat(parts))
        else:
            self.work = dist.all_gather(list(self.recv.view(self.world, self.slot).unbind(0)), src, group=self.group, async_op=True)
        self.k ^= 1

    def finish(self):
        """Wait for the exchange in flight; on the root: the merged table of the last exchange (uint8 numpy, rank
        order), with .counts (rows per rank) and .regions (the five counters per rank) set."""
        self._drain()
        if self.rank != self.root:
            return None
        self.host.copy_(self.recv)
        if self.cuda:
            torch.cuda.current_stream(self.device).synchronize()
        h = self.host.numpy().reshape(self.world, self.slot)
        hdr = np.stack([h[r, :48].view(np.int64) for r in range(self.world)])
        self.counts = [int(x) for x in hdr[:, 0]]
        self.regions = [dict(spliced=int(a[1]), unspliced=int(a[2]), sum_len=int(a[3]), min_len=int(a[4]), max_len=int(a[5])) for a in hdr]
        parts = [h[r, self.HDR:self.HDR + self.counts[r] * self.row_bytes] for r in range(self.world) if self.counts[r]]
        return np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)
