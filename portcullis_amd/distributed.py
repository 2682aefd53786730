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


def gather_region(region, device, group=None):
    """The same reduction as allreduce_region with ONE collective and one host sync: every rank's five counters are
    all-gathered and folded locally."""
    world = dist.get_world_size(group)
    mine = torch.tensor([region["spliced"], region["unspliced"], region["sum_len"], region["min_len"], region["max_len"]],
                        dtype=torch.int64, device=device)
    out = torch.empty((world, 5), dtype=torch.int64, device=device)
    dist.all_gather(list(out.unbind(0)), mine, group=group)
    h = out.cpu()
    return dict(spliced=int(h[:, 0].sum()), unspliced=int(h[:, 1].sum()), sum_len=int(h[:, 2].sum()),
                min_len=int(h[:, 3].min()), max_len=int(h[:, 4].max()))


class RegionExchange:
    """The counters of gather_region without a host sync per call: add() launches one small asynchronous
    all-gather per contig into its own slot, result() folds everything at the end of the run."""

    def __init__(self, slots, device, group=None):
        self.group, self.device = group, torch.device(device)
        self.world = dist.get_world_size(group)
        self.cuda = self.device.type == "cuda"
        self.mine_host = torch.zeros((slots, 5), dtype=torch.int64, pin_memory=self.cuda)
        self.mine = torch.zeros((slots, 5), dtype=torch.int64, device=self.device)
        self.out = torch.zeros((slots, self.world, 5), dtype=torch.int64, device=self.device)
        self.mine_np = self.mine_host.numpy()  # same memory: filled without creating tensors
        self.nccl = dist.get_backend(group) == "nccl"
        self.used = 0
        self.work = []

    def add(self, region):
        k = self.used
        if k >= self.mine.shape[0]:
            raise ValueError("RegionExchange: more contigs than slots")
        self.used += 1
        self.mine_np[k] = (region["spliced"], region["unspliced"], region["sum_len"], region["min_len"], region["max_len"])
        self.mine[k].copy_(self.mine_host[k], non_blocking=True)
        if self.nccl:
            self.work.append(dist.all_gather_into_tensor(self.out[k].view(-1), self.mine[k], group=self.group, async_op=True))
        else:
            self.work.append(dist.all_gather(list(self.out[k].unbind(0)), self.mine[k], group=self.group, async_op=True))

    def result(self):
        for w in self.work:
            w.wait()
        self.work = []
        h = self.out[: self.used].cpu().reshape(-1, 5)
        if len(h) == 0:
            return dict(spliced=0, unspliced=0, sum_len=0, min_len=2**31 - 1, max_len=0)
        return dict(spliced=int(h[:, 0].sum()), unspliced=int(h[:, 1].sum()), sum_len=int(h[:, 2].sum()),
                    min_len=int(h[:, 3].min()), max_len=int(h[:, 4].max()))


class DeviceRows:
    """A device pointer as a torch tensor source (the rows pjb_collect_device returns)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class RowExchange:
    """All-gather of the per-rank junction tables, device to device, off the critical path.

    Rows are PODs of `row_bytes` bytes.  Every rank owns a send slot of `cap` rows plus a 16-byte header (its row
    count) and a receive buffer of world x slot.  start() copies the rank's rows (a uint8 tensor on `device`, e.g.
    the view of pjb_collect_device) into the send slot and launches the collective asynchronously: RCCL moves the
    slots over xGMI while the next contig's kernels run.  Rank `root` brings the gathered buffer to page-locked host
    memory: after every exchange on a side stream, or once in finish().  The next start() (or finish()) waits for
    what is in flight before the buffers are reused.
    """
    HDR = 16

    def __init__(self, row_bytes, cap_rows, device, group=None, root=0, host_copy="every"):
        """host_copy: "every" = the root copies each gathered table to the host (asynchronously), "final" = only
        finish() does (one copy per job: the tables of earlier exchanges stay in HBM until overwritten)."""
        self.host_copy = host_copy
        self.group, self.device, self.row_bytes = group, torch.device(device), row_bytes
        self.world, self.rank, self.root = dist.get_world_size(group), dist.get_rank(group), root
        self.cuda = self.device.type == "cuda"
        self.work = None
        self.copy_done = None
        self.counts = None
        self._alloc(cap_rows)
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None

    def _alloc(self, cap_rows):
        self.cap = int(cap_rows)
        self.slot = self.HDR + self.cap * self.row_bytes
        self.send = torch.zeros(self.slot, dtype=torch.uint8, device=self.device)
        self.recv = torch.zeros(self.world * self.slot, dtype=torch.uint8, device=self.device)
        self.host = None
        if self.rank == self.root:
            self.host = torch.zeros(self.world * self.slot, dtype=torch.uint8, pin_memory=self.cuda)
        self.count_host = torch.zeros(1, dtype=torch.int64, pin_memory=self.cuda)  # staging for the header
        self.count_np = self.count_host.numpy()
        self.count_u8 = self.count_host.view(torch.uint8)
        self.send_hdr = self.send[:8]

    def _wait(self):
        if self.work is not None:
            self.work.wait()  # the current stream waits for the collective (the host does not block on CUDA)
            self.work = None
        if self.copy_done is not None:
            self.copy_done.synchronize()
            self.copy_done = None

    def start(self, rows_u8, n_rows):
        """rows_u8: uint8 tensor with at least n_rows * row_bytes bytes on `device`."""
        self._wait()
        if n_rows > self.cap:  # every rank must grow together: callers size cap from a maximum agreed up front
            raise ValueError(f"RowExchange: {n_rows} rows exceed the agreed capacity {self.cap}")
        nb = n_rows * self.row_bytes
        self.count_np[0] = n_rows  # (reused safely: the stream is synchronised below before start() returns)
        self.send_hdr.copy_(self.count_u8, non_blocking=True)
        if nb:
            self.send[self.HDR:self.HDR + nb] = rows_u8[:nb]
        if self.cuda:
            # rows_u8 usually aliases a buffer its owner rewrites on another stream (pjb_collect_device): the copy
            # out of it (microseconds) must have happened before this returns
            torch.cuda.current_stream(self.device).synchronize()
        else:
            self.send_hdr.copy_(self.count_u8)  # (CPU tensors: plain copy)
        if dist.get_backend(self.group) == "nccl":
            self.work = dist.all_gather_into_tensor(self.recv, self.send, group=self.group, async_op=True)
        else:
            self.work = dist.all_gather(list(self.recv.view(self.world, self.slot).unbind(0)), self.send, group=self.group,
                                        async_op=True)
        if self.rank == self.root and self.host_copy == "every":
            if self.cuda:
                ready = torch.cuda.Event()
                self.side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self.side):
                    self.work.wait()  # on the side stream: the copy follows the collective, not the next kernels
                    self.host.copy_(self.recv, non_blocking=True)
                    ready.record(self.side)
                self.copy_done = ready
                # the main stream must still not overwrite send / recv before the collective is done
            else:
                self.work.wait()
                self.host.copy_(self.recv)

    def finish(self):
        """Wait for the exchange in flight; on the root returns the merged table (uint8 numpy, rank order)."""
        self._wait()
        if self.rank == self.root and self.host_copy != "every":
            self.host.copy_(self.recv)  # after the wait: ordered behind the collective on the current stream
        if self.cuda:
            torch.cuda.current_stream(self.device).synchronize()
        if self.rank != self.root:
            return None
        h = self.host.numpy().reshape(self.world, self.slot)
        counts = [int(h[r, :8].view(np.int64)[0]) for r in range(self.world)]
        self.counts = counts
        parts = [h[r, self.HDR:self.HDR + counts[r] * self.row_bytes] for r in range(self.world) if counts[r]]
        return np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)


class MirrorExchange:
    """RowExchange without a copy or a synchronisation of its own on the sending side: the send slots are handed to
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
