"""ctypes binding of the C ABI (include/portcullis_amd.h).

This is the Python-side stub a maintainer would write to drive the device
path; it adds no computation of its own.  There is deliberately no CPU
fallback: if libportcullis_amd.so is missing, or no MI355X is visible,
loading / creating a context raises.
"""
import ctypes as C
import os

import numpy as np

from .records import ORIENTATION, ReadBatch, pack_seq2

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libportcullis_amd.so")
ABI_VERSION = 4
MAX_QUEUED = 8  # PJB_MAX_QUEUED
N_STAGES = 8
STAGE_NAMES = ["scan_emit", "sort", "group", "anchors", "pair_stats", "finalize", "d2h", "spare"]

EXPORTS = [
    "pjb_create", "pjb_destroy", "pjb_last_error", "pjb_set_refs", "pjb_upload_contig", "pjb_upload_contig_device", "pjb_upload_contig_fasta",
    "pjb_release_contig", "pjb_submit_batch", "pjb_submit_batch_device", "pjb_finish_contig", "pjb_finish_contig_begin",
    "pjb_finish_contig_end", "pjb_finish_ready", "pjb_finish_group_begin", "pjb_finish_group_end", "pjb_collect",
    "pjb_clear_rows", "pjb_get_timing", "pjb_device_count", "pjb_get_kernel_timing", "pjb_reset_kernel_timing",
    "pjb_select_timed_kernels", "pjb_host_alloc", "pjb_host_free", "pjb_host_register", "pjb_host_unregister", "pjb_inflate_bgzf", "pjb_deflate_bgzf", "pjb_submit_bam", "pjb_collect_device", "pjb_set_row_mirror",
    "pjb_extra_finish", "pjb_set_option", "pjb_merge_rows", "pjb_plan_groups", "pjb_bam_begin", "pjb_bam_piece", "pjb_bam_pieces_done", "pjb_bam_end", "pjb_bam_inflate_done", "pjb_filter_set_junctions", "pjb_filter_batch", "pjb_filt_features",
]
N_FEATURES = 34
KMER_TABLE = 3125 * 5
PW_LEN = 32
FEATURE_NAMES = ["Genuine", "rna_usrs", "rna_dist", "rna_rel", "rna_entropy", "rna_rel2raw", "rna_maxminanc", "rna_maxmmes",
                 "rna_missmatch", "rna_intron", "dna_minhamm", "dna_coding", "dna_pws", "dna_ss"] + [f"JAD{i:02d}" for i in range(1, 21)]
FLAG_KERNEL_TIMING = 1
FLAG_NO_CHAINS = 4  # bamfilt's contexts: the chain slots' and the rows' streams are created with the first chain, not by pjb_create
FLAG_EXTRA = 2  # junc --extra: batches carry name_hash, extra_finish() yields mm_score / coverage / up_aln / down_aln


class PjbConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("orientation", C.c_int32),
                ("strandedness", C.c_int32), ("flags", C.c_uint32)]


class PjbBatch(C.Structure):
    _fields_ = [("n_reads", C.c_int64)] + [
        (n, C.c_void_p) for n in ("pos", "flag", "mapq", "xs", "l_qseq", "mtid", "mpos", "cig_off", "cigar", "seq_off", "seq4",
                                  "name_hash", "seq2", "seq_exc")
    ]


class PjbMarkovModels(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("exon", "intron", "donor_t", "donor_f", "acceptor_t", "acceptor_f", "donor_pw", "acceptor_pw")] + [
        (n, C.c_int32) for n in ("exon_size", "intron_size", "donor_pw_size", "acceptor_pw_size")]


class PjbRegionResult(C.Structure):
    _fields_ = [("spliced", C.c_uint64), ("unspliced", C.c_uint64), ("sum_len", C.c_uint64), ("min_len", C.c_int32),
                ("max_len", C.c_int32), ("n_reads", C.c_int64), ("n_pairs", C.c_int64), ("n_junctions", C.c_int64)]


class PjbTiming(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("stage_ms", C.c_float * N_STAGES), ("sort_passes", C.c_int64),
                ("generic_pairs", C.c_int64), ("generic_reads", C.c_int64), ("position_runs", C.c_int64), ("candidates", C.c_int64), ("checked_reads", C.c_int64), ("repeats", C.c_int64), ("repeat_reasons", C.c_int64)]


class PjbKernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_int64), ("total_ms", C.c_double)]


ROW_DTYPE = np.dtype(
    [
        ("refid", "<i4"), ("start", "<i4"), ("end", "<i4"), ("left", "<i4"), ("right", "<i4"),
        ("read_strand", "u1"), ("ss_strand", "u1"), ("cons_strand", "u1"), ("canonical", "u1"),
        ("da1", "u1", (2,)), ("da2", "u1", (2,)), ("suspicious", "u1"), ("_pad", "u1", (3,)),
        ("nb_raw", "<u4"), ("nb_dist", "<u4"), ("nb_ms", "<u4"), ("nb_um", "<u4"), ("nb_bpp", "<u4"), ("nb_ppp", "<u4"),
        ("nb_rel", "<u4"), ("r1pos", "<u4"), ("r1neg", "<u4"), ("r2pos", "<u4"), ("r2neg", "<u4"),
        ("max_min_anc", "<u4"), ("maxmmes", "<u4"), ("hamming5p", "<u4"), ("hamming3p", "<u4"),
        ("nb_up_juncs", "<u4"), ("nb_down_juncs", "<u4"), ("jad", "<u4", (20,)), ("_pad2", "<u4"),
        ("sum_mismatches", "<u8"), ("entropy", "<f8"),
    ]
)
assert ROW_DTYPE.itemsize == 200
EXTRA_DTYPE = np.dtype([("mm_score", "<f8"), ("coverage", "<f8"), ("up_aln", "<u4"), ("down_aln", "<u4")])
assert EXTRA_DTYPE.itemsize == 24


class PjbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pjb error {code}: {msg}")
        self.code = code


_LIB = None


def load():
    """Load the HIP library; raises (never falls back) if it has not been built.

    In a process that also uses torch on the GPU, import torch first: torch ships its own libamdhip64 and the copy
    that is loaded first serves both libraries (torch does not find its devices through the system one)."""
    global _LIB, LIB_PATH
    if _LIB is None:
        # PJB_LIB_PATH: another build of the same library (kernel A/B runs: tools/build_variants.sh); never a different backend
        LIB_PATH = os.environ.get("PJB_LIB_PATH", LIB_PATH)
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the junc hot path is HIP-only; there is no CPU fallback)"
            )
        L = C.CDLL(LIB_PATH)
        L.pjb_last_error.restype = C.c_char_p
        L.pjb_last_error.argtypes = [C.c_void_p]
        L.pjb_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(PjbConfig)]
        L.pjb_destroy.argtypes = [C.c_void_p]
        L.pjb_destroy.restype = None
        L.pjb_set_refs.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.pjb_upload_contig.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        L.pjb_upload_contig_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        L.pjb_upload_contig_fasta.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_int)]
        L.pjb_release_contig.argtypes = [C.c_void_p, C.c_int32]
        L.pjb_submit_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(PjbBatch)]
        L.pjb_submit_batch_device.argtypes = [C.c_void_p, C.c_int32, C.POINTER(PjbBatch)]
        L.pjb_finish_contig.argtypes = [C.c_void_p, C.c_int32, C.POINTER(PjbRegionResult)]
        L.pjb_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.pjb_bam_begin.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
        L.pjb_bam_piece.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.pjb_bam_pieces_done.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.pjb_bam_end.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
        L.pjb_finish_contig_begin.argtypes = [C.c_void_p, C.c_int32]
        L.pjb_finish_contig_end.argtypes = [C.c_void_p, C.c_int32, C.POINTER(PjbRegionResult)]
        L.pjb_finish_ready.argtypes = [C.c_void_p]
        L.pjb_finish_group_begin.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32]
        L.pjb_finish_group_end.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.POINTER(PjbRegionResult)]
        L.pjb_set_row_mirror.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.pjb_collect_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.pjb_inflate_bgzf.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.pjb_deflate_bgzf.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]
        L.pjb_submit_bam.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_int64)]
        L.pjb_collect.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.pjb_extra_finish.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.pjb_filter_set_junctions.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        L.pjb_filter_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(PjbBatch), C.c_int32, C.c_void_p]
        L.pjb_filt_features.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_uint32, C.POINTER(PjbMarkovModels), C.c_void_p]
        L.pjb_host_alloc.restype = C.c_void_p
        L.pjb_host_alloc.argtypes = [C.c_size_t]
        L.pjb_host_free.restype = None
        L.pjb_host_free.argtypes = [C.c_void_p]
        L.pjb_device_count.restype = C.c_int
        L.pjb_device_count.argtypes = []
        L.pjb_clear_rows.argtypes = [C.c_void_p]
        L.pjb_plan_groups.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]
        L.pjb_merge_rows.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(PjbRegionResult)]
        L.pjb_get_timing.argtypes = [C.c_void_p, C.POINTER(PjbTiming)]
        L.pjb_get_kernel_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        L.pjb_reset_kernel_timing.argtypes = [C.c_void_p]
        L.pjb_select_timed_kernels.argtypes = [C.c_void_p, C.c_char_p]
        _LIB = L
    return _LIB


def device_count():
    return load().pjb_device_count()


def merge_rows(gathered, n_ranks, slot_stride_bytes):
    """pjb_merge_rows: the gathered send slots (uint8 numpy, n_ranks * slot_stride_bytes) -> (rows in refid order, folded counters).
    Host arithmetic of the library; needs no context and no device."""
    L = load()
    g = np.ascontiguousarray(gathered, dtype=np.uint8)
    assert g.size >= n_ranks * slot_stride_bytes
    cap = max((int(slot_stride_bytes) - 64) // ROW_DTYPE.itemsize, 0) * int(n_ranks)
    out = np.zeros(max(cap, 1), dtype=ROW_DTYPE)
    n = C.c_int64()
    tot = PjbRegionResult()
    rc = L.pjb_merge_rows(g.ctypes.data, n_ranks, slot_stride_bytes, out.ctypes.data, cap, C.byref(n), C.byref(tot))
    if rc:
        raise PjbError(rc, "pjb_merge_rows: malformed slot header or too many rows")
    return out[: n.value].copy(), {k: getattr(tot, k) for k, _ in PjbRegionResult._fields_}


_FIELDS = [("pos", np.int32), ("flag", np.uint16), ("mapq", np.uint8), ("xs", np.uint8), ("l_qseq", np.int32),
           ("mtid", np.int32), ("mpos", np.int32), ("cig_off", np.uint32), ("cigar", np.uint32),
           ("seq_off", np.uint32), ("seq4", np.uint8)]


class Context:
    """One device context (pjb_ctx).  Mirrors the calls JunctionBuilder::findJunctions makes."""

    def __init__(self, device=0, orientation="UNKNOWN", strandedness=3, flags=0, abi_version=ABI_VERSION):
        """abi_version: what the caller was compiled against (3: a caller that knows nothing of pjb_batch.seq2 / pjb_timing.repeats)."""
        self._L = load()
        self._h = C.c_void_p()
        self._abi = int(abi_version)
        ori = ORIENTATION[orientation] if isinstance(orientation, str) else int(orientation)
        cfg = PjbConfig(self._abi, device, ori, strandedness, flags)
        rc = self._L.pjb_create(C.byref(self._h), C.byref(cfg))
        if rc:
            raise PjbError(rc, self._L.pjb_last_error(None).decode())
        self._keep = []
        self._keep_batch = {}  # tid -> device tensors lent to the context until the contig is collected

    def close(self):
        if self._h:
            self._L.pjb_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise PjbError(rc, self._L.pjb_last_error(self._h).decode(errors="replace"))

    def set_refs(self, ref_lens):
        a = np.ascontiguousarray(ref_lens, dtype=np.int32)
        self._check(self._L.pjb_set_refs(self._h, len(a), a.ctypes.data))

    def upload_contig(self, tid, bases):
        b = bases if isinstance(bases, (bytes, bytearray)) else bytes(bases)
        self._check(self._L.pjb_upload_contig(self._h, tid, b, len(b)))

    def upload_contig_fasta(self, tid, raw, line_bases, line_width, length):
        """raw: the record's sequence lines as they are in the FASTA file.  False: not laid out as stated, nothing uploaded."""
        ok = C.c_int(0)
        self._check(self._L.pjb_upload_contig_fasta(self._h, tid, bytes(raw), len(raw), line_bases, line_width, length, C.byref(ok)))
        return bool(ok.value)

    def upload_contig_device(self, tid, tensor):
        """tensor: torch uint8 CUDA tensor of UPPER-CASE bases; must outlive the context's use of it."""
        self._keep.append(tensor)
        self._check(self._L.pjb_upload_contig_device(self._h, tid, tensor.data_ptr(), tensor.numel()))

    def release_contig(self, tid):
        self._check(self._L.pjb_release_contig(self._h, tid))

    def submit_batch(self, tid, batch: ReadBatch, seq2=None):
        """seq2: None = as PJB_FFI_SEQ2 says (default 1): the batch goes over with its bases in 2 bits as well (records.pack_seq2, what an
        ABI-4 decoder does); False = seq4 only (the compares then run on the 4-bit codes)."""
        if seq2 is None:
            seq2 = os.environ.get("PJB_FFI_SEQ2", "1") != "0"
        pb = PjbBatch()
        if self._abi < 4:  # (an ABI-3 caller's struct ends at name_hash: whatever lies behind it is not the library's to read)
            seq2 = False
            pb.seq2 = pb.seq_exc = 0xDEADBEEF0
        pb.n_reads = batch.n
        keep = []
        for name, dt in _FIELDS:
            a = np.ascontiguousarray(getattr(batch, name), dtype=dt)
            if a.size == 0:
                a = np.zeros(4, dtype=dt)
            keep.append(a)
            setattr(pb, name, a.ctypes.data)
        nh = getattr(batch, "name_hash", None)
        if nh is not None:
            nh = np.ascontiguousarray(nh, dtype=np.uint64)
            if nh.size == 0:
                nh = np.zeros(2, dtype=np.uint64)
            keep.append(nh)
            pb.name_hash = nh.ctypes.data
        if seq2 and batch.n:
            s2, sx = pack_seq2(batch.seq4, batch.seq_off, batch.l_qseq)
            s2 = np.ascontiguousarray(np.concatenate([s2, np.zeros(2, dtype=np.uint16)]))  # (read as words: an even number of granules)
            sx = np.ascontiguousarray(sx) if sx.size else np.zeros(1, dtype=np.uint32)
            keep += [s2, sx]
            pb.seq2, pb.seq_exc = s2.ctypes.data, sx.ctypes.data
        self._check(self._L.pjb_submit_batch(self._h, tid, C.byref(pb)))

    def submit_batch_device(self, tid, tensors, n_reads):
        """tensors: dict name -> torch CUDA tensor with the dtypes of `pjb_batch`; borrowed until finish."""
        pb = PjbBatch()
        pb.n_reads = int(n_reads)
        for name, _ in _FIELDS:
            t = tensors[name]
            self._keep_batch.setdefault(tid, []).append(t)
            setattr(pb, name, t.data_ptr())
        if tensors.get("name_hash") is not None:
            self._keep_batch.setdefault(tid, []).append(tensors["name_hash"])
            pb.name_hash = tensors["name_hash"].data_ptr()
        if tensors.get("seq2") is not None and tensors.get("seq_exc") is not None and os.environ.get("PJB_FFI_SEQ2", "1") != "0":
            self._keep_batch.setdefault(tid, []) .extend([tensors["seq2"], tensors["seq_exc"]])
            pb.seq2, pb.seq_exc = tensors["seq2"].data_ptr(), tensors["seq_exc"].data_ptr()
        self._check(self._L.pjb_submit_batch_device(self._h, tid, C.byref(pb)))

    def finish_contig(self, tid):
        r = PjbRegionResult()
        try:
            self._check(self._L.pjb_finish_contig(self._h, tid, C.byref(r)))
        finally:
            self._keep_batch.pop(tid, None)
        return {k: getattr(r, k) for k, _ in PjbRegionResult._fields_}

    def set_option(self, name, value):
        self._check(self._L.pjb_set_option(self._h, name.encode(), int(value)))

    def finish_contig_begin(self, tid):
        """Queue the contig's kernel chain without waiting (at most MAX_QUEUED contigs queued; collect in the same order)."""
        try:
            self._check(self._L.pjb_finish_contig_begin(self._h, tid))
        except Exception:
            self._keep_batch.pop(tid, None)
            raise

    def finish_contig_end(self, tid):
        r = PjbRegionResult()
        try:
            self._check(self._L.pjb_finish_contig_end(self._h, tid, C.byref(r)))
        finally:
            self._keep_batch.pop(tid, None)
        return {k: getattr(r, k) for k, _ in PjbRegionResult._fields_}

    def finish_ready(self):
        """True if collecting the oldest queued chain would not wait for the device."""
        return bool(self._L.pjb_finish_ready(self._h))

    def finish_group_begin(self, tids):
        """Queue ONE kernel chain over several targets (pjb_finish_group_begin); collect with finish_group_end(tids)."""
        arr = (C.c_int32 * len(tids))(*tids)
        try:
            self._check(self._L.pjb_finish_group_begin(self._h, arr, len(tids)))
        except Exception:
            for t in tids:
                self._keep_batch.pop(t, None)
            raise

    def finish_group_end(self, tids):
        """-> {tid: region result} for the group queued with the same tids."""
        arr = (C.c_int32 * len(tids))(*tids)
        res = (PjbRegionResult * len(tids))()
        try:
            self._check(self._L.pjb_finish_group_end(self._h, arr, len(tids), res))
        finally:
            for t in tids:
                self._keep_batch.pop(t, None)
        return {t: {k: getattr(res[i], k) for k, _ in PjbRegionResult._fields_} for i, t in enumerate(tids)}

    def collect(self, copy=True):
        """Rows built so far.  copy=False returns a view of the context's pinned buffer that is only
        valid until the next finish_contig / clear_rows / close."""
        p = C.c_void_p()
        n = C.c_int64()
        self._check(self._L.pjb_collect(self._h, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, dtype=ROW_DTYPE)
        buf = (C.c_char * (n.value * ROW_DTYPE.itemsize)).from_address(p.value)
        a = np.frombuffer(buf, dtype=ROW_DTYPE, count=n.value)
        return a.copy() if copy else a

    def set_row_mirror(self, device_ptr, cap_bytes):
        """Every following finish_contig writes a 64-byte header (int64 n_rows, spliced, unspliced, sum_len, min_len,
        max_len) and the contig's rows into this device buffer (0 to stop)."""
        self._check(self._L.pjb_set_row_mirror(self._h, C.c_void_p(device_ptr or None), cap_bytes))

    def collect_device(self):
        """(device pointer, row count) of the rows of the contig finished last, still in HBM."""
        p = C.c_void_p()
        n = C.c_int64()
        self._check(self._L.pjb_collect_device(self._h, C.byref(p), C.byref(n)))
        return (p.value or 0), n.value

    def extra_finish(self, copy=True):
        """calcExtraMetrics once every contig of the file is finished (FLAG_EXTRA contexts): one EXTRA_DTYPE record
        per row of collect(), same order.  copy=False: a view of the context's page-locked table (valid until the next
        clear_rows / close)."""
        p = C.c_void_p()
        n = C.c_int64()
        self._check(self._L.pjb_extra_finish(self._h, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, dtype=EXTRA_DTYPE)
        buf = (C.c_char * (n.value * EXTRA_DTYPE.itemsize)).from_address(p.value)
        a = np.frombuffer(buf, dtype=EXTRA_DTYPE, count=n.value)
        return a.copy() if copy else a

    def filter_set_junctions(self, tid, starts, ends):
        """bamfilt: the junctions of target `tid` that passed the filter (any order; duplicates are dropped)."""
        keys = np.unique((np.asarray(starts, dtype=np.int64).astype(np.uint32).astype(np.uint64) << np.uint64(32))
                         | np.asarray(ends, dtype=np.int64).astype(np.uint32).astype(np.uint64))
        self._check(self._L.pjb_filter_set_junctions(self._h, tid, keys.ctypes.data_as(C.c_void_p), len(keys)))

    def filter_batch(self, tid, batch, clip_mode="HARD"):
        """bamfilt: one code per alignment (0 drop, 1 unspliced kept, 2 spliced kept, 3 multiply spliced kept)."""
        pb = PjbBatch()
        pb.n_reads = batch.n
        keep = []
        for name, dt in (("pos", np.int32), ("cig_off", np.uint32), ("cigar", np.uint32)):
            a = np.ascontiguousarray(getattr(batch, name), dtype=dt)
            if a.size == 0:
                a = np.zeros(4, dtype=dt)
            keep.append(a)
            setattr(pb, name, a.ctypes.data)
        out = np.zeros(max(batch.n, 1), dtype=np.uint8)
        mode = {"HARD": 0, "SOFT": 1, "COMPLETE": 2}[clip_mode]
        self._check(self._L.pjb_filter_batch(self._h, tid, C.byref(pb), mode, out.ctypes.data_as(C.c_void_p)))
        return out[: batch.n]

    def filt_features(self, rows, mean_read_length, l95, models):
        """ModelFeatures::setRow for `rows` (ROW_DTYPE): float64 [n, N_FEATURES].  models: dict name -> float64 table
        (exon, intron, donor_t, donor_f, acceptor_t, acceptor_f: KMER_TABLE; donor_pw, acceptor_pw: PW_LEN * 5; missing /
        None = untrained) plus exon_size, intron_size, donor_pw_size, acceptor_pw_size."""
        rows = np.ascontiguousarray(rows, dtype=ROW_DTYPE)
        m = PjbMarkovModels()
        keep = []
        for name in ("exon", "intron", "donor_t", "donor_f", "acceptor_t", "acceptor_f", "donor_pw", "acceptor_pw"):
            t = models.get(name)
            if t is not None:
                t = np.ascontiguousarray(t, dtype=np.float64)
                assert t.size == (PW_LEN * 5 if name.endswith("_pw") else KMER_TABLE), name
                keep.append(t)
                setattr(m, name, t.ctypes.data)
        for name in ("exon_size", "intron_size", "donor_pw_size", "acceptor_pw_size"):
            setattr(m, name, int(models.get(name, 0)))
        out = np.zeros((max(len(rows), 1), N_FEATURES), dtype=np.float64)
        self._check(self._L.pjb_filt_features(self._h, rows.ctypes.data_as(C.c_void_p), len(rows), float(mean_read_length), int(l95),
                                              C.byref(m), out.ctypes.data_as(C.c_void_p)))
        return out[: len(rows)]

    def clear_rows(self):
        self._check(self._L.pjb_clear_rows(self._h))

    def submit_bam_pieces(self, tid, comp, first_uoffset, piece_sizes):
        """submit_bam with the bytes handed over in pieces of the given sizes (cycled; pjb_bam_begin / _piece / _end)."""
        comp = np.frombuffer(bytes(comp), dtype=np.uint8) if not isinstance(comp, np.ndarray) else comp
        self._check(self._L.pjb_bam_begin(self._h, tid, len(comp)))
        at, k, last = 0, 0, C.c_int64()
        keep = []
        while at < len(comp):
            nb = min(int(piece_sizes[k % len(piece_sizes)]), len(comp) - at)
            piece = np.ascontiguousarray(comp[at:at + nb])
            keep.append(piece)
            self._check(self._L.pjb_bam_piece(self._h, tid, piece.ctypes.data_as(C.c_void_p), nb, C.byref(last)))
            at += nb
            k += 1
        n = C.c_int64()
        self._check(self._L.pjb_bam_end(self._h, tid, first_uoffset, C.byref(n)))
        done = C.c_int64()
        self._check(self._L.pjb_bam_pieces_done(self._h, C.byref(done)))
        assert done.value >= last.value  # (everything was consumed by _end)
        return n.value

    def submit_bam(self, tid, comp, first_uoffset):
        """All alignments of target `tid` from the BGZF bytes `comp` (whole blocks, starting with the block
        that holds the target's first record at inflated offset `first_uoffset`): inflate, parse and
        transcode on the device.  Returns the number of alignments added."""
        comp = np.frombuffer(bytes(comp), dtype=np.uint8) if not isinstance(comp, np.ndarray) else comp
        n = C.c_int64()
        self._check(self._L.pjb_submit_bam(self._h, tid, comp.ctypes.data_as(C.c_void_p), len(comp), first_uoffset, C.byref(n)))
        return n.value

    def inflate_bgzf(self, comp):
        """Inflate a run of whole BGZF blocks (bytes-like) on the device; returns the inflated bytes."""
        comp = np.frombuffer(bytes(comp), dtype=np.uint8)
        n = C.c_int64()
        # capacity from the ISIZE fields is not known up front: BGZF blocks inflate to at most 64 KB each,
        # a block is at least 28 bytes
        cap = max(65536, (len(comp) // 28 + 1) * 65536) if len(comp) < (1 << 22) else len(comp) * 12
        out = np.empty(cap, dtype=np.uint8)
        self._check(self._L.pjb_inflate_bgzf(self._h, comp.ctypes.data_as(C.c_void_p), len(comp),
                                             out.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return out[:n.value].tobytes()

    def deflate_bgzf(self, data, block_bytes=0xff00):
        """BGZF-compress bytes on the device: returns (the members back to back -- no EOF block --, their sizes)."""
        data = np.frombuffer(bytes(data), dtype=np.uint8)
        n_blocks = (len(data) + block_bytes - 1) // block_bytes
        cap = max(1, n_blocks) * 65536
        out = np.empty(cap, dtype=np.uint8)
        sizes = np.zeros(max(1, n_blocks), dtype=np.uint32)
        n = C.c_int64()
        self._check(self._L.pjb_deflate_bgzf(self._h, data.ctypes.data_as(C.c_void_p), len(data), block_bytes, out.ctypes.data_as(C.c_void_p), cap,
                                             C.byref(n), sizes.ctypes.data_as(C.c_void_p)))
        return out[:n.value].tobytes(), sizes[:n_blocks]

    def timing(self):
        t = PjbTiming()
        t.repeats = t.repeat_reasons = -77  # (an ABI-3 context must leave what lies behind checked_reads alone)
        self._check(self._L.pjb_get_timing(self._h, C.byref(t)))
        return dict(total_ms=t.total_ms, stage_ms={STAGE_NAMES[i]: t.stage_ms[i] for i in range(N_STAGES)},
                    sort_passes=t.sort_passes, generic_pairs=t.generic_pairs, generic_reads=t.generic_reads, checked_reads=t.checked_reads,
                    position_runs=t.position_runs, candidates=t.candidates, repeats=t.repeats, repeat_reasons=t.repeat_reasons)


def _kernel_timing(self):
    n = C.c_int32()
    self._check(self._L.pjb_get_kernel_timing(self._h, None, 0, C.byref(n)))
    arr = (PjbKernelTime * max(1, n.value))()
    self._check(self._L.pjb_get_kernel_timing(self._h, arr, n.value, C.byref(n)))
    return {arr[i].name.decode(): (arr[i].launches, arr[i].total_ms) for i in range(n.value)}


def _reset_kernel_timing(self):
    self._check(self._L.pjb_reset_kernel_timing(self._h))


def _select_timed_kernels(self, names):
    self._check(self._L.pjb_select_timed_kernels(self._h, ",".join(names).encode()))


Context.select_timed_kernels = _select_timed_kernels
Context.kernel_timing = _kernel_timing
Context.reset_kernel_timing = _reset_kernel_timing


def run_contig(ctx, tid, genome, batches):
    """Convenience: upload genome, submit batches, finish; returns (rows, region)."""
    ctx.upload_contig(tid, genome)
    ctx.clear_rows()
    for b in batches:
        ctx.submit_batch(tid, b)
    reg = ctx.finish_contig(tid)
    return ctx.collect(), reg


GROUP_MAX = 32  # PJB_GROUP_MAX
GROUP_GAP = 4096


def plan_groups(ref_lens, tids, max_bases=1 << 30):
    """pjb_plan_groups: the chains `tids` (in this order) are finished as -- consecutive runs whose sequences (plus the gap
    pjb_finish_group_begin leaves between members) stay below `max_bases` and PJB_GROUP_MAX members; a set that would be one chain of
    more than 0.6 Gb goes as two.  GRCh38's 25 sequences in index order give three groups of about 1 Gb.  The program, bench.py and the
    ranks of a multi-GPU run all plan with this function of the library."""
    tids = [int(t) for t in tids]
    if not tids:
        return []
    lens = np.ascontiguousarray(ref_lens, dtype=np.int32)
    t = np.ascontiguousarray(tids, dtype=np.int32)
    assert t.min() >= 0 and t.max() < len(lens)
    g = np.zeros(len(t), dtype=np.int32)
    n = load().pjb_plan_groups(lens.ctypes.data, t.ctypes.data, len(t), int(max_bases), g.ctypes.data)
    if n < 0:
        raise PjbError(n, "pjb_plan_groups: bad arguments")
    return [[tids[k] for k in range(len(tids)) if g[k] == i] for i in range(n)]
