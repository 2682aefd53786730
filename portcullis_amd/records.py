"""Fixed-width alignment-record batches (structure of arrays).

This is the host-side image of `pjb_batch` (include/portcullis_amd.h): the
fields of a BAM record that the `junc` path reads
(lib/src/bam_alignment.cc:71-100 caches exactly these), in BAM-native
encodings so a decoder can copy them without re-encoding:

* ``cigar``  uint32 ``len<<4|op`` with op indexing ``MIDNSHP=XB``
* ``seq4``   4-bit packed bases, high nibble first (``=ACMGRSVTWYHKDBN``);
  each read's bytes start on a 4-byte boundary and ``seq_off`` counts
  4-byte words (so one batch can address 16 GiB of sequence with 32 bits)
* ``xs``     0 = no ``XS:A`` tag / '?' / '.', 1 = '+', 2 = '-', 3 = any other
  value (the reference throws, lib/include/portcullis/bam/bam_master.hpp:60-72)

Only reads with at least one N operation need sequence bytes; others may
have an empty slice.
"""
from dataclasses import dataclass

import numpy as np

CIGAR_CHARS = "MIDNSHP=XB"
NT16 = "=ACMGRSVTWYHKDBN"
_NT16_CODE = {c: i for i, c in enumerate(NT16)}
XS_CODE = {None: 0, "?": 0, ".": 0, "+": 1, "-": 2}

ORIENTATION = {"SE": 0, "FR": 1, "RF": 2, "FF": 3, "UNKNOWN": 4}


def encode_cigar(s):
    out = []
    num = ""
    for ch in s:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num) << 4) | CIGAR_CHARS.index(ch))
            num = ""
    return np.array(out, dtype=np.uint32)


def decode_cigar(ops):
    return "".join(f"{int(o) >> 4}{CIGAR_CHARS[int(o) & 15]}" for o in ops)


def encode_seq(s):
    codes = np.fromiter((_NT16_CODE[c] for c in s), dtype=np.uint8, count=len(s))
    if len(codes) % 2:
        codes = np.append(codes, np.uint8(0))
    return ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8)


# 2-bit bases (pjb_batch.seq2 / .seq_exc, ABI 4): per seq4 byte the two bases' codes (A 0, C 1, G 2, T 3; anything else 0) in four
# bits, low base first; per BAM code whether it is one of A, C, G, T
_CODE2 = np.zeros(16, dtype=np.uint8)
_CODE2[[1, 2, 4, 8]] = [0, 1, 2, 3]
_VALID2 = np.zeros(16, dtype=bool)
_VALID2[[1, 2, 4, 8]] = True
_b = np.arange(256)
_V2 = (_CODE2[_b >> 4] | (_CODE2[_b & 15] << 2)).astype(np.uint8)       # (BAM: high nibble = the first base of the byte)
_BAD2 = ((~_VALID2[_b >> 4]).astype(np.uint8) + (~_VALID2[_b & 15]).astype(np.uint8)).astype(np.uint8)
del _b


def pack_seq2(seq4, seq_off, l_qseq):
    """What a decoder writes beside seq4 for ABI 4: (seq2 uint16[words of seq4], seq_exc uint32[(n + 31) // 32]) -- the same bases in
    2 bits, a 16-bit granule per seq4 word, and per read whether it must NOT be compared in 2 bits (a base outside ACGT among its
    l_qseq bases, or fewer bases than l_qseq).  A format conversion: nothing of the path's arithmetic."""
    seq4 = np.ascontiguousarray(seq4, dtype=np.uint8)
    seq_off = np.asarray(seq_off, dtype=np.int64)
    lq = np.asarray(l_qseq, dtype=np.int64)
    n = len(lq)
    n_words = int(seq_off[n]) if n else 0
    b = seq4[: 4 * n_words]
    v = _V2[b]
    seq2 = (v[0::2] | (v[1::2] << 4)).astype(np.uint8).view(np.uint16) if n_words else np.zeros(0, dtype=np.uint16)
    # per read: characters outside ACGT among its first l_qseq bases = those of its whole bytes + the high nibble of an odd last one
    bad_cum = np.concatenate([[0], np.cumsum(_BAD2[b], dtype=np.int64)])
    first = 4 * seq_off[:n]
    have = (seq_off[1 : n + 1] - seq_off[:n]) * 8
    full = np.minimum(lq, have).clip(min=0)
    cnt = bad_cum[first + full // 2] - bad_cum[first]
    odd = (full % 2 == 1) & (full > 0)
    idx = np.minimum(first + full // 2, max(len(b) - 1, 0))
    last_hi = (b[idx] >> 4) if len(b) else np.zeros(n, dtype=np.uint8)
    cnt = cnt + (odd & ~_VALID2[last_hi])
    exc = (cnt > 0) | (have < lq) | (lq <= 0)
    bits = np.zeros(((n + 31) // 32) * 32, dtype=np.uint8)
    bits[:n] = exc
    seq_exc = np.packbits(bits, bitorder="little").view(np.uint32) if n else np.zeros(0, dtype=np.uint32)
    return seq2, seq_exc


@dataclass
class ReadBatch:
    """Alignment records of one contig, in BAM file order."""

    pos: np.ndarray       # int32[n]
    flag: np.ndarray      # uint16[n]
    mapq: np.ndarray      # uint8[n]
    xs: np.ndarray        # uint8[n]
    l_qseq: np.ndarray    # int32[n]
    mtid: np.ndarray      # int32[n]
    mpos: np.ndarray      # int32[n]
    cig_off: np.ndarray   # uint32[n+1]
    cigar: np.ndarray     # uint32[cig_off[n]]
    seq_off: np.ndarray   # uint32[n+1]  (4-byte words)
    seq4: np.ndarray      # uint8[4*seq_off[n]]
    name_hash: np.ndarray = None  # uint64[n], optional: std::hash of BamAlignment::deriveName() (junc --extra)

    @property
    def n(self):
        return len(self.pos)

    @property
    def n_refskip(self):
        return int(np.count_nonzero((self.cigar & 15) == 3))

    @staticmethod
    def from_reads(reads):
        """reads: iterable of dicts with keys pos, cigar (str), seq (str or None),
        and optional flag, mapq, xs ('+', '-', None), mtid, mpos."""
        reads = list(reads)
        n = len(reads)
        pos = np.zeros(n, np.int32)
        flag = np.zeros(n, np.uint16)
        mapq = np.zeros(n, np.uint8)
        xs = np.zeros(n, np.uint8)
        lq = np.zeros(n, np.int32)
        mtid = np.full(n, -1, np.int32)
        mpos = np.full(n, -1, np.int32)
        cig_off = np.zeros(n + 1, np.uint32)
        seq_off = np.zeros(n + 1, np.uint32)
        cigs, seqs = [], []
        for i, r in enumerate(reads):
            pos[i] = r["pos"]
            flag[i] = r.get("flag", 0)
            mapq[i] = r.get("mapq", 60)
            x = r.get("xs")
            xs[i] = XS_CODE[x] if x in XS_CODE else 3
            mtid[i] = r.get("mtid", -1)
            mpos[i] = r.get("mpos", -1)
            c = r["cigar"]
            c = encode_cigar(c) if isinstance(c, str) else np.asarray(c, np.uint32)
            cigs.append(c)
            cig_off[i + 1] = cig_off[i] + len(c)
            s = r.get("seq")
            if s is None or s == "*":
                lq[i] = r.get("l_qseq", 0)
                seq_off[i + 1] = seq_off[i]
            else:
                lq[i] = len(s)
                b = encode_seq(s)
                pad = (-len(b)) % 4
                if pad:
                    b = np.concatenate([b, np.zeros(pad, np.uint8)])
                seqs.append(b)
                seq_off[i + 1] = seq_off[i] + len(b) // 4
        cigar = np.concatenate(cigs) if cigs else np.zeros(0, np.uint32)
        seq4 = np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)
        return ReadBatch(pos, flag, mapq, xs, lq, mtid, mpos, cig_off, cigar.astype(np.uint32), seq_off, seq4)

    def slice(self, lo, hi):
        """Records [lo, hi) as an independent batch (offsets rebased)."""
        c0, c1 = int(self.cig_off[lo]), int(self.cig_off[hi])
        s0, s1 = int(self.seq_off[lo]), int(self.seq_off[hi])
        return ReadBatch(
            self.pos[lo:hi].copy(), self.flag[lo:hi].copy(), self.mapq[lo:hi].copy(), self.xs[lo:hi].copy(),
            self.l_qseq[lo:hi].copy(), self.mtid[lo:hi].copy(), self.mpos[lo:hi].copy(),
            (self.cig_off[lo:hi + 1] - np.uint32(c0)).astype(np.uint32), self.cigar[c0:c1].copy(),
            (self.seq_off[lo:hi + 1] - np.uint32(s0)).astype(np.uint32), self.seq4[4 * s0:4 * s1].copy(),
            None if self.name_hash is None else self.name_hash[lo:hi].copy(),
        )
