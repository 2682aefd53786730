"""Device-side BGZF inflate against zlib: every DEFLATE block type, long and short
match distances, many blocks per call, the reference's own BAM fixture, and corrupt input (must fail
with PJB_ERR_BGZF, never hang)."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from fuzzgen import make_reads
from util_bam import BGZF_EOF, _bgzf_block, write_bam

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ffi_mod():
    from portcullis_amd import ffi
    assert ffi.device_count() >= 1
    return ffi


@pytest.fixture(scope="module")
def ctx():
    from portcullis_amd import ffi
    assert ffi.device_count() >= 1
    with ffi.Context(0, "UNKNOWN") as c:
        yield c


def bgzf(payload, level=1, strategy=zlib.Z_DEFAULT_STRATEGY, block=0xFF00):
    out = bytearray()
    for u in range(0, len(payload), block):
        chunk = payload[u:u + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        cdata = co.compress(chunk) + co.flush()
        hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(cdata) + 25)
        out += hdr + cdata + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return bytes(out)


def payloads():
    rng = np.random.default_rng(11)
    text = b"".join(b"read%07d\tchrIII\t%d\t60\t50M200N50M\t=\tACGTTGCA\n" % (i, 1000 + 7 * i) for i in range(40000))
    bamlike = rng.integers(0, 16, 300000, dtype=np.uint8).tobytes() + text[:200000]
    return {
        "random": rng.integers(0, 256, 200000, dtype=np.uint8).tobytes(),          # incompressible: stored blocks
        "text": text,
        "runs": b"".join(bytes([65 + (i % 7)]) * (1 + (i * 37) % 300) for i in range(3000)),  # distance-1 matches
        "periods": b"".join((b"ACGTTGCAAGT"[:1 + i % 9]) * (5 + i % 60) for i in range(4000)),  # distances 1..9
        "bamlike": bamlike,
        "tiny": b"x",
        "zeros": bytes(70000),
    }


@pytest.mark.parametrize("name", list(payloads()))
@pytest.mark.parametrize("mode", ["l1", "l6", "l9", "fixed", "stored", "huffman"])
def test_inflate_matches_zlib(ctx, name, mode):
    data = payloads()[name]
    level, strat = {"l1": (1, zlib.Z_DEFAULT_STRATEGY), "l6": (6, zlib.Z_DEFAULT_STRATEGY), "l9": (9, zlib.Z_DEFAULT_STRATEGY),
                    "fixed": (6, zlib.Z_FIXED), "stored": (0, zlib.Z_DEFAULT_STRATEGY), "huffman": (6, zlib.Z_HUFFMAN_ONLY)}[mode]
    comp = bgzf(data, level, strat)
    assert gzip.decompress(comp) == data
    assert ctx.inflate_bgzf(comp) == data


def test_full_64k_blocks(ctx):
    """Blocks of exactly 65536 inflated bytes (the BGZF maximum; match destinations reach offset 65536)."""
    rng = np.random.default_rng(9)
    data = bytes(70000) + (rng.integers(0, 3, 200000, dtype=np.uint8) + 65).tobytes() + b"ACGT" * 40000
    for level in (1, 6):
        comp = bgzf(data, level, block=65536)
        assert ctx.inflate_bgzf(comp) == data


@pytest.mark.parametrize("per_launch", [None, "128"])
def test_many_blocks_and_eof_markers(ctx, monkeypatch, per_launch):
    if per_launch:
        monkeypatch.setenv("PJB_INF_BLOCKS_PER_LAUNCH", per_launch)  # the scratch area is reused launch after launch
    rng = np.random.default_rng(5)
    parts, comp = [], bytearray()
    for i in range(700):  # more than one 64-lane workgroup per launch, ragged sizes, EOF markers in between
        n = int(rng.integers(1, 60000))
        chunk = (rng.integers(0, 4, n, dtype=np.uint8) + 65).tobytes()
        parts.append(chunk)
        comp += bgzf(chunk, level=1 + i % 9)
        if i % 50 == 0:
            comp += BGZF_EOF
    comp += BGZF_EOF
    assert ctx.inflate_bgzf(bytes(comp)) == b"".join(parts)
    assert ctx.inflate_bgzf(BGZF_EOF) == b""
    assert ctx.inflate_bgzf(b"") == b""


def test_reference_fixture_bam(ctx, golden_dir):
    raw = open(os.path.join(golden_dir, "clipped3.bam"), "rb").read()
    assert ctx.inflate_bgzf(raw) == gzip.decompress(raw)


def test_util_bam_block_writer(ctx):
    data = os.urandom(1000) + b"A" * 5000
    assert ctx.inflate_bgzf(_bgzf_block(data, level=6) + BGZF_EOF) == data


def test_corrupt_input_fails_cleanly(ctx):
    from portcullis_amd import ffi
    data = payloads()["text"][:300000]
    good = bgzf(data, 6)
    rng = np.random.default_rng(3)
    n_err = 0
    for trial in range(40):
        bad = bytearray(good)
        for _ in range(1 + trial % 5):
            pos = int(rng.integers(18, len(bad) - 8))
            bad[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            out = ctx.inflate_bgzf(bytes(bad))
            assert len(out) == len(data)  # a flipped bit may still decode; sizes are pinned by ISIZE
        except ffi.PjbError as e:
            n_err += 1
            assert e.code in (-22,), e
    assert n_err > 0
    for cut in (5, 17, 30, len(good) - 3):  # truncated streams
        with pytest.raises(ffi.PjbError):
            ctx.inflate_bgzf(good[:cut])
    with pytest.raises(ffi.PjbError):
        ctx.inflate_bgzf(b"\x1f\x8b\x08\x00" + bytes(40))  # gzip without the BGZF extra field
    assert ctx.inflate_bgzf(good) == data  # the context still works afterwards


class _Bits:
    """LSB-first bit packing of a DEFLATE stream; Huffman codes go in MSB-first."""

    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, value, nbits):
        self.acc |= value << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 0xff)
            self.acc >>= 8
            self.n -= 8

    def code(self, code, nbits):
        for k in range(nbits - 1, -1, -1):
            self.put((code >> k) & 1, 1)

    def bytes(self):
        return bytes(self.out) + (bytes([self.acc & 0xff]) if self.n else b"")


def _raw_block(payload, isize):
    """A BGZF block around a hand-made DEFLATE payload (CRC is not checked by either decoder here)."""
    hdr = bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0])
    total = len(hdr) + 2 + len(payload) + 8
    return hdr + struct.pack("<H", total - 1) + payload + struct.pack("<II", 0, isize)


def _endless_literals_header():
    """Dynamic block whose literal/length set is { literal 0: '0', end-of-block: '1' } (complete): a zero-filled
    stream behind it decodes as literal 0 for ever, one bit each."""
    b = _Bits()
    b.put(1, 1); b.put(2, 2)                    # BFINAL, dynamic
    b.put(0, 5); b.put(0, 5); b.put(14, 4)      # 257 lit/len codes, 1 distance code, 18 code-length codes
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    cl = {18: 2, 0: 2, 1: 1}
    for sym in order[:18]:
        b.put(cl.get(sym, 0), 3)
    # code-length alphabet codes: '1' -> 0, '0' -> 10, '18' -> 11
    b.code(0, 1)                                # literal 0: length 1
    b.code(3, 2); b.put(138 - 11, 7)            # 138 zeros
    b.code(3, 2); b.put(117 - 11, 7)            # 117 zeros (literals 1..255)
    b.code(0, 1)                                # end-of-block: length 1
    b.code(2, 2)                                # the one distance code: length 0
    return b


def test_streams_that_run_past_their_block(ctx):
    """A block whose symbols never reach an end-of-block code inside its payload must fail cleanly (PJB_ERR_BGZF) after
    reading at most a few hundred bytes past the payload -- never the tens of kilobytes its ISIZE would allow -- and an
    incomplete code set is refused as zlib refuses it."""
    import zlib

    from portcullis_amd import ffi
    b = _endless_literals_header()
    payload = b.bytes() + bytes(24)              # zeros: literal 0, literal 0, ...
    good_tail = bgzf(b"tail" * 1000, 6)
    for comp in (_raw_block(payload, 65536),                                  # last block: only the zero padding follows
                 _raw_block(payload, 65536) + good_tail,                      # another block's bytes follow
                 good_tail + _raw_block(payload, 60000) + BGZF_EOF):
        with pytest.raises(ffi.PjbError) as e:
            ctx.inflate_bgzf(comp)
        assert e.value.code == -22
    # the same header with a proper end: 100 literal zeros then end-of-block -> decodes
    b2 = _endless_literals_header()
    for _ in range(100):
        b2.code(0, 1)
    b2.code(1, 1)
    ok = b2.bytes()
    assert zlib.decompressobj(-15).decompress(ok) == bytes(100)
    assert ctx.inflate_bgzf(_raw_block(ok, 100) + BGZF_EOF) == bytes(100)
    # incomplete literal/length set: literal 0 and end-of-block both 2 bits long, nothing else
    b3 = _Bits()
    b3.put(1, 1); b3.put(2, 2); b3.put(0, 5); b3.put(0, 5); b3.put(14, 4)
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    cl = {18: 2, 0: 2, 2: 1}
    for sym in order[:18]:
        b3.put(cl.get(sym, 0), 3)
    b3.code(0, 1); b3.code(3, 2); b3.put(127, 7); b3.code(3, 2); b3.put(106, 7); b3.code(0, 1); b3.code(2, 2)
    b3.code(0, 2); b3.code(1, 2)
    bad = b3.bytes() + bytes(8)
    with pytest.raises(zlib.error):
        zlib.decompressobj(-15).decompress(bad)
    with pytest.raises(ffi.PjbError):
        ctx.inflate_bgzf(_raw_block(bad, 1) + BGZF_EOF)
    assert ctx.inflate_bgzf(good_tail) == b"tail" * 1000   # the context still works


def test_bam_data_ending_inside_a_record_is_an_error(ffi_mod, orc):
    """pjb_submit_bam: the bytes of a target must not stop inside one of its records (a span cut short by a stale
    index or a truncated file would silently lose the tail)."""
    genome, reads = make_reads(8, n_reads=400)
    for k, r in enumerate(reads):
        r["tid"] = 0
        r["name"] = f"r{k}"
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "t.bam")
        write_bam(path, [("chr1", len(genome))], reads, write_index=False, block_size=2000)
        raw = open(path, "rb").read()
    blocks, o = [], 0
    while o < len(raw):
        bs = (raw[o + 16] | raw[o + 17] << 8) + 1
        blocks.append((o, bs))
        o += bs
    hdr = gzip.decompress(raw[: blocks[0][1]])
    (l_text,) = struct.unpack_from("<i", hdr, 4)
    first = 8 + l_text + 4 + (4 + len("chr1") + 1 + 4)
    with ffi_mod.Context(0, "UNKNOWN") as ctx:
        ctx.set_refs([len(genome)])
        ctx.upload_contig(0, genome.encode())
        raised = 0
        for k in (len(blocks) // 3, len(blocks) // 2, len(blocks) // 2 + 1):
            cut = blocks[k][0]                       # whole blocks, but the record chain is (almost surely) cut
            try:
                n = ctx.submit_bam(0, raw[:cut], first)
                assert n < len(reads)                # the cut fell on a record boundary
            except ffi_mod.PjbError as e:
                assert e.code == -22 and "ends inside" in str(e)
                raised += 1
            ctx.finish_contig(0)
            ctx.clear_rows()
        assert raised >= 1
        n = ctx.submit_bam(0, raw, first)            # the whole file is fine
        assert n == len(reads)
        ctx.finish_contig(0)


# ------------------------------------------------------------------ BAM records on the device
def bam_targets(path):
    """(refs, {tid: (file offset of the block holding the first record, offset inside that block)})."""
    raw = open(path, "rb").read()
    # block table
    blocks, o = [], 0
    while o < len(raw):
        bs = (raw[o + 16] | raw[o + 17] << 8) + 1
        isz = struct.unpack_from("<I", raw, o + bs - 4)[0]
        blocks.append((o, isz))
        o += bs
    ustart = np.cumsum([0] + [b[1] for b in blocks])
    data = gzip.decompress(raw)
    (l_text,) = struct.unpack_from("<i", data, 4)
    p = 8 + l_text
    (n_ref,) = struct.unpack_from("<i", data, p)
    p += 4
    refs = []
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from("<i", data, p)
        name = data[p + 4:p + 4 + l_name - 1].decode()
        (l_ref,) = struct.unpack_from("<i", data, p + 4 + l_name)
        refs.append((name, l_ref))
        p += 8 + l_name
    first = {}
    while p + 4 <= len(data):
        (bs,) = struct.unpack_from("<i", data, p)
        (tid,) = struct.unpack_from("<i", data, p + 4)
        if tid >= 0 and tid not in first:
            b = int(np.searchsorted(ustart, p, side="right") - 1)
            first[tid] = (blocks[b][0], p - int(ustart[b]))
        p += 4 + bs
    return raw, refs, first


def run_targets_from_bam(ctx, orc, path, genomes, orientation="UNKNOWN", piece_sizes=None):
    from parity import assert_rows_equal, region_equal
    from util_bam import read_bam, records_to_batch
    raw, refs, first = bam_targets(path)
    _, recs = read_bam(path)
    ctx.set_refs([l for _, l in refs])
    n_checked = 0
    for tid, (coff, uoff) in sorted(first.items()):
        mine = [r for r in recs if r["tid"] == tid]
        batch = records_to_batch(mine)
        orows, oreg = orc.find_juncs(tid, refs[tid][1], genomes[tid], batch, orientation)
        ctx.upload_contig(tid, genomes[tid])
        ctx.clear_rows()
        n = ctx.submit_bam(tid, raw[coff:], uoff) if piece_sizes is None else ctx.submit_bam_pieces(tid, raw[coff:], uoff, piece_sizes)
        assert n == len(mine)
        dreg = ctx.finish_contig(tid)
        drows = ctx.collect()
        region_equal(dreg, oreg)
        assert_rows_equal(drows, orows)
        n_checked += len(orows)
    return n_checked


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


@pytest.mark.parametrize("block_size", [0xFF00, 700, 4096])
def test_submit_bam_matches_oracle(orc, tmp_path, block_size):
    """Three targets of fuzz reads (every CIGAR op, aux tags of every type before / after XS, SEQ '*'),
    BGZF blocks that cut records anywhere: the device ingest must give the oracle's rows."""
    from fuzzgen import make_reads
    from portcullis_amd import ffi
    from util_bam import write_bam
    reads, refs, genomes = [], [], {}
    for tid in range(3):
        genome, rs = make_reads(40 + tid, n_reads=1500 + 900 * tid, paired=(tid == 1))
        refs.append((f"chr{tid + 1}", len(genome)))
        genomes[tid] = genome.encode() if isinstance(genome, str) else genome
        for k, r in enumerate(rs):
            r = dict(r)
            r["tid"] = tid
            if r.get("mtid", -1) >= 0:
                r["mtid"] = tid
            r["name"] = f"t{tid}r{k}" + "x" * (k % 23)
            other = b"NMC\x03" + b"MDZ" + b"10A5^AC6\x00" + b"ZBBs" + struct.pack("<i", 3) + b"\x01\x00\x02\x00\x03\x00"
            if k % 3 == 0:   # aux fields of several types after XS ...
                r["aux"] = other
            elif k % 3 == 1 and r.get("xs") is not None:   # ... or before it
                r["aux"] = other + b"XSA" + r["xs"].encode()
                r["xs"] = None
            reads.append(r)
    path = str(tmp_path / "a.bam")
    write_bam(path, refs, reads, block_size=block_size)
    with ffi.Context(0, "FR") as ctx:
        assert run_targets_from_bam(ctx, orc, path, genomes, "FR") > 50


@pytest.mark.parametrize("block_size,pieces", [(0xFF00, [100000]), (700, [1, 17, 5000, 3]), (4096, [4096]), (0xFF00, [7, 65536 + 300])])
def test_submit_bam_in_pieces(orc, tmp_path, block_size, pieces):
    """pjb_bam_begin / _piece / _end: the same bytes in pieces that cut BGZF headers, footers and payloads anywhere
    (pieces shorter than a header, a piece boundary on every block boundary, ...) give the same rows."""
    from fuzzgen import make_reads
    from portcullis_amd import ffi
    from util_bam import write_bam
    reads, refs, genomes = [], [], {}
    for tid in range(2):
        genome, rs = make_reads(50 + tid, n_reads=1200 + 700 * tid, paired=(tid == 1))
        refs.append((f"chr{tid + 1}", len(genome)))
        genomes[tid] = genome.encode() if isinstance(genome, str) else genome
        for k, r in enumerate(rs):
            r = dict(r)
            r["tid"] = tid
            if r.get("mtid", -1) >= 0:
                r["mtid"] = tid
            r["name"] = f"t{tid}r{k}"
            reads.append(r)
    path = str(tmp_path / "p.bam")
    write_bam(path, refs, reads, block_size=block_size)
    with ffi.Context(0, "FR") as ctx:
        assert run_targets_from_bam(ctx, orc, path, genomes, "FR", piece_sizes=pieces) > 30
        # protocol errors: a piece for a target that was not begun, more bytes than announced, an end before all bytes
        raw = open(path, "rb").read()
        with pytest.raises(ffi.PjbError):
            ctx.submit_bam_pieces(0, raw[:100], 0, [40])          # truncated block
        ctx._check(ctx._L.pjb_bam_begin(ctx._h, 1, 1000))
        import ctypes as C
        buf = np.frombuffer(raw[:2000], dtype=np.uint8).copy()
        assert ctx._L.pjb_bam_piece(ctx._h, 1, buf.ctypes.data_as(C.c_void_p), 2000, None) != 0   # more than announced (drops the staging)
        assert ctx._L.pjb_bam_piece(ctx._h, 1, buf.ctypes.data_as(C.c_void_p), 10, None) != 0     # ... so: not begun
        ctx._check(ctx._L.pjb_bam_begin(ctx._h, 1, 1000))
        assert ctx._L.pjb_bam_end(ctx._h, 1, 0, None) != 0          # 0 of 1000 bytes arrived
        assert ctx._L.pjb_bam_piece(ctx._h, 0, buf.ctypes.data_as(C.c_void_p), 10, None) != 0      # never begun (the failed run above was dropped)
        assert run_targets_from_bam(ctx, orc, path, genomes, "FR", piece_sizes=pieces) > 30       # and the context still works


def test_submit_bam_reference_fixture(orc, golden_dir):
    from portcullis_amd import ffi
    path = os.path.join(golden_dir, "clipped3.bam")
    raw, refs, first = bam_targets(path)
    rng = np.random.default_rng(4)
    genomes = {t: rng.choice(np.frombuffer(b"ACGT", np.uint8), size=refs[t][1]).tobytes() for t in first}
    with ffi.Context(0, "UNKNOWN") as ctx:
        assert run_targets_from_bam(ctx, orc, path, genomes) >= 1


def test_submit_bam_corrupt_records_fail_cleanly(orc, tmp_path):
    """Valid BGZF / DEFLATE around damaged BAM records: the chain verification (or a size check) must
    refuse the target, or -- if the damage stays inside fields the path does not read -- still
    return a consistent batch; never hang, never crash, and the context stays usable."""
    from fuzzgen import make_reads
    from portcullis_amd import ffi
    from util_bam import write_bam
    genome, rs = make_reads(77, n_reads=3000)
    refs = [("chr1", len(genome))]
    for k, r in enumerate(rs):
        r["tid"] = 0
        r["name"] = f"q{k}"
    path = str(tmp_path / "c.bam")
    write_bam(path, refs, rs)
    raw, _, first = bam_targets(path)
    coff, uoff = first[0]
    data = bytearray(gzip.decompress(raw[coff:]))
    rng = np.random.default_rng(8)
    outcomes = {"error": 0, "ok": 0}
    with ffi.Context(0, "UNKNOWN") as ctx:
        ctx.set_refs([len(genome)])
        ctx.upload_contig(0, genome.encode() if isinstance(genome, str) else genome)
        for trial in range(24):
            bad = bytearray(data)
            for _ in range(1 + trial % 4):
                p = int(rng.integers(uoff, len(bad) - 40))
                if trial % 3 == 0:
                    bad[p:p + 4] = struct.pack("<I", int(rng.integers(0, 1 << 31)))  # likely a block_size / length field
                else:
                    bad[p] = int(rng.integers(0, 256))
            try:
                n = ctx.submit_bam(0, bgzf(bytes(bad), 1), uoff)
                reg = ctx.finish_contig(0)
                assert reg["n_reads"] == n
                outcomes["ok"] += 1
            except ffi.PjbError as e:
                outcomes["error"] += 1
                try:
                    ctx.finish_contig(0)  # drop whatever was submitted
                except ffi.PjbError:
                    pass
            ctx.clear_rows()
        assert outcomes["error"] > 0
        # and the untouched bytes still work on the same context
        n = ctx.submit_bam(0, bgzf(bytes(data), 6), uoff)
        assert n == len(rs)
        ctx.finish_contig(0)


@pytest.mark.parametrize("name", ["bam1.bam", "bam2.bam", "sorted.bam", "unsorted.bam", "clipped3.bam"])
def test_reference_bam_fixtures_inflate(ctx, golden_dir, name):
    raw = open(os.path.join(golden_dir, name), "rb").read()
    assert ctx.inflate_bgzf(raw) == gzip.decompress(raw)


def test_reference_unspliced_fixtures(golden_dir):
    """The reference's E. coli fixtures (100 bwa alignments each, XS:i:<score> tags): `junc` refuses them like
    the reference does -- BamAlignment::init calls getXSStrand, bam_aux2A of a non-'A' XS is 0 and strandFromChar
    throws "Unknown strand" (bam_alignment.cc:226-231, bam_master.hpp:60-72); the unsorted one may also be
    refused for its order, whichever alignment comes first.  File bytes and SoA batches must agree."""
    from portcullis_amd import ffi
    from util_bam import read_bam, records_to_batch
    for name in ("sorted.bam", "unsorted.bam", "bam1.bam", "bam2.bam"):
        path = os.path.join(golden_dir, name)
        raw, refs, first = bam_targets(path)
        _, recs = read_bam(path)
        assert any(r["xs"] == "\x00" for r in recs)  # typed XS:i
        batch = records_to_batch(recs)
        codes = []
        with ffi.Context(0, "UNKNOWN") as ctx:
            ctx.set_refs([l for _, l in refs])
            coff, uoff = first[0]
            for via_bam in (True, False):
                ctx.clear_rows()
                if via_bam:
                    assert ctx.submit_bam(0, raw[coff:], uoff) == len(recs)
                else:
                    ctx.submit_batch(0, batch)
                with pytest.raises(ffi.PjbError) as ei:
                    ctx.finish_contig(0)
                codes.append((ei.value.code, str(ei.value)))
        assert codes[0] == codes[1], codes
        assert codes[0][0] in ((-1, -14) if name != "sorted.bam" else (-1,)), codes


@pytest.mark.parametrize("seg", [1, 2, 5])
def test_false_record_start_is_repaired(orc, tmp_path, monkeypatch, seg):
    """A guessed record start that is not a boundary (injected through the PJB_TEST_FALSE_START hook: the start of
    one 64 KB segment is moved by a byte) is caught by the chain walk and replaced by the boundary the verified walk
    reaches; the result is the oracle's."""
    from fuzzgen import make_reads
    from portcullis_amd import ffi
    from util_bam import write_bam
    genome, rs = make_reads(91, n_reads=6000, L=(60, 150))
    for k, r in enumerate(rs):
        r["tid"] = 0
        r["name"] = f"read{k:06d}"
        if r.get("mtid", -1) >= 0:
            r["mtid"] = 0
    path = str(tmp_path / "f.bam")
    write_bam(path, [("chr1", len(genome))], rs)
    monkeypatch.setenv("PJB_TEST_FALSE_START", str(seg))
    with ffi.Context(0, "UNKNOWN") as ctx:
        assert run_targets_from_bam(ctx, orc, path, {0: genome.encode()}) > 10
