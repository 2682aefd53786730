"""BGZF deflate on the device (pjb_deflate_bgzf; the writing side of the BAM files the stages emit, deps/htslib-1.3/bgzf.c:216-262):
every member must be a valid gzip member with the BC field that inflates -- with zlib and with this library's own inflater --
to exactly the input bytes."""
import gzip
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from portcullis_amd import ffi
    assert ffi.device_count() >= 1
    with ffi.Context(0, "UNKNOWN") as c:
        yield c


def _members(blob, sizes):
    out, o = [], 0
    for s in sizes:
        m = blob[o:o + int(s)]
        assert m[:4] == b"\x1f\x8b\x08\x04" and m[10:16] == b"\x06\x00BC\x02\x00"
        (bsize,) = struct.unpack_from("<H", m, 16)
        assert bsize + 1 == len(m)
        out.append(m)
        o += int(s)
    assert o == len(blob)
    return out


def _check(ctx, data, block_bytes=0xff00):
    blob, sizes = ctx.deflate_bgzf(data, block_bytes)
    n_blocks = (len(data) + block_bytes - 1) // block_bytes
    assert len(sizes) == n_blocks
    members = _members(blob, sizes)
    back = b""
    for k, m in enumerate(members):
        part = data[k * block_bytes:(k + 1) * block_bytes]
        d = zlib.decompressobj(-15)
        got = d.decompress(m[18:-8]) + d.flush()
        assert d.eof and got == part, (k, len(got), len(part))
        crc, isize = struct.unpack_from("<II", m, len(m) - 8)
        assert crc == zlib.crc32(part) and isize == len(part), k
        back += got
    assert back == data
    if data:
        assert gzip.decompress(blob) == data
        assert ctx.inflate_bgzf(blob) == data      # this library's own inflater
    return blob, sizes


def _bam_like(rng, n_records):
    """records with a fixed-layout core, names that count up, repetitive qualities: what a BAM block looks like"""
    out = []
    for k in range(n_records):
        l_seq = int(rng.integers(90, 151))
        name = f"read{k // 2:08d}".encode() + b"\0"
        core = struct.pack("<iiBBHHHiiii", int(rng.integers(0, 3)), 100000 + 37 * k, len(name), 60, 4681, 1, 99, l_seq, 0, 100200 + 37 * k, 300)
        cigar = struct.pack("<I", l_seq << 4)
        seq = rng.integers(0, 256, (l_seq + 1) // 2, dtype=np.uint8).tobytes()
        qual = bytes(rng.choice(np.frombuffer(b"FFFFFFFFFF:F,", dtype=np.uint8), l_seq))
        body = core + name + cigar + seq + qual + b"NHC\x01XSA+"
        out.append(struct.pack("<i", len(body)) + body)
    return b"".join(out)


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 63, 64, 65, 127, 1000, 0xff00 - 1, 0xff00, 0xff00 + 1, 3 * 0xff00 + 17])
def test_sizes_random_and_text(ctx, n):
    rng = np.random.default_rng(n)
    _check(ctx, rng.integers(0, 256, n, dtype=np.uint8).tobytes())                 # incompressible: stored blocks
    text = (b"the quick brown fox jumps over the lazy dog; " * (n // 40 + 1))[:n]
    _check(ctx, text)
    _check(ctx, bytes(n))                                                          # one long run
    _check(ctx, bytes(rng.integers(0, 3, n, dtype=np.uint8)))                      # three letters: short codes, many matches


def test_small_blocks_and_every_length_code(ctx):
    rng = np.random.default_rng(7)
    # matches of every length 4 .. 258 at assorted distances
    parts = []
    unit = rng.integers(0, 256, 600, dtype=np.uint8).tobytes()
    for l in range(4, 259):
        parts.append(rng.integers(0, 256, 5, dtype=np.uint8).tobytes() + unit[:l])
    data = unit + b"".join(parts)
    _check(ctx, data)
    _check(ctx, data, block_bytes=1024)
    _check(ctx, data, block_bytes=4)


def test_skewed_counts_need_the_length_limiter(ctx):
    """Fibonacci-like symbol counts make an unlimited Huffman code deeper than 15 bits"""
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    rng = np.random.default_rng(3)
    syms = np.concatenate([np.full(f, k, dtype=np.uint8) for k, f in enumerate(fib)])
    rng.shuffle(syms)
    data = syms.tobytes()[:0xff00]
    _check(ctx, data)


def test_bam_like_ratio(ctx):
    rng = np.random.default_rng(11)
    data = _bam_like(rng, 12000)
    blob, sizes = _check(ctx, data)
    ref = sum(len(zlib.compress(data[o:o + 0xff00], 6)) for o in range(0, len(data), 0xff00))
    ratio = len(blob) / ref
    print("device / zlib -6:", round(ratio, 3), len(data), len(blob), ref)
    assert ratio < 1.25


@pytest.mark.parametrize("seed", range(24))
def test_structured_fuzz(ctx, seed):
    """mixtures of what a compressor meets: random stretches, copies from near and far (beyond the 32 K window too), runs,
    small alphabets, text; random block sizes"""
    rng = np.random.default_rng(1000 + seed)
    parts, total = [], 0
    want = int(rng.integers(1, 4 * 0xff00))
    while total < want:
        kind = int(rng.integers(0, 6))
        ln = int(rng.integers(1, 3000))
        if kind == 0:
            b = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        elif kind == 1 and total > 0:   # a copy from somewhere before
            whole = b"".join(parts)
            back = int(rng.integers(1, min(len(whole), 70000) + 1))
            b = (whole[-back:] * (ln // back + 1))[:ln]
        elif kind == 2:
            b = bytes([int(rng.integers(0, 256))]) * ln
        elif kind == 3:
            b = bytes(rng.integers(0, 4, ln, dtype=np.uint8) + 65)
        elif kind == 4:
            b = (b"chr%d\t%d\tread_%d\t" % (int(rng.integers(1, 23)), int(rng.integers(0, 10**8)), int(rng.integers(0, 10**6)))) * (ln // 20 + 1)
            b = b[:ln]
        else:
            b = bytes(rng.integers(0, 256, 16, dtype=np.uint8)) * (ln // 16 + 1)
            b = b[:ln]
        parts.append(b)
        total += len(b)
    data = b"".join(parts)[:want]
    block = int(rng.choice([0xff00, 0xff00, 4096, 32768, 1000 * 4]))
    _check(ctx, data, block)


def test_bad_arguments_are_refused(ctx):
    import ctypes as C
    from portcullis_amd import ffi
    data = np.frombuffer(b"abcd" * 5000, dtype=np.uint8)
    out = np.empty(1 << 17, dtype=np.uint8)
    n = C.c_int64()
    sizes = np.zeros(8, dtype=np.uint32)
    L = ctx._L
    def call(block, cap, nbytes=len(data)):
        return L.pjb_deflate_bgzf(ctx._h, data.ctypes.data_as(C.c_void_p), nbytes, block, out.ctypes.data_as(C.c_void_p), cap, C.byref(n),
                                  sizes.ctypes.data_as(C.c_void_p))
    assert call(0xff00 + 4, 1 << 17) != 0          # larger than a BGZF block may be
    assert call(1002, 1 << 17) != 0                # not a multiple of 4
    assert call(0, 1 << 17) != 0
    assert call(0xff00, 64) != 0                   # the output does not fit
    assert b"more than" in L.pjb_last_error(ctx._h)
    assert call(0xff00, 1 << 17) == 0 and n.value > 0
    assert call(0xff00, 1 << 17, 0) == 0 and n.value == 0   # nothing in, nothing out
