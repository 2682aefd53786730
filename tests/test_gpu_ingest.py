"""Device-side BGZF inflate against zlib: every DEFLATE block type, long and short
match distances, many blocks per call, the reference's own BAM fixture, and corrupt input (must fail
with PJB_ERR_BGZF, never hang)."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from util_bam import BGZF_EOF, _bgzf_block

pytestmark = pytest.mark.gpu


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


def test_many_blocks_and_eof_markers(ctx):
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
