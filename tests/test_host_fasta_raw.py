"""Host side of pjb_upload_contig_fasta (no GPU needed): GenomeMapper::rawSpan / readRaw hand over exactly the bytes of a
record's sequence lines, and the rule the device applies to them -- base i is byte (i / LINEBASES) * LINEWIDTH +
i % LINEBASES; every base a graphic character, every terminator byte not one -- gives GenomeMapper::fetchContig's bases
for records laid out as their .fai line says and refuses the others (restated here in numpy: what k0_fasta computes).
The C++ side runs under AddressSanitizer + UndefinedBehaviorSanitizer (PJB_TEST_SANITIZE=0: against the built library)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def delinearize(raw, n, line_bases, line_width):
    """numpy restatement of k0_fasta (portcullis_amd/csrc/pjb_kernels.hip.h): (bases, well_formed)."""
    raw = np.frombuffer(raw, dtype=np.uint8)
    i = np.arange(n, dtype=np.int64)
    src = i // line_bases * line_width + i % line_bases
    inside = src < len(raw)
    bases = np.where(inside, raw[np.minimum(src, max(len(raw) - 1, 0))] if len(raw) else 0, 0).astype(np.uint8)
    ok = bool(inside.all()) and bool(((bases > 32) & (bases < 127)).all())
    # terminators: after every full line that is followed by another base
    ends = i[(i % line_bases == line_bases - 1) & (i + 1 < n)]
    for k in range(line_width - line_bases):
        t = ends // line_bases * line_width + line_bases + k
        tb = np.where(t < len(raw), raw[np.minimum(t, max(len(raw) - 1, 0))] if len(raw) else 0, ord("?"))
        ok = ok and not bool(((tb > 32) & (tb < 127)).any())
    return bases.tobytes(), ok


def build(tmp_path):
    if os.environ.get("PJB_TEST_SANITIZE", "1") != "0":
        # GenomeMapper has no dependency on the device library: the driver and genome_mapper.cc build on their own, under
        # AddressSanitizer + UndefinedBehaviorSanitizer (an out-of-bounds pread destination or a bad offset computation
        # aborts the driver)
        host = os.path.join(ROOT, "portcullis_amd", "host")
        exe = str(tmp_path / "fasta_raw_span_asan")
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                               "-fno-omit-frame-pointer", f"-I{host}/include", f"-I{ROOT}/include", "-o", exe,
                               os.path.join(ROOT, "tests", "cpp", "fasta_raw_span.cc"),
                               os.path.join(host, "src", "genome_mapper.cc"), "-lpthread"])
        return exe
    host = os.path.join(ROOT, "portcullis_amd", "host")
    csrc = os.path.join(ROOT, "portcullis_amd", "csrc")
    if not os.path.exists(os.path.join(host, "libportcullis_host.so")):
        pytest.skip("host library not built")
    exe = str(tmp_path / "fasta_raw_span")
    subprocess.check_call(["g++", "-O1", "-std=c++17", f"-I{host}/include", f"-I{ROOT}/include", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "fasta_raw_span.cc"), f"-L{host}", "-lportcullis_host",
                           f"-L{csrc}", "-lportcullis_amd", f"-Wl,-rpath,{host}", f"-Wl,-rpath,{csrc}"])
    return exe


def test_fasta_raw_spans(tmp_path):
    exe = build(tmp_path)
    rng = np.random.default_rng(7)
    alphabet = np.frombuffer(b"ACGTacgtNnRYKM", dtype=np.uint8)
    recs = {  # name -> (bases, line width in bases, terminator)
        "short_last": (rng.choice(alphabet, 1234).tobytes().decode(), 60, "\n"),
        "crlf": (rng.choice(alphabet, 700).tobytes().decode(), 70, "\r\n"),
        "whole_lines": (rng.choice(alphabet, 50 * 9).tobytes().decode(), 50, "\n"),
        "one_line": (rng.choice(alphabet, 333).tobytes().decode(), 333, "\n"),
        "one_base": ("G", 80, "\n"),
    }
    fa = tmp_path / "g.fa"
    with open(fa, "w", newline="") as f:
        for name, (seq, w, term) in recs.items():
            f.write(f">{name} some description{term}")
            for k in range(0, len(seq), w):
                f.write(seq[k:k + w] + term)
    out = tmp_path / "out"
    out.mkdir()
    lines = subprocess.check_output([exe, str(fa), str(out)] + list(recs), text=True).split("\n")
    whole = open(fa, "rb").read()
    for line in lines:
        if not line:
            continue
        name, off, nbytes, lb, lw, length = line.split()
        off, nbytes, lb, lw, length = int(off), int(nbytes), int(lb), int(lw), int(length)
        seq, w, term = recs[name]
        assert (lb, lw, length) == (min(w, len(seq)) if len(seq) < w else w, (min(w, len(seq)) if len(seq) < w else w) + len(term), len(seq))
        raw = open(out / f"{name}.raw", "rb").read()
        assert raw == whole[off:off + nbytes]                       # readRaw: exactly the file's bytes
        assert raw[:1] == seq[:1].encode() and raw[-1:] == seq[-1:].encode()  # from the first base to the last
        got = open(out / f"{name}.seq", "rb").read()
        assert got == seq.encode()                                  # fetchContig (the reference's loader)
        bases, ok = delinearize(raw, length, lb, lw)
        assert ok and bases == got                                  # the device's rule on the same bytes


def test_fasta_records_not_laid_out_as_indexed(tmp_path):
    """A record whose lines are not all of the first line's width: the index describes the first line only, the rule must
    refuse the record (fetchContig's character filter still reads it correctly)."""
    exe = build(tmp_path)
    seq = "ACGT" * 100
    fa = tmp_path / "r.fa"
    with open(fa, "w") as f:
        f.write(">ragged\n" + seq[:60] + "\n" + seq[60:130] + "\n" + seq[130:] + "\n")   # second line is 70 wide
        f.write(">blank\n" + seq[:60] + "\n" + seq[60:90] + " " + seq[91:120] + "\n" + seq[120:] + "\n")
    out = tmp_path / "out"
    out.mkdir()
    lines = [l for l in subprocess.check_output([exe, str(fa), str(out), "ragged", "blank"], text=True).split("\n") if l]
    seen = set()
    for line in lines:
        f = line.split()
        seen.add(f[0])
        if f[1] in ("short", "none"):  # the span the index implies runs past the end of the file: no raw span, the caller parses
            assert f[0] == "blank"
            continue
        name, off, nbytes, lb, lw, length = f
        raw = open(out / f"{name}.raw", "rb").read()
        _, ok = delinearize(raw, int(length), int(lb), int(lw))
        assert not ok
    assert seen == {"ragged", "blank"}
    assert open(out / "ragged.seq", "rb").read() == seq.encode()


def test_fasta_odd_line_ends_are_left_to_the_filter(tmp_path):
    """Line ends longer than a few bytes: no raw span (the device's terminator check is bounded), fetchContig still reads
    the record."""
    exe = build(tmp_path)
    seq = "GATTACA" * 40
    fa = tmp_path / "o.fa"
    term = "\r" * 9 + "\n"
    with open(fa, "w", newline="") as f:
        f.write(">odd" + "\n" + term.join(seq[k:k + 70] for k in range(0, len(seq), 70)) + term)
    out = tmp_path / "out"
    out.mkdir()
    lines = [l for l in subprocess.check_output([exe, str(fa), str(out), "odd"], text=True).split("\n") if l]
    assert lines == ["odd none"]
