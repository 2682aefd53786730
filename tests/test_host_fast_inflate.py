"""BamReader's own DEFLATE decoder (host/src/fast_inflate.cc) against zlib, under AddressSanitizer + UBSan (no GPU): zlib's
output at every level, strategy and memLevel (stored, fixed and dynamic blocks, several blocks per stream), exact-size input
and output buffers with guard bytes, wrong output sizes, and twelve damaged copies of every stream -- whatever the decoder
accepts, zlib must accept with the same bytes; undamaged streams must be accepted.  tests/cpp/fast_inflate_check.cc."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    host = os.path.join(ROOT, "portcullis_amd", "host")
    out = str(tmp_path_factory.mktemp("fi") / "fast_inflate_check")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", f"-I{host}/include",
                           "-o", out, os.path.join(ROOT, "tests", "cpp", "fast_inflate_check.cc"), os.path.join(host, "src", "fast_inflate.cc"), "-lz"])
    return out


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fast_inflate_against_zlib(exe, seed):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe, str(seed), "250"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0 and p.stdout.startswith("ok:"), p.stdout[-2000:] + p.stderr[-3000:]
