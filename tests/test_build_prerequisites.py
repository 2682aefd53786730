"""A stale-binary guard (round 5's verdict: pjb_deflate.hip.h was not a prerequisite of the library): every source the HIP library
and the host layer are built from must make `make -q` report "out of date" when it is newer than the binary.  No GPU, nothing is built."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def out_of_date_after_touch(make_dir, target, source):
    """`make -q` exit status with `source` stamped newer than `target` (times are put back afterwards)."""
    st_t, st_s = os.stat(target), os.stat(source)
    try:
        os.utime(source, ns=(st_t.st_atime_ns, st_t.st_mtime_ns + 5_000_000_000))
        return subprocess.run(["make", "-q", "-C", make_dir], capture_output=True).returncode
    finally:
        os.utime(source, ns=(st_s.st_atime_ns, st_s.st_mtime_ns))


def test_every_kernel_header_rebuilds_the_library():
    d = os.path.join(ROOT, "portcullis_amd", "csrc")
    lib = os.path.join(d, "libportcullis_amd.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    sources = sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.hip.h"))) + [os.path.join(ROOT, "include", "portcullis_amd.h")]
    assert len(sources) >= 6
    for src in sources:
        assert out_of_date_after_touch(d, lib, src) == 1, f"editing {os.path.relpath(src, ROOT)} would not rebuild libportcullis_amd.so"


def test_the_abi_header_rebuilds_the_host_layer():
    d = os.path.join(ROOT, "portcullis_amd", "host")
    lib = os.path.join(d, "libportcullis_host.so")
    if not os.path.exists(lib):
        pytest.skip("host layer not built")
    hdr = os.path.join(ROOT, "include", "portcullis_amd.h")
    assert out_of_date_after_touch(d, lib, hdr) == 1, "editing include/portcullis_amd.h would not rebuild the host layer"
