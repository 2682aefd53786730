"""The C-ABI library loads (no GPU needed) and exports every function include/portcullis_amd.h
declares; struct sizes seen from Python match the header."""
import ctypes
import os

import pytest
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "portcullis_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pjb_[a-z_]+)\s*\(", hdr))
    assert {"pjb_create", "pjb_submit_batch", "pjb_finish_contig", "pjb_collect"} <= declared
    from portcullis_amd import ffi

    lib = ffi.load()
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert set(ffi.EXPORTS) == declared


def test_struct_sizes():
    from portcullis_amd import ffi

    assert ffi.ROW_DTYPE.itemsize == 200
    assert ctypes.sizeof(ffi.PjbBatch) == 8 + 14 * 8  # (ABI 4: + seq2, seq_exc)
    assert ffi.EXTRA_DTYPE.itemsize == 24
    assert ctypes.sizeof(ffi.PjbRegionResult) == 56
    assert ctypes.sizeof(ffi.PjbConfig) == 20


def test_struct_sizes_match_the_header_compiled(tmp_path):
    """sizeof of every struct the stub mirrors, as gcc sees it in include/portcullis_amd.h (pjb_timing grew in ABI 3)."""
    import subprocess
    from portcullis_amd import ffi

    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "portcullis_amd.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %d\\n", sizeof(pjb_timing), '
                   'sizeof(pjb_batch), sizeof(pjb_region_result), sizeof(pjb_config), sizeof(pjb_junction_row), PJB_ABI_VERSION); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    timing, batch, region, config, row, abi = (int(x) for x in out)
    assert timing == ctypes.sizeof(ffi.PjbTiming)
    assert batch == ctypes.sizeof(ffi.PjbBatch) and region == ctypes.sizeof(ffi.PjbRegionResult) and config == ctypes.sizeof(ffi.PjbConfig)
    assert row == ffi.ROW_DTYPE.itemsize and abi == ffi.ABI_VERSION


def test_no_device_is_a_loud_error():
    """Without a GPU pjb_create must fail with PJB_ERR_NO_DEVICE -- there is no CPU fallback."""
    from portcullis_amd import ffi

    if ffi.device_count() > 0:
        return
    try:
        ffi.Context(0)
    except ffi.PjbError as e:
        assert e.code == -18 and "no CPU fallback" in str(e)
    else:
        raise AssertionError("pjb_create succeeded without a device")


def test_host_library_builds_and_links():
    host = os.path.join(ROOT, "portcullis_amd", "host")
    assert os.path.exists(os.path.join(host, "libportcullis_host.so"))
    ctypes.CDLL(os.path.join(host, "libportcullis_host.so"))


def test_c_caller_of_plan_groups_and_merge_rows(tmp_path):
    """A plain C program (gcc, no HIP headers) links the library and calls the two host-arithmetic entry points of ABI 4 -- the chain plan and
    the receive side of the multi-GPU merge -- without a context or a device: tests/cpp/merge_plan_check.c."""
    import subprocess

    lib_dir = os.path.join(ROOT, "portcullis_amd", "csrc")
    if not os.path.exists(os.path.join(lib_dir, "libportcullis_amd.so")):
        pytest.skip("library not built")
    exe = tmp_path / "merge_plan_check"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "merge_plan_check.c"),
                    "-L", lib_dir, "-lportcullis_amd", f"-Wl,-rpath,{lib_dir}"], check=True)
    p = subprocess.run([str(exe)], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout.strip() == "ok", p.stdout + p.stderr
