"""The C-ABI library loads (no GPU needed) and exports every function include/portcullis_amd.h
declares; struct sizes seen from Python match the header."""
import ctypes
import os
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
    assert ctypes.sizeof(ffi.PjbBatch) == 8 + 12 * 8
    assert ffi.EXTRA_DTYPE.itemsize == 24
    assert ctypes.sizeof(ffi.PjbRegionResult) == 56
    assert ctypes.sizeof(ffi.PjbConfig) == 20


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
