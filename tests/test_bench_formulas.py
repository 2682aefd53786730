"""bench.py's byte formulas: every kernel the chain launches (pjb_api.hip) has one, so that `roofline.step_alg_bytes` prices the whole step."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def chain_kernel_names():
    src = open(os.path.join(ROOT, "portcullis_amd", "csrc", "pjb_api.hip")).read()
    a = src.index("static int queue_contig(")
    b = src.index("static void unqueue_followers(")
    body = src[a:b]
    names = set(re.findall(r'LAUNCH(?:_LDS)?\(c, "([a-z0-9_]+)"', body))
    for scan in re.findall(r'run_scan\(c, "([a-z0-9_]+)"', body):
        names |= {scan + "_reduce", scan + "_apply", scan + "_tiles"}
    return {n for n in names if not n.startswith(("kx_", "k0_"))}


def test_every_chain_kernel_has_a_byte_formula():
    import bench

    names = chain_kernel_names()
    assert {"k1_count", "k1_emit", "k1_generic", "k4b_generic", "k4_pairs", "k6_rows_out"} <= names
    missing = [n for n in sorted(names)
               if bench.algorithmic_bytes(n, 1e6, 3e6, 3e5, 1e6, 4e5, 1e4, 150, 1e4, 5e3, 1e5, 2e4, 5e4, True, 3e8) is None
               and not n.endswith("_tiles")]  # (the scan's middle kernel exists for inputs of more than 4096 tiles only)
    assert not missing, missing


def test_survey_formula_is_the_one_of_section_8d():
    import bench

    # N (18 + 4 c) + P (24 + 128 + 16 + 36 + 1.5 A + 4 c_s + 16 + 32) + 264 J with A = L
    got = bench.survey_bytes(10, 30, 4, 16, 5, 2, 100)
    assert got == 10 * (18 + 4 * 3.0) + 5 * (24 + 128 + 16 + 36 + 150.0 + 4 * 4.0 + 16 + 32) + 2 * 264
