"""Reads of device memory that nobody wrote.  PJB_POISON=1 (a test hook of the library) fills every new device buffer with a
pattern instead of leaving it as the allocator hands it out -- usually zeroed, which hides such reads until a used page comes
along (round 4: `junc` died of a Memory access fault in one run of thirty; with the pattern the same run failed every time).
The hot path through the C ABI and the whole program run once more under the pattern; both in child processes, because the
hook is read when the library is loaded."""
import os
import subprocess
import sys

import pytest

from junctools_cases import fuzz_contigs
from util_bam import make_prep_dir

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "portcullis_amd", "host", "portcullis_amd")

def test_hot_path_under_poison():
    env = dict(os.environ, PJB_POISON="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "poison_child.py")], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0 and "poison ok" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


@pytest.mark.parametrize("ingest", ["host", "device"])
def test_program_under_poison(tmp_path, ingest):
    contigs = fuzz_contigs(seeds=(31, 32, 33, 34, 35), n_reads=2500) + fuzz_contigs(seeds=(36,), n_reads=200)
    contigs = [(f"chr{k + 1}", g, [dict(r, tid=k, mtid=(k if r.get("mtid", -1) >= 0 else -1)) for r in rr]) for k, (_, g, rr) in enumerate(contigs)]
    reads = [r for _, _, rr in contigs for r in rr]
    prep = make_prep_dir(str(tmp_path / "prep"), [(n, len(g)) for n, g, _ in contigs], [(n, g) for n, g, _ in contigs], reads)
    outs = []
    for poison in ("0", "1"):
        out = str(tmp_path / f"out{poison}" / "pc")
        p = subprocess.run([EXE, "junc", "-o", out, "--orientation", "FR", "-t", "4", "--ingest", ingest, prep], capture_output=True, text=True,
                           timeout=600, env=dict(os.environ, PJB_POISON=poison))
        assert p.returncode == 0, f"PJB_POISON={poison}: " + p.stdout[-1000:] + p.stderr[-2000:]
        outs.append(open(out + ".junctions.tab", "rb").read())
    assert outs[0] == outs[1] and outs[0].count(b"\n") > 10
