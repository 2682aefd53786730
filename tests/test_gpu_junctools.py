"""The device path against the junctools vectors (see test_oracle_junctools.py): the same inputs as a prepared BAM + FASTA ->
`portcullis_amd junc` (HIP kernels behind the C ABI, C++ JunctionSystem writer) -> .junctions.tab, which must be, line for line,
what the reference's own TabJunction parsed and re-serialised."""
import os
import subprocess

import pytest

from junctools_cases import fuzz_contigs, micro_reads
from test_oracle_junctools import FIX, FIX2, check_bed_against_fixture, check_gff_against_fixture, check_tab_against_fixture
from util_bam import make_prep_dir

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "portcullis_amd", "host", "portcullis_amd")


def run_junc(prep, out, orientation, *opts):
    assert os.path.exists(EXE), f"{EXE} missing: run __graft_entry__.build()"
    p = subprocess.run([EXE, "junc", "-o", out, "--orientation", orientation, "-t", "2", *opts, prep], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    return open(out + ".junctions.tab").read()


@pytest.mark.parametrize("ingest", ["host", "device"])
def test_micro_fixtures_tab_is_what_junctools_parsed(tmp_path, ingest):
    name, genome, reads = micro_reads()
    prep = make_prep_dir(str(tmp_path / "prep"), [(name, len(genome))], [(name, genome)], reads)
    tab = run_junc(prep, str(tmp_path / "out" / "pc"), "FR", "--ingest", ingest, "--intron_gff")
    check_tab_against_fixture(tab, FIX["cases"]["micro_FR"])
    # the program's .bed and intron .gff3: line for line what the reference's BedJunction / GFFJunction parsed
    check_bed_against_fixture(open(str(tmp_path / "out" / "pc") + ".junctions.bed").read(), FIX2["cases"]["micro_FR"])
    check_gff_against_fixture(open(str(tmp_path / "out" / "pc") + ".junctions.intron.gff3").read(), FIX2["cases"]["micro_FR"])


@pytest.mark.parametrize("ingest", ["host", "device"])
def test_fuzz_targets_tab_is_what_junctools_parsed(tmp_path, ingest):
    contigs = fuzz_contigs()
    reads = [r for _, _, rr in contigs for r in rr]
    prep = make_prep_dir(str(tmp_path / "prep"), [(n, len(g)) for n, g, _ in contigs], [(n, g) for n, g, _ in contigs], reads)
    tab = run_junc(prep, str(tmp_path / "out" / "pc"), "FR", "--ingest", ingest, "--intron_gff")
    check_tab_against_fixture(tab, FIX["cases"]["fuzz3_FR"])
    check_bed_against_fixture(open(str(tmp_path / "out" / "pc") + ".junctions.bed").read(), FIX2["cases"]["fuzz3_FR"])
    check_gff_against_fixture(open(str(tmp_path / "out" / "pc") + ".junctions.intron.gff3").read(), FIX2["cases"]["fuzz3_FR"])
