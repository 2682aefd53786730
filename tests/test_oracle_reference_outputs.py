"""Pins the CPU oracle against outputs of the REAL reference recorded in
SURVEY.md (Appendix A micro-fixtures 1 and 2, Appendix B clipped3.bam row).
Those numbers were produced by the reference's own code during the survey;
the reference cannot be rebuilt in this image (Boost is absent), so they are
kept here as literal expectations."""
import os

import numpy as np

from fixtures_micro import micro1, micro2
from oracle import oracle as orc
from portcullis_amd.records import ReadBatch
from util_bam import read_bam, records_to_batch

U32_MINUS1 = 4294967295


def run(genome, reads, orientation="UNKNOWN", ref_len=None):
    b = ReadBatch.from_reads(reads)
    rows, reg = orc.find_juncs(0, ref_len or len(genome), genome, b, orientation)
    mean = reg["sum_len"] / (reg["spliced"] + reg["unspliced"])
    rows = orc.finalize(rows, mean)
    return rows, reg


def by_key(rows):
    return {(int(r["start"]), int(r["end"])): r for r in rows}


def test_micro_fixture_1(spombe30k):
    _, genome = spombe30k
    rows, reg = run(genome, micro1(genome))
    assert len(rows) == 3
    j0, j1, j2 = rows
    assert (j0["start"], j0["end"], j0["left"], j0["right"]) == (1030, 1129, 1000, 1169)
    assert j0["nb_raw"] == 1 and j0["nb_ms"] == 1 and j0["maxmmes"] == 30
    assert j0["nb_down_juncs"] == 1 and j0["dist_up"] == 36 and j0["dist_down"] == U32_MINUS1
    assert (j1["start"], j1["end"], j1["left"], j1["right"]) == (1165, 1364, 1100, 1394)
    assert j1["nb_raw"] == 1 and j1["nb_um"] == 0 and j1["nb_rel"] == 0 and j1["r2neg"] == 1
    assert j1["dist_up"] == 0 and j1["dist_down"] == 36 and j1["dist_nearest"] == 0
    assert (j2["start"], j2["end"], j2["left"], j2["right"]) == (1170, 1369, 990, 1399)
    assert j2["nb_raw"] == 3 and j2["nb_dist"] == 3 and j2["nb_ms"] == 1
    assert f"{j2['entropy']:g}" == "0.918296"
    assert f"{j2['mean_mismatches']:g}" == "34.3333"
    assert j2["sum_mismatches"] == 103
    assert j2["maxmmes"] == 30 and j2["nb_up_juncs"] == 1
    assert list(j2["jad"][:10]) == [3] * 10 and list(j2["jad"][10:]) == [2] * 10
    assert j2["suspicious"] == 0
    for r in rows:
        assert r["mean_readlen"] == 122.0
    # flag-0 reads land in nb_r2_pos
    assert j2["r2pos"] == 3 and j0["r2pos"] == 1
    assert reg["spliced"] == 4 and reg["unspliced"] == 0 and reg["sum_len"] == 490


def test_micro_fixture_2(spombe30k):
    _, genome = spombe30k
    rows, _ = run(genome, micro2(genome), orientation="FR")
    k = by_key(rows)
    t = k[(5050, 5149)]
    assert t["nb_raw"] == 3 and t["nb_dist"] == 3 and t["nb_um"] == 3
    assert t["nb_ppp"] == 0 and t["nb_rel"] == 0 and t["r2pos"] == 3
    assert t["entropy"] == 0.0 and t["maxmmes"] == 50 and list(t["jad"]) == [3] * 20
    h = k[(8004, 8103)]
    assert (h["read_strand"], h["ss_strand"], h["cons_strand"]) == (2, 2, 2)
    assert h["max_min_anc"] == 4 and h["maxmmes"] == 4
    assert list(h["jad"]) == [1] * 4 + [0] * 16
    assert h["hamming5p"] == 3 and h["hamming3p"] == 5
    p = k[(12060, 12259)]
    assert p["nb_raw"] == 4 and p["nb_dist"] == 4 and p["nb_um"] == 3 and p["nb_bpp"] == 3
    assert p["nb_ppp"] == 2 and p["nb_rel"] == 2
    assert p["r1pos"] == 2 and p["r1neg"] == 1 and p["r2neg"] == 1 and p["r2pos"] == 0
    assert p["entropy"] == 1.5
    z = k[(20050, 20199)]
    assert z["mean_mismatches"] == 1.0 and z["maxmmes"] == 49
    assert list(z["jad"]) == [1] * 5 + [0] * 15
    assert z["suspicious"] == 1 and z["pfp"] == 0


def test_micro_fixture_2_unknown_orientation(spombe30k):
    # orientation UNKNOWN disables the portcullis proper-pair check (bam_master.hpp:149-153)
    _, genome = spombe30k
    rows, _ = run(genome, micro2(genome), orientation="UNKNOWN")
    k = by_key(rows)
    assert k[(5050, 5149)]["nb_rel"] == 3 and k[(5050, 5149)]["nb_ppp"] == 0
    assert k[(12060, 12259)]["nb_rel"] == 3


CLIPPED3_ROW = (
    "0 0 Chr4 18585056 6442658 6442841 184 6442559 6442940 - ? - GG AG N 0 0 0 135 37 135 0 131 4 83 0 131 0.97037 "
    "0 85 50 0 4.89824 0.281481 0 49 49 0 5 6 0 0 0 0 0 0 0 0 0 0 0 0 0 0 1 135 132 131 131 126 126 126 121 114 114 "
    "103 99 95 91 88 86 74 69 69 65"
).split()


def test_clipped3_genome_independent_columns(golden_dir):
    """reference row for tests/resources/clipped3.bam.  The survey's
    synthetic genome is not reproducible, so only read-derived columns are compared."""
    refs, recs = read_bam(os.path.join(golden_dir, "clipped3.bam"))
    assert refs == [("Chr4", 18585056)]
    assert len(recs) == 2128
    recs = [r for r in recs if r["tid"] == 0]
    batch = records_to_batch(recs)
    rng = np.random.default_rng(4)
    genome = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=refs[0][1]).tobytes()
    rows, reg = orc.find_juncs(0, refs[0][1], genome, batch, "UNKNOWN")
    rows = orc.finalize(rows, reg["sum_len"] / (reg["spliced"] + reg["unspliced"]))
    assert len(rows) == 1 and reg["spliced"] == 135
    tab = orc.write_tab(rows, ["Chr4"], [18585056]).decode().split("\n")
    assert tab[-1] == "" and tab[-2] == ""  # header, row, extra empty line
    got = tab[1].split("\t")
    assert len(got) == 75 and len(tab[0].split("\t")) == 75
    hdr = tab[0].split("\t")
    genome_dep = {"ss-strand", "consensus-strand", "ss1", "ss2", "canonical_ss", "mean_mismatches", "maxmmes",
                  "hamming5p", "hamming3p", "suspicious", "pfp"} | {f"JAD{i:02d}" for i in range(1, 21)}
    for name, g, e in zip(hdr, got, CLIPPED3_ROW):
        if name not in genome_dep:
            assert g == e, (name, g, e)
    # JAD01 = every alignment has minMatch >= 1 only if the first base either side matches; not pinned.
    assert got[hdr.index("max_min_anc")] == "49"
