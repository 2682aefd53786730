"""Second witness for the .tab contract: tests/golden/junctools_tab.json holds what the REFERENCE's own Python parser
(scripts/junctools/junctools/junction.py:579-799, TabJunction) made of the oracle's .tab for fixed inputs -- its header, its
parse of every column, its re-serialisation of every row (tests/golden/make_junctools_fixture.py made it in the build
container, where /root/reference exists).  Here: the oracle's .tab for the same inputs must still be exactly those rows."""
import json
import os

import pytest

from junctools_cases import build_cases

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "junctools_tab.json")))


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def check_tab_against_fixture(tab_text, case):
    """tab_text: the whole .junctions.tab; case: a fixture entry.  Header, row count, every row as junctools re-serialised it,
    and junctools' typed view of the columns it converts."""
    lines = tab_text.split("\n")
    assert lines[0] == FIX["header"], "header differs from junctools' TabJunction.file_header()"
    assert len(lines[0].split("\t")) == FIX["n_columns"] == 75
    assert lines[-1] == "" and lines[-2] == "", "saveAll ends the table with an empty line (junction_system.cc:356)"
    rows = lines[1:-2]
    assert len(rows) == case["n_rows"]
    for line, want in zip(rows, case["rows"]):
        assert line == want["str"]
        c = line.split("\t")
        assert (c[0], int(c[1]), c[2], int(c[3]), int(c[4]), int(c[5]), int(c[6]), int(c[7]), int(c[8])) == (
            want["id"], want["refid"], want["refseq"], want["reflen"], want["start"], want["end"], want["size"], want["left"], want["right"])
        assert (c[9], c[10], c[11], c[12], c[13]) == (want["read_strand"], want["ss_strand"], want["strand"], want["ss1"], want["ss2"])
        names = list(want["metrics"])
        assert len(names) == 41 and FIX["header"].split("\t")[14:55] == sorted(names, key=FIX["header"].split("\t").index)
        for k, name in enumerate(FIX["header"].split("\t")[14:55]):
            assert c[14 + k] == want["metrics"][name], (name, c[14 + k], want["metrics"][name])
        assert c[55:75] == want["jo"]
        assert int(want["metrics"]["nb_raw_aln"]) == want["raw"] and int(c[26]) == want["reliable"]
        assert float(want["metrics"]["entropy"]) == want["entropy"]


@pytest.mark.parametrize("name", sorted(FIX["cases"]))
def test_oracle_tab_is_what_junctools_parsed(orc, name):
    refs, genomes, batches, orientation = build_cases()[name]
    assert orientation == FIX["cases"][name]["orientation"]
    rows, _ = orc.run_prep_like(refs, genomes, batches, orientation)
    tab = orc.write_tab(rows, [n for n, _ in refs], [l for _, l in refs]).decode()
    check_tab_against_fixture(tab, FIX["cases"][name])


def test_fixture_covers_the_format():
    """the vectors exercise what the format can hold: all three strands, canonical / semi / non-canonical sites, the 4294967295
    sentinel, fractional entropy and mean_mismatches, suspicious rows"""
    rows = [r for c in FIX["cases"].values() for r in c["rows"]]
    assert {r["strand"] for r in rows} >= {"+", "-", "?"}
    assert {r["ss_type"] for r in rows} >= {"C", "S", "N"}
    assert any("4294967295" in (r["metrics"]["dist_2_up_junc"], r["metrics"]["dist_2_down_junc"]) for r in rows)
    assert any("." in r["metrics"]["entropy"] for r in rows) and any("." in r["metrics"]["mean_mismatches"] for r in rows)
    assert any(r["metrics"]["suspicious"] == "1" for r in rows)


# ---- the .junctions.bed and .junctions.intron.gff3 witnesses (tests/golden/junctools_bed_gff.json): what the reference's own
# BedJunction.parse_line (scripts/junctools/junctools/junction.py:414-456) and GFFJunction.parse_line (:531-577) made of every line
FIX2 = json.load(open(os.path.join(HERE, "golden", "junctools_bed_gff.json")))


def check_bed_against_fixture(bed_text, case):
    lines = bed_text.split("\n")
    assert lines == [e["line"] for e in case["bed"]]
    for line, e in zip(lines, case["bed"]):
        c = line.split("\t")
        p = e["parsed"]
        if p is None:  # the track line / the empty end: junctools skips anything that has not 6 or 12 columns
            assert len(c) not in (6, 12)
            continue
        # junctools' reading of the 12 columns: thickStart / thickEnd are the intron, chromStart / chromEnd the anchors' outer ends
        assert len(c) == 12 and (c[0], int(c[6]), int(c[7]) - 1, int(c[1]), int(c[2]) - 1, c[5], c[3], float(c[4])) == (
            p["refseq"], p["start"], p["end"], p["left"], p["right"], p["strand"], p["id"], p["score"])
        assert [int(x) for x in c[8].split(",")] == p["rgb"]
        assert p["start"] != p["left"], "portcullis writes the exon-anchored style (junctools: EBED), not tophat's"


def check_gff_against_fixture(gff_text, case):
    lines = gff_text.split("\n")
    assert lines == [e["line"] for e in case["intron_gff"]]
    for line, e in zip(lines, case["intron_gff"]):
        p = e["parsed"]
        if p is None:
            assert line.startswith("#") or len(line.split("\t")) <= 1
            continue
        c = line.split("\t")
        assert len(c) == 9 and c[2] == "intron"
        assert (c[0], int(c[3]) - 1, int(c[4]) - 1, c[6], c[1], c[7]) == (p["refseq"], p["start"], p["end"], p["strand"], p["source"], p["frame"])
        assert (float(c[5]) if c[5] != "." else 0.0) == p["score"] and c[8].split(";") == p["attrs"]
        mult = [a.split("=")[1] for a in p["attrs"] if a.startswith("mult")]
        assert not mult or int(mult[0]) == p["raw"]


@pytest.mark.parametrize("name", sorted(FIX2["cases"]))
def test_oracle_bed_and_gff_are_what_junctools_parsed(orc, name):
    refs, genomes, batches, orientation = build_cases()[name]
    rows, _ = orc.run_prep_like(refs, genomes, batches, orientation)
    names = [n for n, _ in refs]
    check_bed_against_fixture(orc.write_bed(rows, names, version="1.2.4").decode(), FIX2["cases"][name])
    check_gff_against_fixture(orc.write_intron_gff(rows, names).decode(), FIX2["cases"][name])
    # the three files name the same introns: junctools' three parsers agree line by line
    tab = FIX["cases"][name]["rows"]
    bed = [e["parsed"] for e in FIX2["cases"][name]["bed"] if e["parsed"]]
    gff = [e["parsed"] for e in FIX2["cases"][name]["intron_gff"] if e["parsed"]]
    assert len(tab) == len(bed) == len(gff)
    for t, b, g in zip(tab, bed, gff):
        assert (t["refseq"], t["start"], t["end"]) == (b["refseq"], b["start"], b["end"]) == (g["refseq"], g["start"], g["end"])
        assert (t["left"], t["right"]) == (b["left"], b["right"])
