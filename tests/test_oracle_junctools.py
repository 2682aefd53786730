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
