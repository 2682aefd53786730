#!/usr/bin/env python3
"""Second witness for the .junctions.tab contract (build container only: it imports the REFERENCE's own parser).

The reference ships two implementations of the .tab format: the C++ writer (lib/include/portcullis/junction.hpp:1260-1319,
lib/src/junction.cc:1224-1230) that the oracle restates, and -- independent of it -- the Python reader/writer of junctools
(scripts/junctools/junctools/junction.py:579-799, class TabJunction).  This script feeds the oracle's .tab of fixed inputs to
the reference's TabJunction, as imported from /root/reference, and records what IT makes of every line:

  * TabJunction().file_header()            -- junctools' own idea of the 75 column names, in order
  * parse_line(line)                       -- must accept the column count; int() conversions of the coordinate columns
  * the parsed attributes and str(junction) -- junctools' re-serialisation of the row

into tests/golden/junctools_tab.json; and, for the same inputs, what the reference's BedJunction.parse_line makes of every line of the
oracle's .junctions.bed and its GFFJunction.parse_line of every line of the .introns.gff3 (tests/golden/junctools_bed_gff.json).  tests/test_oracle_junctools.py re-derives the same .tab text from the oracle (CPU) and
tests/test_gpu_junctools.py from the device rows through the C++ writer, and both compare with these vectors.  Nothing of
junctools travels: the fixture holds inputs' names and expected outputs only.

    python tests/golden/make_junctools_fixture.py        (needs /root/reference)
"""
import importlib.util
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/scripts/junctools/junctools/junction.py"


def load_reference_parser():
    spec = importlib.util.spec_from_file_location("ref_junctools_junction", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def cases():
    """name -> (refs, genomes, batches, orientation): the inputs both tests rebuild (tests/junctools_cases.py)."""
    from junctools_cases import build_cases
    return build_cases()


def main():
    from oracle import oracle as orc
    ref = load_reference_parser()
    header = ref.TabJunction().file_header()
    out = {"_made_by": "tests/golden/make_junctools_fixture.py", "_reference_parser": "scripts/junctools/junctools/junction.py:579-799 (TabJunction)",
           "header": header, "n_columns": len(header.split("\t")), "cases": {}}
    out2 = {"_made_by": "tests/golden/make_junctools_fixture.py",
            "_reference_parsers": "scripts/junctools/junctools/junction.py:414-456 (BedJunction.parse_line), :531-577 (GFFJunction.parse_line)", "cases": {}}
    for name, (refs, genomes, batches, orientation) in cases().items():
        rows, _tot = orc.run_prep_like(refs, genomes, batches, orientation)
        tab = orc.write_tab(rows, [n for n, _ in refs], [l for _, l in refs]).decode()
        lines = tab.split("\n")
        assert lines[0] == header, "the oracle's header line is not junctools' file_header()"
        assert lines[-1] == "" and lines[-2] == "", "saveAll ends the table with an empty line"
        parsed = []
        for line in lines[1:-2]:
            tj = ref.TabJunction()
            assert tj.parse_line(line) is tj
            assert str(tj) == line, "junctools' re-serialisation differs from the line it parsed"
            assert tj.size() == int(line.split("\t")[6])
            parsed.append({
                "id": tj.id, "refid": tj.refid, "refseq": tj.refseq, "reflen": tj.reflen, "start": tj.start, "end": tj.end, "size": tj.size(),
                "left": tj.left, "right": tj.right, "read_strand": tj.read_strand, "ss_strand": tj.ss_strand, "strand": tj.strand,
                "ss1": tj.ss1, "ss2": tj.ss2, "metrics": dict(zip(ref.TabJunction.metric_names(), tj.metrics)),
                "jo": list(tj.jo), "raw": tj.getRaw(), "reliable": tj.getReliable(), "entropy": tj.getEntropy(), "maxmmes": tj.getMaxMMES(),
                "min_hamming": tj.getMinHamming(), "nb_samples": tj.getNbSamples(), "ss_type": tj.getSSType(), "str": str(tj)})
        out["cases"][name] = {"orientation": orientation, "n_rows": len(parsed), "rows": parsed}
        print(f"{name}: {len(parsed)} rows parsed and re-serialised by the reference's TabJunction")
        # ---- the other two writers' witnesses: BedJunction.parse_line (junction.py:414-456) on the oracle's .junctions.bed,
        # GFFJunction.parse_line (junction.py:531-577) on its .introns.gff3 -- what the reference's readers make of every line
        names = [n for n, _ in refs]
        bed_lines = orc.write_bed(rows, names, version="1.2.4").decode().split("\n")
        bed = []
        for line in bed_lines:
            bj = ref.BedJunction()
            if bj.parse_line(line) is None:  # (the track line, the empty end)
                bed.append({"line": line, "parsed": None})
                continue
            bed.append({"line": line, "parsed": {"refseq": bj.refseq, "start": bj.start, "end": bj.end, "left": bj.left, "right": bj.right,
                                                  "strand": bj.strand, "id": bj.id, "score": bj.score, "style": str(bj.style),
                                                  "rgb": [bj.red, bj.green, bj.blue]}})
        gff_lines = orc.write_intron_gff(rows, names).decode().split("\n")
        gff = []
        for line in gff_lines:
            gj = ref.GFFJunction()
            if gj.parse_line(line) is None:  # (comment lines, the empty end)
                gff.append({"line": line, "parsed": None})
                continue
            gff.append({"line": line, "parsed": {"refseq": gj.refseq, "start": gj.start, "end": gj.end, "strand": gj.strand, "source": gj.source,
                                                  "feature": gj.feature, "score": gj.score, "frame": gj.frame, "raw": gj.raw, "attrs": list(gj.attrs)}})
        n_bed = sum(1 for e in bed if e["parsed"])
        n_gff = sum(1 for e in gff if e["parsed"])
        assert n_bed == len(parsed) == n_gff, (n_bed, len(parsed), n_gff)
        for e, g_, t in zip([e for e in bed if e["parsed"]], [e for e in gff if e["parsed"]], parsed):  # the three files name the same introns
            assert (e["parsed"]["refseq"], e["parsed"]["start"], e["parsed"]["end"]) == (t["refseq"], t["start"], t["end"]) == \
                   (g_["parsed"]["refseq"], g_["parsed"]["start"], g_["parsed"]["end"])
            assert (e["parsed"]["left"], e["parsed"]["right"]) == (t["left"], t["right"])
        out2["cases"][name] = {"bed": bed, "intron_gff": gff}
        print(f"{name}: {n_bed} .bed lines parsed by BedJunction, {n_gff} .gff3 lines by GFFJunction")
    with open(os.path.join(HERE, "junctools_tab.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
        f.write("\n")
    with open(os.path.join(HERE, "junctools_bed_gff.json"), "w") as f:
        json.dump(out2, f, indent=0, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
