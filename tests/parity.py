"""Row-by-row comparison of the HIP path's rows with the oracle's."""
import numpy as np

INT_FIELDS = ["refid", "start", "end", "left", "right", "read_strand", "ss_strand", "cons_strand", "canonical",
              "suspicious", "nb_raw", "nb_dist", "nb_ms", "nb_um", "nb_bpp", "nb_ppp", "nb_rel", "r1pos", "r1neg",
              "r2pos", "r2neg", "max_min_anc", "maxmmes", "hamming5p", "hamming3p", "nb_up_juncs", "nb_down_juncs",
              "sum_mismatches"]
ENTROPY_TOL = 1e-6  # north_star: floating metrics within 1e-6


def sort_rows(rows):
    return rows[np.lexsort((rows["end"], rows["start"], rows["refid"]))]


def assert_rows_equal(dev, orc, entropy_tol=ENTROPY_TOL):
    dev, orc = sort_rows(dev), sort_rows(orc)
    assert len(dev) == len(orc), f"junction count differs: device {len(dev)} oracle {len(orc)}"
    for f in INT_FIELDS:
        bad = np.nonzero(dev[f] != orc[f])[0]
        assert bad.size == 0, (f"field {f} differs at {bad.size} junctions; first: key="
                               f"({orc['start'][bad[0]]},{orc['end'][bad[0]]}) device={dev[f][bad[0]]} oracle={orc[f][bad[0]]}")
    for f in ("da1", "da2", "jad"):
        bad = np.nonzero((dev[f] != orc[f]).any(axis=1))[0]
        assert bad.size == 0, (f"field {f} differs at {bad.size} junctions; first: key="
                               f"({orc['start'][bad[0]]},{orc['end'][bad[0]]}) device={dev[f][bad[0]]} oracle={orc[f][bad[0]]}")
    d = np.abs(dev["entropy"] - orc["entropy"])
    assert (d <= entropy_tol).all(), f"entropy differs by up to {d.max()}"
    return float(d.max()) if len(d) else 0.0


def region_equal(dev_reg, orc_reg):
    for k in ("spliced", "unspliced", "sum_len", "min_len", "max_len"):
        assert dev_reg[k] == orc_reg[k], (k, dev_reg[k], orc_reg[k])
