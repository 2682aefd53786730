"""JunctionSystem::determineStrandedness in the oracle (lib/src/junction_system.cc:455-560) on hand-made totals."""
import numpy as np

from oracle import oracle as orc


def _rows(spec):
    """spec: list of (ss_strand, r1pos, r1neg, r2pos, r2neg)."""
    r = np.zeros(len(spec), dtype=orc.ROW_DTYPE)
    for k, (ss, a, b, c, d) in enumerate(spec):
        r["ss_strand"][k], r["r1pos"][k], r["r1neg"][k], r["r2pos"][k], r["r2neg"][k] = ss, a, b, c, d
    return r


POS, NEG, UNK = 0, 1, 2


def test_protocol_table():
    ds = orc.determine_strandedness
    assert ds(_rows([])) == (4, 3)                                                    # no alignments: unknown / unknown
    assert ds(_rows([(UNK, 5, 5, 0, 0)])) == (4, 3)                                   # only unknown splice sites count nothing
    assert ds(_rows([(POS, 10, 0, 0, 0), (NEG, 0, 10, 0, 0)])) == (0, 2)              # SE, R1 agrees: secondstrand
    assert ds(_rows([(POS, 1, 9, 0, 0), (NEG, 9, 1, 0, 0)])) == (0, 1)                # SE, R1 disagrees: firststrand
    assert ds(_rows([(POS, 6, 4, 0, 0), (NEG, 5, 5, 0, 0)])) == (0, 3)                # SE: R2 ratios are 0/0 = NaN -> not "unstranded"
    assert ds(_rows([(POS, 9, 1, 1, 9), (NEG, 1, 9, 9, 1)])) == (1, 2)                # FR secondstrand
    assert ds(_rows([(POS, 1, 9, 9, 1), (NEG, 9, 1, 1, 9)])) == (1, 1)                # FR firststrand
    assert ds(_rows([(POS, 9, 1, 9, 1), (NEG, 1, 9, 1, 9)])) == (3, 2)                # FF secondstrand
    assert ds(_rows([(POS, 1, 9, 1, 9), (NEG, 9, 1, 9, 1)])) == (3, 1)                # FF firststrand
    assert ds(_rows([(POS, 5, 5, 6, 4), (NEG, 4, 6, 5, 5)])) == (1, 0)                # paired, no correlation: unstranded
    assert ds(_rows([(POS, 9, 1, 5, 5), (NEG, 1, 9, 5, 5)])) == (1, 3)                # R1 stranded, R2 not: unknown
    # the 0.5 thresholds are strict / inclusive as written: ratio exactly 0.5 is neither "> 0.5" nor "> 0.5" unstranded-exempt
    assert ds(_rows([(POS, 3, 1, 1, 3), (NEG, 1, 3, 3, 1)])) == (1, 0)                # all |ratios| == 0.5 -> unstranded
