"""Hand-worked checks of the oracle's restatement of the filt-stage Markov models and feature rows
(lib/src/markov_model.cc, lib/src/model_features.cc:67-212, lib/src/junction.cc:953-956,1328-1391).  The reference holds
no test or vector for this code (tests/kmer_tests.cpp is about the k-mer hash only), so these values are derived from the
source by hand: row f4 of the oracle is "parity unpinned" in the sense of the project rules."""
import math

import numpy as np

from oracle import oracle as orc


def _row(start, end, cons=0, raw=10, **kw):
    r = np.zeros(1, dtype=orc.ROW_DTYPE)
    r["refid"], r["start"], r["end"], r["cons_strand"], r["nb_raw"] = 0, start, end, cons, raw
    r["mean_readlen"] = 100.0
    for k, v in kw.items():
        r[k] = v
    return r


def test_position_and_kmer_models_by_hand():
    # one passing junction at 30..49 on a 100-base contig: donor window = [27, 50], acceptor window = [29, 51]
    g = "ACGT" * 25
    rows = _row(30, 49)
    F, M, l95 = orc.filt_features([len(g)], {0: g}, rows, [0], [], [0], [])
    assert l95 == 20                                      # the only intron size
    don = g[27:51]
    pw = M["donor_pw"].reshape(32, 5)
    for i in range(1, len(don)):                          # PosMarkovModel::train, order 1: position i saw exactly this base
        assert pw[i]["ACGT".index(don[i])] == 1.0 and pw[i].sum() == 1.0
    assert pw[0].sum() == 0 and M["donor_pw_size"] == 23 and M["acceptor_pw_size"] == 22
    # scoring the training window itself: every probability is 1 -> log(1) = 0 for both position models;
    # the true k-mer models also give 0, the false models are untrained: no_count = 19 (18) -> score 1 / (n * 0.5)
    assert F[0][12] == 0.0
    nd, na = len(don) - 5, len(g[29:52]) - 5
    assert abs(F[0][13] - ((0 - math.log(1.0 / (nd * 0.5))) + (0 - math.log(1.0 / (na * 0.5))))) < 1e-12
    assert F[0][11] == 0.0                                # coding-potential models untrained -> column is 0
    assert F[0][9] == 0.0                                 # intron size (20) <= L95 (20)


def test_row_getters_and_jad_deviation():
    g = "ACGTTGCA" * 40
    jad = np.array([8, 8, 7, 7, 6, 5, 5, 4, 3, 3, 2, 2, 2, 1, 1, 1, 0, 0, 0, 0], dtype=np.uint32)
    rows = _row(100, 180, raw=8, nb_ms=2, nb_dist=5, nb_rel=4, entropy=1.25, max_min_anc=33, maxmmes=21, mean_mismatches=0.5,
                hamming5p=7, hamming3p=4)
    rows["jad"][0] = jad
    other = _row(10, 25)
    both = np.concatenate([other, rows])
    F, M, l95 = orc.filt_features([len(g)], {0: g}, both, [0, 1], [], [], [])
    assert l95 == 81                                      # sizes (16, 81): index int(2 * 0.95) = 1
    f = F[1]
    assert list(f[:9]) == [0.0, 6.0, 5.0, 4.0, 1.25, 0.5, 33.0, 21.0, 0.5]
    assert f[9] == 0.0 and f[10] == 4.0 and f[11] == 0.0  # size 81 <= L95; min hamming; no coding model
    assert F[0][9] == 0.0                                  # size 16 <= 81
    for i in range(20):                                   # calcJunctionAnchorDepthLogDeviation, junction.cc:1384-1391
        ni = float(jad[i]) if jad[i] else 1e-12
        want = math.log2(ni / (8.0 * (1.0 - i / (100.0 / 2.0))))
        assert abs(f[14 + i] - want) < 1e-12
    # untrained position models are "touched" by calcSplicingScores before isPWModelEmpty() is asked: the columns hold
    # the scores of empty models (PosMarkovModel: product 0 -> -300 each; KmerMarkovModel: true - false cancel)
    assert f[12] == -600.0 and f[13] == 0.0


def test_intron_score_above_threshold():
    g = "ACGT" * 300
    rows = np.concatenate([_row(100, 100 + s - 1) for s in (40,) * 19 + (400,)])
    F, _, l95 = orc.filt_features([len(g)], {0: g}, rows, list(range(20)), [], [], [])
    assert l95 == 400 and (F[:, 9] == 0).all()           # sorted sizes[int(20 * 0.95)] = sizes[19] = 400
    F, _, l95 = orc.filt_features([len(g)], {0: g}, rows, list(range(19)), [], [], [])
    assert l95 == 40 and F[19][9] == math.log(400 - 40)  # calcIntronScore, junction.cc:953-956
