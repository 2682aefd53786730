"""filt-stage feature rows: ModelFeatures::setRow on the device (pjb_filt_features) against the
oracle, with Markov models trained by the oracle on the same junctions.  Integer-valued columns are exact; the columns
that go through log / log2 and products of probabilities are compared to 1e-6 (north_star's tolerance for floating
metrics)."""
import numpy as np
import pytest

from fuzzgen import make_reads, to_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ffi():
    from portcullis_amd import ffi as f
    assert f.device_count() >= 1
    return f


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as o
    return o


def _setup(ffi, orc, seeds, paired=False, upper=True):
    contigs, rows_all, dev_rows = [], [], []
    tot_len = tot_n = 0
    ctx = ffi.Context(0, "FR" if paired else "UNKNOWN")
    lens = []
    for tid, seed in enumerate(seeds):
        genome, reads = make_reads(seed, n_reads=2500, paired=paired, glen=24000)
        if upper:
            genome = genome.upper()
        contigs.append(genome)
        lens.append(len(genome))
    ctx.set_refs(lens)
    ctx.clear_rows()
    for tid, seed in enumerate(seeds):
        genome, reads = make_reads(seed, n_reads=2500, paired=paired, glen=24000)
        b = to_batch(reads)
        rows, reg = orc.find_juncs(tid, len(contigs[tid]), contigs[tid], b, "FR" if paired else "UNKNOWN")
        rows_all.append(rows)
        tot_len += reg["sum_len"]
        tot_n += reg["spliced"] + reg["unspliced"]
        ctx.upload_contig(tid, contigs[tid].encode())
        ctx.submit_batch(tid, b)
        ctx.finish_contig(tid)
        ctx.upload_contig(tid, contigs[tid].encode())      # (finish does not release it; kept for the feature windows)
    orows = orc.finalize(np.concatenate(rows_all), tot_len / tot_n)
    drows = ctx.collect()
    assert (drows["start"] == orows["start"]).all() and (drows["refid"] == orows["refid"]).all()
    return ctx, contigs, lens, orows, drows


def _compare(F, G):
    assert F.shape == G.shape
    exact = [0, 1, 2, 3, 6, 7, 10]
    assert (F[:, exact] == G[:, exact]).all()
    assert (np.isnan(F) == np.isnan(G)).all() and (np.isinf(F) == np.isinf(G)).all()   # log2 of a negative expectation etc.
    fin = np.isfinite(G)
    assert (F[~fin & ~np.isnan(G)] == G[~fin & ~np.isnan(G)]).all()
    d = np.where(fin, np.abs(np.where(fin, F, 0) - np.where(fin, G, 0)), 0.0)
    tol = 1e-6 * np.maximum(1.0, np.abs(np.where(fin, G, 0)))
    bad = np.argwhere(d > tol)
    assert bad.size == 0, (bad[:5], F[tuple(bad[0])], G[tuple(bad[0])])
    return float(d.max())


@pytest.mark.parametrize("seeds,paired", [((61, 62, 63), False), ((64, 65), True)])
def test_feature_rows_match_oracle(ffi, orc, seeds, paired):
    ctx, contigs, lens, orows, drows = _setup(ffi, orc, seeds, paired)
    try:
        n = len(orows)
        idx = np.arange(n)
        good = idx[orows["nb_raw"] >= 3]
        bad = idx[orows["nb_raw"] < 3]
        assert len(good) > 10 and len(bad) >= 3
        sizes = orows["end"] - orows["start"] + 1
        small = idx[sizes <= np.median(sizes)]            # L95 from the shorter half, so that some introns score above it
        G, models, l95 = orc.filt_features(lens, dict(enumerate(contigs)), orows, small, good, good, bad)
        assert models["exon_size"] > 100 and models["donor_pw_size"] == 23 and l95 > 0
        F = ctx.filt_features(drows, float(orows["mean_readlen"][0]), l95, models)
        worst = _compare(F, G)
        assert worst < 1e-6
        assert (np.abs(G[:, 11]) > 0).any() and (G[:, 12] != 0).any() and (G[:, 13] != 0).any() and (G[:, 9] > 0).any()
        assert (orows["cons_strand"] == 1).any() and (orows["cons_strand"] == 0).any()   # both orientations of the windows
        # untrained models: coding column 0, position / signal columns hold the scores of the empty models
        G0, m0, _ = orc.filt_features(lens, dict(enumerate(contigs)), orows, [], [], [], [])
        F0 = ctx.filt_features(drows, float(orows["mean_readlen"][0]), 0, {})
        _compare(F0, G0)
        assert (G0[:, 11] == 0).all() and (G0[:, 9] == 0).all() and (G0[:, 12] == -600.0).all()
    finally:
        ctx.close()


def test_feature_windows_at_contig_edges(ffi, orc):
    """Junctions a few bases from either end of a short contig: every window is clamped like faidx_fetch_seq clamps it."""
    from portcullis_amd.records import ReadBatch
    g = ("ACGTTGCAAGGCTTAACCGGTTAACG" * 8)[:200]
    reads = [dict(pos=0, cigar="4M20N30M", seq="A" * 34, xs="+"), dict(pos=2, cigar="3M19N30M", seq="C" * 33, xs="-"),
             dict(pos=150, cigar="20M25N5M", seq="G" * 25, xs="-"), dict(pos=151, cigar="20M20N9M", seq="T" * 29, xs="+")]
    b = ReadBatch.from_reads(reads)
    rows, reg = orc.find_juncs(0, len(g), g, b, "UNKNOWN")
    orows = orc.finalize(rows, 30.0)
    with ffi.Context(0, "UNKNOWN") as ctx:
        ctx.set_refs([len(g)])
        ctx.upload_contig(0, g.encode())
        ctx.clear_rows()
        ctx.submit_batch(0, b)
        ctx.finish_contig(0)
        ctx.upload_contig(0, g.encode())
        drows = ctx.collect()
        idx = np.arange(len(orows))
        G, models, l95 = orc.filt_features([len(g)], {0: g}, orows, idx, idx, idx[:2], idx[2:])
        F = ctx.filt_features(drows, 30.0, l95, models)
        _compare(F, G)
        with pytest.raises(ffi.PjbError):
            ctx.release_contig(0)
            ctx.filt_features(drows, 30.0, l95, models)   # genome gone


def test_model_features_class_from_tab(tmp_path, orc):
    """The host-side mirror of the reference's ModelFeatures (train on the host, matrix from the device) on a .tab the
    junc program wrote, against the oracle fed with the same junctions and the same positive / negative sets."""
    import os
    import subprocess

    from util_bam import make_prep_dir
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "portcullis_amd", "host")
    csrc = os.path.join(root, "portcullis_amd", "csrc")
    exe = str(tmp_path / "model_features")
    subprocess.check_call(["g++", "-O1", "-std=c++17", f"-I{host}/include", f"-I{root}/include", "-o", exe,
                           os.path.join(root, "tests", "cpp", "model_features.cc"), f"-L{host}", "-lportcullis_host",
                           f"-L{csrc}", "-lportcullis_amd", f"-Wl,-rpath,{host}", f"-Wl,-rpath,{csrc}"])
    refs, contigs, reads, rows_all = [], [], [], []
    tot_len = tot_n = 0
    for tid, seed in enumerate([71, 72]):
        genome, rr = make_reads(seed, n_reads=2500, glen=22000)
        genome = genome.upper()
        for k, r in enumerate(rr):
            r["tid"] = tid
        refs.append((f"chr{tid + 1}", len(genome)))
        contigs.append((f"chr{tid + 1}", genome))
        reads += rr
        rows, reg = orc.find_juncs(tid, len(genome), genome, to_batch(rr), "UNKNOWN")
        rows_all.append(rows)
        tot_len += reg["sum_len"]
        tot_n += reg["spliced"] + reg["unspliced"]
    prep = make_prep_dir(str(tmp_path / "prep"), refs, contigs, reads)
    out = str(tmp_path / "junc" / "pc")
    p = subprocess.run([os.path.join(host, "portcullis_amd"), "junc", "-o", out, prep], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    txt = str(tmp_path / "features.txt")
    p = subprocess.run([exe, os.path.join(prep, "portcullis.genome.fa"), out + ".junctions.tab", txt], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-2000:]
    lines = open(txt).read().strip().split("\n")
    F = np.array([[float(v) for v in l.split("\t")] for l in lines[1:]])
    orows = orc.finalize(np.concatenate(rows_all), tot_len / tot_n)
    idx = np.arange(len(orows))
    good, bad = idx[orows["nb_raw"] >= 3], idx[orows["nb_raw"] < 3]
    G, models, l95 = orc.filt_features([l for _, l in refs], {t: g for t, (_, g) in enumerate(contigs)}, orows, idx, good, good, bad)
    assert lines[0] == f"# L95={l95} exon={models['exon_size']} intron={models['intron_size']} donorPW={models['donor_pw_size']}"
    assert F.shape == G.shape
    loose = [4, 8]                                        # entropy and mean_mismatches pass through the .tab's 6 significant digits
    for k in range(F.shape[1]):
        tol = (1e-5 if k in loose else 1e-6) * np.maximum(1.0, np.abs(G[:, k]))
        fin = np.isfinite(G[:, k])
        assert (np.isfinite(F[:, k]) == fin).all()
        assert (np.abs(F[fin, k] - G[fin, k]) <= tol[fin]).all(), (k, np.abs(F[fin, k] - G[fin, k]).max())
