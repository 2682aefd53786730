"""Inputs for the `junc --extra` tests: read names (with multi-mapping), placed-unmapped records and the oracle
side of the comparison."""
import numpy as np

from fuzzgen import make_reads, to_batch


def add_names(reads, rng, tag, pool=None, share=0.25, unmapped_frac=0.02):
    """Give every read a name; `share` of them reuse a name from `pool` (multi-mapped fragments, also across
    contigs).  A few unspliced reads are marked unmapped (placed next to their mate)."""
    pool = [] if pool is None else pool
    for k, r in enumerate(reads):
        if pool and rng.random() < share:
            r["name"] = pool[int(rng.integers(0, len(pool)))]
        else:
            r["name"] = f"{tag}.{k}"
            pool.append(r["name"])
        if "N" not in r["cigar"] and rng.random() < unmapped_frac:
            r["flag"] = r.get("flag", 0) | 0x4
    return pool


def name_hashes(orc, reads):
    return np.array([orc.name_hash(r["name"], r.get("flag", 0)) for r in reads], dtype=np.uint64)


def batch_with_names(orc, reads):
    b = to_batch(reads)
    b.name_hash = name_hashes(orc, reads)
    return b


def oracle_extra(orc, contigs, orientation="UNKNOWN"):
    """contigs: list of (genome str, reads list or None) by tid.  Returns finalised oracle rows with the extra
    columns filled, and the read-length totals."""
    rows_all, soa, nh, lens = [], {}, {}, []
    spliced = unspliced = sum_len = 0
    max_len = 0
    for tid, (genome, reads) in enumerate(contigs):
        lens.append(len(genome))
        if not reads:
            continue
        b = batch_with_names(orc, reads)
        soa[tid] = b
        nh[tid] = b.name_hash
        rows, reg = orc.find_juncs(tid, len(genome), genome, soa[tid], orientation)
        rows_all.append(rows)
        spliced += reg["spliced"]
        unspliced += reg["unspliced"]
        sum_len += reg["sum_len"]
        max_len = max(max_len, reg["max_len"])
    rows = np.concatenate(rows_all) if rows_all else np.zeros(0, dtype=orc.ROW_DTYPE)
    rows = orc.finalize(rows, sum_len / max(spliced + unspliced, 1))
    rows = orc.extra(lens, soa, nh, rows, max_len)
    return rows, lens


def device_extra(ffi, orc, contigs, orientation="UNKNOWN", split=None, queue=1, dense=False):
    """The same contigs through a PJB_FLAG_EXTRA context.  Returns (rows, extra rows).  queue > 1: that many targets'
    kernel chains queued at once (pjb_finish_contig_begin / _end); dense: the depth-vector path for every target
    (pjb_set_option "extra_dense")."""
    with ffi.Context(0, orientation, flags=ffi.FLAG_EXTRA) as ctx:
        ctx.set_refs([len(g) for g, _ in contigs])
        ctx.clear_rows()
        if dense:
            ctx.set_option("extra_dense", 1)
        queued = []
        for tid, (genome, reads) in enumerate(contigs):
            ctx.upload_contig(tid, genome.encode())
            if reads:
                b = batch_with_names(orc, reads)
                if split and b.n > 4:
                    cuts = sorted(set([0, b.n] + [int(b.n * f) for f in split]))
                    for lo, hi in zip(cuts[:-1], cuts[1:]):
                        ctx.submit_batch(tid, b.slice(lo, hi))
                else:
                    ctx.submit_batch(tid, b)
            if queue <= 1:
                ctx.finish_contig(tid)
                continue
            if len(queued) >= queue:
                ctx.finish_contig_end(queued.pop(0))
            ctx.finish_contig_begin(tid)
            queued.append(tid)
        for tid in queued:
            ctx.finish_contig_end(tid)
        rows = ctx.collect()
        extra = ctx.extra_finish()
    return rows, extra


def assert_extra_equal(rows, extra, orows):
    assert len(rows) == len(orows) == len(extra)
    assert (rows["refid"] == orows["refid"]).all() and (rows["start"] == orows["start"]).all() and (rows["end"] == orows["end"]).all()
    for f in ("up_aln", "down_aln"):
        bad = np.nonzero(extra[f] != orows[f])[0]
        assert bad.size == 0, (f, bad.size, rows["refid"][bad[0]], rows["start"][bad[0]], rows["end"][bad[0]], extra[f][bad[0]], orows[f][bad[0]])
    # doubles: the device evaluates the reference's own expressions in IEEE f64 -> identical bits expected;
    # north_star's tolerance for floating metrics is 1e-6
    for f in ("mm_score", "coverage"):
        d = np.abs(extra[f] - orows[f])
        assert (d <= 1e-6).all(), (f, float(d.max()), int(np.argmax(d)), extra[f][np.argmax(d)], orows[f][np.argmax(d)])
    return float(np.abs(extra["coverage"] - orows["coverage"]).max()) if len(rows) else 0.0
