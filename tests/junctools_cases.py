"""The inputs of the junctools witness (tests/golden/make_junctools_fixture.py, test_oracle_junctools.py, test_gpu_junctools.py):
the two micro-fixtures of SURVEY.md Appendix A on the 30 kb S. pombe genome (orientation FR) and a three-target fuzz set."""
import os

from fixtures_micro import micro1, micro2
from fuzzgen import make_reads, to_batch

HERE = os.path.dirname(os.path.abspath(__file__))


def spombe30k():
    name, seq = None, []
    for line in open(os.path.join(HERE, "golden", "spombe_III_30k.fa")):
        if line.startswith(">"):
            name = line[1:].split()[0]
        else:
            seq.append(line.strip())
    return name, "".join(seq)


def micro_reads():
    name, genome = spombe30k()
    reads = micro1(genome) + micro2(genome)
    reads.sort(key=lambda r: r["pos"])
    for r in reads:
        r["tid"] = 0
    return name, genome, reads


def fuzz_contigs(seeds=(11, 12, 13), n_reads=1500):
    """[(name, genome, reads)] -- tid = index; mates on "another" target point at the next one"""
    out = []
    for tid, seed in enumerate(seeds):
        genome, rr = make_reads(seed, n_reads=n_reads, paired=True, glen=20000 + 1000 * tid)
        for r in rr:
            r["tid"] = tid
            if r.get("mtid", -1) >= 0:
                r["mtid"] = tid if r["mtid"] == 0 else (tid + 1) % len(seeds)
        out.append((f"chr{tid + 1}", genome, rr))
    return out


def build_cases():
    cases = {}
    name, genome, reads = micro_reads()
    cases["micro_FR"] = ([(name, len(genome))], {0: genome}, {0: to_batch(reads)}, "FR")
    contigs = fuzz_contigs()
    cases["fuzz3_FR"] = ([(n, len(g)) for n, g, _ in contigs], {t: g for t, (_, g, _) in enumerate(contigs)},
                         {t: to_batch(rr) for t, (_, _, rr) in enumerate(contigs)}, "FR")
    return cases
