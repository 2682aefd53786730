"""The two micro-fixtures whose reference outputs are recorded in SURVEY.md
Appendix A (produced by the real reference during the survey).  Genome =
tests/golden/spombe_III_30k.fa (first 30 kb of the reference's
tests/resources/spombe.III.fa); read bases are copied from the genome,
inserted / soft-clipped bases are 'A'."""
from portcullis_amd.records import CIGAR_CHARS, encode_cigar


def read_from_genome(genome, pos, cigar, **kw):
    g = genome.upper() if isinstance(genome, str) else genome.decode().upper()
    seq = []
    r = pos
    for op in encode_cigar(cigar):
        l, c = int(op) >> 4, CIGAR_CHARS[int(op) & 15]
        if c in "M=X":
            seq.append(g[r:r + l])
            r += l
        elif c in "IS":
            seq.append("A" * l)
        elif c in "DN":
            r += l
    seq = "".join(seq)
    sub = kw.pop("sub", None)
    if sub is not None:
        b = seq[sub]
        seq = seq[:sub] + ("C" if b != "C" else "G") + seq[sub + 1:]
    d = dict(pos=pos, cigar=cigar, seq=seq, flag=0, mapq=60, xs="+", mtid=-1, mpos=-1)
    d.update(kw)
    return d


def micro1(genome):
    return [
        read_from_genome(genome, 990, "180M200N30M"),
        read_from_genome(genome, 1000, "30M100N40M200N30M"),
        read_from_genome(genome, 1100, "5S65M200N30M2S", flag=16, mapq=3),
        read_from_genome(genome, 1120, "20M2D28M200N10M1I19M"),
    ]


def micro2(genome):
    return [
        read_from_genome(genome, 5000, "50M100N50M"),
        read_from_genome(genome, 5000, "50M100N40M"),
        read_from_genome(genome, 5000, "50M100N50M"),
        read_from_genome(genome, 8000, "4M100N96M", xs=None),
        read_from_genome(genome, 12000, "60M200N40M", flag=99, mtid=0, mpos=12500),
        read_from_genome(genome, 12010, "50M200N50M", flag=147, mtid=0, mpos=11000),
        read_from_genome(genome, 12020, "40M200N60M", flag=83, mtid=0, mpos=13000),
        read_from_genome(genome, 12030, "30M200N70M", flag=1345, mapq=10, mtid=0, mpos=12600),
        read_from_genome(genome, 20000, "50M150N50M", sub=44),
    ]
