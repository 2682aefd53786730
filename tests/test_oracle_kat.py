"""Pins the CPU oracle against every known-answer test the reference holds for
the junc path.  Expected strings/values are the
reference tests' own literals."""
import pytest

from oracle import oracle as orc


# /root/reference/tests/bam_tests.cpp:181-202
def test_padding1():
    cigar = "2S14M2I1M1737N8M14S"
    query = "AGAAAGTGGAGAAAAGAATTTGGTGTGGATGATCTTATCACAACCATTCTTTCTGGTGAGACAGAAGC"
    genomic = "AAAGTGGAGAAAAGAATTTGGTGTGGATGATCTTATCACAACCATTCTTTCTGGTGAGACAGAAGC"
    q, left, right = orc.padded_query_seq(cigar, 609263, 1787, query, 609263, 609304)
    g = orc.padded_genome_seq(cigar, 609263, 1787, genomic, 609263, 609304, left, right)
    assert len(q) == len(g)
    assert q == "AAAGTGGAGAAAAGAAT"
    assert g == "AAAGTGGAGAAAAGXXA"


# /root/reference/tests/bam_tests.cpp:204-225
def test_padding2():
    cigar = "14S13M1I2601N9M4918N13M18S"
    query = "ATTGGGGTGTAGATAATTTTATAAAAATTTTTATTTAGGAGGAAAAAAAGGCCGTTTCCAAATATTAC"
    genomic = "AATTTTATAAAAAAACGGAACTCCGGC"
    q, left, right = orc.padded_query_seq(cigar, 750577, 7586, query, 750577, 750603)
    g = orc.padded_genome_seq(cigar, 750577, 7586, genomic, 750577, 750603, left, right)
    assert len(q) == len(g)
    assert q == "AATTTTATAAAAAT"
    assert g == "AATTTTATAAAAAX"


# /root/reference/tests/bam_tests.cpp:227-248
def test_padding3():
    cigar = "30S8M25N2M5D28M"
    query = "ACAAAAACAGAAAAAAAAAGAAAAAAAAATACCAAAACCAACGCCTTCACTTAAAGACAAATATTCAA"
    genomic = "TACCAAAG"
    q, left, right = orc.padded_query_seq(cigar, 4776643, 98, query, 4776673, 4776680)
    g = orc.padded_genome_seq(cigar, 4776643, 98, genomic, 4776673, 4776680, left, right)
    assert len(q) == len(g)
    assert q == "CAXXX"
    assert g == "CAAAG"


# /root/reference/tests/intron_tests.cpp:53-65
def test_intron_min_anchor_and_size():
    assert orc.min_anchor(10, 20, 4, 40) == 6
    assert 20 - 10 + 1 == 11  # Intron::size(), intron.hpp


def test_intron_min_anchor_throws():
    # lib/src/intron.cc:67-83
    with pytest.raises(orc.OracleError):
        orc.min_anchor(10, 20, 11, 40)
    with pytest.raises(orc.OracleError):
        orc.min_anchor(10, 20, 4, 19)


# /root/reference/tests/junction_tests.cpp:48-88
def test_donor_acceptor():
    C, S, N = 0, 1, 2
    assert orc.donor_acceptor("GT", "AG")[0] == C
    assert orc.donor_acceptor("CT", "AC")[0] == C
    with pytest.raises(orc.OracleError):
        orc.donor_acceptor("GTA", "AG")
    assert orc.donor_acceptor("CT", "AG")[0] != C
    assert orc.donor_acceptor("GT", "AC")[0] != C
    with pytest.raises(orc.OracleError):
        orc.donor_acceptor("", "")
    # semi-canonical classes, lib/include/portcullis/junction.hpp:73-79
    for a, b in (("AT", "AC"), ("GT", "AT"), ("GC", "AG"), ("CT", "GC")):
        assert orc.donor_acceptor(a, b)[0] == S
    assert orc.donor_acceptor("GG", "AG")[0] == N


def test_donor_acceptor_strands():
    POS, NEG, UNK = 0, 1, 2
    css, ss, cons, da1, da2 = orc.donor_acceptor("CT", "AC", read_strand=UNK)
    assert (ss, cons) == (NEG, NEG)
    assert (da1, da2) == (b"GT", b"AG")  # rev-comp swapped, junction.cc:513-514
    css, ss, cons, da1, da2 = orc.donor_acceptor("CT", "AC", read_strand=POS)
    assert (ss, cons) == (NEG, UNK) and (da1, da2) == (b"CT", b"AC")
    css, ss, cons, da1, da2 = orc.donor_acceptor("GG", "AG", read_strand=NEG)
    assert (ss, cons) == (UNK, NEG) and (da1, da2) == (b"CT", b"CC")


# /root/reference/tests/junction_tests.cpp:90-108 (ordering only) + SURVEY App. A values
def test_entropy():
    e1 = orc.entropy([13, 15, 17, 19])
    e2 = orc.entropy([16, 16, 16, 16])
    assert e1 > e2
    assert e2 == 0.0
    assert e1 == 1.5  # quirk grouping {2,1,1}
    assert abs(orc.entropy([990, 1000, 1120]) - 0.918296) < 5e-7
    assert orc.entropy([5]) == 0.0


# /root/reference/tests/seq_utils_tests.cpp:30-48
def test_seq_utils():
    assert orc.hamming("ATGC", "ATGC") == 0
    assert orc.hamming("ATGC", "ATGG") == 1
    assert orc.hamming("ATGC", "CGTA") == 4
    assert orc.revcomp("ATGC") == b"GCAT"
    with pytest.raises(orc.OracleError):
        orc.hamming("ATG", "AT")
    assert orc.hamming("atgc", "ATGC") == 0  # upper-cases both, seq_utils.hpp:68-69
