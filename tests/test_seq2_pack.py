"""pjb_batch.seq2 / .seq_exc (ABI 4): the packers a decoder uses (records.pack_seq2 on the host, synth.pack_seq2_torch for generated
records) against a per-base restatement of the format in include/portcullis_amd.h.  No GPU."""
import numpy as np

from portcullis_amd.records import NT16, encode_seq, pack_seq2


def slots(seqs, pad_byte):
    seq4, off = [], [0]
    for s in seqs:
        b = encode_seq(s) if s else np.zeros(0, np.uint8)
        seq4.append(np.concatenate([b, np.full((-len(b)) % 4, pad_byte, np.uint8)]))
        off.append(off[-1] + len(seq4[-1]) // 4)
    return np.concatenate(seq4) if seq4 else np.zeros(0, np.uint8), np.array(off, dtype=np.uint32)


def check(seqs, l_qseq, seq4, off):
    s2, sx = pack_seq2(seq4, off, l_qseq)
    assert s2.dtype == np.uint16 and len(s2) == off[-1] and len(sx) == (len(seqs) + 31) // 32
    for r, s in enumerate(seqs):
        exc = (int(sx[r >> 5]) >> (r & 31)) & 1
        have = (int(off[r + 1]) - int(off[r])) * 8
        want_exc = l_qseq[r] <= 0 or have < l_qseq[r] or any(c not in "ACGT" for c in s[: l_qseq[r]])
        assert exc == int(want_exc), (r, s, exc)
        for i, c in enumerate(s[: l_qseq[r]]):
            if c in "ACGT":  # (what is stored for any other letter does not matter)
                assert (int(s2[int(off[r]) + i // 8]) >> (2 * (i % 8))) & 3 == "ACGT".index(c), (r, i)


def test_pack_seq2_known_cases():
    seqs = ["ACGTACGTAC", "ACGNN", "TTTTTTTTT", "", "ACGTACGTACGTACGTA", "=ACG", "A", "ACGTACGT", "ACGTACGTN", "GGGGGGGGGGGGGGGG"]
    for pad in (0x00, 0xEE, 0x11, 0xFF):  # whatever lies behind a read's last base must not matter
        seq4, off = slots(seqs, pad)
        check(seqs, [len(s) for s in seqs], seq4, off)


def test_pack_seq2_fewer_bases_than_l_qseq():
    seqs = ["ACGTACGT", "ACGT"]
    seq4, off = slots(seqs, 0)
    s2, sx = pack_seq2(seq4, off, [30, 4])  # the first record says 30 bases and carries 8
    assert int(sx[0]) & 3 == 1


def test_pack_seq2_random_reads():
    rng = np.random.default_rng(5)
    seqs = []
    for _ in range(700):
        n = int(rng.integers(0, 70))
        p_bad = float(rng.choice([0.0, 0.0, 0.02, 0.3]))
        s = "".join(rng.choice(list("ACGT"), size=n))
        s = "".join(c if rng.random() >= p_bad else str(rng.choice(list(NT16))) for c in s)
        seqs.append(s)
    seq4, off = slots(seqs, int(rng.integers(0, 256)))
    check(seqs, [len(s) for s in seqs], seq4, off)


def test_synth_batches_carry_the_same_packing():
    from portcullis_amd import synth

    d = synth.generate(synth.CONFIGS["C2-tiny"])
    rb = synth.batch_to_numpy(d["batch"])
    s2, sx = pack_seq2(rb.seq4, rb.seq_off, rb.l_qseq)
    t2 = d["batch"]["seq2"].numpy().view(np.uint16)
    assert len(t2) % 2 == 0 and np.array_equal(t2[: len(s2)], s2) and not t2[len(s2):].any()
    assert np.array_equal(d["batch"]["seq_exc"].numpy().view(np.uint32), sx)
    spliced = np.diff(rb.seq_off.astype(np.int64)) > 0
    bits = np.unpackbits(sx.view(np.uint8), bitorder="little")[: rb.n].astype(bool)
    assert not (bits & spliced).any() and (bits | spliced).all()  # generated reads are pure ACGT; records without bases are marked
