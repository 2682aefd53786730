"""plan_groups (pjb_plan_groups through portcullis_amd/ffi.py): the chain plan of the program, of bench.py and of a multi-GPU rank -- host
arithmetic of the library, no GPU."""
import pytest


def test_grch38_gives_three_chains_of_about_a_gigabase():
    from portcullis_amd import ffi, synth

    lens = [c.contig_len for c in synth.c3_contig_configs()]
    groups = ffi.plan_groups(lens, list(range(len(lens))))
    assert [len(g) for g in groups] == [5, 7, 13]
    assert [t for g in groups for t in g] == list(range(25))                       # consecutive, in index order, nothing lost
    for g in groups:
        assert sum(((lens[t] + ffi.GROUP_GAP + 63) & ~63) for t in g) <= 1 << 30   # what pjb_finish_group_begin adds up
    seven = ffi.plan_groups(lens, list(range(25)), 1 << 29)
    assert [len(g) for g in seven] == [2, 2, 3, 3, 4, 7, 4]


@pytest.mark.parametrize("lens,tids,max_bases,want", [
    ([100] * 5, [0, 1, 2, 3, 4], 1 << 30, [[0, 1, 2, 3, 4]]),
    ([100] * 40, list(range(40)), 1 << 30, [list(range(32)), list(range(32, 40))]),        # PJB_GROUP_MAX members at most
    ([1 << 29, 1 << 29, 1 << 29], [0, 1, 2], 1 << 30, [[0], [1], [2]]),                      # the gap between members counts
    ([10, (1 << 31) - 1, 10], [0, 1, 2], 1 << 30, [[0], [1], [2]]),                         # a target longer than the limit stands alone
    ([5, 5, 5], [2, 0], 1 << 30, [[2, 0]]),                                                 # only the targets named, in the order given
    ([0, 0], [0, 1], 1 << 30, [[0, 1]]),                                                    # empty targets count one base
    # a set that would be ONE chain of more than 0.6 Gb goes as chains of at most 0.55 of its bases (two, or three when the targets do not
    # divide that way): the share of a rank of three or four
    ([300_000_000, 200_000_000, 150_000_000, 100_000_000], [0, 1, 2, 3], 1 << 30, [[0], [1, 2], [3]]),
    ([300_000_000, 200_000_000], [0, 1], 1 << 30, [[0, 1]]),                                # 0.5 Gb: one chain
    ([700_000_000], [0], 1 << 30, [[0]]),                                                   # a single target is never split
    ([], [], 1 << 30, []),
])
def test_edge_cases(lens, tids, max_bases, want):
    from portcullis_amd import ffi

    assert ffi.plan_groups(lens, tids, max_bases) == want
