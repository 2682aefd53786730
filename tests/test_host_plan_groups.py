"""plan_groups (portcullis_amd/ffi.py): the groups a caller hands to pjb_finish_group_begin -- host logic, no GPU."""
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
    ([10, 1 << 31, 10], [0, 1, 2], 1 << 30, [[0], [1], [2]]),                               # a target longer than the limit stands alone
    ([5, 5, 5], [2, 0], 1 << 30, [[2, 0]]),                                                 # only the targets named, in the order given
    ([0, 0], [0, 1], 1 << 30, [[0, 1]]),                                                    # empty targets count one base
])
def test_edge_cases(lens, tids, max_bases, want):
    from portcullis_amd import ffi

    assert ffi.plan_groups(lens, tids, max_bases) == want
