"""The placement protocol of k1_walk (portcullis_amd/csrc/pjb_kernels.hip.h: tile descriptors, group accumulators, group
descriptors that are upgraded from "pairs of the group" to "pairs up to and including the group", group prefixes) as a
state machine in Python, run under random schedules: every tile must learn exactly the number of pairs before it, whatever
the interleaving of the tiles' memory operations, with only a few tiles resident at a time (a tile holds its slot while it
waits, as a workgroup does) and tile numbers handed out by ticket.  No GPU needed; this pins the protocol, the GPU tests
(test_options_do_not_change_rows with "fused_k1") pin the kernel."""
import random

import pytest

VALID, PREFIX = 1 << 63, 1 << 62
VALUE = (1 << 40) - 1


def tile_program(t, n_tiles, pairs, G, W, mem, out):
    """One tile, as a generator that yields before every access to shared memory (one atomic step each)."""
    g, gi = divmod(t, G)
    tp = pairs[t]
    yield
    mem["tile_desc"][t] = VALID | tp
    yield
    old = mem["grp_acc"][g]
    mem["grp_acc"][g] = old + ((1 << 40) | tp)  # (one atomic add)
    gsize = min(G, n_tiles - g * G)
    if (old >> 40) + 1 == gsize:
        yield
        mem["grp_desc"][g] = VALID | (PREFIX if g == 0 else 0) | ((old & VALUE) + tp)
    if gi == 0:
        excl_g = 0
        gg = g - 1
        while gg >= 0:
            lanes = [gg - lane for lane in range(W)]
            seen = []
            for my in lanes:  # every lane polls its own word; the wave goes on when all have a valid one
                if my < 0:
                    seen.append(0)
                    continue
                while True:
                    yield
                    d = mem["grp_desc"][my]
                    if d & VALID:
                        break
                seen.append(d)
            stop = next((k for k, d in enumerate(seen) if d & PREFIX), W)
            excl_g += sum(d & VALUE for k, d in enumerate(seen) if k <= stop and lanes[k] >= 0)
            if stop < W:
                break
            gg -= W
        yield
        mem["grp_excl"][g] = VALID | excl_g
        if g > 0:
            yield
            mem["grp_desc"][g - 1] = VALID | PREFIX | excl_g
    else:
        while True:
            yield
            d = mem["grp_excl"][g]
            if d & VALID:
                break
        excl_g = d & VALUE
    v = 0
    for lane in range(gi):
        while True:
            yield
            d = mem["tile_desc"][g * G + lane]
            if d & VALID:
                break
        v += d & VALUE
    out[t] = excl_g + v


def run(n_tiles, G, W, resident, seed):
    rng = random.Random(seed)
    pairs = [rng.choice([0, 0, 1, 3, 300, 5000]) for _ in range(n_tiles)]
    ng = (n_tiles + G - 1) // G
    mem = {"tile_desc": [0] * n_tiles, "grp_acc": [0] * ng, "grp_desc": [0] * ng, "grp_excl": [0] * ng}
    out = [None] * n_tiles
    active, ticket, steps = [], 0, 0
    while ticket < n_tiles or active:
        while ticket < n_tiles and len(active) < resident and (not active or rng.random() < 0.5):
            active.append(tile_program(ticket, n_tiles, pairs, G, W, mem, out))  # tickets: tiles start in order
            ticket += 1
        k = rng.randrange(len(active))
        if rng.random() < 0.3:  # favour the youngest tile now and then: the worst case for waiting on older ones
            k = len(active) - 1
        try:
            next(active[k])
        except StopIteration:
            active.pop(k)
        steps += 1
        assert steps < 400 * n_tiles * max(W, G), "no progress: the protocol deadlocked under this schedule"
    want, acc = [], 0
    for p in pairs:
        want.append(acc)
        acc += p
    assert out == want


@pytest.mark.parametrize("n_tiles,G,W,resident", [(1, 4, 2, 1), (5, 4, 2, 2), (37, 4, 2, 3), (64, 4, 3, 5), (130, 8, 2, 7),
                                                  (200, 4, 64, 9), (257, 64, 64, 16), (96, 3, 1, 2)])
def test_every_tile_learns_its_prefix(n_tiles, G, W, resident):
    for seed in range(25):
        run(n_tiles, G, W, resident, seed)
