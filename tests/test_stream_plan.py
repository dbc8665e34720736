"""The persistent kernels' schedule (csrc/bsc_stream.h: whole rounds of tiles dealt round-robin, the
left-over tiles split along their units or run as one more round) property-tested on the host:
bsc_stream_plan exposes the C++ partition, the cursor walk below restates what StreamCursor does on
the device.  Every (tile, unit) must be visited exactly once, a workgroup must hold at most two
partial pieces (its two slab slots), and the pieces of a split tile must tile its units in order."""
import ctypes

import numpy as np
import pytest

from bayesic_amd import _ffi


def plan(tiles, n_kt, slots):
    out = (ctypes.c_int32 * 6)()
    _ffi.check(_ffi.load_library().bsc_stream_plan(tiles, n_kt, slots, out), "bsc_stream_plan")
    return dict(zip(["n_wg", "rounds", "tail_tiles", "sk_stream", "sk_q", "sk_r"], list(out)))


def first_unit(p, n_kt, w):
    if p["sk_stream"]:
        return w * p["sk_q"] + min(w, p["sk_r"])
    return min(w, p["tail_tiles"]) * n_kt


def walk(p, n_kt, w):
    """[(tile, first unit, one past the last)] of workgroup w, in its order."""
    segs = [(r * p["n_wg"] + w, 0, n_kt) for r in range(p["rounds"])]
    u0, u1 = first_unit(p, n_kt, w), first_unit(p, n_kt, w + 1)
    base = p["rounds"] * p["n_wg"]
    u = u0
    while u < u1:
        t, kt = divmod(u, n_kt)
        take = min(n_kt - kt, u1 - u)
        segs.append((base + t, kt, kt + take))
        u += take
    return segs


CASES = [(1, 1, 512), (4, 31250, 512), (1024, 128, 512), (1089, 132, 512), (782, 196, 512), (38318, 4, 512),
         (513, 7, 512), (511, 1000, 512), (600, 5, 512), (3, 2, 8), (17, 9, 8), (100, 1, 16), (9, 300, 8), (16, 6, 8)]


@pytest.mark.parametrize("tiles,n_kt,slots", CASES)
def test_every_unit_is_visited_once_and_pieces_fit_the_slab(tiles, n_kt, slots):
    p = plan(tiles, n_kt, slots)
    assert 1 <= p["n_wg"] <= slots
    assert p["rounds"] * p["n_wg"] + p["tail_tiles"] == tiles
    seen = np.zeros((tiles, n_kt), np.int32)
    pieces = {}
    for w in range(p["n_wg"]):
        partial = 0
        for t, a, b in walk(p, n_kt, w):
            assert 0 <= t < tiles and 0 <= a < b <= n_kt
            seen[t, a:b] += 1
            if (a, b) != (0, n_kt):
                partial += 1
                pieces.setdefault(t, []).append((w, a, b))
        assert partial <= 2, (w, partial)                       # two slab slots per workgroup
    assert (seen == 1).all()
    for t, ps in pieces.items():                                # a split tile: consecutive workgroups, in unit order
        ps.sort()
        assert ps[0][1] == 0 and ps[-1][2] == n_kt
        assert all(x[2] == y[1] for x, y in zip(ps, ps[1:]))
        assert [x[0] for x in ps] == list(range(ps[0][0], ps[0][0] + len(ps)))
    if not p["sk_stream"]:
        assert not pieces


def test_random_shapes():
    rs = np.random.RandomState(0)
    for _ in range(300):
        slots = int(rs.choice([8, 16, 64, 512]))
        tiles = int(rs.randint(1, 5 * slots))
        n_kt = int(rs.choice([1, 2, 3, 5, 8, 33, 500]))
        p = plan(tiles, n_kt, slots)
        total = sum(b - a for w in range(p["n_wg"]) for _, a, b in walk(p, n_kt, w))
        assert total == tiles * n_kt
        loads = [sum(b - a for _, a, b in walk(p, n_kt, w)) for w in range(p["n_wg"])]
        if p["sk_stream"]:
            assert max(loads) - min(loads) <= 1                 # the tail is dealt evenly
