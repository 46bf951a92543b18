"""CPU: the hole-free tile maps of the trailing-update kernels, replayed on the host through
gpx_debug_tile_map (same index functions the kernels call, same XCD chunking): every owned
tile is enumerated exactly once and nothing else is.  No GPU needed — pure index arithmetic."""
import ctypes as C

import numpy as np
import pytest


def tile_map(gpx, kind, tm, tn=0, P=1, tpb=1, c=0):
    cap = int(tm) * int(max(tn, tm)) + 64
    out = np.empty((cap, 2), dtype=np.int32)
    count = C.c_int64(0)
    rc = gpx.gpx_debug_tile_map(kind, tm, tn, P, tpb, c, out.ctypes.data_as(C.POINTER(C.c_int32)), cap,
                                C.byref(count))
    assert rc == 0
    return out[:count.value]


@pytest.mark.parametrize("tm", [1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 100, 256, 504])
def test_triangle_map_is_a_bijection(gpx, tm):
    pairs = tile_map(gpx, 0, tm)
    want = {(i, j) for i in range(tm) for j in range(i + 1)}
    got = [tuple(p) for p in pairs.tolist()]
    assert len(got) == len(set(got)) == len(want) and set(got) == want


def staircase(tm, tn, P, tpb, c):
    return {(ti, tj) for ti in range(tm)
            for tj in range(min(((ti // tpb) * P + c) * tpb + ti % tpb, tn - 1) + 1)}


@pytest.mark.parametrize("tm,tn,P,tpb,c", [
    (8, 8, 1, 8, 0),            # one rank, nb = 1024: the plain triangle
    (64, 64, 1, 8, 0),
    (60, 500, 8, 4, 3),         # P = 8, nb = 512: ragged part spans several super-columns
    (32, 256, 8, 4, 0),
    (33, 129, 4, 2, 1),         # tile rows not a multiple of 8
    (16, 4, 2, 4, 0),           # the STRIP: only one block column wide
    (5, 3, 3, 1, 2),
    (256, 2040, 8, 8, 7),
    (256, 2048, 8, 8, 0),       # BASELINE.json configs[3] per rank: N = 262144, P = 8, nb = 1024
    (256, 2048, 8, 8, 8),       #   -> 256 local tile rows x 2048 tile columns, every offset class
    (248, 1984, 8, 8, 3),
    (256, 8, 8, 8, 5),          #   its STRIP (one block column)
])
def test_staircase_map_is_a_bijection(gpx, tm, tn, P, tpb, c):
    pairs = tile_map(gpx, 1, tm, tn, P, tpb, c)
    want = staircase(tm, tn, P, tpb, c)
    got = [tuple(p) for p in pairs.tolist()]
    assert len(got) == len(set(got)) == len(want) and set(got) == want


def test_staircase_map_random(gpx):
    rng = np.random.default_rng(5)
    for _ in range(60):
        tpb = int(rng.choice([1, 2, 4, 8, 16]))
        P = int(rng.integers(1, 9))
        nloc_blocks = int(rng.integers(1, 7))
        tm = nloc_blocks * tpb
        c = int(rng.integers(0, P + 1))
        full = ((nloc_blocks - 1) * P + c + 1) * tpb          # width that holds the whole staircase
        tn = int(rng.integers(1, full + 1))
        pairs = tile_map(gpx, 1, tm, tn, P, tpb, c)
        want = staircase(tm, tn, P, tpb, c)
        got = [tuple(p) for p in pairs.tolist()]
        assert len(got) == len(set(got)) == len(want) and set(got) == want, (tm, tn, P, tpb, c)


def dealt_staircase(tm, tn, P, tpb, r, lbf, gc0, snake):
    from gaussianprocesspathmodelling_amd import dist as gdist
    blocks = gdist.blocks_owned(r, 100000 // max(1, tpb) + 4 * P, P, snake)   # (far more blocks than any case asks for)
    want = set()
    for ti in range(tm):
        lim = (blocks[lbf + ti // tpb] - gc0) * tpb + ti % tpb
        want |= {(ti, tj) for tj in range(min(lim, tn - 1) + 1)}
    return want


def test_staircase_under_both_dealings(gpx):
    """Round 4: the staircase of a rank's trailing update under the cyclic and the snake dealing of row blocks
    (gpx_debug_stair_map: local tile row -> the row block's place in the dealing), against the Python mirror of the dealing —
    every owned tile once — and the per-panel balance of TILES over the ranks that the snake is for (P = 8, nb = 512, N = 65536)."""
    rng = np.random.default_rng(9)
    for case in range(80):
        snake = case % 2
        tpb = int(rng.choice([1, 2, 4, 8]))
        P = int(rng.integers(1, 9))
        r = int(rng.integers(0, P))
        from gaussianprocesspathmodelling_amd import dist as gdist
        own = gdist.blocks_owned(r, 40 * P, P, bool(snake))
        p = int(rng.integers(0, 6 * P))                      # panel being applied: columns from block gc0 = p + 2 on
        gc0 = p + 2
        lbf = gdist.lb0(p + 1, r, P, bool(snake))            # first own block beyond block p + 1
        nrows = int(rng.integers(1, 6))
        tm = nrows * tpb
        tn = int(rng.integers(1, (own[lbf + nrows - 1] - gc0 + 1) * tpb + 1))
        cap = tm * tn + 64
        out = np.empty((cap, 2), dtype=np.int32)
        count = C.c_int64(0)
        rc = gpx.gpx_debug_stair_map(tm, tn, P, tpb, r, lbf, gc0, snake, out.ctypes.data_as(C.POINTER(C.c_int32)), cap,
                                     C.byref(count))
        assert rc == 0, (tm, tn, P, tpb, r, lbf, gc0, snake)
        got = [tuple(q) for q in out[:count.value].tolist()]
        want = dealt_staircase(tm, tn, P, tpb, r, lbf, gc0, bool(snake))
        assert len(got) == len(set(got)) == len(want) and set(got) == want, (tm, tn, P, tpb, r, lbf, gc0, snake)

    def tiles_per_rank(snake, nblk=128, P=8, tpb=4):
        from gaussianprocesspathmodelling_amd import dist as gdist
        worst = mean = 0
        for p in range(0, nblk - 2, 9):
            per = []
            for r in range(P):
                lbf = gdist.lb0(p + 1, r, P, snake)
                nown = gdist.lb0(nblk - 1, r, P, snake) - lbf
                if nown <= 0:
                    per.append(0)
                    continue
                tm, tn = nown * tpb, (nblk - p - 2) * tpb
                cap = tm * tn + 64
                out = np.empty((cap, 2), dtype=np.int32)
                count = C.c_int64(0)
                assert gpx.gpx_debug_stair_map(tm, tn, P, tpb, r, lbf, p + 2, int(snake),
                                               out.ctypes.data_as(C.POINTER(C.c_int32)), cap, C.byref(count)) == 0
                per.append(count.value)
            worst += max(per)
            mean += sum(per) / P
        return worst / mean
    assert tiles_per_rank(True) < 1.02 < 1.07 < tiles_per_rank(False)


def test_first_super_tiles_are_compact(gpx):
    """The L2 argument of DESIGN.md §3.1: 64 consecutive launch slots of one XCD chunk touch at
    most 16 panel row blocks (8 rows + 8 columns of tiles) while super-tiles are full."""
    tm = 128
    pairs = tile_map(gpx, 0, tm)
    total = len(pairs)                       # no masked slots when tm is a multiple of 8
    chunk = total // 8                       # logical ids of XCD x: [x*chunk, (x+1)*chunk)
    logical = np.empty((total, 2), dtype=np.int64)
    q, r = divmod(total, 8)
    for b in range(total):                   # invert the XCD chunking: launch order -> logical order
        x = b & 7
        lin = (x * (q + 1) if x < r else r * (q + 1) + (x - r) * q) + (b >> 3)
        logical[lin] = pairs[b]
    first = logical[:64]
    assert len(set(first[:, 0])) == 8 and len(set(first[:, 1])) == 8


@pytest.mark.parametrize("tm,ts", [(504, 8), (16, 8), (9, 8), (100, 8), (33, 4), (24, 16), (8, 1), (7, 2), (65, 8)])
def test_fused_strip_plus_rest_map_is_a_bijection(gpx, tm, ts):
    """gemm_nt_fused_kernel (round 3): blocks [0, S) are the strip (first ts tile columns, masked to
    tj <= ti), the rest the triangle beyond it, each part with its own XCD chunking — together the lower
    triangle of the tm x tm tile grid, every tile once; and the strip's tiles all come before the rest's
    in launch order (the hardware dispatches in block order: that is what retires the strip first)."""
    pairs = tile_map(gpx, 2, tm, ts)
    want = {(i, j) for i in range(tm) for j in range(i + 1)}
    got = [tuple(p) for p in pairs.tolist()]
    assert len(got) == len(set(got)) == len(want) and set(got) == want
    is_strip = [j < ts for _, j in got]
    nstrip = sum(is_strip)
    assert all(is_strip[:nstrip]) and not any(is_strip[nstrip:])
