"""The eight-wide acceleration tree over the leaves of a reference BVH (raytracer_amd/csrc/rt_wide.h) and the walk in
stretches of top-level entries, compiled for the host by tests/hostsim and compared with the oracle bit for bit:
replacing the reference's inner nodes must change no hit, no primitive id and no random draw."""
import ctypes as C

import numpy as np
import pytest

import helpers
from helpers import SMALL, HostSim, assert_hits_equal, build_pair, cameras, path_rays

BVH_CONFIGS = {"C1": 1, "C4": 2, "C5": 2}   # config -> top-level BVH entries that must take a wide tree


@pytest.fixture(scope="module")
def pairs():
    cache = {}

    def get(cfg):
        if cfg not in cache:
            hs, os_ = build_pair(cfg, **SMALL[cfg])
            cache[cfg] = (hs, os_, HostSim(hs.builder))
        return cache[cfg]

    return get


def _stretch_mode():
    L = helpers._sim_lib()
    L.sim_wide_stretches.restype = C.c_ulonglong
    L.sim_wide_steps.restype = C.c_ulonglong
    return L


@pytest.mark.parametrize("cfg", sorted(BVH_CONFIGS))
def test_stretches_with_wide_trees_match_oracle(pairs, cfg):
    hs, os_, sim = pairs(cfg)
    L = _stretch_mode()
    _, oc = cameras(hs)
    L.sim_set_walk_mode(3)
    try:
        for bounce in (0, 1, 2):
            o, d, rng3 = path_rays(os_, oc, 3000, bounce)
            want = os_.builder.intersect(o, d, 0.001, np.inf, rng3)
            got = sim.intersect(o, d, 0.001, np.inf, rng3)
            assert L.sim_wide_stretches() == BVH_CONFIGS[cfg]
            assert L.sim_wide_steps() > 0
            assert_hits_equal(got, want, exact=True, what=f"{cfg} bounce {bounce}")
            assert got["counters"]["prim_tests"] <= want["counters"]["prim_tests"]
    finally:
        L.sim_set_walk_mode(0)


@pytest.mark.parametrize("cfg", sorted(BVH_CONFIGS))
def test_wide_trees_with_arbitrary_intervals_and_directions(pairs, cfg):
    """Random origins, directions with zero components (1/d infinite in every slab test) and random finite
    intervals, negative t included."""
    hs, os_, sim = pairs(cfg)
    L = _stretch_mode()
    rs = np.random.RandomState(23)
    n = 4000
    lo, hi = (-15.0, 15.0) if cfg == "C1" else (-50.0, 600.0)
    o = rs.uniform(lo, hi, (n, 3))
    d = rs.standard_normal((n, 3))
    d[::7, 0] = 0.0
    d[::11, 1] = 0.0
    d[::13, 2] = -0.0
    o[::14] = np.round(o[::14])   # origins on round coordinates with axis-parallel directions: 0 * inf in the slabs
    tmin = rs.uniform(-100.0, 10.0, n)
    tmax = tmin + rs.uniform(0.0, 2000.0, n)
    rng3 = np.stack([np.arange(n), np.full(n, 3), np.zeros(n)], axis=1).astype(np.uint32)
    want = os_.builder.intersect(o, d, tmin, tmax, rng3)
    L.sim_set_walk_mode(3)
    try:
        got = sim.intersect(o, d, tmin, tmax, rng3)
    finally:
        L.sim_set_walk_mode(0)
    assert_hits_equal(got, want, exact=True, what=cfg)


def test_wide_tree_over_mixed_leaves_and_fallback_for_other_trees():
    """A BVH of spheres, quads, triangles and boxes in one list under an instance chain takes a wide tree (leaves are
    runs of several kinds and box-shaped lists); a BVH whose objects are themselves instances or BVHs does not and
    keeps the entry-by-entry walk.  Either way the hits are the oracle's."""
    from oracle import oracle as orc
    from raytracer_amd import HostBuilder
    from raytracer_amd.philox import SceneRng

    def make(b):
        rng = SceneRng(321)
        m = b.mat_lambertian(b.tex_solid(0.5, 0.5, 0.5))
        lst = b.list_new()
        for k in range(150):
            c = rng.vector(-10.0, 10.0)
            which = k % 4
            if which == 0:
                b.list_add(lst, b.sphere(c, rng.range(0.3, 1.5), m))
            elif which == 1:
                b.list_add(lst, b.quad(c, rng.vector(-2, 2), rng.vector(-2, 2), m))
            elif which == 2:
                b.list_add(lst, b.triangle(c, rng.vector(-2, 2), rng.vector(-2, 2), m))
            else:
                b.list_add(lst, b.slice(b.box(c, (c[0] + 1.0, c[1] + 2.0, c[2] + 0.5), m)))
        inst = b.translate(b.rotate_y(b.bvh(lst), 33.0), (-3.0, 0.0, 4.0))
        nested = b.list_new()   # objects of this BVH are translated spheres: chains change below the root
        for k in range(40):
            b.list_add(nested, b.translate(b.sphere(rng.vector(-8.0, 8.0), 0.8, m), (0.5, 0.0, 0.0)))
        tris = b.list_new()
        for k in range(90):
            b.list_add(tris, b.triangle(rng.vector(-9.0, 9.0), rng.vector(-2, 2), rng.vector(-2, 2), m))
        world = b.list_new()
        b.list_add(world, b.quad((-30.0, -12.0, -30.0), (60.0, 0.0, 0.0), (0.0, 0.0, 60.0), m))
        b.list_add(world, inst)
        b.list_add(world, b.bvh(nested))
        b.list_add(world, b.sphere((0.0, 0.0, 0.0), 1.0, m))
        b.list_add(world, b.bvh(tris))
        b.set_world(b.slice(world))

    hb, ob = HostBuilder(), orc.OracleBuilder()
    make(hb)
    make(ob)
    sim = HostSim(hb)
    rs = np.random.RandomState(4)
    n = 6000
    o = rs.uniform(-25, 25, (n, 3))
    d = -o + rs.uniform(-8, 8, (n, 3))
    rng3 = np.stack([np.arange(n), np.zeros(n), np.zeros(n)], axis=1).astype(np.uint32)
    want = ob.intersect(o, d, 0.001, np.inf, rng3)
    assert want["hit"].mean() > 0.5
    L = _stretch_mode()
    L.sim_set_walk_mode(3)
    try:
        got = sim.intersect(o, d, 0.001, np.inf, rng3)
        # the mixed BVH (a leaf's items of another kind start a level-1 node of their own) and the triangle BVH; the
        # one over instances keeps the entry-by-entry walk
        assert L.sim_wide_stretches() == 2
    finally:
        L.sim_set_walk_mode(0)
    assert_hits_equal(got, want, exact=True, what="mixed scene")


@pytest.mark.parametrize("cfg", sorted(BVH_CONFIGS))
def test_nearest_first_order_finds_the_same_hits_with_fewer_tests(pairs, cfg):
    """RT_FLAG_FAST_BVH visits a wide tree's children nearest first.  The closest hit is the same wherever no two
    primitives are hit at exactly the same parameter (on these scenes: everywhere, bar the shared faces of the floor
    boxes of final_scene, where either box may be reported); fewer primitives are tested on the way."""
    hs, os_, sim = pairs(cfg)
    L = _stretch_mode()
    _, oc = cameras(hs)
    o, d, rng3 = path_rays(os_, oc, 6000, 1)
    want = os_.builder.intersect(o, d, 0.001, np.inf, rng3)
    L.sim_set_walk_mode(3)
    try:
        plain = sim.intersect(o, d, 0.001, np.inf, rng3)
        L.sim_set_fast_bvh(1)
        fast = sim.intersect(o, d, 0.001, np.inf, rng3)
    finally:
        L.sim_set_fast_bvh(0)
        L.sim_set_walk_mode(0)
    assert_hits_equal(plain, want, exact=True, what=cfg)
    assert np.array_equal(fast["hit"], want["hit"]) and np.array_equal(fast["t"], want["t"])
    differing = (fast["prim"] != want["prim"]).mean()
    print(cfg, "prim ids differing in fast mode:", differing, "prim tests", fast["counters"]["prim_tests"], "vs", plain["counters"]["prim_tests"])
    assert differing <= (0.01 if cfg == "C5" else 0.0)
    assert fast["counters"]["prim_tests"] <= plain["counters"]["prim_tests"]


@pytest.mark.parametrize("count, kinds, seed", [
    (1, "s", 1), (1, "t", 2), (2, "q", 3), (3, "t", 4), (8, "t", 5), (9, "t", 6), (16, "q", 7), (17, "t", 8), (33, "s", 9),
    (64, "b", 10), (65, "tq", 11), (130, "b", 12), (257, "sq", 13), (700, "t", 14), (300, "stqb", 15), (1200, "s", 16)])
def test_wide_trees_of_many_shapes_and_sizes(count, kinds, seed):
    """BVHs of 1 to 1200 objects of one or several kinds (s sphere, t triangle, q quad, b six-sided box) next to two plain
    objects: a tree of one node, leaves that fill a level-1 node exactly, by one too many, trees of several levels.
    Rays from inside, from far outside and along the axes (zero direction components); custom intervals."""
    from oracle import oracle as orc
    from raytracer_amd import HostBuilder
    from raytracer_amd.philox import SceneRng

    def make(b):
        rng = SceneRng(1000 + seed)
        m = b.mat_lambertian(b.tex_solid(0.5, 0.5, 0.5))
        lst = b.list_new()
        for k in range(count):
            c = rng.vector(-6.0, 6.0)
            kind = kinds[k % len(kinds)]
            if kind == "s":
                b.list_add(lst, b.sphere(c, rng.range(0.1, 0.9), m))
            elif kind == "t":
                b.list_add(lst, b.triangle(c, rng.vector(-1, 1), rng.vector(-1, 1), m))
            elif kind == "q":
                b.list_add(lst, b.quad(c, rng.vector(-1, 1), rng.vector(-1, 1), m))
            else:
                b.list_add(lst, b.slice(b.box(c, (c[0] + rng.range(0.1, 1.0), c[1] + rng.range(0.1, 1.0), c[2] + rng.range(0.1, 1.0)), m)))
        world = b.list_new()
        b.list_add(world, b.sphere((0.0, -1000.0, 0.0), 990.0, m))
        b.list_add(world, b.bvh(lst))
        b.list_add(world, b.quad((-8.0, -8.0, 9.0), (16.0, 0.0, 0.0), (0.0, 16.0, 0.0), m))
        b.set_world(b.slice(world))

    hb, ob = HostBuilder(), orc.OracleBuilder()
    make(hb)
    make(ob)
    sim = HostSim(hb)
    rs = np.random.RandomState(seed)
    n = 1500
    o = rs.uniform(-7, 7, (n, 3))
    d = rs.normal(size=(n, 3))
    o[500:800] = rs.uniform(-1, 1, (300, 3)) * 800.0          # far outside: the f32 box tests run out of precision
    d[500:800] = -o[500:800] + rs.uniform(-6, 6, (300, 3))
    axis = rs.randint(0, 3, 300)
    d[800:1100] = 0.0
    d[np.arange(800, 1100), axis] = rs.choice([-1.0, 1.0], 300)  # along an axis: two infinite reciprocals
    tmin = np.full(n, 0.001)
    tmax = np.full(n, np.inf)
    tmin[1100:] = rs.uniform(-3.0, 3.0, n - 1100)
    tmax[1100:] = tmin[1100:] + rs.uniform(0.0, 12.0, n - 1100)
    rng3 = np.stack([np.arange(n), np.zeros(n), np.zeros(n)], axis=1).astype(np.uint32)
    want = ob.intersect(o, d, tmin, tmax, rng3)
    L = _stretch_mode()
    L.sim_set_walk_mode(3)
    try:
        got = sim.intersect(o, d, tmin, tmax, rng3)
        assert L.sim_wide_stretches() == 1
        L.sim_set_fast_bvh(1)
        fast = sim.intersect(o, d, tmin, tmax, rng3)
    finally:
        L.sim_set_fast_bvh(0)
        L.sim_set_walk_mode(0)
    assert 0.2 < want["hit"].mean() < 0.98
    assert_hits_equal(got, want, exact=True, what=f"{count} x {kinds}")
    assert np.array_equal(fast["hit"], want["hit"])
    same = fast["prim"] == want["prim"]
    assert same.mean() > 0.995 and np.array_equal(fast["t"][same], want["t"][same])


def test_box_sides_are_chosen_without_changing_any_hit():
    """A BVH of create_box lists (floor boxes like final_scene's, touching their neighbours) walked through its wide
    tree: of a box's six quads the walk tests the pair(s) holding the sides a ray can enter (or, from inside, leave) by,
    and the others only if those give no hit.  Rays aimed exactly at corners and at points on edges, rays along the axes
    and along face diagonals, origins inside boxes, on faces and on corners, origins far away: every hit equals the
    oracle's bit for bit, and far fewer than six quads are tested per box visited."""
    from oracle import oracle as orc
    from raytracer_amd import HostBuilder

    from helpers import box_floor_rays, box_floor_scene
    hb, ob = HostBuilder(), orc.OracleBuilder()
    box_floor_scene(hb)
    box_floor_scene(ob)
    sim = HostSim(hb)
    o, d, tmin, tmax = box_floor_rays()
    n = len(o)
    rng3 = np.stack([np.arange(n), np.zeros(n), np.zeros(n)], axis=1).astype(np.uint32)
    want = ob.intersect(o, d, tmin, tmax, rng3)
    L = _stretch_mode()
    L.sim_set_walk_mode(3)
    try:
        got = sim.intersect(o, d, tmin, tmax, rng3)
        assert L.sim_wide_stretches() == 1
    finally:
        L.sim_set_walk_mode(0)
    assert 0.5 < want["hit"].mean() < 0.999
    assert_hits_equal(got, want, exact=True, what="boxes")
    # the entry-by-entry walk with its cull boxes (a ray that enters a leaf's box exactly where the closest hit so far
    # lies -- every ray aimed at a shared corner -- must still be offered the later boxes there: a later quad takes the tie)
    plain = sim.intersect(o, d, tmin, tmax, rng3)
    assert_hits_equal(plain, want, exact=True, what="boxes, entry by entry")
    assert got["counters"]["prim_tests"] < 0.2 * want["counters"]["prim_tests"]
