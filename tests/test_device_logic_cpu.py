"""The per-lane device functions of raytracer_amd/csrc/rt_core.h (tree walk in its three forms, hit-record
reconstruction, materials, textures, camera), compiled for the host by tests/hostsim and compared with the oracle
bit for bit.  This is a logic check that runs without a GPU; the GPU tests repeat it on the real kernels."""
import ctypes as C

import numpy as np
import pytest

import helpers
from raytracer_amd import _lib
from helpers import SMALL, HostSim, assert_hits_equal, build_pair, cameras, path_rays

CONFIGS = ["C1", "C2", "C3", "C4", "C5"]


@pytest.fixture(scope="module")
def pairs():
    cache = {}

    def get(cfg):
        if cfg not in cache:
            hs, os_ = build_pair(cfg, **SMALL[cfg])
            cache[cfg] = (hs, os_, HostSim(hs.builder))
        return cache[cfg]

    return get


@pytest.mark.parametrize("cfg", CONFIGS)
@pytest.mark.parametrize("mode", ["generic-uniform", "generic-divergent", "lockstep", "lane-state-machine"])
def test_walk_matches_oracle(pairs, cfg, mode):
    """Every form of the walk returns the oracle's hits with the cull boxes in use; with the cull boxes switched
    off it also performs exactly the oracle's box and primitive tests."""
    hs, os_, sim = pairs(cfg)
    if mode.startswith("lockstep") and cfg in ("C1", "C4", "C5"):
        pytest.skip("the lockstep walk is only used for trees without BVH entries")
    L = helpers._sim_lib()
    L.sim_set_walk_mode({"generic-uniform": 0, "generic-divergent": 0, "lockstep": 1, "lane-state-machine": 2}[mode])
    L.sim_set_uniform_mode(0 if mode == "generic-divergent" else 1)
    try:
        _, oc = cameras(hs)
        for bounce in (0, 2):
            o, d, rng3 = path_rays(os_, oc, 1500, bounce)
            want = os_.builder.intersect(o, d, 0.001, np.inf, rng3)
            got = sim.intersect(o, d, 0.001, np.inf, rng3)
            assert_hits_equal(got, want, exact=True, what=f"{cfg} {mode} bounce {bounce}")
            assert got["counters"]["node_tests"] <= want["counters"]["node_tests"]
            assert got["counters"]["prim_tests"] <= want["counters"]["prim_tests"]
            sim.set_cull(False)
            got = sim.intersect(o, d, 0.001, np.inf, rng3)
            assert_hits_equal(got, want, exact=True, what=f"{cfg} {mode} bounce {bounce}, no culling")
            assert got["counters"]["node_tests"] == want["counters"]["node_tests"]
            assert got["counters"]["prim_tests"] == want["counters"]["prim_tests"]
            sim.set_cull(True)
    finally:
        sim.set_cull(True)
        L.sim_set_walk_mode(0)
        L.sim_set_uniform_mode(1)


@pytest.mark.parametrize("cfg", CONFIGS)
def test_cull_boxes_save_work(pairs, cfg):
    """What the cull boxes are for: fewer primitive tests on the benchmark scenes (and never more)."""
    hs, os_, sim = pairs(cfg)
    _, oc = cameras(hs)
    o, d, rng3 = path_rays(os_, oc, 1500, 1)
    sim.set_cull(False)
    try:
        base = sim.intersect(o, d, 0.001, np.inf, rng3)["counters"]
    finally:
        sim.set_cull(True)
    with_cull = sim.intersect(o, d, 0.001, np.inf, rng3)["counters"]
    print(cfg, "no cull", base, "cull", with_cull)
    assert with_cull["prim_tests"] <= base["prim_tests"]
    if cfg in ("C4", "C5"):
        assert with_cull["prim_tests"] < 0.7 * base["prim_tests"]


def test_leaf_boxes_cut_triangle_tests(pairs):
    """BVH leaves of the two meshes in cornell_model carry a cull box per pair of triangles; the per-lane state
    machine tests the eight boxes of a leaf in one step and visits only the pairs whose box the ray can touch.  Same
    hits as the oracle bit for bit, well under half of the primitive tests, and with the cull boxes switched off
    exactly the oracle's."""
    hs, os_, sim = pairs("C4")
    L = helpers._sim_lib()
    _, oc = cameras(hs)
    o, d, rng3 = path_rays(os_, oc, 1500, 1)
    want = os_.builder.intersect(o, d, 0.001, np.inf, rng3)
    L.sim_set_walk_mode(2)
    try:
        got = sim.intersect(o, d, 0.001, np.inf, rng3)
        L.sim_set_leaf_boxes(0)
        plain = sim.intersect(o, d, 0.001, np.inf, rng3)
        L.sim_set_leaf_boxes(1)
        sim.set_cull(False)
        exact = sim.intersect(o, d, 0.001, np.inf, rng3)
    finally:
        sim.set_cull(True)
        L.sim_set_leaf_boxes(1)
        L.sim_set_walk_mode(0)
    for name, r in (("leaf boxes", got), ("no leaf boxes", plain), ("no culling", exact)):
        assert_hits_equal(r, want, exact=True, what=name)
    assert exact["counters"]["prim_tests"] == want["counters"]["prim_tests"]
    assert got["counters"]["prim_tests"] < 0.6 * plain["counters"]["prim_tests"] < want["counters"]["prim_tests"]


def test_ray_signatures_prune_the_lockstep_walk(pairs):
    """A lockstep wavefront whose rays all have a signature bit clear walks a copy of the tree without that
    entry's subtree (here: a "wavefront" of one ray).  Same hits as the oracle, and on cornell_smoke, whose two
    media are the signature entries, far fewer boundary tests than with the signatures left at all ones."""
    hs, os_, sim = pairs("C2")
    L = helpers._sim_lib()
    _, oc = cameras(hs)
    o, d, rng3 = path_rays(os_, oc, 1500, 1)
    want = os_.builder.intersect(o, d, 0.001, np.inf, rng3)
    L.sim_set_walk_mode(1)
    try:
        got = sim.intersect(o, d, 0.001, np.inf, rng3)
        L.sim_set_signatures(0)
        plain = sim.intersect(o, d, 0.001, np.inf, rng3)
    finally:
        L.sim_set_signatures(1)
        L.sim_set_walk_mode(0)
    assert_hits_equal(got, want, exact=True, what="with signatures")
    assert_hits_equal(plain, want, exact=True, what="without signatures")
    assert plain["counters"]["prim_tests"] == want["counters"]["prim_tests"]
    assert got["counters"]["prim_tests"] < 0.6 * plain["counters"]["prim_tests"]


@pytest.mark.parametrize("cfg", CONFIGS)
def test_arbitrary_intervals_and_directions(pairs, cfg):
    """Rays that do not come from the camera: random origins inside the scene's extent, random directions, random
    finite intervals (exercises negative t, rays starting inside boxes and spheres, axis-parallel directions)."""
    hs, os_, sim = pairs(cfg)
    rs = np.random.RandomState(17)
    n = 2000
    lo, hi = (-15.0, 15.0) if cfg == "C1" else (-50.0, 600.0)
    o = rs.uniform(lo, hi, (n, 3))
    d = rs.standard_normal((n, 3))
    d[::7, 0] = 0.0   # axis-parallel components: 1/d = inf in the slab test
    d[::11, 1] = 0.0
    tmin = rs.uniform(-100.0, 10.0, n)
    tmax = tmin + rs.uniform(0.0, 2000.0, n)
    rng3 = np.stack([np.arange(n), np.full(n, 3), np.zeros(n)], axis=1).astype(np.uint32)
    want = os_.builder.intersect(o, d, tmin, tmax, rng3)
    got = sim.intersect(o, d, tmin, tmax, rng3)
    assert_hits_equal(got, want, exact=True, what=cfg)


@pytest.mark.parametrize("cfg", CONFIGS)
def test_per_sample_radiance(pairs, cfg):
    hs, os_, sim = pairs(cfg)
    hc, oc = cameras(hs)
    rs = np.random.RandomState(2)
    n = 600
    pixel = rs.randint(0, hc.width * hc.height, n).astype(np.uint32)
    sample = rs.randint(0, 50, n).astype(np.uint32)
    want, want_rays = os_.builder.trace_samples(oc, pixel, sample, hs.max_depth, hs.background)
    got, got_rays = sim.trace_samples(hc, pixel, sample, hs.max_depth, hs.background)
    assert np.array_equal(got_rays, want_rays)
    # same libm on both sides here, so only the order of the throughput product differs (a0*(a1*(..)) in the
    # reference's recursion, ((a0*a1)*..) in the loop): a few ulp
    np.testing.assert_allclose(got, want, rtol=1e-13, atol=0)


def test_random_scene_with_every_primitive_and_instance(pairs):
    """A scene no benchmark has: all four primitive types in a BVH under nested Translate/RotateY, a medium whose
    boundary is itself a BVH of boxes, checker-of-checker and perlin textures."""
    from oracle import oracle as orc
    from raytracer_amd import HostBuilder
    from raytracer_amd.philox import SceneRng

    def make(b):
        rng = SceneRng(99)
        tex = b.tex_checker(1.5, b.tex_checker(0.3, b.tex_solid(1, 0, 0), b.tex_solid(0, 1, 0)), b.tex_perlin(0.7, [rng.range(-1, 1) for _ in range(768)]))
        mats = [b.mat_lambertian(tex), b.mat_metal(0.8, 0.7, 0.6, 0.3), b.mat_dielectric(1.4), b.mat_diffuse_light(b.tex_solid(4, 4, 4)), b.mat_isotropic(tex)]
        lst = b.list_new()
        for k in range(120):
            c = rng.vector(-10.0, 10.0)
            m = mats[k % 5]
            which = k % 4
            if which == 0:
                b.list_add(lst, b.sphere(c, rng.range(0.3, 1.5), m))
            elif which == 1:
                b.list_add(lst, b.quad(c, rng.vector(-2, 2), rng.vector(-2, 2), m))
            elif which == 2:
                b.list_add(lst, b.triangle(c, rng.vector(-2, 2), rng.vector(-2, 2), m))
            else:
                b.list_add(lst, b.slice(b.box(c, (c[0] + 1.0, c[1] + 2.0, c[2] + 0.5), m)))
        inst = b.translate(b.rotate_y(b.translate(b.bvh(lst), (1.0, -2.0, 0.5)), 33.0), (-3.0, 0.0, 4.0))
        fog_boxes = b.list_new()
        for k in range(20):
            c = rng.vector(-8.0, 8.0)
            b.list_add(fog_boxes, b.slice(b.box(c, (c[0] + 3.0, c[1] + 3.0, c[2] + 3.0), mats[0])))
        fog = b.constant_medium(b.rotate_y(b.bvh(fog_boxes), -20.0), 0.3, b.tex_solid(0.9, 0.9, 0.9))
        world = b.list_new()
        b.list_add(world, inst)
        b.list_add(world, b.y_cone((0.0, 6.0, 0.0), (3.0, 3.0, 3.0), mats[1]))
        b.list_add(world, fog)
        b.list_add(world, b.quadric_sphere((4.0, 1.0, -2.0), 2.0, mats[2]))
        b.set_world(b.slice(world))

    hb, ob = HostBuilder(), orc.OracleBuilder()
    make(hb)
    make(ob)
    sim = HostSim(hb)
    rs = np.random.RandomState(4)
    n = 4000
    o = rs.uniform(-25, 25, (n, 3))
    d = -o + rs.uniform(-6, 6, (n, 3))
    rng3 = np.stack([np.arange(n), np.zeros(n), np.zeros(n)], axis=1).astype(np.uint32)
    L = helpers._sim_lib()
    for mode in (0, 2):
        L.sim_set_walk_mode(mode)
        try:
            got = sim.intersect(o, d, 0.001, np.inf, rng3)
        finally:
            L.sim_set_walk_mode(0)
        want = ob.intersect(o, d, 0.001, np.inf, rng3)
        assert want["hit"].mean() > 0.3
        assert_hits_equal(got, want, exact=True, what=f"random scene mode {mode}")


def test_media_boundary_hits_through_edges_corners_and_tangents():
    """The one-pass scan for a medium's two boundary hits against the oracle's two hit() calls, on rays chosen to
    land two boundary hits within 0.0001 of each other (box edges and corners, sphere tangents, rays starting on a
    face), where the scan has to notice that it cannot decide and fall back to two passes."""
    from oracle import oracle as orc
    from raytracer_amd import HostBuilder

    def make(b):
        m = b.mat_lambertian(b.tex_solid(0.5, 0.5, 0.5))
        world = b.list_new()
        box = b.slice(b.box((0.0, 0.0, 0.0), (2.0, 3.0, 4.0), m))
        b.list_add(world, b.constant_medium(b.translate(b.rotate_y(box, 30.0), (1.0, 0.5, -2.0)), 0.8, b.tex_solid(1, 1, 1)))
        b.list_add(world, b.constant_medium(b.sphere((6.0, 1.0, 0.0), 1.5, m), 0.5, b.tex_solid(0, 0, 1)))
        two = b.list_new()   # a boundary of two overlapping spheres and a triangle: several candidates per ray
        b.list_add(two, b.sphere((-5.0, 0.0, 0.0), 1.0, m))
        b.list_add(two, b.sphere((-4.2, 0.0, 0.0), 1.0, m))
        b.list_add(two, b.triangle((-6.0, -1.0, 0.3), (3.0, 0.0, 0.0), (0.0, 2.5, 0.0), m))
        b.list_add(world, b.constant_medium(b.slice(two), 0.7, b.tex_solid(0, 1, 0)))
        b.list_add(world, b.quad((-20.0, -5.0, -20.0), (40.0, 0.0, 0.0), (0.0, 0.0, 40.0), m))
        b.set_world(b.slice(world))

    hb, ob = HostBuilder(), orc.OracleBuilder()
    make(hb)
    make(ob)
    sim = HostSim(hb)
    rs = np.random.RandomState(11)
    th = np.deg2rad(30.0)
    corners = []
    for x in (0.0, 2.0):
        for y in (0.0, 3.0):
            for z in (0.0, 4.0):  # object -> world: RotateY then Translate
                corners.append((np.cos(th) * x + np.sin(th) * z + 1.0, y + 0.5, -np.sin(th) * x + np.cos(th) * z - 2.0))
    corners = np.array(corners)
    targets = [corners]
    for a in range(8):
        for b_ in range(a + 1, 8):
            if np.isclose(np.abs(corners[a] - corners[b_]).sum(), 0) or (np.count_nonzero(np.abs(corners[a] - corners[b_]) > 1e-9) > 2 and False):
                continue
            f = rs.uniform(0, 1, (6, 1))
            targets.append(corners[a] * f + corners[b_] * (1 - f))   # points on edges and diagonals
    targets = np.concatenate(targets)
    # tangents to the spheres: points on the silhouette circle seen from the origin of each ray
    o_list, d_list = [], []
    for tgt in targets:
        for _ in range(6):
            o = rs.uniform(-12, 12, 3)
            o_list.append(o)
            d_list.append((tgt - o) * (1.0 + 1e-9 * rs.standard_normal()))
            o_list.append(o)
            d_list.append((tgt - o) + 1e-6 * rs.standard_normal(3))   # within 0.0001 of the edge
    for c, r in (((6.0, 1.0, 0.0), 1.5), ((-5.0, 0.0, 0.0), 1.0), ((-4.2, 0.0, 0.0), 1.0)):
        c = np.array(c)
        for _ in range(300):
            o = c + rs.standard_normal(3) * 6.0
            oc = c - o
            dist = np.linalg.norm(oc)
            if dist <= r * 1.01:
                continue
            # a direction on the tangent cone, nudged in or out by up to ~1e-7
            perp = np.cross(oc, rs.standard_normal(3))
            perp /= np.linalg.norm(perp)
            ang = np.arcsin(r / dist) + rs.choice([0.0, 1e-7, -1e-7, 3e-5, -3e-5])
            o_list.append(o)
            d_list.append(np.cos(ang) * oc / dist + np.sin(ang) * perp)
    # rays that start on a face or inside
    for _ in range(400):
        p = corners[0] * 0 + np.array([1.0, 0.5, -2.0]) + rs.uniform(0, 1, 3) * np.array([2.0, 3.0, 1.0])
        o_list.append(p)
        d_list.append(rs.standard_normal(3))
    o = np.array(o_list)
    d = np.array(d_list)
    n = len(o)
    rng3 = np.stack([np.arange(n), np.ones(n), np.zeros(n)], axis=1).astype(np.uint32)
    want = ob.intersect(o, d, 0.001, np.inf, rng3)
    assert (want["ctr"] > 0).mean() > 0.3     # many rays do cross a medium (a draw is made)
    L = helpers._sim_lib()
    L.sim_fallbacks.restype = C.c_ulonglong
    for mode in (0, 1, 2):
        L.sim_set_walk_mode(mode)
        before = L.sim_fallbacks()
        try:
            got = sim.intersect(o, d, 0.001, np.inf, rng3)
            fallbacks = L.sim_fallbacks() - before
            L.sim_set_one_pass(0)
            got_two = sim.intersect(o, d, 0.001, np.inf, rng3)
            assert L.sim_fallbacks() - before == fallbacks
            sim.set_cull(False)   # the reference's own test counts, whichever way the two hits are found
            L.sim_set_one_pass(1)
            got_nc = sim.intersect(o, d, 0.001, np.inf, rng3)
        finally:
            sim.set_cull(True)
            L.sim_set_one_pass(1)
            L.sim_set_walk_mode(0)
        assert_hits_equal(got, want, exact=True, what=f"one pass, mode {mode}")
        assert_hits_equal(got_two, want, exact=True, what=f"two passes, mode {mode}")
        assert_hits_equal(got_nc, want, exact=True, what=f"one pass without cull boxes, mode {mode}")
        assert got_nc["counters"]["prim_tests"] == want["counters"]["prim_tests"]
        assert 0 < fallbacks < 0.2 * n, fallbacks   # the undecidable cases occur here, and stay the exception


def test_state_machine_after_a_medium_whose_boundary_changes_chain():
    """A medium whose boundary sits in an instance chain, followed by a BVH in that same chain: the walk comes out of
    the medium in the boundary's ray space and goes straight into box tests there."""
    from oracle import oracle as orc
    from raytracer_amd import HostBuilder
    from raytracer_amd.philox import SceneRng

    def make(b):
        rng = SceneRng(5)
        m = b.mat_lambertian(b.tex_solid(0.5, 0.5, 0.5))
        off = (3.0, -1.0, 2.0)
        balls = b.list_new()
        for _ in range(60):
            b.list_add(balls, b.sphere(rng.vector(-6.0, 6.0), rng.range(0.2, 0.9), m))
        world = b.list_new()
        b.list_add(world, b.constant_medium(b.translate(b.sphere((0.0, 0.0, 0.0), 4.0, m), off), 0.4, b.tex_solid(1, 1, 1)))
        b.list_add(world, b.translate(b.bvh(balls), off))
        b.set_world(b.slice(world))

    hb, ob = HostBuilder(), orc.OracleBuilder()
    make(hb)
    make(ob)
    nodes = hb.flat("nodes")
    kinds = [_lib.NODE_KINDS[n.kind] for n in nodes]
    med, bvh = kinds.index("medium"), kinds.index("bvh")
    assert nodes[med].chain == 0 and nodes[med + 1].chain == nodes[bvh].chain != 0
    sim = HostSim(hb)
    rs = np.random.RandomState(8)
    n = 3000
    o = rs.uniform(-20, 20, (n, 3))
    d = (np.array([3.0, -1.0, 2.0]) - o) + rs.uniform(-5, 5, (n, 3))
    rng3 = np.stack([np.arange(n), np.zeros(n), np.zeros(n)], axis=1).astype(np.uint32)
    want = ob.intersect(o, d, 0.001, np.inf, rng3)
    L = helpers._sim_lib()
    for mode in (0, 2):
        L.sim_set_walk_mode(mode)
        try:
            got = sim.intersect(o, d, 0.001, np.inf, rng3)
        finally:
            L.sim_set_walk_mode(0)
        assert_hits_equal(got, want, exact=True, what=f"mode {mode}")
