"""Shared test plumbing: paired scene construction (product host layer + oracle), the CPU simulator of the device
functions (tests/hostsim), ray-batch generation and comparison helpers."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from raytracer_amd import HostBuilder, scenes  # noqa: E402
from raytracer_amd import _lib  # noqa: E402
from raytracer_amd.builder import camera_new as host_camera_new  # noqa: E402
from oracle import oracle as orc  # noqa: E402

SMALL = {  # config -> keyword overrides that keep CPU tests quick
    "C1": dict(width=64, spp=2),
    "C2": dict(width=48, spp=2),
    "C3": dict(width=48, spp=2),
    "C4": dict(width=48, spp=2),
    "C5": dict(width=48, spp=2),
}


def build_pair(config, **kw):
    """The same scene described to the product's host layer and to the oracle."""
    hs = scenes.build(config, HostBuilder(), **kw)
    os_ = scenes.build(config, orc.OracleBuilder(), **kw)
    return hs, os_


def cameras(spec):
    """(rt_camera from the product host layer, orc_camera from the oracle) for a SceneSpec."""
    hc = host_camera_new(image_width=spec.width, **spec.camera)
    oc = orc.camera_new(image_width=spec.width, **spec.camera)
    return hc, oc


class sim_scene(C.Structure):
    _fields_ = [(n, C.c_void_p) if n not in ("n_nodes", "n_quads") else (n, C.c_uint64) for n in
                ["nodes", "n_nodes", "boxes", "spheres", "quads", "tris", "quadrics", "sphere_mat", "sphere_id",
                 "quad_mat", "quad_id", "tri_mat", "tri_id", "quadric_mat", "quadric_id", "media", "ops", "chains",
                 "mats", "texs", "texels", "image_desc", "perlin_vec", "perlin_perm", "cull_boxes", "n_quads"]]


_SIM = None


def _sim_lib():
    global _SIM
    if _SIM is None:
        so = os.path.join(ROOT, "tests", "hostsim", "libhostsim.so")
        src = os.path.join(ROOT, "tests", "hostsim", "hostsim.cpp")
        deps = [src] + [os.path.join(ROOT, *p) for p in (("raytracer_amd", "csrc", "rt_core.h"),
                                                         ("raytracer_amd", "csrc", "rt_records.h"),
                                                         ("raytracer_amd", "csrc", "rt_wide.h"),
                                                         ("include", "rt_amd.h"))]
        if not os.path.exists(so) or os.path.getmtime(so) <= max(os.path.getmtime(d) for d in deps):
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-host-only", "-x", "hip", "-O2", "-std=c++17", "-fPIC",
                                   "-shared", "-Wl,-Bsymbolic", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-o", so, src])
        _SIM = C.CDLL(so)
    return _SIM


class HostSim:
    """Runs rt_core.h's walk / finalize / scatter on the CPU over a HostBuilder's flattened arrays."""

    ORDER = {"nodes": 0, "boxes": 1, "spheres": 2, "quads": 3, "tris": 4, "quadrics": 5, "media": 6, "ops": 7,
             "chains": 8, "mats": 9, "texs": 10, "sphere_id": 11, "quad_id": 12, "tri_id": 13, "quadric_id": 14,
             "sphere_mat": 15, "quad_mat": 16, "tri_mat": 17, "quadric_mat": 18, "texels": 19, "image_desc": 20,
             "perlin_vec": 21, "perlin_perm": 22, "cull_boxes": 23}

    def __init__(self, builder):
        self.builder = builder  # keeps the arrays alive
        self.sc = sim_scene()
        for name, what in self.ORDER.items():
            p, n = builder.flat_ptr(what)
            setattr(self.sc, name, p)
            if name == "nodes":
                self.sc.n_nodes = n
            if name == "quads":
                self.sc.n_quads = n
        self.L = _sim_lib()

    def intersect(self, o, d, tmin, tmax, rng3=None, seed=42):
        o = np.ascontiguousarray(o, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(d, dtype=np.float64).reshape(-1, 3)
        n = o.shape[0]
        tmin = np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, dtype=np.float64), (n,)))
        tmax = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, dtype=np.float64), (n,)))
        if rng3 is not None:
            rng3 = np.ascontiguousarray(rng3, dtype=np.uint32).reshape(n, 3)
        out = {"hit": np.zeros(n, np.uint8), "prim": np.zeros(n, np.uint32), "t": np.zeros(n), "p": np.zeros((n, 3)),
               "n": np.zeros((n, 3)), "uv": np.zeros((n, 2)), "front": np.zeros(n, np.uint8),
               "ctr": np.zeros(n, np.uint32)}
        tl = (C.c_uint64 * 3)()
        vp = C.c_void_p
        self.L.sim_intersect(C.byref(self.sc), C.c_uint64(seed), vp(o.ctypes.data), vp(d.ctypes.data),
                             vp(tmin.ctypes.data), vp(tmax.ctypes.data),
                             vp(rng3.ctypes.data) if rng3 is not None else None, C.c_uint64(n),
                             vp(out["hit"].ctypes.data), vp(out["prim"].ctypes.data), vp(out["t"].ctypes.data),
                             vp(out["p"].ctypes.data), vp(out["n"].ctypes.data), vp(out["uv"].ctypes.data),
                             vp(out["front"].ctypes.data), vp(out["ctr"].ctypes.data), tl)
        out["counters"] = {"node_tests": tl[0], "prim_tests": tl[1], "cull_tests": tl[2]}
        return out

    def set_cull(self, on):
        """on=False makes every cull box infinite (RT_FLAG_NO_CULL): the tests counted are then the reference's."""
        self.L.sim_set_cull(1 if on else 0)

    def trace_samples(self, cam, pixel, sample, max_depth, background, seed=42):
        pixel = np.ascontiguousarray(pixel, dtype=np.uint32)
        sample = np.ascontiguousarray(sample, dtype=np.uint32)
        n = pixel.size
        rad, nr = np.zeros((n, 3)), np.zeros(n, np.uint32)
        bg = (C.c_double * 3)(*[float(x) for x in background])
        vp = C.c_void_p
        self.L.sim_trace_samples(C.byref(self.sc), C.byref(cam), C.c_uint64(seed), C.c_uint32(max_depth), bg,
                                 C.c_uint64(n), vp(pixel.ctypes.data), vp(sample.ctypes.data), vp(rad.ctypes.data),
                                 vp(nr.ctypes.data))
        return rad, nr


def path_rays(ospec, ocam, n, bounce, seed=42, rng=None):
    """Rays entering bounce `bounce` (0 = camera rays) of n random (pixel, sample) pairs, captured from the oracle,
    with the random-stream position (pixel, sample, draw index) each ray's traversal starts from."""
    rs = np.random.RandomState(1234 + bounce if rng is None else rng)
    npix = ocam.width * ocam.height
    pixel = rs.randint(0, npix, size=n).astype(np.uint32)
    sample = rs.randint(0, max(ospec.spp, 1), size=n).astype(np.uint32)
    _, _, cap = ospec.builder.trace_samples(ocam, pixel, sample, ospec.max_depth, ospec.background, seed=seed,
                                            capture_bounce=bounce)
    ok = cap["valid"].astype(bool)
    rng3 = np.stack([pixel, sample, cap["ctr"]], axis=1).astype(np.uint32)
    return cap["o"][ok], cap["d"][ok], rng3[ok]


def assert_hits_equal(got, want, exact=True, what="", uv_exact=True):
    """Bit-exact (or <= 1e-12 relative when exact=False) comparison of two intersect() results.  uv_exact=False
    allows a few ulp on (u, v): a sphere's come from acos/atan2, which differ between libm and the device library."""
    assert np.array_equal(got["hit"], want["hit"]), f"{what}: hit flags differ at {np.flatnonzero(got['hit'] != want['hit'])[:8]}"
    assert np.array_equal(got["prim"], want["prim"]), f"{what}: primitive ids differ at {np.flatnonzero(got['prim'] != want['prim'])[:8]}"
    assert np.array_equal(got["front"], want["front"]), f"{what}: front_face differs"
    assert np.array_equal(got["ctr"], want["ctr"]), f"{what}: random draw counts differ"
    for k in ("t", "p", "n", "uv"):
        if k == "uv" and not uv_exact:
            np.testing.assert_allclose(got[k], want[k], rtol=0, atol=1e-15, err_msg=f"{what}: uv")
            continue
        if exact:
            assert np.array_equal(got[k], want[k]), f"{what}: {k} not bit-identical (max abs diff {np.abs(got[k] - want[k]).max()})"
        else:
            np.testing.assert_allclose(got[k], want[k], rtol=1e-12, atol=1e-300, err_msg=f"{what}: {k}")


BOX_FLOOR_SIDE = 1.25
BOX_FLOOR_HEIGHTS = np.random.RandomState(21).uniform(0.5, 3.0, (9, 9))


def box_floor_scene(b):
    """A 9 x 9 floor of create_box lists of different heights that touch their neighbours (final_scene's ground in
    small), under a BVH, and a sphere above it."""
    side, heights = BOX_FLOOR_SIDE, BOX_FLOOR_HEIGHTS
    m = b.mat_lambertian(b.tex_solid(0.5, 0.5, 0.5))
    lst = b.list_new()
    for i in range(9):
        for j in range(9):
            x, z = -5.0 + i * side, -5.0 + j * side
            b.list_add(lst, b.slice(b.box((x, 0.0, z), (x + side, float(heights[i, j]), z + side), m)))
    world = b.list_new()
    b.list_add(world, b.bvh(lst))
    b.list_add(world, b.sphere((0.0, 30.0, 0.0), 2.0, m))
    b.set_world(b.slice(world))


def box_floor_rays(n=4000, seed=22):
    """Rays that look for trouble on box_floor_scene: aimed exactly at corners and at points on edges, along the axes
    and the space diagonals, from inside boxes, from top faces and corners, from far away; some with their own intervals.
    Returns o, d, tmin, tmax."""
    side, heights = BOX_FLOOR_SIDE, BOX_FLOOR_HEIGHTS
    rs = np.random.RandomState(seed)
    corners = np.array([[-5.0 + i * side, float(rs.choice([0.0, heights[min(i, 8), min(j, 8)]])), -5.0 + j * side]
                        for i in range(10) for j in range(10)])
    o = np.column_stack([rs.uniform(-7, 7, n), rs.uniform(0.2, 6.0, n), rs.uniform(-7, 7, n)])
    d = rs.normal(size=(n, 3))
    k = np.arange(n)
    aim = corners[rs.randint(0, len(corners), n)]
    d[:800] = aim[:800] - o[:800]                                           # exactly at a corner
    edge = aim[800:1400] + np.column_stack([rs.uniform(0, side, 600), np.zeros(600), np.zeros(600)])
    d[800:1400] = edge - o[800:1400]                                        # at a point on an edge along x
    ax = rs.randint(0, 3, 500)
    d[1400:1900] = 0.0
    d[k[1400:1900], ax] = rs.choice([-1.0, 1.0], 500)                       # along an axis
    o[1400:1650] = aim[1400:1650] + 0.0                                     # ... from a corner itself
    d[1900:2200] = rs.choice([-1.0, 1.0], (300, 3))                         # along a space diagonal
    o[2200:2700, 1] = rs.uniform(0.01, 0.45, 500)                           # from inside a box (every box is taller than 0.5)
    o[2200:2700, 0] = rs.uniform(-5, 6.25, 500)
    o[2200:2700, 2] = rs.uniform(-5, 6.25, 500)
    i_, j_ = rs.randint(0, 9, 300), rs.randint(0, 9, 300)
    o[2700:3000] = np.column_stack([-5.0 + i_ * side + rs.uniform(0, side, 300), heights[i_, j_], -5.0 + j_ * side + rs.uniform(0, side, 300)])
    d[2700:3000, 1] = np.abs(d[2700:3000, 1]) * rs.choice([1.0, 0.05], 300)  # leaving a top face, some nearly level
    o[3000:3300] = rs.uniform(-1, 1, (300, 3)) * 900.0 + np.array([0.0, 1000.0, 0.0])
    d[3000:3300] = corners[rs.randint(0, len(corners), 300)] - o[3000:3300]  # from far away at a corner
    tmin = np.full(n, 0.001)
    tmax = np.full(n, np.inf)
    tmin[3300:] = rs.uniform(0.0, 4.0, n - 3300)
    tmax[3300:] = tmin[3300:] + rs.uniform(0.0, 8.0, n - 3300)
    return o, d, tmin, tmax
