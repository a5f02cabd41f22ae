"""GPU parity tests: the HIP path, called through the C ABI of include/rt_amd.h, against the CPU oracle on the same
seeded inputs.  Bar (BASELINE.json north_star): hit flag and primitive id exact, t and normal within 1e-5 relative;
these tests hold the stricter bar the design aims for -- bit-identical t / p / normal / uv wherever no
transcendental function is involved."""
import numpy as np
import pytest

from helpers import SMALL, assert_hits_equal, build_pair, cameras, path_rays

pytestmark = pytest.mark.gpu

CONFIGS = ["C1", "C2", "C3", "C4", "C5"]


@pytest.fixture(scope="module")
def gpu_scenes():
    cache = {}

    def get(cfg):
        if cfg not in cache:
            hs, os_ = build_pair(cfg, **SMALL[cfg])
            cache[cfg] = (hs, os_, hs.builder.commit(0))
        return cache[cfg]

    yield get
    for _, _, sc in cache.values():
        sc.close()


def test_device_arithmetic_is_ieee():
    """a/b, sqrt(a) and 1/a must be correctly rounded on the device: every accept/reject branch of the
    intersection code depends on it.  log() only has to be close (it feeds the medium's free-flight distance)."""
    from raytracer_amd import debug_arith
    rs = np.random.RandomState(7)
    n = 1 << 20
    a = np.abs(rs.standard_normal(n)) * 10.0 ** rs.uniform(-8, 8, n) + 1e-300
    b = rs.standard_normal(n) * 10.0 ** rs.uniform(-8, 8, n)
    b[b == 0] = 1.0
    q, r, c, l = debug_arith(a, b)
    assert np.array_equal(q, a / b)
    assert np.array_equal(r, np.sqrt(a))
    assert np.array_equal(c, 1.0 / a)
    ref = np.log(a)
    assert np.max(np.abs(l - ref) / np.maximum(np.abs(ref), 1e-300)) < 1e-14


@pytest.mark.parametrize("cfg", CONFIGS)
@pytest.mark.parametrize("bounce", [0, 1, 3])
def test_intersect_batch_matches_oracle(gpu_scenes, cfg, bounce):
    hs, os_, scene = gpu_scenes(cfg)
    _, oc = cameras(hs)
    o, d, rng3 = path_rays(os_, oc, 20000, bounce)
    want = os_.builder.intersect(o, d, 0.001, np.inf, rng3)
    got = scene.intersect(o, d, 0.001, np.inf, rng3)
    has_media = cfg in ("C2", "C5")
    if not has_media:
        assert_hits_equal(got, want, exact=True, what=f"{cfg} bounce {bounce}", uv_exact=(cfg != "C1"))
    else:
        # a medium hit's t comes from log(U); everything that is not a medium hit must still be bit-identical
        assert np.array_equal(got["hit"], want["hit"])
        assert np.array_equal(got["prim"], want["prim"])
        assert np.array_equal(got["ctr"], want["ctr"])
        assert np.array_equal(got["front"], want["front"])
        medium_ids = _medium_prim_ids(hs)
        is_medium = np.isin(want["prim"], medium_ids)
        for k in ("t", "p", "n"):
            assert np.array_equal(got[k][~is_medium], want[k][~is_medium]), k
            np.testing.assert_allclose(got[k][is_medium], want[k][is_medium], rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(got["uv"], want["uv"], rtol=0, atol=1e-15)  # sphere uv: acos/atan2


def _medium_prim_ids(hspec):
    return np.array([m.prim_id for m in hspec.builder.flat("media")], dtype=np.uint32)


@pytest.mark.parametrize("cfg", CONFIGS)
def test_per_sample_radiance_matches_oracle(gpu_scenes, cfg):
    hs, os_, scene = gpu_scenes(cfg)
    hc, oc = cameras(hs)
    rs = np.random.RandomState(11)
    n = 4000
    pixel = rs.randint(0, hc.width * hc.height, n).astype(np.uint32)
    sample = rs.randint(0, 64, n).astype(np.uint32)
    want, want_rays = os_.builder.trace_samples(oc, pixel, sample, hs.max_depth, hs.background)
    got, got_rays = scene.trace_samples(hc, pixel, sample, hs.max_depth, hs.background)
    same_len = got_rays == want_rays
    # paths may part ways only through last-bit differences of log/sin/pow; that has to be very rare
    assert same_len.mean() > 0.995, f"{(~same_len).sum()} of {n} paths took a different number of bounces"
    np.testing.assert_allclose(got[same_len], want[same_len], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("cfg", CONFIGS)
@pytest.mark.parametrize("schedule", ["queues", "queues-split", "whole-paths"])
def test_film_equals_sum_of_samples(gpu_scenes, cfg, schedule):
    """rt_render must give, bit for bit, the in-order sum of the per-sample radiances rt_trace_samples computes with
    one lane per whole path -- on every schedule: the queue-based one (queues + compaction + film accumulation; for
    scenes without BVH entries with queues sorted by ray signature and one kernel per bounce or, on request, a
    traverse and a shade launch) and whole paths in one kernel (on request, for scenes without BVH entries)."""
    hs, os_, scene = gpu_scenes(cfg)
    hc, _ = cameras(hs)
    spp = 3
    film, stats = scene.render(hc, spp, hs.max_depth, hs.background, spp_begin=5, batch_paths=2 * hc.width * hc.height,
                               whole_paths=schedule == "whole-paths",  # forced either way
                               split_bounce=schedule == "queues-split")
    no_bvh = cfg in ("C2", "C3")
    assert (stats["path_launches"] > 0) == (schedule == "whole-paths" and no_bvh)
    assert (stats["bounce_launches"] > 0) == (schedule == "queues" and no_bvh)
    assert stats["finish_launches"] == (1 if (schedule == "queues" or not no_bvh) else 0)  # the tail in one launch
    assert sum(stats[k] > 0 for k in ("traverse_launches", "path_launches", "bounce_launches")) == 1
    npix = hc.width * hc.height
    pixel = np.tile(np.arange(npix, dtype=np.uint32), spp)
    sample = np.repeat(np.arange(5, 5 + spp, dtype=np.uint32), npix)
    rad, nrays = scene.trace_samples(hc, pixel, sample, hs.max_depth, hs.background)
    acc = np.full((npix, 3), -0.0)
    for s in range(spp):
        acc = acc + rad[s * npix:(s + 1) * npix]
    assert np.array_equal(film.reshape(npix, 3), acc)
    assert stats["paths"] == npix * spp
    assert stats["rays"] == int(nrays.sum())


@pytest.mark.parametrize("cfg", ["C2", "C3"])
def test_film_matches_oracle_render(gpu_scenes, cfg):
    hs, os_, scene = gpu_scenes(cfg)
    hc, oc = cameras(hs)
    spp = 4
    film, _ = scene.render(hc, spp, hs.max_depth, hs.background)
    want, _ = os_.builder.render(oc, spp, hs.max_depth, hs.background)
    close = np.isclose(film, want, rtol=1e-9, atol=1e-12).all(axis=2)
    assert close.mean() > 0.995
    from oracle import oracle as orc
    rgb = scene.tonemap(film, spp)
    rgb_want = orc.tonemap(want, spp)
    assert (np.abs(rgb.astype(int) - rgb_want.astype(int)).max(axis=2) <= 1)[close].all()
    assert np.array_equal(scene.tonemap(want, spp), rgb_want)  # the tonemap kernel itself is exact


def test_counters_match_oracle(gpu_scenes):
    hs, os_, scene = gpu_scenes("C4")
    hc, oc = cameras(hs)
    # camera rays only: no transcendental involved.  With the cull boxes off the kernels do the reference's work
    # test for test; with them on, the same film from less work.
    film, st = scene.render(hc, 1, 1, hs.background, counters=True, cull=False)
    ofilm, ost = os_.builder.render(oc, 1, 1, hs.background)
    assert st["rays"] == ost["rays"]
    assert st["node_tests"] == ost["node_tests"]
    assert st["prim_tests"] == ost["prim_tests"]
    film_c, st_c = scene.render(hc, 1, 1, hs.background, counters=True)
    assert np.array_equal(film_c, film) and np.array_equal(film, ofilm)
    assert st_c["prim_tests"] < 0.7 * st["prim_tests"] and st_c["node_tests"] < st["node_tests"]


@pytest.mark.parametrize("cfg", CONFIGS)
def test_cull_boxes_do_not_change_the_film(gpu_scenes, cfg):
    hs, _, scene = gpu_scenes(cfg)
    hc, _ = cameras(hs)
    film, _ = scene.render(hc, 3, hs.max_depth, hs.background)
    film_nc, _ = scene.render(hc, 3, hs.max_depth, hs.background, cull=False)
    assert np.array_equal(film, film_nc)


@pytest.mark.parametrize("cfg", CONFIGS)
def test_golden_ray_batches(gpu_scenes, cfg):
    """The committed fixtures (tests/golden/oracle_rays_*.npz, made by tools/make_golden.py from the oracle)."""
    import os
    from helpers import ROOT
    g = np.load(os.path.join(ROOT, "tests", "golden", f"oracle_rays_{cfg}.npz"))
    hs, _, scene = gpu_scenes(cfg)
    medium_ids = _medium_prim_ids(hs)
    for bounce in (0, 1, 3):
        r = scene.intersect(g[f"o{bounce}"], g[f"d{bounce}"], 0.001, np.inf, g[f"rng{bounce}"])
        for k in ("hit", "prim", "front", "ctr"):
            assert np.array_equal(r[k], g[f"{k}{bounce}"]), (cfg, bounce, k)
        solid = ~np.isin(g[f"prim{bounce}"], medium_ids)
        for k in ("t", "p", "n"):
            assert np.array_equal(r[k][solid], g[f"{k}{bounce}"][solid]), (cfg, bounce, k)
            np.testing.assert_allclose(r[k], g[f"{k}{bounce}"], rtol=1e-12, atol=1e-9)


def test_render_agrees_with_reference_image():
    """cornell_quadric at the settings of the reference's own render (600 px, 200 spp, depth 50; main.rs:851-852)
    against the block means of images/cornell_box_two_cylinders_medium_quality.png kept in tests/golden."""
    import os
    from helpers import ROOT
    from raytracer_amd import HostBuilder, camera_new, scenes
    ref = np.load(os.path.join(ROOT, "tests", "golden", "reference_cylinders_blockmeans.npz"))
    spec = scenes.build("C3", HostBuilder(), width=600)
    cam = camera_new(image_width=600, **spec.camera)
    scene = spec.builder.commit(0)
    film, _ = scene.render(cam, 200, 50, spec.background)
    rgb = scene.tonemap(film, 200).astype(np.float64)
    blocks = rgb.reshape(6, 100, 6, 100, 3).mean(axis=(1, 3))
    diff = np.abs(blocks - ref["blocks"])
    # both images are 200-spp estimates with independent noise; a 100x100 block averages 10^4 pixels
    assert diff.max() < 3.0, diff.max()
    assert diff.mean() < 1.0, diff.mean()
    assert np.abs(rgb.mean(axis=(0, 1)) - ref["mean"]).max() < 0.75
    assert abs((rgb == 255).mean() - float(ref["frac255"])) < 0.005
    scene.close()


def test_render_agrees_with_the_reference_models_image_above_the_models():
    """cornell_model at the settings of images/cornell_models_medium_quality.png (600 px, 200 spp, depth 50) against the
    block means of that image's top third (tests/golden/reference_models_blockmeans.npz): box, light and camera are
    C4's, the meshes below are the author's own (stand-ins here), so the pin is coarse; see models_image_pin."""
    import os
    from helpers import ROOT
    from raytracer_amd import HostBuilder, camera_new, scenes
    from test_oracle_pins import models_image_pin
    ref = np.load(os.path.join(ROOT, "tests", "golden", "reference_models_blockmeans.npz"))
    spec = scenes.build("C4", HostBuilder(), width=600)
    cam = camera_new(image_width=600, **spec.camera)
    scene = spec.builder.commit(0)
    try:
        film, _ = scene.render(cam, 200, 50, spec.background)
        worst = models_image_pin(scene.tonemap(film, 200).astype(np.float64), ref)
        print("largest block difference against the reference's image:", worst)
    finally:
        scene.close()


def test_image_rmse_gpu_vs_oracle_at_equal_spp(gpu_scenes):
    """Image check at equal spp and equal random streams.  The two renders trace the same paths except where a
    last-bit difference between libm and the device library (log in the medium, sin in the perlin texture, pow in
    Schlick) sends a path the other way; in final_scene, whose glass and fuzzy-metal spheres amplify such
    differences, that happens to a fraction of a percent of the samples, each of which then contributes
    independent noise.  Bounds: RMSE in 8-bit grey levels and the share of pixels off by more than one level."""
    from oracle import oracle as orc
    for cfg, rmse_max, off_max in (("C1", 0.5, 0.002), ("C5", 4.0, 0.03)):
        hs, os_, scene = gpu_scenes(cfg)
        hc, oc = cameras(hs)
        spp = 16
        film, _ = scene.render(hc, spp, hs.max_depth, hs.background)
        want, _ = os_.builder.render(oc, spp, hs.max_depth, hs.background)
        a = scene.tonemap(film, spp).astype(np.float64)
        b = orc.tonemap(want, spp).astype(np.float64)
        rmse = np.sqrt(np.mean((a - b) ** 2))
        off = (np.abs(a - b).max(axis=2) > 1).mean()
        print(cfg, "rmse", rmse, "off", off)
        assert rmse < rmse_max, (cfg, rmse)
        assert off < off_max, (cfg, off)
        assert 20 * np.log10(255.0 / max(rmse, 1e-9)) > 36.0  # PSNR


def test_full_size_headline_workload_properties():
    """BASELINE's configs[1] at its full size (cornell_smoke 600x600, 200 spp, 72 M paths), through size-independent
    properties: the film does not depend on how the work is cut up (queue size, schedule, cull boxes: bit for bit;
    two half renders: to rounding, since only the association of the sum changes) and repeats exactly."""
    from raytracer_amd import HostBuilder, camera_new, scenes
    spec = scenes.build("C2", HostBuilder(), width=600)
    cam = camera_new(image_width=600, **spec.camera)
    scene = spec.builder.commit(0)
    try:
        spp = 200
        film, st = scene.render(cam, spp, spec.max_depth, spec.background)
        assert st["paths"] == 600 * 600 * spp and st["rays"] > 4 * st["paths"]
        again, st2 = scene.render(cam, spp, spec.max_depth, spec.background)
        assert np.array_equal(film, again) and st2["rays"] == st["rays"]
        small_queue, st3 = scene.render(cam, spp, spec.max_depth, spec.background, batch_paths=3_000_000)
        assert np.array_equal(film, small_queue) and st3["rays"] == st["rays"]
        assert st3["bounce_launches"] > st["bounce_launches"] > 0
        whole, st4 = scene.render(cam, spp, spec.max_depth, spec.background, whole_paths=True)
        assert np.array_equal(film, whole) and st4["rays"] == st["rays"] and st4["path_launches"] > 0
        split, st5 = scene.render(cam, spp, spec.max_depth, spec.background, whole_paths=False, split_bounce=True)
        assert np.array_equal(film, split) and st5["traverse_launches"] > 0 and st5["bounce_launches"] == 0
        no_cull, _ = scene.render(cam, spp, spec.max_depth, spec.background, cull=False)
        assert np.array_equal(film, no_cull)
        a, sa = scene.render(cam, 100, spec.max_depth, spec.background, spp_begin=0)
        b, sb = scene.render(cam, 100, spec.max_depth, spec.background, spp_begin=100)
        assert sa["rays"] + sb["rays"] == st["rays"]
        np.testing.assert_allclose(a + b, film, rtol=1e-12, atol=1e-12)
        # a box lit from inside, seen through its open side: nothing is negative or NaN, and all but the margin
        # around the opening received light
        assert np.isfinite(film).all() and (film >= 0).all() and (film.sum(axis=2) > 0).mean() > 0.85
    finally:
        scene.close()


def test_full_resolution_bvh_scene_properties():
    """cornell_model at BASELINE's 800x800 (20 of its 2000 spp): the cull boxes and the queue size leave the film
    bit-identical, and the run repeats exactly, on the kernel that hands rays to lanes in whatever order they free up."""
    from raytracer_amd import HostBuilder, camera_new, scenes
    spec = scenes.build("C4", HostBuilder(), width=800)
    cam = camera_new(image_width=800, **spec.camera)
    scene = spec.builder.commit(0)
    try:
        film, st = scene.render(cam, 20, spec.max_depth, spec.background)
        again, _ = scene.render(cam, 20, spec.max_depth, spec.background)
        assert np.array_equal(film, again)
        no_cull, st_nc = scene.render(cam, 20, spec.max_depth, spec.background, cull=False, batch_paths=5_000_000)
        assert np.array_equal(film, no_cull) and st_nc["rays"] == st["rays"]
        # the room's six quads are tested by the shade launches, after the meshes' trees; on request they get a launch of
        # their own ahead of the trees, which is the list's own order: the same film
        assert st["shade_quads"] == 6 and st_nc["shade_quads"] == 0  # (without cull boxes: no work lists, the list's order)
        in_order, st_o = scene.render(cam, 20, spec.max_depth, spec.background, split_bounce=True)
        assert st_o["shade_quads"] == 0 and np.array_equal(film, in_order) and st_o["rays"] == st["rays"]
    finally:
        scene.close()


def _room_with_a_tree(b, kind):
    """A lit room of six quads, then one BVH: of boxes standing on the floor (their bottom faces lie in the floor's plane:
    a ray from below meets floor and face at exactly the same parameter) or of triangles lying in that plane.
    kind "mixed": the triangles, with a checkered floor, a metal wall and glass, metal and Lambertian triangles."""
    white = b.mat_lambertian(b.tex_solid(0.73, 0.73, 0.73))
    red = b.mat_lambertian(b.tex_solid(0.65, 0.05, 0.05))
    light = b.mat_diffuse_light(b.tex_solid(7.0, 7.0, 7.0))
    world = b.list_new()
    mixed = kind == "mixed"
    floor = b.mat_lambertian(b.tex_checker(0.8, b.tex_solid(0.2, 0.3, 0.1), b.tex_solid(0.9, 0.9, 0.9))) if mixed else white
    wall = b.mat_metal(0.8, 0.85, 0.88, 0.1) if mixed else red
    others = [b.mat_dielectric(1.5), b.mat_metal(0.7, 0.6, 0.5, 0.3), white] if mixed else [white]
    b.list_add(world, b.quad((0.0, 0.0, 0.0), (8.0, 0.0, 0.0), (0.0, 0.0, 8.0), floor))      # floor, y = 0
    b.list_add(world, b.quad((0.0, 8.0, 0.0), (8.0, 0.0, 0.0), (0.0, 0.0, 8.0), white))      # ceiling
    b.list_add(world, b.quad((0.0, 0.0, 8.0), (8.0, 0.0, 0.0), (0.0, 8.0, 0.0), white))      # back
    b.list_add(world, b.quad((0.0, 0.0, 0.0), (0.0, 8.0, 0.0), (0.0, 0.0, 8.0), wall))       # left
    b.list_add(world, b.quad((8.0, 0.0, 0.0), (0.0, 8.0, 0.0), (0.0, 0.0, 8.0), red))        # right
    b.list_add(world, b.quad((3.0, 7.5, 3.0), (2.0, 0.0, 0.0), (0.0, 0.0, 2.0), light))
    inner = b.list_new()
    rs = np.random.RandomState(5)
    for i in range(40):
        x, z = 0.5 + (i % 7), 0.5 + (i // 7)
        if kind == "boxes":
            b.list_add(inner, b.slice(b.box((x, 0.0, z), (x + 0.5, float(rs.uniform(0.5, 3.0)), z + 0.5), red)))
        else:
            b.list_add(inner, b.triangle((x, 0.0, z), (0.75, 0.0, 0.0), (0.0, 0.0, 0.75), red))
            b.list_add(inner, b.triangle((x, float(rs.uniform(0.5, 3.0)), z), (0.75, 0.0, 0.0), (0.0, 0.5, 0.75), others[i % len(others)]))
    b.list_add(world, b.bvh(inner))
    b.set_world(b.slice(world))


@pytest.mark.parametrize("kind", ["boxes", "triangles", "mixed"])
def test_quads_ahead_of_trees_tested_after_them_give_the_list_order_result(kind):
    """Where a run of quads precedes nothing but eight-wide trees of triangles (or spheres), a render walks the trees
    first and tests the quads inside the shade launches.  The closest hit does not depend on the order except between
    hits at exactly equal parameters, which the reference's list order decides (hittable_list.rs:88-102): the trees are
    handed (tmin, t of the closest quad), a triangle tests a half-open interval (triangle.rs:131), so the quad keeps a
    tie.  Trees of quads stay in the list's order: parallelogram.rs:101-103 tests a closed interval, and a hit at the
    cap passes or not with the leaf box's interval (here it does not: the boxes' bottom faces lie in the plane of their
    BVH boxes, where the reference loses them).  Seen from below the floor, camera rays meet the floor and a box's
    bottom face, or a triangle lying in the floor, at exactly the same parameter; the films of both schedules agree bit
    for bit and with the oracle's."""
    from raytracer_amd import HostBuilder
    from oracle import oracle as orc
    from raytracer_amd.builder import camera_new as host_camera_new
    hb, ob = HostBuilder(), orc.OracleBuilder()
    _room_with_a_tree(hb, kind)
    _room_with_a_tree(ob, kind)
    camera = dict(aspect_ratio=1.0, vfov=60.0, defocus_angle=0.0, focus_distance=10.0, look_from=(4.0, -6.0, 4.0),
                  look_at=(4.0, 0.0, 4.5), view_up=(0.0, 0.0, 1.0))
    # (more than a million paths: fewer go straight to the kernel that follows the last paths to their ends, and the
    # launches this test is about never run)
    hc, oc = host_camera_new(image_width=640, **camera), orc.camera_new(image_width=640, **camera)
    bg = (0.1, 0.1, 0.1)
    scene = hb.commit(0)
    try:
        film, st = scene.render(hc, 4, 8, bg)
        in_order, st_o = scene.render(hc, 4, 8, bg, split_bounce=True)
        # ("mixed": plain and other materials on walls and triangles alike, so the shade phase is two launches as well --
        # the plain code over every ray, the rest from its list, both testing the quads first)
        assert st["shade_quads"] == (0 if kind == "boxes" else 6) and st_o["shade_quads"] == 0
        assert np.array_equal(film, in_order) and st["rays"] == st_o["rays"]
        assert st["traverse_launches"] >= 2 and st_o["traverse_launches"] >= 2
        want, _ = ob.render(oc, 4, 8, bg)
        assert np.isclose(film, want, rtol=1e-9, atol=1e-12).all(axis=2).mean() > 0.995
        # the ties are there: camera rays from below that hit the floor's plane where a tree's primitive lies in it too
        n = 4000
        rs = np.random.RandomState(3)
        o = np.column_stack([rs.uniform(0.5, 7.0, n), np.full(n, -6.0), rs.uniform(0.5, 6.0, n)])
        d = np.tile([0.0, 1.0, 0.0], (n, 1))
        got, ref = scene.intersect(o, d, 0.001, np.inf), ob.intersect(o, d, 0.001, np.inf)
        assert np.array_equal(got["prim"], ref["prim"]) and np.array_equal(got["t"], ref["t"])
        assert (ref["t"] == 6.0).all() and (ref["prim"] < 6).all()  # the floor keeps every tie
    finally:
        scene.close()


@pytest.mark.parametrize("cfg", ["C2", "C5"])
def test_many_small_batches_through_the_counter_ring(gpu_scenes, cfg):
    """The host queues launches ahead of the ray counts it has read and rotates through a few sets of device
    counters.  A queue that holds one sample of every pixel at a time forces one batch per iteration for many
    iterations (sets reused many times, batches finishing while later ones are in flight, luminance sections
    recycled); a depth of 1 and of 2 ends paths before the first counts are read.  Same film bit for bit as with
    everything in one batch, and the same ray count."""
    hs, _, scene = gpu_scenes(cfg)
    hc, _ = cameras(hs)
    npix = hc.width * hc.height
    for depth in (hs.max_depth, 2, 1):
        want, st_w = scene.render(hc, 37, depth, hs.background, spp_begin=3)
        got, st_g = scene.render(hc, 37, depth, hs.background, spp_begin=3, batch_paths=npix)
        assert np.array_equal(got, want), depth
        assert st_g["rays"] == st_w["rays"] and st_g["paths"] == st_w["paths"] == 37 * npix
        launches = st_g["bounce_launches"] + st_g["traverse_launches"]
        assert launches >= 37 and launches > st_w["bounce_launches"] + st_w["traverse_launches"]
        assert st_g["bounces"] == st_w["bounces"] <= depth


def test_render_multi_splits_samples_and_adds_shards_in_rank_order():
    """rt_render_multi on a one-GPU box: the same scene committed three times to device 0 stands in for three GPUs.
    The result must be, bit for bit, the rank-ordered sum of the three shard renders (7 samples -> 3 + 2 + 2)."""
    from raytracer_amd import HostBuilder, camera_new, scenes
    from raytracer_amd.scene import render_multi
    spec = scenes.build("C5", HostBuilder(), **SMALL["C5"])
    cam = camera_new(image_width=spec.width, **spec.camera)
    devs = [spec.builder.commit(0) for _ in range(3)]
    try:
        film, st = render_multi(devs, cam, 7, spec.max_depth, spec.background, spp_begin=2)
        parts = [devs[0].render(cam, n, spec.max_depth, spec.background, spp_begin=b) for b, n in ((2, 3), (5, 2), (7, 2))]
        want = (parts[0][0] + parts[1][0]) + parts[2][0]
        assert np.array_equal(film, want)
        assert st["rays"] == sum(p[1]["rays"] for p in parts) and st["paths"] == 7 * cam.width * cam.height
        whole, _ = devs[0].render(cam, 7, spec.max_depth, spec.background, spp_begin=2)
        np.testing.assert_allclose(film, whole, rtol=1e-12, atol=1e-12)
        one, _ = render_multi(devs[:1], cam, 7, spec.max_depth, spec.background, spp_begin=2)
        assert np.array_equal(one, whole)
        with pytest.raises(Exception):
            render_multi([devs[0], devs[0]], cam, 4, spec.max_depth, spec.background)
    finally:
        for d in devs:
            d.close()


def test_render_multi_reduces_the_films_with_rccl(monkeypatch):
    """With every scene on a device of its own rt_render_multi sums the films with one RCCL reduce.  A one-GPU box has
    one such device: RT_MULTI_RCCL=force takes a single scene through the same calls (library loaded at run time,
    communicator, grouped reduce with root 0, stream wait), and the film must come out as rt_render gives it."""
    from raytracer_amd import HostBuilder, _lib, camera_new, scenes
    from raytracer_amd.scene import render_multi
    spec = scenes.build("C2", HostBuilder(), **SMALL["C2"])
    cam = camera_new(image_width=spec.width, **spec.camera)
    scene = spec.builder.commit(0)
    try:
        whole, _ = scene.render(cam, 5, spec.max_depth, spec.background, spp_begin=1)
        before = _lib.amd().rt_multi_rccl_reductions()
        monkeypatch.setenv("RT_MULTI_RCCL", "force")
        film, st = render_multi([scene], cam, 5, spec.max_depth, spec.background, spp_begin=1)
        assert _lib.amd().rt_multi_rccl_reductions() == before + 1
        assert np.array_equal(film, whole) and st["paths"] == 5 * cam.width * cam.height
        film2, _ = render_multi([scene], cam, 5, spec.max_depth, spec.background, spp_begin=1)  # the communicator is kept
        assert _lib.amd().rt_multi_rccl_reductions() == before + 2 and np.array_equal(film2, whole)
        monkeypatch.setenv("RT_MULTI_RCCL", "1")
        film3, _ = render_multi([scene], cam, 5, spec.max_depth, spec.background, spp_begin=1)
        assert _lib.amd().rt_multi_rccl_reductions() == before + 2 and np.array_equal(film3, whole)
    finally:
        scene.close()


def test_edge_cases_empty_inputs_and_limits(gpu_scenes):
    """What the reference does at the edges: an empty world shows the background (camera.rs:364-366 on a miss);
    depth 0 is black (camera.rs:350-352); zero samples leave the fold's start value -0.0 (point3/mod.rs:96-109);
    empty ray batches and sample lists are no-ops; bad arguments are errors, not crashes."""
    from raytracer_amd import HostBuilder, RtError, camera_new
    b = HostBuilder()
    b.set_world(b.slice(b.list_new()))
    scene = b.commit(0)
    cam = camera_new(image_width=16, aspect_ratio=1.0, vfov=40.0, look_from=(0, 0, -5), look_at=(0, 0, 0), view_up=(0, 1, 0),
                     defocus_angle=0.0, focus_distance=1.0)
    try:
        film, st = scene.render(cam, 3, 5, (0.25, 0.5, 1.0))
        assert np.array_equal(film, np.broadcast_to(np.array([0.75, 1.5, 3.0]), film.shape))
        assert st["rays"] == st["paths"] == 3 * cam.width * cam.height
        r = scene.intersect(np.zeros((4, 3)), np.ones((4, 3)), 0.001, np.inf)
        assert not r["hit"].any() and (r["prim"] == 0xFFFFFFFF).all()
    finally:
        scene.close()
    hs, os_, scene = gpu_scenes("C2")
    hc, oc = cameras(hs)
    film, st = scene.render(hc, 2, 0, hs.background)
    assert np.array_equal(film, np.zeros_like(film)) and not np.signbit(film).any() and st["rays"] == 0
    film, st = scene.render(hc, 0, 5, hs.background)
    assert (film == 0).all() and np.signbit(film).all() and st["paths"] == 0
    r = scene.intersect(np.zeros((0, 3)), np.zeros((0, 3)), 0.001, np.inf)
    assert r["hit"].shape == (0,)
    rad, nr = scene.trace_samples(hc, np.zeros(0, np.uint32), np.zeros(0, np.uint32), 5, hs.background)
    assert rad.shape == (0, 3)
    with pytest.raises(RtError):
        scene.trace_samples(hc, np.array([hc.width * hc.height], np.uint32), np.array([0], np.uint32), 5, hs.background)
    # a ray that starts exactly on a wall and one parallel to it: same answers as the oracle
    o = np.array([[0.0, 100.0, 100.0], [278.0, 278.0, -800.0], [555.0, 10.0, 10.0]])
    d = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
    assert_hits_equal(scene.intersect(o, d, 0.0, np.inf), os_.builder.intersect(o, d, 0.0, np.inf), exact=True, what="walls")


def test_box_floor_corners_edges_and_ties_match_the_oracle():
    """rt_intersect_batch on a floor of touching boxes under a BVH (the eight-wide walk, which of a box's six quads tests
    the sides a ray can meet first and the others only if those are missed) for rays aimed at corners, at edges, along
    axes and diagonals, from inside boxes and from far away: exact against the oracle, ties at shared corners included
    (a later quad takes a tie: parallelogram.rs:101-103)."""
    from raytracer_amd import HostBuilder
    from oracle import oracle as orc
    from helpers import box_floor_rays, box_floor_scene
    hb, ob = HostBuilder(), orc.OracleBuilder()
    box_floor_scene(hb)
    box_floor_scene(ob)
    o, d, tmin, tmax = box_floor_rays()
    scene = hb.commit(0)
    try:
        got, want = scene.intersect(o, d, tmin, tmax), ob.intersect(o, d, tmin, tmax)
        assert_hits_equal(got, want, exact=True, what="box floor")
        ties = (want["t"] == 1.0) & (want["hit"] == 1)
        assert ties.sum() > 100  # (many rays aimed at a corner or an edge reach it at exactly t = 1)
    finally:
        scene.close()
