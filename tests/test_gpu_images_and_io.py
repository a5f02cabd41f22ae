"""GPU tests beyond the small-image parity suite: images at equal high sample counts against the oracle for all five
BASELINE configurations, intersection batches at real pixel footprints, the output step of the host layer, the
remaining quadric constructors, a tree deeper than a workgroup's LDS would hold, and the ordering of renders into one
device film."""
import json
import os

import numpy as np
import pytest

from helpers import assert_hits_equal, build_pair, cameras, path_rays

pytestmark = pytest.mark.gpu

# Image bounds per configuration at 256 spp, 160 pixels wide, equal random streams on both sides.  The two renders
# trace the same paths except where a last-bit difference between libm and the device library (log in a medium, sin
# in the perlin texture, pow in Schlick's term, acos/atan2 in a sphere's u, v) sends a path the other way; each such
# path then contributes independent noise.  rmse8: RMSE of the 8-bit images in grey levels; psnr: dB; lin: RMSE of the
# linear radiance averages over pixels neither image clamps (point3/color.rs:61-65), relative to their mean;
# same: least share of pixels whose radiance sums agree to 1e-9.
IMAGE_BOUNDS = {   # (measured on an MI355X: rmse8 0.0 and same 1.0 on all five; lin 6e-17 .. 7e-14)
    "C1": dict(rmse8=0.02, psnr=80.0, lin=1e-9, same=0.999),
    "C2": dict(rmse8=0.02, psnr=80.0, lin=1e-9, same=0.999),
    "C3": dict(rmse8=0.02, psnr=80.0, lin=1e-9, same=0.999),
    "C4": dict(rmse8=0.02, psnr=80.0, lin=1e-9, same=0.999),
    "C5": dict(rmse8=0.1, psnr=65.0, lin=1e-6, same=0.99),
}


@pytest.mark.parametrize("cfg", sorted(IMAGE_BOUNDS))
def test_images_at_equal_high_spp_match_the_oracle(cfg):
    from oracle import oracle as orc
    width, spp = 160, 256
    hs, os_ = build_pair(cfg, width=width, spp=spp)
    hc, oc = cameras(hs)
    scene = hs.builder.commit(0)
    try:
        film, st = scene.render(hc, spp, hs.max_depth, hs.background)
        want, _ = os_.builder.render(oc, spp, hs.max_depth, hs.background, threads=min(16, os.cpu_count() or 1))
        a = scene.tonemap(film, spp).astype(np.float64)
        b = orc.tonemap(want, spp).astype(np.float64)
        rmse8 = float(np.sqrt(np.mean((a - b) ** 2)))
        psnr = float(20 * np.log10(255.0 / max(rmse8, 1e-9)))
        unclamped = ((film / spp) < 0.999 ** 2).all(axis=2) & ((want / spp) < 0.999 ** 2).all(axis=2)
        lin = float(np.sqrt(np.mean(((film - want)[unclamped] / spp) ** 2)) / max(np.mean(want[unclamped] / spp), 1e-12))
        same = float(np.isclose(film, want, rtol=1e-9, atol=1e-12).all(axis=2).mean())
        print(json.dumps({"config": cfg, "rmse8": rmse8, "psnr": psnr, "lin_rel_rmse": lin, "same": same, "paths": st["paths"]}))
        bound = IMAGE_BOUNDS[cfg]
        assert rmse8 <= bound["rmse8"] and psnr >= bound["psnr"] and lin <= bound["lin"] and same >= bound["same"], \
            (cfg, rmse8, psnr, lin, same)
    finally:
        scene.close()


@pytest.mark.parametrize("cfg", ["C1", "C2", "C3", "C4", "C5"])
def test_intersect_batches_at_real_pixel_footprints(cfg):
    """Rays of a 240-pixel-wide image (thin features and grazing hits at real footprints), camera rays and two
    scattered generations: the oracle's hits bit for bit wherever no logarithm is involved."""
    hs, os_ = build_pair(cfg, width=240, spp=4)
    _, oc = cameras(hs)
    scene = hs.builder.commit(0)
    try:
        for bounce in (0, 1, 2):
            o, d, rng3 = path_rays(os_, oc, 60000, bounce)
            want = os_.builder.intersect(o, d, 0.001, np.inf, rng3)
            got = scene.intersect(o, d, 0.001, np.inf, rng3)
            if cfg in ("C2", "C5"):     # a medium hit's t comes from log(U): everything else must be bit-identical
                assert np.array_equal(got["hit"], want["hit"]) and np.array_equal(got["prim"], want["prim"])
                assert np.array_equal(got["ctr"], want["ctr"]) and np.array_equal(got["front"], want["front"])
                np.testing.assert_allclose(got["t"], want["t"], rtol=1e-12, atol=0)
                plain = got["ctr"] == rng3[:, 2]      # no draw was made on the way: no medium was crossed
                assert np.array_equal(got["t"][plain], want["t"][plain]) and np.array_equal(got["n"][plain], want["n"][plain])
            else:
                assert_hits_equal(got, want, exact=True, what=f"{cfg} bounce {bounce}", uv_exact=(cfg != "C1"))
    finally:
        scene.close()


def test_thrender_writes_the_tonemapped_film_as_png(tmp_path, capfd):
    """Camera::thrender of the host layer (camera.rs:248-271): images/image.png holds rt_tonemap_rgb8 of the film the
    same scene renders through the C ABI, and the metrics line says what was traced."""
    from PIL import Image
    hs, _ = build_pair("C3", width=96, spp=8)
    hc, _ = cameras(hs)
    (tmp_path / "images").mkdir()
    out = tmp_path / "images" / "image.png"
    st = hs.builder.thrender(hs.builder.world, hc, 8, hs.max_depth, hs.background, out, print_metrics=True)
    scene = hs.builder.commit(0)
    try:
        film, st2 = scene.render(hc, 8, hs.max_depth, hs.background)
        want = scene.tonemap(film, 8)
    finally:
        scene.close()
    assert np.array_equal(np.asarray(Image.open(out)), want)
    assert st["paths"] == st2["paths"] == 96 * 96 * 8 and st["rays"] == st2["rays"]
    line = [l for l in capfd.readouterr().out.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["event"] == "render_done" and rec["paths"] == st["paths"] and rec["rays"] == st["rays"] and rec["width"] == 96
    with pytest.raises(Exception):      # the directory must exist, as for the reference (camera.rs:270 unwraps)
        hs.builder.thrender(hs.builder.world, hc, 1, hs.max_depth, hs.background, tmp_path / "nowhere" / "image.png")


def test_cone_and_quadric_sphere_match_oracle():
    """y_cone and quadric_sphere (quadric.rs:138-177) on the GPU: hits bit for bit, images to the last-bit paths."""
    from oracle import oracle as orc
    from raytracer_amd import HostBuilder, camera_new

    def make(b):
        white = b.mat_lambertian(b.tex_solid(0.73, 0.73, 0.73))
        red = b.mat_lambertian(b.tex_solid(0.65, 0.05, 0.05))
        light = b.mat_diffuse_light(b.tex_solid(7.0, 7.0, 7.0))
        objs = [b.quad((-20.0, -3.0, -20.0), (40.0, 0.0, 0.0), (0.0, 0.0, 40.0), white),
                b.quad((-5.0, 9.0, -5.0), (10.0, 0.0, 0.0), (0.0, 0.0, 10.0), light),
                b.y_cone((0.0, 4.0, 0.0), (0.5, 1.0, 0.5), red),
                b.quadric_sphere((3.5, -1.0, 1.0), 2.0, b.mat_metal(0.8, 0.8, 0.9, 0.1)),
                b.y_cylinder((-4.0, 0.0, 2.0), 1.2, b.mat_dielectric(1.5))]
        b.set_world(b.slice_of(objs))

    hb, ob = HostBuilder(), orc.OracleBuilder()
    make(hb)
    make(ob)
    cam_kw = dict(aspect_ratio=1.0, vfov=50.0, defocus_angle=0.0, focus_distance=10.0, look_from=(0.0, 3.0, -14.0),
                  look_at=(0.0, 1.0, 0.0), view_up=(0.0, 1.0, 0.0))
    hc = camera_new(image_width=120, **cam_kw)
    oc = orc.camera_new(image_width=120, **cam_kw)
    scene = hb.commit(0)
    try:
        rs = np.random.RandomState(9)
        n = 40000
        o = rs.uniform(-12, 12, (n, 3))
        d = rs.uniform(-4, 4, (n, 3)) - o
        want = ob.intersect(o, d, 0.001, np.inf)
        got = scene.intersect(o, d, 0.001, np.inf)
        assert_hits_equal(got, want, exact=True, what="cone scene")
        prims = set(np.unique(want["prim"][want["hit"] == 1]).tolist())
        assert {2, 3, 4} <= prims            # the cone, the quadric sphere and the cylinder are all hit
        bg = (0.05, 0.05, 0.1)
        film, _ = scene.render(hc, 32, 20, bg)
        ref, _ = ob.render(oc, 32, 20, bg)
        assert np.isclose(film, ref, rtol=1e-9, atol=1e-12).all(axis=2).mean() > 0.97
    finally:
        scene.close()


def test_deep_tree_through_the_tail_kernel():
    """BVHs nested in BVHs until the walk needs more interval levels than 256 lanes' worth fit a workgroup's LDS: the
    tail kernel, which otherwise runs 256-thread workgroups, must still render it (and equal the in-order sum of the
    per-sample radiances bit for bit)."""
    from raytracer_amd import HostBuilder, camera_new
    from raytracer_amd.philox import SceneRng
    b = HostBuilder()
    rng = SceneRng(77)
    white = b.mat_lambertian(b.tex_solid(0.7, 0.7, 0.7))
    light = b.mat_diffuse_light(b.tex_solid(5.0, 5.0, 5.0))
    inner = None
    for level in range(7):
        lst = b.list_new()
        if inner is not None:
            b.list_add(lst, inner)
        for _ in range(40):
            b.list_add(lst, b.sphere(rng.vector(-6.0, 6.0), rng.range(0.2, 0.6), white))
        inner = b.bvh(lst)
    b.set_world(b.slice_of([b.quad((-30.0, -7.0, -30.0), (60.0, 0.0, 0.0), (0.0, 0.0, 60.0), white),
                            b.quad((-6.0, 12.0, -6.0), (12.0, 0.0, 0.0), (0.0, 0.0, 12.0), light), inner]))
    levels = max(n.level for n in b.flat("nodes")) + 2
    assert levels * 256 * 16 > 64 * 1024, levels     # more than a 256-lane workgroup may have
    cam = camera_new(aspect_ratio=1.0, image_width=64, vfov=45.0, defocus_angle=0.0, focus_distance=10.0,
                     look_from=(0.0, 2.0, -20.0), look_at=(0.0, 0.0, 0.0), view_up=(0.0, 1.0, 0.0))
    scene = b.commit(0)
    try:
        spp, depth, bg = 3, 12, (0.1, 0.1, 0.2)
        film, st = scene.render(cam, spp, depth, bg)
        assert st["finish_launches"] == 1
        npix = cam.width * cam.height
        pixel = np.tile(np.arange(npix, dtype=np.uint32), spp)
        sample = np.repeat(np.arange(spp, dtype=np.uint32), npix)
        rad, nrays = scene.trace_samples(cam, pixel, sample, depth, bg)
        acc = np.full((npix, 3), -0.0)
        for s in range(spp):
            acc = acc + rad[s * npix:(s + 1) * npix]
        assert np.array_equal(film.reshape(npix, 3), acc) and st["rays"] == int(nrays.sum())
    finally:
        scene.close()


def test_two_renders_into_one_device_film_through_render_sharded():
    """render_sharded (world size 1) twice into the same torch tensor: the second render must not start before the
    caller has seen the first, and each leaves exactly its own film (the renderer works on a stream of its own)."""
    import torch
    from raytracer_amd.dist import render_scene_sharded, render_sharded
    hs, _ = build_pair("C2", width=96, spp=4)
    hc, _ = cameras(hs)
    scene = hs.builder.commit(0)
    try:
        want_a, _ = scene.render(hc, 6, hs.max_depth, hs.background, spp_begin=0)
        want_b, _ = scene.render(hc, 4, hs.max_depth, hs.background, spp_begin=6)
        film = torch.full((hc.height, hc.width, 3), 123.0, dtype=torch.float64, device="cuda:0")
        seen = []
        for begin, count in ((0, 6), (6, 4)):
            def render_range(b0, c0, begin=begin):
                scene.render_into(hc, c0, hs.max_depth, hs.background, film.data_ptr(), spp_begin=begin + b0)
            render_sharded(render_range, film, count)
            seen.append(film.cpu().numpy().copy())
        assert np.array_equal(seen[0], want_a) and np.array_equal(seen[1], want_b)
        whole, _ = render_scene_sharded(scene, hc, 10, hs.max_depth, hs.background)
        np.testing.assert_allclose(whole.cpu().numpy(), want_a + want_b, rtol=1e-12, atol=1e-12)
    finally:
        scene.close()


def test_render_sharded_over_the_nccl_backend_with_one_rank():
    """The multi-process route (one process per GPU, film all-reduced by RCCL through torch.distributed) with the one
    rank a one-GPU box can hold: process group on the "nccl" backend, device film, all-reduce, barrier -- the calls
    bench.py --gpus N makes -- and the film must be the single-process film."""
    import torch
    import torch.distributed as dist
    from raytracer_amd.dist import render_scene_sharded, shard_range
    hs, _ = build_pair("C4", width=64, spp=4)
    hc, _ = cameras(hs)
    scene = hs.builder.commit(0)
    port = 29600 + os.getpid() % 300
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        want, _ = scene.render(hc, 5, hs.max_depth, hs.background)
        film, stats = render_scene_sharded(scene, hc, 5, hs.max_depth, hs.background)
        assert shard_range(5, 0, 1) == (0, 5) and stats["paths"] == 5 * hc.width * hc.height
        dist.barrier()
        torch.cuda.synchronize()
        assert np.array_equal(film.cpu().numpy(), want)
    finally:
        dist.destroy_process_group()
        scene.close()


@pytest.mark.parametrize("count, kinds", [(1, "t"), (9, "t"), (130, "b"), (300, "stqb"), (1200, "s"), (5000, "tq")])
def test_wide_kernel_on_random_trees_of_every_kind(count, kinds):
    """k_traverse_wide through rt_intersect_batch on BVHs of spheres, triangles, quads and boxes (leaves of one kind and
    of several, trees of one node and of several levels), with rays from inside, from far away, along the axes and with
    intervals of the caller's: hit, primitive, t, point and normal exactly the oracle's."""
    from oracle import oracle as orc
    from raytracer_amd import HostBuilder
    from raytracer_amd.philox import SceneRng

    def make(b):
        rng = SceneRng(77 + count)
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
    scene = hb.commit(0)
    try:
        rs = np.random.RandomState(count)
        n = 20000
        o = rs.uniform(-7, 7, (n, 3))
        d = rs.normal(size=(n, 3))
        o[5000:8000] = rs.uniform(-1, 1, (3000, 3)) * 800.0
        d[5000:8000] = -o[5000:8000] + rs.uniform(-6, 6, (3000, 3))
        axis = rs.randint(0, 3, 3000)
        d[8000:11000] = 0.0
        d[np.arange(8000, 11000), axis] = rs.choice([-1.0, 1.0], 3000)
        tmin = np.full(n, 0.001)
        tmax = np.full(n, np.inf)
        tmin[11000:] = rs.uniform(-3.0, 3.0, n - 11000)
        tmax[11000:] = tmin[11000:] + rs.uniform(0.0, 12.0, n - 11000)
        rng3 = np.stack([np.arange(n), np.zeros(n), np.zeros(n)], axis=1).astype(np.uint32)
        want = ob.intersect(o, d, tmin, tmax, rng3)
        got = scene.intersect(o, d, tmin, tmax, rng3)
        assert 0.2 < want["hit"].mean() < 0.98
        assert_hits_equal(got, want, exact=True, what=f"{count} x {kinds}", uv_exact=False)  # (the ground is a sphere: acos, atan2)
    finally:
        scene.close()


def test_bench_line_keeps_its_contract():
    """bench.py's measure() on reduced workloads: the fields the driver reads, a primary roofline fraction that is a
    fraction (fp64 valu, between 0 and 1, never the algorithmic-bytes figure), the HBM roofline from measured traffic
    likewise, and for a tree with BVH entries the fast mode's report."""
    import argparse
    import sys
    import torch
    from helpers import ROOT
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    args = argparse.Namespace(width=120, batch_paths=0, scaling="weak", fast_bvh=False, whole_paths=False, queues=False,
                              split_bounce=False, no_fast_report=False, fast_report=True)
    for config, spp in (("C2", 16), ("C4", 8)):
        out = bench.measure(args, config, spp, 2, 1, torch, None, 0, 0, 1, False)
        assert out["metric"] == "Mrays/s" and out["unit"] == "Mrays/s" and out["n_gpus"] == 1 and out["steps"] == 2
        assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["dtype"] == "f64" and out["vs_baseline"] is None
        assert out["value"] > 0 and out["ms_per_step"] > 0 and out["config"]["paths_per_step"] > 0
        assert "workload" in out["config"] and "model" not in out["config"]
        roof = out["roofline"]
        assert roof["bound"] == "fp64 valu" and roof["unit"] == "TFLOP/s" and roof["peak"] == bench.FP64_PEAK_TFLOPS
        assert roof["frac"] is not None and 0.0 < roof["frac"] <= 1.0
        assert 0.0 < roof["frac_executed"] <= 1.0 and roof["frac_executed"] <= roof["frac_algorithmic"] * 1.0001
        assert abs(roof["achieved"] / roof["peak"] - roof["frac"]) < 1e-12
        assert "cache_served_bytes_per_ray" in roof and roof["avg_launch_ms"] > 0 and roof["launches"] > 0
        hbm = out.get("roofline_hbm")
        if hbm is not None:
            assert hbm["bound"] == "hbm" and hbm["peak"] == bench.HBM_PEAK_GBS and 0.0 < hbm["frac"] <= 1.0
            assert roof["traffic"] == hbm["traffic"] > 0
        if config == "C4":
            fast = out["fast_bvh"]
            assert fast["prim_id_mismatch_rate"] <= 1e-3 and fast["rays_compared"] > 0 and fast["speedup"] > 0.5
