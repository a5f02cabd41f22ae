#!/usr/bin/env python3
"""Development aid (run on the GPU box): many random scenes against the oracle, beyond what the test suite holds.

Part 1, ray batches: BVHs of random primitives (every mix of kinds, 1 .. 4000 of them, with and without an instance
chain around the tree) between a ground sphere and a quad, 20 000 rays each as in
tests/test_gpu_images_and_io.py::test_wide_kernel_on_random_trees_of_every_kind; hit, primitive, t, point and normal
must be the oracle's bit for bit.

Part 2, renders: a lit room of quads around one or two trees with random materials and textures; the film must not
depend on the schedule (one batch, which goes straight to the kernel that follows the last paths to their ends; many
small batches, which run the per-bounce launches; split bounces), must equal the in-order sum of rt_trace_samples bit
for bit, and must agree with the oracle's film except where a last-bit difference of a transcendental sent a path
another way.

usage: python3 tools/soak_gpu.py [--seeds N] [--first K] [--width W --spp S] [--no-batches]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import assert_hits_equal  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from raytracer_amd import HostBuilder  # noqa: E402
from raytracer_amd.builder import camera_new as host_camera_new  # noqa: E402
from raytracer_amd.philox import SceneRng  # noqa: E402

KINDS = ["t", "s", "q", "b", "ts", "tq", "sq", "stqb", "bq", "tb"]


def add_prim(b, lst, kind, rng, m, spread):
    c = rng.vector(-spread, spread)
    if kind == "s":
        b.list_add(lst, b.sphere(c, rng.range(0.1, 0.9), m))
    elif kind == "t":
        b.list_add(lst, b.triangle(c, rng.vector(-1, 1), rng.vector(-1, 1), m))
    elif kind == "q":
        b.list_add(lst, b.quad(c, rng.vector(-1, 1), rng.vector(-1, 1), m))
    else:
        b.list_add(lst, b.slice(b.box(c, (c[0] + rng.range(0.1, 1.0), c[1] + rng.range(0.1, 1.0), c[2] + rng.range(0.1, 1.0)), m)))


def batch_case(seed):
    rs0 = np.random.RandomState(1000 + seed)
    kinds = KINDS[seed % len(KINDS)]
    count = int(rs0.choice([1, 2, 7, 9, 17, 64, 130, 300, 1200, 4000]))
    chain = int(rs0.randint(0, 3))  # 0: none, 1: translate, 2: rotate + translate

    def make(b):
        rng = SceneRng(500 + seed)
        m = b.mat_lambertian(b.tex_solid(0.5, 0.5, 0.5))
        lst = b.list_new()
        for k in range(count):
            add_prim(b, lst, kinds[k % len(kinds)], rng, m, 6.0)
        world = b.list_new()
        b.list_add(world, b.sphere((0.0, -1000.0, 0.0), 990.0, m))
        tree = b.bvh(lst)
        if chain == 2:
            tree = b.rotate_y(tree, 25.0)
        if chain >= 1:
            tree = b.translate(tree, (0.5, -0.25, 1.0))
        b.list_add(world, tree)
        b.list_add(world, b.quad((-8.0, -8.0, 9.0), (16.0, 0.0, 0.0), (0.0, 16.0, 0.0), m))
        b.set_world(b.slice(world))

    hb, ob = HostBuilder(), orc.OracleBuilder()
    make(hb)
    make(ob)
    scene = hb.commit(0)
    try:
        rs = np.random.RandomState(seed)
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
        assert_hits_equal(got, want, exact=True, what=f"batch seed {seed}: {count} x {kinds}, chain {chain}", uv_exact=False)
    finally:
        scene.close()
    return f"{count} x {kinds}, chain {chain}"


def room(b, seed):
    """Six quads and a light, then one or two BVHs of one kind of primitive each (or a mix), materials drawn per seed; every fourth seed
    with two constant media among the top-level entries."""
    rs = np.random.RandomState(2000 + seed)
    rng = SceneRng(900 + seed)
    plain_only = seed % 3 == 0
    white = b.mat_lambertian(b.tex_solid(0.73, 0.73, 0.73))
    red = b.mat_lambertian(b.tex_solid(0.65, 0.05, 0.05))
    light = b.mat_diffuse_light(b.tex_solid(9.0, 9.0, 9.0))
    pool = [white, red]
    if not plain_only:
        pool += [b.mat_metal(0.8, 0.85, 0.88, 0.2), b.mat_dielectric(1.5),
                 b.mat_lambertian(b.tex_checker(0.9, b.tex_solid(0.2, 0.3, 0.1), b.tex_solid(0.9, 0.9, 0.9)))]
        if seed % 3 == 2:
            pool.append(b.mat_lambertian(b.tex_perlin(4.0, [rng.range(-1.0, 1.0) for _ in range(768)])))
    world = b.list_new()
    walls_first = seed % 2 == 0
    walls = [((0.0, 0.0, 0.0), (8.0, 0.0, 0.0), (0.0, 0.0, 8.0), pool[seed % len(pool)]),
             ((0.0, 8.0, 0.0), (8.0, 0.0, 0.0), (0.0, 0.0, 8.0), white),
             ((0.0, 0.0, 8.0), (8.0, 0.0, 0.0), (0.0, 8.0, 0.0), white),
             ((0.0, 0.0, 0.0), (0.0, 8.0, 0.0), (0.0, 0.0, 8.0), red),
             ((8.0, 0.0, 0.0), (0.0, 8.0, 0.0), (0.0, 0.0, 8.0), pool[(seed + 1) % len(pool)]),
             ((3.0, 7.9, 3.0), (2.0, 0.0, 0.0), (0.0, 0.0, 2.0), light)]
    if walls_first:
        for q in walls:
            b.list_add(world, b.quad(*q))
    kinds = ["t", "s", "tq", "b", "ts"][seed % 5]
    for tree in range(1 + seed % 2):
        lst = b.list_new()
        for k in range(int(rs.choice([3, 20, 90, 400]))):
            c = (float(rs.uniform(1.0, 7.0)), float(rs.uniform(0.2, 5.0)), float(rs.uniform(1.0, 7.0)))
            m = pool[int(rs.randint(0, len(pool)))]
            kind = kinds[k % len(kinds)]
            if kind == "s":
                b.list_add(lst, b.sphere(c, float(rs.uniform(0.1, 0.5)), m))
            elif kind == "t":
                b.list_add(lst, b.triangle(c, rng.vector(-0.8, 0.8), rng.vector(-0.8, 0.8), m))
            elif kind == "q":
                b.list_add(lst, b.quad(c, rng.vector(-0.8, 0.8), rng.vector(-0.8, 0.8), m))
            else:
                b.list_add(lst, b.slice(b.box(c, (c[0] + 0.4, c[1] + 0.6, c[2] + 0.3), m)))
        node = b.bvh(lst)
        if tree == 1:
            node = b.translate(b.rotate_y(node, 15.0), (0.2, 0.0, -0.1))
        b.list_add(world, node)
    if seed % 4 == 3:  # media: a ball of fog, and a rotated box of smoke as in cornell_smoke
        ball = b.sphere((5.5, 2.0, 5.0), 1.2, b.mat_dielectric(1.5))
        if seed % 8 == 3:
            b.list_add(world, ball)
        b.list_add(world, b.constant_medium(ball, 0.6, b.tex_solid(0.2, 0.4, 0.9)))
        smoke = b.translate(b.rotate_y(b.slice(b.box((0.0, 0.0, 0.0), (1.5, 3.0, 1.5), white)), 18.0), (1.5, 0.0, 3.0))
        b.list_add(world, b.constant_medium(smoke, 0.9, b.tex_solid(0.0, 0.0, 0.0)))
    if not walls_first:
        for q in walls:
            b.list_add(world, b.quad(*q))
    b.set_world(b.slice(world))


def render_case(seed, width=96, spp=6):
    hb, ob = HostBuilder(), orc.OracleBuilder()
    room(hb, seed)
    room(ob, seed)
    camera = dict(aspect_ratio=1.0, vfov=70.0, defocus_angle=0.0, focus_distance=10.0, look_from=(4.0, 4.0, 0.3),
                  look_at=(4.0, 3.5, 8.0), view_up=(0.0, 1.0, 0.0))
    depth = 12
    hc, oc = host_camera_new(image_width=width, **camera), orc.camera_new(image_width=width, **camera)
    bg = (0.02, 0.02, 0.03)
    npix = hc.width * hc.height
    scene = hb.commit(0)
    try:
        film, st = scene.render(hc, spp, depth, bg)
        small, st2 = scene.render(hc, spp, depth, bg, batch_paths=npix)
        split, st3 = scene.render(hc, spp, depth, bg, batch_paths=2 * npix, split_bounce=True)
        assert np.array_equal(film, small) and st["rays"] == st2["rays"], f"render seed {seed}: small batches differ"
        assert np.array_equal(film, split) and st["rays"] == st3["rays"], f"render seed {seed}: split bounces differ"
        assert st2["traverse_launches"] > st["traverse_launches"]
        pixel = np.tile(np.arange(npix, dtype=np.uint32), spp)
        sample = np.repeat(np.arange(spp, dtype=np.uint32), npix)
        rad, nrays = scene.trace_samples(hc, pixel, sample, depth, bg)
        acc = np.full((npix, 3), -0.0)
        for s in range(spp):
            acc = acc + rad[s * npix:(s + 1) * npix]
        assert np.array_equal(film.reshape(npix, 3), acc), f"render seed {seed}: film is not the sum of its samples"
        assert st["rays"] == int(nrays.sum())
        want, ost = ob.render(oc, spp, depth, bg)
        close = np.isclose(film, want, rtol=1e-9, atol=1e-12).all(axis=2).mean()
        assert close > 0.99, f"render seed {seed}: only {close:.4f} of the pixels agree with the oracle"
    finally:
        scene.close()
    return f"agree {close:.4f}, rays {st['rays']}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--width", type=int, default=96, help="of the renders (520 at 4 spp: more than a million paths, one batch runs the per-bounce launches)")
    ap.add_argument("--spp", type=int, default=6)
    ap.add_argument("--no-batches", action="store_true")
    a = ap.parse_args()
    t0 = time.time()
    for seed in range(a.first, a.first + (0 if a.no_batches else a.seeds)):
        print(f"[{time.time() - t0:6.1f} s] batch {seed}: {batch_case(seed)}", flush=True)
    for seed in range(a.first, a.first + a.seeds):
        print(f"[{time.time() - t0:6.1f} s] render {seed}: {render_case(seed, a.width, a.spp)}", flush=True)
    print("soak ok")


if __name__ == "__main__":
    main()
