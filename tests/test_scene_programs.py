"""The reference's other scene programs (src/main.rs:1373-1397 lists all eighteen entries; five are the benchmark
configurations the other test files cover): every one of them described through the builder interface, walked by the
device functions on the CPU (tests/hostsim) and by the HIP kernels, and compared with the oracle like the benchmark
scenes are."""
import numpy as np
import pytest

from helpers import HostSim, assert_hits_equal, build_pair, cameras, path_rays
from raytracer_amd import scenes

EXTRA = [n for n in scenes.SCENES if scenes.SCENES[n] not in scenes.CONFIGS.values()]
SMALL_KW = dict(width=48, spp=2)


def test_every_scene_program_of_the_reference_is_present():
    # main.rs:1378-1395, in the order of its match arms (final_scene appears three times with other arguments)
    assert list(scenes.SCENES) == [
        "many_spheres", "checkered_spheres", "earth", "perlin_spheres", "para", "simple_light", "cornell_box",
        "cornell_smoke", "final_scene", "cornell_quadric", "debug_quadric", "cornell_triangle", "profiler_scene",
        "cornell_model", "spherical_mirror", "debug_model", "bust"]
    assert len(EXTRA) == 12


def test_defaults_are_the_reference_settings():
    from oracle import oracle as orc
    want = {  # name: (width, spp, max_depth, aspect, vfov)   -- main.rs, the let-bindings of each function
        "checkered_spheres": (400, 100, 50, 16 / 9, 20.0), "earth": (400, 100, 50, 16 / 9, 20.0),
        "perlin_spheres": (400, 100, 50, 16 / 9, 20.0), "para": (400, 100, 50, 1.0, 80.0),
        "simple_light": (400, 100, 50, 16 / 9, 20.0), "cornell_box": (600, 20, 50, 1.0, 40.0),
        "debug_quadric": (600, 50, 50, 1.0, 40.0), "cornell_triangle": (300, 20, 4, 1.0, 40.0),
        "profiler_scene": (400, 20, 4, 1.0, 40.0), "spherical_mirror": (800, 800, 50, 1.0, 40.0),
        "debug_model": (600, 20, 4, 16 / 9, 40.0), "bust": (600, 20, 4, 16 / 9, 40.0)}
    for name in EXTRA:
        spec = scenes.build(name, orc.OracleBuilder())
        got = (spec.width, spec.spp, spec.max_depth, spec.camera["aspect_ratio"], spec.camera["vfov"])
        assert got == want[name], name


@pytest.fixture(scope="module")
def pairs():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = build_pair(name, **SMALL_KW)
        return cache[name]

    return get


def _has_media(hs):
    return len(hs.builder.flat("media")) > 0


def _has_spheres(hs):
    return len(hs.builder.flat("spheres")) > 0


@pytest.mark.parametrize("name", EXTRA)
def test_device_functions_on_cpu_match_oracle(pairs, name):
    hs, os_ = pairs(name)
    sim = HostSim(hs.builder)
    hc, oc = cameras(hs)
    for bounce in (0, 1):
        o, d, rng3 = path_rays(os_, oc, 1200, bounce)
        if len(o) == 0:
            continue
        want = os_.builder.intersect(o, d, 0.001, np.inf, rng3)
        got = sim.intersect(o, d, 0.001, np.inf, rng3)
        assert_hits_equal(got, want, exact=True, what=f"{name} bounce {bounce}")
    rs = np.random.RandomState(5)
    n = 400
    pixel = rs.randint(0, hc.width * hc.height, n).astype(np.uint32)
    sample = rs.randint(0, 50, n).astype(np.uint32)
    want, want_rays = os_.builder.trace_samples(oc, pixel, sample, hs.max_depth, hs.background)
    got, got_rays = sim.trace_samples(hc, pixel, sample, hs.max_depth, hs.background)
    assert np.array_equal(got_rays, want_rays)
    np.testing.assert_allclose(got, want, rtol=1e-13, atol=0)


@pytest.fixture(scope="module")
def gpu_scenes(pairs):
    cache = {}

    def get(name):
        if name not in cache:
            hs, os_ = pairs(name)
            cache[name] = (hs, os_, hs.builder.commit(0))
        return cache[name]

    yield get
    for _, _, sc in cache.values():
        sc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", EXTRA)
def test_gpu_intersections_match_oracle(gpu_scenes, name):
    hs, os_, scene = gpu_scenes(name)
    _, oc = cameras(hs)
    for bounce in (0, 2):
        o, d, rng3 = path_rays(os_, oc, 8000, bounce)
        if len(o) == 0:
            continue
        want = os_.builder.intersect(o, d, 0.001, np.inf, rng3)
        got = scene.intersect(o, d, 0.001, np.inf, rng3)
        if not _has_media(hs):
            assert_hits_equal(got, want, exact=True, what=f"{name} bounce {bounce}", uv_exact=not _has_spheres(hs))
            continue
        for k in ("hit", "prim", "ctr", "front"):
            assert np.array_equal(got[k], want[k]), k
        medium_ids = np.array([m.prim_id for m in hs.builder.flat("media")], dtype=np.uint32)
        is_medium = np.isin(want["prim"], medium_ids)
        for k in ("t", "p", "n"):
            assert np.array_equal(got[k][~is_medium], want[k][~is_medium]), k
            np.testing.assert_allclose(got[k][is_medium], want[k][is_medium], rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(got["uv"], want["uv"], rtol=0, atol=1e-15)


@pytest.mark.gpu
@pytest.mark.parametrize("name", EXTRA)
def test_gpu_radiance_and_film_match_oracle(gpu_scenes, name):
    hs, os_, scene = gpu_scenes(name)
    hc, oc = cameras(hs)
    npix = hc.width * hc.height
    spp = 2
    pixel = np.tile(np.arange(npix, dtype=np.uint32), spp)
    sample = np.repeat(np.arange(spp, dtype=np.uint32), npix)
    want, want_rays = os_.builder.trace_samples(oc, pixel, sample, hs.max_depth, hs.background)
    got, got_rays = scene.trace_samples(hc, pixel, sample, hs.max_depth, hs.background)
    same_len = got_rays == want_rays
    assert same_len.mean() > 0.995, f"{(~same_len).sum()} of {len(pixel)} paths took a different number of bounces"
    np.testing.assert_allclose(got[same_len], want[same_len], rtol=1e-9, atol=1e-12)
    # the film of the default schedule is the in-order sum of exactly those samples, and the cull boxes are invisible
    film, stats = scene.render(hc, spp, hs.max_depth, hs.background)
    acc = np.full((npix, 3), -0.0)
    for s in range(spp):
        acc = acc + got[s * npix:(s + 1) * npix]
    assert np.array_equal(film.reshape(npix, 3), acc)
    assert stats["rays"] == int(got_rays.sum())
    film_nc, _ = scene.render(hc, spp, hs.max_depth, hs.background, cull=False)
    assert np.array_equal(film, film_nc)
