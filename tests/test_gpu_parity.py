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
def test_wavefront_film_equals_sum_of_samples(gpu_scenes, cfg):
    """rt_render (wavefront queues + compaction + film accumulation) must give, bit for bit, the in-order sum of the
    per-sample radiances rt_trace_samples computes with one lane per whole path."""
    hs, os_, scene = gpu_scenes(cfg)
    hc, _ = cameras(hs)
    spp = 3
    film, stats = scene.render(hc, spp, hs.max_depth, hs.background, spp_begin=5, batch_paths=2 * hc.width * hc.height)
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
    film, st = scene.render(hc, 1, 1, hs.background, counters=True)  # camera rays only: no transcendental involved
    _, ost = os_.builder.render(oc, 1, 1, hs.background)
    assert st["rays"] == ost["rays"]
    assert st["node_tests"] == ost["node_tests"]
    assert st["prim_tests"] == ost["prim_tests"]
