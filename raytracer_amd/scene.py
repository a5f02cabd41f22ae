"""A scene committed to one GPU, and the calls of the C ABI (include/rt_amd.h) that run on it."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import RtError, rt_render_opts, rt_stats


def _opts(spp_begin, spp_count, max_depth, seed, background, batch_paths=0, flags=0):
    o = rt_render_opts()
    o.spp_begin, o.spp_count, o.max_depth, o.flags = spp_begin, spp_count, max_depth, flags
    o.seed = seed
    o.background[0], o.background[1], o.background[2] = [float(x) for x in background]
    o.batch_paths = batch_paths
    return o


def _flags(counters, cull, whole_paths, split_bounce=False, fast_bvh=False):
    """whole_paths: None = the library's choice, True / False = force one kernel per batch / the queue schedule;
    split_bounce: a traverse and a shade launch per bounce where the queue schedule would use one bounce kernel."""
    return ((_lib.RT_FLAG_COUNTERS if counters else 0) | (0 if cull else _lib.RT_FLAG_NO_CULL) |
            (0 if whole_paths is None else (_lib.RT_FLAG_WHOLE_PATHS if whole_paths else _lib.RT_FLAG_QUEUES)) |
            (_lib.RT_FLAG_SPLIT_BOUNCE if split_bounce else 0) | (_lib.RT_FLAG_FAST_BVH if fast_bvh else 0))


def _stats_dict(st):
    return {k: getattr(st, k) for k, _ in rt_stats._fields_}


class Scene:
    def __init__(self, handle, device):
        self.L = _lib.amd()
        self.h = handle
        self.device = device

    def close(self):
        if self.h:
            self.L.rt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise RtError(f"rt_amd error {rc}: {_lib.last_error()}")

    def intersect(self, o, d, tmin, tmax, rng3=None, seed=42):
        """Hittable::hit on a batch of rays -> dict(hit, prim, t, p, n, uv, front, ctr)."""
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
        self._ck(self.L.rt_intersect_batch(self.h, seed, o.ctypes.data, d.ctypes.data, tmin.ctypes.data,
                                           tmax.ctypes.data, rng3.ctypes.data if rng3 is not None else None, n,
                                           out["hit"].ctypes.data, out["prim"].ctypes.data, out["t"].ctypes.data,
                                           out["p"].ctypes.data, out["n"].ctypes.data, out["uv"].ctypes.data,
                                           out["front"].ctypes.data, out["ctr"].ctypes.data))
        return out

    def render(self, cam, spp, max_depth, background, seed=42, spp_begin=0, batch_paths=0, counters=False,
               cull=True, whole_paths=None, split_bounce=False, fast_bvh=False):
        """Camera::render: returns (film of radiance SUMS, shape (H, W, 3), stats dict).  cull=False switches the
        cull boxes off: the same film from exactly the box and primitive tests the reference performs.
        whole_paths=True traces whole paths in one kernel (scenes without BVH entries only), False forces the
        queue-based schedule, None leaves the choice to the library; split_bounce=True makes the queue-based schedule
        of a scene without BVH entries use a traverse and a shade launch per bounce instead of its one bounce kernel."""
        film = np.zeros((cam.height, cam.width, 3))
        st = rt_stats()
        o = _opts(spp_begin, spp, max_depth, seed, background, batch_paths, _flags(counters, cull, whole_paths, split_bounce, fast_bvh))
        self._ck(self.L.rt_render(self.h, C.byref(cam), C.byref(o), film.ctypes.data, C.byref(st)))
        return film, _stats_dict(st)

    def render_into(self, cam, spp, max_depth, background, film_device_ptr, seed=42, spp_begin=0, batch_paths=0,
                    counters=False, cull=True, whole_paths=None, split_bounce=False, fast_bvh=False):
        """Same, into a caller-owned device buffer of H*W*3 f64 (e.g. a torch tensor's data_ptr())."""
        st = rt_stats()
        o = _opts(spp_begin, spp, max_depth, seed, background, batch_paths, _flags(counters, cull, whole_paths, split_bounce, fast_bvh))
        self._ck(self.L.rt_render_device(self.h, C.byref(cam), C.byref(o), film_device_ptr, C.byref(st)))
        return _stats_dict(st)

    def set_fast_bvh(self, on):
        """RT_FLAG_FAST_BVH for every later call on this scene (rt_intersect_batch takes no flags)."""
        self._ck(self.L.rt_scene_set_fast_bvh(self.h, 1 if on else 0))

    def trace_samples(self, cam, pixel, sample, max_depth, background, seed=42):
        pixel = np.ascontiguousarray(pixel, dtype=np.uint32)
        sample = np.ascontiguousarray(sample, dtype=np.uint32)
        n = pixel.size
        rad = np.zeros((n, 3))
        nr = np.zeros(n, np.uint32)
        o = _opts(0, 1, max_depth, seed, background)
        self._ck(self.L.rt_trace_samples(self.h, C.byref(cam), C.byref(o), n, pixel.ctypes.data, sample.ctypes.data,
                                         rad.ctypes.data, nr.ctypes.data))
        return rad, nr

    def tonemap(self, film, spp):
        film = np.ascontiguousarray(film, dtype=np.float64)
        h, w, _ = film.shape
        rgb = np.zeros((h, w, 3), np.uint8)
        self._ck(self.L.rt_tonemap_rgb8(self.h, film.ctypes.data, w, h, spp, rgb.ctypes.data))
        return rgb


def render_multi(scenes, cam, spp, max_depth, background, seed=42, spp_begin=0, batch_paths=0):
    """rt_render_multi: one process, scenes[r] committed to GPU r; the samples are split over them and the shard films
    added in rank order.  Returns (film of radiance sums (H, W, 3), stats summed over the shards)."""
    film = np.zeros((cam.height, cam.width, 3))
    st = rt_stats()
    o = _opts(spp_begin, spp, max_depth, seed, background, batch_paths, 0)
    handles = (C.c_void_p * len(scenes))(*[s.h for s in scenes])
    rc = _lib.amd().rt_render_multi(handles, len(scenes), C.byref(cam), C.byref(o), film.ctypes.data, C.byref(st))
    if rc != 0:
        raise RtError(f"rt_amd error {rc}: {_lib.last_error()}")
    return film, _stats_dict(st)


def debug_arith(a, b, device=0):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    n = a.size
    q, r, c, l = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    rc = _lib.amd().rt_debug_arith(device, a.ctypes.data, b.ctypes.data, n, q.ctypes.data, r.ctypes.data,
                                   c.ctypes.data, l.ctypes.data)
    if rc != 0:
        raise RtError(_lib.last_error())
    return q, r, c, l
