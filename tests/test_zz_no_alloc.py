"""rt_intersect_batch_noalloc, the entry point for callers that cannot allocate device memory: rays and results in the
caller's device memory, the kernels' own needs in a static device array of the library.  Its results must be those of
rt_intersect_batch bit for bit -- on two calls in a row, the second smaller, so that whatever the first call left in
the shared array is seen not to matter -- and that the call makes no device allocation: the library counts its own
hipMalloc calls (rt_debug_device_allocs), and the device's free memory and torch's allocator do not move either."""
import ctypes as C

import numpy as np
import pytest

from helpers import SMALL, build_pair, cameras, path_rays
from raytracer_amd import _lib

pytestmark = pytest.mark.gpu


def _call(scene, torch, o, d, tmin, tmax, rng3, seed=42):
    n = o.shape[0]
    dev = f"cuda:{scene.device}"
    od6 = torch.from_numpy(np.ascontiguousarray(np.concatenate([o.T, d.T], axis=0))).to(dev)       # 6 x n
    tmin_d = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, np.float64), (n,)))).to(dev)
    tmax_d = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, np.float64), (n,)))).to(dev)
    rng = torch.from_numpy(np.ascontiguousarray(rng3.T.astype(np.int64))).to(dev).to(torch.int32)   # 3 x n u32 bit patterns
    out = {"hit": torch.zeros(n, dtype=torch.uint8, device=dev), "prim": torch.zeros(n, dtype=torch.int32, device=dev),
           "t": torch.zeros(n, dtype=torch.float64, device=dev), "p": torch.zeros((n, 3), dtype=torch.float64, device=dev),
           "n": torch.zeros((n, 3), dtype=torch.float64, device=dev), "uv": torch.zeros((n, 2), dtype=torch.float64, device=dev),
           "front": torch.zeros(n, dtype=torch.uint8, device=dev)}
    torch.cuda.synchronize()
    # three probes of "allocates nothing": the library's own count of its hipMalloc calls, the device's free memory
    # (hipMemGetInfo) and torch's allocator
    before = (_lib.amd().rt_debug_device_allocs(), torch.cuda.mem_get_info(scene.device)[0], torch.cuda.memory_allocated(scene.device))
    rc = _lib.amd().rt_intersect_batch_noalloc(scene.h, seed, od6.data_ptr(), tmin_d.data_ptr(), tmax_d.data_ptr(),
                                               rng.data_ptr(), n, out["hit"].data_ptr(), out["prim"].data_ptr(),
                                               out["t"].data_ptr(), out["p"].data_ptr(), out["n"].data_ptr(),
                                               out["uv"].data_ptr(), out["front"].data_ptr())
    assert rc == 0, _lib.last_error()
    after = (_lib.amd().rt_debug_device_allocs(), torch.cuda.mem_get_info(scene.device)[0], torch.cuda.memory_allocated(scene.device))
    assert after == before, (before, after)
    res = {k: v.cpu().numpy() for k, v in out.items()}
    res["prim"] = res["prim"].view(np.uint32)
    res["ctr"] = rng[2].cpu().numpy().view(np.uint32)
    return res


@pytest.mark.parametrize("cfg", ["C2", "C4", "C5"])
def test_noalloc_entry_point_equals_intersect_batch(cfg):
    import torch
    hs, os_ = build_pair(cfg, **SMALL[cfg])
    _, oc = cameras(hs)
    scene = hs.builder.commit(0)
    try:
        for count, bounce in ((20000, 1), (3000, 0)):     # two calls in a row, the second with fewer rays
            o, d, rng3 = path_rays(os_, oc, count, bounce)
            allocs = _lib.amd().rt_debug_device_allocs()
            want = scene.intersect(o, d, 0.001, np.inf, rng3)
            assert _lib.amd().rt_debug_device_allocs() > allocs    # (the probe does see the allocating entry point)
            got = _call(scene, torch, o, d, 0.001, np.inf, rng3)
            for k in ("hit", "prim", "front", "ctr", "t", "p", "n", "uv"):
                assert np.array_equal(got[k], want[k]), f"{cfg} bounce {bounce}: {k} differs"
            assert want["hit"].any()
    finally:
        scene.close()


def test_noalloc_entry_point_refuses_more_rays_than_its_scratch_holds():
    import torch
    hs, _ = build_pair("C3", **SMALL["C3"])
    scene = hs.builder.commit(0)
    try:
        n = 65537
        z = torch.zeros(6 * n, dtype=torch.float64, device="cuda:0")
        r = torch.zeros(3 * n, dtype=torch.int32, device="cuda:0")
        rc = _lib.amd().rt_intersect_batch_noalloc(scene.h, 42, z.data_ptr(), z.data_ptr(), z.data_ptr(), r.data_ptr(), n,
                                                   None, None, None, None, None, None, None)
        assert rc == -4 and "65536" in _lib.last_error()
    finally:
        scene.close()
