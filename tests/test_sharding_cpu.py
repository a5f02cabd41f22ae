"""The multi-GPU path on the CPU: two to four processes, gloo backend, the oracle standing in for the GPU renderer.  Checks
that the sample ranges tile [0, spp) and that the all-reduced film of the shards equals the film of one process
rendering every sample (the random stream of a sample depends on its global index only)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT
from raytracer_amd.dist import render_sharded, shard_range


def test_shard_ranges_tile_the_samples():
    for spp in (1, 2, 7, 200, 1000, 10000):
        for world in (1, 2, 3, 4, 8):
            covered = []
            for r in range(world):
                b, c = shard_range(spp, r, world)
                covered.extend(range(b, b + c))
            assert covered == list(range(spp)), (spp, world)
            counts = [shard_range(spp, r, world)[1] for r in range(world)]
            assert max(counts) - min(counts) <= 1


def _worker(rank, world, port, spp, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as orc
    from raytracer_amd import scenes
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec = scenes.build("C2", orc.OracleBuilder(), width=24)
        cam = orc.camera_new(image_width=24, **spec.camera)
        film = torch.zeros((cam.height, cam.width, 3), dtype=torch.float64)

        def render_range(begin, count):
            f, _ = spec.builder.render(cam, count, spec.max_depth, spec.background, spp_begin=begin, threads=2)
            film.copy_(torch.from_numpy(f))

        begin, count = render_sharded(render_range, film, spp)
        assert (begin, count) == shard_range(spp, rank, world)
        if rank == 0:
            np.save(out_path, film.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world, spp", [(2, 5), (3, 7), (4, 6), (4, 3)])
def test_shards_sum_to_the_single_process_film(tmp_path, world, spp):
    """Sample counts that do not divide by the number of ranks (2: 3 + 2; 3: 3 + 2 + 2; 4: 2 + 2 + 1 + 1), and more
    ranks than samples (4 ranks, 3 samples: the last rank renders nothing and contributes a zero film)."""
    out = str(tmp_path / "film.npy")
    port = 29500 + (os.getpid() % 2000) + 7 * world + spp
    mp.spawn(_worker, args=(world, port, spp, out), nprocs=world, join=True)
    sharded = np.load(out)
    from oracle import oracle as orc
    from raytracer_amd import scenes
    spec = scenes.build("C2", orc.OracleBuilder(), width=24)
    cam = orc.camera_new(image_width=24, **spec.camera)
    whole, _ = spec.builder.render(cam, spp, spec.max_depth, spec.background)
    assert sharded.shape == whole.shape and whole.sum() > 0
    # identical paths; only the association of the sum differs ((s0+s1+s2)+(s3+s4) vs ((((s0+s1)+s2)+s3)+s4))
    np.testing.assert_allclose(sharded, whole, rtol=1e-13, atol=1e-300)
    assert np.array_equal(orc.tonemap(sharded, spp), orc.tonemap(whole, spp))
