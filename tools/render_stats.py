#!/usr/bin/env python3
"""Prints the library's own statistics of one render (development aid).  usage: render_stats.py <config> <spp> [batch_paths]"""
import json
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import bench
from raytracer_amd import HostBuilder, camera_new, scenes

cfg, spp = sys.argv[1], int(sys.argv[2])
bp = int(sys.argv[3]) if len(sys.argv) > 3 else 0
key, width, _ = bench.WORKLOADS[cfg]
spec = scenes.build(key, HostBuilder(), width=width)
cam = camera_new(image_width=width, **spec.camera)
scene = spec.builder.commit(0)
film = torch.empty((cam.height, cam.width, 3), dtype=torch.float64, device="cuda:0")
for k in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = scene.render_into(cam, spp, spec.max_depth, spec.background, film.data_ptr(), batch_paths=bp)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(json.dumps({"config": cfg, "spp": spp, "wall_ms": dt * 1e3, "mrays_per_s": st["rays"] / dt / 1e6,
                  **{k: st[k] for k in ("ms_total", "ms_traverse", "ms_shade", "ms_finish", "traverse_launches", "finish_launches", "bounces", "rays", "paths", "list_rays")}}))
