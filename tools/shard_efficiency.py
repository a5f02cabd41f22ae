#!/usr/bin/env python3
"""What strong scaling over N GPUs can reach at best, measured on ONE GPU: a rank of an N-GPU job renders spp / N of the
samples of every pixel, so the job takes as long as its slowest rank, t(spp / N), and the efficiency against one GPU
rendering everything is at most t(spp) / (N * t(spp / N)) -- what is lost is the fixed cost of a render (set-up, the
launches of the nearly empty last bounces, the film reduction) that every rank pays in full.  The all-reduce of the
films over the links is NOT in this figure; bench.py --gpus N --scaling strong on a multi-GPU node measures the whole.

usage (on the GPU box, from the repository root):  python3 tools/shard_efficiency.py [--out profiles/<name>.json]
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--configs", default="C2,C3,C4,C5")
    ap.add_argument("--ranks", default="2,4,8")
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--c5-spp", type=int, default=1000, help="samples per pixel for C5 (its 10000 take minutes)")
    args = ap.parse_args()
    import torch
    from raytracer_amd import HostBuilder, camera_new, scenes
    from raytracer_amd.dist import shard_range
    import bench
    rows = []
    for config in args.configs.split(","):
        key, width, spp = bench.WORKLOADS[config]
        if config == "C5":
            spp = args.c5_spp
        spec = scenes.build(key, HostBuilder(), width=width)
        cam = camera_new(image_width=width, **spec.camera)
        scene = spec.builder.commit(0)
        film = torch.empty((cam.height, cam.width, 3), dtype=torch.float64, device="cuda:0")

        def timed(begin, count):
            ts = []
            for _ in range(args.repeats + 1):  # the first run is a warm-up
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                scene.render_into(cam, count, spec.max_depth, spec.background, film.data_ptr(), spp_begin=begin)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            return statistics.median(ts[1:])

        whole = timed(0, spp)
        row = {"config": config, "scene": spec.name, "width": cam.width, "height": cam.height, "spp": spp,
               "seconds_one_gpu": whole, "ranks": {}}
        for n in [int(x) for x in args.ranks.split(",")]:
            # the slowest rank is one with the larger share; rank 0 has it (shard_range gives the remainder to the first ranks)
            begin, count = shard_range(spp, 0, n)
            t = timed(begin, count)
            row["ranks"][str(n)] = {"spp_of_rank_0": count, "seconds_rank_0": t, "efficiency_bound": whole / (n * t),
                                    "speedup_bound": whole / t}
        rows.append(row)
        print(json.dumps(row), flush=True)
        scene.close()
    out = {"what": "upper bound of the strong-scaling efficiency of the sample-sharded render, t(spp) / (N t(spp / N)), "
                   "measured on one MI355X (median of %d runs each); the film all-reduce is not included" % args.repeats,
           "rows": rows}
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
