#!/usr/bin/env python3
"""Benchmark of the hot path: one step = one full render (ray generation -> [traverse+intersect -> shade+compact] x
depth -> film accumulation, plus the film all-reduce when N > 1) of a BASELINE.json configuration.

  python bench.py --gpus 1 --steps K --warmup W            # default workload: configs[1] (cornell_smoke 600px 200spp)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Weak scaling: every rank renders the configuration's full sample count on its own range of global sample indices
(rank r renders samples [r*spp, (r+1)*spp)), then the f64 films are summed with one RCCL all-reduce; the job has
rendered N*spp samples per pixel.  Rank 0 prints one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {  # BASELINE.json configs -> (scene function key, width, spp)
    "C1": ("C1", 400, 10), "C2": ("C2", 600, 200), "C3": ("C3", 600, 1000), "C4": ("C4", 800, 2000), "C5": ("C5", 800, 10000),
}
S_NODE, S_STATE, S_HIT = 64, 88, 56  # algorithmic bytes: BVH entry, path state, hit record
HBM_PEAK_GBS = 8000.0
FP64_PEAK_TFLOPS = 78.6


def cpu_baseline(config, width, max_seconds=20.0):
    """Times the CPU oracle (C++ restatement of the reference, a thread pool over rows standing in for Rayon) on a
    bounded sample of the same workload: same scene and resolution, fewer samples per pixel."""
    from oracle import oracle as orc
    from raytracer_amd import scenes
    spec = scenes.build(config, orc.OracleBuilder(), width=width)
    cam = orc.camera_new(image_width=width, **spec.camera)
    cores = min(16, os.cpu_count() or 1)
    t0 = time.time()
    _, st = spec.builder.render(cam, 1, spec.max_depth, spec.background, threads=cores)
    dt = time.time() - t0
    spp = 1
    if dt < max_seconds / 3:
        spp = max(1, min(64, int(max_seconds / max(dt, 1e-3)) - 1))
        t0 = time.time()
        _, st = spec.builder.render(cam, spp, spec.max_depth, spec.background, spp_begin=1, threads=cores)
        dt = time.time() - t0
    return {"value": st["rays"] / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "mpaths_per_s": st["paths"] / dt / 1e6,
            "sample": f"{spec.name} {cam.width}x{cam.height}, {spp} spp of the workload's samples, depth {spec.max_depth}, "
                      f"{st['rays']} rays in {dt:.1f} s; C++ restatement of the reference, not Rayon"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C2", help="a BASELINE.json configuration (C1..C5, the metric's workloads) or the "
                    "name of another scene program of the reference, run at the reference's own settings")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel per GPU (invalidates the metric)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--batch-paths", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--whole-paths", action="store_true", help="RT_FLAG_WHOLE_PATHS: one kernel per batch (BVH-free scenes)")
    ap.add_argument("--queues", action="store_true", help="RT_FLAG_QUEUES: the queue-based schedule whatever the scene")
    ap.add_argument("--split-bounce", action="store_true", help="RT_FLAG_SPLIT_BOUNCE: separate traverse and shade launches")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the only rendering path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from raytracer_amd import HostBuilder, camera_new, scenes
    if args.config in WORKLOADS:
        key, width, spp = WORKLOADS[args.config]
    elif args.config in scenes.SCENES:
        probe = scenes.build(args.config, HostBuilder())
        key, width, spp = args.config, probe.width, probe.spp
        probe.builder.close()
    else:
        raise SystemExit(f"unknown --config {args.config}: one of {sorted(WORKLOADS)} or {sorted(scenes.SCENES)}")
    width = args.width or width
    spp = args.spp or spp
    spec = scenes.build(key, HostBuilder(), width=width)
    cam = camera_new(image_width=width, **spec.camera)
    scene = spec.builder.commit(local_rank)
    film = torch.empty((cam.height, cam.width, 3), dtype=torch.float64, device=f"cuda:{local_rank}")

    from raytracer_amd.dist import render_sharded

    schedule = True if args.whole_paths else (False if args.queues else None)

    def step():
        st = {}

        def render_range(begin, count):
            st.update(scene.render_into(cam, count, spec.max_depth, spec.background, film.data_ptr(), spp_begin=begin,
                                        batch_paths=args.batch_paths, whole_paths=schedule,
                                        split_bounce=args.split_bounce))

        render_sharded(render_range, film, spp * world)  # rank r renders samples [r*spp, (r+1)*spp), then RCCL all-reduce
        return st

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        stats.append(step())
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([sum(s["rays"] for s in stats), sum(s["paths"] for s in stats)], dtype=torch.float64,
                           device=f"cuda:{local_rank}")
        dist.all_reduce(tot)
        total_rays, total_paths = float(tot[0].item()), float(tot[1].item())
    else:
        total_rays, total_paths = float(sum(s["rays"] for s in stats)), float(sum(s["paths"] for s in stats))

    if rank == 0:
        # per-ray algorithmic work of this workload, from the device counters (one untimed pass at reduced spp)
        # with the cull boxes off, i.e. the box and primitive tests the reference's algorithm performs; the timed
        # steps skip part of that work (cull boxes), which is reported beside it
        cst = scene.render_into(cam, min(spp, 4), spec.max_depth, spec.background, film.data_ptr(), counters=True,
                                cull=False)
        n_node = cst["node_tests"] / cst["rays"]
        prim_bytes = cst["prim_bytes"] / cst["rays"]
        dst = scene.render_into(cam, min(spp, 4), spec.max_depth, spec.background, film.data_ptr(), counters=True)
        done = {"node_visits_per_ray": dst["node_tests"] / dst["rays"], "cull_tests_per_ray": dst["cull_tests"] / dst["rays"],
                "prim_tests_per_ray": dst["prim_tests"] / dst["rays"], "prim_bytes_per_ray": dst["prim_bytes"] / dst["rays"],
                "reference_prim_tests_per_ray": cst["prim_tests"] / cst["rays"]}
        # algorithmic flops per ray: 18 per box test, the primitive tests as counted, ~100 for shading one bounce
        flops_ray = 18.0 * n_node + cst["prim_flops"] / cst["rays"] + 100.0
        b_trav = S_NODE * n_node + prim_bytes + S_STATE + S_HIT        # traverse+intersect launch
        b_ray = S_NODE * n_node + prim_bytes + 2 * S_STATE + 2 * S_HIT  # whole bounce
        rays_local = sum(s["rays"] for s in stats)
        fused = sum(s["path_launches"] for s in stats) > 0
        if fused:  # whole paths in one kernel: the dominant kernel does the whole bounce
            kernel = "k_paths_lockstep"
            launches, ms_trav, b_kernel = sum(s["path_launches"] for s in stats), \
                sum(s["ms_paths"] for s in stats), b_ray
        elif sum(s["bounce_launches"] for s in stats) > 0:  # traverse + shade of one bounce in one kernel
            kernel = "k_bounce_lockstep"
            launches, ms_trav, b_kernel = sum(s["bounce_launches"] for s in stats), \
                sum(s["ms_bounce"] for s in stats), b_ray
        else:
            kernel = "k_traverse_lockstep" if stats[0]["lockstep"] else "k_traverse_persistent"
            launches, ms_trav, b_kernel = sum(s["traverse_launches"] for s in stats), \
                sum(s["ms_traverse"] for s in stats), b_trav
        if launches == 0:  # so small a workload that the tail kernel took all of it
            kernel = "k_finish"
            launches, ms_trav, b_kernel = sum(s["finish_launches"] for s in stats), sum(s["ms_finish"] for s in stats), b_ray
        avg_launch_ms = ms_trav / launches
        achieved = (rays_local / launches) * b_kernel / (avg_launch_ms * 1e-3) / 1e9
        traffic = None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):
            with open(tj) as fh:
                traffic = json.load(fh).get(args.config, {}).get(kernel + "_hbm_bytes_per_launch")
        out = {
            "metric": "Mrays/s", "value": total_rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "mpaths_per_s": total_paths / elapsed / 1e6,
            "config": {"workload": f"{args.config} {spec.name} {cam.width}x{cam.height}, {spp} spp per GPU, depth {spec.max_depth}",
                       "spp_total": spp * world, "paths_per_step": int(total_paths / args.steps), "seed": 42,
                       "parallelism": f"spp-shard x{world}" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "bytes_per_ray_kernel": b_kernel, "bytes_per_ray_traverse": b_trav, "bytes_per_ray_bounce": b_ray, "node_tests_per_ray": n_node,
                         "prim_bytes_per_ray": prim_bytes, "avg_launch_ms": avg_launch_ms, "launches": launches,
                         "traverse_share_of_step": ms_trav / (elapsed * 1e3),
                         "shade_ms_per_step": sum(s["ms_shade"] for s in stats) / args.steps,
                         "work_done_with_cull_boxes": done},
            # the bound that actually binds (all scenes are cache-resident): whole-step algorithmic flops against
            # the FP64 vector peak, which counts an FMA as two flops although contraction is not allowed here
            "roofline_fp64": {"bound": "fp64 valu", "achieved": total_rays / elapsed * flops_ray / 1e12 / world,
                              "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s per GPU",
                              "frac": total_rays / elapsed * flops_ray / 1e12 / world / FP64_PEAK_TFLOPS,
                              "flops_per_ray": flops_ray},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(key, width)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
