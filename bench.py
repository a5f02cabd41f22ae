#!/usr/bin/env python3
"""Benchmark of the hot path: one step = one full render (ray generation -> [traverse+intersect -> shade+compact] x
depth -> film accumulation, plus the film all-reduce when N > 1) of a BASELINE.json configuration.

  python bench.py --gpus 1 --steps K --warmup W            # default workload: configs[1] (cornell_smoke 600px 200spp)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Scaling.  Default is weak: every rank renders the configuration's full sample count on its own range of global sample
indices (rank r renders samples [r*spp, (r+1)*spp)), so the job has rendered N*spp samples per pixel.  `--scaling
strong` keeps the configuration's sample count for the whole job and gives rank r its contiguous share of it
(raytracer_amd.dist.shard_range; the reference's precedent is thrender_no_rayon, camera.rs:319-328).  Either way the
f64 films are summed with one RCCL all-reduce and rank 0 prints one JSON line.

At N = 1 the line also carries, under "also", the same measurement for the BVH workloads the headline configuration
does not exercise (C4 and C5 at their full sizes), each timed after the headline's timed region, with its own roofline
and CPU baseline; their rates are repeated as flat keys (c4_mrays_per_s, ...) of the line."""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {  # BASELINE.json configs -> (scene function key, width, spp)
    "C1": ("C1", 400, 10), "C2": ("C2", 600, 200), "C3": ("C3", 600, 1000), "C4": ("C4", 800, 2000), "C5": ("C5", 800, 10000),
}
S_NODE, S_STATE, S_HIT = 64, 88, 56  # bytes of a BVH entry, a path state, a hit record in the reference's terms
HBM_PEAK_GBS = 8000.0
FP64_PEAK_TFLOPS = 78.6   # vector FP64 peak of one MI355X (an FMA counted as two flops)
FLOPS_BOX, FLOPS_SHADE = 18.0, 100.0  # algorithmic flops of one slab test / of shading one bounce


def host_cpu():
    """(model name, logical CPUs of the machine, CPUs this process may run on)."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = logical
    return model, logical, usable


def physical_cores():
    """Distinct (physical id, core id) pairs of /proc/cpuinfo; None where the file does not say."""
    seen, phys, core = set(), None, None
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if core is not None:
                        seen.add((phys, core))
                    phys = core = None
        if core is not None:
            seen.add((phys, core))
    except OSError:
        return None
    return len(seen) or None


def launch_ranks(n, argv):
    """`bench.py --gpus N` outside a launcher: starts N ranks (one process per GPU) with torch.distributed.run as a
    child process and returns its exit code.  Runs before anything in this process has touched the GPU
    (torch.cuda.device_count() does not initialise it); rank 0 of the child prints the JSON line."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if n > have:
        print(f"bench.py: --gpus {n} asked for, {have} GPU(s) visible: refusing to measure fewer GPUs than asked for",
              file=sys.stderr, flush=True)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd)


def reduce_over_ranks(elapsed, rays, paths, dist, device):
    """The job's time is the slowest rank's (MAX over ranks), its rays and paths the sum over ranks.  `dist` is
    torch.distributed with an initialised process group, or None for a single process."""
    if dist is None:
        return elapsed, rays, paths
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tot = torch.tensor([rays, paths], dtype=torch.float64, device=device)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    return float(t.item()), float(tot[0].item()), float(tot[1].item())


def cpu_baseline(config, width, max_seconds=18.0):
    """Times the CPU oracle (C++ restatement of the reference, a thread pool over rows standing in for Rayon) on a
    bounded sample of the same workload: same scene and resolution, fewer samples per pixel.  All CPUs this process may
    run on (median of three runs), and one thread beside it."""
    from oracle import oracle as orc
    from raytracer_amd import scenes
    spec = scenes.build(config, orc.OracleBuilder(), width=width)
    cam = orc.camera_new(image_width=width, **spec.camera)
    model, logical, usable = host_cpu()
    threads = max(1, usable)

    def timed(spp, begin, nthreads):
        t0 = time.time()
        _, st = spec.builder.render(cam, spp, spec.max_depth, spec.background, spp_begin=begin, threads=nthreads)
        return st, time.time() - t0

    st, dt = timed(1, 0, threads)                      # sizes the sample: about a quarter of the budget per run
    spp = max(1, min(64, int(max_seconds / 4.0 / max(dt, 1e-3))))
    runs = []
    for k in range(3):
        st, dt = timed(spp, 1 + k * spp, threads)
        runs.append((st["rays"] / dt / 1e6, st["paths"] / dt / 1e6, st["rays"], dt))
    runs.sort()
    rate, prate, rays, dt = runs[1]
    # one thread: a row band's worth of the same image is not available through the oracle's interface, so the whole
    # image at 1 spp when that fits the budget, else reported as not measured
    one = None
    if dt / spp * threads < max_seconds:               # estimated single-thread time of 1 spp
        st1, dt1 = timed(1, 0, 1)
        one = st1["rays"] / dt1 / 1e6
    return {"value": rate, "unit": "Mrays/s", "cores": threads, "kind": "port", "mpaths_per_s": prate,
            "value_1thread": one, "cpu_model": model, "host_logical_cpus": logical, "host_physical_cores": physical_cores(),
            "usable_cpus": usable,
            "runs_mrays_per_s": [r[0] for r in runs],
            "sample": f"{spec.name} {cam.width}x{cam.height}, {spp} spp of the workload's samples, depth {spec.max_depth}, "
                      f"{rays} rays in {dt:.1f} s on {threads} threads (median of 3); C++ restatement of the reference, not Rayon"}


def profiled(config):
    """What profiles/traffic.json holds for this workload (written by tools/make_traffic.py from rocprofv3 passes of
    this same command): HBM bytes per ray of the dominant kernel and the profiled VALU issue account."""
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tj):
        return {}
    with open(tj) as fh:
        return json.load(fh).get(config, {})


def fast_bvh_report(scene, cam, spec, film, spp, parity_rate):
    """RT_FLAG_FAST_BVH (nearest-first traversal, not the default): its rate on the same workload at a bounded sample
    count, and how often it reports another primitive than the parity mode -- on camera rays through random pixels and
    on rays scattered from their hit points, through rt_intersect_batch."""
    import numpy as np
    rs = np.random.RandomState(7)
    n = 200_000
    px = rs.uniform(0, cam.width, n)
    py = rs.uniform(0, cam.height, n)
    c = np.array(cam.center[:])
    d = (np.array(cam.pixel00[:])[None, :] + px[:, None] * np.array(cam.delta_u[:])[None, :] +
         py[:, None] * np.array(cam.delta_v[:])[None, :]) - c[None, :]
    o = np.repeat(c[None, :], n, axis=0)
    total = differing = 0
    for _ in range(2):
        scene.set_fast_bvh(False)
        a = scene.intersect(o, d, 0.001, np.inf)
        scene.set_fast_bvh(True)
        b = scene.intersect(o, d, 0.001, np.inf)
        scene.set_fast_bvh(False)
        total += len(o)
        differing += int((a["prim"] != b["prim"]).sum())
        hit = a["hit"] == 1
        if not hit.any():
            break
        v = rs.standard_normal((int(hit.sum()), 3))
        o, d = a["p"][hit], a["n"][hit] + v / np.linalg.norm(v, axis=1, keepdims=True)
    sub = max(1, min(spp, 200))
    scene.render_into(cam, sub, spec.max_depth, spec.background, film.data_ptr(), fast_bvh=True)  # warm-up
    t0 = time.perf_counter()
    st = scene.render_into(cam, sub, spec.max_depth, spec.background, film.data_ptr(), fast_bvh=True)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    st0 = scene.render_into(cam, sub, spec.max_depth, spec.background, film.data_ptr())
    dt0 = time.perf_counter() - t0
    return {"flag": "RT_FLAG_FAST_BVH (opt-in, not the reference's tie rule)", "prim_id_mismatch_rate": differing / max(total, 1),
            "rays_compared": total, "value": st["rays"] / dt / 1e6, "unit": "Mrays/s", "spp": sub,
            "parity_mode_same_spp": st0["rays"] / dt0 / 1e6, "speedup": (st["rays"] / dt) / (st0["rays"] / dt0)}


SECTOR = 64  # bytes the memory system moves for a lone store (a finished path's radiance, written at the path's own slot)
LUM_SECTORS = 1.25  # the 24 bytes of a radiance start at a multiple of 24: two of every eight slots straddle two sectors


def traffic_model(kernel, stats, rays, paths):
    """HBM bytes per ray of the dominant kernel's launches, from the layout of what they read and write (the scene itself
    stays in cache): the queue holds 96 B per ray (origin, direction, throughput: 9 f64; pixel, sample, slot, birth,
    draw index, signature: 6 u32), a closest hit 16 B (t, reference, chain), a work-list entry 4 B, a finished path's
    radiance 3 consecutive f64 stored at its own slot -- a lone store of one sector, sometimes two.  Every ray read was written once (camera
    rays by k_raygen), so rays - paths rays are written by the bounce / shade launches."""
    if not rays:
        return None
    out_rays = rays - paths
    if kernel.startswith("k_bounce_lockstep"):
        total = 96.0 * rays + 96.0 * out_rays + LUM_SECTORS * SECTOR * paths
        what = "96 B read per ray + 96 B written per ray that goes on + 1.25 sectors of 64 B per finished path"
    elif kernel.startswith("k_paths_lockstep"):
        total = LUM_SECTORS * SECTOR * paths
        what = "1.25 sectors of 64 B per finished path (path state stays in registers)"
    elif kernel.startswith("traverse phase"):
        lists = float(sum(s.get("list_rays", 0) for s in stats))
        if stats and stats[0].get("shade_quads", 0):
            # the traverse launches walk the eight-wide trees only (the top level's quads are tested by the shade
            # launches, which read the ray anyway): per ray of a work list, list entry 4 B, ray 48 B and t of the closest
            # hit so far 8 B read (gathers), closest hit 16 B written
            total = (4.0 + 48.0 + 8.0 + 16.0) * lists
            what = (f"76 B per work-list entry of the eight-wide launches ({lists / rays:.2f} entries per ray; nothing else runs "
                    "in the traverse phase: the top level's quads are tested in the shade launches); gathers counted at "
                    "their useful bytes, not at sector granularity")
        else:
            # first stretch (lockstep, every ray): ray 48 B + stream position 12 B + signature 4 B read, closest hit 16 B
            # written; each eight-wide launch, per ray of its work list: list entry 4 B, ray 48 B and closest hit 16 B
            # read (gathers), closest hit 16 B written
            total = (64.0 + 16.0) * rays + (4.0 + 48.0 + 16.0 + 16.0) * lists
            what = ("80 B per ray in the stretch every ray goes through + 84 B per work-list entry of the eight-wide launches "
                    f"({lists / rays:.2f} entries per ray); gathers counted at their useful bytes, not at sector granularity")
    else:
        return None
    return {"bytes_per_ray": total / rays, "what": what, "rays": rays, "paths": paths}


def measure(args, config, spp_override, steps, warmup, torch, dist, rank, local_rank, world, with_cpu):
    """Times `steps` renders of one workload; returns the JSON fields of that workload (rank 0) or None."""
    from raytracer_amd import HostBuilder, camera_new, scenes
    from raytracer_amd.dist import render_sharded, shard_range
    if config in WORKLOADS:
        key, width, spp = WORKLOADS[config]
    elif config in scenes.SCENES:
        probe = scenes.build(config, HostBuilder())
        key, width, spp = config, probe.width, probe.spp
        probe.builder.close()
    else:
        raise SystemExit(f"unknown --config {config}: one of {sorted(WORKLOADS)} or {sorted(scenes.SCENES)}")
    width = args.width or width
    spp = spp_override or spp
    spec = scenes.build(key, HostBuilder(), width=width)
    cam = camera_new(image_width=width, **spec.camera)
    scene = spec.builder.commit(local_rank)
    film = torch.empty((cam.height, cam.width, 3), dtype=torch.float64, device=f"cuda:{local_rank}")
    schedule = True if args.whole_paths else (False if args.queues else None)
    strong = args.scaling == "strong"
    spp_job = spp if strong else spp * world

    def step():
        st = {}

        def render_range(begin, count):
            st.update(scene.render_into(cam, count, spec.max_depth, spec.background, film.data_ptr(), spp_begin=begin,
                                        batch_paths=args.batch_paths, whole_paths=schedule,
                                        split_bounce=args.split_bounce, fast_bvh=args.fast_bvh))

        render_sharded(render_range, film, spp_job)  # this rank's contiguous share of the job's samples, then all-reduce
        return st

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    stats = []
    for _ in range(steps):
        stats.append(step())
    fence()
    elapsed = time.perf_counter() - t0
    stats = [s for s in stats if s]  # (a rank whose share of a strong-scaling job is empty renders nothing)
    rays_local = float(sum(s["rays"] for s in stats))
    paths_local = float(sum(s["paths"] for s in stats))
    elapsed, total_rays, total_paths = reduce_over_ranks(elapsed, rays_local, paths_local, dist, f"cuda:{local_rank}")
    if rank != 0:
        scene.close()
        return None

    out = {
        "metric": "Mrays/s", "value": total_rays / elapsed / 1e6 if elapsed > 0 else 0.0, "unit": "Mrays/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": elapsed / max(steps, 1) * 1e3,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic", "mpaths_per_s": total_paths / elapsed / 1e6 if elapsed > 0 else 0.0,
        "config": {"workload": f"{config} {spec.name} {cam.width}x{cam.height}, "
                               f"{spp_job} spp in all ({shard_range(spp_job, 0, world)[1]} on rank 0), depth {spec.max_depth}",
                   "spp_total": spp_job, "paths_per_step": int(total_paths / max(steps, 1)), "seed": 42,
                   "parallelism": f"spp-shard x{world}" if world > 1 else "single GPU"},
    }
    # Per-ray work of this workload from the device counters (untimed passes at reduced spp).  "executed": the same
    # kernels and launches as the timed steps with their tallies switched on (the eight-wide walk included, its f32 child
    # box tests priced at half an f64 slab test).  "algorithmic": with the cull boxes off, where the box and primitive
    # tests are exactly the reference algorithm's (the entry-by-entry walk).
    cspp = max(1, min(spp, 4))
    ref = scene.render_into(cam, cspp, spec.max_depth, spec.background, film.data_ptr(), counters=True, cull=False)
    done = scene.render_into(cam, cspp, spec.max_depth, spec.background, film.data_ptr(), counters=True)
    n_node = ref["node_tests"] / ref["rays"]
    prim_bytes = ref["prim_bytes"] / ref["rays"]
    flops_alg = FLOPS_BOX * n_node + ref["prim_flops"] / ref["rays"] + FLOPS_SHADE
    flops_exec = (FLOPS_BOX * (done["node_tests"] + done["cull_tests"]) + 0.5 * FLOPS_BOX * done["wide_tests"] +
                  done["prim_flops"]) / done["rays"] + FLOPS_SHADE
    b_ray = S_NODE * n_node + prim_bytes + 2 * S_STATE + 2 * S_HIT
    # the dominant kernel (or, for trees with BVH entries, the launches of the traverse phase of one bounce) and its
    # mean duration, measured with HIP events on the library's stream during the timed steps
    whole = True   # the kernel does whole bounces (walk + shade), so a ray's whole flop count is its work
    if not stats:
        kernel, launches, ms_kernel = "none", 0, 0.0
    elif sum(s["path_launches"] for s in stats) > 0:
        kernel, launches, ms_kernel = "k_paths_lockstep", sum(s["path_launches"] for s in stats), sum(s["ms_paths"] for s in stats)
    elif sum(s["bounce_launches"] for s in stats) > 0:
        kernel, launches, ms_kernel = "k_bounce_lockstep", sum(s["bounce_launches"] for s in stats), sum(s["ms_bounce"] for s in stats)
    elif sum(s["traverse_launches"] for s in stats) > 0:
        whole = False
        kernel = "k_traverse_lockstep" if stats[0]["lockstep"] else "traverse phase of a bounce (k_traverse_lockstep / k_traverse_wide / k_traverse_persistent, one launch per stretch of the tree)"
        launches, ms_kernel = sum(s["traverse_launches"] for s in stats), sum(s["ms_traverse"] for s in stats)
    else:  # so small a workload that the tail kernel took all of it
        kernel, launches, ms_kernel = "k_finish", sum(s["finish_launches"] for s in stats), sum(s["ms_finish"] for s in stats)
    if launches > 0 and ms_kernel > 0:
        avg_ms = ms_kernel / launches
        rays_per_launch = rays_local / launches
        # (what the shade launches do is not the traverse phase's work: shading, and the top level's quads where those
        # are left to them -- 60 flops each, as the counting pass prices them in the traverse launches it runs)
        shade = 0.0 if whole else FLOPS_SHADE + 60.0 * stats[0].get("shade_quads", 0)
        alg = rays_per_launch * (flops_alg - shade) / (avg_ms * 1e-3) / 1e12
        exe = rays_per_launch * (flops_exec - shade) / (avg_ms * 1e-3) / 1e12
        prof = profiled(config)
        roof = {"bound": "fp64 valu", "kernel": kernel, "achieved": exe, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": exe / FP64_PEAK_TFLOPS, "frac_is": "executed", "frac_executed": exe / FP64_PEAK_TFLOPS,
                "frac_algorithmic": alg / FP64_PEAK_TFLOPS, "achieved_algorithmic": alg,
                "flops_per_ray_algorithmic": flops_alg, "flops_per_ray_executed": flops_exec,
                "avg_launch_ms": avg_ms, "launches": launches, "rays_per_launch": rays_per_launch,
                "kernel_share_of_step": ms_kernel / (elapsed * 1e3),
                "shade_ms_per_step": sum(s["ms_shade"] for s in stats) / max(steps, 1),
                "traverse_ms_per_step": sum(s["ms_traverse"] for s in stats) / max(steps, 1),
                "traffic": None,
                "note": "frac = executed flops (what the timed kernels do after culling: 18 flops per f64 box test, 9 per f32 "
                        "child box of an eight-wide tree; 35 / 60 / 70 / 30 per sphere / planar / quadric / medium test; 100 "
                        "per bounce shaded) x rays per launch / mean launch time / peak, on every workload; "
                        "frac_algorithmic prices the reference's own test counts instead and can exceed 1 where most of them "
                        "are culled; FMA contraction is off, so half of the peak is the ceiling for this arithmetic",
                "work_per_ray": {"reference_node_tests": n_node, "reference_prim_tests": ref["prim_tests"] / ref["rays"],
                                 "executed_node_tests": done["node_tests"] / done["rays"],
                                 "executed_cull_tests": done["cull_tests"] / done["rays"],
                                 "executed_wide_child_tests": done["wide_tests"] / done["rays"],
                                 "executed_prim_tests": done["prim_tests"] / done["rays"]},
                "cache_served_bytes_per_ray": b_ray}
        if prof.get("valu_issue"):
            roof["valu_issue_profiled"] = prof["valu_issue"]
        # HBM bytes of the same launches from what the library knows of its queues (bytes per ray of this run's own
        # ray, path and work-list counts), with the profiled figure beside it
        model = traffic_model(kernel, stats, rays_local, paths_local)
        hbm = None
        if model is not None:
            traffic = model["bytes_per_ray"] * rays_per_launch
            roof["traffic"] = traffic
            gbs = traffic / (avg_ms * 1e-3) / 1e9
            hbm = {"bound": "hbm", "kernel": kernel, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                   "traffic": traffic, "traffic_model": model, "traffic_profiled": None}
            bpr = prof.get("hbm_bytes_per_ray")
            if bpr and prof.get("kernel", "").split("<")[0].split(" ")[0] == kernel.split(" ")[0]:
                hbm["traffic_profiled"] = {"bytes_per_ray": bpr, "bytes_per_launch": bpr * rays_per_launch,
                                           "taken_at_commit": prof.get("commit"),
                                           "source": "rocprofv3 FETCH_SIZE (doubled, gfx950) + WRITE_SIZE of this command, per ray "
                                                     "of the profiled run (profiles/traffic.json)"}
        for r in (roof, hbm):
            if r is not None and not (0.0 <= r["frac"] <= 1.0):  # a fraction above 1 would be a modelling error: not reported
                r["frac_rejected"] = r["frac"]
                r["frac"] = None
        out["roofline"] = roof
        if hbm is not None:
            out["roofline_hbm"] = hbm
    else:
        out["roofline"] = None
    if stats and not stats[0]["lockstep"] and world == 1 and args.fast_report:
        out["fast_bvh"] = fast_bvh_report(scene, cam, spec, film, spp, out["value"])
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline(key, width)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"] if out["cpu_baseline"]["value"] else None
    scene.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="GPUs of this node to use, one rank each (default: the launcher's WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C2", help="a BASELINE.json configuration (C1..C5, the metric's workloads) or the "
                    "name of another scene program of the reference, run at the reference's own settings")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (invalidates the metric)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--batch-paths", type=int, default=0)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--fast-bvh", action="store_true", help="RT_FLAG_FAST_BVH for the timed steps (invalidates the metric: "
                    "ties between primitives are not decided as the reference decides them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fast-report", action="store_true", help="also render with RT_FLAG_FAST_BVH (untimed) and report its "
                    "speed-up and primitive-id mismatch rate under \"fast_bvh\" (trees with BVH entries)")
    ap.add_argument("--no-fast-report", action="store_true", help="(the default now; kept so that older command lines still run)")
    ap.add_argument("--no-also", action="store_true", help="skip the BVH workloads timed after the headline one")
    ap.add_argument("--also-c5-spp", type=int, default=0, help="samples per pixel of the C5 entry under \"also\" (0: the configuration's 10000)")
    ap.add_argument("--whole-paths", action="store_true", help="RT_FLAG_WHOLE_PATHS: one kernel per batch (BVH-free scenes)")
    ap.add_argument("--queues", action="store_true", help="RT_FLAG_QUEUES: the queue-based schedule whatever the scene")
    ap.add_argument("--split-bounce", action="store_true", help="RT_FLAG_SPLIT_BOUNCE: separate traverse and shade launches")
    args = ap.parse_args()
    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: this process starts the N ranks itself, before it has made any GPU call
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s): refusing to print a line "
                         f"for another GPU count than the one asked for")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, {torch.cuda.device_count()} visible: the HIP path is "
                         f"the only rendering path and every rank renders on a GPU of its own")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the only rendering path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    with_cpu = world == 1 and not args.no_cpu_baseline
    out = measure(args, args.config, args.spp, args.steps, args.warmup, torch, dist, rank, local_rank, world, with_cpu)
    if world == 1 and not args.no_also and args.config == "C2" and not args.spp and not args.width:
        also = []
        # (C5 at its full 10000 spp is one step of nine seconds, timed without a warm-up render)
        for cfg, spp, steps, warm in (("C4", 0, 2, 1), ("C5", args.also_c5_spp, 1 if not args.also_c5_spp else 2, 0 if not args.also_c5_spp else 1)):
            r = measure(args, cfg, spp, steps, warm, torch, dist, rank, local_rank, world, with_cpu)
            r["valid_metric_line"] = spp == 0  # (a reduced sample count gives a rate, not the configuration's line)
            also.append(r)
            key = cfg.lower()
            out[f"{key}_mrays_per_s"] = r["value"]
            out[f"{key}_mpaths_per_s"] = r["mpaths_per_s"]
            out[f"{key}_ms_per_step"] = r["ms_per_step"]
            out[f"{key}_spp"] = r["config"]["spp_total"]
            out[f"{key}_roofline_frac"] = (r.get("roofline") or {}).get("frac")
            out[f"{key}_shade_share_of_step"] = ((r.get("roofline") or {}).get("shade_ms_per_step") or 0.0) / r["ms_per_step"]
            out[f"{key}_gpu_over_cpu"] = r.get("gpu_over_cpu")
        out["also"] = also
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
