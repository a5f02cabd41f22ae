#!/usr/bin/env python3
"""Writes profiles/traffic.json (read by bench.py for roofline.traffic / roofline_hbm and the VALU issue account) from
the PMC summaries tools/profile.sh leaves under gpurun_out/prof_<tag>/ (copied to profiles/).

Per workload, for the dominant kernel bench.py names (for trees with BVH entries: all launches of a bounce's traverse
phase together):
  hbm_bytes_per_ray = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / rays of the profiled run (warm-up steps included).  rocprofv3 reports
      both counters in KiB, and on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes, so the read side is
      doubled.  The two counters come from separate --pmc passes of the same command.
  valu_issue: SQ_INSTS_VALU x 4 cycles / (256 CUs x 4 SIMDs x 2.4 GHz x the kernel's summed duration): the share of
      the SIMDs' issue slots the kernel's vector instructions take ("VALU busy"), with the numbers it is made of;
      lanes_active = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU).

usage: tools/make_traffic.py <config>=<summary.json>,<bench.json> [...]   (bench.json: the line bench.py printed in
the kernel-trace pass of that profile)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRAVERSE = ("k_traverse_lockstep", "k_traverse_wide", "k_traverse_persistent")


def main():
    path_out = os.path.join(ROOT, "profiles", "traffic.json")
    out = {}
    try:
        commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        commit = None
    if os.path.exists(path_out):
        with open(path_out) as fh:
            out = json.load(fh)
    for arg in sys.argv[1:]:
        cfg, paths = arg.split("=", 1)
        summary, bench = paths.split(",")
        with open(summary) as fh:
            kernels = json.load(fh)["kernels"]
        with open(bench) as fh:
            line = json.loads([l for l in fh if '"metric"' in l][-1])
        roof = line["roofline"]
        wanted = TRAVERSE if roof["kernel"].startswith("traverse phase") else (roof["kernel"].split(" ")[0],)
        agg = {}
        ms = 0.0
        names = []
        for name, k in kernels.items():
            base = name.split("<")[0]
            # instantiations whose first template argument (COUNT) is true only run in the untimed counting passes
            # (k_traverse_wide's COUNT is its second argument, the first being the cycle profile)
            counting = name.replace(" ", "").endswith(",true>") if base == "k_traverse_wide" else "<true" in name
            if base not in wanted or counting or not k["dispatches"]:
                continue
            names.append(name)
            ms += k["total_ms"]
            for c, v in k["counters"].items():
                agg[c] = agg.get(c, 0.0) + v
        if not names or "FETCH_SIZE" not in agg:
            print(f"{cfg}: no counters for {wanted}", file=sys.stderr)
            continue
        # the counters cover the warm-up steps of the profiled command as well as the timed ones
        rays = roof["rays_per_launch"] * roof["launches"] * (line["steps"] + line["warmup"]) / max(line["steps"], 1)
        entry = {"kernel": roof["kernel"], "profiled_kernels": sorted(names), "profiled_rays": rays, "profiled_kernel_ms": ms,
                 "hbm_bytes_per_ray": (2.0 * agg["FETCH_SIZE"] + agg["WRITE_SIZE"]) * 1024.0 / rays,
                 "source": os.path.relpath(summary, ROOT), "commit": commit}
        if "SQ_INSTS_VALU" in agg:
            entry["valu_issue"] = {
                "valu_insts": agg["SQ_INSTS_VALU"], "waves": agg.get("SQ_WAVES"), "cycles_per_inst": 4, "simds": 1024,
                "clock_ghz": 2.4, "kernel_ms": ms, "frac": agg["SQ_INSTS_VALU"] * 4.0 / (1024 * 2.4e9 * ms * 1e-3),
                "lanes_active": agg["SQ_THREAD_CYCLES_VALU"] / (64.0 * agg["SQ_ACTIVE_INST_VALU"]) if agg.get("SQ_ACTIVE_INST_VALU") else None,
                "valu_insts_per_ray": agg["SQ_INSTS_VALU"] / rays}
        out[cfg] = entry
    with open(path_out, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
