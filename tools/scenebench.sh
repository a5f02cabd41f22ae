#!/bin/bash
# The reference's other twelve scene programs at the settings main.rs gives them, one JSON line each (run on the GPU
# box from the repository root).  usage: tools/scenebench.sh <tag>
tag=$1
out=gpurun_out/scenes_$tag.jsonl
rm -f $out
run() { timeout -k 10 300 python3 -u bench.py --config $1 --steps 2 --warmup 1 2>gpurun_out/scenes_$tag.$1.err | grep '"metric"' >> $out || exit 1; echo "$1 done"; }
run checkered_spheres && run earth && run perlin_spheres && run para && run simple_light && run cornell_box && \
run debug_quadric && run cornell_triangle && run profiler_scene && run spherical_mirror && run debug_model && run bust && python3 - <<PY
import json
for l in open("$out"):
    d = json.loads(l); r = d["roofline"]; c = d.get("cpu_baseline", {})
    print(d["config"]["workload"][:60], "| Mrays/s %.0f | Mpaths/s %.1f | ms/step %.1f | kernel %s avg %.3f ms | cpu %.1f Mrays/s (%d thr) | x%.0f" % (
        d["value"], d["mpaths_per_s"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], c.get("value", 0), c.get("cores", 0), d.get("gpu_over_cpu", 0)))
PY
