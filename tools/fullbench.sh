#!/bin/bash
# Full-size runs of all five BASELINE configurations, one JSON line each (run on the GPU box from the repository root).
# usage: tools/fullbench.sh <tag>
tag=$1
out=gpurun_out/full_$tag.jsonl
rm -f $out
run() { timeout -k 10 900 python3 -u bench.py --config $1 --steps $2 --warmup $3 --no-also 2>gpurun_out/full_$tag.$1.err | grep '"metric"' >> $out || exit 1; echo "$1 done"; }
run C1 5 1 && run C2 3 1 && run C3 3 1 && run C4 2 1 && run C5 1 0 && python3 - <<PY
import json
for l in open("$out"):
    d = json.loads(l); r = d["roofline"]; c = d.get("cpu_baseline", {})
    print(d["config"]["workload"][:34], "| Mrays/s %.0f | Mpaths/s %.1f | ms/step %.1f | kernel %s avg %.3f ms | frac %.2f | cpu %.1f Mrays/s (%d thr) | x%.0f" % (
        d["value"], d["mpaths_per_s"], d["ms_per_step"], r["kernel"][:24], r["avg_launch_ms"], r["frac"], c.get("value", 0), c.get("cores", 0), d.get("gpu_over_cpu", 0)))
PY
