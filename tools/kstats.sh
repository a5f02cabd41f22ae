#!/bin/bash
# development aid: per-kernel totals of one reduced bench run (rocprofv3 --kernel-trace --stats); run on the GPU box
# usage: tools/kstats.sh <config> <spp> <tag>
root=$(pwd); out=$root/gpurun_out/ks_$3; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --config $1 --spp $2 --steps 1 --warmup 0 --no-cpu-baseline --no-also --no-fast-report > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
cd $root
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-60s calls %5s total_ms %9.2f avg_us %9.1f pct %5s" % (r["Name"].split("(")[0].replace("void ", "")[:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
rm -rf $out/trace
