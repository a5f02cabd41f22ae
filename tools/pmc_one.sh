#!/bin/bash
# usage: tools/pmc_one.sh <config> <spp> <tag> <counter> [counter ...]   one --pmc pass of a short bench run, summed per kernel
cfg=$1; spp=$2; tag=$3; shift 3
root=$(pwd); out=$root/gpurun_out/pmc_$tag; mkdir -p $out; export TMPDIR=/tmp
cd /tmp
timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $out/p -- python3 $root/bench.py --config $cfg --spp $spp --steps 1 --warmup 0 --no-cpu-baseline > $out/log 2>&1 || { tail -3 $out/log; exit 1; }
cd $root
python3 tools/pmc_summary.py $out/p > $out/summary.json
rm -rf $out/p
python3 - <<PY
import json
d=json.load(open("$out/summary.json"))
for k,v in sorted(d["kernels"].items(), key=lambda kv:-sum(kv[1]["counters"].values()))[:4]:
    print(k[:40], {c: "%.3g"%x for c,x in v["counters"].items()})
PY
