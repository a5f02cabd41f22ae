#!/bin/bash
# usage: tools/ktrace.sh <tag> <config> <spp>   (GPU box, repository root): per-kernel times of one bench step
tag=$1; cfg=$2; spp=$3
root=$(pwd); out=$root/gpurun_out/kt_$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --config $cfg --spp $spp --steps 1 --warmup 0 --no-cpu-baseline --no-also > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
cd $root
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $out/stats.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(r['Name'][:70], 'calls',r['Calls'],'total_ms %.2f'%(float(r['TotalDurationNs'])/1e6),'avg_ms %.3f'%(float(r['AverageNs'])/1e6), 'pct',r['Percentage'])
PY
grep metric $out/trace.log | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('Mrays/s %.0f ms/step %.1f'%(d['value'],d['ms_per_step']))"
rm -rf $out/trace
