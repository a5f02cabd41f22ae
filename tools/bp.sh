#!/bin/bash
# usage: tools/bp.sh <config> <spp> <batch_paths>...   (GPU box): quick look at how the number of paths in flight changes the rate
cfg=$1; spp=$2; shift 2
for bp in "$@"; do
  timeout -k 10 300 python -u bench.py --config $cfg --spp $spp --steps 1 --warmup 1 --no-cpu-baseline --no-also --batch-paths $bp 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']
    print('$cfg bp=$bp', 'Mrays/s %.0f'%d['value'], 'ms/step %.1f'%d['ms_per_step'], 'avg_launch_ms %.3f'%r['avg_launch_ms'], 'launches', r['launches'])
" || exit 1
done
