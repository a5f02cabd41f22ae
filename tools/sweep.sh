#!/bin/bash
# usage: tools/sweep.sh <config> <spp>  (GPU box): thresholds of the wide kernel (development aid)
cfg=$1; spp=$2
for tf in 8 16 24 32; do for ts in 16 24 32; do
  echo -n "tf=$tf ts=$ts: "
  RT_WIDE_TF=$tf RT_WIDE_TS=$ts timeout -k 10 200 python -u bench.py --config $cfg --spp $spp --steps 1 --warmup 1 --no-cpu-baseline --no-also 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('Mrays/s %.0f'%d['value'])" || exit 1
done; done
