#!/bin/bash
# development aid (needs a build with `make DEV=1`): wide-kernel thresholds against throughput
# usage: tools/sweep_wide.sh "<TS values>" "<TF values>" [config:spp ...]
tss=$1; tfs=$2; shift 2
cfgs=${@:-C4:200 C5:64}
for ts in $tss; do for tf in $tfs; do
  for c in $cfgs; do
    RT_WIDE_TS=$ts RT_WIDE_TF=$tf timeout -k 10 300 python -u bench.py --config ${c%%:*} --spp ${c##*:} --steps 1 --warmup 1 --no-cpu-baseline --no-also --no-fast-report 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']
    print('TS=$ts TF=$tf', d['config']['workload'][:16], 'Mrays/s %.0f'%d['value'], 'trav_ms %.1f'%r['traverse_ms_per_step'], 'shade_ms %.1f'%r['shade_ms_per_step'])
" || exit 1
  done
done; done
