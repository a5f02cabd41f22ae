#!/bin/bash
# quick overview: reduced-spp runs of several configs (not valid metric lines; for development only)
# usage: tools/quickbench.sh <tag> [config:spp ...]
tag=$1; shift
cfgs=${@:-C2:200 C3:200 C4:200 C5:64 C1:10}
out=gpurun_out/quick_$tag.log
rm -f $out
for c in $cfgs; do
  timeout -k 10 400 python -u bench.py --config ${c%%:*} --spp ${c##*:} --steps 1 --warmup 1 --no-cpu-baseline --no-also 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']
    print(d['config']['workload'][:22], 'Mrays/s %.0f'%d['value'], 'Mpaths/s %.0f'%d['mpaths_per_s'], 'ms/step %.1f'%d['ms_per_step'], 'kernel_share %.2f'%r['kernel_share_of_step'], 'shade_ms %.1f'%r['shade_ms_per_step'], 'avg_launch_ms %.3f'%r['avg_launch_ms'], 'fp64 frac %.3f / %.3f'%(r['frac_algorithmic'],r['frac_executed']), ('fast x%.2f mismatch %.2e'%(d['fast_bvh']['speedup'], d['fast_bvh']['prim_id_mismatch_rate'])) if 'fast_bvh' in d else '')
" >> $out || exit 1
done
cat $out
