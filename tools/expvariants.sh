#!/bin/bash
# development aid: runs a reduced bench once per experimental build of the library found in build/exp/ (on the GPU box)
# usage: tools/expvariants.sh "<config:spp ...>" [variant names...]
cfgs=$1; shift
cp raytracer_amd/csrc/librt_amd.so /tmp/librt_amd_base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/librt_amd_base.so raytracer_amd/csrc/librt_amd.so; else cp build/exp/librt_amd_$v.so raytracer_amd/csrc/librt_amd.so; fi
  for c in $cfgs; do
    timeout -k 10 300 python -u bench.py --config ${c%%:*} --spp ${c##*:} --steps 1 --warmup 1 --no-cpu-baseline --no-also --no-fast-report 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']
    print('$v', d['config']['workload'][:16], 'Mrays/s %.0f'%d['value'], 'ms %.1f'%d['ms_per_step'], 'trav_ms %.1f'%r['traverse_ms_per_step'], 'shade_ms %.1f'%r['shade_ms_per_step'])
" || exit 1
  done
done
cp /tmp/librt_amd_base.so raytracer_amd/csrc/librt_amd.so
