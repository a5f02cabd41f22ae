#!/bin/bash
# Register / LDS / occupancy report of every kernel instantiation (device-only compile, nothing is written in-tree).
# usage: tools/kernel_resources.sh [name-filter]
root=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I$root/include -I$root/raytracer_amd/csrc \
  --cuda-device-only -c $root/raytracer_amd/csrc/rt_amd.hip -o /tmp/rt_dev_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c "
import re, sys, subprocess
flt = sys.argv[1] if len(sys.argv) > 1 else ''
cur = None
rows = {}
for line in sys.stdin:
    if 'error' in line: print(line.rstrip())
    m = re.search(r'remark: +(.*?) \[-Rpass', line)
    if not m: continue
    t = m.group(1)
    if t.startswith('Function Name:'):
        cur = t.split(':', 1)[1].strip(); rows[cur] = {}
    elif cur and ':' in t:
        k, v = t.split(':', 1); rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
    if flt in dem:
        print('%-48s VGPR %3s spill %3s | SGPR spill %3s | scratch %5s B | LDS %5s B | waves/SIMD %s' % (
            dem.replace('void ', ''), r.get('VGPRs'), r.get('VGPRs Spill'), r.get('SGPRs Spill'), r.get('ScratchSize [bytes/lane]'),
            r.get('LDS Size [bytes/block]'), r.get('Occupancy [waves/SIMD]')))
" "$1"
rm -f /tmp/rt_dev_$$.o
