#!/bin/bash
# usage: tools/pmc_extra.sh <config> <spp> <tag>   (on the GPU box, from the repository root)
# Extra --pmc passes for hunting a bottleneck: instruction fetch, scalar unit, VALU mix, texture path stalls, L2.
cfg=$1; spp=$2; tag=$3
root=$(pwd)
out=$root/gpurun_out/pmcx_$tag
mkdir -p $out
export TMPDIR=/tmp
cmd="python3 $root/bench.py --config $cfg --spp $spp --steps 1 --warmup 0 --no-cpu-baseline --no-also --no-fast-report"
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- $cmd > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_BUSY_sum TCC_REQ_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $pmc --output-format csv -d $out/pmc$i -- $cmd > $out/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $out/pmc$i.log; }
done
cd $root
python3 tools/pmc_summary.py $out/trace $out/pmc* > $out/summary.json
rm -rf $out/trace $out/pmc?
python3 - <<PY
import json
d=json.load(open("$out/summary.json"))
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["total_ms"])[:3]:
    print(k[:40], "n",v["dispatches"],"total_ms %.2f"%v["total_ms"])
    for c,x in sorted(v["counters"].items()): print("    %-44s %.4g"%(c,x))
PY
