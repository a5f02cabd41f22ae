#!/bin/bash
# usage: [STEPS=k WARMUP=w] tools/profile.sh <config> <spp> <tag>   (run on the GPU box from the repository root)
# kernel-trace + stats pass and separate --pmc passes of the same bench command; summaries land in gpurun_out/prof_<tag>/
cfg=$1; spp=$2; tag=$3
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cmd="python3 $root/bench.py --config $cfg --spp $spp --steps ${STEPS:-1} --warmup ${WARMUP:-0} --no-cpu-baseline --no-also --no-fast-report"
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $cmd > $out/trace.log 2>&1 || { tail -5 $out/trace.log; exit 1; }
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $pmc --output-format csv -d $out/pmc$i -- $cmd > $out/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $out/pmc$i.log; }
done
cd $root
python3 tools/pmc_summary.py $out/trace $out/pmc* > $out/summary.json
grep '"metric"' $out/trace.log | tail -1 > $out/bench.json
find $out -name "*_kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/trace $out/pmc?  # keep only the summaries (raw CSVs are large)
python3 - <<PY
import json
d=json.load(open("$out/summary.json"))
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["total_ms"])[:4]:
    print(k[:40], "n",v["dispatches"],"total_ms %.2f"%v["total_ms"], {c: "%.3g"%x for c,x in v["counters"].items()})
PY
