#!/usr/bin/env python3
"""Per-kernel digest of a summary.json written by tools/pmc_summary.py (development aid).
usage: tools/pmc_report.py <summary.json>"""
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["total_ms"]):
    c = v["counters"]
    if "SQ_WAVES" not in c or c["SQ_WAVES"] == 0 or v["total_ms"] < 0.5:
        continue
    w = c["SQ_WAVES"]
    g = c.get("GRBM_GUI_ACTIVE", 0) / 8.0  # summed over 8 XCDs
    print(k[:34], "n", v["dispatches"], "ms %.1f" % v["total_ms"], "waves %.3g" % w,
          "VALU/wave %.0f" % (c["SQ_INSTS_VALU"] / w), "VMEM/wave %.0f" % (c["SQ_INSTS_VMEM_RD"] / w),
          "SMEM/wave %.0f" % (c["SQ_INSTS_SMEM"] / w), "LDS/wave %.0f" % (c["SQ_INSTS_LDS"] / w),
          "| of wave cycles: busy %.2f wait_any %.2f wait_inst %.2f" % (c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]),
          "| valu_active/busy_cu_cycles %.2f" % (c["SQ_ACTIVE_INST_VALU"] / max(c["SQ_BUSY_CYCLES"], 1)),
          "| lanes/valu %.1f" % (c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"] * 4 if c.get("SQ_ACTIVE_INST_VALU") else 0),
          "| L2 hit %.3f" % (c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)),
          "| TCP lines/VMEM %.1f" % (c["TCP_TOTAL_CACHE_ACCESSES_sum"] / max(c["SQ_INSTS_VMEM_RD"], 1)),
          "| TA busy frac %.2f" % (c["TA_TA_BUSY_sum"] / 256.0 / max(g, 1)),
          "| FETCH MB %.1f WRITE MB %.1f" % (c.get("FETCH_SIZE", 0) / 1024.0, c.get("WRITE_SIZE", 0) / 1024.0))
