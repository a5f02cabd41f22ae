#!/usr/bin/env python3
"""Sums rocprofv3 counter_collection CSV rows per (kernel, counter) and kernel_trace durations per kernel.
usage: pmc_summary.py <dir> [more dirs...]  -> JSON on stdout"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    out = {"counters": defaultdict(lambda: defaultdict(float)), "dispatches": defaultdict(int),
           "duration_ns": defaultdict(float)}
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    out["counters"][k][row["Counter_Name"]] += float(row["Counter_Value"])
                    key = (k, row["Dispatch_Id"])
                    if key not in seen:
                        seen.add(key)
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    out["dispatches"][k] += 1
                    out["duration_ns"][k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    res = {"kernels": {}}
    for k in set(out["counters"]) | set(out["dispatches"]):
        res["kernels"][k] = {"dispatches": out["dispatches"].get(k, 0), "total_ms": out["duration_ns"].get(k, 0) / 1e6,
                             "avg_ms": (out["duration_ns"].get(k, 0) / 1e6 / out["dispatches"][k]) if out["dispatches"].get(k) else None,
                             "counters": dict(out["counters"].get(k, {}))}
    json.dump(res, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
