#!/usr/bin/env python3
"""Writes profiles/traffic.json (read by bench.py for roofline.traffic) from the PMC summaries under profiles/.

HBM bytes per launch of a kernel = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / dispatches: rocprofv3 reports both counters
in KiB, and on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes, so the read side is doubled.  The two
counters come from separate --pmc passes of the same command (tools/profile.sh).

usage: tools/make_traffic.py <config>=<summary.json> [...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = {}
    for arg in sys.argv[1:]:
        cfg, path = arg.split("=", 1)
        with open(path) as fh:
            kernels = json.load(fh)["kernels"]
        entry = {}
        for name, k in kernels.items():
            c = k["counters"]
            if not k["dispatches"] or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                continue
            base = name.split("<")[0]
            # the traverse / whole-path instantiations whose first template argument (COUNT) is true only run in the
            # untimed counting passes
            if base != "k_shade" and "<true" in name:
                continue
            entry[base + "_hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / k["dispatches"]
            entry[base + "_avg_ms"] = k["avg_ms"]
        out[cfg] = entry
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
