#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per counter) into per-kernel, per-grid-size averages.

Usage (on the GPU box, each counter in its own pass as MI355X_MICROARCH.md prescribes):
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$c -o p --output-format csv -- \
            python bench.py --inflight 1 --steps 4 --warmup 1 --cpu-instances 0 --no-profile
    done
    python scripts/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > profiles/<round>/pmc_summary.json

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB; on gfx950 FETCH_SIZE under-counts by 2x (64 B per 128 B
request), so `hbm_read_bytes` below applies the x2 correction from the guide; WRITE_SIZE is left as reported.
"""
import collections
import csv
import glob
import json
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main(dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                key = (short(row["Kernel_Name"]), int(row["Grid_Size"]) // max(1, int(row["Workgroup_Size"])))
                acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for (kern, blocks), ctrs in sorted(acc.items()):
        e = {"launches": max(len(v) for v in ctrs.values())}
        for c, v in ctrs.items():
            e[c + "_avg"] = sum(v) / len(v)
        if "FETCH_SIZE" in ctrs:
            e["hbm_read_bytes"] = 2.0 * 1024.0 * e["FETCH_SIZE_avg"]
        if "WRITE_SIZE" in ctrs:
            e["hbm_write_bytes"] = 1024.0 * e["WRITE_SIZE_avg"]
        out["%s @ %d blocks" % (kern, blocks)] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
