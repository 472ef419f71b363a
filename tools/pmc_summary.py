#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes -> one JSON (the file bench.py reads `roofline.traffic` from).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d OUT/fetch -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d OUT/write -- python3 bench.py ...          (separate pass: TCC has 4 slots)
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d OUT/mfma -- ...
    python tools/pmc_summary.py OUT/fetch OUT/write OUT/mfma > profiles/rNN_bench_n1_hbm_traffic.json

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))     # kernel -> counter -> [sum, launches]
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(float)                       # a counter may be reported once per XCD / instance
            names = {}
            for row in csv.DictReader(open(f)):
                key = (row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = row["Kernel_Name"]
            for (disp, ctr), v in per_dispatch.items():
                a = acc[names[disp]][ctr]
                a[0] += v
                a[1] += 1
    out = {"passes": sys.argv[1:], "kernels": {}}
    for k, ctrs in acc.items():
        e = {}
        for c, (s, n) in ctrs.items():
            e[c + "_avg_per_launch"] = s / n
            e["launches"] = n
        if "FETCH_SIZE_avg_per_launch" in e and "WRITE_SIZE_avg_per_launch" in e:
            e["hbm_bytes_per_launch_corrected"] = (2.0 * e["FETCH_SIZE_avg_per_launch"] + e["WRITE_SIZE_avg_per_launch"]) * 1024.0
        out["kernels"][k] = e
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
