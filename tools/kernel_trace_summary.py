#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> per (kernel, grid, workgroup) launch count and average / min duration.

    python tools/kernel_trace_summary.py OUT_DIR > profiles/rNN_kernel_bench_trace.json

rocprofv3's own --stats table averages over a kernel NAME; tools/kernel_bench.py launches the same kernel at several shapes
(four gather tables, four top-k shapes) and bench.py runs the persistent tile kernel on two workloads (cfg2: 21 M pairs per
launch, cfg5: 256-268 M pairs per launch, same 256-workgroup grid), so the durations are grouped by launch geometry and,
within one geometry, by duration band (a new band starts where the sorted durations jump by more than 2x).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    acc = defaultdict(list)
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6      # ms
                grid = "%sx%sx%s" % (row.get("Grid_Size_X", row.get("Grid_Size", "?")), row.get("Grid_Size_Y", "1"), row.get("Grid_Size_Z", "1"))
                wg = "%sx%sx%s" % (row.get("Workgroup_Size_X", row.get("Workgroup_Size", "?")), row.get("Workgroup_Size_Y", "1"),
                                   row.get("Workgroup_Size_Z", "1"))
                acc[(row["Kernel_Name"], grid, wg)].append(dur)
    out = []
    for (name, grid, wg), allv in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        allv.sort()
        bands, cur = [], [allv[0]]
        for x in allv[1:]:
            if x > 2.0 * cur[-1]:
                bands.append(cur)
                cur = []
            cur.append(x)
        bands.append(cur)
        for b, v in enumerate(bands):
            out.append({"kernel": name, "grid_threads": grid, "workgroup": wg, "duration_band": "%d of %d" % (b + 1, len(bands)),
                        "launches": len(v), "avg_ms": sum(v) / len(v), "median_ms": v[len(v) // 2], "min_ms": v[0], "max_ms": v[-1],
                        "total_ms": sum(v)})
    json.dump({"source": sys.argv[1:], "kernels": out}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
