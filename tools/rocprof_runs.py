#!/usr/bin/env python3
"""Per-RUN rocprofv3 figures for kernels that tools/kernel_bench.py launches at several shapes.

    python tools/rocprof_runs.py KERNEL_SUBSTRING TRACE_DIR [FETCH_DIR WRITE_DIR] > profiles/rNN_<kernel>_rocprof.json

A run = consecutive dispatches of the same kernel (kernel_bench launches every shape 23 times back to back: 3 warm-up + 20
timed).  Durations come from the --kernel-trace pass, HBM bytes from the separate --pmc FETCH_SIZE / WRITE_SIZE passes of the
same program (same dispatch order), corrected as MI355X_MICROARCH.md prescribes: both counters are in KiB, FETCH_SIZE reports
half of the bytes of wide coalesced reads on gfx950 (doubled here), WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import csv
import glob
import json
import os
import sys


def rows_of(d, suffix):
    out = []
    for f in glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def runs(rows, sub, value):
    """[(first dispatch ordinal among matches, [values])] for consecutive dispatches whose kernel name contains `sub`."""
    rows = sorted(rows, key=lambda r: int(r["Dispatch_Id"]) if "Dispatch_Id" in r else int(r["Start_Timestamp"]))
    out, cur, prev_match = [], None, False
    for r in rows:
        m = sub in r["Kernel_Name"]
        if m:
            if not prev_match:
                cur = []
                out.append(cur)
            cur.append(value(r))
        prev_match = m
    return out


def main():
    sub, trace = sys.argv[1], sys.argv[2]
    dur = runs(rows_of(trace, "kernel_trace.csv"), sub, lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    res = {"kernel": sub, "passes": sys.argv[2:], "runs": []}
    fetch = write = None
    if len(sys.argv) >= 5:
        fetch = runs(rows_of(sys.argv[3], "counter_collection.csv"), sub, lambda r: float(r["Counter_Value"]) if r["Counter_Name"] == "FETCH_SIZE" else 0.0)
        write = runs(rows_of(sys.argv[4], "counter_collection.csv"), sub, lambda r: float(r["Counter_Value"]) if r["Counter_Name"] == "WRITE_SIZE" else 0.0)
    for i, d in enumerate(dur):
        timed = sorted(d[3:]) if len(d) > 3 else sorted(d)
        e = {"run": i, "launches": len(d), "avg_ms": sum(d) / len(d), "median_ms_timed": timed[len(timed) // 2], "min_ms": min(d)}
        if fetch and i < len(fetch) and write and i < len(write):
            # a counter is reported once per XCD/instance and dispatch: sum over the run, divide by the launches
            f = sum(fetch[i]) / len(d) * 1024.0
            w = sum(write[i]) / len(d) * 1024.0
            e.update({"FETCH_SIZE_bytes_per_launch_raw": f, "WRITE_SIZE_bytes_per_launch": w,
                      "hbm_bytes_per_launch_corrected": 2.0 * f + w,
                      "hbm_GBps_from_counters": (2.0 * f + w) / (e["median_ms_timed"] * 1e-3) / 1e9})
        res["runs"].append(e)
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
