#!/bin/bash
# Kernel-by-kernel trace of the probe workloads' steps (what exactly a step launches, with durations and gaps):
#   bash tools/step_trace.sh gpurun_out/steps "autorec hybrid cfg1"
set -e -o pipefail
OUT=$(realpath -m "$1"); WORK=${2:-autorec hybrid cfg1 cfg2}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for w in $WORK; do
  timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/$w" -- python3 "$ROOT/tools/ab_probe.py" $w > "$OUT/$w.json" 2> "$OUT/$w.err"
  echo "$w traced"
done
cd "$ROOT"
python3 - "$OUT" $WORK <<'PY'
import csv, glob, sys
out = sys.argv[1]
for w in sys.argv[2:]:
    fs = glob.glob("%s/%s/*/*kernel_trace.csv" % (out, w))
    if not fs:
        continue
    rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
    print("== %s: last 24 launches" % w)
    prev = None
    for r in rows[-24:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("  %-70s %9.1f us  gap %7.1f us  grid %s" % (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70], (e - s) / 1e3,
                                                        ((s - prev) / 1e3) if prev else 0.0, r["Grid_Size_X"]))
        prev = e
PY
