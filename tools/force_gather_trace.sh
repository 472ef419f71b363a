#!/bin/bash
# kernel trace of bench.py's cfg2 steps in a one-rank RCCL group with the all-gather forced: prints the launches of the last steps
# (start, duration, gap to the previous kernel's end, name).   bash tools/force_gather_trace.sh [out]
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/${1:-gpurun_out/fgt}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
RECO_FORCE_GATHER=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29743 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-split --no-cfg5 --no-kernels > "$OUT/bench.json" 2> "$OUT/bench.err"
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
f = sorted(glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-28:]
prev_end = None
for r in tail:
    b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (b - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%10.1f us  dur %9.1f  gap %8.1f  q%s  %s" % ((b - int(tail[0]["Start_Timestamp"])) / 1e3, (e - b) / 1e3, gap, r.get("Queue_Id", "?"), r["Kernel_Name"][:90]))
    prev_end = e if prev_end is None else max(prev_end, e)
PY
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
