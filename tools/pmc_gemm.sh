#!/bin/bash
# SQ counters of one GEMM shape (one --pmc pass, 8 SQ counters):   bash tools/pmc_gemm.sh OUT M N K
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/${1:?out dir}; M=$2; N=$3; K=$4
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVES -d "$OUT/pmc" -- python3 "$ROOT/tools/gemm_one.py" $M $N $K 8 > "$OUT/run.log" 2>&1
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT/pmc" > "$OUT/summary.json"
python3 - "$OUT/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["kernels"].items():
    if "linear_kernel" in k or "splitk" in k:
        busy = v.get("SQ_BUSY_CYCLES_avg_per_launch", 0)
        print(k[:100]); print("  ", {kk.replace("_avg_per_launch", ""): round(vv) for kk, vv in v.items()})
        if busy:
            print("   MFMA busy share: %.3f (of 32 x SQ_BUSY_CYCLES)" % (v.get("SQ_VALU_MFMA_BUSY_CYCLES_avg_per_launch", 0) / (32 * busy)))
PY
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
