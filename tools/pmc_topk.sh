#!/bin/bash
# SQ counters of the stand-alone top-k kernels for two library variants (run on the GPU box from the repo root):
#   bash tools/pmc_topk.sh gpurun_out/pmc_topk base sel
set -e -o pipefail
OUT=$(realpath -m "$1"); shift
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export RECO_HIP_LIB=$ROOT/recommender_amd/lib/ab/$v.so
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$OUT/$v" -- python3 "$ROOT/tools/ab_probe.py" topk > "$OUT/$v.json" 2> "$OUT/$v.err"
  echo "$v counters done"
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/${v}_b" -- python3 "$ROOT/tools/ab_probe.py" topk > "$OUT/${v}_b.json" 2> "$OUT/${v}_b.err"
  echo "$v counters b done"
done
cd "$ROOT"
for v in "$@"; do
  python3 tools/pmc_summary.py "$OUT/$v" "$OUT/${v}_b" > "$OUT/$v.summary.json"
done
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
echo done
