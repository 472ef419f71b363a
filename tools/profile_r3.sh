#!/bin/bash
# Round-3 rocprofv3 evidence for the driver's bench command (run on the GPU box from the repo root):
#   bash tools/profile_r3.sh gpurun_out/r3p
# 1. the bench line itself (no profiler), 2. kernel trace + stats of the same command, 3. separate --pmc passes (FETCH_SIZE,
# WRITE_SIZE, MFMA busy, instruction counts) over the cfg2 AND cfg5 regions.  Every GPU step has its own time limit and the
# script stops at the first failure.  Summaries go to $OUT; copy what is to be judged into profiles/.
set -e -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/r3p}")
ROOT=$(pwd)
mkdir -p "$OUT"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 > "$OUT/bench_prof.json" 2> "$OUT/bench_prof.err"
echo "bench trace done"
PMC_ARGS="--steps 5 --warmup 1 --precondition 8 --no-cpu-baseline --no-split --no-kernels --cfg5-steps 2"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/bench_$c" -- python3 "$ROOT/bench.py" $PMC_ARGS > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err"
  echo "bench $c done"
done
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$OUT/bench_mfma" -- python3 "$ROOT/bench.py" $PMC_ARGS > "$OUT/bench_mfma.json" 2> "$OUT/bench_mfma.err"
echo "bench mfma done"
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$OUT/bench_insts" -- python3 "$ROOT/bench.py" $PMC_ARGS > "$OUT/bench_insts.json" 2> "$OUT/bench_insts.err"
echo "bench instruction counts done"
# the per-kernel sections alone, traced (kernel names and durations behind gather / cfg1 / cfg3 / cfg4 of the bench line)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kernels_trace" -- python3 "$ROOT/tools/bench_kernels.py" > "$OUT/bench_kernels.json" 2> "$OUT/bench_kernels.err"
echo "kernel sections trace done"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/kernels_$c" -- python3 "$ROOT/tools/bench_kernels.py" > "$OUT/kernels_$c.json" 2> "$OUT/kernels_$c.err"
  echo "kernel sections $c done"
done
cd "$ROOT"
python3 tools/kernel_trace_summary.py "$OUT/bench_trace" > "$OUT/bench_trace.json"
python3 tools/kernel_trace_summary.py "$OUT/kernels_trace" > "$OUT/kernels_trace.json"
python3 tools/pmc_summary.py "$OUT/bench_FETCH_SIZE" "$OUT/bench_WRITE_SIZE" "$OUT/bench_mfma" > "$OUT/bench_hbm_traffic.json"
python3 tools/pmc_summary.py "$OUT/kernels_FETCH_SIZE" "$OUT/kernels_WRITE_SIZE" > "$OUT/kernels_hbm_traffic.json"
python3 tools/pmc_summary.py "$OUT/bench_insts" > "$OUT/bench_sq_insts.json"
find "$OUT" -name "*stats*.csv" | head -20
# raw traces are large: keep the summaries and the stats tables only
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
echo "profile_r3 done"
