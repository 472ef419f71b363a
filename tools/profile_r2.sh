#!/bin/bash
# rocprofv3 evidence for every kernel that has a roofline target.  Run on the GPU box from the repo root:
#   bash tools/profile_r2.sh gpurun_out/r2b
# Each GPU step has its own time limit; the script stops at the first failure.
set -e -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/r2b}")
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats over the per-kernel bench (program directly after --)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kb_trace" -- python3 "$ROOT/tools/kernel_bench.py" > "$OUT/kernel_bench.jsonl" 2> "$OUT/kernel_bench.err"
echo "kernel_bench trace done"
# 2. HBM counters of the gather kernels, one pass per counter
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/kb_fetch" -- python3 "$ROOT/tools/kernel_bench.py" gather > "$OUT/kb_fetch.jsonl" 2> "$OUT/kb_fetch.err"
echo "gather FETCH_SIZE done"
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/kb_write" -- python3 "$ROOT/tools/kernel_bench.py" gather > "$OUT/kb_write.jsonl" 2> "$OUT/kb_write.err"
echo "gather WRITE_SIZE done"
# 3. the headline command: kernel stats, then HBM traffic and MFMA busy passes
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -- python3 "$ROOT/bench.py" > "$OUT/bench_prof.json" 2> "$OUT/bench_prof.err"
echo "bench trace done"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/bench_$c" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-split --no-cfg5 > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err"
  echo "bench $c done"
done
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$OUT/bench_mfma" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-split --no-cfg5 > "$OUT/bench_mfma.json" 2> "$OUT/bench_mfma.err"
echo "bench mfma done"
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$OUT/bench_insts" -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-split --no-cfg5 > "$OUT/bench_insts.json" 2> "$OUT/bench_insts.err"
echo "bench instruction counts done"
cd "$ROOT"
python3 tools/kernel_trace_summary.py "$OUT/kb_trace" > "$OUT/kernel_bench_trace.json"
python3 tools/kernel_trace_summary.py "$OUT/bench_trace" > "$OUT/bench_trace.json"
python3 tools/pmc_summary.py "$OUT/kb_fetch" "$OUT/kb_write" > "$OUT/gather_hbm_traffic.json"
python3 tools/pmc_summary.py "$OUT/bench_FETCH_SIZE" "$OUT/bench_WRITE_SIZE" "$OUT/bench_mfma" > "$OUT/bench_hbm_traffic.json"
python3 tools/pmc_summary.py "$OUT/bench_insts" > "$OUT/bench_sq_insts.json"
find "$OUT" -name "*stats*.csv" | head
# raw traces are large: keep the summaries and the stats tables only
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
echo "profile_r2 done"
