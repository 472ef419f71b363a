#!/bin/bash
# Round-3 end-to-end check on the GPU box: GPU tests, smoke, bench plain vs under torch.distributed.run (N = 1), the 2-rank
# shared-GPU rehearsal of the N > 1 code path (gloo: numbers meaningless), and a kernel trace of the torchrun form.
set -e -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/r3a}")
ROOT=$(pwd)
mkdir -p "$OUT"
if [ "${SKIP_TESTS:-0}" != 1 ]; then
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > "$OUT/gpu_tests.log" 2>&1 || { tail -30 "$OUT/gpu_tests.log"; exit 1; }
tail -3 "$OUT/gpu_tests.log"
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 || { tail -20 "$OUT/smoke.log"; exit 1; }
tail -1 "$OUT/smoke.log"
fi
TR="python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
for rep in 1 2; do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_plain_$rep.json" 2> "$OUT/bench_plain_$rep.err" || { tail -20 "$OUT/bench_plain_$rep.err"; exit 1; }
  timeout -k 10 300 $TR --master-port $((29710 + rep)) bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_torchrun_$rep.json" 2> "$OUT/bench_torchrun_$rep.err" || { tail -20 "$OUT/bench_torchrun_$rep.err"; exit 1; }
  echo "plain / torchrun pair $rep done"
done
export RECO_GATHER_TIMEOUT_S=20
timeout -k 10 400 python3 bench.py --gpus 2 --share-gpus --steps 5 --warmup 2 --no-cpu-baseline --launch-timeout 150 > "$OUT/bench_n2_rehearsal.json" 2> "$OUT/bench_n2_rehearsal.err" || { tail -20 "$OUT/bench_n2_rehearsal.err"; exit 1; }
echo "2-rank rehearsal done"
RECO_RESERVE_CUS=8 timeout -k 10 400 python3 bench.py --gpus 2 --share-gpus --steps 5 --warmup 2 --no-cpu-baseline --no-split --no-cfg5 --launch-timeout 150 > "$OUT/bench_n2_rehearsal_reserve8.json" 2> "$OUT/bench_n2_rehearsal_reserve8.err" || { tail -20 "$OUT/bench_n2_rehearsal_reserve8.err"; exit 1; }
cd /tmp && export TMPDIR=/tmp
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29741 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pg_trace" -- python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-split --no-cfg5 --no-kernels > "$OUT/pg_trace.json" 2> "$OUT/pg_trace.err"
cd "$ROOT"
python3 tools/kernel_trace_summary.py "$OUT/pg_trace" > "$OUT/pg_trace_summary.json"
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
python3 - "$OUT" <<'EOF'
import json, sys, os, glob
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "bench_*.json")) + [os.path.join(out, "pg_trace.json")]):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    c5 = d.get("cfg5_top100") or {}
    print("%-34s n=%d cfg2 %.4f ms/step kernel %.4f frac %.4f | cfg5 %s ms | %s" % (os.path.basename(f), d["n_gpus"], d["ms_per_step"], d["roofline"]["kernel_ms"],
          d["roofline"]["frac"], c5.get("ms_per_step"), d["config"]["parallelism"][:90]))
EOF
echo "r3_check done"
