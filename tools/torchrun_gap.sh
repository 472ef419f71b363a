#!/bin/bash
# the driver's SCALE run starts bench.py under `python -m torch.distributed.run` also at N = 1; that form's step
# must equal the plain form's.  Runs both forms back to back (alternating, twice), the in-between forms that isolate the agent
# from the process group, and one kernel trace of the process-group form.  On the GPU box, from the repo root:
#   bash tools/torchrun_gap.sh gpurun_out/r3gap
set -e -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/r3gap}")
ROOT=$(pwd)
mkdir -p "$OUT"
ARGS="--gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-split ${GAP_EXTRA_ARGS:-}"
TR="python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
for rep in 1 2; do
  timeout -k 10 200 python3 bench.py $ARGS > "$OUT/plain_$rep.json" 2> "$OUT/plain_$rep.err"
  echo "plain $rep done"
  timeout -k 10 200 $TR --master-port $((29600 + rep)) bench.py $ARGS > "$OUT/torchrun_$rep.json" 2> "$OUT/torchrun_$rep.err"
  echo "torchrun $rep done"
done
# the process group without the agent (same environment a rank sees, set by hand)
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 timeout -k 10 200 python3 bench.py $ARGS > "$OUT/pg_noagent.json" 2> "$OUT/pg_noagent.err"
echo "process group without agent done"
# the agent's environment without its thread setting
OMP_NUM_THREADS=16 timeout -k 10 200 $TR --master-port 29621 bench.py $ARGS > "$OUT/torchrun_omp16.json" 2> "$OUT/torchrun_omp16.err"
echo "torchrun OMP 16 done"
# blocking gather under torchrun
RECO_BENCH_SYNC_GATHER=1 timeout -k 10 200 $TR --master-port 29631 bench.py $ARGS > "$OUT/torchrun_sync.json" 2> "$OUT/torchrun_sync.err"
echo "torchrun blocking gather done"
# kernel trace of the process-group form (program directly after --)
cd /tmp && export TMPDIR=/tmp
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29641 timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$OUT/pg_trace" -- python3 "$ROOT/bench.py" --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-split --no-cfg5 > "$OUT/pg_trace.json" 2> "$OUT/pg_trace.err"
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$OUT/plain_trace" -- python3 "$ROOT/bench.py" --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-split --no-cfg5 > "$OUT/plain_trace.json" 2> "$OUT/plain_trace.err"
cd "$ROOT"
python3 - "$OUT" <<'EOF'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    c5 = d.get("cfg5_top100") or {}
    print("%-22s cfg2 %.4f ms/step (kernel %.4f)  cfg5 %s ms/step" % (os.path.basename(f), d["ms_per_step"], d["roofline"]["kernel_ms"], c5.get("ms_per_step")))
EOF
echo "torchrun_gap done"
