#!/bin/bash
# quick round-3 check on the GPU box: GPU tests, then bench plain vs torchrun (N = 1)
set -e -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/r3a}")
mkdir -p "$OUT"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > "$OUT/gpu_tests.log" 2>&1 || { tail -30 "$OUT/gpu_tests.log"; exit 1; }
tail -3 "$OUT/gpu_tests.log"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err" || { tail -20 "$OUT/bench_plain.err"; exit 1; }
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernels > "$OUT/bench_torchrun.json" 2> "$OUT/bench_torchrun.err" || { tail -20 "$OUT/bench_torchrun.err"; exit 1; }
python3 - "$OUT" <<'EOF'
import json, sys, os
out = sys.argv[1]
for f in ("bench_plain.json", "bench_torchrun.json"):
    d = json.loads(open(os.path.join(out, f)).read().strip().splitlines()[-1])
    print(f, "cfg2 %.4f ms/step kernel %.4f frac %.4f | cfg5 %.2f ms" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["cfg5_top100"]["ms_per_step"]))
    for k in ("gather", "cfg1_protocol", "cfg3_autorec", "cfg4_hybrid"):
        if k in d:
            v = d[k]
            print("  ", k, {x: (round(v[x], 4) if isinstance(v[x], float) else v[x]) for x in ("kernel", "kernel_ms", "frac", "ms_per_step", "error") if x in v})
EOF
