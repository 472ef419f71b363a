#!/bin/bash
# torchrun N=1 with the top-k all-gather forced through a one-rank RCCL group (RECO_FORCE_GATHER=1) against the default (no collective
# at one rank), and the pipeline depth (RECO_GATHER_DEPTH) under the forced gather; alternating rounds.   bash tools/force_gather_ab.sh [out]
set -e -o pipefail
OUT=${1:-gpurun_out/fg}
mkdir -p $OUT
for r in 1 2; do
  for v in "0 2" "1 2" "1 3"; do
    set -- $v
    RECO_FORCE_GATHER=$1 RECO_GATHER_DEPTH=$2 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$r bench.py --gpus 1 --no-cpu-baseline --no-kernels --no-split > $OUT/force$1_depth$2_$r.json 2> $OUT/force$1_depth$2_$r.err
    echo "force=$1 depth=$2 round $r done"
  done
done
