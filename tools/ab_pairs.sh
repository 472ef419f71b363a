#!/bin/bash
# same-box A/B of library variants (recommender_amd/lib/ab/<name>.so, selected through RECO_HIP_LIB) on the pair-kernel lines of
# the kernel bench:   bash tools/ab_pairs.sh base aonly bonly
set -e -o pipefail
for r in 1 2; do
  for v in "$@"; do
    export RECO_HIP_LIB=$PWD/recommender_amd/lib/ab/$v.so
    timeout -k 10 120 python tools/kernel_bench.py tower 2>/dev/null | V=$v python -c "
import sys,json,os
for l in sys.stdin:
    d=json.loads(l)
    if 'pairs_kernel' in d['kernel'] and 'split' not in d['kernel']: print(os.environ['V'], d['kernel'][:40], d['shape'][:30], round(d['ms'],4))
"
  done
done
