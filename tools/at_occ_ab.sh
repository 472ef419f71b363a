#!/bin/bash
# same-box A/B: transposed-A GEMM variants compiled for 1 / 4 workgroups per CU, and three K slices instead of four (diagnostic build)
set -e -o pipefail
for r in 1 2; do
  for v in at1 at4; do
    RECO_HIP_LIB=$PWD/recommender_amd/lib/ab/$v.so timeout -k 10 300 python3 tools/ab_probe.py autorec enc 2>/dev/null
  done
  RECO_GEMM_SPLITS=3 RECO_HIP_LIB=$PWD/recommender_amd/lib/ab/at1diag.so timeout -k 10 300 python3 tools/ab_probe.py autorec enc 2>/dev/null | sed 's/at1diag.so/at1-3splits/'
done
