#!/bin/bash
# the long hunt: the seeded sweeps of tests/test_gpu_parity.py with 25 times the cases from another seed (GPU box, ~2 minutes)
#   bash tools/fuzz_run.sh [seed] [mult]
set -e -o pipefail
RECO_FUZZ_SEED=${1:-1} RECO_FUZZ_MULT=${2:-25} timeout -k 10 900 python3 -u -m pytest tests/test_gpu_parity.py tests/test_zz_no_alloc.py -x -q -m gpu --timeout 600 --timeout-method thread -k "sweep"
