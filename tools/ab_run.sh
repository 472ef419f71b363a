#!/bin/bash
# same-box A/B of library variants built by tools/build_variant.sh: alternating rounds of tools/ab_probe.py
#   AB_WORK="cfg2 cfg1" bash tools/ab_run.sh base new
set -e -o pipefail
for r in 1 2; do
  for v in "$@"; do
    RECO_HIP_LIB=$PWD/recommender_amd/lib/ab/$v.so timeout -k 10 300 python3 tools/ab_probe.py ${AB_WORK:-cfg2 cfg1} 2>/dev/null
  done
done
