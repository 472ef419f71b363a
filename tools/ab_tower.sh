#!/bin/bash
# same-box A/B of library variants (recommender_amd/lib/ab/<name>.so, selected through RECO_HIP_LIB) on the tower lines of the
# kernel bench, two alternating rounds:   AB_FILTER=split bash tools/ab_tower.sh base new
set -e -o pipefail
for r in 1 2; do
  for v in "$@"; do
    export RECO_HIP_LIB=$PWD/recommender_amd/lib/ab/$v.so
    timeout -k 10 120 python tools/kernel_bench.py tower 2>/dev/null | V=$v python -c "
import sys,json,os,re
for l in sys.stdin:
    d=json.loads(l)
    if re.search(os.environ.get('AB_FILTER','pairs'), d['kernel']): print(os.environ['V'], d['kernel'][:48], d['shape'][:24], round(d['ms'],4))
"
  done
done
