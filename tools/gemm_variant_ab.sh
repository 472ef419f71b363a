#!/bin/bash
# same-box A/B of library variants (recommender_amd/lib/ab/<name>.so) on the GEMM lines of the kernel bench:  bash tools/gemm_variant_ab.sh base dma
set -e -o pipefail
for r in 1 2; do for v in "$@"; do
RECO_HIP_LIB=$PWD/recommender_amd/lib/ab/$v.so timeout -k 10 120 python tools/kernel_bench.py autorec tower 2>/dev/null | V=$v python -c "
import sys,json,os
o=[]
for l in sys.stdin:
    d=json.loads(l); k=d['kernel']
    if k.startswith('linear_kernel (') or k.startswith('hybrid encoders') or 'end to end, rating matrix kept sparse' in k or k.startswith('hybrid score_all,'): o.append('%s=%.4f'%(k[:30],d['ms']))
print(os.environ['V'],' | '.join(o))
"
done; done
