#!/bin/bash
set -e -o pipefail
for r in 1 2; do for v in base g_NO_STORE g_NO_LOAD; do
RECO_HIP_LIB=$PWD/recommender_amd/lib/ab/$v.so timeout -k 10 120 python tools/kernel_bench.py autorec 2>/dev/null | V=$v python -c "
import sys,json,os
o=[]
for l in sys.stdin:
    d=json.loads(l); k=d['kernel']
    if k.startswith('linear_kernel (') : o.append('%s=%.4f'%(k[15:40],d['ms']))
print(os.environ['V'],' '.join(o))
"
done; done
