#!/bin/bash
# Timing-only GEMM ablations against the shipped library, alternating on one box.  Build the variants first (csrc/):
#   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DRECO_ABL_GEMM_NO_STORE -c gemm.hip -o /tmp/g.o   (and _NO_LOAD)
#   hipcc --offload-arch=gfx950 -shared -fPIC <the other objects of build/csrc> /tmp/g.o -o ../lib/ab/g_NO_STORE.so ; cp ../lib/libreco_hip.so ../lib/ab/base.so
# The variants compute wrong results by construction: never ship them (recommender_amd/lib/ab/ is deleted after use).
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
