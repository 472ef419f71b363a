#!/usr/bin/env python3
"""One GEMM shape a few times (for rocprofv3 --pmc passes): python3 tools/gemm_one.py M N K [reps] [precision]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommender_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
prec = sys.argv[5] if len(sys.argv) > 5 else "f32"
dev = torch.device("cuda", 0)
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev)
for _ in range(reps):
    ops.linear(x, w, b, "clamp", 1.0, 5.0, out=out, precision=prec)
torch.cuda.synchronize()
