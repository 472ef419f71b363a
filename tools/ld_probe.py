#!/usr/bin/env python3
"""AutoRec decoder with swapped operands: does the output row stride matter?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommender_amd import ops
dev = torch.device("cuda", 0)
h = torch.randn(3706, 500, device=dev); wd = torch.randn(6040, 500, device=dev) * 0.05; bd = torch.randn(6040, device=dev)
def t(fn, reps=30):
    for _ in range(10): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
for ld in (3706, 3708, 3712, 3744, 3776, 3840, 3968, 4096):
    buf = torch.empty((6040, ld), device=dev)
    print("swapped ld", ld, "%.4f ms" % t(lambda: ops.linear(wd, h, bd, "clamp", 1.0, 5.0, bias_per_row=True, out=buf[:, :3706])), flush=True)
    print("swapped ld", ld, "no bias %.4f ms" % t(lambda: ops.linear(wd, h, None, "clamp", 1.0, 5.0, out=buf[:, :3706])), flush=True)
for ld in (6040, 6048, 6144):
    buf = torch.empty((3706, ld), device=dev)
    print("plain ld", ld, "%.4f ms" % t(lambda: ops.linear(h, wd, bd, "clamp", 1.0, 5.0, out=buf[:, :6040])), flush=True)
for (M, N, K) in ((4096, 4096, 4096), (4096, 4096, 512), (6040, 256, 3706), (3706, 500, 6040), (8192, 1024, 1024), (6038, 128, 128)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.02
    print("square-ish", (M, N, K), "%.4f ms" % t(lambda: ops.linear(A, W), reps=15), flush=True)
