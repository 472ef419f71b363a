#!/usr/bin/env python3
"""Experiment: the AutoRec decoder GEMM writing rows of 3706 floats (14,824 B: every 128-byte store segment straddles two cache
lines) against the same rows padded to 3712 floats (whole lines)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from recommender_amd import ops

M, N, K = 6040, 3706, 500
g = torch.Generator(device="cuda").manual_seed(1)
A = torch.randn(M, K, device="cuda", generator=g)
W = torch.randn(N, K, device="cuda", generator=g) * 0.05
b = torch.randn(N, device="cuda", generator=g)
Ws, As, bs = A, W, torch.randn(M, device="cuda", generator=g)       # swapped form: out[users, items] = W_dec-like operand order


def t(fn, reps=50):
    for _ in range(400):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)


for rep in range(2):
    for ld in (3706, 3712, 3840):
        buf = torch.empty((M, ld), device="cuda")
        out = buf[:, :N]
        print(json.dumps({"ld": ld, "ms": t(lambda: ops.linear(A, W, b, act="clamp", lo=1.0, hi=5.0, out=out)),
                          "ms_bias_per_row": t(lambda: ops.linear(A, W, bs, act="clamp", lo=1.0, hi=5.0, out=out, bias_per_row=True))}), flush=True)
