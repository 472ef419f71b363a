#!/usr/bin/env python3
"""Replays test_linear_random_shape_sweep's random stream up to one case and reports where the linear kernel's error sits.
    python tools/repro_linear_case.py <seed offset> <case>"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from recommender_amd import ops

off, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(2026 + off)
for case in range(want + 1):
    M = int(rng.integers(1, 400)); N = int(rng.integers(1, 300))
    K = int(rng.choice([rng.integers(1, 64), rng.integers(64, 700), rng.integers(1024, 3000)]))
    x = rng.standard_normal((M, K + 5)).astype(np.float32); w = rng.standard_normal((N, K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
print("case", case, "M N K", M, N, K, "view" if case % 3 == 0 else "contiguous", "out view" if case % 4 == 1 else "own out")
xg = torch.from_numpy(x).cuda()[:, 3:3 + K] if case % 3 == 0 else torch.from_numpy(np.ascontiguousarray(x[:, 3:3 + K])).cuda()
out = torch.full((M, N + 4), 7.0, device="cuda")[:, :N] if case % 4 == 1 else None
y = ops.linear(xg, torch.from_numpy(w).cuda(), torch.from_numpy(b).cuda(), "relu", out=out).cpu().numpy()
xs = x[:, 3:3 + K].astype(np.float64)
ref = np.maximum(xs @ w.astype(np.float64).T + b, 0)
scale = np.abs(xs) @ np.abs(w).astype(np.float64).T + np.abs(b)
err = np.abs(y - ref)
i = np.unravel_index(np.argmax(err / (4e-7 * scale + 1e-6)), err.shape)
print("worst", i, "got", y[i], "ref", ref[i], "err", err[i], "tol", 4e-7 * scale[i] + 1e-6, "scale", scale[i])
print("outputs over tol:", int((err > 4e-7 * scale + 1e-6).sum()), "of", err.size, " mean err/scale %.3g  max err/scale %.3g" % ((err / scale).mean(), (err / scale).max()))
# the same chain in numpy fp32, k-ordered fmaf-free (float32 multiply-add): what ANY sequential fp32 sum gives
acc = np.zeros((M, N), np.float32)
xs32 = x[:, 3:3 + K]
for k in range(K):
    acc += xs32[:, k:k + 1] * w[:, k][None, :]
seq = np.maximum(acc + b, 0)
e2 = np.abs(seq - ref)
print("numpy sequential fp32: max err/scale %.3g, over tol %d; at the worst element err %.3g" % ((e2 / scale).max(), int((e2 > 4e-7 * scale + 1e-6).sum()), e2[i]))
print("kernel vs numpy sequential at the worst element:", y[i], seq[i])
