#!/usr/bin/env python3
"""The product's GEMM shapes under each forced tile shape (needs a -DRECO_DIAG_GEMM_TILE build; RECO_GEMM_TILE is read per call):
wall ms per call (kernel + split-K reduce), for checking the launcher's shape model against measurements."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from recommender_amd import ops

g = torch.Generator(device="cuda").manual_seed(1)
R = torch.randn(6040, 3706, device="cuda", generator=g)
SHAPES = [  # name, M, N, K, transposed A
    ("hybrid user encoder 6040x256 K=3706", 6040, 256, 3706, False),
    ("hybrid item encoder 3706x256 K=6040 (A^T)", 3706, 256, 6040, True),
    ("autorec encoder 3706x500 K=6040 (A^T)", 3706, 500, 6040, True),
    ("autorec encoder 3706x500 K=6040", 3706, 500, 6040, False),
    ("autorec decoder 6040x3706 K=500", 6040, 3706, 500, False),
    ("hybrid user decoder 6040x3706 K=256", 6040, 3706, 256, False),
    ("hybrid item decoder 3706x6040 K=256", 3706, 6040, 256, False),
    ("projection 6038x128 K=128", 6038, 128, 128, False),
    ("hidden 6040x128 K=256", 6040, 128, 256, False),
]


def t(fn, reps=40):
    for _ in range(150):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)


for name, M, N, K, at in SHAPES:
    if at:
        A = R if (K, M) == (6040, 3706) else torch.randn(K, M, device="cuda", generator=g)
    else:
        A = R if (M, K) == (6040, 3706) else torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) * 0.05
    b = torch.randn(N, device="cuda", generator=g)
    row = {"shape": name}
    for tile in ("default", "128x128", "128x64", "64x128", "64x64"):
        if tile == "default":
            os.environ.pop("RECO_GEMM_TILE", None)
        else:
            os.environ["RECO_GEMM_TILE"] = tile
        row[tile] = t(lambda: ops.linear(A, W, b, act="relu", x_transposed=at))
    os.environ.pop("RECO_GEMM_TILE", None)
    print(json.dumps(row), flush=True)
