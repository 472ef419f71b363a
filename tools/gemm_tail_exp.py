#!/usr/bin/env python3
"""Experiment: the AutoRec decoder GEMM as ONE launch against a main launch of whole rounds of 128x128 tiles + a tail launch with a
smaller tile shape (needs a -DRECO_DIAG_GEMM_TILE build: the tail's shape is forced through RECO_GEMM_TILE).  Kernel times from
the library's hipEvent hook."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from recommender_amd import _lib, ops

M, N, K = 6040, 3706, 500
g = torch.Generator(device="cuda").manual_seed(1)
A = torch.randn(M, K, device="cuda", generator=g)
W = torch.randn(N, K, device="cuda", generator=g) * 0.05
b = torch.randn(N, device="cuda", generator=g)
L = _lib.lib()


def run(split_rows, tail_tile, reps=30):
    def step():
        if split_rows is None:
            os.environ.pop("RECO_GEMM_TILE", None)
            return [ops.linear(A, W, b, act="clamp", lo=1.0, hi=5.0)]
        os.environ.pop("RECO_GEMM_TILE", None)
        o1 = ops.linear(A[:split_rows], W, b, act="clamp", lo=1.0, hi=5.0)
        if tail_tile:
            os.environ["RECO_GEMM_TILE"] = tail_tile
        o2 = ops.linear(A[split_rows:], W, b, act="clamp", lo=1.0, hi=5.0)
        os.environ.pop("RECO_GEMM_TILE", None)
        return [o1, o2]
    for _ in range(300):
        step()
    torch.cuda.synchronize()
    cap = reps * 4
    _lib.check(L.reco_profile_enable(cap))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    ms = (ctypes.c_float * cap)(); n = ctypes.c_int(0)
    _lib.check(L.reco_profile_collect(ms, cap, ctypes.byref(n)))
    _lib.check(L.reco_profile_enable(0))
    per = n.value // reps
    ks = [float(np.mean([ms[i + r * per] for r in range(reps)])) for i in range(per)]
    return {"split_rows": split_rows, "tail_tile": tail_tile, "kernels_ms": [round(x, 4) for x in ks], "sum_ms": round(sum(ks), 4),
            "wall_ms": round(ev0.elapsed_time(ev1) / reps, 4)}


print(json.dumps(run(None, None)), flush=True)
for rows in (35 * 128, 32 * 128, 40 * 128):
    for tile in (None, "128x64", "64x128", "64x64"):
        print(json.dumps(run(rows, tile)), flush=True)
print(json.dumps(run(None, None)), flush=True)
