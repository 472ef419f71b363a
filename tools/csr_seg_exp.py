#!/usr/bin/env python3
"""CSR encoder layers against the segment length cap (ops.CsrMatrix.SEG_LEN): fewer, longer segments mean fewer partial sums (none
at all when every row is one segment: no reduce launch) but a worse balance on the popular rows.  Kernel times from the hook."""
import ctypes
import json
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tools"))
import numpy as np
import torch

from recommender_amd import _lib, ops
from bench_kernels import _ml1m_like_matrix, timed, KIND_NAMES

dev = torch.device("cuda", 0)
R = _ml1m_like_matrix(dev)
g = torch.Generator(device=dev).manual_seed(3)
cases = [("hybrid user side N=256", False, 256), ("hybrid item side N=256", True, 256), ("autorec item side N=500", True, 500)]
for seg in [int(x) for x in os.environ.get("SEGS", "256,512,1024,2048,1073741824").split(",")]:
    ops.CsrMatrix.SEG_LEN = seg
    ops.CsrMatrix.SEG_LEN_SKEWED = min(seg, int(os.environ.get("SKEWED", seg)))
    for name, tr, N in cases:
        K = R.shape[0] if tr else R.shape[1]
        X = ops.CsrMatrix.from_dense(R, transpose=tr, k_ranges=int(os.environ["KR"]) if "KR" in os.environ else ops.csr_k_ranges(K, N))
        Wt = torch.randn(K, N, device=dev, generator=g) * 0.05
        b = torch.randn(N, device=dev, generator=g)
        ms, seq = timed(lambda: ops.csr_linear(X, Wt, b, act="relu"), 20, 3, dev)
        print(json.dumps({"seg_len": seg, "case": name, "segments": X.n_seg, "wall_ms": round(ms, 4),
                          "kernels_ms": [(KIND_NAMES.get(k, "?"), round(v, 4)) for k, v, _ in seq]}), flush=True)
