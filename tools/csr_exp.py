#!/usr/bin/env python3
"""CSR encoder layer against the number of k ranges (XCD placement) for the AutoRec and Hybrid weight sizes: the sweep behind ops.csr_k_ranges (DESIGN section 3 K-c')."""
import os, sys, time, json
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tools"))
import torch, numpy as np
from recommender_amd import ops, _lib
from bench_kernels import _ml1m_like_matrix
dev = torch.device("cuda", 0)
R = _ml1m_like_matrix(dev)
def t(fn, reps=30):
    for _ in range(5): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
for side, tr, K in (("user", False, 3706), ("item", True, 6040)):
    for N in (256, 500):
        W = torch.randn(N, K, device=dev) * 0.02; b = torch.randn(N, device=dev)
        Wt = ops.transpose(W)
        for kr in (1, 2, 4, 8):
            csr = ops.CsrMatrix.from_dense(R, transpose=tr, k_ranges=kr)
            ms = t(lambda: ops.csr_linear(csr, Wt, b, "relu"))
            print(side, "N", N, "k_ranges", kr, "segs", csr.n_seg, "single", csr.single_segment_rows, "ms %.4f" % ms, flush=True)
        X = R.t().contiguous() if tr else R
        ms = t(lambda: ops.linear(X, W, b, "relu"))
        print(side, "N", N, "dense ms %.4f" % ms, flush=True)
