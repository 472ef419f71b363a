#!/usr/bin/env python3
"""Phase timeline of mlp2_kernel from a -DRECO_DIAG_MLP2_TRACE build (tools/build_variant.sh WORK trace -DRECO_DIAG_MLP2_TRACE):
   RECO_HIP_LIB=recommender_amd/lib/ab/trace.so python tools/mlp2_trace.py [rows] [K1] [N2] [N3]
Stamps of wave 0 of every workgroup (100 MHz wall clock): start, rows staged, layer 1 done, second barrier passed, end."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from recommender_amd import _lib, ops

rows, K1, N2, N3 = (int(x) for x in (sys.argv[1:5] + ["6040", "256", "128", "64"][len(sys.argv) - 1:]))
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.randn(rows, K1, device="cuda", generator=g)
W2 = torch.randn(N2, K1, device="cuda", generator=g) * 0.05
b2 = torch.randn(N2, device="cuda", generator=g)
W3 = torch.randn(N3, N2, device="cuda", generator=g) * 0.05
b3 = torch.randn(N3, device="cuda", generator=g)
for _ in range(int(os.environ.get("TRACE_WARM", "300"))):
    ops.mlp2(X, W2, b2, W3, b3, "relu", None)
torch.cuda.synchronize()
L = _lib.lib()
L.reco_debug_mlp2_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
n = min((rows + 31) // 32, 1024)
buf = np.zeros(8 * n, dtype=np.uint64)
assert L.reco_debug_mlp2_trace(buf.ctypes.data, 8 * n) == 0
t = buf.reshape(n, 8)[:, :5].astype(np.int64)
t = (t - t[:, 0].min()) * 10 / 1e3          # us
names = ["start", "staged", "layer1", "barrier2", "end"]
out = {"workgroups": int(n), "span_us": float(t[:, 4].max())}
for i, nm in enumerate(names):
    out[nm + "_us"] = {p: round(float(np.percentile(t[:, i], p)), 2) for p in (0, 50, 100)}
d = np.diff(t, axis=1)
for i, nm in enumerate(["stage", "layer1", "wait2", "layer2+store"]):
    out["phase_" + nm + "_us"] = {p: round(float(np.percentile(d[:, i], p)), 2) for p in (0, 50, 100)}
print(json.dumps(out))
