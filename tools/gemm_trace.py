#!/usr/bin/env python3
"""Per-workgroup phase timeline of linear_kernel from a -DRECO_DIAG_GEMM_TRACE build (tools/build_variant.sh WORK trace
-DRECO_DIAG_GEMM_TRACE):   RECO_HIP_LIB=recommender_amd/lib/ab/trace.so python tools/gemm_trace.py [M N K] [swap]
Default: the AutoRec decoder as the product runs it (operands swapped: M = 6040 users, N = 3706 items, K = 500, bias per row).
Stamps of thread 0 (100 MHz wall clock): start, first tile staged, K loop done, stores issued; plus the CU it ran on."""
import collections
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from recommender_amd import _lib, ops

M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else ("6040", "3706", "500")))
g = torch.Generator(device="cuda").manual_seed(1)
A = torch.randn(M, K, device="cuda", generator=g)
W = torch.randn(N, K, device="cuda", generator=g) * 0.05
b = torch.randn(N, device="cuda", generator=g)
for _ in range(int(os.environ.get("TRACE_WARM", "400"))):
    ops.linear(A, W, b, act="clamp", lo=1.0, hi=5.0)
torch.cuda.synchronize()
L = _lib.lib()
L.reco_debug_gemm_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
n = min(int(os.environ.get("TRACE_WGS", ((M + 127) // 128) * ((N + 127) // 128))), 8192)
buf = np.zeros(8 * n, dtype=np.uint64)
assert L.reco_debug_gemm_trace(buf.ctypes.data, 8 * n) == 0
rec = buf.reshape(n, 8)
t = rec[:, :4].astype(np.int64)
t = (t - t[:, 0].min()) * 10 / 1e3          # us
hw = rec[:, 4]
hwid = (hw & 0xffffffff).astype(np.int64); xcc = ((hw >> 32) & 0xf).astype(np.int64)
cu = list(zip(xcc.tolist(), ((hwid >> 13) & 7).tolist(), ((hwid >> 12) & 1).tolist(), ((hwid >> 8) & 15).tolist()))
out = {"workgroups": int(n), "span_us": float(t[:, 3].max())}
for i, nm in enumerate(["start", "staged", "loop_end", "end"]):
    out[nm + "_us"] = {p: round(float(np.percentile(t[:, i], p)), 2) for p in (0, 10, 50, 90, 100)}
d = np.diff(t, axis=1)
for i, nm in enumerate(["prologue", "k_loop", "epilogue"]):
    out["phase_" + nm + "_us"] = {p: round(float(np.percentile(d[:, i], p)), 2) for p in (0, 10, 50, 90, 100)}
out["life_us"] = {p: round(float(np.percentile(t[:, 3] - t[:, 0], p)), 2) for p in (0, 10, 50, 90, 100)}
per_cu = collections.Counter(cu)
out["cus_used"] = len(per_cu)
out["workgroups_per_cu_hist"] = dict(sorted(collections.Counter(per_cu.values()).items()))
# per CU: when its last workgroup ended
last = collections.defaultdict(float)
for c, e in zip(cu, t[:, 3]):
    last[c] = max(last[c], e)
lv = np.array(list(last.values()))
out["cu_last_end_us"] = {p: round(float(np.percentile(lv, p)), 2) for p in (0, 10, 50, 90, 100)}
step = max(int(out["span_us"] // 40), 1)
out["alive_every_%dus" % step] = [int(((t[:, 0] <= u) & (t[:, 3] > u)).sum()) for u in range(0, int(out["span_us"]) + 1, step)]
out["in_k_loop_every_%dus" % step] = [int(((t[:, 1] <= u) & (t[:, 2] > u)).sum()) for u in range(0, int(out["span_us"]) + 1, step)]
print(json.dumps(out))
