#!/usr/bin/env python3
"""Per-wave timeline of the lane top-k kernel from a RECO_DIAG_LANE_TRACE build (tools/build_variant.sh WORK trace
-DRECO_DIAG_LANE_TRACE):   RECO_HIP_LIB=recommender_amd/lib/ab/trace.so python tools/lane_trace.py [rows] [N]
Prints when waves start and end (100 MHz wall clock), how long they live and how they were placed on XCDs / CUs / SIMDs."""
import collections
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from recommender_amd import _lib, ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 115056
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t = torch.randn(rows, N, device="cuda")
for _ in range(200):
    ops.topk_rows(t, 10)
torch.cuda.synchronize()
L = _lib.lib()
L.reco_debug_lane_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
n = min((rows + 63) // 64, 4096)
buf = np.zeros(3 * n, dtype=np.uint64)
assert L.reco_debug_lane_trace(buf.ctypes.data, 3 * n) == 0
t0, t1, hw = buf[0::3].astype(np.int64), buf[1::3].astype(np.int64), buf[2::3]
base = t0.min()
t0 = (t0 - base) * 10            # ns
t1 = (t1 - base) * 10
hwid = (hw & 0xffffffff).astype(np.int64)
xcc = ((hw >> 32) & 0xf).astype(np.int64)
simd = (hwid >> 4) & 3
cu = (hwid >> 8) & 15
sh = (hwid >> 12) & 1
se = (hwid >> 13) & 7
life = t1 - t0
out = {"waves": int(n), "span_us": float(t1.max()) / 1e3,
       "start_us_percentiles": {p: float(np.percentile(t0, p)) / 1e3 for p in (0, 10, 50, 90, 99, 100)},
       "end_us_percentiles": {p: float(np.percentile(t1, p)) / 1e3 for p in (0, 10, 50, 90, 99, 100)},
       "life_us_percentiles": {p: float(np.percentile(life, p)) / 1e3 for p in (0, 10, 50, 90, 100)}}
per_cu = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
per_simd = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist()))
out["xcc_counts"] = dict(sorted(collections.Counter(xcc.tolist()).items()))
out["cus_used"] = len(per_cu)
out["waves_per_cu_hist"] = dict(sorted(collections.Counter(per_cu.values()).items()))
out["waves_per_simd_hist"] = dict(sorted(collections.Counter(per_simd.values()).items()))
# concurrency: how many waves are alive at each microsecond
alive = [int(((t0 <= u * 1000) & (t1 > u * 1000)).sum()) for u in range(0, int(t1.max() // 1000) + 1)]
out["alive_per_us"] = alive
print(json.dumps(out))
