#!/usr/bin/env python3
"""topk_rows kernel time against row length and k at a fixed row count (the library's hipEvent hook): what a row costs before its first
element and what each 256 elements add.   python tools/scan_probe.py"""
import ctypes, sys
sys.path.insert(0, ".")
import numpy as np
import torch
from recommender_amd import _lib, ops
L = _lib.lib()
dev = torch.device("cuda", 0)


def kernel_ms(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    cap = reps * 4
    _lib.check(L.reco_profile_enable(cap))
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms = (ctypes.c_float * cap)(); kinds = (ctypes.c_int32 * cap)(); n = ctypes.c_int(0)
    _lib.check(L.reco_profile_collect_kinds(ms, kinds, cap, ctypes.byref(n)))
    _lib.check(L.reco_profile_enable(0))
    scan = [ms[i] for i in range(n.value) if kinds[i] == _lib.PROF_TOPK]
    return float(np.median(scan)) if scan else float("nan")


rows = 6038
for N in (1024, 1280, 2048, 3533, 4096, 8192):
    ld = (N + 31) // 32 * 32
    s = torch.randn(rows, ld, device=dev)[:, :N]
    line = []
    for k in (1, 10, 64, 100):
        line.append("k=%d %.4f" % (k, kernel_ms(lambda: ops.topk_rows(s, k))))
    print("rows %d N %5d: %s ms" % (rows, N, "  ".join(line)), flush=True)
# sorted rows: ascending scores = every element enters (worst case), descending = nothing enters after the first k
s = torch.sort(torch.randn(rows, 3552, device=dev)[:, :3533], dim=1).values.contiguous()
pad = torch.zeros(rows, 3552, device=dev); pad[:, :3533] = s
print("ascending rows (every element enters): k=10 %.4f ms" % kernel_ms(lambda: ops.topk_rows(pad[:, :3533], 10)))
pad[:, :3533] = s.flip(1)
print("descending rows (nothing enters after the first k): k=10 %.4f ms" % kernel_ms(lambda: ops.topk_rows(pad[:, :3533], 10)))
