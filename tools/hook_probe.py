#!/usr/bin/env python3
"""Does arming the library's hipEvent hook change what it measures?  Wall time per step with and without the hook."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommender_amd import _lib, ops
from recommender_amd.helpers import NCF
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = NCF(6038, 3533, 32, 3, 0.0, "NeuMF-end").eval().to(dev)
t = m._tower(dev)
L = _lib.lib()
ws = t.topk_workspace(6038, 3533, 10)
out = (torch.empty((6038, 10), dtype=torch.int32, device=dev), torch.empty((6038, 10), dtype=torch.float32, device=dev))
g = torch.Generator().manual_seed(1)
u = torch.randint(0, 6038, (115056,), generator=g).repeat_interleave(100).to(dev)
i = torch.randint(0, 3533, (115056 * 100,), generator=g).to(dev)

def cfg2(): t.score_topk(None, 10, None, None, check_ids=False, ws=ws, out=out)
def cfg1(): ops.topk_rows(t.score_pairs(u, i, check_ids=False).view(115056, 100), 10)

def wall(fn, steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3

for name, fn in (("cfg2", cfg2), ("cfg1", cfg1)):
    for _ in range(60): fn()
    for rep in range(2):
        a = wall(fn, 20)
        _lib.check(L.reco_profile_enable(200))
        b = wall(fn, 20)
        ms = (ctypes.c_float * 200)(); kinds = (ctypes.c_int32 * 200)(); n = ctypes.c_int(0)
        _lib.check(L.reco_profile_collect_kinds(ms, kinds, 200, ctypes.byref(n)))
        _lib.check(L.reco_profile_enable(0))
        per = {}
        for j in range(n.value): per.setdefault(kinds[j], []).append(ms[j])
        c = wall(fn, 20)
        print(name, "no hook %.4f | hook %.4f | no hook %.4f ms/step; by kind:" % (a, b, c), {k: round(sum(v) / len(v), 4) for k, v in per.items()}, flush=True)
