#!/usr/bin/env python3
"""Soak: every scheduled-at-run-time path (tile work queue, fused top-k hand-out, CSR partial sums, side-stream encoders) repeated
many times on one box; every repetition must give the first one's bits.   python tools/soak.py [reps]"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from recommender_amd.helpers import NCF
from recommender_amd.autorec.utils.model import AutoRec
from recommender_amd.hybridautorecncf.model import HybridAutoRecNCF

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
torch.manual_seed(0)
rng = np.random.default_rng(0)


def same(a, b):
    return all(torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x, y.view(torch.int32) if y.dtype == torch.float32 else y) for x, y in zip(a, b))


def soak(name, fn):
    t0 = time.time()
    first = [t.clone() for t in fn()]
    bad = 0
    for r in range(reps):
        got = fn()
        if not same(got, first):
            bad += 1
    torch.cuda.synchronize()
    print("%-58s %4d reps  %s  (%.1f s)" % (name, reps, "identical" if bad == 0 else "%d DIFFER" % bad, time.time() - t0), flush=True)
    return bad


m = NCF(6038, 3533, 32, 3, 0.0, "NeuMF-end").eval()
with torch.no_grad():
    for n, p in m.named_parameters():
        if "embed" in n:
            p.mul_(25.0)
m = m.cuda()
tower = m._tower(dev)
bad = 0
bad += soak("cfg2 score_topk (tile kernel work queue + scan)", lambda: tower.score_topk(None, 10))
bad += soak("cfg2 fused score + top-k (visitor slots)", lambda: tower.score_topk(None, 10, fused=True))
users = torch.from_numpy(rng.integers(0, 6038, 4000)).cuda()
bad += soak("cfg2 user list, k = 100, fused", lambda: tower.score_topk(users, 100, fused=True))
pu = torch.from_numpy(rng.integers(0, 6038, 1 << 20)).cuda(); pi = torch.from_numpy(rng.integers(0, 3533, 1 << 20)).cuda()
bad += soak("pair list 2^20", lambda: (m(pu, pi),))
big = NCF(4096, 300_000, 32, 3, 0.0, "NeuMF-end").eval().cuda()
bt = big._tower(dev)
bu = torch.arange(512, device=dev)
bad += soak("512 users x 300k items top-100 (fused, several visitors per block)", lambda: bt.score_topk(bu, 100))
R = ((rng.random((6040, 3706)) < 0.045) * rng.integers(1, 6, (6040, 3706))).astype(np.float32)
Rd = torch.from_numpy(R).cuda()
a = AutoRec(6040, 3706, 500, True).eval().cuda()
bad += soak("AutoRec recommend_topk dense", lambda: a.recommend_topk(Rd, 10))
a.attach_ratings(Rd)
bad += soak("AutoRec recommend_topk CSR (partial sums + reduce)", lambda: a.recommend_topk(Rd, 10))
h = HybridAutoRecNCF(6040, 3706, 64, [128, 64], rating_output=False).eval().cuda()
bad += soak("Hybrid score_all dense (two streams)", lambda: h.score_all(Rd, k=10, mask_seen=True))
h.attach_ratings(Rd)
bad += soak("Hybrid score_all CSR (two streams)", lambda: h.score_all(Rd, k=10, mask_seen=True))
print("soak", "ok" if bad == 0 else "FAILED: %d repetitions differed" % bad)
sys.exit(1 if bad else 0)
