#!/usr/bin/env python3
"""The 11.5 M-pair protocol list through forward(user, item) a few times (for rocprofv3 --pmc passes): python3 tools/pair_one.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommender_amd.helpers import NCF
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = NCF(6038, 3533, 32, 3, 0.0, "NeuMF-end").eval().to(dev)
t = m._tower(dev)
g = torch.Generator().manual_seed(1)
u = torch.randint(0, 6038, (115056,), generator=g).repeat_interleave(100).to(dev)
i = torch.randint(0, 3533, (115056 * 100,), generator=g).to(dev)
for _ in range(reps):
    t.score_pairs(u, i, check_ids=False)
torch.cuda.synchronize()
