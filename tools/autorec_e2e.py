#!/usr/bin/env python3
"""AutoRec cfg3 end to end (reconstruct_all + seen mask + top-10) a few times, for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommender_amd.autorec.utils.model import AutoRec
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = AutoRec(6040, 3706, 500, True).eval().to(dev)
R = ((torch.rand(6040, 3706, device=dev) < 0.045) * torch.randint(1, 6, (6040, 3706), device=dev)).float()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3):
    m.recommend_topk(R, 10)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    m.recommend_topk(R, 10)
e1.record(); torch.cuda.synchronize()
print("ms per call: %.4f" % (e0.elapsed_time(e1) / reps))
