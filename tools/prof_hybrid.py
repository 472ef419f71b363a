import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommender_amd.hybridautorecncf.model import HybridAutoRecNCF
from recommender_amd.autorec.utils.model import AutoRec
torch.manual_seed(0)
dev = torch.device("cuda", 0)
h = HybridAutoRecNCF(6040, 3706, 64, [128, 64], rating_output=False).eval().to(dev)
R = (torch.rand(6040, 3706, device=dev) < 0.045).float()
for _ in range(5):
    h.score_all(R, k=10)
a = AutoRec(6040, 3706, 500, True).eval().to(dev)
for _ in range(5):
    a.recommend_topk(R * 3, 10)
torch.cuda.synchronize()
