"""Hybrid head shape (dg = dm = 64, two layers): reco_tower_score_topk_fused against the two-kernel form, alternating in one process."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.fused_ab import timed  # noqa: E402


def main():
    import recommender_amd.hybridautorecncf.model as hm
    from recommender_amd import ops
    torch.manual_seed(0)
    head = hm.NCF(64, (128, 64), 0.0, rating_output=True).eval().cuda()
    lin = [m for m in head.mlp if isinstance(m, torch.nn.Linear)]
    for U, N, k, reps in [(6040, 3706, 10, 20), (6040, 3706, 100, 20), (512, 500_000, 100, 3)]:
        zu = torch.randn(U, 64, device="cuda"); zi = torch.randn(N, 64, device="cuda")
        t = ops.Tower(user_gmf=zu, item_gmf=zi, user_mlp=zu, item_mlp=zi,
                      layers=[(l.weight.detach(), l.bias.detach()) for l in lin], w_predict=head.output.weight.detach(),
                      b_predict=head.output.bias.detach(), act="rating", lo=1.0, hi=5.0)
        ws = t.topk_workspace(U, N, k); wsf = t.topk_workspace(U, N, k, fused=True)
        out = (torch.empty((U, k), dtype=torch.int32, device="cuda"), torch.empty((U, k), dtype=torch.float32, device="cuda"))
        res = {"fused_ms": [], "own_choice_ms": []}
        for _ in range(2):
            res["fused_ms"].append(round(timed(lambda: t.score_topk(None, k, ws=wsf, out=out, check_ids=False, fused=True), reps), 4))
            res["own_choice_ms"].append(round(timed(lambda: t.score_topk(None, k, ws=ws, out=out, check_ids=False), reps), 4))
        print(json.dumps({"users": U, "items": N, "k": k, **res}), flush=True)
        del t, ws, wsf


if __name__ == "__main__":
    main()
