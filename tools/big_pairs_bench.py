"""Pair-list forward over tables of 2^23 rows or more (64-bit row offsets, rows from HBM): 9 M users x 100 k items,
11.5 M random pairs and 11.5 M pairs in runs of 100 per user."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.fused_ab import timed  # noqa: E402


def main():
    from recommender_amd import ops
    from recommender_amd.helpers import NCF
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(5)
    U, N, B = 9_000_000, 100_000, 11_505_600

    def table(rows, dim):
        return torch.randn(rows, dim, device=dev, generator=g) * 0.25

    torch.manual_seed(0)
    small = NCF(8, 8, 32, 3, 0.0, "NeuMF-end").eval()
    layers = [(l.weight.detach().to(dev), l.bias.detach().to(dev)) for l in small.MLP_layers if isinstance(l, torch.nn.Linear)]
    tower = ops.Tower(user_gmf=table(U, 32), item_gmf=table(N, 32), user_mlp=table(U, 128), item_mlp=table(N, 128), layers=layers,
                      w_predict=small.predict_layer.weight.detach().to(dev), b_predict=small.predict_layer.bias.detach().to(dev))
    items = torch.randint(0, N, (B,), device=dev, generator=g)
    rnd = torch.randint(0, U, (B,), device=dev, generator=g)
    runs = torch.randint(0, U, (B // 100,), device=dev, generator=g).repeat_interleave(100)
    res = {"lib": os.environ.get("RECO_HIP_LIB", "shipped")}
    for name, users in (("random_users_ms", rnd), ("runs_of_100_ms", runs)):
        res[name] = [round(timed(lambda: tower.score_pairs(users, items, check_ids=False), 10), 4) for _ in range(2)]
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
