"""End-to-end time of the batched evaluators, host arrays in -> HR@10 / NDCG@10 out (uploads, kernels, metrics):
evaluate_protocol at the full 99-negative protocol size (115,056 cases x 100 candidates, ML-1M id space), and the AutoRec / Hybrid
full-catalog evaluators on a synthetic 6,040 x 3,706 rating matrix (4.5 % dense train, 0.5 % dense test)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from recommender_amd.evaluate import evaluate_protocol
    from recommender_amd.helpers import NCF
    torch.manual_seed(0)
    m = NCF(6038, 3533, 32, 3, 0.0, "NeuMF-end").eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "embed" in n:
                p.mul_(25.0)
    m = m.cuda()
    rng = np.random.default_rng(0)
    cases = 115_056
    users = rng.integers(0, 6038, cases)
    items = rng.integers(0, 3533, (cases, 100))
    evaluate_protocol(m, users, items)                 # warm-up (library load, packed weights)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        hr, nd, _, _ = evaluate_protocol(m, users, items)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"what": "evaluate_protocol end to end, host lists in -> HR/NDCG out", "cases": cases, "ms": [round(t, 2) for t in ts],
                      "hr": round(hr, 4), "ndcg": round(nd, 4)}))

    from recommender_amd.autorec.utils.model import AutoRec
    from recommender_amd.evaluate import autorec_ranking_metrics, hybrid_ranking_metrics
    from recommender_amd.hybridautorecncf.model import HybridAutoRecNCF
    train = ((rng.random((6040, 3706)) < 0.045) * rng.integers(1, 6, (6040, 3706))).astype(np.float32)
    test = ((rng.random((6040, 3706)) < 0.005) * rng.integers(1, 6, (6040, 3706))).astype(np.float32)
    for name, model, fn in (("autorec_ranking_metrics", AutoRec(6040, 3706, 500).eval().cuda(), autorec_ranking_metrics),
                            ("hybrid_ranking_metrics", HybridAutoRecNCF(6040, 3706, 64, (128, 64), dropout_rate=0.0).eval().cuda(), hybrid_ranking_metrics)):
        fn(model, train, test)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            hr, nd = fn(model, train, test)
            ts.append((time.perf_counter() - t0) * 1e3)
        print(json.dumps({"what": name + " end to end, host matrices in -> HR/NDCG out", "ms": [round(t, 2) for t in ts],
                          "hr": round(hr, 4), "ndcg": round(nd, 4)}))


if __name__ == "__main__":
    main()
