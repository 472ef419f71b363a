#!/usr/bin/env python3
"""Request latency of the serving callers, host wall clock including the D2H of the answer.

    python tools/latency_bench.py > profiles/rNN_latency.jsonl

One line per caller: p50 / p90 in microseconds over `reps` back-to-back requests (each ends in a .cpu(), i.e. a sync),
plus the same call on the reference's own path restated with stock torch modules on the host CPU (oracle/torch_ref.py)
where that exists.  Synthetic ML-1M-shaped weights (seeded), as everywhere.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from recommender_amd.api_core import AutoRecInferenceEngine, NCFInferenceEngine
from recommender_amd.autorec.utils.model import AutoRec
from recommender_amd.helpers import NCF

dev = torch.device("cuda", 0)


def wall(fn, reps=200, warm=10):
    for _ in range(warm):
        fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    t = np.asarray(t) * 1e6
    return float(np.percentile(t, 50)), float(np.percentile(t, 90))


def emit(caller, shape, p50, p90, cpu_us=None, note=""):
    print(json.dumps({"caller": caller, "shape": shape, "p50_us": round(p50, 1), "p90_us": round(p90, 1),
                      "cpu_reference_us": None if cpu_us is None else round(cpu_us, 1), "note": note}), flush=True)


def main():
    torch.manual_seed(0)
    m = NCF(6038, 3533, 32, 3, 0.0, "NeuMF-end").eval().to(dev)
    eng = NCFInferenceEngine(m)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    from oracle import torch_ref
    rng = np.random.default_rng(0)

    users = rng.integers(0, 6038, 512).tolist()
    it = iter(users * 4)
    p50, p90 = wall(lambda: eng.recommend(next(it), 10))
    all_items = torch.arange(3533)

    def cpu_recommend(u=7):
        s = torch_ref.ncf_forward(sd, torch.full((3533,), u, dtype=torch.int64), all_items).numpy()
        return np.argsort(s)[::-1][:10]
    c50, _ = wall(cpu_recommend, reps=30, warm=3)
    emit("NCFInferenceEngine.recommend", "1 user x 3533 items, k=10", p50, p90, c50, "fused score + top-k, one D2H of k ids/scores")

    cand = rng.integers(0, 3533, 100).tolist()
    it = iter(users * 4)
    p50, p90 = wall(lambda: eng.predict_batch(next(it), cand))
    emit("NCFInferenceEngine.predict_batch", "1 user x 100 items", p50, p90)
    it = iter(users * 4)
    p50, p90 = wall(lambda: eng.predict(next(it), 11))
    emit("NCFInferenceEngine.predict", "1 pair", p50, p90)

    u64 = torch.tensor(users[:64]);
    p50, p90 = wall(lambda: [x.cpu() for x in m.score_topk(u64, 10)])
    emit("NCF.score_topk", "64 users x 3533 items, k=10", p50, p90)

    torch.manual_seed(0)
    a = AutoRec(6040, 3706, 500, True).eval().to(dev)
    R = ((torch.rand(6040, 3706) < 0.045) * torch.randint(1, 6, (6040, 3706))).float()
    t0 = time.perf_counter()
    aeng = AutoRecInferenceEngine(a, R.to(dev))
    torch.cuda.synchronize()
    emit("AutoRecInferenceEngine.__init__ (cache_hidden)", "X[3706,6040] -> H[3706,500]", (time.perf_counter() - t0) * 1e6, 0.0)
    it = iter(users * 4)
    p50, p90 = wall(lambda: aeng.recommend(next(it), 10))
    emit("AutoRecInferenceEngine.recommend", "1 user x 3706 items, k=10", p50, p90, None,
         "reference: 3,706 B=1 forwards + 3,706 host syncs per request (inference_engine.py:266-294)")
    it = iter(users * 4)
    p50, p90 = wall(lambda: aeng.predict_batch(next(it), cand))
    emit("AutoRecInferenceEngine.predict_batch", "1 user x 100 items", p50, p90)


if __name__ == "__main__":
    main()
