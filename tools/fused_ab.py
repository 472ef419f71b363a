"""A/B in one process, same clocks: reco_tower_score_topk_fused (one kernel scores and selects, no score rows in memory)
against reco_tower_score_topk (score rows in the workspace + the stand-alone select).  One JSON line per shape.
NOTE: for mask-free calls whose score rows need several user chunks reco_tower_score_topk takes the fused form by itself
(api.hip fused_topk_preferred), so for those shapes both legs run the same kernel."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


SHAPES = [(6038, 3533, 6038, 10, 20), (6038, 3533, 6038, 100, 20), (2048, 1_000_000, 1024, 100, 3),
          (2048, 200_000, 16, 100, 10), (100_000, 3533, 100_000, 10, 5)]


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="", help="comma-separated indices into SHAPES (default: all)")
    ap.add_argument("--reps", type=int, default=0, help="override the per-shape repeat count (counter passes: 2)")
    ap.add_argument("--custom", default="", help="extra shapes as users:items:k:reps, comma separated (instead of SHAPES)")
    args = ap.parse_args()
    from recommender_amd.helpers import NCF
    dev = torch.device("cuda", 0)
    picked = [SHAPES[int(i)] for i in args.shapes.split(",")] if args.shapes else SHAPES
    if args.custom:
        picked = []
        for spec in args.custom.split(","):
            u, n, k, r = (int(x) for x in spec.split(":"))
            picked.append((max(u, 2048), n, u, k, r))
    for n_users, n_items, U, k, reps in picked:
        reps = args.reps or reps
        torch.manual_seed(0)
        m = NCF(n_users, n_items, 32, 3, 0.0, "NeuMF-end").eval()
        with torch.no_grad():
            for n, p in m.named_parameters():
                if "embed" in n:
                    p.mul_(25.0)
        m = m.to(dev)
        t = m._tower(dev)
        users = (0, U)
        ws = t.topk_workspace(U, n_items, k)
        wsf = t.topk_workspace(U, n_items, k, fused=True)
        out = (torch.empty((U, k), dtype=torch.int32, device=dev), torch.empty((U, k), dtype=torch.float32, device=dev))
        res = {}
        for rnd in range(2):
            res.setdefault("fused_ms", []).append(timed(lambda: t.score_topk(users, k, ws=wsf, out=out, check_ids=False, fused=True), reps))
            res.setdefault("two_kernel_ms", []).append(timed(lambda: t.score_topk(users, k, ws=ws, out=out, check_ids=False), reps))
        print(json.dumps({"users": U, "items": n_items, "k": k, **{a: [round(x, 4) for x in v] for a, v in res.items()},
                          "workspace_MB": {"fused": round(wsf.numel() / 1e6, 1), "two_kernel": round(ws.numel() / 1e6, 1)}}), flush=True)
        del m, t, ws, wsf


if __name__ == "__main__":
    main()
