#!/usr/bin/env python3
"""Kernel times of the hot paths through the library's own hipEvent hook, one JSON line per workload; run once per library
variant (RECO_HIP_LIB) by tools/ab_run.sh.      python tools/ab_probe.py [cfg2 cfg1 hybrid rows autorec enc topk] """
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from recommender_amd import _lib, ops
dev = torch.device("cuda", 0)
L = _lib.lib()
KN = {1: "tower", 2: "gather", 3: "linear", 4: "linear_reduce", 5: "csr", 6: "csr_reduce", 7: "topk", 8: "topk_merge", 9: "encoder"}


def measure(name, fn, steps=20, warm=30, cap=64):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    _lib.check(L.reco_profile_enable(steps * cap))
    t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    ms = (ctypes.c_float * (steps * cap))(); kinds = (ctypes.c_int32 * (steps * cap))(); n = ctypes.c_int(0)
    _lib.check(L.reco_profile_collect_kinds(ms, kinds, steps * cap, ctypes.byref(n)))
    _lib.check(L.reco_profile_enable(0))
    per = n.value // steps
    seq = [(KN.get(kinds[i], "?"), round(float(np.median([ms[i + s * per] for s in range(steps)])), 4)) for i in range(per)] if per * steps == n.value else []
    print(json.dumps({"lib": os.path.basename(_lib.LIB_PATH), "workload": name, "wall_ms": round(wall, 4), "kernels_ms": seq,
                      "sum_ms": round(sum(x[1] for x in seq), 4)}), flush=True)


def main(which):
    from recommender_amd.helpers import NCF
    torch.manual_seed(0)
    m = NCF(6038, 3533, 32, 3, 0.0, "NeuMF-end").eval().to(dev)
    t = m._tower(dev)
    if "cfg2" in which:
        ws = t.topk_workspace(6038, 3533, 10)
        out = (torch.empty((6038, 10), dtype=torch.int32, device=dev), torch.empty((6038, 10), dtype=torch.float32, device=dev))
        measure("cfg2 score_topk 6038x3533 k=10", lambda: t.score_topk(None, 10, None, None, check_ids=False, ws=ws, out=out), warm=60)
    if "cfg1" in which:
        g = torch.Generator().manual_seed(1)
        u = torch.randint(0, 6038, (115056,), generator=g).repeat_interleave(100).to(dev)
        i = torch.randint(0, 3533, (115056 * 100,), generator=g).to(dev)
        measure("cfg1 pairs 11.5M + top-10", lambda: ops.topk_rows(t.score_pairs(u, i, check_ids=False).view(115056, 100), 10))
        ur = torch.randint(0, 6038, (115056 * 100,), generator=g).to(dev)
        measure("pairs 11.5M random users", lambda: t.score_pairs(ur, i, check_ids=False))
    if {"hybrid", "rows", "enc"} & set(which):
        from recommender_amd.hybridautorecncf.model import HybridAutoRecNCF
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from bench_kernels import _ml1m_like_matrix
        h = HybridAutoRecNCF(6040, 3706, 64, [128, 64], rating_output=False).eval().to(dev)
        R = _ml1m_like_matrix(dev)
        if "hybrid" in which:
            measure("hybrid score_all dense k=10", lambda: h.score_all(R, k=10))
        if "enc" in which:
            measure("hybrid encoders dense", lambda: (h.user_autorec.encode(R), h.item_autorec.encode(R, transposed=True)))
        h.attach_ratings(R)
        if "hybrid" in which:
            measure("hybrid score_all csr k=10", lambda: h.score_all(R, k=10))
        if "enc" in which:
            pair = h.attach_ratings(R)
            measure("hybrid encoders csr", lambda: (h.user_autorec.encode(pair[0]), h.item_autorec.encode(pair[1])))
        if "rows" in which:
            zu = h.user_autorec.encode(R); zi = h.item_autorec.encode(R.t().contiguous())
            gp = torch.Generator().manual_seed(2)
            zub = zu[torch.randint(0, 6040, (1 << 20,), generator=gp).to(dev)].contiguous()
            zib = zi[torch.randint(0, 3706, (1 << 20,), generator=gp).to(dev)].contiguous()
            th = h.ncf._tower(dev)
            measure("hybrid score_rows 1M pairs", lambda: th.score_rows(zub, zib, zub, zib))
    if "autorec" in which:
        from recommender_amd.autorec.utils.model import AutoRec
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from bench_kernels import _ml1m_like_matrix
        torch.manual_seed(0)
        a = AutoRec(6040, 3706, 500, True).eval().to(dev)
        R = _ml1m_like_matrix(dev)
        measure("autorec recommend_topk dense", lambda: a.recommend_topk(R, 10))
        a.attach_ratings(R)
        measure("autorec recommend_topk csr", lambda: a.recommend_topk(R, 10))
    if "gather" in which:
        U, I, B = 4_000_000, 1_000_000, 1 << 22
        g = torch.Generator(device=dev).manual_seed(11)
        eug, eig = torch.randn(U, 32, device=dev, generator=g), torch.randn(I, 32, device=dev, generator=g)
        eum, eim = torch.randn(U, 128, device=dev, generator=g), torch.randn(I, 128, device=dev, generator=g)
        uu = torch.randint(0, U, (B,), device=dev, generator=g); ii = torch.randint(0, I, (B,), device=dev, generator=g)
        gp = torch.empty((B, 32), dtype=torch.float32, device=dev)
        measure("neumf gather 4M pairs HBM tables", lambda: ops.neumf_gather(eug, eig, eum, eim, uu, ii, check_ids=False, gmf_out=gp), steps=10, warm=20)
        measure("gather_rows D=128 4M rows", lambda: ops.gather_rows(eum, uu, check_ids=False), steps=10, warm=20)
        del eug, eig, eum, eim, uu, ii, gp
        torch.cuda.empty_cache()
    if "dec" in which:
        h = torch.randn(3706, 500, device=dev); wd = torch.randn(6040, 500, device=dev) * 0.05; bd = torch.randn(6040, device=dev)
        buf = torch.empty((6040, 3712), device=dev)
        measure("decoder swapped (M=6040 N=3706, bias per row, ldc 3712)", lambda: ops.linear(wd, h, bd, "clamp", 1.0, 5.0, bias_per_row=True, out=buf[:, :3706]))
        measure("decoder plain (M=3706 N=6040)", lambda: ops.linear(h, wd, bd, "clamp", 1.0, 5.0))
        x = torch.randn(3706, 6040, device=dev); we = torch.randn(500, 6040, device=dev) * 0.02; be = torch.randn(500, device=dev)
        measure("encoder dense (M=3706 N=500 K=6040)", lambda: ops.linear(x, we, be, "sigmoid"))
        xt = x.t().contiguous()
        measure("encoder dense, A transposed", lambda: ops.linear(xt, we, be, "sigmoid", x_transposed=True))
    if "topk" in which:
        s = torch.randn(115056, 100, device=dev)
        measure("topk 115056x100 k=10", lambda: ops.topk_rows(s, 10))
        s2 = torch.randn(6038, 3552, device=dev)[:, :3533]
        measure("topk 6038x3533 k=10", lambda: ops.topk_rows(s2, 10))
        R = ((torch.rand(6040, 3706, device=dev) < 0.045) * torch.randint(1, 6, (6040, 3706), device=dev)).float()
        s3 = torch.randn(6040, 3712, device=dev)[:, :3706]
        measure("topk 6040x3706 k=10 masked by the rating matrix", lambda: ops.topk_rows_seen(s3, 10, R))
        s4 = torch.randn(256, 1_000_000, device=dev)
        measure("topk 256x1M k=100", lambda: ops.topk_rows(s4, 100))
        s5 = torch.randn(2048, 65536, device=dev)
        measure("topk 2048x65536 k=100", lambda: ops.topk_rows(s5, 100))


if __name__ == "__main__":
    main(sys.argv[1:] or ["cfg2", "cfg1"])
