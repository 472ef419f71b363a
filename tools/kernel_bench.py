#!/usr/bin/env python3
"""Per-kernel roofline measurements (not the headline bench): every hot kernel against its own bound.

    python tools/kernel_bench.py > profiles/rNN_kernel_bench.jsonl

Timing: torch.cuda.Event pairs on the current stream (the stream the wrappers launch on), median of `reps` after warm-up.
Algorithmic bytes / flops per unit are the figures of SURVEY 8(d) / DESIGN.md section 3.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from recommender_amd import ops

HBM_PEAK = 8000.0      # GB/s, MI355X_MICROARCH.md (6.29 TB/s measured float4 copy)
L2_PEAK = 34500.0      # GB/s, aggregate L2 bandwidth of the guide: the peak for kernels whose operands are L2 / Infinity-Cache resident
MFMA_F32_PEAK = 157.3  # TFLOP/s
dev = torch.device("cuda", 0)


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def emit(kernel, shape, ms, bound, work, unit_work, note="", algorithmic=None):
    """`work` is what the kernel EXECUTES (bytes moved / FLOPs issued): `frac` is a hardware fraction.  Tower kernels execute
    fewer FLOPs than the reference's count (layer 1 is factorised): the reference's count goes to
    `algorithmic` and its ratio to the peak is labelled as such."""
    if bound == "hbm":
        ach = work / (ms * 1e-3) / 1e9; peak = HBM_PEAK; unit = "GB/s"
    elif bound == "l2":            # operands come out of L2 / Infinity Cache: never priced against the HBM peak
        ach = work / (ms * 1e-3) / 1e9; peak = L2_PEAK; unit = "GB/s"
    else:
        ach = work / (ms * 1e-3) / 1e12; peak = MFMA_F32_PEAK; unit = "TFLOP/s"
    rec = {"kernel": kernel, "shape": shape, "ms": round(ms, 4), "bound": bound, "achieved": round(ach, 2), "peak": peak,
           "unit": unit, "frac": round(ach / peak, 4), "work": unit_work, "note": note}
    if algorithmic is not None:
        rec["algorithmic_ratio_to_peak"] = round(algorithmic / (ms * 1e-3) / 1e12 / peak, 4)
    print(json.dumps(rec), flush=True)


NEUMF_EXEC, NEUMF_ALG = 20768, 86176          # FLOP per pair: executed by the fused kernels / the reference's count
HYBRID_EXEC, HYBRID_ALG = 2 * 128 * 64 + 64 + 2 * 128 + 128, 49472


def bench_gathers():
    g = torch.Generator(device="cpu").manual_seed(0)
    for rows, D, B, tag in ((6038, 128, 1 << 23, "ML-1M user MLP table (3 MB, L2-resident)"),
                            (3533, 32, 1 << 24, "ML-1M item GMF table (0.45 MB, L2-resident)"),
                            (4_000_000, 128, 1 << 23, "2 GB table (HBM)"), (10_000_000, 32, 1 << 24, "1.28 GB table (HBM)")):
        table = torch.randn(rows, D, device=dev)
        idx = torch.randint(0, rows, (B,), generator=g).to(dev)
        ms = timeit(lambda: ops.gather_rows(table, idx, check_ids=False))
        resident = rows * D * 4 < (64 << 20)        # the table fits the caches: the row READS are L2 / Infinity-Cache traffic
        emit("gather_rows_kernel", "rows=%d D=%d B=%d" % (rows, D, B), ms, "l2" if resident else "hbm", B * (8 + 8 * D), "8 + 8*D bytes per row", tag)
        del table, idx
    U, I, B = 4_000_000, 1_000_000, 1 << 22
    eug, eig = torch.randn(U, 32, device=dev), torch.randn(I, 32, device=dev)
    eum, eim = torch.randn(U, 128, device=dev), torch.randn(I, 128, device=dev)
    u = torch.randint(0, U, (B,), generator=g).to(dev); i = torch.randint(0, I, (B,), generator=g).to(dev)
    ms = timeit(lambda: ops.neumf_gather(eug, eig, eum, eim, u, i, check_ids=False))
    emit("neumf_gather_kernel", "U=%d I=%d B=%d" % (U, I, B), ms, "hbm", B * (16 + 1280 + 128 + 1024),
         "16 B ids + 1280 B rows read + 1152 B written per pair (materialised gmf_prod + mlp_in)", "HBM-resident tables")


def bench_autorec():
    from recommender_amd.autorec.utils.model import AutoRec
    torch.manual_seed(0)
    m = AutoRec(6040, 3706, 500, True).eval().to(dev)
    X = ((torch.rand(3706, 6040, device=dev) < 0.045) * torch.randint(1, 6, (3706, 6040), device=dev)).float()
    ms = timeit(lambda: m.reconstruct(X))
    emit("linear_kernel x2 (AutoRec cfg3)", "X[3706,6040] -> 500 -> 6040, sigmoid + clamp fused", ms, "mfma", 2 * 2 * 3706 * 6040 * 500,
         "44.77 GFLOP per full reconstruction")
    R = X.t().contiguous()
    ms = timeit(lambda: m.recommend_topk(R, 10))
    emit("AutoRec cfg3 end to end (reconstruct_all + seen-mask + top-10)", "R[6040,3706]", ms, "mfma", 2 * 2 * 3706 * 6040 * 500,
         "44.77 GFLOP per full reconstruction", "users/s = %.3g" % (6040 / ms * 1e3))
    m.attach_ratings(R)
    ms = timeit(lambda: m.recommend_topk(R, 10))
    emit("AutoRec cfg3 end to end, rating matrix kept sparse (CSR encoder + dense decoder + seen-mask + top-10)", "R[6040,3706] %.1f %% dense" % (100 * float((R != 0).float().mean())),
         ms, "mfma", 2 * 2 * 3706 * 6040 * 500, "44.77 GFLOP algorithmic (the encoder executes 2*nnz*500)", "users/s = %.3g" % (6040 / ms * 1e3))
    csr = ops.CsrMatrix.from_dense(R, transpose=True)
    wet = ops.transpose(m.encoder[0].weight.detach())
    ms = timeit(lambda: ops.csr_linear(csr, wet, m.encoder[0].bias.detach(), "sigmoid"))
    emit("csr_linear_kernel (AutoRec encoder over observed entries)", "M=3706 K=6040 N=500 nnz=%d segs=%d" % (csr.nnz, csr.n_seg), ms, "l2",
         csr.nnz * (500 * 4 + 8) + 3706 * 500 * 4, "nnz * (2000 B weight row + 8 B entry) + output; weight rows come from L2 / Infinity Cache, not HBM",
         "dense MFMA form of the same layer: see the encoder lines below; equivalent dense rate = %.1f TFLOP/s" % (2 * 3706 * 6040 * 500 / ms / 1e9))
    flat = ops.CsrMatrix.from_dense(R, transpose=True, k_ranges=1)
    ms = timeit(lambda: ops.csr_linear(flat, wet, m.encoder[0].bias.detach(), "sigmoid"))
    emit("csr_linear_kernel, no k ranges (every workgroup walks all of Wt)", "M=3706 K=6040 N=500 nnz=%d segs=%d" % (flat.nnz, flat.n_seg), ms, "l2",
         flat.nnz * (500 * 4 + 8) + 3706 * 500 * 4, "as above", "A/B for the XCD placement of the segment table")
    import recommender_amd.autorec.utils.model as _am
    _am._csr_cache.pop(m, None)
    we, be = m.encoder[0].weight.detach(), m.encoder[0].bias.detach()
    wd, bd = m.decoder[0].weight.detach(), m.decoder[0].bias.detach()
    h = ops.linear(X, we, be, "sigmoid")
    ms = timeit(lambda: ops.linear(X, we, be, "sigmoid"))
    emit("linear_kernel (AutoRec encoder)", "M=3706 N=500 K=6040", ms, "mfma", 2 * 3706 * 6040 * 500, "2*M*N*K")
    ms = timeit(lambda: ops.linear(R, we, be, "sigmoid", x_transposed=True))
    emit("linear_kernel (AutoRec encoder, A read transposed from R[users,items])", "M=3706 N=500 K=6040", ms, "mfma", 2 * 3706 * 6040 * 500, "2*M*N*K")
    ms = timeit(lambda: ops.linear(h, wd, bd, "clamp", 1.0, 5.0))
    emit("linear_kernel (AutoRec decoder)", "M=3706 N=6040 K=500", ms, "mfma", 2 * 3706 * 6040 * 500, "2*M*N*K")
    for tag, fn, flop in (("AutoRec cfg3 x2", lambda: m.reconstruct(X), 2 * 2 * 3706 * 6040 * 500),
                          ("AutoRec encoder", lambda: ops.linear(X, we, be, "sigmoid", precision="bf16x3"), 2 * 3706 * 6040 * 500),
                          ("AutoRec encoder, transposed A", lambda: ops.linear(R, we, be, "sigmoid", x_transposed=True, precision="bf16x3"), 2 * 3706 * 6040 * 500),
                          ("AutoRec decoder", lambda: ops.linear(h, wd, bd, "clamp", 1.0, 5.0, precision="bf16x3"), 2 * 3706 * 6040 * 500)):
        old = ops.set_tower_precision("bf16x3")
        ms = timeit(fn)
        ops.set_tower_precision(old)
        emit("linear_kernel<SPLIT> bf16x3 (%s)" % tag, "cfg3 shapes", ms, "mfma", flop, "2*M*N*K per GEMM", "fp32-equivalent FLOP/s against the fp32 MFMA peak")
    A = torch.randn(4096, 4096, device=dev); W = torch.randn(4096, 4096, device=dev)
    ms = timeit(lambda: ops.linear(A, W, precision="bf16x3"), reps=10)
    emit("linear_kernel<SPLIT> bf16x3 (square)", "M=N=K=4096", ms, "mfma", 2 * 4096 ** 3, "2*M*N*K")
    ms = timeit(lambda: ops.linear(A, W), reps=10)
    emit("linear_kernel (square)", "M=N=K=4096", ms, "mfma", 2 * 4096 ** 3, "2*M*N*K", "guide's untuned LDS-tiled f32-MFMA GEMM: 122 TF")


def bench_topk():
    for rows, N, k in ((6038, 3533, 10), (6040, 3706, 10), (256, 1_000_000, 100), (115056, 100, 10)):
        s = torch.randn(rows, N, device=dev)
        ms = timeit(lambda: ops.topk_rows(s, k))
        emit(("topk_lane_select_kernel" if (N % 4 == 0 and N <= 160) else "topk_lane_kernel") if (rows >= 32768 and N <= 512 and k <= 16) else "topk_scan_kernel(+merge)", "rows=%d N=%d k=%d" % (rows, N, k), ms, "hbm", rows * (4 * N + 8 * k), "4*N + 8*K bytes per row")
        del s


def bench_tower():
    from recommender_amd.helpers import NCF
    torch.manual_seed(0)
    m = NCF(6038, 3533, 32, 3, 0.0, "NeuMF-end").eval().to(dev)
    t = m._tower(dev)
    B = 115056 * 100
    g = torch.Generator().manual_seed(1)
    u = torch.randint(0, 6038, (115056,), generator=g).repeat_interleave(100).to(dev)
    i = torch.randint(0, 3533, (B,), generator=g).to(dev)
    ms = timeit(lambda: t.score_pairs(u, i, check_ids=False))
    emit("tower_pairs_q_kernel<32,true>", "B=%d (115,056 cases x 100)" % B, ms, "mfma", B * NEUMF_EXEC, "20,768 FLOP/pair executed (86,176 algorithmic)",
         "pairs/s = %.3g" % (B / ms * 1e3), B * NEUMF_ALG)
    ur = torch.randint(0, 6038, (B,), generator=g).to(dev)
    ms = timeit(lambda: t.score_pairs(ur, i, check_ids=False))
    emit("tower_pairs_q_kernel<32,true>", "B=%d random users" % B, ms, "mfma", B * NEUMF_EXEC, "20,768 FLOP/pair executed", "pairs/s = %.3g" % (B / ms * 1e3), B * NEUMF_ALG)
    out = torch.empty(6038, 3533, device=dev)
    ms = timeit(lambda: t.score_matrix(None, None, check_ids=False, out=out))
    emit("tower_tile_kernel<32,true> (+projections)", "6038 x 3533", ms, "mfma", 6038 * 3533 * NEUMF_EXEC, "20,768 FLOP/pair executed",
         "pairs/s = %.3g" % (6038 * 3533 / ms * 1e3), 6038 * 3533 * NEUMF_ALG)
    ops.set_tower_precision("bf16x3")
    ts = m._tower(dev)
    ms = timeit(lambda: ts.score_pairs(u, i, check_ids=False))
    emit("tower_pairs_split_q_kernel<32,true> (bf16x3)", "B=%d (115,056 cases x 100)" % B, ms, "mfma", B * NEUMF_EXEC,
         "20,768 fp32-equivalent FLOP/pair executed, against the FP32 MFMA peak", "pairs/s = %.3g" % (B / ms * 1e3), B * NEUMF_ALG)
    ms = timeit(lambda: ts.score_matrix(None, None, check_ids=False, out=out))
    emit("tower_tile_split_kernel<32,true> (bf16x3, +projections)", "6038 x 3533", ms, "mfma", 6038 * 3533 * NEUMF_EXEC,
         "20,768 fp32-equivalent FLOP/pair executed, against the FP32 MFMA peak", "pairs/s = %.3g" % (6038 * 3533 / ms * 1e3), 6038 * 3533 * NEUMF_ALG)
    ops.set_tower_precision("f32")
    from recommender_amd.hybridautorecncf.model import HybridAutoRecNCF
    h = HybridAutoRecNCF(6040, 3706, 64, [128, 64], rating_output=False).eval().to(dev)
    R = (torch.rand(6040, 3706, device=dev) < 0.045).float()
    ms = timeit(lambda: h.score_all(R), reps=10)
    emit("hybrid score_all (encoders + tower_tile_kernel<64,false>)", "6040 x 3706", ms, "mfma", 6040 * 3706 * HYBRID_EXEC + 23.72e9,
         "16,832 FLOP/pair executed + 23.72 GFLOP encoders", "pairs/s = %.3g" % (6040 * 3706 / ms * 1e3), 6040 * 3706 * HYBRID_ALG + 23.72e9)
    ms = timeit(lambda: (h.user_autorec.encode(R), h.item_autorec.encode(R, transposed=True)), reps=10)
    emit("hybrid encoders only (6 linear_kernel launches + split-K reduces)", "R[6040,3706] -> zu[6040,64], zi[3706,64]", ms, "mfma", 23.72e9,
         "23.72 GFLOP", "score_all = this + the head below: nothing else runs per call")
    h.attach_ratings(R)
    ms = timeit(lambda: h.score_all(R), reps=10)
    emit("hybrid score_all, rating matrix kept sparse (CSR first encoder layers)", "6040 x 3706", ms, "mfma", 6040 * 3706 * HYBRID_EXEC + 23.72e9,
         "16,832 FLOP/pair executed + 23.72 GFLOP encoders (algorithmic)", "pairs/s = %.3g" % (6040 * 3706 / ms * 1e3), 6040 * 3706 * HYBRID_ALG + 23.72e9)
    zu = h.user_autorec.encode(R); zi = h.item_autorec.encode(R.t().contiguous())
    ms = timeit(lambda: h.ncf.score_matrix(zu, zi), reps=10)
    emit("tower_tile_kernel<64,false> (+projections)", "6040 x 3706", ms, "mfma", 6040 * 3706 * HYBRID_EXEC, "16,832 FLOP/pair executed", "", 6040 * 3706 * HYBRID_ALG)
    Bp = 1 << 20                                  # Hybrid forward over pair rows (every pair brings its latents): score_rows
    gp = torch.Generator().manual_seed(2)
    zub = zu[torch.randint(0, 6040, (Bp,), generator=gp).to(dev)].contiguous()
    zib = zi[torch.randint(0, 3706, (Bp,), generator=gp).to(dev)].contiguous()
    th = h.ncf._tower(dev)
    ms = timeit(lambda: th.score_rows(zub, zib, zub, zib), reps=10)
    emit("tower_pairs_kernel<64,false> + per-pair layer-1 projections (Hybrid forward, score_rows)", "B=%d pair rows" % Bp, ms, "mfma",
         Bp * HYBRID_ALG, "49,472 FLOP/pair, all executed (nothing to factorise: every pair has its own rows)", "pairs/s = %.3g" % (Bp / ms * 1e3))
    ops.set_tower_precision("bf16x3")
    ms = timeit(lambda: h.ncf.score_matrix(zu, zi), reps=10)
    emit("tower_tile_split_kernel<64,false> (bf16x3, +projections)", "6040 x 3706", ms, "mfma", 6040 * 3706 * HYBRID_EXEC,
         "16,832 fp32-equivalent FLOP/pair executed, against the FP32 MFMA peak", "", 6040 * 3706 * HYBRID_ALG)
    ops.set_tower_precision("f32")


if __name__ == "__main__":
    which = sys.argv[1:] or ["gather", "autorec", "topk", "tower"]
    for w in which:
        {"gather": bench_gathers, "autorec": bench_autorec, "topk": bench_topk, "tower": bench_tower}[w]()
