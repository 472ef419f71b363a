#!/usr/bin/env python3
"""The other roofline targets of BASELINE.json's north_star, measured by bench.py after its headline regions:

  gather         K-a  fused 4-table NeuMF gather out of HBM-resident tables                      bound: HBM
  cfg1_protocol  K-b' 115,056 x 100 pair-list forward + segmented top-10 (the 99-negative protocol) bound: fp32 MFMA
  cfg3_autorec   K-c  AutoRec full reconstruction + seen mask + top-10 (dense and CSR encoder)    bound: fp32 MFMA
  cfg4_hybrid    K-b  Hybrid score_all: encoders once + fused head over 6040 x 3706 + top-10      bound: fp32 MFMA

Every section runs W warm-up steps and K timed steps between device synchronisations; `ms_per_step` is wall time, and every
kernel launch inside the timed steps is bracketed by the library's own hipEvent hook on the launch stream
(reco_profile_enable / reco_profile_collect_kinds), so `kernel_ms` is the KERNEL's duration averaged over `launches_timed`
launches, not the call's.  `frac` = achieved / peak with the work one launch executes (stated per kernel in `work`).
Reference call sites: src/helpers/ncf_model.py:207-227 (gathers + tower), src/autorec/utils/model.py:31-33,
src/autorec/utils/helper.py:275-340, src/hybridautorecncf/evaluate_hybridautorecncf_rating.py:167-208.

    python tools/bench_kernels.py            # stand-alone: one JSON object with the four sections
"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from recommender_amd import _lib, ops

WARM_SECONDS = 0.08            # load before a section's timed steps (see timed())
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md
MFMA_F32_PEAK = 157.3          # TFLOP/s dense fp32 MFMA
NEUMF_EXEC, NEUMF_ALG = 20768, 86176
HYBRID_EXEC, HYBRID_ALG = 2 * 128 * 64 + 64 + 2 * 128 + 128, 49472
KIND_NAMES = {_lib.PROF_TOWER: "tower", _lib.PROF_GATHER: "gather", _lib.PROF_LINEAR: "linear", _lib.PROF_LINEAR_REDUCE: "linear_reduce",
              _lib.PROF_CSR: "csr", _lib.PROF_CSR_REDUCE: "csr_reduce", _lib.PROF_TOPK: "topk", _lib.PROF_TOPK_MERGE: "topk_merge",
              _lib.PROF_ENCODER: "encoder"}


def timed(fn, steps, warmup, dev, launches_per_step_cap=24):
    """(wall ms per step, [(kind, mean kernel ms, launches)] by position within a step)."""
    L = _lib.lib()
    # warm-up by TIME as well as by count: the sections start after host-side set-up during which the GPU idles, and the device
    # clock needs ~30 ms of load to settle again (the same effect bench.py's --precondition removes from the headline region);
    # one rank only runs these sections, so a data-dependent step count cannot desynchronise a collective
    t_warm, n_warm = time.perf_counter(), 0
    while n_warm < warmup or time.perf_counter() - t_warm < WARM_SECONDS:
        fn()
        n_warm += 1
        if n_warm % 8 == 0:
            torch.cuda.synchronize(dev)
    cap = steps * launches_per_step_cap
    # wall time per step WITHOUT the hook (an event pair costs ~4 us of stream time per launch: 4 % of a 0.4-ms step with four
    # kernels), then the same steps once more with it for the per-kernel durations
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    _lib.check(L.reco_profile_enable(cap))
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    ms = (ctypes.c_float * cap)()
    kinds = (ctypes.c_int32 * cap)()
    n = ctypes.c_int(0)
    _lib.check(L.reco_profile_collect_kinds(ms, kinds, cap, ctypes.byref(n)))
    _lib.check(L.reco_profile_enable(0))
    per = n.value // steps if steps else 0
    seq = []
    if per and per * steps == n.value:
        for i in range(per):
            seq.append((int(kinds[i]), float(np.mean([ms[i + s * per] for s in range(steps)])), steps))
    return el / steps * 1e3, seq


def kernel_rec(name, kind_ms, bound, work, work_text):
    """One kernel's roofline record; kind_ms = (kind, kernel_ms, launches)."""
    kind, kms, launches = kind_ms
    if bound == "hbm":
        ach, peak, unit = work / (kms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
    else:
        ach, peak, unit = work / (kms * 1e-3) / 1e12, MFMA_F32_PEAK, "TFLOP/s"
    return {"kernel": name, "kind": KIND_NAMES.get(kind, str(kind)), "kernel_ms": kms, "launches_timed": launches, "bound": bound,
            "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "work": work_text}


def _pick(seq, kind, nth=0):
    hits = [x for x in seq if x[0] == kind]
    return hits[nth] if len(hits) > nth else None


def section_gather(dev, steps, warmup):
    U, I, B = 4_000_000, 1_000_000, 1 << 22
    g = torch.Generator(device=dev).manual_seed(11)
    eug, eig = torch.randn(U, 32, device=dev, generator=g), torch.randn(I, 32, device=dev, generator=g)
    eum, eim = torch.randn(U, 128, device=dev, generator=g), torch.randn(I, 128, device=dev, generator=g)
    u = torch.randint(0, U, (B,), device=dev, generator=g); i = torch.randint(0, I, (B,), device=dev, generator=g)
    gp = torch.empty((B, 32), dtype=torch.float32, device=dev)
    keep = []

    def step():
        keep[:] = [ops.neumf_gather(eug, eig, eum, eim, u, i, check_ids=False, gmf_out=gp)]

    ms, seq = timed(step, steps, warmup, dev)
    per_pair = 16 + 1280 + 128 + 1024
    rec = kernel_rec("neumf_gather_kernel", _pick(seq, _lib.PROF_GATHER), "hbm", B * per_pair,
                     "%d pairs x (16 B ids + 1,280 B of rows read + 1,152 B written: gmf_prod [B,32] + mlp_in [B,256])" % B)
    out = {"workload": "fused 4-table NeuMF gather, tables 4M x {32,128} + 1M x {32,128} fp32 (3.2 GB, HBM-resident), %d random pairs, "
                       "outputs materialised as the reference does (ncf_model.py:207-224)" % B,
           "ms_per_step": ms, "steps": steps, "bytes_per_pair": per_pair, "pairs_per_s": B / ms * 1e3}
    out.update(rec)
    del eug, eig, eum, eim, u, i, gp, keep
    torch.cuda.empty_cache()
    return out


def section_cfg1(model, dev, steps, warmup):
    cases, width, k = 115056, 100, 10
    B = cases * width
    g = torch.Generator().manual_seed(1)
    users = torch.randint(0, 6038, (cases,), generator=g)
    items = torch.randint(0, 3533, (cases, width), generator=g)
    u = users.repeat_interleave(width).to(dev); i = items.view(-1).to(dev)
    t = model._tower(dev)

    def step():
        s = t.score_pairs(u, i, check_ids=False)
        return ops.topk_rows(s.view(cases, width), k)

    ms, seq = timed(step, steps, warmup, dev)
    pair = kernel_rec("tower_pairs_q_kernel<32,true,12 waves>", _pick(seq, _lib.PROF_TOWER), "mfma", B * NEUMF_EXEC,
                      "%d pairs x 20,768 FLOP executed (86,176 algorithmic: layer 1 is read from the handle's P/Q projections)" % B)
    lane = kernel_rec("topk_lane_select_kernel<10>", _pick(seq, _lib.PROF_TOPK), "hbm", cases * (4 * width + 8 * k),
                      "%d lists x (400 B of scores read + 80 B written)" % cases)
    # the whole evaluator once, host lists in -> HR / NDCG out, not part of the timed steps
    from recommender_amd.evaluate import evaluate_protocol
    users_h, items_h = users.numpy().astype(np.int32), items.numpy().astype(np.int32)    # as data_io.read_test_negative returns them
    evaluate_protocol(model, users_h, items_h, k)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    hr, nd, _, _ = evaluate_protocol(model, users_h, items_h, k)
    e2e = (time.perf_counter() - t0) * 1e3
    out = {"workload": "cfg1 at the reference's full protocol size: %d test cases x %d candidates = %d pairs, forward(user, item) + per-case "
                       "top-%d (ncf.ipynb cell 8); ids resident in HBM" % (cases, width, B, k),
           "ms_per_step": ms, "steps": steps, "pairs_per_s": B / ms * 1e3,
           "algorithmic": {"flop_per_pair": NEUMF_ALG, "ratio_to_peak": B * NEUMF_ALG / (pair["kernel_ms"] * 1e-3) / 1e12 / MFMA_F32_PEAK,
                           "note": "reference FLOP count over the fp32 MFMA peak; not a hardware fraction"},
           "topk": lane,
           "evaluate_protocol_e2e_ms": e2e, "hr_at_10": hr, "ndcg_at_10": nd,
           "e2e_note": "host int32 id lists (as the protocol reader returns them) in -> HR@10 / NDCG@10 out, one call (PCIe upload + kernels + host metrics); random-init weights"}
    out.update(pair)
    return out


def _ml1m_like_matrix(dev, seed=1):
    """cfg3's X: [6040, 3706] ratings uniform in {1..5}, ~1.0 M non-zeros (4.5 %), item popularity ~ Zipf(1.1)."""
    rng = np.random.default_rng(seed)
    U, I = 6040, 3706
    w = 1.0 / np.arange(1, I + 1) ** 1.1
    target, lo, hi = 1.0e6 / U, 0.0, 1.0e6                 # expected ratings per user; popularity capped at 0.95
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if np.minimum(w * mid, 0.95).sum() < target:
            lo = mid
        else:
            hi = mid
    p = np.minimum(w * hi, 0.95)[rng.permutation(I)]
    mask = rng.random((U, I)) < p[None, :]
    vals = rng.integers(1, 6, (U, I)).astype(np.float32)
    return torch.from_numpy(mask * vals).to(dev)


def section_cfg3(dev, steps, warmup):
    from recommender_amd.autorec.utils.model import AutoRec
    torch.manual_seed(0)
    m = AutoRec(6040, 3706, 500, True).eval()
    for lin in (m.encoder[0], m.decoder[0]):
        torch.nn.init.xavier_uniform_(lin.weight)
    m = m.to(dev)
    R = _ml1m_like_matrix(dev)
    nnz = int((R != 0).sum().item())
    gemm = 2 * 3706 * 6040 * 500
    res = {"workload": "cfg3: I-AutoRec(6040, 3706, 500) full reconstruction R_hat = clamp(dec(sigmoid(enc(X)))) + seen mask + top-10 per "
                       "user; X = synthetic ML-1M-like ratings 1..5, %d non-zeros (%.2f %% dense)" % (nnz, 100.0 * nnz / R.numel()),
           "algorithmic_flop": 2 * gemm, "steps": steps}
    for tag in ("dense", "csr"):
        if tag == "csr":
            m.attach_ratings(R)
        ms, seq = timed(lambda: m.recommend_topk(R, 10), steps, warmup, dev)
        part = {"ms_per_step": ms, "users_per_s": 6040 / ms * 1e3,
                "step_frac_of_mfma_peak_on_algorithmic_flop": 2 * gemm / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK, "kernels": []}
        lin = [x for x in seq if x[0] == _lib.PROF_LINEAR]
        if tag == "dense" and len(lin) >= 2:
            enc = lin[0]
            red = _pick(seq, _lib.PROF_LINEAR_REDUCE)
            enc_ms = enc[1] + (red[1] if red else 0.0)
            part["kernels"].append(kernel_rec("linear_kernel (encoder, A read transposed out of R, split-K + reduce)", (enc[0], enc_ms, enc[2]),
                                              "mfma", gemm, "2 x 3706 x 6040 x 500 FLOP"))
        if tag == "csr":
            c = _pick(seq, _lib.PROF_CSR)
            red = _pick(seq, _lib.PROF_CSR_REDUCE)
            if c:
                c_ms = c[1] + (red[1] if red else 0.0)
                rec = kernel_rec("csr_linear_wave_kernel (+reduce): encoder over the observed entries", (c[0], c_ms, c[2]), "hbm",
                                 nnz * (500 * 4 + 8) + 3706 * 500 * 4, "nnz x (2,000 B weight row + 8 B entry) + output")
                rec["bound"] = "l2"
                rec["peak"] = 34500.0
                rec["frac"] = rec["achieved"] / rec["peak"]
                rec["peak_note"] = "weight rows come out of L2 / Infinity Cache, not HBM: peak = aggregate L2 bandwidth of the guide (~34.5 TB/s)"
                part["kernels"].append(rec)
        if lin:
            dec = lin[-1]
            part["kernels"].append(kernel_rec("linear_kernel (decoder, operands swapped so R_hat lands in [users, items], clamp fused)", dec, "mfma",
                                              gemm, "2 x 6040 x 3706 x 500 FLOP"))
        tk = _pick(seq, _lib.PROF_TOPK)
        if tk:
            part["kernels"].append(kernel_rec("topk_scan_kernel<masked> (seen mask read off the rating matrix)", tk, "hbm",
                                              6040 * (8 * 3706 + 80), "6040 rows x (3706 scores + 3706 ratings read, 80 B written)"))
        res[tag] = part
    d = res["csr"]["kernels"]
    dec = [k for k in d if k["kernel"].startswith("linear_kernel (decoder")]
    if dec:
        res.update({k: dec[0][k] for k in ("kernel", "kernel_ms", "launches_timed", "bound", "achieved", "peak", "unit", "frac")})
    res["ms_per_step"] = res["csr"]["ms_per_step"]
    return res


def section_cfg4(dev, steps, warmup):
    from recommender_amd.hybridautorecncf.model import HybridAutoRecNCF
    torch.manual_seed(0)
    h = HybridAutoRecNCF(6040, 3706, 64, [128, 64], rating_output=False).eval().to(dev)
    R = _ml1m_like_matrix(dev)
    pairs = 6040 * 3706
    enc_flop = 2 * (6040 * (3706 * 256 + 256 * 128 + 128 * 64) + 3706 * (6040 * 256 + 256 * 128 + 128 * 64))
    res = {"workload": "cfg4: Hybrid AutoRec->NCF score_all: both encoders once (6040 users, 3706 items -> 64-d latents), fused head over "
                       "all %d pairs, top-10 per user; same X as cfg3" % pairs,
           "algorithmic_flop": pairs * HYBRID_ALG + enc_flop, "steps": steps}
    for tag in ("dense", "csr"):
        if tag == "csr":
            h.attach_ratings(R)
        # the step as the product runs it (item encoder on a side stream beside the user encoder), then once more with both sides
        # on one stream for the per-kernel records: durations of kernels that overlap on the GPU do not add up to anything
        ms, _ = timed(lambda: h.score_all(R, k=10), steps, warmup, dev, 40)
        was = os.environ.get("RECO_HYBRID_STREAMS")
        os.environ["RECO_HYBRID_STREAMS"] = "0"
        try:
            ms_one, seq = timed(lambda: h.score_all(R, k=10), steps, warmup, dev, 40)
        finally:
            if was is None:
                os.environ.pop("RECO_HYBRID_STREAMS", None)
            else:
                os.environ["RECO_HYBRID_STREAMS"] = was
        part = {"ms_per_step": ms, "ms_per_step_one_stream": ms_one, "pairs_per_s": pairs / ms * 1e3, "kernels": [],
                "kernels_note": "per-kernel durations from the one-stream run (RECO_HYBRID_STREAMS=0); ms_per_step is the two-stream step"}
        tw = _pick(seq, _lib.PROF_TOWER)
        first_tower = next((n for n, x in enumerate(seq) if x[0] == _lib.PROF_TOWER), len(seq))
        enc_ms = sum(x[1] for x in seq[:max(first_tower - 2, 0)])           # everything before the head's two projection GEMMs
        if enc_ms > 0:
            part["kernels"].append(kernel_rec("encoders: %d launches (%s)" % (max(first_tower - 2, 0), "+".join(KIND_NAMES.get(x[0], "?") for x in seq[:max(first_tower - 2, 0)])),
                                              (0, enc_ms, steps), "mfma", enc_flop, "23.7 GFLOP algorithmic (dense first layers)"))
        if tw:
            part["kernels"].append(kernel_rec("tower_tile_kernel<64,false>", tw, "mfma", pairs * HYBRID_EXEC,
                                              "%d pairs x 16,832 FLOP executed (49,472 algorithmic)" % pairs))
        tk = _pick(seq, _lib.PROF_TOPK)
        if tk:
            part["kernels"].append(kernel_rec("topk_scan_kernel", tk, "hbm", 6040 * (4 * 3706 + 80), "6040 rows x 3706 scores read"))
        res[tag] = part
    tw = [k for k in res["csr"]["kernels"] if k["kernel"].startswith("tower_tile")]
    if tw:
        res.update({k: tw[0][k] for k in ("kernel", "kernel_ms", "launches_timed", "bound", "achieved", "peak", "unit", "frac")})
    res["ms_per_step"] = res["csr"]["ms_per_step"]
    # Hybrid forward over pair rows (the reference's own signature, evaluate_hybridautorecncf_rating.py:197-208): score_rows
    Bp = 1 << 20
    gp = torch.Generator().manual_seed(2)
    zu = h.user_autorec.encode(R); zi = h.item_autorec.encode(R.t().contiguous())
    zub = zu[torch.randint(0, 6040, (Bp,), generator=gp).to(dev)].contiguous()
    zib = zi[torch.randint(0, 3706, (Bp,), generator=gp).to(dev)].contiguous()
    th = h.ncf._tower(dev)
    ms, seq = timed(lambda: th.score_rows(zub, zib, zub, zib), steps, warmup, dev)
    call_ms = sum(x[1] for x in seq)
    rows = {"workload": "Hybrid head over %d pair rows, every pair brings its own latents (nothing to factorise)" % Bp, "ms_per_step": ms,
            "kernels_ms": [(KIND_NAMES.get(x[0], "?"), x[1]) for x in seq], "pairs_per_s": Bp / ms * 1e3}
    rows.update(kernel_rec("score_rows: " + "+".join(KIND_NAMES.get(x[0], "?") for x in seq), (0, call_ms, steps), "mfma", Bp * HYBRID_ALG,
                           "%d pairs x 49,472 FLOP, all executed" % Bp))
    res["score_rows"] = rows
    return res


def run(model, dev, steps=10, warmup=3):
    out = {}
    for name, fn in (("gather", lambda: section_gather(dev, steps, warmup)), ("cfg1_protocol", lambda: section_cfg1(model, dev, steps, warmup)),
                     ("cfg3_autorec", lambda: section_cfg3(dev, steps, warmup)), ("cfg4_hybrid", lambda: section_cfg4(dev, steps, warmup))):
        try:
            out[name] = fn()
        except Exception as exc:                                  # a failing section must not take the headline line with it
            out[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    from recommender_amd.helpers import NCF
    device = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = NCF(6038, 3533, 32, 3, 0.0, "NeuMF-end").eval()
    with torch.no_grad():
        for pn, p in net.named_parameters():
            if "embed" in pn:
                p.mul_(25.0)
    print(json.dumps(run(net.to(device), device)))
