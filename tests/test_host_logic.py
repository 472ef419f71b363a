"""CPU-side tests: C ABI surface, module mirrors (state_dict layouts, pickling, init order), loaders, drop-in shim."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, golden_sd


def test_library_exports_every_declared_symbol():
    from recommender_amd import _lib
    header = open(os.path.join(ROOT, "include", "reco_hip.h")).read()
    declared = set(re.findall(r"\b(reco_[a-z0-9_]+)\s*\(", header))
    declared -= {"reco_tower_desc"}
    assert declared, "no declarations parsed"
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "libreco_hip.so does not export %s" % name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert L.reco_abi_version() == _lib.ABI_VERSION == 5


def test_tower_desc_layout_matches_header():
    import ctypes
    from recommender_amd._lib import TowerDesc
    # 4 pointers, 2 int64, 2+1+3 int32, 3+3 pointers, 2 pointers, int32 + 2 floats (natural alignment, LP64)
    assert ctypes.sizeof(TowerDesc) == 4 * 8 + 2 * 8 + 6 * 4 + 6 * 8 + 2 * 8 + 4 + 2 * 4 + 4
    assert TowerDesc.W.offset == 4 * 8 + 2 * 8 + 6 * 4


def test_c_abi_argument_errors_without_gpu():
    from recommender_amd import _lib
    L = _lib.lib()
    assert L.reco_topk_rows(None, 4, 10, 10, 0, None, 0, None, None, None, 0, None) == _lib.EINVAL
    assert b"k must be" in L.reco_last_error() or b"null" in L.reco_last_error()
    assert L.reco_gather_rows(None, 10, 30, None, 5, None, None, None) == _lib.EINVAL
    assert L.reco_tower_score_pairs(None, None, None, 3, None, None, None) == _lib.EINVAL
    assert L.reco_tower_topk_workspace_bytes(6038, 3533, 10) > 6038 * 3533 * 4
    with pytest.raises(ValueError):
        _lib.check(_lib.EINVAL)


def test_ncf_state_dict_layouts(g2):
    from recommender_amd.helpers import NCF
    end = NCF(257, 131, 32, 3, 0.1, "NeuMF-end")
    assert list(end.state_dict().keys()) == list(g2["key_list_end"])
    gmf = NCF(97, 61, 32, 3, 0.1, "GMF"); mlp = NCF(97, 61, 32, 3, 0.1, "MLP")
    pre = NCF(97, 61, 32, 3, 0.1, "NeuMF-pre", gmf, mlp)
    assert list(pre.state_dict().keys()) == list(g2["key_list_pre"])
    assert {k: tuple(v.shape) for k, v in end.state_dict().items()}["MLP_layers.4.weight"] == (64, 128)
    # NeuMF-pre initialisation from the pair (ncf_model.py:158-186), checked on the fixture built by the reference
    g = NCF(97, 61, 32, 3, 0.1, "GMF"); g.load_state_dict({k: torch.from_numpy(v) for k, v in golden_sd(g2, "gmf").items()})
    m = NCF(97, 61, 32, 3, 0.1, "MLP"); m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_sd(g2, "mlp").items()})
    p = NCF(97, 61, 32, 3, 0.1, "NeuMF-pre", g, m)
    ref = golden_sd(g2, "pre")
    for k, v in p.state_dict().items():
        assert np.array_equal(v.numpy(), ref[k]), k
    with pytest.raises(ValueError):
        NCF(5, 5, 32, 3, 0.1, "bogus")


def test_ncf_init_consumes_rng_like_the_reference(g2):
    """Same constructor order and initialisers as ncf_model.py:61-186 => same weights from the same seed.

    The fixture model 'end' was built by the reference under torch.manual_seed(0) and then rescaled
    (embeddings x25, random biases): undo that and compare bit for bit.
    """
    from recommender_amd.helpers import NCF
    torch.manual_seed(0)
    mine = NCF(257, 131, 32, 3, 0.1, "NeuMF-end").state_dict()
    ref = golden_sd(g2, "end")
    for k in ("embed_user_GMF.weight", "embed_item_MLP.weight"):
        assert np.array_equal((mine[k] * 25.0).numpy(), ref[k]), k
    for k in ("MLP_layers.1.weight", "MLP_layers.7.weight", "predict_layer.weight"):
        assert np.array_equal(mine[k].numpy(), ref[k]), k


def test_forward_refuses_to_run_without_gpu_or_in_train_mode():
    from recommender_amd.helpers import NCF
    m = NCF(10, 7, 32, 3, 0.5, "NeuMF-end")
    with pytest.raises(RuntimeError, match="eval"):
        m(torch.tensor([1]), torch.tensor([2]))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m.eval()(torch.tensor([1]), torch.tensor([2]))


def test_loaders_real_gmf_whole_module_pickle(g1, tmp_path):
    """models/GMF.pth is a whole-module pickle naming helpers.ncf_model.NCF: weights_only load, our class."""
    from recommender_amd.helpers import NCF
    from recommender_amd.loaders import load_ncf_model, ncf_from_state_dict
    m = load_ncf_model(os.path.join(GOLDEN, "GMF.pth"))
    assert isinstance(m, NCF) and m.model_name == "GMF" and not m.training
    assert np.array_equal(m.embed_user_GMF.weight.detach().numpy(), g1["embed_user_GMF__weight"])
    assert tuple(m.embed_item_GMF.weight.shape) == (3533, 32)
    assert load_ncf_model(str(tmp_path / "missing.pth")) is None
    # shape (2): plain state_dict, with and without the rating epilogue; nested keys of NeuMF-pre ignored
    sd = {k: v for k, v in m.state_dict().items()}
    torch.save(sd, tmp_path / "GMF-Rating.pth")
    from recommender_amd.ncf.rating_model import NCFRating
    r = load_ncf_model(str(tmp_path / "GMF-Rating.pth"))
    assert isinstance(r, NCFRating) and r._rating_epilogue() == (1.0, 5.0)
    bad = tmp_path / "bad.pth"; bad.write_bytes(b"not a checkpoint")
    assert load_ncf_model(str(bad)) is None


def test_loaders_neumf_pre_state_dict_and_autorec_checkpoint(g2, g3, tmp_path):
    from recommender_amd.loaders import load_autorec_model, ncf_from_state_dict
    pre = {k: torch.from_numpy(v) for k, v in golden_sd(g2, "pre").items()}
    assert len(pre) == 26
    m = ncf_from_state_dict(pre)
    assert m.model_name == "NeuMF-end" and len(m.state_dict()) == 12
    sd = {k: torch.from_numpy(v) for k, v in golden_sd(g3, "a").items()}
    torch.save({"epoch": 49, "model_state_dict": sd, "optimizer_state_dict": {}, "val_rmse": 0.874113}, tmp_path / "AutoRec-best.pth")
    a = load_autorec_model(str(tmp_path / "AutoRec-best.pth"), 193, 89, 64)
    assert a is not None and not a.training and tuple(a.encoder[0].weight.shape) == (64, 193)
    assert load_autorec_model(str(tmp_path / "AutoRec-best.pth"), 6040, 3706, 500) is None     # shape mismatch -> None like the reference


def test_dropin_shim_resolves_reference_import_paths(tmp_path):
    """With recommender_amd/dropin first on sys.path, the reference's import lines and its pickles get the HIP classes."""
    code = r'''
import sys, os
sys.path.insert(0, os.path.join(%r, "recommender_amd", "dropin")); sys.path.insert(1, %r)
import torch
from helpers import NCF, hit, ndcg
from helpers.ncf_model import NCF as N2
from autorec.utils.model import AutoRec
import recommender_amd.helpers.ncf_model as real
assert NCF is real.NCF and N2 is real.NCF
from recommender_amd.loaders import _safe_globals
# the reference's shipped checkpoint is only ever read by a weights-only loader: the pickle's class path
# helpers.ncf_model.NCF (what src/api/core/model_loader.py:28 resolves by import) is on the allow-list and maps to the HIP class
with torch.serialization.safe_globals(_safe_globals()):
    m = torch.load(%r, weights_only=True)
assert type(m) is real.NCF and m.model_name == "GMF", type(m)
m2 = NCF(12, 9, 32, 3, 0.0, "NeuMF-end")
torch.save(m2, %r)
with torch.serialization.safe_globals(_safe_globals()):
    m3 = torch.load(%r, weights_only=True)
assert type(m3) is real.NCF and len(m3.state_dict()) == 12
print("ok")
''' % (ROOT, ROOT, os.path.join(GOLDEN, "GMF.pth"), str(tmp_path / "w.pth"), str(tmp_path / "w.pth"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_autorec_and_hybrid_layouts(g3, g4):
    from recommender_amd.autorec.utils.model import AutoRec
    from recommender_amd.hybridautorecncf.model import HybridAutoRecNCF
    a = AutoRec(6040, 3706, 500, True)
    assert {k: tuple(v.shape) for k, v in a.state_dict().items()} == {
        "encoder.0.weight": (500, 6040), "encoder.0.bias": (500,), "decoder.0.weight": (6040, 500), "decoder.0.bias": (6040,)}
    assert tuple(AutoRec(6040, 3706, 500, False).encoder[0].weight.shape) == (500, 3706)
    h = HybridAutoRecNCF(6040, 3706, 64, [128, 64])
    assert sum(p.numel() for p in h.parameters()) == 5190099        # hybridautorecncf.ipynb cell 12 output
    for tag, (U, I), hd, rating in (("rating", (83, 47), (48, 40), True), ("logit", (40, 26), (256, 128), False)):
        m = HybridAutoRecNCF(U, I, 64, [128, 64], hidden_dims=hd, rating_output=rating)
        assert set(m.state_dict().keys()) == set(golden_sd(g4, tag).keys())


def test_ranking_metrics(g5):
    from recommender_amd.helpers import hit, ndcg
    from recommender_amd.helpers.ranking_metrics import hit_multi, hit_ndcg_batch, ndcg_multi
    hr = [hit(int(g), p.tolist()) for g, p in zip(g5["gts"], g5["preds"])]
    nd = [float(ndcg(int(g), p.tolist())) for g, p in zip(g5["gts"], g5["preds"])]
    assert np.array_equal(hr, g5["hr"]) and np.array_equal(nd, g5["ndcg"])
    bh, bn = hit_ndcg_batch(g5["preds"], g5["gts"])
    assert np.array_equal(bh, g5["hr"]) and np.array_equal(bn, g5["ndcg"])
    mg = [[x for x in row if x >= 0] for row in g5["multi_gt"]]
    assert np.array_equal([hit_multi(g, p) for g, p in zip(mg, g5["preds"])], g5["multi_hr"])
    assert np.allclose([ndcg_multi(g, p) for g, p in zip(mg, g5["preds"])], g5["multi_ndcg"], rtol=0, atol=1e-15)


def test_header_is_plain_c_and_a_c_consumer_can_bind_the_library(tmp_path):
    """The boundary is a C ABI: the header compiles as C99, and a C program (no Python, no torch) binds the library the way a
    cgo / JNI / N-API stub would: dlopen + the declared prototypes.  No compute call is made (no GPU here)."""
    import ctypes
    import shutil
    from recommender_amd import _lib
    from recommender_amd._lib import TowerDesc
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    header_dir = os.path.join(ROOT, "include")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                    os.path.join(header_dir, "reco_hip.h")], check=True)
    src = tmp_path / "consumer.c"
    src.write_text(r'''
#include <dlfcn.h>
#include <stdio.h>
#include "reco_hip.h"
int main(int argc, char** argv) {
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "%s\n", dlerror()); return 2; }
    int (*abi)(void) = (int (*)(void))dlsym(h, "reco_abi_version");
    const char* (*err)(void) = (const char* (*)(void))dlsym(h, "reco_last_error");
    size_t (*tk)(int64_t, int64_t, int) = (size_t (*)(int64_t, int64_t, int))dlsym(h, "reco_topk_workspace_bytes");
    size_t (*tw)(int64_t, int64_t, int) = (size_t (*)(int64_t, int64_t, int))dlsym(h, "reco_tower_topk_workspace_bytes");
    int (*pairs)(reco_tower_t*, const int64_t*, const int64_t*, int64_t, float*, void*) =
        (int (*)(reco_tower_t*, const int64_t*, const int64_t*, int64_t, float*, void*))dlsym(h, "reco_tower_score_pairs");
    if (!abi || !err || !tk || !tw || !pairs) return 3;
    int rc = pairs(NULL, NULL, NULL, 4, NULL, NULL);            /* refused with a code and a message, nothing launched */
    printf("%d %zu %zu %zu %zu %d %s\n", abi(), sizeof(reco_tower_desc_t), tk(0, 100, 10), tk(0, 0, 0), tw(6038, 3533, 10), rc, err());
    (void)argc;
    return 0;
}
''')
    exe = tmp_path / "consumer"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", header_dir, str(src), "-o", str(exe), "-ldl"], check=True)
    out = subprocess.run([str(exe), _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout.split(None, 6)
    assert int(out[0]) == _lib.ABI_VERSION
    assert int(out[1]) == ctypes.sizeof(TowerDesc)
    assert int(out[2]) == 0 and int(out[3]) == 0                 # zero rows / zero everything: sized, not divided by
    assert int(out[4]) == _lib.lib().reco_tower_topk_workspace_bytes(6038, 3533, 10)
    assert int(out[5]) == _lib.EINVAL and "null handle" in out[6]


def test_multi_gt_metrics_batch_equals_the_per_user_loops_bit_for_bit():
    """multi_gt_metrics_batch replaces 6,040 Python iterations of hit_multi / ndcg_multi in the AutoRec / Hybrid evaluators:
    same float64 numbers, not merely close ones (users without test items, fewer test items than k, no hit, all hits)."""
    from recommender_amd.helpers.ranking_metrics import hit_multi, multi_gt_metrics_batch, ndcg_multi
    rng = np.random.default_rng(3)
    for users, items, k, density in [(200, 50, 10, 0.1), (64, 30, 10, 0.5), (50, 12, 10, 0.02), (33, 40, 1, 0.2), (10, 128, 100, 0.3)]:
        test = (rng.random((users, items)) < density) * rng.integers(1, 6, (users, items))
        test[0] = 0                                                  # a user without test items
        test[1] = 3                                                  # every item is ground truth
        top = np.stack([rng.permutation(items)[:k] for _ in range(users)]).astype(np.int32)
        hr, nd, has = multi_gt_metrics_batch(top, test)
        for u in range(users):
            gt = np.flatnonzero(test[u] > 0)
            assert has[u] == (gt.size > 0)
            assert hr[u] == hit_multi(gt, top[u])
            assert nd[u] == ndcg_multi(gt, top[u]), (u, nd[u], ndcg_multi(gt, top[u]))
