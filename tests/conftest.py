import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def golden_sd(npz, prefix):
    """state_dict (reference key names) stored under '<prefix>__a__b' keys."""
    out = {}
    for k in npz.files:
        if k.startswith(prefix + "__") and (k.endswith("weight") or k.endswith("bias")):
            out[k[len(prefix) + 2:].replace("__", ".")] = npz[k]
    return out


def assert_close(a, b, rtol=1e-5, atol=1e-6, what=""):
    """SURVEY 8(d) parity gate: |a-b| <= rtol*max(|a|,|b|) + atol."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    tol = rtol * np.maximum(np.abs(a), np.abs(b)) + atol
    bad = err > tol
    if bad.any():
        i = np.unravel_index(np.argmax(err - tol), a.shape)
        raise AssertionError("%s: %d/%d outside tolerance; worst at %s: %r vs %r (err %.3g tol %.3g)" % (
            what, bad.sum(), a.size, i, a[i], b[i], err[i], tol[i]))


@pytest.fixture(scope="session")
def g1():
    return load_golden("g1_gmf_real.npz")


@pytest.fixture(scope="session")
def g2():
    return load_golden("g2_neumf_small.npz")


@pytest.fixture(scope="session")
def g2r():
    return load_golden("g2r_rating.npz")


@pytest.fixture(scope="session")
def g3():
    return load_golden("g3_autorec.npz")


@pytest.fixture(scope="session")
def g4():
    return load_golden("g4_hybrid.npz")


@pytest.fixture(scope="session")
def g5():
    return load_golden("g5_metrics.npz")


@pytest.fixture(scope="session")
def g2b():
    return load_golden("g2b_other_shapes.npz")


@pytest.fixture(scope="session")
def g6():
    return load_golden("g6_files.npz")


@pytest.fixture(params=["f32", "bf16x3"])
def tower_precision(request):
    """Runs a tower test once per multiply mode of the MFMA layers (include/reco_hip.h RECO_PREC_*), same gates for both."""
    from recommender_amd import ops
    old = ops.set_tower_precision(request.param)
    yield request.param
    ops.set_tower_precision(old)
