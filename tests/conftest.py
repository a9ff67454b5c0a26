import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# north_star tolerance: 1e-4 relative, fp32.  Per tensor:
#   max|y - ref| <= RTOL * max|ref|   
RTOL = 1e-4


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def rel_err(y, ref):
    y = np.asarray(y, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert y.shape == ref.shape, (y.shape, ref.shape)
    if ref.size == 0:
        return 0.0
    scale = max(float(np.abs(ref).max()), 1e-30)
    d = np.abs(y - ref)
    d = np.where(np.isnan(y) & np.isnan(ref), 0.0, d)   # NaN matches NaN
    return float(d.max() / scale)


def assert_close(y, ref, tol=RTOL, what=""):
    e = rel_err(y, ref)
    assert e <= tol, "%s rel err %.3e > %.1e" % (what, e, tol)


@pytest.fixture(scope="session")
def golden_layers():
    z = np.load(os.path.join(GOLDEN, "layers.npz"))
    meta = json.loads(bytes(z["__meta__"]).decode())
    return z, meta


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(autouse=True)
def _default_plan_environment(monkeypatch):
    """Tests describe the DEFAULT plan compiler: experiment switches inherited from the caller's shell are dropped
    (a test that wants one sets it itself)."""
    for k in ("PLANER_HIP_GRAPH", "PLANER_HIP_FUSE", "PLANER_HIP_Q4", "PLANER_HIP_WINOGRAD", "PLANER_HIP_WINOGRAD4",
              "PLANER_HIP_ROWPACK", "PLANER_HIP_TAPMAJOR", "PLANER_HIP_STREAMS", "PLANER_HIP_CONV_ALGO",
              "PLANER_HIP_WINO_CHAIN", "PLANER_HIP_WINO_LDS", "PLANER_HIP_WINO_G", "PLANER_HIP_WINO_BD", "PLANER_HIP_FEED_PACK", "PLANER_HIP_SMALLCIN_WIDE", "PLANER_HIP_WINO_GEMM_AS"):
        monkeypatch.delenv(k, raising=False)
