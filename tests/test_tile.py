"""Tiled large-image inference: the oracle's restatement of util.tile against
vectors produced by the reference (CPU), and planer_amd.util.tile -- resampling, window cuts and
blending as HIP kernels -- against the same vectors on a real MI355X."""
import os

import numpy as np
import pytest

from oracle import planer_np as onp
from tests.cases import tile_cases
from tests.conftest import RTOL, ROOT, assert_close

CASES = tile_cases()


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "tile.npz"))


def conv_f_np(K, B, up):
    def f(win):
        x = win[None, None] if win.ndim == 2 else win.transpose(2, 0, 1)[None]
        y = onp.relu(onp.conv2d(np.ascontiguousarray(x), K, B, pads=[1, 1, 1, 1]))
        if up > 1:
            y = onp.upsample(y, np.array([1, 1, up, up], np.float32), "nearest")
        return np.ascontiguousarray(y[0].transpose(1, 2, 0))
    return f


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_tile_matches_reference(case, golden):
    name, img, K, B, up, kw = case
    np.testing.assert_array_equal(img, golden[name + "/img"])          # seeded inputs did not drift
    out = onp.tile(conv_f_np(K, B, up), img.copy(), **kw)
    assert out.shape == golden[name + "/out"].shape
    assert_close(out, golden[name + "/out"], 2e-6, name)


def test_grid_geometry_matches_reference_formula():
    from planer_amd import util
    for H, W, h, w, mar in [(150, 170, 64, 64, 6), (135, 165, 64, 64, 8), (1000, 64, 64, 64, 0), (65, 65, 64, 64, 12)]:
        assert util.grid_slice(H, W, h, w, mar) == onp.grid_slice(H, W, h, w, mar)
        rows = util.make_slice(H, h, mar)
        assert rows[0].start == 0 and rows[-1].stop == H and all(r.stop - r.start == h for r in rows)
        assert all(a.stop - b.start >= mar for a, b in zip(rows, rows[1:]))      # neighbours overlap by >= margin


# ---- GPU ---------------------------------------------------------------------------------------
def conv_f_dev(pa, K, B, up, batched):
    dK, dB = pa.asarray(K), pa.asarray(B)
    scale = np.array([1, 1, up, up], np.float32)

    def f(win):
        if batched:                                           # (n, h, w[, c]) -> (n, c, h, w)
            x = pa.Unsqueeze(win, [1]) if win.ndim == 3 else pa.Transpose(win, [0, 3, 1, 2])
        else:
            x = pa.Unsqueeze(win, [0, 1]) if win.ndim == 2 else pa.Unsqueeze(pa.Transpose(win, [2, 0, 1]), [0])
        y = pa.ReLU(pa.Conv2d(x, dK, dB, pads=[1, 1, 1, 1]))
        if up > 1:
            y = pa.UpSample(y, scale, "nearest")
        return pa.Transpose(y, [0, 2, 3, 1]) if batched else pa.Transpose(y[0], [1, 2, 0])
    return f


@pytest.mark.gpu
@pytest.mark.parametrize("batched", [False, True], ids=["per_window", "batched"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_device_tile_matches_reference(case, batched, golden):
    import planer_amd as pa
    from planer_amd import util
    name, img, K, B, up, kw = case
    seen = []
    wrapped = util.tile(progress=lambda i, n: seen.append((i, n)), batched=batched, **kw)(conv_f_dev(pa, K, B, up, batched))
    out = wrapped(img.copy())                                 # host image in -> host image out
    ref = golden[name + "/out"]
    assert isinstance(out, np.ndarray) and out.shape == ref.shape and out.dtype == ref.dtype
    assert_close(out, ref, RTOL, name)
    dev_out = wrapped(pa.asarray(img))                        # device image in -> device image out
    assert isinstance(dev_out, pa.DeviceArray)
    np.testing.assert_array_equal(dev_out.get(), out)
    if seen:                                                  # progress callback as in the reference
        n = seen[0][1]
        assert [s[0] for s in seen[:n]] == list(range(1, n + 1))


@pytest.mark.gpu
def test_device_resize_matches_oracle():
    import planer_amd as pa
    from planer_amd import util
    rng = np.random.default_rng(3)
    for shape, size in [((40, 50), (48, 64)), ((33, 21, 3), (80, 17)), ((64, 64, 2), (64, 100)), ((9, 7), (4, 3))]:
        img = rng.standard_normal(shape).astype(np.float32)
        got = util.resize(pa.asarray(img), size).get()
        want = onp.image_resize(img, size)
        assert got.shape == want.shape
        assert_close(got, want, 2e-6, "resize %s -> %s" % (shape, size))
