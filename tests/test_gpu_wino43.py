"""Mixed-tile Winograd -- F(4,3) x F(3,3) segments on maps whose sides are 7, 14 or 21 (csrc/wino43_kernels.h, ConvQ4 w_layout 11,
q4.Wino43In / Gemm / Out / Chain) -- on a real MI355X.

Against the oracle (layer.Conv2d layer.py:22-26 -> util.conv_for util.py:17-44, + BatchNorm / Add / ReLU / LeakyReLU): 3e-5 of
max|ref| like the F(4x4,3x3) path (F(3,3)'s constants are the smaller ones).  Stages against the one-call conv and the chained
kernel against two one-call convs: bit for bit (the same arithmetic in the same order)."""
import numpy as np
import pytest

from tests.conftest import RTOL, assert_close
from tests.test_gpu_wino_chain import TAILS, _act, _operands, _oracle, pa  # noqa: F401  (shared fixtures / helpers)

pytestmark = pytest.mark.gpu

SHAPES = [(32, 256, 14, 14, 256), (32, 512, 7, 7, 512), (2, 8, 21, 14, 12), (3, 12, 7, 21, 8), (1, 4, 7, 7, 4), (5, 20, 14, 7, 24),
          (1, 64, 21, 21, 32)]


def _u43(pa, dev):
    from planer_amd import q4
    return q4.prepare_winograd43_q4_weights(dev["k"])


def _mono43(dev, u, tail, xq=None):
    from planer_amd import q4
    return q4.ConvQ4(dev["xq"] if xq is None else xq, u, dev["b"], dev["scale"], dev["shift"], dev["resq"], pads=(1, 1, 1, 1),
                     act=_act(tail), alpha=0.1, w_layout=11)


@pytest.mark.parametrize("shape", SHAPES, ids=["x".join(map(str, s)) for s in SHAPES])
def test_lds_transforms_equal_the_whole_tile_kernels_bit_for_bit(pa, shape, monkeypatch):
    """PLANER_HIP_WINO43_LDS=0: lone input / output transforms on the whole-tile register kernels; 1 (default): on the LDS kernel
    (one row of a tile per thread).  Same row functions, same order: V, M -> y and the conv agree bit for bit."""
    from planer_amd import q4
    n, cin, h, w, cout = shape
    rng = np.random.default_rng(5 + sum(shape))
    for tail in TAILS[2:4]:
        host, dev = _operands(pa, rng, n, cin, h, w, cout, tail)
        u = _u43(pa, dev)
        got = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("PLANER_HIP_WINO43_LDS", mode)
            v = q4.Wino43In(dev["xq"])
            y = q4.Wino43Out(q4.Wino43Gemm(v, u), dev["b"], dev["scale"], dev["shift"], dev["resq"], act=_act(tail), alpha=0.1)
            got[mode] = (v.get(), y.get())
        np.testing.assert_array_equal(got["0"][0], got["1"][0], err_msg="V %s" % (shape,))
        np.testing.assert_array_equal(got["0"][1], got["1"][1], err_msg="y %s %s" % (shape, tail))


@pytest.mark.parametrize("shape", SHAPES, ids=["x".join(map(str, s)) for s in SHAPES])
def test_mixed_tile_conv_matches_oracle_and_its_stages(pa, shape):
    from planer_amd import q4
    n, cin, h, w, cout = shape
    rng = np.random.default_rng(sum(shape))
    for tail in TAILS:
        host, dev = _operands(pa, rng, n, cin, h, w, cout, tail)
        u = _u43(pa, dev)
        y = _mono43(dev, u, tail)
        assert q4.logical_shape(y) == (n, cout, h, w)
        assert_close(q4.from_q4(y).get(), _oracle(host, tail), 3e-5, "%s %s" % (shape, tail))
        v = q4.Wino43In(dev["xq"])
        m = q4.Wino43Gemm(v, u)
        assert v.meta == (n, cin, h, w) and m.meta == (n, cout, h, w) and v.size == 121 * cin * n * (h // 7) * (w // 7)
        got = q4.Wino43Out(m, dev["b"], dev["scale"], dev["shift"], dev["resq"], act=_act(tail), alpha=0.1)
        np.testing.assert_array_equal(got.get(), y.get(), err_msg="%s %s" % (shape, tail))
        # ... and agrees with the F(4x4,3x3) path on the same operands
        y4 = q4.ConvQ4(dev["xq"], dev["u"], dev["b"], dev["scale"], dev["shift"], dev["resq"], pads=(1, 1, 1, 1), act=_act(tail),
                       alpha=0.1, w_layout=7)
        assert_close(y.get(), y4.get(), 3e-5, "vs F(4x4) %s %s" % (shape, tail))


@pytest.mark.parametrize("shape", SHAPES, ids=["x".join(map(str, s)) for s in SHAPES])
def test_mixed_tile_chain_equals_two_convs_bit_for_bit(pa, shape):
    from planer_amd import q4
    n, cin, h, w, cout = shape
    rng = np.random.default_rng(11 + sum(shape))
    for tail in TAILS:
        host1, dev1 = _operands(pa, rng, n, cin, h, w, cout, tail)
        host2, dev2 = _operands(pa, rng, n, cout, h, w, cout, ("bn", "relu"))
        u1, u2 = _u43(pa, dev1), _u43(pa, dev2)
        y1_want = _mono43(dev1, u1, tail)
        y2_want = _mono43(dev2, u2, ("bn", "relu"), xq=y1_want).get()
        for keep in (True, False):
            m1 = q4.Wino43Gemm(q4.Wino43In(dev1["xq"]), u1)
            out = q4.Wino43Chain(m1, dev1["b"], dev1["scale"], dev1["shift"], dev1["resq"], act=_act(tail), alpha=0.1, keep_y=keep)
            y1, v2 = out if keep else (None, out)
            if keep:
                np.testing.assert_array_equal(y1.get(), y1_want.get(), err_msg="y1 %s %s" % (shape, tail))
            np.testing.assert_array_equal(v2.get(), q4.Wino43In(y1_want).get(), err_msg="V2 %s %s" % (shape, tail))
            y2 = q4.Wino43Out(q4.Wino43Gemm(v2, u2), dev2["b"], dev2["scale"], dev2["shift"], None, act=_act(("relu",)))
            np.testing.assert_array_equal(y2.get(), y2_want, err_msg="y2 %s %s keep=%s" % (shape, tail, keep))
        want = _oracle(host2, ("bn", "relu"), x=_oracle(host1, tail))
        assert_close(q4.from_q4(y2).get(), want, RTOL, "%s %s" % (shape, tail))


def test_mixed_tiles_are_refused_elsewhere(pa):
    from planer_amd import q4
    assert q4.winograd43_eligible((1, 8, 14, 7), (8, 8, 3, 3), strides=(1, 1), pads=(1, 1, 1, 1))
    assert not q4.winograd43_eligible((1, 8, 16, 14), (8, 8, 3, 3), strides=(1, 1), pads=(1, 1, 1, 1))
    assert not q4.winograd43_eligible((1, 8, 14, 14), (8, 8, 3, 3), strides=(2, 2), pads=(1, 1, 1, 1))
    x = q4.to_q4(pa.asarray(np.zeros((1, 8, 16, 16), np.float32)))
    with pytest.raises((NotImplementedError, ValueError)):
        q4.Wino43In(x)
    u = q4.prepare_winograd43_q4_weights(pa.asarray(np.zeros((8, 8, 3, 3), np.float32)))
    with pytest.raises(ValueError):
        q4.ConvQ4(x, u, pads=(1, 1, 1, 1), w_layout=11)
