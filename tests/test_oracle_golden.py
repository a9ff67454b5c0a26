"""Pin the CPU oracle (oracle/planer_np.py) against vectors captured from the
reference itself (tools/capture_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import planer_np as onp
from planer_amd.irgen import customnet, resnet18, yolov3, blob_sha256
from tests.cases import layer_cases, sample_index
from tests.conftest import assert_close, load_golden

CASES = layer_cases()


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_layer_matches_reference(case, golden_layers):
    name, kind, args, params = case
    z, meta = golden_layers
    for i, a in enumerate(args):                       # seeded inputs did not drift
        np.testing.assert_array_equal(a, z["%s/in%d" % (name, i)])
    ins = [a.copy() for a in args]
    with np.errstate(all="ignore"):
        out = onp.OPS[kind](*ins, **params)
    outs = tuple(out) if isinstance(out, (tuple, list)) else (out,)
    assert len(outs) == meta[name]["n_out"]
    for i, o in enumerate(outs):
        assert_close(o, z["%s/out%d" % (name, i)], 2e-6, name)
    assert bool(outs[0] is ins[0]) == meta[name]["inplace"]


def _net(graph, blob):
    net = onp.OracleNet()
    net.load_json(graph["input"], graph["inits"], graph["layers"], graph["flow"])
    net.load_weights(blob)
    return net


def _check_packed(outs, z, tol):
    for i, o in enumerate(outs):
        o = np.ascontiguousarray(o)
        assert tuple(z["out%d_shape" % i]) == o.shape
        idx = sample_index(o.size)
        scale = float(z["out%d_absmax" % i])
        err = np.abs(o.reshape(-1)[idx] - z["out%d_sample" % i]).max() / scale
        assert err <= tol, err
        s = z["out%d_sum" % i]
        assert abs(o.astype(np.float64).sum() - s[0]) <= tol * s[1]


def test_customnet_matches_reference():
    g, b = customnet.build()
    z = load_golden("customnet_b1.npz")
    assert blob_sha256(b) == str(z["sha"])
    y = _net(g, b)(customnet.make_input(1))
    assert y.shape == (1, 128, 64, 64)
    _check_packed([y], z, 2e-6)
    # without the trailing `return` a batch-1 result loses its batch dim (net.py:101)
    g2 = dict(g, layers=g["layers"][:-1], flow=g["flow"][:-1])
    y2 = _net(g2, b)(customnet.make_input(1))
    assert y2.shape == (128, 64, 64)
    _check_packed([y2], load_golden("customnet_b1_noreturn.npz"), 2e-6)


@pytest.mark.parametrize("n", [1, 2])
def test_resnet18_logits_match_reference(n):
    g, b = resnet18.build()
    z = load_golden("resnet18_b%d.npz" % n)
    assert b.size == resnet18.BLOB_BYTES and blob_sha256(b) == str(z["sha"])
    y = _net(g, b)(resnet18.make_input(n))
    assert y.shape == (n, 1000)
    assert_close(y, z["logits"], 5e-6)


def test_yolov3_160_matches_reference():
    g, b = yolov3.build()
    assert blob_sha256(b) == str(load_golden("yolov3_b1.npz")["sha"])
    y = _net(g, b)(yolov3.make_input(1, size=160))
    assert [o.shape for o in y] == [(1, 255, 5, 5), (1, 255, 10, 10), (1, 255, 20, 20)]
    _check_packed(list(y), load_golden("yolov3_b1_160.npz"), 5e-6)


def test_asymmetric_pads_rejected():
    x = np.zeros((1, 1, 5, 5), np.float32)
    with pytest.raises(ValueError):
        onp.conv2d(x, np.zeros((1, 1, 3, 3), np.float32), pads=(1, 1, 0, 0))
