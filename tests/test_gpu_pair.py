"""Two sibling convolutions in one launch (q4.ConvQ4Pair, csrc/conv_q4_kernel.h conv_q4_pair_kernel) on a real MI355X:
each output against the oracle (util.conv_for, util.py:17-44, + folded BatchNorm / ReLU) to 1e-4 of max|ref| and against
the same conv launched alone; ResNet-18 with and without pairing."""
import numpy as np
import pytest

from oracle import planer_np as onp
from tests.conftest import RTOL, assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import planer_amd
    planer_amd.hip.context()
    return planer_amd


CASES = [(32, 64, 56, 56, 128, 2), (32, 128, 28, 28, 256, 2), (32, 256, 14, 14, 512, 2), (2, 8, 9, 11, 12, 2), (1, 6, 7, 7, 5, 3),
         (3, 16, 16, 16, 40, 2)]


@pytest.mark.parametrize("case", CASES, ids=["x".join(map(str, c)) for c in CASES])
def test_pair_equals_the_two_convs(pa, case):
    from planer_amd import q4
    n, cin, h, w, cout, stride = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    k1 = (rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).astype(np.float32)
    k2 = (rng.standard_normal((cout + 4, cin, 1, 1)) * (2.0 / cin) ** 0.5).astype(np.float32)
    s1, s2 = [rng.uniform(0.5, 1.5, (1, c, 1, 1)).astype(np.float32) for c in (cout, cout + 4)]
    t1, t2 = [(rng.standard_normal((1, c, 1, 1)) * 0.1).astype(np.float32) for c in (cout, cout + 4)]
    b2 = rng.standard_normal(cout + 4).astype(np.float32)
    xq = q4.to_q4(pa.asarray(x))
    K1, K2 = q4.prepare_q4_weights(pa.asarray(k1)), q4.prepare_q4_weights(pa.asarray(k2))
    d = {name: pa.asarray(v) for name, v in dict(s1=s1, s2=s2, t1=t1, t2=t2, b2=b2).items()}
    p1 = dict(strides=[stride, stride], pads=[1, 1, 1, 1], act=1, alpha=0.0)
    p2 = dict(strides=[stride, stride], pads=[0, 0, 0, 0], act=0, alpha=0.0)
    y1, y2 = q4.ConvQ4Pair(xq, K1, None, d["s1"], d["t1"], K2, d["b2"], d["s2"], d["t2"], para1=p1, para2=p2)
    plan = pa.hip.context().last_conv_plan()
    assert plan.startswith("pair[")
    want1 = onp.relu(np.ascontiguousarray(onp.batchnorm(onp.conv2d(x, k1, None, strides=(stride, stride), pads=(1, 1, 1, 1)), s1, t1)))
    want2 = onp.batchnorm(onp.conv2d(x, k2, b2, strides=(stride, stride), pads=(0, 0, 0, 0)), s2, t2)
    assert_close(q4.from_q4(y1).get(), want1, RTOL, "3x3 of %s [%s]" % (case, plan))
    assert_close(q4.from_q4(y2).get(), want2, RTOL, "1x1 of %s [%s]" % (case, plan))
    alone1 = q4.ConvQ4(xq, K1, None, d["s1"], d["t1"], None, **p1)
    alone2 = q4.ConvQ4(xq, K2, d["b2"], d["s2"], d["t2"], None, **p2)
    assert_close(y1.get(), alone1.get(), 2e-6, "pair vs lone 3x3")
    assert_close(y2.get(), alone2.get(), 2e-6, "pair vs lone 1x1")


def test_resnet18_with_and_without_pairing(pa, monkeypatch):
    import planer_amd
    from planer_amd.irgen import resnet18
    g, b = resnet18.build()
    x = resnet18.make_input(8)
    ref = onp.OracleNet()
    ref.load_json(g["input"], g["inits"], g["layers"], g["flow"])
    ref.load_weights(b)
    want = ref(x.copy())
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PLANER_HIP_PAIR", mode)
        net = planer_amd.from_graph(g, b)
        net.streams = "1x1"
        outs[mode] = net(planer_amd.asarray(x.copy())).get()
        assert net.conv_pairs == (3 if mode == "1" else 0)
        assert_close(outs[mode], want, RTOL, "PLANER_HIP_PAIR=%s" % mode)
    assert_close(outs["1"], outs["0"], 1e-5, "paired vs unpaired logits")
