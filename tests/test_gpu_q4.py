"""Channel-quad (Q4) layout on a real MI355X: layout conversions are bit-exact, the Q4
implicit-GEMM conv (every tile configuration, split-K, fused tail, groups, odd channel counts)
matches the reference vectors / the oracle to 1e-4 of max|ref| (tests/conftest.RTOL)."""
import numpy as np
import pytest

from oracle import planer_np as onp
from tests.cases import layer_cases
from tests.conftest import RTOL, assert_close

pytestmark = pytest.mark.gpu
CASES = layer_cases()
CONV_CASES = [c for c in CASES if c[1] == "conv"]


@pytest.fixture(scope="module")
def pa():
    import planer_amd
    planer_amd.hip.context()
    return planer_amd


def q4_host(x):
    """numpy statement of the layout: (N,C,H,W) -> (N,ceil(C/4),H,W,4), zero padded."""
    n, c, h, w = x.shape
    cq = (c + 3) // 4
    pad = np.zeros((n, cq * 4, h, w), x.dtype)
    pad[:, :c] = x
    return np.ascontiguousarray(pad.reshape(n, cq, 4, h, w).transpose(0, 1, 3, 4, 2))


@pytest.mark.parametrize("shape", [(2, 1, 5, 7), (1, 3, 9, 4), (3, 4, 6, 6), (2, 5, 3, 11), (2, 64, 7, 7), (1, 255, 13, 13)])
def test_layout_round_trip_is_bit_exact(pa, shape):
    from planer_amd import q4
    x = np.random.default_rng(sum(shape)).standard_normal(shape).astype(np.float32)
    xq = q4.to_q4(pa.asarray(x))
    assert xq.chan == shape[1] and xq.shape == q4_host(x).shape
    np.testing.assert_array_equal(xq.get(), q4_host(x))
    np.testing.assert_array_equal(q4.from_q4(xq).get(), x)


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_q4_conv_vs_reference_vectors(pa, case, golden_layers):
    from planer_amd import q4
    name, kind, args, params = case
    z, _ = golden_layers
    ref = z["%s/out0" % name]
    x, k = args[0], args[1]
    b = pa.asarray(args[2]) if len(args) > 2 and args[2] is not None else None
    group = params.get("group", 1)
    if not q4.q4_conv_eligible(k.shape, **params):
        with pytest.raises(NotImplementedError):
            q4.ConvQ4(q4.to_q4(pa.asarray(x)), q4.prepare_q4_weights(pa.asarray(k), group), b, **params)
        return
    yq = q4.ConvQ4(q4.to_q4(pa.asarray(x)), q4.prepare_q4_weights(pa.asarray(k), group), b, **params)
    assert q4.logical_shape(yq) == ref.shape
    assert_close(q4.from_q4(yq).get(), ref, RTOL, name)
    # padding lanes of the last quad stay zero (consumers multiply them by zero weights)
    raw = yq.get()
    np.testing.assert_array_equal(raw, q4_host(q4.from_q4(yq).get()))


def _cfg_names(pa):
    import ctypes
    lib = pa._lib.load()
    names = []
    for c in range(lib.pl_conv2d_num_configs()):
        buf = ctypes.create_string_buffer(32)
        lib.pl_conv2d_config_name(c, buf, 32)
        names.append(buf.value.decode())
    return names


def test_q4_conv_every_tile_config_and_split_k(pa):
    from planer_amd import q4
    ctx = pa.hip.context()
    names = _cfg_names(pa)
    qnames = [n for n in names if n.startswith("q") or n.startswith("k")]      # k32x32x8: K split inside the workgroup
    assert len(qnames) >= 9 and "k32x32x8" in qnames
    rng = np.random.default_rng(11)
    shapes = [((3, 32, 14, 14), (40, 32, 3, 3), dict(strides=[1, 1], pads=[1, 1, 1, 1])),
              ((2, 3, 33, 35), (20, 3, 7, 7), dict(strides=[2, 2], pads=[3, 3, 3, 3])),
              ((2, 64, 7, 7), (130, 64, 1, 1), dict(strides=[1, 1], pads=[0, 0, 0, 0])),
              ((2, 20, 13, 11), (70, 20, 3, 3), dict(strides=[2, 2], pads=[1, 1, 1, 1])),
              ((2, 6, 10, 9), (9, 6, 3, 5), dict(strides=[1, 2], pads=[1, 2, 1, 2])),
              ((2, 64, 9, 9), (48, 32, 3, 3), dict(strides=[1, 1], pads=[2, 2, 2, 2], dilations=[2, 2], group=2))]
    try:
        for xs, ks, p in shapes:
            x = rng.standard_normal(xs).astype(np.float32)
            k = (rng.standard_normal(ks) * 0.1).astype(np.float32)
            b = rng.standard_normal(ks[0]).astype(np.float32)
            ref = np.ascontiguousarray(onp.conv2d(x, k, b, **p))
            xq, db = q4.to_q4(pa.asarray(x)), pa.asarray(b)
            kq = q4.prepare_q4_weights(pa.asarray(k), p.get("group", 1))
            for name in qnames:
                for split in ((1,) if name.startswith("k") else (1, 2, 3)):
                    ctx.set_conv_config(names.index(name), split)
                    y = q4.from_q4(q4.ConvQ4(xq, kq, db, **p)).get()
                    assert_close(y, ref, RTOL, "cfg %s split %d %s" % (name, split, xs))
    finally:
        ctx.set_conv_config(-1, 0)


def test_q4_fused_tail_and_hybrid_plans_are_deterministic(pa):
    from planer_amd import q4
    ctx = pa.hip.context()
    names = _cfg_names(pa)
    rng = np.random.default_rng(23)
    for cout in (128, 126):
        x = rng.standard_normal((8, 64, 28, 28)).astype(np.float32)
        k = (rng.standard_normal((cout, 64, 3, 3)) * 0.05).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, (1, cout, 1, 1)).astype(np.float32)
        sh = rng.standard_normal((1, cout, 1, 1)).astype(np.float32)
        res = rng.standard_normal((8, cout, 28, 28)).astype(np.float32)
        ref = onp.relu(onp.batchnorm(np.ascontiguousarray(onp.conv2d(x, k, pads=[1, 1, 1, 1])), sc, sh) + res)
        xq, kq, rq = q4.to_q4(pa.asarray(x)), q4.prepare_q4_weights(pa.asarray(k)), q4.to_q4(pa.asarray(res))
        dsc, dsh = pa.asarray(sc), pa.asarray(sh)
        try:
            for name, dp, split, occ in [("q64x64x16", 0, 4, 0), ("q64x64x16", 64, 6, 0), ("q128x64x16", 0, 9, 0),
                                         ("q64x64x32", 128, 2, 4), ("q128x128x16", 0, 1, 0), ("q128x32x32", 8, 5, 2)]:
                ctx.set_conv_plan(names.index(name), dp, split, occ)
                first = None
                for it in range(6):
                    yq = q4.ConvQ4(xq, kq, None, dsc, dsh, rq, pads=[1, 1, 1, 1], act=1)
                    y = q4.from_q4(yq).get()
                    if first is None:
                        first = y
                        assert_close(y, ref, RTOL, "%s dp%d s%d" % (name, dp, split))
                        np.testing.assert_array_equal(yq.get(), q4_host(y))
                    else:
                        np.testing.assert_array_equal(y, first)
        finally:
            ctx.set_conv_config(-1, 0)
    # autotuned plan + leaky relu
    y = q4.from_q4(q4.ConvQ4(xq, kq, None, dsc, dsh, None, pads=[1, 1, 1, 1], act=2, alpha=0.1)).get()
    ref = onp.leakyrelu(onp.batchnorm(np.ascontiguousarray(onp.conv2d(x, k, pads=[1, 1, 1, 1])), sc, sh), 0.1)
    assert_close(y, ref, RTOL, "autotuned leaky")


def test_winograd_q4_matches_oracle(pa):
    """Winograd F(2x2,3x3) on Q4 tensors (float4 transforms around one grouped 1x1 Q4 conv), odd and
    even maps, channel counts that are multiples of 4 but not of 16, with and without the fused tail."""
    from planer_amd import q4
    rng = np.random.default_rng(19)
    for (n, cin, h, w, cout) in [(2, 16, 7, 7, 24), (3, 20, 14, 14, 44), (1, 64, 9, 13, 64), (2, 48, 28, 28, 32)]:
        x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
        k = (rng.standard_normal((cout, cin, 3, 3)) * 0.1).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, (1, cout, 1, 1)).astype(np.float32)
        sh = rng.standard_normal((1, cout, 1, 1)).astype(np.float32)
        res = rng.standard_normal((n, cout, h, w)).astype(np.float32)
        xq = q4.to_q4(pa.asarray(x))
        U = q4.prepare_winograd_q4_weights(pa.asarray(k))
        y = q4.from_q4(q4.ConvQ4(xq, U, pa.asarray(b), pads=[1, 1, 1, 1], w_layout=4)).get()
        ref = np.ascontiguousarray(onp.conv2d(x, k, b, pads=[1, 1, 1, 1]))
        assert_close(y, ref, RTOL, "winograd q4 %s" % ((n, cin, h, w, cout),))
        y = q4.from_q4(q4.ConvQ4(xq, U, None, pa.asarray(sc), pa.asarray(sh), q4.to_q4(pa.asarray(res)),
                                 pads=[1, 1, 1, 1], act=1, w_layout=4)).get()
        ref = onp.relu(onp.batchnorm(np.ascontiguousarray(onp.conv2d(x, k, pads=[1, 1, 1, 1])), sc, sh) + res)
        assert_close(y, ref, RTOL, "winograd q4 fused")
    with pytest.raises(ValueError):
        q4.ConvQ4(xq, U, pads=[0, 0, 0, 0], w_layout=4)


def test_pointwise_q4_layers_match_nchw_kernels(pa):
    """Every HBM-bound Q4 layer equals its NCHW namesake bit for bit (same arithmetic, other layout);
    channel counts that are not multiples of 4 keep their padding lanes zero."""
    from planer_amd import q4
    rng = np.random.default_rng(5)
    for c in (3, 8, 13):
        x = rng.standard_normal((2, c, 12, 10)).astype(np.float32)
        dx = pa.asarray(x)
        xq = q4.to_q4(dx)
        pairs = [(q4.MaxpoolQ4(xq, (3, 3), (1, 1, 1, 1), (2, 2)), pa.Maxpool(dx, (3, 3), (1, 1, 1, 1), (2, 2))),
                 (q4.MaxpoolQ4(xq), pa.Maxpool(dx)),
                 (q4.AveragePoolQ4(xq, (2, 2), (0, 0, 0, 0), (2, 2)), pa.AveragePool(dx, (2, 2), (0, 0, 0, 0), (2, 2))),
                 (q4.UpSampleQ4(xq, np.array([1, 1, 2, 3], np.float32)), pa.UpSample(dx, np.array([1, 1, 2, 3], np.float32))),
                 (q4.LeakyReLUQ4(xq, 0.1), pa.LeakyReLU(dx, 0.1)),
                 (q4.AddQ4(xq, q4.to_q4(pa.asarray(x[::-1].copy()))), pa.Add(dx, pa.asarray(x[::-1].copy())))]
        k = pa.asarray(rng.uniform(0.5, 1.5, (1, c, 1, 1)).astype(np.float32))
        b = pa.asarray(rng.standard_normal((1, c, 1, 1)).astype(np.float32))
        pairs.append((q4.BatchNormQ4(xq, k, b), pa.BatchNorm(dx, k, b)))
        for got, want in pairs:
            np.testing.assert_array_equal(q4.from_q4(got).get(), want.get())
            np.testing.assert_array_equal(got.get(), q4_host(want.get()))
        np.testing.assert_array_equal(q4.GlobalAveragePoolQ4(xq).get(), pa.GlobalAveragePool(dx).get())
        r = q4.ReLUQ4(q4.to_q4(pa.asarray(x)))
        np.testing.assert_array_equal(q4.from_q4(r).get(), pa.ReLU(pa.asarray(x.copy())).get())
    a, b2 = rng.standard_normal((2, 8, 5, 6)).astype(np.float32), rng.standard_normal((2, 12, 5, 6)).astype(np.float32)
    got = q4.ConcatenateQ4(q4.to_q4(pa.asarray(a)), q4.to_q4(pa.asarray(b2)), axis=1)
    np.testing.assert_array_equal(q4.from_q4(got).get(), np.concatenate([a, b2], axis=1))
    np.testing.assert_array_equal(q4.from_q4(q4.SigmoidQ4(q4.to_q4(pa.asarray(a)))).get(), pa.Sigmoid(pa.asarray(a)).get())


def test_stem_conv_maxpool_marching_kernel_matches_oracle(pa):
    """conv_stem_pool_kernel (ConvPoolQ4): the row-packed 7x7 / stride 2 stem + its tail + maxpool(3x3, s2, p1) in one
    persistent kernel.  Heights that give partial strips and odd conv maps, channel counts that are not multiples of 64,
    relu / no activation (negative maxima then meet the zero padding and the -1e4 start, util.py:82-95), bias / bn tails;
    against the oracle, and against the conv kernel followed by the pool kernel (same values up to the K summation order)."""
    from planer_amd import q4
    rng = np.random.default_rng(123)
    para = dict(strides=[2, 2], pads=[3, 3, 3, 3], dilations=[1, 1], group=1)
    pool = dict(w=[3, 3], pads=[1, 1, 1, 1], strides=[2, 2])
    # (round 5: any width -- narrower maps take fewer 16-column blocks per workgroup, wider ones are cut into column chunks
    #  that recompute two conv columns at each border; odd widths end in masked columns)
    for n, h, cout, act, tail, wd in [(2, 224, 64, 1, "bn", 224), (1, 64, 64, 0, "bias", 224), (3, 100, 72, 1, "bn", 224),
                                      (2, 30, 8, 0, "none", 224), (1, 226, 132, 2, "bn", 224), (32, 224, 64, 1, "bn", 224),
                                      (2, 160, 64, 1, "bn", 160), (2, 256, 64, 1, "bn", 256), (1, 416, 64, 2, "bn", 416),
                                      (2, 64, 64, 0, "bias", 64), (1, 50, 36, 1, "bn", 37), (1, 90, 64, 0, "none", 231),
                                      (1, 70, 64, 1, "bn", 450), (1, 33, 12, 2, "bn", 1000), (2, 20, 64, 1, "bn", 12),
                                      (1, 40, 64, 1, "bn", 452), (2, 21, 20, 0, "bias", 8), (1, 47, 64, 2, "bn", 1000)]:
        x = rng.standard_normal((n, 3, h, wd)).astype(np.float32)
        K = (rng.standard_normal((cout, 3, 7, 7)) * 0.1).astype(np.float32)
        B = rng.standard_normal(cout).astype(np.float32) if tail == "bias" else None
        sc = rng.uniform(0.5, 1.5, (1, cout, 1, 1)).astype(np.float32) if tail == "bn" else None
        sh = (rng.standard_normal((1, cout, 1, 1)) * 0.3).astype(np.float32) if tail == "bn" else None
        dx, Kq = pa.asarray(x), q4.prepare_rowpack_weights(pa.asarray(K))
        dB, dsc, dsh = (pa.asarray(a) if a is not None else None for a in (B, sc, sh))
        assert q4.stem_pool_eligible(x.shape, K.shape, **para)
        one = q4.ConvPoolQ4(dx, Kq, dB, dsc, dsh, act=act, alpha=0.1, **para)
        assert "maxpool" in pa.hip.context().last_conv_plan(), pa.hip.context().last_conv_plan()
        two = q4.MaxpoolQ4(q4.ConvQ4(dx, Kq, dB, dsc, dsh, None, act=act, alpha=0.1, w_layout=6, **para), **pool)
        assert one.shape == two.shape and one.chan == two.chan
        conv = np.ascontiguousarray(onp.conv2d(x, K, B, **para))
        if tail == "bn":
            conv = onp.batchnorm(conv, sc, sh)
        conv = onp.relu(conv) if act == 1 else onp.leakyrelu(conv, 0.1) if act == 2 else conv
        want = onp.maxpool(conv, **pool)
        got = q4.from_q4(one).get()
        assert_close(got, want, RTOL, "stem + maxpool %s" % ((n, h, cout, act, tail, wd),))
        assert_close(one.get(), two.get(), 1e-5, "one kernel vs conv kernel + pool kernel")
        np.testing.assert_array_equal(one.get(), q4_host(got))                 # padding lanes of the last quad stay zero
        # the same kernel reading the NCHW tensor itself (w_layout 12: no row-packed copy, its own k order) where W % 4 == 0
        assert q4.stem_pool_nchw_eligible(x.shape, K.shape, **para) == (wd % 4 == 0)
        if wd % 4 == 0:
            Kn = q4.prepare_stem_nchw_weights(pa.asarray(K))
            direct = q4.ConvPoolQ4(dx, Kn, dB, dsc, dsh, act=act, alpha=0.1, w_layout=12, **para)
            assert "maxpool(nchw)" in pa.hip.context().last_conv_plan(), pa.hip.context().last_conv_plan()
            assert direct.shape == one.shape and direct.chan == one.chan
            gotn = q4.from_q4(direct).get()
            assert_close(gotn, want, RTOL, "stem + maxpool, NCHW input %s" % ((n, h, cout, act, tail, wd),))
            assert_close(direct.get(), one.get(), 1e-5, "NCHW-reading kernel vs row-packed kernel")
            np.testing.assert_array_equal(direct.get(), q4_host(gotn))
            # strips of 14 pooled rows (what throughput plans take at batch 32): half the workgroups, the same arithmetic
            tall = q4.ConvPoolQ4(dx, Kn, dB, dsc, dsh, act=act, alpha=0.1, w_layout=12, strip_rows=14, **para)
            assert "of 14 rows" in pa.hip.context().last_conv_plan(), pa.hip.context().last_conv_plan()
            np.testing.assert_array_equal(tall.get(), direct.get())
    with pytest.raises(NotImplementedError):
        q4.ConvPoolQ4(pa.asarray(np.zeros((1, 3, 64, 30), np.float32)), q4.prepare_stem_nchw_weights(pa.asarray(K[:8])), w_layout=12, **para)
    assert q4.stem_pool_eligible((1, 3, 224, 200), (64, 3, 7, 7), **para)
    assert not q4.stem_pool_eligible((1, 3, 224, 224), (64, 3, 3, 3), **dict(para, pads=[1, 1, 1, 1]))
    assert not q4.stem_pool_eligible((1, 3, 224, 224), (64, 3, 7, 7), **dict(para, strides=[1, 1]))
    with pytest.raises(NotImplementedError):
        q4.ConvPoolQ4(pa.asarray(np.zeros((1, 4, 64, 64), np.float32)), Kq, **para)


def test_rowpack_stem_conv_matches_oracle(pa):
    """Row-packed convolution for 1..3 input channels (pl_conv2d_rowpack_q4_f32): NCHW in, Q4 out;
    K runs (filter row, quads of the kw*Cin row segment) over a zero-padded NHWC copy of the input."""
    from planer_amd import q4
    ctx = pa.hip.context()
    names = _cfg_names(pa)
    rng = np.random.default_rng(31)
    shapes = [((2, 3, 33, 35), (20, 3, 7, 7), dict(strides=[2, 2], pads=[3, 3, 3, 3])),
              ((1, 1, 20, 21), (6, 1, 3, 3), dict(strides=[1, 1], pads=[1, 1, 1, 1])),
              ((2, 2, 15, 17), (9, 2, 5, 5), dict(strides=[2, 2], pads=[2, 2, 2, 2])),
              ((2, 3, 16, 16), (64, 3, 3, 3), dict(strides=[1, 1], pads=[1, 1, 1, 1])),
              ((1, 3, 12, 40), (8, 3, 3, 5), dict(strides=[1, 2], pads=[0, 2, 0, 2]))]
    for xs, ks, p in shapes:
        x = rng.standard_normal(xs).astype(np.float32)
        k = (rng.standard_normal(ks) * 0.1).astype(np.float32)
        b = rng.standard_normal(ks[0]).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, (1, ks[0], 1, 1)).astype(np.float32)
        sh = rng.standard_normal((1, ks[0], 1, 1)).astype(np.float32)
        dx, wq = pa.asarray(x), q4.prepare_rowpack_weights(pa.asarray(k))
        ref = np.ascontiguousarray(onp.conv2d(x, k, b, **p))
        yq = q4.ConvQ4(dx, wq, pa.asarray(b), w_layout=6, **p)
        assert_close(q4.from_q4(yq).get(), ref, RTOL, "rowpack %s" % (xs,))
        np.testing.assert_array_equal(yq.get(), q4_host(q4.from_q4(yq).get()))
        ref2 = onp.relu(onp.batchnorm(np.ascontiguousarray(onp.conv2d(x, k, **p)), sc, sh))
        y2 = q4.from_q4(q4.ConvQ4(dx, wq, None, pa.asarray(sc), pa.asarray(sh), None, act=1, w_layout=6, **p)).get()
        assert_close(y2, ref2, RTOL, "rowpack fused %s" % (xs,))
    # every channel-quad tile configuration and split-K on the stem-like shape
    xs, ks, p = shapes[0]
    x = rng.standard_normal(xs).astype(np.float32)
    k = (rng.standard_normal(ks) * 0.1).astype(np.float32)
    ref = np.ascontiguousarray(onp.conv2d(x, k, **p))
    dx, wq = pa.asarray(x), q4.prepare_rowpack_weights(pa.asarray(k))
    try:
        for name in [n for n in names if n.startswith("q")]:
            for split in (1, 2, 3):
                ctx.set_conv_config(names.index(name), split)
                y = q4.from_q4(q4.ConvQ4(dx, wq, w_layout=6, **p)).get()
                assert_close(y, ref, RTOL, "rowpack cfg %s split %d" % (name, split))
    finally:
        ctx.set_conv_config(-1, 0)
    with pytest.raises(ValueError):
        q4.ConvQ4(dx, wq, w_layout=6, strides=[2, 2], pads=[3, 3, 3, 3], dilations=[2, 2])


def test_winograd_f43_matches_oracle(pa):
    """Winograd F(4x4,3x3) on Q4 tensors: map sizes that are / are not multiples of 4, with and
    without the fused tail, and the real ResNet layer3/4 shapes (where its larger transform
    constants matter most: K = 2304 / 4608).  Same 1e-4 * max|ref| bar; the measured error is printed."""
    from planer_amd import q4
    rng = np.random.default_rng(37)
    worst = 0.0
    for (n, cin, h, w, cout) in [(2, 16, 7, 7, 24), (3, 20, 14, 13, 44), (1, 64, 9, 12, 64), (2, 48, 28, 28, 32),
                                 (4, 256, 14, 14, 256), (4, 512, 7, 7, 512)]:
        x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
        k = (rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, (1, cout, 1, 1)).astype(np.float32)
        sh = rng.standard_normal((1, cout, 1, 1)).astype(np.float32)
        res = rng.standard_normal((n, cout, h, w)).astype(np.float32)
        xq = q4.to_q4(pa.asarray(x))
        U = q4.prepare_winograd4_q4_weights(pa.asarray(k))
        y = q4.from_q4(q4.ConvQ4(xq, U, pa.asarray(b), pads=[1, 1, 1, 1], w_layout=7)).get()
        ref = np.ascontiguousarray(onp.conv2d(x, k, b, pads=[1, 1, 1, 1]))
        worst = max(worst, float(np.abs(y - ref).max() / np.abs(ref).max()))
        assert_close(y, ref, RTOL, "winograd F(4,3) %s" % ((n, cin, h, w, cout),))
        y = q4.from_q4(q4.ConvQ4(xq, U, None, pa.asarray(sc), pa.asarray(sh), q4.to_q4(pa.asarray(res)),
                                 pads=[1, 1, 1, 1], act=1, w_layout=7)).get()
        ref = onp.relu(onp.batchnorm(np.ascontiguousarray(onp.conv2d(x, k, pads=[1, 1, 1, 1])), sc, sh) + res)
        assert_close(y, ref, RTOL, "winograd F(4,3) fused")
    print("winograd F(4,3): worst error %.2e of max|ref|" % worst)
    assert worst < 3e-5                       # keep a 3x margin to the 1e-4 bar


def test_winograd_f43_filter_stationary_gemm_matches_oracle(pa, monkeypatch):
    """wino4_gemm_as_kernel (128 input channels: one frequency's filter block stays in LDS, V streams through): forced on
    for a ragged column count (T = 11 x 49 = 539, not a multiple of the 32-column sub-tile), two 128-row blocks, a map
    with tile padding, and the real layer2 shape at batch 8 and at batch 32 (what the bench runs); against the oracle and against the tiled kernel (different
    summation order: equal within 3e-5 of max|ref|)."""
    from planer_amd import q4
    rng = np.random.default_rng(41)
    for (n, h, w, cout) in [(11, 26, 26, 128), (3, 13, 17, 256), (8, 28, 28, 128), (32, 28, 28, 128)]:
        x = rng.standard_normal((n, 128, h, w)).astype(np.float32)
        k = (rng.standard_normal((cout, 128, 3, 3)) * (2.0 / (9 * 128)) ** 0.5).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, (1, cout, 1, 1)).astype(np.float32)
        sh = rng.standard_normal((1, cout, 1, 1)).astype(np.float32)
        res = rng.standard_normal((n, cout, h, w)).astype(np.float32)
        xq, U, rq = q4.to_q4(pa.asarray(x)), q4.prepare_winograd4_q4_weights(pa.asarray(k)), q4.to_q4(pa.asarray(res))
        outs = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("PLANER_HIP_WINO_GEMM_AS", flag)
            outs[flag] = q4.from_q4(q4.ConvQ4(xq, U, None, pa.asarray(sc), pa.asarray(sh), rq, pads=[1, 1, 1, 1], act=1, w_layout=7)).get()
            assert ("as128x32" in pa.hip.context().last_conv_plan()) == (flag == "1"), pa.hip.context().last_conv_plan()
        ref = onp.relu(onp.batchnorm(np.ascontiguousarray(onp.conv2d(x, k, pads=[1, 1, 1, 1])), sc, sh) + res)
        assert_close(outs["1"], ref, RTOL, "filter-stationary GEMM %s" % ((n, h, w, cout),))
        # two summation orders of F(4x4,3x3) (each within ~1.2e-5 of the oracle): 1.04e-5 apart at worst over the 3.2 M
        # outputs of the batch-32 shape
        assert np.abs(outs["1"] - outs["0"]).max() <= 3e-5 * np.abs(ref).max()


def test_winograd_1d_f43_fused_matches_oracle(pa):
    """Fused 1-D Winograd F(4,3) along W (conv_w1d4_kernel): widths that are / are not multiples of 4,
    Cout not a multiple of 64 or 4, K tails, fused tail; error bar as for the 2-D F(4,3)."""
    from planer_amd import q4
    rng = np.random.default_rng(41)
    worst = 0.0
    for (n, cin, h, w, cout) in [(2, 16, 7, 7, 24), (3, 20, 14, 13, 44), (1, 64, 9, 12, 64), (2, 48, 28, 28, 130),
                                 (1, 8, 5, 1, 6), (2, 4, 6, 3, 3), (2, 64, 56, 56, 64)]:
        x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
        k = (rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, (1, cout, 1, 1)).astype(np.float32)
        sh = rng.standard_normal((1, cout, 1, 1)).astype(np.float32)
        res = rng.standard_normal((n, cout, h, w)).astype(np.float32)
        xq = q4.to_q4(pa.asarray(x))
        U = q4.prepare_w1d4_q4_weights(pa.asarray(k))
        yq = q4.ConvQ4(xq, U, pa.asarray(b), pads=[1, 1, 1, 1], w_layout=8)
        y = q4.from_q4(yq).get()
        ref = np.ascontiguousarray(onp.conv2d(x, k, b, pads=[1, 1, 1, 1]))
        worst = max(worst, float(np.abs(y - ref).max() / np.abs(ref).max()))
        assert_close(y, ref, RTOL, "winograd-1d F(4,3) %s" % ((n, cin, h, w, cout),))
        np.testing.assert_array_equal(yq.get(), q4_host(y))
        y = q4.from_q4(q4.ConvQ4(xq, U, None, pa.asarray(sc), pa.asarray(sh), q4.to_q4(pa.asarray(res)),
                                 pads=[1, 1, 1, 1], act=1, w_layout=8)).get()
        ref = onp.relu(onp.batchnorm(np.ascontiguousarray(onp.conv2d(x, k, pads=[1, 1, 1, 1])), sc, sh) + res)
        assert_close(y, ref, RTOL, "winograd-1d F(4,3) fused")
    print("winograd-1d F(4,3): worst error %.2e of max|ref|" % worst)
    assert worst < 3e-5


def test_q4_stem_maxpool_block_kernel_is_bit_exact(pa):
    """maxpool_q4_k3s2p1_2x1 (a thread = two vertically adjacent outputs from one 5x3 window): odd / even extents, negative
    inputs (zero padding and the -1e4 start of util.py:88,95 show), against the oracle and the NCHW kernel."""
    from planer_amd import q4
    rng = np.random.default_rng(9)
    for shape in [(2, 8, 13, 15), (1, 5, 7, 9), (3, 4, 6, 6), (2, 64, 112, 112), (1, 3, 1, 1), (1, 4, 2, 5)]:
        x = rng.standard_normal(shape).astype(np.float32)
        x[0, 0] = -np.abs(x[0, 0]) - 0.5
        yq = q4.MaxpoolQ4(q4.to_q4(pa.asarray(x)), (3, 3), (1, 1, 1, 1), (2, 2))
        want = onp.maxpool(x, (3, 3), (1, 1, 1, 1), (2, 2))
        np.testing.assert_array_equal(q4.from_q4(yq).get(), want)
        np.testing.assert_array_equal(yq.get(), q4_host(want))


def test_upsample_concat_one_kernel_is_bit_exact(pa):
    """UpConcatQ4 = concat([upsample(a), b], channels) in one launch, against the two numpy calls; also the plain
    two-input concat (no upsampling) that ConcatenateQ4 now routes through the same kernel."""
    from planer_amd import q4
    rng = np.random.default_rng(5)
    for n, ca, cb, h, w, fh, fw in [(1, 256, 512, 13, 13, 2, 2), (2, 8, 4, 5, 7, 2, 3), (3, 4, 12, 6, 6, 1, 1), (1, 128, 256, 26, 26, 2, 2)]:
        a = rng.standard_normal((n, ca, h, w)).astype(np.float32)
        b = rng.standard_normal((n, cb, h * fh, w * fw)).astype(np.float32)
        k = np.array([1, 1, fh, fw], np.float32)
        got = q4.from_q4(q4.UpConcatQ4(q4.to_q4(pa.asarray(a)), pa.asarray(k), q4.to_q4(pa.asarray(b)))).get()
        np.testing.assert_array_equal(got, np.concatenate([onp.OPS["upsample"](a, k, mode="nearest"), b], 1))
    a, b = rng.standard_normal((2, 8, 9, 5)).astype(np.float32), rng.standard_normal((2, 12, 9, 5)).astype(np.float32)
    got = q4.from_q4(q4.ConcatenateQ4(q4.to_q4(pa.asarray(a)), q4.to_q4(pa.asarray(b)), axis=1)).get()
    np.testing.assert_array_equal(got, np.concatenate([a, b], 1))
