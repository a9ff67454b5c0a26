"""Seeded random sweep of the conv path on a real MI355X: shapes nobody hand-picked.

Every case draws (N, Cin, H, W, Cout, kernel, stride, pad, dilation, groups, tail) from a seeded generator, runs
the convolution through EVERY kernel family that accepts it -- the NCHW operator (`Conv2d`), the fused-epilogue
NCHW kernel, the channel-quad direct kernel, the row-packed small-Cin kernel, the fused 1-D Winograd F(2,3) / F(4,3)
kernels and the 2-D F(2x2,3x3) / F(4x4,3x3) pipelines -- under the autotuned launch plan, and compares each with the
oracle (tolerance 1e-4 of max|ref|, tests/conftest.RTOL).  Odd maps, channel counts that are not multiples of 4 or
of a tile, maps smaller than a Winograd tile and batches of one are all in the draw.
"""
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


def draw(seed):
    r = np.random.default_rng(seed)
    k = int(r.choice([1, 3, 3, 3, 5, 7]))
    kh, kw = (k, k) if r.random() < 0.85 else (int(r.choice([1, 3, 5])), int(r.choice([1, 3, 5])))
    st = int(r.choice([1, 1, 1, 2]))
    dil = int(r.choice([1, 1, 1, 2])) if max(kh, kw) > 1 else 1
    group = int(r.choice([1, 1, 1, 2, 4]))
    cin = int(r.choice([1, 2, 3, 4, 6, 8, 12, 16, 20, 32, 48, 64])) * (group if group > 1 else 1)
    if group > 1:
        cin = max(4 * group, cin // 4 * 4)               # grouped Q4 convs need Cin/group % 4 == 0
    cout = int(r.choice([1, 3, 4, 7, 8, 16, 24, 33, 64, 70, 130])) * 1
    if group > 1:
        cout = max(4 * group, (cout // (4 * group) + 1) * 4 * group)
    h, w = int(r.integers(1, 30)), int(r.integers(1, 30))
    ph, pw = (kh - 1) * dil // 2, (kw - 1) * dil // 2    # "same"-style, symmetric (the only defined case)
    if r.random() < 0.25:
        ph, pw = int(r.integers(0, ph + 2)), int(r.integers(0, pw + 2))
    h = max(h, (kh - 1) * dil + 1 - 2 * ph)
    w = max(w, (kw - 1) * dil + 1 - 2 * pw)
    n = int(r.choice([1, 1, 2, 3, 5]))
    tail = int(r.integers(0, 6))                         # 0 none, 1 bias, 2 bn, 3 bn+relu, 4 bn+res+relu, 5 bias+bn+leaky
    return dict(n=n, cin=cin, h=h, w=w, cout=cout, kh=kh, kw=kw, st=st, dil=dil, group=group, ph=ph, pw=pw, tail=tail)


def reference(c, x, K, B, sc, sh, res):
    y = np.ascontiguousarray(onp.conv2d(x, K, B, group=c["group"], strides=[c["st"]] * 2, dilations=[c["dil"]] * 2,
                                        pads=[c["ph"], c["pw"], c["ph"], c["pw"]]))
    if sc is not None:
        y = onp.batchnorm(y, sc, sh)
    if res is not None:
        y = y + res
    if c["tail"] in (3, 4):
        y = onp.relu(y)
    if c["tail"] == 5:
        y = onp.leakyrelu(y, 0.1)
    return y


@pytest.mark.parametrize("seed", range(120))
def test_random_conv_every_eligible_kernel_family(pa, seed):
    from planer_amd import q4
    c = draw(1000 + seed)
    r = np.random.default_rng(seed)
    x = r.standard_normal((c["n"], c["cin"], c["h"], c["w"])).astype(np.float32)
    K = (r.standard_normal((c["cout"], c["cin"] // c["group"], c["kh"], c["kw"])) * 0.3).astype(np.float32)
    B = r.standard_normal(c["cout"]).astype(np.float32) if c["tail"] in (1, 5) else None
    sc = r.uniform(0.5, 1.5, (1, c["cout"], 1, 1)).astype(np.float32) if c["tail"] >= 2 else None
    sh = (r.standard_normal((1, c["cout"], 1, 1)) * 0.3).astype(np.float32) if c["tail"] >= 2 else None
    para = dict(group=c["group"], strides=[c["st"]] * 2, dilations=[c["dil"]] * 2, pads=[c["ph"], c["pw"], c["ph"], c["pw"]])
    ho = (c["h"] + 2 * c["ph"] - (c["kh"] - 1) * c["dil"] - 1 + c["st"]) // c["st"]
    wo = (c["w"] + 2 * c["pw"] - (c["kw"] - 1) * c["dil"] - 1 + c["st"]) // c["st"]
    res = r.standard_normal((c["n"], c["cout"], ho, wo)).astype(np.float32) if c["tail"] == 4 else None
    want = reference(c, x, K, B, sc, sh, res)
    act, alpha = {3: 1, 4: 1, 5: 2}.get(c["tail"], 0), 0.1
    dx, dK = pa.asarray(x), pa.asarray(K)
    dB, dsc, dsh, dres = [None if a is None else pa.asarray(a) for a in (B, sc, sh, res)]
    what = "seed %d %s" % (seed, c)
    ran = []
    # -- the reference-shaped operator (NCHW in, NCHW out), plain conv + bias only
    if c["tail"] in (0, 1):
        assert_close(pa.Conv2d(dx, dK, dB, **para).get(), want, RTOL, "Conv2d " + what)
        ran.append("conv2d")
    # -- fused-epilogue NCHW kernel
    y = pa.layer.ConvFused(dx, dK, dB, dsc, dsh, dres, act=act, alpha=alpha, **para)
    assert_close(y.get(), want, RTOL, "ConvFused " + what)
    ran.append("fused-nchw")
    # -- channel-quad families
    grouped_ok = c["group"] == 1 or ((c["cin"] // c["group"]) % 4 == 0 and (c["cout"] // c["group"]) % 4 == 0)
    if grouped_ok:
        xq = q4.to_q4(dx)
        rq = q4.to_q4(dres) if dres is not None else None
        fams = [(2, lambda: q4.prepare_q4_weights(dK, c["group"]), xq)]
        if q4.rowpack_eligible(K.shape, **para) and c["cin"] < 4:
            fams.append((6, lambda: q4.prepare_rowpack_weights(dK), dx))
        if q4.w1d_q4_eligible(K.shape, **para):
            fams += [(8, lambda: q4.prepare_w1d4_q4_weights(dK), xq)]
        if q4.winograd_q4_eligible(K.shape, **para):
            fams += [(4, lambda: q4.prepare_winograd_q4_weights(dK), xq), (7, lambda: q4.prepare_winograd4_q4_weights(dK), xq),
                     (9, lambda: q4.prepare_wf4_q4_weights(dK), xq)]
        for lay, prep, xin in fams:
            try:
                yq = q4.ConvQ4(xin, prep(), dB, dsc, dsh, rq, act=act, alpha=alpha, w_layout=lay, **para)
            except NotImplementedError:
                # the fully fused F(4x4,3x3) kernel declines geometries whose 32-tile patch does not fit its LDS (the
                # picker then takes another family, Net._pick_conv_algo); every other family must take what it is eligible for
                assert lay == 9
                continue
            plan = pa.hip.context().last_conv_plan()
            assert_close(q4.from_q4(yq).get(), want, RTOL, "ConvQ4 w_layout %d [%s] %s" % (lay, plan, what))
            ran.append(lay)
    assert len(ran) >= 1


@pytest.mark.parametrize("seed", range(40))
def test_random_pointwise_layers_nchw_and_q4_are_bit_exact(pa, seed):
    """maxpool / averagepool / upsample / concat / add / batchnorm / relu / leakyrelu on random shapes: the NCHW kernels vs
    the oracle (bit-exact: no accumulation order involved except averagepool's, which follows the reference's tap
    order) and the channel-quad kernels vs the NCHW kernels."""
    from planer_amd import q4
    r = np.random.default_rng(5000 + seed)
    n, ch, h, w = int(r.choice([1, 2, 3])), int(r.choice([1, 3, 4, 5, 8, 13, 64])), int(r.integers(3, 40)), int(r.integers(3, 40))
    x = r.standard_normal((n, ch, h, w)).astype(np.float32)
    dx = pa.asarray(x)
    xq = q4.to_q4(dx)
    k, s, p = [(2, 2, 0), (3, 2, 1), (3, 1, 1), (2, 1, 0), (3, 3, 0)][seed % 5]
    pool = dict(w=[k, k], strides=[s, s], pads=[p] * 4)
    for kind, fn, ref in (("maxpool", q4.MaxpoolQ4, onp.maxpool), ("averagepool", q4.AveragePoolQ4, onp.OPS["averagepool"])):
        want = ref(x.copy(), **pool)
        got = pa.layer_map[kind](dx, **pool).get()
        np.testing.assert_array_equal(got, want, err_msg="%s %s %s" % (kind, x.shape, pool))
        np.testing.assert_array_equal(q4.from_q4(fn(xq, **pool)).get(), got)
    f = [1, 1, int(r.choice([1, 2, 3])), int(r.choice([1, 2, 3]))]
    up = pa.UpSample(dx, pa.asarray(np.array(f, np.float32)), mode="nearest").get()
    np.testing.assert_array_equal(up, onp.OPS["upsample"](x, np.array(f, np.float32), mode="nearest"))
    np.testing.assert_array_equal(q4.from_q4(q4.UpSampleQ4(xq, pa.asarray(np.array(f, np.float32)))).get(), up)
    ch2 = int(r.choice([1, 2, 4, 7, 8]))
    x2 = r.standard_normal((n, ch2, h, w)).astype(np.float32)
    cat = pa.Concatenate(dx, pa.asarray(x2), axis=1).get()
    np.testing.assert_array_equal(cat, np.concatenate([x, x2], 1))
    if ch % 4 == 0 and ch2 % 4 == 0:                     # Q4 concat joins whole quads
        np.testing.assert_array_equal(q4.from_q4(q4.ConcatenateQ4(xq, q4.to_q4(pa.asarray(x2)), axis=1)).get(), cat)
    y = r.standard_normal(x.shape).astype(np.float32)
    np.testing.assert_array_equal(q4.from_q4(q4.AddQ4(xq, q4.to_q4(pa.asarray(y)))).get(), x + y)
    sc, sh = r.uniform(0.5, 1.5, (1, ch, 1, 1)).astype(np.float32), r.standard_normal((1, ch, 1, 1)).astype(np.float32)
    bn = onp.batchnorm(x, sc, sh)
    np.testing.assert_array_equal(pa.BatchNorm(dx, pa.asarray(sc), pa.asarray(sh)).get(), bn)
    np.testing.assert_array_equal(q4.from_q4(q4.BatchNormQ4(xq, pa.asarray(sc), pa.asarray(sh))).get(), bn)
    np.testing.assert_array_equal(q4.from_q4(q4.LeakyReLUQ4(q4.to_q4(pa.asarray(x)), alpha=0.1)).get(), onp.leakyrelu(x, 0.1))
    np.testing.assert_array_equal(q4.from_q4(q4.ReLUQ4(q4.to_q4(pa.asarray(x)))).get(), onp.relu(x.copy()))


@pytest.mark.parametrize("seed", range(24))
def test_random_dense_small_batch_and_general(pa, seed):
    """layer.Dense on random (batch, K, N): batches up to 64 with K % 8 == 0 take the dedicated small-batch MFMA kernel,
    everything else the implicit-GEMM path; both against the oracle, with and without bias."""
    r = np.random.default_rng(9000 + seed)
    m = int(r.choice([1, 2, 5, 31, 32, 33, 64, 65, 100]))
    k = int(r.choice([64, 72, 96, 512, 1000, 1024, 77]))
    n = int(r.choice([10, 32, 33, 40, 255, 1000, 1031]))
    x = r.standard_normal((m, k)).astype(np.float32)
    W = (r.standard_normal((n, k)) * 0.1).astype(np.float32)
    B = r.standard_normal(n).astype(np.float32)
    y = pa.Dense(pa.asarray(x), pa.asarray(W), pa.asarray(B)).get()
    plan = pa.hip.context().last_conv_plan()
    assert ("dense32x32" in plan) == (m <= 64 and n >= 32 and k % 8 == 0), (plan, m, k, n)
    assert_close(y, onp.dense(x, W, B), RTOL, "dense %s [%s]" % ((m, k, n), plan))
    again = pa.Dense(pa.asarray(x), pa.asarray(W), pa.asarray(B)).get()
    np.testing.assert_array_equal(y, again)              # fixed summation order


def _try(fn):
    """An input a kernel does not cover must raise NotImplementedError / ValueError (never return something else)."""
    try:
        return fn()
    except (NotImplementedError, ValueError):
        return None


@pytest.mark.parametrize("seed", range(60))
def test_random_second_wave_ops_match_the_oracle(pa, seed):
    """Slice / Pad / Tile / Expand / Transpose / Reshape / Squeeze / Unsqueeze / Split (pure data movement: bit-exact),
    Softmax / LogSoftmax / Reduce* over random axes, Gather with random indices, Where / comparisons, ConvTranspose with
    random stride / pad / dilation / output_padding -- random ranks and shapes, each against the oracle.  A combination
    a kernel does not cover may raise; it may not return a different answer."""
    r = np.random.default_rng(7000 + seed)
    nd = int(r.integers(1, 5))
    shape = tuple(int(v) for v in r.integers(1, 9, nd))
    x = r.standard_normal(shape).astype(np.float32)
    dx = lambda: pa.asarray(x.copy())
    lm, O = pa.layer_map, onp.OPS
    i64 = lambda *v: np.array(v, np.int64)

    def same(kind, args, dargs, exact=True, **para):
        try:
            want = O[kind](*[a.copy() if isinstance(a, np.ndarray) else a for a in args], **para)
        except ValueError:                     # a geometry the reference itself cannot run (e.g. an empty conv output)
            return
        got = _try(lambda: lm[kind](*dargs, **para))
        if got is None:
            return
        wants = want if isinstance(want, (list, tuple)) else [want]
        gots = got if isinstance(got, (list, tuple)) else [got]
        assert len(wants) == len(gots), kind
        for g_, w_ in zip(gots, wants):
            g_ = g_.get() if hasattr(g_, "get") else np.asarray(g_)
            assert g_.shape == np.asarray(w_).shape, (kind, g_.shape, np.asarray(w_).shape, para)
            if exact:
                np.testing.assert_array_equal(g_, w_, err_msg="%s %s %s" % (kind, shape, para))
            else:
                assert_close(g_, w_, RTOL, "%s %s %s" % (kind, shape, para))

    # slice: random axes subset, negative / out-of-range bounds, negative steps
    axes = sorted(r.choice(nd, int(r.integers(1, nd + 1)), replace=False).tolist())
    st, en, sp = [], [], []
    for a in axes:
        step = int(r.choice([1, 1, 2, 3, -1, -2]))
        lo, hi = int(r.integers(-shape[a] - 1, shape[a] + 2)), int(r.integers(-shape[a] - 1, shape[a] + 2))
        st.append(lo); en.append(hi); sp.append(step)
    sl = [i64(*st), i64(*en), i64(*axes), i64(*sp)]
    same("slice", [x] + sl, [dx()] + sl)
    # pad (constant), tile, expand
    pads = i64(*r.integers(0, 3, 2 * nd).tolist())
    same("pad", [x, pads], [dx(), pads], constant_value=float(r.integers(-2, 3)))
    rep = i64(*r.integers(1, 4, nd).tolist())
    same("tile", [x, rep], [dx(), rep])
    ex_shape = [int(v) for v in r.integers(1, 4, int(r.integers(0, 2)))] + [s if r.random() < 0.7 else s for s in shape]
    xe = x[tuple(slice(0, 1) if r.random() < 0.4 else slice(None) for _ in shape)]
    same("expand", [xe, i64(*ex_shape)], [pa.asarray(np.ascontiguousarray(xe)), i64(*ex_shape)])
    # transpose / reshape / squeeze / unsqueeze
    perm = r.permutation(nd).tolist()
    same("transpose", [x], [dx()], axis=perm)
    same("reshape", [x, i64(-1)], [dx(), i64(-1)])
    if nd >= 2:
        same("reshape", [x, i64(0, -1)], [dx(), i64(0, -1)])
    ux = int(r.integers(0, nd + 1))
    same("unsqueeze", [x], [dx()], axes=[ux])
    ones = [i for i, s in enumerate(shape) if s == 1]
    if ones:
        same("squeeze", [x], [dx()], axes=[ones[0]])
    # split along a random axis (the reference's leading slice is along axis 0)
    ax = int(r.integers(0, nd))
    parts = []
    left = shape[ax]
    while left > 0:
        p = int(r.integers(1, left + 1)); parts.append(p); left -= p
    if sum(parts) <= shape[0]:
        same("split", [x], [dx()], split=parts, axis=ax)
    # softmax / logsoftmax / reductions
    ax = int(r.integers(-nd, nd))
    same("softmax", [x], [dx()], exact=False, axis=ax)
    same("logsoftmax", [x], [dx()], exact=False, axis=ax)
    red_axes = sorted(r.choice(nd, int(r.integers(1, nd + 1)), replace=False).tolist())
    kd = bool(r.integers(0, 2))
    for kind in ("reducesum", "reducemean"):
        same(kind, [x], [dx()], exact=False, axes=red_axes, keepdims=kd)
    for kind in ("reducemax", "reducemin"):
        same(kind, [x], [dx()], axes=red_axes, keepdims=kd)
    # gather, comparisons, where
    ax = int(r.integers(0, nd))
    idx = r.integers(-shape[ax], shape[ax], tuple(int(v) for v in r.integers(1, 4, int(r.integers(0, 3)))))
    same("gather", [x, idx.astype(np.int64)], [dx(), idx.astype(np.int64)], axis=ax)
    q = np.round(x)
    y = np.round(r.standard_normal(shape)).astype(np.float32)
    for kind in ("equal", "greater", "greaterorequal"):
        same(kind, [q, y], [pa.asarray(q), pa.asarray(y)])
    m = r.random(shape) > 0.5
    same("where", [m, x, y], [pa.asarray(m), dx(), pa.asarray(y)])
    same("where", [m, x, np.array([0.5], np.float32)], [pa.asarray(m), dx(), np.array([0.5], np.float32)])
    # convtranspose: random geometry
    n, ci, co = int(r.integers(1, 3)), int(r.choice([1, 3, 4, 8])), int(r.choice([1, 2, 5, 8]))
    k = int(r.choice([1, 2, 3, 4]))
    s_, d_ = int(r.choice([1, 2, 3])), int(r.choice([1, 1, 2]))
    pd = int(r.integers(0, (k - 1) * d_ + 1))
    op = int(r.integers(0, max(s_, d_)))
    xt = r.standard_normal((n, ci, int(r.integers(1, 8)), int(r.integers(1, 8)))).astype(np.float32)
    Kt = (r.standard_normal((ci, co, k, k)) * 0.3).astype(np.float32)
    Bt = r.standard_normal(co).astype(np.float32)
    para = dict(strides=[s_, s_], dilations=[d_, d_], pads=[pd] * 4, output_padding=[op, op])
    same("convtranspose", [xt, Kt, Bt], [pa.asarray(xt), pa.asarray(Kt), pa.asarray(Bt)], exact=False, **para)
