"""Per-layer parity on a real MI355X: HIP kernels (through the C ABI) vs the
vectors captured from the reference and vs the oracle.  Tolerance: 1e-4
relative to max|ref| per tensor (north_star), set in tests/conftest.RTOL."""
import numpy as np
import pytest

from oracle import planer_np as onp
from tests.cases import layer_cases
from tests.conftest import RTOL, assert_close

pytestmark = pytest.mark.gpu
CASES = layer_cases()


@pytest.fixture(scope="module")
def pa():
    import planer_amd
    planer_amd.hip.context()          # raises loudly without a GPU / library
    return planer_amd


def run_hip(pa, kind, args, params):
    dev = [pa.asarray(a.copy()) for a in args]
    out = pa.layer_map[kind](*dev, **params)
    outs = tuple(out) if isinstance(out, (tuple, list)) else (out,)
    return dev, outs


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_layer_vs_reference_vectors(pa, case, golden_layers):
    name, kind, args, params = case
    z, meta = golden_layers
    dev, outs = run_hip(pa, kind, args, params)
    assert len(outs) == meta[name]["n_out"]
    for i, o in enumerate(outs):
        ref = z["%s/out%d" % (name, i)]
        assert o.shape == ref.shape
        assert_close(o.get(), ref, RTOL, name)
    # ReLU aliasing: the reference returns its (mutated) input object
    assert bool(outs[0] is dev[0]) == meta[name]["inplace"]


EXACT = ["relu", "leakyrelu_0.1", "leakyrelu_default", "add", "batchnorm", "maxpool_k2s2", "maxpool_k3s2p1_neg",
         "maxpool_k3s2p1", "maxpool_clamp_-1e4", "upsample_x2", "upsample_2x3", "concat_axis1", "concat_axis0_3",
         "flatten", "add_bcast_channel", "sub", "mul", "div", "sub_bcast_channel", "mul_bcast_channel",
         "div_bcast_channel", "sub_scalar_lhs", "mul_scalar_lhs", "div_scalar_lhs", "add_scalar", "sqrt",
         "reciprocal", "hardsigmoid", "hardsigmoid_ab", "clip", "clip_relu6", "reducemax_hw", "reducemin_last",
         "transpose_0231", "transpose_10", "reshape_keep0", "squeeze", "unsqueeze", "resize_nearest_x2",
         "resize_asym_floor", "slice_basic", "slice_step_neg", "slice_default_axes", "pad_hw", "pad_value", "tile_2d",
         "tile_more_reps", "expand_channel", "expand_lower_rank", "split_axis1", "split_axis0",
         "reducemax_axis0", "shape", "gather_axis0", "gather_axis2_neg_2d_idx", "gather_shape_scalar", "cast_f32_i64",
         "cast_i64_f32", "cast_f32_bool", "range", "equal", "greater_scalar", "greaterorequal", "equal_shape_tensors",
         "where", "where_scalar_rhs", "constantofshape_f32", "constantofshape_i64", "erf", "concat_shape_tensors",
         "mul_shape_tensors", "scatternd_rows", "scatternd_elements", "nonzero_f32", "nonzero_bool", "nonzero_none",
         "nonzero_i64_1d", "topk_last", "topk_axis1", "topk_smallest_quirk", "topk_yolo_candidates", "topk_long_row",
         "topk_long_row_smallest", "add_bcast_rows", "sub_bcast_outer", "mul_bcast_cross", "div_bcast_lower_rank_lhs",
         "add_bcast_batch", "mul_bcast_spatial", "resize_linear_frac", "resize_linear_down", "resize_linear_size",
         # round 6: nearest Resize under every shifting (transform, rounding) pair, np.pad's index-map modes
         "resize_asym_ceil", "resize_asym_prefer_ceil", "resize_asym_prefer_floor", "resize_half_floor", "resize_half_ceil",
         "resize_half_floor_rows_only", "resize_asym_ceil_cols_only", "resize_unknown_modes", "resize_nearest_truncated",
         "pad_reflect_hw", "pad_edge_hw", "pad_symmetric", "pad_wrap", "pad_reflect_wide", "pad_reflect_len1"]


@pytest.mark.parametrize("name", EXACT)
def test_copy_and_compare_ops_are_bit_exact(pa, name, golden_layers):
    """Ops without accumulation must reproduce the reference bit for bit."""
    z, meta = golden_layers
    _, kind, args, params = [c for c in CASES if c[0] == name][0]
    _, outs = run_hip(pa, kind, args, params)
    for i, o in enumerate(outs):
        np.testing.assert_array_equal(o.get(), z["%s/out%d" % (name, i)])


def _cfg_names(pa):
    import ctypes
    lib = pa._lib.load()
    names = []
    for c in range(lib.pl_conv2d_num_configs()):
        buf = ctypes.create_string_buffer(32)
        lib.pl_conv2d_config_name(c, buf, 32)
        names.append(buf.value.decode())
    return names


def test_conv_every_tile_config_and_split_k(pa):
    """All tile shapes of both implicit-GEMM kernels (generic OIHW and tap-major) and the
    split-K path agree with the oracle."""
    ctx = pa.hip.context()
    names = _cfg_names(pa)
    assert any(n.startswith("t") for n in names) and any(not n.startswith("t") for n in names)
    rng = np.random.default_rng(7)
    shapes = [((3, 32, 14, 14), (40, 32, 3, 3), dict(strides=[1, 1], pads=[1, 1, 1, 1])),
              ((2, 3, 33, 35), (20, 3, 7, 7), dict(strides=[2, 2], pads=[3, 3, 3, 3])),
              ((2, 64, 7, 7), (130, 64, 1, 1), dict(strides=[1, 1], pads=[0, 0, 0, 0])),
              ((2, 64, 13, 11), (70, 64, 3, 3), dict(strides=[2, 2], pads=[1, 1, 1, 1])),
              ((2, 64, 9, 9), (48, 32, 3, 3), dict(strides=[1, 1], pads=[2, 2, 2, 2], dilations=[2, 2], group=2))]
    try:
        for xs, ks, p in shapes:
            x = rng.standard_normal(xs).astype(np.float32)
            k = (rng.standard_normal(ks) * 0.1).astype(np.float32)
            b = rng.standard_normal(ks[0]).astype(np.float32)
            ref = np.ascontiguousarray(onp.conv2d(x, k, b, **p))
            dx, dk, db = pa.asarray(x), pa.asarray(k), pa.asarray(b)
            dkt = pa.prepare_conv_weights(dk) if ks[1] % 16 == 0 else None
            for cfg, name in enumerate(names):
                if name.startswith("q"):
                    continue                      # channel-quad configs: tests/test_gpu_q4.py
                tap = name.startswith("t")
                if tap and (dkt is None or ks[1] % int(name.split("x")[-1])):
                    continue
                for split in (1, 2, 3):
                    ctx.set_conv_config(cfg, split)
                    y = pa.ConvFused(dx, dkt if tap else dk, db, w_layout=int(tap), **p).get()
                    assert_close(y, ref, RTOL, "cfg %s split %d %s" % (name, split, xs))
    finally:
        ctx.set_conv_config(-1, 0)


def test_hybrid_split_k_plans_are_deterministic(pa):
    """Hybrid launch plans (data-parallel prefix + split-K tail with compact slabs and the tile
    reduce): many tiles, many slices, occupancy pins, repeated launches; every run must be
    bit-identical and match the oracle."""
    ctx = pa.hip.context()
    names = _cfg_names(pa)
    rng = np.random.default_rng(21)
    x = rng.standard_normal((8, 64, 28, 28)).astype(np.float32)
    k = (rng.standard_normal((128, 64, 3, 3)) * 0.05).astype(np.float32)
    res = rng.standard_normal((8, 128, 28, 28)).astype(np.float32)
    ref = onp.relu(np.ascontiguousarray(onp.conv2d(x, k, pads=[1, 1, 1, 1])) + res)
    dx, dk, dres = pa.asarray(x), pa.asarray(k), pa.asarray(res)
    dkt = pa.prepare_conv_weights(dk)
    try:
        for name, dp, split, occ in [("t64x64x16", 0, 4, 0), ("t64x64x16", 64, 6, 0), ("t128x64x16", 0, 9, 0),
                                     ("64x64", 0, 3, 0), ("t64x64x32", 128, 2, 4), ("128x32", 8, 5, 2)]:
            tap = name.startswith("t")
            ctx.set_conv_plan(names.index(name), dp, split, occ)
            first = None
            for it in range(12):
                y = pa.ConvFused(dx, dkt if tap else dk, None, None, None, dres, pads=[1, 1, 1, 1], act=1,
                                 w_layout=int(tap)).get()
                if first is None:
                    first = y
                    assert_close(y, ref, RTOL, "%s dp%d s%d" % (name, dp, split))
                else:
                    np.testing.assert_array_equal(y, first)
    finally:
        ctx.set_conv_config(-1, 0)


def test_winograd_path_matches_direct_conv(pa):
    """Winograd F(2x2,3x3) pipeline (input transform, 16 grouped GEMMs on the MFMA kernel, output
    transform + fused tail) vs the oracle, odd and even maps, with and without the epilogue."""
    rng = np.random.default_rng(17)
    for (n, cin, h, w, cout) in [(2, 16, 7, 7, 24), (3, 32, 14, 14, 40), (1, 64, 9, 13, 64), (2, 48, 28, 28, 32)]:
        x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
        k = (rng.standard_normal((cout, cin, 3, 3)) * 0.1).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, (1, cout, 1, 1)).astype(np.float32)
        sh = rng.standard_normal((1, cout, 1, 1)).astype(np.float32)
        res = rng.standard_normal((n, cout, h, w)).astype(np.float32)
        dx = pa.asarray(x)
        U = pa.prepare_winograd_weights(pa.asarray(k))
        y = pa.ConvFused(dx, U, pa.asarray(b), pads=[1, 1, 1, 1], w_layout=3).get()
        ref = np.ascontiguousarray(onp.conv2d(x, k, b, pads=[1, 1, 1, 1]))
        assert_close(y, ref, RTOL, "winograd %s" % ((n, cin, h, w, cout),))
        y = pa.ConvFused(dx, U, None, pa.asarray(sc), pa.asarray(sh), pa.asarray(res), pads=[1, 1, 1, 1], act=1,
                         w_layout=3).get()
        ref = onp.relu(onp.batchnorm(np.ascontiguousarray(onp.conv2d(x, k, pads=[1, 1, 1, 1])), sc, sh) + res)
        assert_close(y, ref, RTOL, "winograd fused")
    with pytest.raises(ValueError):
        pa.ConvFused(dx, U, pads=[0, 0, 0, 0], w_layout=3)


def test_autotune_and_heuristic_agree(pa):
    ctx = pa.hip.context()
    lib = pa._lib.load()
    rng = np.random.default_rng(9)
    x = rng.standard_normal((4, 64, 20, 20)).astype(np.float32)
    k = (rng.standard_normal((96, 64, 3, 3)) * 0.1).astype(np.float32)
    ref = np.ascontiguousarray(onp.conv2d(x, k, pads=[1, 1, 1, 1]))
    dx, dk = pa.asarray(x), pa.asarray(k)
    dkt = pa.prepare_conv_weights(dk)
    for tune in (1, 0, 1):
        lib.pl_set_autotune(ctx.handle, tune)
        assert_close(pa.Conv2d(dx, dk, pads=[1, 1, 1, 1]).get(), ref, RTOL)
        assert_close(pa.ConvFused(dx, dkt, pads=[1, 1, 1, 1], w_layout=1).get(), ref, RTOL)
    lib.pl_set_autotune(ctx.handle, 1)


def test_conv_fused_epilogue_matches_layer_by_layer(pa):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 16, 12, 12)).astype(np.float32)
    k = (rng.standard_normal((24, 16, 3, 3)) * 0.1).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, (1, 24, 1, 1)).astype(np.float32)
    sh = rng.standard_normal((1, 24, 1, 1)).astype(np.float32)
    res = rng.standard_normal((2, 24, 12, 12)).astype(np.float32)
    ctx = pa.hip.context()
    for split in (1, 2):
        ctx.set_conv_config(-1, split)
        for act, alpha in ((0, 0.0), (1, 0.0), (2, 0.1)):
            y = pa.ConvFused(pa.asarray(x), pa.asarray(k), None, pa.asarray(sc), pa.asarray(sh), pa.asarray(res),
                             pads=[1, 1, 1, 1], act=act, alpha=alpha).get()
            ref = onp.add(onp.batchnorm(np.ascontiguousarray(onp.conv2d(x, k, pads=[1, 1, 1, 1])), sc, sh), res)
            ref = onp.relu(ref) if act == 1 else (onp.leakyrelu(ref, alpha) if act == 2 else ref)
            assert_close(y, ref, RTOL)
    ctx.set_conv_config(-1, 0)


def test_config2_single_conv_full_size(pa):
    """BASELINE config 2: Conv2d 3->64 k3 s1 p1 on (8,3,224,224), vs the oracle."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((8, 3, 224, 224)).astype(np.float32)
    k = (rng.standard_normal((64, 3, 3, 3)) * 0.1).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    y = pa.Conv2d(pa.asarray(x), pa.asarray(k), pa.asarray(b), strides=[1, 1], pads=[1, 1, 1, 1]).get()
    assert y.shape == (8, 64, 224, 224)
    assert_close(y, np.ascontiguousarray(onp.conv2d(x, k, b, pads=[1, 1, 1, 1])), RTOL)


def test_conv_linearity_at_resnet_layer_size(pa):
    """Size-independent property: conv(a*x1 + x2) == a*conv(x1) + conv(x2)."""
    rng = np.random.default_rng(11)
    k = pa.asarray((rng.standard_normal((128, 128, 3, 3)) * 0.03).astype(np.float32))
    x1 = rng.standard_normal((32, 128, 28, 28)).astype(np.float32)
    x2 = rng.standard_normal((32, 128, 28, 28)).astype(np.float32)
    f = lambda a: pa.Conv2d(pa.asarray(a), k, strides=[1, 1], pads=[1, 1, 1, 1]).get()
    assert_close(f(2.5 * x1 + x2), 2.5 * f(x1) + f(x2), RTOL)


def test_maxpool_fast_path_is_bit_exact(pa):
    """The vectorised 3x3/s2/p1 kernel (even maps, Wo % 4 == 0) vs the oracle, negatives included so
    the zero-padding / -1e4 rule shows at the borders."""
    rng = np.random.default_rng(4)
    for shape in [(2, 5, 16, 16), (3, 4, 24, 8), (1, 2, 112, 112)]:
        x = (rng.standard_normal(shape) * 3 - 1).astype(np.float32)
        y = pa.Maxpool(pa.asarray(x), w=[3, 3], pads=[1, 1, 1, 1], strides=[2, 2]).get()
        np.testing.assert_array_equal(y, onp.maxpool(x, (3, 3), (1, 1, 1, 1), (2, 2)))
    x = np.full((1, 1, 8, 8), -2e4, np.float32)
    y = pa.Maxpool(pa.asarray(x), w=[3, 3], pads=[1, 1, 1, 1], strides=[2, 2]).get()
    np.testing.assert_array_equal(y, onp.maxpool(x, (3, 3), (1, 1, 1, 1), (2, 2)))


def test_unsupported_inputs_fail_loudly(pa):
    x = pa.asarray(np.zeros((1, 4, 8, 8), np.float32))
    k = pa.asarray(np.zeros((4, 4, 3, 3), np.float32))
    with pytest.raises(NotImplementedError):          # asymmetric pads: undefined in the reference
        pa.Conv2d(x, k, pads=[1, 1, 0, 0])
    with pytest.raises(NotImplementedError):
        pa.Maxpool(x, w=[2, 2], pads=[1, 0, 0, 0])
    with pytest.raises(TypeError):                    # the reference indexes B / initial_h / initial_c per direction
        pa.layer_map["lstm"](x.reshape(4, 16, 4), x.reshape(1, 16, 16), x.reshape(1, 16, 16))
    with pytest.raises(NotImplementedError):          # arange(k) * -largest only means something for 0 / 1
        pa.layer_map["topk"](x, np.array([2]), largest=2)
    with pytest.raises(IndexError):                   # np.take would raise too
        pa.layer_map["topk"](x, np.array([9]))
    with pytest.raises(IndexError):
        pa.layer_map["scatternd"](x, np.array([[[0, 4]]]), pa.asarray(np.zeros((1, 1, 8, 8), np.float32)))
    with pytest.raises(NotImplementedError):          # an update that would need broadcasting
        pa.layer_map["scatternd"](x, np.array([[[0, 1]]]), pa.asarray(np.zeros((1, 1, 1, 8), np.float32)))
    assert pa.layer.NOT_ON_DEVICE == []
    with pytest.raises(ValueError):
        pa.Softmax(x, axis=4)
    with pytest.raises(NotImplementedError):          # stacks with different leading dimensions
        pa.MatMul(pa.asarray(np.zeros((2, 3, 4, 5), np.float32)), pa.asarray(np.zeros((3, 2, 5, 6), np.float32)))
    with pytest.raises((ValueError, NotImplementedError)):
        pa.Conv2d(x, pa.asarray(np.zeros((4, 3, 3, 3), np.float32)))


def test_empty_and_ragged(pa):
    e = pa.asarray(np.zeros((0, 4, 8, 8), np.float32))
    assert pa.ReLU(e).shape == (0, 4, 8, 8)
    k = pa.asarray(np.ones((4, 4, 3, 3), np.float32))
    assert pa.Conv2d(e, k, pads=[1, 1, 1, 1]).shape == (0, 4, 8, 8)
    # unaligned views: a[1] of an odd-sized tensor starts 4*35 bytes in
    a = np.random.default_rng(5).standard_normal((3, 5, 7)).astype(np.float32)
    d = pa.asarray(a)
    np.testing.assert_array_equal(pa.LeakyReLU(d[1], 0.1).get(), onp.leakyrelu(a[1].copy(), 0.1))
    np.testing.assert_array_equal(pa.Add(d[1], d[2]).get(), a[1] + a[2])


def test_small_cin_store_stream_kernel_matches_oracle(pa, monkeypatch):
    """conv_smallcin_nchw_kernel (forced here with PLANER_HIP_SMALLCIN=1; by default it serves big outputs only): 3x3 / stride 1 on 1..4 input channels, the
    BASELINE config-2 shape class -- ragged widths, pad 0 / 1, channel counts that are not multiples of 64."""
    monkeypatch.setenv("PLANER_HIP_SMALLCIN", "1")
    monkeypatch.setenv("PLANER_HIP_SMALLCIN_VALU", "0")      # (the vector-ALU kernel would take the shapes with Wo % 4 == 0)
    rng = np.random.default_rng(77)
    for n, c, h, w, co, pad, bias in [(2, 3, 9, 11, 20, 1, True), (1, 1, 5, 300, 70, 0, False), (3, 4, 17, 16, 64, 1, True),
                                      (2, 2, 40, 7, 130, 1, True), (2, 3, 64, 64, 64, 1, True)]:
        x = rng.standard_normal((n, c, h, w)).astype(np.float32)
        k = (rng.standard_normal((co, c, 3, 3)) * 0.1).astype(np.float32)
        b = rng.standard_normal(co).astype(np.float32) if bias else None
        y = pa.Conv2d(pa.asarray(x), pa.asarray(k), pa.asarray(b) if bias else None, pads=[pad] * 4)
        assert pa.hip.context().last_conv_plan().startswith("smallcin3x3")
        assert_close(y.get(), np.ascontiguousarray(onp.conv2d(x, k, b, pads=[pad] * 4)), RTOL, str((n, c, h, w, co)))


def test_small_cin_vector_alu_kernel_matches_oracle(pa, monkeypatch):
    """conv_smallcin_valu_kernel (the default for big outputs whose rows are whole pixel quads; forced here): no MFMA, a lane =
    4 pixels x 8 channels, filter broadcast from LDS.  Every Cin 1..4, pad 0 / 1, channel counts that are not multiples of
    8 or of the workgroup's channel block, quad counts that are not multiples of 256, every channels-per-workgroup setting,
    bias / no bias; config 2 at full size against the oracle and against the MFMA store-stream kernel (same k order: equal
    within rounding of the different FMA grouping -- both are single fmaf chains, so bit-equal)."""
    monkeypatch.setenv("PLANER_HIP_SMALLCIN_VALU", "1")
    rng = np.random.default_rng(78)
    for n, c, h, w, co, pad, bias in [(3, 4, 17, 16, 64, 1, True), (2, 2, 40, 8, 130, 1, True), (2, 3, 64, 64, 64, 1, False),
                                      (1, 1, 6, 302, 70, 0, True), (3, 3, 50, 44, 20, 1, True), (2, 3, 9, 14, 5, 0, True),
                                      (8, 3, 224, 224, 64, 1, True)]:
        x = rng.standard_normal((n, c, h, w)).astype(np.float32)
        k = (rng.standard_normal((co, c, 3, 3)) * 0.1).astype(np.float32)
        b = rng.standard_normal(co).astype(np.float32) if bias else None
        args = (pa.asarray(x), pa.asarray(k), pa.asarray(b) if bias else None)
        ref = np.ascontiguousarray(onp.conv2d(x, k, b, pads=[pad] * 4))
        first = None
        for cpb in ("", "8", "16", "64"):
            monkeypatch.setenv("PLANER_HIP_SCV_CPB", cpb) if cpb else monkeypatch.delenv("PLANER_HIP_SCV_CPB", raising=False)
            y = pa.Conv2d(*args, pads=[pad] * 4).get()
            assert pa.hip.context().last_conv_plan().startswith("smallcin3x3valu"), pa.hip.context().last_conv_plan()
            assert_close(y, ref, RTOL, str((n, c, h, w, co, cpb)))
            first = y if first is None else first
            np.testing.assert_array_equal(y, first)       # the channel blocking does not change a value
    monkeypatch.delenv("PLANER_HIP_SCV_CPB", raising=False)
    monkeypatch.setenv("PLANER_HIP_SMALLCIN_VALU", "0")
    monkeypatch.setenv("PLANER_HIP_SMALLCIN", "1")
    y1 = pa.Conv2d(*args, pads=[1] * 4).get()
    assert pa.hip.context().last_conv_plan().startswith("smallcin3x3w "), pa.hip.context().last_conv_plan()
    np.testing.assert_array_equal(first, y1)


def test_sorting_and_data_dependent_ops_edge_cases(pa):
    """TopK with NaNs / repeated values / k = n / long rows, NonZero with nothing, everything and NaN, ScatterND with
    no updates, LSTM with a single step -- against numpy's own answers."""
    rng = np.random.default_rng(91)
    topk, nonzero, scat, lstm = (pa.layer_map[k] for k in ("topk", "nonzero", "scatternd", "lstm"))
    # NaN sorts last (numpy): with largest=1 the NaNs come first, then the greatest finite values
    x = rng.standard_normal((3, 37)).astype(np.float32)
    x[0, 5] = x[2, 0] = x[2, 36] = np.nan
    v, i = topk(pa.asarray(x), np.array([4]))
    want_v, want_i = onp.topk(x, np.array(4))
    np.testing.assert_array_equal(v.get(), want_v)
    np.testing.assert_array_equal(x[np.arange(3)[:, None], i.get()], want_v)          # indices point at those values
    # repeated values: the values agree with numpy, the indices point at equal values and are all different
    y = np.round(rng.standard_normal((2, 300)) * 2).astype(np.float32)
    v, i = topk(pa.asarray(y), np.array([300]))                                        # k = n: a full sort
    np.testing.assert_array_equal(v.get(), np.sort(y, axis=-1)[:, ::-1])
    ii = i.get()
    np.testing.assert_array_equal(np.take_along_axis(y, ii, -1), v.get())
    assert all(len(set(row.tolist())) == 300 for row in ii)
    # a row longer than the LDS sort holds, with duplicates of the maximum
    z = rng.standard_normal(40000).astype(np.float32)
    z[[7, 39999, 20000]] = 9.0
    v, i = topk(pa.asarray(z), np.array([5]))
    np.testing.assert_array_equal(v.get(), np.sort(z)[::-1][:5])
    assert sorted(i.get()[:3].tolist()) == [7, 20000, 39999]
    # NonZero: nothing, everything, NaN counts, bool input, one element
    for a in (np.zeros((4, 5), np.float32), np.ones((3, 2, 2), np.float32), np.array([0.0, np.nan, -0.0, 1e-38], np.float32),
              rng.standard_normal((70, 130)) > 2.5, np.array([3.0], np.float32)):
        got = nonzero(pa.asarray(a)).get()
        want = np.array(np.nonzero(a))
        assert got.dtype == np.int64 and got.shape == want.shape
        np.testing.assert_array_equal(got, want)
    # ScatterND with an empty update list returns a copy
    d = rng.standard_normal((3, 4)).astype(np.float32)
    out = scat(pa.asarray(d), np.zeros((1, 0, 2), np.int64), pa.asarray(np.zeros((1, 0), np.float32)))
    np.testing.assert_array_equal(out.get(), d)
    # LSTM, one time step, batch 1
    L, N, D, H = 1, 1, 5, 4
    args = [rng.standard_normal(s).astype(np.float32) * 0.5 for s in ((L, N, D), (1, 4 * H, D), (1, 4 * H, H), (1, 8 * H))]
    h0, c0 = rng.standard_normal((1, N, H)).astype(np.float32), rng.standard_normal((1, N, H)).astype(np.float32)
    got = lstm(*[pa.asarray(a) for a in args], np.array([1]), pa.asarray(h0), pa.asarray(c0), hidden_size=H)
    want = onp.lstm(*args, np.array([1]), h0, c0, hidden_size=H)
    for g_, w_ in zip(got, want):
        assert g_.shape == w_.shape
        assert_close(g_.get(), w_, RTOL, "lstm one step")


def test_general_broadcasting_random_pairs(pa):
    """Add / Sub / Mul / Div under numpy's broadcasting rules (layer.py:93-107) on 40 random shape pairs: ranks 0-6
    on either side, axes of extent 1 sprinkled in, against numpy itself -- bit for bit."""
    rng = np.random.default_rng(20260929)
    ops = [("add", np.add), ("sub", np.subtract), ("mul", np.multiply), ("div", np.divide)]
    for trial in range(40):
        nd = int(rng.integers(1, 7))
        full = [int(rng.integers(1, 6)) for _ in range(nd)]
        def operand():
            keep = int(rng.integers(0, nd + 1))
            shp = [d if rng.random() < 0.6 else 1 for d in full[nd - keep:]]
            return rng.standard_normal(shp).astype(np.float32)
        a, b = operand(), operand()
        kind, fn = ops[trial % 4]
        if kind == "div":
            b = np.abs(b) + 0.5
        want = fn(a, b)
        got = pa.layer_map[kind](pa.asarray(a), pa.asarray(b))
        assert tuple(got.shape) == want.shape, (kind, a.shape, b.shape)
        assert np.array_equal(got.get(), want), (kind, a.shape, b.shape)


def test_broadcast_mismatch_raises_like_numpy(pa):
    with pytest.raises(ValueError):
        pa.layer_map["add"](pa.asarray(np.zeros((2, 3), np.float32)), pa.asarray(np.zeros((4,), np.float32)))


def test_linear_upsample_and_broadcast_add_inside_a_net(pa):
    """conv -> relu -> linear upsample x2 -> add of a (1, 1, H, 1) constant -> conv, as a Net through the plan compiler
    (the two new steps stay on the NCHW kernels between channel-quad convs) against the oracle's interpreter."""
    rng = np.random.default_rng(7)
    inits = [("K1", (rng.standard_normal((8, 4, 3, 3)) * 0.2).astype(np.float32)), ("B1", rng.standard_normal(8).astype(np.float32)),
             ("k", np.array([1, 1, 2, 2], np.float32)), ("row", rng.standard_normal((1, 1, 12, 1)).astype(np.float32)),
             ("K2", (rng.standard_normal((8, 8, 3, 3)) * 0.2).astype(np.float32)), ("B2", rng.standard_normal(8).astype(np.float32))]
    cp = {"group": 1, "strides": [1, 1], "dilations": [1, 1], "pads": [1, 1, 1, 1]}
    graph = {"input": ["x"], "inits": [[n, list(a.shape), str(a.dtype)] for n, a in inits],
             "layers": [["c1", "conv", cp], ["r1", "relu", {}], ["up", "upsample", {"mode": "linear"}], ["ad", "add", {}],
                        ["c2", "conv", cp], ["return", "return", {}]],
             "flow": [[["x", "K1", "B1"], ["c1"], "a"], ["a", ["r1"], "b"], [["b", "k"], ["up"], "u"],
                      [["u", "row"], ["ad"], "s"], [["s", "K2", "B2"], ["c2"], "y"], [["y"], ["return"], "plrst"]]}
    blob = np.concatenate([a.reshape(-1).view(np.uint8) for _, a in inits])
    x = rng.standard_normal((2, 4, 6, 7)).astype(np.float32)
    ref = onp.OracleNet()
    ref.load_json(graph["input"], graph["inits"], graph["layers"], graph["flow"])
    ref.load_weights(blob)
    want = ref(x.copy())
    net = pa.from_graph(graph, blob)
    for rnd in range(2):
        got = net(x)
        assert_close(got, want, RTOL, "linear upsample net")
