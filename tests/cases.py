"""Per-layer parity cases shared by tools/capture_golden.py (which runs them
through the imported reference) and the tests (which run them through the
oracle and the HIP path).  Inputs are regenerated from seeds; the golden
files additionally store them so drift in numpy's generators is detected.
"""
import numpy as np


def _r(seed, *shape, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def layer_cases():
    """-> list of (name, kind, [arrays], params)"""
    c = []

    def conv(name, xs, ks, bias, seed, **p):
        args = [_r(seed, *xs), _r(seed + 1, *ks, scale=0.2)]
        if bias:
            args.append(_r(seed + 2, ks[0]))
        p.setdefault("group", 1)
        p.setdefault("dilations", [1, 1])
        c.append((name, "conv", args, p))

    conv("conv_3x3_s1_p1_bias", (2, 8, 13, 17), (16, 8, 3, 3), True, 10, strides=[1, 1], pads=[1, 1, 1, 1])
    conv("conv_3x3_s1_p1_nobias", (2, 8, 13, 17), (16, 8, 3, 3), False, 20, strides=[1, 1], pads=[1, 1, 1, 1])
    conv("conv_3x3_s2_p1", (2, 16, 17, 19), (24, 16, 3, 3), False, 30, strides=[2, 2], pads=[1, 1, 1, 1])
    conv("conv_7x7_s2_p3_cin3", (2, 3, 31, 29), (16, 3, 7, 7), False, 40, strides=[2, 2], pads=[3, 3, 3, 3])
    conv("conv_1x1_s1_p0", (2, 16, 9, 11), (32, 16, 1, 1), True, 50, strides=[1, 1], pads=[0, 0, 0, 0])
    conv("conv_1x1_s2_p0", (2, 16, 14, 14), (32, 16, 1, 1), False, 60, strides=[2, 2], pads=[0, 0, 0, 0])
    conv("conv_3x3_cin3_p1_bias", (2, 3, 16, 16), (64, 3, 3, 3), True, 70, strides=[1, 1], pads=[1, 1, 1, 1])
    conv("conv_3x3_p0", (1, 4, 10, 12), (8, 4, 3, 3), True, 80, strides=[1, 1], pads=[0, 0, 0, 0])
    conv("conv_depthwise_g8", (2, 8, 11, 13), (8, 1, 3, 3), True, 90, strides=[1, 1], pads=[1, 1, 1, 1], group=8)
    conv("conv_group2", (2, 8, 9, 9), (12, 4, 3, 3), False, 100, strides=[1, 1], pads=[1, 1, 1, 1], group=2)
    conv("conv_dil2_p2", (2, 6, 15, 15), (10, 6, 3, 3), True, 110, strides=[1, 1], pads=[2, 2, 2, 2], dilations=[2, 2])
    conv("conv_3x5_s1x2", (1, 5, 12, 18), (7, 5, 3, 5), True, 120, strides=[1, 2], pads=[1, 2, 1, 2])
    conv("conv_wide_k", (3, 96, 7, 7), (40, 96, 3, 3), False, 130, strides=[1, 1], pads=[1, 1, 1, 1])
    conv("conv_b1_7x7img", (1, 32, 7, 7), (48, 32, 3, 3), True, 140, strides=[1, 1], pads=[1, 1, 1, 1])

    c.append(("dense", "dense", [_r(200, 5, 96), _r(201, 40, 96, scale=0.1), _r(202, 40)], {"shp": [96, 40]}))
    c.append(("dense_b1", "dense", [_r(203, 1, 64), _r(204, 10, 64, scale=0.1), _r(205, 10)], {}))
    c.append(("matmul", "matmul", [_r(210, 7, 33), _r(211, 33, 21)], {}))
    c.append(("batchnorm", "batchnorm", [_r(220, 2, 12, 7, 9), _r(221, 1, 12, 1, 1), _r(222, 1, 12, 1, 1)], {}))
    c.append(("relu", "relu", [_r(230, 2, 5, 9, 7)], {}))
    c.append(("leakyrelu_0.1", "leakyrelu", [_r(240, 2, 5, 9, 7)], {"alpha": 0.1}))
    c.append(("leakyrelu_default", "leakyrelu", [_r(241, 3, 33)], {}))
    sg = _r(250, 2, 3, 8, 8, scale=4.0)
    sg.flat[:8] = [88.0, -88.0, 100.0, -100.0, 0.0, -0.0, 87.0, -87.0]
    c.append(("sigmoid", "sigmoid", [sg], {}))
    c.append(("add", "add", [_r(260, 2, 6, 5, 7), _r(261, 2, 6, 5, 7)], {}))
    c.append(("add_bcast_channel", "add", [_r(262, 2, 6, 5, 7), _r(263, 1, 6, 1, 1)], {}))
    c.append(("maxpool_k2s2", "maxpool", [_r(270, 2, 4, 12, 14)], {"w": [2, 2], "pads": [0, 0, 0, 0], "strides": [2, 2]}))
    c.append(("maxpool_k3s2p1_neg", "maxpool", [-np.abs(_r(271, 2, 4, 13, 15)) - 0.5],
              {"w": [3, 3], "pads": [1, 1, 1, 1], "strides": [2, 2]}))
    c.append(("maxpool_k3s2p1", "maxpool", [_r(272, 2, 4, 14, 14)], {"w": [3, 3], "pads": [1, 1, 1, 1], "strides": [2, 2]}))
    big = _r(273, 1, 2, 6, 6)
    big.flat[:3] = [-2e4, -3e4, -1.5e4]
    c.append(("maxpool_clamp_-1e4", "maxpool", [np.minimum(big, -1.2e4)], {"w": [2, 2], "pads": [0, 0, 0, 0], "strides": [2, 2]}))
    c.append(("averagepool_k3s2p1", "averagepool", [_r(280, 2, 4, 13, 15)], {"w": [3, 3], "pads": [1, 1, 1, 1], "strides": [2, 2]}))
    c.append(("upsample_x2", "upsample", [_r(290, 2, 5, 6, 7), np.array([1, 1, 2, 2], np.float32)], {"mode": "nearest"}))
    c.append(("upsample_2x3", "upsample", [_r(291, 1, 3, 5, 4), np.array([1, 1, 2, 3], np.float32)], {"mode": "nearest"}))
    c.append(("concat_axis1", "concat", [_r(300, 2, 3, 5, 7), _r(301, 2, 6, 5, 7)], {"axis": 1}))
    c.append(("concat_axis0_3", "concat", [_r(302, 2, 4), _r(303, 1, 4), _r(304, 3, 4)], {"axis": 0}))
    c.append(("gap", "gap", [_r(310, 2, 9, 7, 7)], {}))
    c.append(("gap_big", "gap", [_r(311, 1, 3, 28, 31)], {}))
    c.append(("flatten", "flatten", [_r(320, 2, 4, 3, 5)], {}))
    # ---- second-wave operators ----
    pos = np.abs(_r(400, 2, 5, 6, 7)) + 0.1
    for kind, arr in (("exp", _r(401, 2, 5, 6, 7)), ("log", pos), ("tanh", _r(402, 3, 50, scale=2)),
                      ("sqrt", pos), ("reciprocal", pos)):
        c.append((kind, kind, [arr], {}))
    c.append(("hardsigmoid", "hardsigmoid", [_r(403, 2, 5, 6, 7, scale=3)], {}))
    c.append(("hardsigmoid_ab", "hardsigmoid", [_r(404, 4, 33, scale=3)], {"alpha": 0.1666, "beta": 0.5}))
    c.append(("clip", "clip", [_r(405, 2, 5, 6, 7, scale=2)], {"min": -0.5, "max": 1.25}))
    c.append(("clip_relu6", "clip", [_r(406, 3, 40, scale=5)], {"min": 0, "max": 6}))
    a4, b4 = _r(410, 2, 6, 5, 7), _r(411, 2, 6, 5, 7)
    for kind in ("sub", "mul", "div"):
        c.append((kind, kind, [a4, np.where(np.abs(b4) < 0.1, 0.5, b4).astype(np.float32)], {}))
        c.append((kind + "_bcast_channel", kind, [a4, (np.abs(_r(412, 1, 6, 1, 1)) + 0.5).astype(np.float32)], {}))
        c.append((kind + "_scalar_lhs", kind, [np.array([1.5], np.float32), (np.abs(b4) + 0.5).astype(np.float32)], {}))
    c.append(("pow_scalar", "pow", [pos, np.array([2.0], np.float32)], {}))
    c.append(("pow_tensor", "pow", [pos, (np.abs(_r(413, 2, 5, 6, 7)) + 0.2).astype(np.float32)], {}))
    c.append(("add_scalar", "add", [a4, np.array([0.25], np.float32)], {}))
    c.append(("softmax", "softmax", [_r(420, 6, 1000, scale=3)], {}))
    c.append(("softmax_3d", "softmax", [_r(421, 2, 7, 33, scale=2)], {"axis": -1}))
    c.append(("logsoftmax", "logsoftmax", [_r(422, 5, 91, scale=3)], {"axis": 1}))
    c.append(("reducemean_hw", "reducemean", [_r(430, 2, 9, 7, 5)], {"axes": [2, 3], "keepdims": True}))
    c.append(("reducesum_last", "reducesum", [_r(431, 4, 77)], {"axes": [-1], "keepdims": False}))
    c.append(("reducemax_hw", "reducemax", [_r(432, 2, 3, 8, 9)], {"axes": [-2, -1], "keepdims": True}))
    c.append(("reducemin_last", "reducemin", [_r(433, 3, 5, 40)], {"axes": [2], "keepdims": True}))
    c.append(("transpose_0231", "transpose", [_r(440, 2, 5, 6, 7)], {"axis": [0, 2, 3, 1]}))
    c.append(("transpose_10", "transpose", [_r(441, 13, 29)], {"axis": [1, 0]}))
    c.append(("reshape_keep0", "reshape", [_r(450, 2, 12, 5), np.array([0, 3, -1], np.int64)], {}))
    c.append(("squeeze", "squeeze", [_r(451, 3, 1, 5)], {"axes": [1]}))
    c.append(("unsqueeze", "unsqueeze", [_r(452, 3, 5)], {"axes": [0, 3]}))
    c.append(("resize_nearest_x2", "resize", [_r(460, 2, 3, 5, 6), np.zeros(0, np.float32),
                                              np.array([1, 1, 2, 2], np.float32)], {"mode": "nearest"}))
    c.append(("resize_asym_floor", "resize", [_r(461, 1, 2, 4, 4), np.zeros(0, np.float32),
                                              np.array([1, 1, 3, 2], np.float32)],
              {"mode": "nearest", "coordinate_transformation_mode": "asymmetric", "nearest_mode": "floor"}))
    # nearest with a non-zero shift (util.offset / util.pix_offset, util.py:155-192): every (transform, rounding) pair whose
    # offset is not 0, both signs, one axis alone and both together (the vacated corner rule), truncated factors
    rz = lambda name, seed, shp, kk, tm, rm: c.append((name, "resize", [_r(seed, *shp), np.zeros(0, np.float32),
                                                        np.array([1, 1] + list(kk), np.float32)],
                                                       {"mode": "nearest", "coordinate_transformation_mode": tm, "nearest_mode": rm}))
    rz("resize_asym_ceil", 462, (2, 3, 4, 5), (3, 2), "asymmetric", "ceil")
    rz("resize_asym_prefer_ceil", 463, (1, 2, 5, 4), (2, 4), "asymmetric", "round_prefer_ceil")
    rz("resize_asym_prefer_floor", 464, (1, 2, 3, 6), (3, 5), "asymmetric", "round_prefer_floor")
    rz("resize_half_floor", 465, (2, 2, 4, 3), (2, 4), "half_pixel", "floor")
    rz("resize_half_ceil", 466, (1, 3, 5, 5), (3, 3), "half_pixel", "ceil")
    rz("resize_half_floor_rows_only", 467, (1, 2, 4, 6), (3, 1), "half_pixel", "floor")
    rz("resize_asym_ceil_cols_only", 468, (1, 2, 4, 6), (1, 3), "asymmetric", "ceil")
    rz("resize_unknown_modes", 469, (1, 2, 3, 4), (2, 2), "align_corners", "ceil")
    rz("resize_nearest_truncated", 459, (1, 2, 4, 5), (2.7, 3.2), "asymmetric", "ceil")
    # ---- general numpy broadcasting (layer.py:93-111) and linear up-sampling (util.py:121-153, 194-219) ----
    c.append(("add_bcast_rows", "add", [_r(470, 2, 3, 4, 5), _r(471, 4, 1)], {}))
    c.append(("sub_bcast_outer", "sub", [_r(472, 3, 1), _r(473, 1, 4)], {}))
    c.append(("mul_bcast_cross", "mul", [_r(474, 2, 1, 4, 1), _r(475, 1, 3, 1, 5)], {}))
    c.append(("div_bcast_lower_rank_lhs", "div", [_r(476, 5), np.abs(_r(477, 2, 3, 4, 5)) + 0.5], {}))
    c.append(("add_bcast_batch", "add", [_r(478, 1, 6, 5, 7), _r(479, 3, 1, 5, 1)], {}))
    c.append(("mul_bcast_spatial", "mul", [_r(480, 2, 6, 5, 7), _r(481, 2, 1, 5, 7)], {}))
    c.append(("pow_bcast_rows", "pow", [np.abs(_r(482, 2, 3, 4)) + 0.1, _r(483, 3, 1)], {}))
    c.append(("upsample_linear_x2", "upsample", [_r(484, 2, 3, 5, 6), np.array([1, 1, 2, 2], np.float32)], {"mode": "linear"}))
    c.append(("upsample_linear_3x2", "upsample", [_r(485, 1, 2, 4, 7), np.array([1, 1, 3, 2], np.float32)], {"mode": "linear"}))
    c.append(("upsample_linear_1x2", "upsample", [_r(486, 1, 2, 4, 5), np.array([1, 1, 1, 2], np.float32)], {"mode": "linear"}))
    c.append(("upsample_linear_4x1", "upsample", [_r(487, 2, 2, 3, 5), np.array([1, 1, 4, 1], np.float32)], {"mode": "linear"}))
    c.append(("upsample_linear_trunc", "upsample", [_r(488, 1, 2, 4, 5), np.array([1, 1, 2.7, 2.2], np.float32)], {"mode": "linear"}))
    c.append(("resize_linear_x2", "resize", [_r(489, 2, 3, 5, 6), np.zeros(0, np.float32),
                                             np.array([1, 1, 2, 2], np.float32)], {"mode": "linear"}))
    c.append(("resize_linear_frac", "resize", [_r(490, 2, 3, 6, 8), np.zeros(0, np.float32),
                                               np.array([1, 1, 1.5, 2.25], np.float32)], {"mode": "linear"}))
    c.append(("resize_linear_down", "resize", [_r(491, 1, 2, 9, 11), np.zeros(0, np.float32),
                                               np.array([1, 1, 0.5, 0.7], np.float32)], {"mode": "linear"}))
    c.append(("resize_linear_size", "resize", [_r(492, 1, 3, 5, 7), np.zeros(0, np.float32), np.zeros(0, np.float32),
                                               np.array([1, 3, 13, 10], np.int64)], {"mode": "linear"}))
    # ---- structural ops on the strided-map kernel + ConvTranspose2d ----
    i64 = lambda *v: np.array(v, np.int64)
    c.append(("slice_basic", "slice", [_r(500, 2, 6, 9, 11), i64(1, 2), i64(5, 9), i64(1, 3), i64(1, 1)], {}))
    c.append(("slice_step_neg", "slice", [_r(501, 3, 8, 10), i64(0, -2, 1), i64(3, 0, 100), i64(0, 1, 2), i64(2, -3, 4)], {}))
    c.append(("slice_default_axes", "slice", [_r(502, 5, 7), i64(1, 2), i64(-1, 6)], {}))
    c.append(("pad_hw", "pad", [_r(510, 2, 3, 5, 6), i64(0, 0, 1, 2, 0, 0, 3, 0)], {}))
    c.append(("pad_value", "pad", [_r(511, 3, 4), i64(2, 1, 0, 3)], {"constant_value": -1.5}))
    # np.pad's index-map modes (layer.py:241-245 hands `mode` through): borders wider than the axis fold back more than once
    c.append(("pad_reflect_hw", "pad", [_r(512, 2, 3, 5, 6), i64(0, 0, 2, 1, 0, 0, 1, 3)], {"mode": "reflect"}))
    c.append(("pad_edge_hw", "pad", [_r(513, 2, 3, 5, 6), i64(0, 1, 2, 1, 0, 0, 3, 2)], {"mode": "edge"}))
    c.append(("pad_symmetric", "pad", [_r(514, 3, 4, 5), i64(1, 2, 0, 0, 3, 6)], {"mode": "symmetric"}))
    c.append(("pad_wrap", "pad", [_r(515, 4, 5), i64(3, 7, 5, 2)], {"mode": "wrap"}))
    c.append(("pad_reflect_wide", "pad", [_r(516, 3, 4), i64(7, 9, 8, 10)], {"mode": "reflect"}))
    c.append(("pad_reflect_len1", "pad", [_r(517, 1, 4), i64(2, 1, 3, 0)], {"mode": "reflect"}))
    c.append(("tile_2d", "tile", [_r(520, 3, 5), i64(2, 3)], {}))
    c.append(("tile_more_reps", "tile", [_r(521, 2, 3), i64(2, 1, 2)], {}))
    c.append(("expand_channel", "expand", [_r(530, 1, 4, 1, 1), i64(2, 4, 3, 5)], {}))
    c.append(("expand_lower_rank", "expand", [_r(531, 5), i64(3, 1, 5)], {}))
    c.append(("split_axis1", "split", [_r(540, 9, 7, 4)], {"split": [2, 4, 1], "axis": 1}))
    c.append(("split_axis0", "split", [_r(541, 6, 5)], {"split": [1, 3], "axis": 0}))
    c.append(("convtranspose_k2s2", "convtranspose", [_r(550, 2, 8, 7, 9), _r(551, 8, 6, 2, 2, scale=0.2)], {}))
    c.append(("convtranspose_k3s2p1_op1_bias", "convtranspose",
              [_r(552, 1, 16, 6, 5), _r(553, 16, 12, 3, 3, scale=0.1), _r(554, 12)],
              {"strides": [2, 2], "pads": [1, 1, 1, 1], "output_padding": [1, 1]}))
    c.append(("convtranspose_k4s2p1", "convtranspose", [_r(555, 2, 4, 5, 5), _r(556, 4, 3, 4, 4, scale=0.2)],
              {"strides": [2, 2], "pads": [1, 1, 1, 1]}))
    c.append(("convtranspose_s1_dil2", "convtranspose", [_r(557, 1, 3, 8, 8), _r(558, 3, 5, 3, 3, scale=0.2), _r(559, 5)],
              {"strides": [1, 1], "dilations": [2, 2]}))
    # ---- lifted restrictions: softmax / reductions over any axis, stacked matmul ----
    c.append(("softmax_axis1_4d", "softmax", [_r(600, 2, 7, 5, 6, scale=2)], {"axis": 1}))
    c.append(("logsoftmax_axis0", "logsoftmax", [_r(601, 9, 14, scale=2)], {"axis": 0}))
    c.append(("reducesum_axis1", "reducesum", [_r(602, 3, 8, 5, 4)], {"axes": [1], "keepdims": True}))
    c.append(("reducemean_axes02", "reducemean", [_r(603, 4, 6, 10)], {"axes": [0, 2], "keepdims": False}))
    c.append(("reducemax_axis0", "reducemax", [_r(604, 5, 33)], {"axes": [0], "keepdims": False}))
    c.append(("matmul_stack_2d", "matmul", [_r(610, 2, 3, 5, 17), _r(611, 17, 9)], {}))
    c.append(("matmul_stack_stack", "matmul", [_r(612, 2, 3, 6, 20), _r(613, 2, 3, 20, 7)], {}))
    # ---- operators of ONNX-exported detection heads ----
    c.append(("shape", "shape", [_r(620, 2, 3, 5, 7)], {}))
    c.append(("gather_axis0", "gather", [_r(621, 6, 5, 4), i64(4, 0, 4)], {}))
    c.append(("gather_axis2_neg_2d_idx", "gather", [_r(622, 3, 4, 9), np.array([[0, -1], [3, 8]], np.int64)], {"axis": 2}))
    c.append(("gather_shape_scalar", "gather", [i64(2, 255, 13, 13), np.array(1, np.int64)], {"axis": 0}))
    c.append(("cast_f32_i64", "cast", [_r(623, 4, 9, scale=5)], {"dtype": "int64"}))
    c.append(("cast_i64_f32", "cast", [i64(3, -7, 100000)], {"dtype": "float32"}))
    c.append(("cast_f32_bool", "cast", [np.array([0.0, -0.0, 1.5, -2.0, 1e-30], np.float32)], {"dtype": "bool"}))
    c.append(("range", "range", [np.array(2, np.int64), np.array(17, np.int64), np.array(4, np.int64)], {}))
    q = np.round(_r(624, 3, 40) * 2).astype(np.float32)
    c.append(("equal", "equal", [q, np.round(_r(625, 3, 40) * 2).astype(np.float32)], {}))
    c.append(("greater_scalar", "greater", [q, np.array([0.5], np.float32)], {}))
    c.append(("greaterorequal", "greaterorequal", [q, np.round(_r(626, 3, 40) * 2).astype(np.float32)], {}))
    c.append(("equal_shape_tensors", "equal", [i64(2, 3, 4), i64(2, 5, 4)], {}))
    c.append(("where", "where", [_r(627, 5, 33) > 0.2, _r(628, 5, 33), _r(629, 5, 33)], {}))
    c.append(("where_scalar_rhs", "where", [_r(630, 2, 3, 8) > 0, _r(631, 2, 3, 8), np.array([-1.0], np.float32)], {}))
    c.append(("constantofshape_f32", "constantofshape", [i64(2, 3, 4)], {"value": 1.5, "dtype": "float32"}))
    c.append(("constantofshape_i64", "constantofshape", [i64(5)], {"value": 0, "dtype": "int64"}))
    c.append(("erf", "erf", [_r(632, 4, 50, scale=1.5)], {}))
    c.append(("instancenorm", "instancenormalization", [_r(633, 2, 6, 7, 9), (np.abs(_r(634, 6)) + 0.5).astype(np.float32),
                                                       _r(635, 6)], {"epsilon": 1e-5}))
    c.append(("concat_shape_tensors", "concat", [i64(2), i64(255), i64(13, 13)], {"axis": 0}))
    c.append(("mul_shape_tensors", "mul", [i64(13, 13), i64(2, 2)], {}))
    # ---- sorting / data-dependent shapes / recurrence ----
    sc = np.array([[[1, 2], [3, 0], [1, 2], [-1, -2]]], np.int64)          # row (1, 2) written twice: the last wins
    c.append(("scatternd_rows", "scatternd", [_r(640, 4, 5, 6), sc, _r(641, 1, 4, 6)], {}))
    c.append(("scatternd_elements", "scatternd", [_r(642, 3, 7), np.array([[[0, 0], [2, 6], [1, 3], [2, 6], [0, -1]]], np.int64),
                                                  _r(643, 1, 5)], {}))
    nz = _r(644, 3, 4, 50)
    nz[np.abs(nz) < 0.8] = 0
    c.append(("nonzero_f32", "nonzero", [nz], {}))
    c.append(("nonzero_bool", "nonzero", [_r(645, 2, 2100) > 1.5], {}))
    c.append(("nonzero_none", "nonzero", [np.zeros((2, 3, 4), np.float32)], {}))
    c.append(("nonzero_i64_1d", "nonzero", [np.array([0, 3, 0, 0, -1, 7], np.int64)], {}))
    c.append(("topk_last", "topk", [_r(646, 4, 50), np.array([5], np.int64)], {}))
    c.append(("topk_axis1", "topk", [_r(647, 3, 20, 7), np.array([3], np.int64)], {"axis": 1}))
    c.append(("topk_smallest_quirk", "topk", [_r(648, 3, 33), np.array([4], np.int64)], {"largest": 0}))
    c.append(("topk_yolo_candidates", "topk", [_r(649, 2, 10647), np.array([100], np.int64)], {"axis": -1, "largest": 1, "sorted": 1}))
    c.append(("topk_long_row", "topk", [_r(650, 20000), np.array([6], np.int64)], {}))
    c.append(("topk_long_row_smallest", "topk", [_r(651, 17000), np.array([3], np.int64)], {"largest": 0}))
    L_, N_, D_, H_ = 5, 3, 6, 8

    def lstm_args(seed, dirs):
        return [_r(seed, L_, N_, D_), _r(seed + 1, dirs, 4 * H_, D_, scale=0.4), _r(seed + 2, dirs, 4 * H_, H_, scale=0.4),
                _r(seed + 3, dirs, 8 * H_, scale=0.2), np.full((N_,), L_, np.int64), _r(seed + 4, dirs, N_, H_, scale=0.5),
                _r(seed + 5, dirs, N_, H_, scale=0.5)]
    c.append(("lstm_forward", "lstm", lstm_args(660, 1), {"hidden_size": H_, "direction": "forward"}))
    c.append(("lstm_reverse", "lstm", lstm_args(670, 1), {"hidden_size": H_, "direction": "reverse"}))
    c.append(("lstm_bidirectional", "lstm", lstm_args(680, 2), {"hidden_size": H_, "direction": "bidirectional"}))
    return c


def sample_index(size, count=4096, seed=12345):
    """Deterministic flat indices used to sample big whole-net outputs."""
    if size <= count:
        return np.arange(size)
    return np.sort(np.random.default_rng(seed).choice(size, count, replace=False))


def tile_cases():
    """(name, image, K, B, upsample factor of f, tile kwargs): scenarios for util.tile (util.py:291-348).
    f(window) = relu(conv3x3(window) + B) [-> nearest upsample], returned as H x W x Cout."""
    return [
        # many windows, no resampling, float margin
        ("tile_gray_multi", _r(600, 150, 170), _r(601, 3, 1, 3, 3, scale=0.3), _r(602, 3), 1,
         dict(sample=1, window=64, margin=0.1)),
        # resample up by 1.5, f doubles the resolution (k = 2), integer margin, colour image
        ("tile_rgb_resample_k2", _r(610, 90, 110, 3), _r(611, 2, 3, 3, 3, scale=0.3), _r(612, 2), 2,
         dict(sample=1.5, window=64, margin=8)),
        # smaller than the window: grown to a multiple of glob, one window, result resized back
        ("tile_small_glob", _r(620, 40, 50), _r(621, 2, 1, 3, 3, scale=0.3), _r(622, 2), 1,
         dict(sample=1, window=64, glob=16, margin=0.1)),
        # explicit target size
        ("tile_size_tuple", _r(630, 70, 61, 2), _r(631, 2, 2, 3, 3, scale=0.3), _r(632, 2), 1,
         dict(sample=(100, 96), window=48, margin=0.2)),
    ]
