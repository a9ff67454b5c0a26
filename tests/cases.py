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
    return c


def sample_index(size, count=4096, seed=12345):
    """Deterministic flat indices used to sample big whole-net outputs."""
    if size <= count:
        return np.arange(size)
    return np.sort(np.random.default_rng(seed).choice(size, count, replace=False))
