"""Channel-quad ("Q4") tensors: the compiled plan's internal activation layout.

A Q4 tensor holds the reference's (N, C, H, W) array as a DeviceArray of shape
(N, ceil(C/4), H, W, 4) -- channel c sits in quad c // 4, lane c % 4, padding
lanes are zero -- with `.chan = C`.  Nothing here is a reference op: the plan
compiler (plan.assign_layouts) rewrites runs of layers that have Q4 kernels
to the `*_q4` kinds below and puts `to_q4` / `from_q4` at the edges, so what
`Net.__call__` takes and returns is NCHW exactly as in net.py:94-101.

Why: include/planer_hip.h ("channel-quad activations") and DESIGN.md section 4 --
one 16-byte load per (pixel, 4 channels) instead of four 4-byte loads keeps the
fp32 MFMA pipe 13-17 % busier, and every HBM-bound layer moves float4s.
"""
import ctypes

import numpy

from . import _lib
from .hip import DeviceArray, empty
from .layer import ACT_NONE, _f32, _host_values, _ptr, conv_out_hw


def is_q4(a):
    return isinstance(a, DeviceArray) and a.chan is not None


def _new_q4(n, c, h, w, ctx):
    y = empty((n, (c + 3) // 4, h, w, 4), ctx=ctx)
    y.chan = c
    return y


def logical_shape(a):
    n, _, h, w, _ = a.shape
    return (n, a.chan, h, w)


def to_q4(x):
    """NCHW -> Q4 (one HBM pass)."""
    _f32(x)
    if is_q4(x):
        return x
    n, c, h, w = x.shape
    y = _new_q4(n, c, h, w, x.ctx)
    if y.size:
        _lib.call("pl_nchw_to_q4_f32", x.ctx.handle, x.ptr, y.ptr, n, c, h * w)
    return y


def from_q4(xq):
    """Q4 -> NCHW."""
    if not is_q4(xq):
        return xq
    n, c, h, w = logical_shape(xq)
    y = empty((n, c, h, w), ctx=xq.ctx)
    if y.size:
        _lib.call("pl_q4_to_nchw_f32", xq.ctx.handle, xq.ptr, y.ptr, n, c, h * w)
    return y


def q4_conv_eligible(k_shape, group=1, **_):
    cout, cin_g = k_shape[0], k_shape[1]
    return len(k_shape) == 4 and (group == 1 or (cin_g % 4 == 0 and (cout // group) % 4 == 0))


def prepare_q4_weights(K, group=1):
    """OIHW filters -> wq[group][tap*ceil(Cin_g/4) + cin/4][Cout/group][4] (zero padded), made once
    per model.  The returned array keeps the logical OIHW shape; its allocation is the packed size."""
    _f32(K)
    cout, cin_g, kh, kw = K.shape
    n = ctypes.c_size_t()
    _lib.call("pl_conv2d_q4_filter_elems", cout, cin_g, kh, kw, int(group), ctypes.byref(n))
    out = empty((n.value,), ctx=K.ctx)
    _lib.call("pl_conv2d_prepare_q4_f32", K.ctx.handle, K.ptr, cout, cin_g, kh, kw, int(group), out.ptr)
    out.shape = K.shape
    return out


def winograd_q4_eligible(k_shape, group=1, strides=(1, 1), dilations=(1, 1), pads=(0, 0, 0, 0), **_):
    """3x3 / stride 1 / pad 1 / no dilation / no groups, Cin and Cout multiples of 4."""
    cout, cin_g, kh, kw = k_shape
    return (kh == 3 and kw == 3 and group == 1 and cin_g % 4 == 0 and cout % 4 == 0 and list(strides) == [1, 1]
            and list(dilations) == [1, 1] and list(pads) == [1, 1, 1, 1])


def prepare_winograd_q4_weights(K):
    """OIHW 3x3 filters -> Winograd-domain Q4 filters [16][k-quad][Cout][4] (ConvQ4 w_layout=4)."""
    _f32(K)
    cout, cin, kh, kw = K.shape
    if (kh, kw) != (3, 3) or cin % 4 or cout % 4:
        raise ValueError("winograd Q4 filters need 3x3 kernels, Cin % 4 == 0 and Cout % 4 == 0")
    n = ctypes.c_size_t()
    _lib.call("pl_conv2d_winograd_q4_filter_elems", cout, cin, ctypes.byref(n))
    out = empty((n.value,), ctx=K.ctx)
    _lib.call("pl_conv2d_prepare_winograd_q4_f32", K.ctx.handle, K.ptr, cout, cin, out.ptr)
    out.shape = K.shape
    return out


def prepare_winograd4_q4_weights(K):
    """OIHW 3x3 filters -> Winograd F(4x4,3x3) Q4 filters [36][k-quad][Cout][4] (ConvQ4 w_layout=7)."""
    _f32(K)
    cout, cin, kh, kw = K.shape
    if (kh, kw) != (3, 3) or cin % 4 or cout % 4:
        raise ValueError("winograd Q4 filters need 3x3 kernels, Cin % 4 == 0 and Cout % 4 == 0")
    n = ctypes.c_size_t()
    _lib.call("pl_conv2d_winograd4_q4_filter_elems", cout, cin, ctypes.byref(n))
    out = empty((n.value,), ctx=K.ctx)
    _lib.call("pl_conv2d_prepare_winograd4_q4_f32", K.ctx.handle, K.ptr, cout, cin, out.ptr)
    out.shape = K.shape
    return out


def winograd43_eligible(x_shape, k_shape, min_columns=0, **para):
    """Mixed-tile Winograd (csrc/wino43_kernels.h): a 3x3 / stride 1 / pad 1 / group 1 conv on a map whose sides are 7, 14 or 21.
    `min_columns`: the plan compiler only offers it where each of the 121 per-frequency GEMMs has that many tile columns
    (N * (H / 7) * (W / 7)): its filters are 3.4x those of F(4x4,3x3), and with few columns per filter the GEMM lives on filter
    bandwidth -- ResNet-18's layer4 at batch 32 (32 columns, 127 MB of filters per conv) wins 4 us per conv in isolation and
    loses 2 % of the pipelined rate, layer3 (128 columns) wins both ways."""
    return (len(x_shape) == 4 and x_shape[2] in (7, 14, 21) and x_shape[3] in (7, 14, 21) and winograd_q4_eligible(k_shape, **para)
            and x_shape[0] * (x_shape[2] // 7) * (x_shape[3] // 7) >= min_columns)


def prepare_winograd43_q4_weights(K):
    """OIHW 3x3 filters -> mixed-tile Winograd filters [121][k-quad][Cout][4] (ConvQ4 w_layout=11)."""
    _f32(K)
    cout, cin, kh, kw = K.shape
    if (kh, kw) != (3, 3) or cin % 4 or cout % 4:
        raise ValueError("winograd Q4 filters need 3x3 kernels, Cin % 4 == 0 and Cout % 4 == 0")
    n = ctypes.c_size_t()
    _lib.call("pl_conv2d_winograd43_q4_filter_elems", cout, cin, ctypes.byref(n))
    out = empty((n.value,), ctx=K.ctx)
    _lib.call("pl_conv2d_prepare_winograd43_q4_f32", K.ctx.handle, K.ptr, cout, cin, out.ptr)
    out.shape = K.shape
    return out


def prepare_wf4_q4_weights(K):
    """OIHW 3x3 filters -> fully fused F(4x4,3x3) filters [Cout/64][Cin/4][36][4][4][16] (ConvQ4 w_layout=9)."""
    _f32(K)
    cout, cin, kh, kw = K.shape
    if (kh, kw) != (3, 3) or cin % 4 or cout % 4:
        raise ValueError("winograd Q4 filters need 3x3 kernels, Cin % 4 == 0 and Cout % 4 == 0")
    n = ctypes.c_size_t()
    _lib.call("pl_conv2d_wf4_filter_elems", cout, cin, ctypes.byref(n))
    out = empty((n.value,), ctx=K.ctx)
    _lib.call("pl_conv2d_prepare_wf4_f32", K.ctx.handle, K.ptr, cout, cin, out.ptr)
    out.shape = K.shape
    return out


def rowpack_eligible(k_shape, group=1, strides=(1, 1), dilations=(1, 1), pads=(0, 0, 0, 0), **_):
    """Convs on 1..3 input channels (the stem): group 1, no dilation, symmetric pads."""
    cout, cin_g, kh, kw = k_shape
    pads = list(pads)
    return (group == 1 and cin_g < 4 and list(dilations) == [1, 1] and len(pads) == 4
            and pads[0] == pads[2] and pads[1] == pads[3])


def prepare_rowpack_weights(K):
    """OIHW filters with Cin < 4 -> row-packed [kh*ceil(kw*Cin/4)][Cout][4] (ConvQ4 w_layout=6)."""
    _f32(K)
    cout, cin, kh, kw = K.shape
    n = ctypes.c_size_t()
    _lib.call("pl_conv2d_rowpack_filter_elems", cout, cin, kh, kw, ctypes.byref(n))
    out = empty((n.value,), ctx=K.ctx)
    _lib.call("pl_conv2d_prepare_rowpack_f32", K.ctx.handle, K.ptr, cout, cin, kh, kw, out.ptr)
    out.shape = K.shape
    return out


def prepare_stem_nchw_weights(K):
    """OIHW stem filters [Cout][3][7][7] -> [48][Cout][4] in the k order of the stem + max-pool kernel that reads the NCHW
    input itself (ConvPoolQ4 w_layout=12, csrc/conv_stem_pool_kernel.h)."""
    _f32(K)
    cout, cin, kh, kw = K.shape
    if (cin, kh, kw) != (3, 7, 7):
        raise ValueError("the NCHW stem kernel takes [Cout][3][7][7] filters")
    n = ctypes.c_size_t()
    _lib.call("pl_conv2d_stem_nchw_filter_elems", cout, ctypes.byref(n))
    out = empty((n.value,), ctx=K.ctx)
    _lib.call("pl_conv2d_prepare_stem_nchw_f32", K.ctx.handle, K.ptr, cout, out.ptr)
    out.shape = K.shape
    return out


def prepare_w1d4_q4_weights(K):
    """OIHW 3x3 filters -> fused 1-D Winograd F(4,3) filters [6][row*Cin/4 + cin/4][Cout][4] (w_layout=8)."""
    _f32(K)
    cout, cin, kh, kw = K.shape
    if (kh, kw) != (3, 3) or cin % 4:
        raise ValueError("1-D winograd filters need 3x3 kernels and Cin % 4 == 0")
    n = ctypes.c_size_t()
    _lib.call("pl_conv2d_w1d4_q4_filter_elems", cout, cin, ctypes.byref(n))
    out = empty((n.value,), ctx=K.ctx)
    _lib.call("pl_conv2d_prepare_w1d4_q4_f32", K.ctx.handle, K.ptr, cout, cin, out.ptr)
    out.shape = K.shape
    return out


def w1d_q4_eligible(k_shape, group=1, strides=(1, 1), dilations=(1, 1), pads=(0, 0, 0, 0), **_):
    cout, cin_g, kh, kw = k_shape
    return (kh == 3 and kw == 3 and group == 1 and cin_g % 4 == 0 and list(strides) == [1, 1]
            and list(dilations) == [1, 1] and list(pads) == [1, 1, 1, 1])


def pack_rows(x, geom=None, src_ptr=None, ctx=None):
    """The row-packed (zero-padded NHWC) image the stem kernel reads (pl_rowpack_input_f32), kept BESIDE a plan's static
    NCHW input `x` as `x.packed = (geom, image)`, geom = (kw, stride_w, pad_top, pad_left).  First call (geom given):
    allocates the image and fills it from `x`.  Later calls re-fill it from `src_ptr` -- the batch a caller feeds the
    plan -- on `ctx`'s stream: the re-layout then IS the copy into the plan, the NCHW tensor is not written."""
    n, c, h, w = x.shape
    if x.packed is None:
        elems = ctypes.c_size_t()
        _lib.call("pl_rowpack_input_elems", n, c, h, w, geom[0], geom[1], geom[2], geom[3], ctypes.byref(elems))
        x.packed = (tuple(geom), empty((int(elems.value),), ctx=x.ctx))
    g, img = x.packed
    cx = ctx or x.ctx
    _lib.call("pl_rowpack_input_f32", cx.handle, x.ptr if src_ptr is None else src_ptr, img.ptr, n, c, h, w, g[0], g[1], g[2], g[3])
    return img


def ConvQ4(xq, Kq, B=None, scale=None, shift=None, resq=None, group=1, strides=(1, 1),
           dilations=(1, 1), pads=(0, 0, 0, 0), act=ACT_NONE, alpha=0.0, w_layout=2, **_):
    """layer.ConvFused on Q4 tensors: act((conv(x,K)+B)*scale + shift + res), all activations Q4.
    w_layout=2: Kq from prepare_q4_weights(); w_layout=4: Winograd filters from
    prepare_winograd_q4_weights(); 6 row-packed stem, 7 staged / 9 fused F(4x4,3x3), 8 fused 1-D F(4,3)."""
    _f32(xq, Kq, B, scale, shift, resq)
    if w_layout == 6:
        # row-packed stem: the input is the reference's NCHW tensor, the output is Q4
        if is_q4(xq) or (resq is not None and not is_q4(resq)):
            raise TypeError("row-packed ConvQ4 takes an NCHW input (and a Q4 residual)")
        if not rowpack_eligible(Kq.shape, group, strides, dilations, pads):
            raise ValueError("row-packed filters serve group 1 / dilation 1 / Cin < 4 convs only")
        n, cin, h, w = xq.shape
        cout, cin_g, kh, kw = Kq.shape
        if cin_g != cin:
            raise ValueError("conv: weight %s does not match input %s" % (Kq.shape, xq.shape))
        pads, strides = [int(p) for p in pads], [int(s) for s in strides]
        ho, wo = conv_out_hw(h, w, kh, kw, strides, [1, 1], pads)
        y = _new_q4(n, cout, ho, wo, xq.ctx)
        if resq is not None and resq.shape != y.shape:
            raise ValueError("fused residual shape %s != conv output %s" % (resq.shape, y.shape))
        if xq.packed is not None and xq.packed[0] == (kw, strides[1], pads[0], pads[1]):
            # a plan's static input whose row-packed image is kept up to date by whoever feeds the plan (pack_rows)
            _lib.call("pl_conv2d_rowpacked_q4_f32", xq.ctx.handle, xq.packed[1].ptr, n, cin, h, w, Kq.ptr, cout, kh, kw, _ptr(B),
                      y.ptr, strides[0], strides[1], pads[0], pads[1], _ptr(scale), _ptr(shift), _ptr(resq), int(act), float(alpha))
            return y
        _lib.call("pl_conv2d_rowpack_q4_f32", xq.ctx.handle, xq.ptr, n, cin, h, w, Kq.ptr, cout, kh, kw, _ptr(B), y.ptr,
                  strides[0], strides[1], pads[0], pads[1], _ptr(scale), _ptr(shift), _ptr(resq), int(act), float(alpha))
        return y
    if not is_q4(xq) or (resq is not None and not is_q4(resq)):
        raise TypeError("ConvQ4 needs Q4 activations (planer_amd.q4.to_q4)")
    n, cin, h, w = logical_shape(xq)
    cout, cin_g, kh, kw = Kq.shape
    if cin_g * group != cin:
        raise ValueError("conv: weight %s does not match input %s with group=%d" % (Kq.shape, (n, cin, h, w), group))
    pads = [int(p) for p in pads]
    strides = [int(s) for s in strides]
    dilations = [int(d) for d in dilations]
    ho, wo = conv_out_hw(h, w, kh, kw, strides, dilations, pads)
    y = _new_q4(n, cout, ho, wo, xq.ctx)
    if resq is not None and resq.shape != y.shape:
        raise ValueError("fused residual shape %s != conv output %s" % (resq.shape, y.shape))
    if w_layout == 8:
        if not w1d_q4_eligible(Kq.shape, group, strides, dilations, pads):
            raise ValueError("1-D winograd filters serve 3x3 / stride 1 / pad 1 / group 1 convs only")
        _lib.call("pl_conv2d_w1d4_q4_f32", xq.ctx.handle, xq.ptr, n, cin, h, w, Kq.ptr, cout, _ptr(B), y.ptr,
                  _ptr(scale), _ptr(shift), _ptr(resq), int(act), float(alpha))
        return y
    if w_layout == 11 and (h not in (7, 14, 21) or w not in (7, 14, 21)):
        raise ValueError("mixed-tile winograd filters serve maps whose sides are 7, 14 or 21")
    if w_layout in (4, 7, 9, 11):
        if not winograd_q4_eligible(Kq.shape, group, strides, dilations, pads):
            raise ValueError("winograd Q4 filters serve 3x3 / stride 1 / pad 1 / group 1 convs only")
        if w_layout == 9 and any(a is not None and a.ptr % 16 for a in (B, scale, shift)):
            raise ValueError("the fused F(4x4,3x3) kernel reads bias / scale / shift as 16-byte quads: misaligned parameter")
        _lib.call({4: "pl_conv2d_winograd_q4_f32", 7: "pl_conv2d_winograd4_q4_f32", 9: "pl_conv2d_wf4_q4_f32",
                   11: "pl_conv2d_winograd43_q4_f32"}[w_layout], xq.ctx.handle, xq.ptr, n, cin, h, w, Kq.ptr, cout, _ptr(B), y.ptr,
                  _ptr(scale), _ptr(shift), _ptr(resq), int(act), float(alpha))
        return y
    _lib.call("pl_conv2d_q4_f32", xq.ctx.handle, xq.ptr, n, cin, h, w, Kq.ptr, cout, kh, kw,
              _ptr(B), y.ptr, strides[0], strides[1], dilations[0], dilations[1],
              pads[0], pads[1], pads[2], pads[3], int(group),
              _ptr(scale), _ptr(shift), _ptr(resq), int(act), float(alpha))
    return y


def stem_pool_eligible(x_shape, k_shape, group=1, strides=(1, 1), dilations=(1, 1), pads=(0, 0, 0, 0), **_):
    """Whether the row-packed conv + maxpool(3x3 / s2 / p1) kernel (csrc/conv_stem_pool_kernel.h) takes this conv."""
    if len(x_shape) != 4 or int(group) != 1 or list(dilations) != [1, 1] or len(pads) != 4 or pads[0] != pads[2] or pads[1] != pads[3]:
        return False
    cout, cin, kh, kw = k_shape
    ok = ctypes.c_int()
    _lib.call("pl_conv2d_rowpacked_pool_supported", int(cin), int(x_shape[2]), int(x_shape[3]), int(cout), int(kh), int(kw),
              int(strides[0]), int(strides[1]), int(pads[0]), int(pads[1]), ctypes.byref(ok))
    return bool(ok.value) and x_shape[1] == cin


def stem_pool_nchw_eligible(x_shape, k_shape, group=1, strides=(1, 1), dilations=(1, 1), pads=(0, 0, 0, 0), **_):
    """Whether the stem + max-pool kernel can read this NCHW input itself (W % 4 == 0 on top of stem_pool_eligible)."""
    if not stem_pool_eligible(x_shape, k_shape, group, strides, dilations, pads):
        return False
    cout, cin, kh, kw = k_shape
    ok = ctypes.c_int()
    _lib.call("pl_conv2d_stem_pool_nchw_supported", int(cin), int(x_shape[2]), int(x_shape[3]), int(cout), int(kh), int(kw),
              int(strides[0]), int(strides[1]), int(pads[0]), int(pads[1]), ctypes.byref(ok))
    return bool(ok.value)


def ConvPoolQ4(x, Kq, B=None, scale=None, shift=None, group=1, strides=(1, 1), dilations=(1, 1), pads=(0, 0, 0, 0),
               act=ACT_NONE, alpha=0.0, w_layout=10, out=None, src_ptr=None, ctx=None, strip_rows=0, **_):
    """Row-packed stem conv (ConvQ4 w_layout 6: NCHW input, filter from prepare_rowpack_weights) with its fused tail, followed
    by layer.Maxpool(w=(3, 3), strides=(2, 2), pads=(1, 1, 1, 1)) (layer.py:71-72), in ONE kernel: only the pooled Q4 tensor is
    written.  Emitted by the plan compiler (Net._fuse_stem_pool) where the max-pool is the conv's only reader.
    w_layout 12: the kernel reads the NCHW tensor itself (filter from prepare_stem_nchw_weights; W % 4 == 0) -- no row-packed
    copy.  A plan's static input then carries `x.prefed = (feed, pooled)`: whoever feeds the plan runs this kernel from the
    caller's batch straight into `pooled` (`out` / `src_ptr` / `ctx` below), and the captured pass starts behind it."""
    _f32(x, Kq, B, scale, shift)
    if w_layout == 12 and out is None and getattr(x, "prefed", None) is not None:
        return x.prefed[1]                         # a plan's static input: the feed has already run this step
    if is_q4(x) or not stem_pool_eligible(x.shape, Kq.shape, group, strides, dilations, pads):
        raise NotImplementedError("conv + maxpool in one kernel: NCHW 3-channel input, 7x7 / stride 2 / pad 3")
    if any(a is not None and a.ptr % 16 for a in (B, scale, shift)):
        raise ValueError("the stem + max-pool kernel reads bias / scale / shift as 16-byte quads: misaligned parameter")
    n, cin, h, w = x.shape
    cout, _, kh, kw = Kq.shape
    pads, strides = [int(p) for p in pads], [int(s) for s in strides]
    ho, wo = conv_out_hw(h, w, kh, kw, strides, [1, 1], pads)
    cx = ctx or x.ctx
    if w_layout == 12:
        xptr = x.ptr if src_ptr is None else src_ptr
        if w % 4 or xptr % 16:
            raise NotImplementedError("the NCHW stem + max-pool kernel needs W % 4 == 0 and a 16-byte aligned input")
        y = out if out is not None else _new_q4(n, cout, (ho + 1) // 2, (wo + 1) // 2, cx)
        _lib.call("pl_conv2d_stem_pool_nchw_q4_f32", cx.handle, xptr, n, h, w, Kq.ptr, cout, _ptr(B), y.ptr, _ptr(scale),
                  _ptr(shift), int(act), float(alpha), int(strip_rows))
        return y
    geom = (kw, strides[1], pads[0], pads[1])
    if x.packed is not None and x.packed[0] == geom:
        img = x.packed[1]                          # a plan's static input: whoever feeds the plan keeps the image current
    else:
        elems = ctypes.c_size_t()
        _lib.call("pl_rowpack_input_elems", n, cin, h, w, geom[0], geom[1], geom[2], geom[3], ctypes.byref(elems))
        img = empty((int(elems.value),), ctx=x.ctx)
        _lib.call("pl_rowpack_input_f32", x.ctx.handle, x.ptr, img.ptr, n, cin, h, w, geom[0], geom[1], geom[2], geom[3])
    y = _new_q4(n, cout, (ho + 1) // 2, (wo + 1) // 2, x.ctx)
    _lib.call("pl_conv2d_rowpacked_pool_q4_f32", x.ctx.handle, img.ptr, n, cin, h, w, Kq.ptr, cout, kh, kw, _ptr(B), y.ptr,
              strides[0], strides[1], pads[0], pads[1], _ptr(scale), _ptr(shift), int(act), float(alpha))
    return y


def stem_pool_feeder(x_shape, Kq, B, scale, shift, act, alpha, pooled, strip_rows=0):
    """feed(src_ptr, ctx): the NCHW stem + max-pool kernel from the batch at `src_ptr` into the persistent tensor `pooled`, on
    `ctx`'s stream -- what `DeviceArray.prefed` holds for a plan's static input (Net._pack_static_inputs)."""
    n, _, h, w = (int(v) for v in x_shape)
    cout = int(Kq.shape[0])

    def feed(src_ptr, ctx):
        _lib.call("pl_conv2d_stem_pool_nchw_q4_f32", ctx.handle, src_ptr, n, h, w, Kq.ptr, cout, _ptr(B), pooled.ptr, _ptr(scale),
                  _ptr(shift), int(act), float(alpha), int(strip_rows))
    return feed


def ConvQ4Pair(xq, K1, B1, scale1, shift1, K2, B2, scale2, shift2, para1=None, para2=None, **_):
    """Two fused convs (ConvQ4, w_layout 2, group 1, no dilation, no residual) that read the SAME Q4 input, in one
    launch -> (y1, y2).  Emitted by plan.pair_sibling_convs where a graph forks into two convs (ResNet's stride-2
    3x3 conv and the 1x1 stride-2 projection of the same block)."""
    _f32(xq, K1, B1, scale1, shift1, K2, B2, scale2, shift2)
    if not is_q4(xq):
        raise TypeError("ConvQ4Pair needs a Q4 activation")
    n, cin, h, w = logical_shape(xq)
    outs, args = [], []
    for K, B, sc, sh, para in ((K1, B1, scale1, shift1, para1 or {}), (K2, B2, scale2, shift2, para2 or {})):
        cout, cin_g, kh, kw = K.shape
        strides = [int(v) for v in para.get("strides", (1, 1))]
        pads = [int(v) for v in para.get("pads", (0, 0, 0, 0))]
        if (cin_g != cin or int(para.get("group", 1)) != 1 or [int(v) for v in para.get("dilations", (1, 1))] != [1, 1]
                or pads[0] != pads[2] or pads[1] != pads[3] or int(para.get("act", 0)) & ~3):
            raise ValueError("ConvQ4Pair: group 1, dilation 1, symmetric pads, no residual; weight %s on input %s"
                             % (K.shape, (n, cin, h, w)))
        ho, wo = conv_out_hw(h, w, kh, kw, strides, [1, 1], pads)
        y = _new_q4(n, cout, ho, wo, xq.ctx)
        outs.append(y)
        args += [K.ptr, cout, kh, kw, strides[0], strides[1], pads[0], pads[1], _ptr(B), _ptr(sc), _ptr(sh),
                 int(para.get("act", 0)), float(para.get("alpha", 0.0)), y.ptr]
    _lib.call("pl_conv2d_q4_pair_f32", xq.ctx.handle, xq.ptr, n, cin, h, w, *args)
    return tuple(outs)


# ---- Winograd F(4x4,3x3) stage by stage (plan-internal; plan.chain_winograd emits these) -----------
def _wino_tensor(n, c, h, w, ctx):
    """V or M of an (n, c, h, w) activation: [36][c/4][T][4], T = n * ceil(h/4) * ceil(w/4)."""
    e = ctypes.c_size_t()
    _lib.call("pl_wino4_elems", n, c, h, w, ctypes.byref(e))
    t = empty((max(e.value, 1),), ctx=ctx)
    t.meta = (n, c, h, w)
    return t


def wino4_chain_supported(shape, ctx):
    """Can the LDS transform kernel (whole planes per workgroup) take an (N, C, H, W) activation?"""
    n, c, h, w = shape
    if c % 4:
        return False
    ok = ctypes.c_int()
    _lib.call("pl_wino4_chain_supported", ctx.handle, n, c, h, w, ctypes.byref(ok))
    return bool(ok.value)


def Wino4In(xq):
    """B^T d B of every 6x6 tile of a Q4 activation -> V."""
    _f32(xq)
    if not is_q4(xq):
        raise TypeError("Wino4In needs a Q4 activation")
    n, c, h, w = logical_shape(xq)
    v = _wino_tensor(n, c, h, w, xq.ctx)
    _lib.call("pl_wino4_input_q4_f32", xq.ctx.handle, xq.ptr, n, c, h, w, v.ptr)
    return v


def Wino4Gemm(v, Kq, **_):
    """The 36 per-frequency GEMMs: V (Cin) x Winograd-domain filters -> M (Cout)."""
    n, cin, h, w = v.meta
    cout, cin_k, kh, kw = Kq.shape
    if cin_k != cin or (kh, kw) != (3, 3):
        raise ValueError("conv: weight %s does not match input %s" % (Kq.shape, (n, cin, h, w)))
    m = _wino_tensor(n, cout, h, w, v.ctx)
    _lib.call("pl_wino4_gemm_q4_f32", v.ctx.handle, v.ptr, n, cin, h, w, Kq.ptr, cout, m.ptr)
    return m


def _wino_tail_check(m, resq):
    n, c, h, w = m.meta
    if resq is not None and (not is_q4(resq) or logical_shape(resq) != (n, c, h, w)):
        raise ValueError("fused residual %s != conv output %s" % (getattr(resq, "shape", None), (n, c, h, w)))
    return n, c, h, w


def Wino4Out(m, B=None, scale=None, shift=None, resq=None, act=ACT_NONE, alpha=0.0, **_):
    """A^T m A + the conv's fused tail -> y (Q4)."""
    _f32(B, scale, shift, resq)
    n, c, h, w = _wino_tail_check(m, resq)
    y = _new_q4(n, c, h, w, m.ctx)
    _lib.call("pl_wino4_output_q4_f32", m.ctx.handle, m.ptr, n, c, h, w, _ptr(B), _ptr(scale), _ptr(shift), _ptr(resq),
              int(act), float(alpha), y.ptr)
    return y


def Wino4Chain(m, B=None, scale=None, shift=None, resq=None, act=ACT_NONE, alpha=0.0, keep_y=True, **_):
    """Wino4Out and the Wino4In of the next 3x3 conv in one kernel: -> (y, V) or, with keep_y=False
    (nothing else reads y), V alone -- y then never exists in memory."""
    _f32(B, scale, shift, resq)
    n, c, h, w = _wino_tail_check(m, resq)
    y = _new_q4(n, c, h, w, m.ctx) if keep_y else None
    v = _wino_tensor(n, c, h, w, m.ctx)
    _lib.call("pl_wino4_chain_q4_f32", m.ctx.handle, m.ptr, n, c, h, w, _ptr(B), _ptr(scale), _ptr(shift), _ptr(resq),
              int(act), float(alpha), _ptr(y), v.ptr)
    return (y, v) if keep_y else v


# ---- mixed-tile Winograd (maps of 7 / 14 / 21 a side) stage by stage: the same four stages, other kernels ----
def _wino43_tensor(n, c, h, w, ctx):
    e = ctypes.c_size_t()
    _lib.call("pl_wino43_elems", n, c, h, w, ctypes.byref(e))
    t = empty((max(e.value, 1),), ctx=ctx)
    t.meta = (n, c, h, w)
    return t


def Wino43In(xq):
    _f32(xq)
    if not is_q4(xq):
        raise TypeError("Wino43In needs a Q4 activation")
    n, c, h, w = logical_shape(xq)
    v = _wino43_tensor(n, c, h, w, xq.ctx)
    _lib.call("pl_wino43_input_q4_f32", xq.ctx.handle, xq.ptr, n, c, h, w, v.ptr)
    return v


def Wino43Gemm(v, Kq, **_):
    n, cin, h, w = v.meta
    cout, cin_k, kh, kw = Kq.shape
    if cin_k != cin or (kh, kw) != (3, 3):
        raise ValueError("conv: weight %s does not match input %s" % (Kq.shape, (n, cin, h, w)))
    m = _wino43_tensor(n, cout, h, w, v.ctx)
    _lib.call("pl_wino43_gemm_q4_f32", v.ctx.handle, v.ptr, n, cin, h, w, Kq.ptr, cout, m.ptr)
    return m


def Wino43Out(m, B=None, scale=None, shift=None, resq=None, act=ACT_NONE, alpha=0.0, **_):
    _f32(B, scale, shift, resq)
    n, c, h, w = _wino_tail_check(m, resq)
    y = _new_q4(n, c, h, w, m.ctx)
    _lib.call("pl_wino43_output_q4_f32", m.ctx.handle, m.ptr, n, c, h, w, _ptr(B), _ptr(scale), _ptr(shift), _ptr(resq),
              int(act), float(alpha), y.ptr)
    return y


def Wino43Chain(m, B=None, scale=None, shift=None, resq=None, act=ACT_NONE, alpha=0.0, keep_y=True, **_):
    _f32(B, scale, shift, resq)
    n, c, h, w = _wino_tail_check(m, resq)
    y = _new_q4(n, c, h, w, m.ctx) if keep_y else None
    v = _wino43_tensor(n, c, h, w, m.ctx)
    _lib.call("pl_wino43_chain_q4_f32", m.ctx.handle, m.ptr, n, c, h, w, _ptr(B), _ptr(scale), _ptr(shift), _ptr(resq),
              int(act), float(alpha), _ptr(y), v.ptr)
    return (y, v) if keep_y else v


def Conv1x1WinoIn(xq, Kq, B=None, scale=None, shift=None, act=ACT_NONE, alpha=0.0, wino=4, **_):
    """ConvQ4 (1x1, stride 1, group 1, fused bias / scale / shift / activation, no residual) and the Wino4In of the 3x3 conv
    that is its only reader, in one kernel -> V (csrc/conv1x1_wino_in_kernel.h).  Emitted by plan.fuse_conv1x1_wino_in for the
    1x1 -> 3x3 pairs of a detection net's blocks at small maps; the 1x1 conv's own output never exists."""
    _f32(xq, Kq, B, scale, shift)
    if not is_q4(xq):
        raise TypeError("Conv1x1WinoIn needs a Q4 activation")
    n, cin, h, w = logical_shape(xq)
    cout, cin_k, kh, kw = Kq.shape
    if (kh, kw) != (1, 1) or cin_k != cin or cout % 4 or int(wino) != 4:
        raise ValueError("Conv1x1WinoIn: 1x1 filter %s on input %s, Cout %% 4 == 0, F(4x4,3x3) tiles" % (Kq.shape, (n, cin, h, w)))
    v = _wino_tensor(n, cout, h, w, xq.ctx)
    _lib.call("pl_conv1x1_wino_in_q4_f32", xq.ctx.handle, xq.ptr, n, cin, h, w, Kq.ptr, cout, _ptr(B), _ptr(scale), _ptr(shift),
              int(act), float(alpha), int(wino), v.ptr)
    return v


# ---- HBM-bound layers on Q4 tensors ---------------------------------------------------------
def _like(x, shape=None):
    y = empty(shape or x.shape, ctx=x.ctx)
    y.chan = x.chan
    return y


def _pool_q4(xq, w, pads, strides, mode):
    _f32(xq)
    n, c, h, wd = logical_shape(xq)
    kh, kw = int(w[0]), int(w[1])
    sh, sw = int(strides[0]), int(strides[1])
    pads = [int(p) for p in pads]
    ho = (h + pads[0] + pads[2] - kh + sh) // sh        # util.py:84
    wo = (wd + pads[1] + pads[3] - kw + sw) // sw       # util.py:85
    y = _new_q4(n, c, ho, wo, xq.ctx)
    _lib.call("pl_pool2d_q4_f32", xq.ctx.handle, xq.ptr, y.ptr, n, c, h, wd, kh, kw, sh, sw,
              pads[0], pads[1], pads[2], pads[3], mode)
    return y


def MaxpoolQ4(xq, w=(2, 2), pads=(0, 0, 0, 0), strides=(2, 2)):
    """layer.Maxpool (layer.py:71-72) on a Q4 tensor."""
    return _pool_q4(xq, w, pads, strides, 0)


def AveragePoolQ4(xq, w=(2, 2), pads=(0, 0, 0, 0), strides=(2, 2)):
    """layer.AveragePool (layer.py:74-75) on a Q4 tensor."""
    return _pool_q4(xq, w, pads, strides, 1)


def GlobalAveragePoolQ4(xq):
    """layer.GlobalAveragePool (layer.py:77-78): Q4 in, plain (N, C, 1, 1) out."""
    _f32(xq)
    n, c, h, w = logical_shape(xq)
    y = empty((n, c, 1, 1), ctx=xq.ctx)
    _lib.call("pl_gap_q4_f32", xq.ctx.handle, xq.ptr, y.ptr, n, c, h * w)
    return y


def UpSampleQ4(xq, k, mode="nearest"):
    """layer.UpSample (layer.py:80-82) on a Q4 tensor."""
    _f32(xq)
    if mode != "nearest":
        raise NotImplementedError("upsample mode %r is not on the HIP path" % mode)
    kv = _host_values(k)
    if kv.size == 0:
        raise ValueError("upsample needs scales (the reference's size-only branch is broken, layer.py:81)")
    fh, fw = [int(v) for v in kv[-2:].astype(int).tolist()]
    n, c, h, w = logical_shape(xq)
    y = _new_q4(n, c, h * fh, w * fw, xq.ctx)
    _lib.call("pl_upsample_nearest_q4_f32", xq.ctx.handle, xq.ptr, y.ptr, n, c, h, w, fh, fw)
    return y


def BatchNormQ4(xq, K, B):
    """layer.BatchNorm (layer.py:125-127) on a Q4 tensor (only reached when it could not be fused)."""
    _f32(xq, K, B)
    n, c, h, w = logical_shape(xq)
    if K.size != c or B.size != c:
        raise ValueError("batchnorm: K/B must hold one value per channel")
    y = _like(xq)
    _lib.call("pl_scale_shift_q4_f32", xq.ctx.handle, xq.ptr, y.ptr, K.ptr, B.ptr, n, c, h * w)
    return y


def ReLUQ4(xq):
    """layer.ReLU (layer.py:44-46): in place on the padded buffer (relu(0) = 0 keeps the padding)."""
    _lib.call("pl_relu_f32", xq.ctx.handle, xq.ptr, xq.ptr, xq.size)
    return xq


def LeakyReLUQ4(xq, alpha=0.2):
    y = _like(xq)
    _lib.call("pl_leakyrelu_f32", xq.ctx.handle, xq.ptr, y.ptr, xq.size, float(alpha))
    return y


def SigmoidQ4(xq):
    """Only scheduled for C % 4 == 0 (sigmoid(0) = 0.5 would dirty the padding lanes)."""
    y = _like(xq)
    _lib.call("pl_sigmoid_f32", xq.ctx.handle, xq.ptr, y.ptr, xq.size)
    return y


def AddQ4(x1, x2):
    """layer.Add (layer.py:93-95), equal shapes, both Q4."""
    if not (is_q4(x1) and is_q4(x2)) or x1.shape != x2.shape or x1.chan != x2.chan:
        raise ValueError("AddQ4 needs two Q4 tensors of one shape")
    y = _like(x1)
    _lib.call("pl_add_f32", x1.ctx.handle, x1.ptr, x2.ptr, y.ptr, x1.size)
    return y


def ConcatenateQ4(*xs, axis=1):
    """layer.Concatenate (layer.py:90-91) along channels; every input needs C % 4 == 0 so that the
    quads of consecutive inputs abut."""
    if axis != 1 or any(not is_q4(a) or a.chan % 4 for a in xs):
        raise ValueError("ConcatenateQ4: channel axis, C % 4 == 0 inputs only")
    n, _, h, w, _ = xs[0].shape
    if any(a.shape[0] != n or a.shape[2:] != xs[0].shape[2:] for a in xs):
        raise ValueError("concat: shapes differ off the axis: %s" % [logical_shape(b) for b in xs])
    total = sum(a.chan for a in xs)
    if len(xs) == 2:
        return UpConcatQ4(xs[0], None, xs[1])
    y = _new_q4(n, total, h, w, xs[0].ctx)
    off, pitch = 0, (total // 4) * h * w * 4
    for a in xs:
        width = a.shape[1] * h * w * 4
        if width and n:
            _lib.call("pl_copy2d_f32", y.ctx.handle, y.ptr + off * 4, pitch, a.ptr, width, width, n)
        off += width
    return y


def UpConcatQ4(aq, k, bq, mode="nearest", axis=1):
    """layer.Concatenate([layer.UpSample(a, k), b], axis=1) on Q4 tensors in one kernel (k = None: no upsampling).
    Emitted by the plan compiler for upsample -> concat routes; also serves every two-input Q4 concat."""
    _f32(aq, bq)
    if axis != 1 or mode != "nearest" or not is_q4(aq) or not is_q4(bq) or aq.chan % 4 or bq.chan % 4:
        raise ValueError("UpConcatQ4: two Q4 tensors with C % 4 == 0, channel axis, nearest mode")
    fh = fw = 1
    if k is not None:
        kv = _host_values(k)
        if kv.size == 0:
            raise ValueError("upsample needs scales (the reference's size-only branch is broken, layer.py:81)")
        fh, fw = [int(v) for v in kv[-2:].astype(int).tolist()]
    n, ca, ha, wa = logical_shape(aq)
    nb, cb, h, w = logical_shape(bq)
    if nb != n or (ha * fh, wa * fw) != (h, w):
        raise ValueError("concat: shapes differ off the axis: %s (x%d, x%d) vs %s" % ((n, ca, ha, wa), fh, fw, (nb, cb, h, w)))
    y = _new_q4(n, ca + cb, h, w, aq.ctx)
    _lib.call("pl_concat2_q4_f32", aq.ctx.handle, aq.ptr, bq.ptr, y.ptr, n, ca, cb, h, w, fh, fw)
    return y


# kind -> Q4 implementation, for plan.assign_layouts (conv kinds are handled by the plan compiler)
Q4_LAYERS = {"maxpool": MaxpoolQ4, "averagepool": AveragePoolQ4, "gap": GlobalAveragePoolQ4,
             "upsample": UpSampleQ4, "batchnorm": BatchNormQ4, "relu": ReLUQ4, "leakyrelu": LeakyReLUQ4,
             "sigmoid": SigmoidQ4, "add": AddQ4, "concat": ConcatenateQ4}


def register(layer_map):
    """Plan-internal kinds (never present in a user's IR)."""
    layer_map.update({"to_q4": to_q4, "from_q4": from_q4, "conv_q4": ConvQ4, "upconcat_q4": UpConcatQ4,
                      "wino4_in": Wino4In, "wino4_gemm": Wino4Gemm, "wino4_out": Wino4Out, "wino4_chain": Wino4Chain,
                      "conv_q4_pair": ConvQ4Pair, "conv_pool_q4": ConvPoolQ4, "conv1x1_wino_in": Conv1x1WinoIn,
                      "wino43_in": Wino43In, "wino43_gemm": Wino43Gemm, "wino43_out": Wino43Out, "wino43_chain": Wino43Chain})
    layer_map.update({k + "_q4": f for k, f in Q4_LAYERS.items()})
