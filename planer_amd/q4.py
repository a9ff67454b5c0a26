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


def ConvQ4(xq, Kq, B=None, scale=None, shift=None, resq=None, group=1, strides=(1, 1),
           dilations=(1, 1), pads=(0, 0, 0, 0), act=ACT_NONE, alpha=0.0, **_):
    """layer.ConvFused on Q4 tensors: act((conv(x,K)+B)*scale + shift + res), all activations Q4."""
    _f32(xq, Kq, B, scale, shift, resq)
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
    _lib.call("pl_conv2d_q4_f32", xq.ctx.handle, xq.ptr, n, cin, h, w, Kq.ptr, cout, kh, kw,
              _ptr(B), y.ptr, strides[0], strides[1], dilations[0], dilations[1],
              pads[0], pads[1], pads[2], pads[3], int(group),
              _ptr(scale), _ptr(shift), _ptr(resq), int(act), float(alpha))
    return y
