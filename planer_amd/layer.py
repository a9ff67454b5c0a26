"""planer's operator table on MI355X.

Same contract as the reference's `layer_map` (layer.py:262-281): every entry
is `f(*arrays, **json_params)` with the reference's positional order, keyword
names and defaults, and `wrap(f, kind)(**params)` gives the Layer objects
`Net` instantiates (layer.py:6-13).  Here every function takes and returns
`DeviceArray`s and enqueues hand-written HIP kernels through the C ABI
(include/planer_hip.h); no activation is ever computed on the host.  (Integer
shape tensors -- what `Shape` returns and what is derived from it -- are
evaluated on their host mirrors, see "shape-domain tensors" below.)

Every kind of the reference's table (layer.py:262-281) is here; an unknown
kind raises KeyError in Net like the reference (net.py:15), and inputs a kernel
does not cover raise NotImplementedError instead of silently running on the CPU.
"""
import numpy

from . import _lib
from .hip import DeviceArray, asarray, empty

ACT_NONE, ACT_RELU, ACT_LEAKY = _lib.ACT_NONE, _lib.ACT_RELU, _lib.ACT_LEAKY


def wrap(f, layername="layer"):
    """layer.wrap (layer.py:6-13): bind json params to an operator."""
    class Layer:
        name = layername

        def __init__(self, **key):
            self.key = key

        def para(self):
            return self.key

        def forward(self, *x):
            return f(*x, **self.key)

        __call__ = forward
    Layer.__name__ = "Layer_" + layername
    return Layer


# ---- helpers -----------------------------------------------------------------
def _f32(*arrays):
    for a in arrays:
        if a is not None:
            if not isinstance(a, DeviceArray):
                raise TypeError("expected DeviceArray, got %s (use planer_amd.asarray)" % type(a).__name__)
            if a.dtype != numpy.float32:
                raise NotImplementedError("the HIP path computes in float32, got %s" % a.dtype)


def _ptr(a):
    return None if a is None else a.ptr


def _host_values(t):
    """Small parameter tensors (UpSample's scale vector) are read on the host;
    weights keep a host mirror so this never synchronises the stream."""
    if isinstance(t, DeviceArray):
        return t.host if t.host is not None else t.get()
    return numpy.asarray(t)


def conv_out_hw(h, w, kh, kw, strides, dilations, pads):
    """util.py:25-26"""
    ho = (h + pads[0] + pads[2] - (kh - 1) * dilations[0] - 1 + strides[0]) // strides[0]
    wo = (w + pads[1] + pads[3] - (kw - 1) * dilations[1] - 1 + strides[1]) // strides[1]
    return ho, wo


# ---- MFMA-bound ------------------------------------------------------------------
def Conv2d(x, K, B=None, group=1, strides=(1, 1), dilations=(1, 1), pads=(0, 0, 0, 0)):
    """layer.Conv2d (layer.py:22-26): implicit-GEMM conv + bias on fp32 MFMA."""
    return ConvFused(x, K, B, group=group, strides=strides, dilations=dilations, pads=pads)


def prepare_conv_weights(K):
    """OIHW filters -> the tap-major layout [Cout][kh*kw][Cin/g] the fast conv
    kernel reads (done once per model by Net's plan compiler).  The returned
    array keeps the logical OIHW shape; only the bytes are permuted."""
    _f32(K)
    cout, cin_g, kh, kw = K.shape
    if cin_g % 16:
        raise ValueError("tap-major filters need Cin/group % 16 == 0")
    if kh * kw == 1:
        return K                                   # identical in both layouts
    out = empty(K.shape, ctx=K.ctx)
    _lib.call("pl_conv2d_prepare_weights_f32", K.ctx.handle, K.ptr, cout, cin_g, kh, kw, out.ptr)
    return out


def winograd_eligible(k_shape, group=1, strides=(1, 1), dilations=(1, 1), pads=(0, 0, 0, 0)):
    """3x3 / stride 1 / pad 1 / no dilation / no groups, Cin % 16 == 0."""
    cout, cin_g, kh, kw = k_shape
    return (kh == 3 and kw == 3 and group == 1 and cin_g % 16 == 0 and list(strides) == [1, 1]
            and list(dilations) == [1, 1] and list(pads) == [1, 1, 1, 1])


def prepare_winograd_weights(K):
    """OIHW 3x3 filters -> Winograd F(2x2,3x3) domain U[16][Cout][Cin] (w_layout=3).  The
    returned array keeps the logical OIHW shape; its allocation holds 16*Cout*Cin floats."""
    _f32(K)
    cout, cin, kh, kw = K.shape
    if (kh, kw) != (3, 3) or cin % 16:
        raise ValueError("winograd filters need 3x3 kernels and Cin % 16 == 0")
    out = empty((16 * cout * cin,), ctx=K.ctx)
    _lib.call("pl_conv2d_prepare_winograd_f32", K.ctx.handle, K.ptr, cout, cin, out.ptr)
    out.shape = K.shape
    return out


def ConvFused(x, K, B=None, scale=None, shift=None, res=None, group=1, strides=(1, 1),
              dilations=(1, 1), pads=(0, 0, 0, 0), act=ACT_NONE, alpha=0.0, w_layout=0):
    """Conv2d with BatchNorm / Add / (Leaky)ReLU folded into its epilogue:
    act((conv(x,K)+B)*scale + shift + res).  Emitted by Net's plan compiler for
    the chains conv->batchnorm->[add]->[relu|leakyrelu]; not a reference op.
    w_layout=1: K holds tap-major bytes from prepare_conv_weights();
    w_layout=3: K holds Winograd-domain filters from prepare_winograd_weights()."""
    _f32(x, K, B, scale, shift, res)
    n, cin, h, w = x.shape
    cout, cin_g, kh, kw = K.shape
    if cin_g * group != cin:
        raise ValueError("conv: weight %s does not match input %s with group=%d" % (K.shape, x.shape, group))
    pads = [int(p) for p in pads]
    strides = [int(s) for s in strides]
    dilations = [int(d) for d in dilations]
    ho, wo = conv_out_hw(h, w, kh, kw, strides, dilations, pads)
    y = empty((n, cout, ho, wo), ctx=x.ctx)
    if res is not None and res.size != y.size:
        raise ValueError("fused residual shape %s != conv output %s" % (res.shape, y.shape))
    _lib.call("pl_conv2d_fused_f32", x.ctx.handle, x.ptr, n, cin, h, w, K.ptr, cout, kh, kw,
              _ptr(B), y.ptr, strides[0], strides[1], dilations[0], dilations[1],
              pads[0], pads[1], pads[2], pads[3], int(group),
              _ptr(scale), _ptr(shift), _ptr(res), int(act), float(alpha), int(w_layout))
    return y


def Dense(x, K, B, shp=None):
    """layer.Dense (layer.py:15-18): x @ K.T + B; `shp` is ignored there too."""
    _f32(x, K, B)
    m, k = x.shape
    n = K.shape[0]
    y = empty((m, n), ctx=x.ctx)
    _lib.call("pl_gemm_f32", x.ctx.handle, x.ptr, m, k, K.ptr, n, 1, _ptr(B), y.ptr)
    return y


def MatMul(x, y):
    """layer.MatMul (layer.py:20): np.matmul -- 2-D operands, or stacks of matrices whose leading
    dimensions are equal or absent on one side (one MFMA GEMM per matrix of the stack)."""
    _f32(x, y)
    if x.ndim < 2 or y.ndim < 2:
        raise NotImplementedError("MatMul on the HIP path needs operands with at least 2 dimensions")
    m, k = x.shape[-2:]
    if y.shape[-2] != k:
        raise ValueError("matmul: inner dimensions differ: %s @ %s" % (x.shape, y.shape))
    n = y.shape[-1]
    if x.ndim == 2 and y.ndim == 2:
        out = empty((m, n), ctx=x.ctx)
        _lib.call("pl_gemm_f32", x.ctx.handle, x.ptr, m, k, y.ptr, n, 0, None, out.ptr)
        return out
    lead = x.shape[:-2] if x.ndim >= y.ndim else y.shape[:-2]
    if (x.ndim > 2 and y.ndim > 2 and x.shape[:-2] != y.shape[:-2]):
        raise NotImplementedError("matmul: stacks must have equal leading dimensions (no broadcasting between stacks)")
    if y.ndim == 2:                                     # (..., m, k) @ (k, n): one GEMM over all rows
        out = empty(x.shape[:-1] + (n,), ctx=x.ctx)
        rows = x.size // k
        _lib.call("pl_gemm_f32", x.ctx.handle, x.ptr, rows, k, y.ptr, n, 0, None, out.ptr)
        return out
    count = int(numpy.prod(lead, dtype=numpy.int64))
    out = empty(tuple(lead) + (m, n), ctx=x.ctx)
    for b in range(count):
        xa = x.ptr + (b * m * k * 4 if x.ndim > 2 else 0)
        ya = y.ptr + b * k * n * 4
        _lib.call("pl_gemm_f32", x.ctx.handle, xa, m, k, ya, n, 0, None, out.ptr + b * m * n * 4)
    return out


# ---- HBM-bound ---------------------------------------------------------------------
def BatchNorm(x, K, B):
    """layer.BatchNorm (layer.py:125-127): x*K + B with K,B folded (1,C,1,1)."""
    _f32(x, K, B)
    c = x.shape[1]
    if K.size != c or B.size != c:
        raise ValueError("batchnorm: K/B must hold one value per channel")
    inner = x.size // (x.shape[0] * c) if x.size else 1
    y = empty(x.shape, ctx=x.ctx)
    _lib.call("pl_scale_shift_f32", x.ctx.handle, x.ptr, y.ptr, K.ptr, B.ptr, x.shape[0], c, max(inner, 1))
    return y


def ReLU(x):
    """layer.ReLU (layer.py:44-46): in place, returns the SAME array object."""
    _f32(x)
    _lib.call("pl_relu_f32", x.ctx.handle, x.ptr, x.ptr, x.size)
    return x


def LeakyReLU(x, alpha=0.2):
    """layer.LeakyReLU (layer.py:48-51), out of place."""
    _f32(x)
    y = empty(x.shape, ctx=x.ctx)
    _lib.call("pl_leakyrelu_f32", x.ctx.handle, x.ptr, y.ptr, x.size, float(alpha))
    return y


def Sigmoid(x):
    """layer.Sigmoid (layer.py:61-64), out of place."""
    _f32(x)
    y = empty(x.shape, ctx=x.ctx)
    _lib.call("pl_sigmoid_f32", x.ctx.handle, x.ptr, y.ptr, x.size)
    return y


def Add(x1, x2):
    """layer.Add (layer.py:93-95): same shape or a (1|N,C,1,1) operand broadcast over the other (the two
    forms planer graphs contain) on their own kernels, any other numpy-broadcastable pair through `_binary`."""
    _f32(x1, x2)
    if x1.shape == x2.shape:
        y = empty(x1.shape, ctx=x1.ctx)
        _lib.call("pl_add_f32", x1.ctx.handle, x1.ptr, x2.ptr, y.ptr, x1.size)
        return y
    big, small = (x1, x2) if x1.size >= x2.size else (x2, x1)
    if (big.ndim >= 2 and small.ndim == big.ndim and small.shape[0] == 1
            and small.shape[1] == big.shape[1] and small.size == big.shape[1]):
        c = big.shape[1]
        y = empty(big.shape, ctx=big.ctx)
        _lib.call("pl_add_channel_f32", big.ctx.handle, big.ptr, small.ptr, y.ptr,
                  big.shape[0], c, big.size // (big.shape[0] * c))
        return y
    return _binary(x1, x2, 0)


def _pool(x, w, pads, strides, mode):
    _f32(x)
    n, c, h, wd = x.shape
    kh, kw = int(w[0]), int(w[1])
    sh, sw = int(strides[0]), int(strides[1])
    pads = [int(p) for p in pads]
    ho = (h + pads[0] + pads[2] - kh + sh) // sh        # util.py:84
    wo = (wd + pads[1] + pads[3] - kw + sw) // sw       # util.py:85
    y = empty((n, c, ho, wo), ctx=x.ctx)
    _lib.call("pl_pool2d_f32", x.ctx.handle, x.ptr, y.ptr, n * c, h, wd, kh, kw, sh, sw,
              pads[0], pads[1], pads[2], pads[3], mode)
    return y


def Maxpool(x, w=(2, 2), pads=(0, 0, 0, 0), strides=(2, 2)):
    """layer.Maxpool (layer.py:71-72): zero padding, accumulator starts at -1e4."""
    return _pool(x, w, pads, strides, 0)


def AveragePool(x, w=(2, 2), pads=(0, 0, 0, 0), strides=(2, 2)):
    """layer.AveragePool (layer.py:74-75): padding counted in the mean."""
    return _pool(x, w, pads, strides, 1)


def GlobalAveragePool(x):
    """layer.GlobalAveragePool (layer.py:77-78): mean over H,W, keepdims."""
    _f32(x)
    n, c = x.shape[:2]
    inner = x.size // (n * c) if x.size else 1
    y = empty((n, c) + (1,) * (x.ndim - 2), ctx=x.ctx)
    _lib.call("pl_gap_f32", x.ctx.handle, x.ptr, y.ptr, n * c, max(inner, 1))
    return y


def _linear_weights(fh, fw):
    """The float16 interpolation table of util.make_upmat (util.py:121-132) as float32, laid out
    (terms, fh, fw): sample fractions are a float16 linspace over (0.5/k, 1 - 0.5/k), the four corner weights
    their float16 products; with a factor of 1 on one axis only the other axis' two weights remain."""
    fy = numpy.linspace(0.5 / fh, 1 - 0.5 / fh, fh, dtype=numpy.float16)
    fx = numpy.linspace(0.5 / fw, 1 - 0.5 / fw, fw, dtype=numpy.float16)
    if fh == 1:
        tab = numpy.stack([1 - fx, fx])
    elif fw == 1:
        tab = numpy.stack([1 - fy, fy])
    else:
        gy, gx = (1 - fy)[:, None], (1 - fx)[None, :]
        tab = numpy.stack([gx * gy, fx[None, :] * gy, gx * fy[:, None], fx[None, :] * fy[:, None]])
    return numpy.ascontiguousarray(tab.reshape(tab.shape[0], -1), dtype=numpy.float32)


def _linear_positions(n, size, dtype=numpy.float32):
    """Sample positions of util.upsample_size along one axis (util.py:200-210): linspace in the image's dtype,
    clip, floor of the clipped-below-the-last-pixel position -> (lower index int32, fraction float32)."""
    k = size / n
    pos = numpy.linspace(-0.5 + 0.5 / k, n - 0.5 - 0.5 / k, size, dtype=dtype)
    pos = numpy.clip(pos, 0, n - 1, out=pos)
    lo = numpy.floor(numpy.clip(pos, 0, n - 1.001)).astype(int)
    pos -= lo
    return lo.astype(numpy.int32), pos


def _upsample_linear(x, fh, fw):
    """util.upsample, mode "linear" (util.py:212-219): integer factors -> upsample_blinear, anything else ->
    upsample_size at round(k * size)."""
    n, c, h, w = x.shape
    if fh == int(fh) and fw == int(fw):
        fh, fw = int(fh), int(fw)
        if fh == 1 and fw == 1:
            return x
        y = empty((n, c, h * fh, w * fw), ctx=x.ctx)
        if fh * fw > 64:
            raise NotImplementedError("linear upsample: fh * fw <= 64 on the HIP path, got %d x %d" % (fh, fw))
        tab = _linear_weights(fh, fw)
        _lib.call("pl_upsample_linear_f32", x.ctx.handle, x.ptr, y.ptr, n * c, h, w, fh, fw,
                  tab.ctypes.data_as(_lib.POINTER(_lib.c_float)))
        return y
    oh, ow = int(round(fh * h)), int(round(fw * w))
    if h < 2 or w < 2:
        raise ValueError("linear resize needs at least 2 x 2 pixels (the reference indexes row / column + 1)")
    ra, rs = _linear_positions(h, oh)
    ca, cs = _linear_positions(w, ow)
    dev = [asarray(a, ctx=x.ctx) for a in (ra, rs, ca, cs)]
    y = empty((n, c, oh, ow), ctx=x.ctx)
    _lib.call("pl_resize_linear_f32", x.ctx.handle, x.ptr, y.ptr, n * c, h, w, oh, ow, *[d.ptr for d in dev])
    return y


def UpSample(x, k, mode="nearest"):
    """layer.UpSample (layer.py:80-82): nearest-neighbour integer up-scaling or bilinear ("linear"),
    factors = last two entries of the tensor `k`."""
    _f32(x)
    if mode not in ("nearest", "linear"):
        raise NotImplementedError("upsample mode %r is not on the HIP path" % mode)
    kv = _host_values(k)
    if kv.size == 0:
        raise ValueError("upsample needs scales (the reference's size-only branch is broken, layer.py:81)")
    fh, fw = [int(v) for v in kv[-2:].astype(int).tolist()]       # truncated, layer.py:82
    if mode == "linear":
        return _upsample_linear(x, fh, fw)
    n, c, h, w = x.shape
    y = empty((n, c, h * fh, w * fw), ctx=x.ctx)
    _lib.call("pl_upsample_nearest_f32", x.ctx.handle, x.ptr, y.ptr, n * c, h, w, fh, fw)
    return y


def Concatenate(*xs, axis=0):
    """layer.Concatenate (layer.py:90-91); default axis 0 as in the reference."""
    _f32(*xs)
    nd = xs[0].ndim
    axis = axis + nd if axis < 0 else axis
    lead, trail = xs[0].shape[:axis], xs[0].shape[axis + 1:]
    for a in xs:
        if a.shape[:axis] != lead or a.shape[axis + 1:] != trail:
            raise ValueError("concat: shapes differ off the axis: %s" % [b.shape for b in xs])
    outer = int(numpy.prod(lead, dtype=numpy.int64)) if lead else 1
    tail = int(numpy.prod(trail, dtype=numpy.int64)) if trail else 1
    total = sum(a.shape[axis] for a in xs)
    y = empty(lead + (total,) + trail, ctx=xs[0].ctx)
    off, pitch = 0, total * tail
    for a in xs:
        width = a.shape[axis] * tail
        if width and outer:
            _lib.call("pl_copy2d_f32", y.ctx.handle, y.ptr + off * 4, pitch, a.ptr, width, width, outer)
        off += width
    return y


def Flatten(x):
    """layer.Flatten (layer.py:59): a view."""
    return x.reshape((x.shape[0], -1))


def Identity(x):
    return x


def Return(*x):
    """layer.Return (layer.py:260)."""
    return x


# ---- second-wave operators -----------------------------------------
def _broadcast_modes(x1, x2):
    """-> (out_shape, outer, C, inner, mode1, mode2) for the broadcast forms planer graphs use most:
    equal shapes, a one-element operand, or a (1|-,C,1,...) per-channel operand; None for any other
    pair (general numpy broadcasting, `_binary_general`)."""
    def mode(a, ref):
        if a.shape == ref.shape:
            return 0
        if a.size == 1:
            return 2
        nd = ref.ndim
        shp = (1,) * (nd - a.ndim) + a.shape
        if nd >= 2 and len(shp) == nd and shp[1] == ref.shape[1] and a.size == ref.shape[1]:
            return 1
        return None
    ref = x1 if x1.size >= x2.size else x2
    if tuple(numpy.broadcast_shapes(tuple(x1.shape), tuple(x2.shape))) != tuple(ref.shape):
        return None
    m1, m2 = mode(x1, ref), mode(x2, ref)
    if m1 is None or m2 is None:
        return None
    c = ref.shape[1] if ref.ndim >= 2 else 1
    outer = ref.shape[0] if ref.ndim >= 2 else 1
    inner = ref.size // (outer * c) if ref.size else 1
    return ref.shape, outer, c, max(inner, 1), m1, m2


def _binary_general(x1, x2, op):
    """x1 (op) x2 under numpy's broadcasting rules (layer.py:93-111 rely on them): result axes that both
    operands walk the same way are merged, what is left must fit pl_binary_bcast_f32's six axes."""
    out = tuple(numpy.broadcast_shapes(tuple(x1.shape), tuple(x2.shape)))     # raises ValueError like numpy
    nd = len(out)
    y = empty(out, ctx=x1.ctx)
    if not y.size:
        return y
    def strides(a):
        shp = (1,) * (nd - a.ndim) + tuple(a.shape)
        st = _contig_strides(shp)
        return [0 if shp[d] == 1 else st[d] for d in range(nd)]
    sa, sb = strides(x1), strides(x2)
    dims = [[out[d], sa[d], sb[d]] for d in range(nd) if out[d] != 1] or [[1, 0, 0]]
    merged = [dims[0]]
    for n, a, b in dims[1:]:
        m = merged[-1]
        if m[1] == a * n and m[2] == b * n:       # the outer axis steps over exactly the inner one's extent
            merged[-1] = [m[0] * n, a, b]
        else:
            merged.append([n, a, b])
    if len(merged) > 6:
        raise NotImplementedError("broadcast %s with %s needs more than 6 axes on the HIP path" % (x1.shape, x2.shape))
    k = len(merged)
    _lib.call("pl_binary_bcast_f32", x1.ctx.handle, x1.ptr, x2.ptr, y.ptr, k,
              (_lib.c_int * k)(*[m[0] for m in merged]), (_lib.c_longlong * k)(*[m[1] for m in merged]),
              (_lib.c_longlong * k)(*[m[2] for m in merged]), op)
    return y


def _binary(x1, x2, op):
    _f32(x1, x2)
    forms = _broadcast_modes(x1, x2)
    if forms is None:
        return _binary_general(x1, x2, op)
    shape, outer, c, inner, m1, m2 = forms
    y = empty(shape, ctx=x1.ctx)
    if y.size:
        _lib.call("pl_binary_f32", x1.ctx.handle, x1.ptr, x2.ptr, y.ptr, outer, c, inner, op, m1, m2)
    return y


def Sub(x1, x2):
    """layer.Sub (layer.py:97-99)"""
    return _binary(x1, x2, 1)


def Mul(x1, x2):
    """layer.Mul (layer.py:101-103)"""
    return _binary(x1, x2, 2)


def Div(x1, x2):
    """layer.Div (layer.py:105-107)"""
    return _binary(x1, x2, 3)


def Pow(x, p):
    """layer.Pow (layer.py:109-111)"""
    return _binary(x, p, 4)


def _unary(x, op, p0=0.0, p1=0.0, inplace=False):
    _f32(x)
    y = x if inplace else empty(x.shape, ctx=x.ctx)
    _lib.call("pl_unary_f32", x.ctx.handle, x.ptr, y.ptr, x.size, op, float(p0), float(p1))
    return y


def Exp(x):
    """layer.Exp (layer.py:178-180)"""
    return _unary(x, 0)


def Log(x):
    """layer.Log (layer.py:182-184)"""
    return _unary(x, 1)


def Tanh(x):
    """layer.Tanh (layer.py:174-176)"""
    return _unary(x, 2)


def Sqrt(x):
    """layer.Sqrt (layer.py:53)"""
    return _unary(x, 3)


def Reciprocal(x):
    """layer.Reciprocal (layer.py:186)"""
    return _unary(x, 4)


def HardSigmoid(x, alpha=0.2, beta=0.5):
    """layer.HardSigmoid (layer.py:66-69): clip(x*alpha + beta, 0, 1)"""
    return _unary(x, 5, alpha, beta)


def Clip(x, min=0, max=1):
    """layer.Clip (layer.py:247-251), numpy branch: IN PLACE on x like the reference."""
    return _unary(x, 6, min, max, inplace=True)


def _softmax_any_axis(x, axis, log):
    _f32(x)
    nd = x.ndim
    axis = axis + nd if axis < 0 else axis
    if not 0 <= axis < nd:
        raise ValueError("softmax: axis out of range")
    if axis != nd - 1:                                  # move the axis last, reduce, move it back
        perm = [d for d in range(nd) if d != axis] + [axis]
        inv = [perm.index(d) for d in range(nd)]
        return Transpose(_softmax_any_axis(Transpose(x, perm), -1, log), inv)
    cols = x.shape[-1]
    y = empty(x.shape, ctx=x.ctx)
    _lib.call("pl_softmax_f32", x.ctx.handle, x.ptr, y.ptr, (x.size // cols if cols else 0), cols, log)
    return y


def Softmax(x, axis=-1):
    """layer.Softmax (layer.py:141-146)"""
    return _softmax_any_axis(x, axis, 0)


def LogSoftmax(x, axis=-1):
    """layer.LogSoftmax (layer.py:148-153)"""
    return _softmax_any_axis(x, axis, 1)


def _reduce(x, axes, keepdims, op):
    """ReduceSum/Mean/Max/Min (layer.py:113-123): x.<op>(axis=tuple(axes), keepdims).  The kernel reduces a
    trailing block of axes; any other set is moved there by one transpose first."""
    _f32(x)
    nd = x.ndim
    axes = sorted(set(a + nd if a < 0 else a for a in tuple(axes)))      # tuple(axes) as in the reference
    if any(not 0 <= a < nd for a in axes):
        raise ValueError("reduction: axis out of range")
    kept = [d for d in range(nd) if d not in axes]
    src = x if axes == list(range(nd - len(axes), nd)) else Transpose(x, kept + axes)
    cols = int(numpy.prod([x.shape[a] for a in axes], dtype=numpy.int64))
    rows = x.size // cols if cols else 0
    y = empty(tuple(x.shape[d] for d in kept), ctx=x.ctx)
    _lib.call("pl_reduce_f32", x.ctx.handle, src.ptr, y.ptr, rows, max(cols, 1), op)
    if keepdims:
        y = y.reshape([1 if d in axes else x.shape[d] for d in range(nd)])
    return y


def ReduceSum(x, axes=-1, keepdims=True):
    return _reduce(x, axes, keepdims, 0)


def ReduceMean(x, axes=-1, keepdims=True):
    return _reduce(x, axes, keepdims, 1)


def ReduceMax(x, axes=-1, keepdims=True):
    return _reduce(x, axes, keepdims, 2)


def ReduceMin(x, axes=-1, keepdims=True):
    return _reduce(x, axes, keepdims, 3)


def Transpose(x, axis):
    """layer.Transpose (layer.py:194): x.transpose(axis), materialised contiguous."""
    _f32(x)
    perm = [int(a) for a in axis]
    if sorted(perm) != list(range(x.ndim)):
        raise ValueError("transpose: bad permutation %s" % (perm,))
    y = empty(tuple(x.shape[a] for a in perm), ctx=x.ctx)
    n = x.ndim
    shp = (_lib.c_int * n)(*x.shape)
    prm = (_lib.c_int * n)(*perm)
    _lib.call("pl_transpose_f32", x.ctx.handle, x.ptr, y.ptr, n, shp, prm)
    return y


def Reshape(x, shp):
    """layer.Reshape (layer.py:188-192): a 0 in `shp` keeps that input dim; a view."""
    shp = [int(v) for v in _host_values(shp).tolist()]
    for i in range(len(shp)):
        shp[i] = shp[i] or x.shape[i]
    return x.reshape(shp)


def Squeeze(x, axes=[0]):
    """layer.Squeeze (layer.py:133-134): np.squeeze(x, axis=axes[0])"""
    a = axes[0] + x.ndim if axes[0] < 0 else axes[0]
    if x.shape[a] != 1:
        raise ValueError("cannot select an axis to squeeze out which has size not equal to one")
    return x.reshape(x.shape[:a] + x.shape[a + 1:])


def Unsqueeze(x, axes=None):
    """layer.Unsqueeze (layer.py:129-131): np.expand_dims(x, tuple(axes))"""
    axes = numpy.array(axes).tolist()
    axes = [axes] if isinstance(axes, int) else list(axes)
    nd = x.ndim + len(axes)
    axes = sorted(a + nd if a < 0 else a for a in axes)
    it, shape = iter(x.shape), []
    for d in range(nd):
        shape.append(1 if d in axes else next(it))
    return x.reshape(shape)


def _nearest_shift(k, trans_mode, round_mode):
    """util.offset (util.py:155-170): the source index of output index 0 under the coordinate transform and the rounding
    rule, found by probing the integers -64 .. 63 (host arithmetic on 128 numbers; names the reference does not know
    apply nothing, the int16 cast truncates)."""
    pos = numpy.arange(-64, 64)
    if trans_mode == "half_pixel":
        pos = (pos + 0.5) / k - 0.5
    if trans_mode == "asymmetric":
        pos = pos / k
    if round_mode == "round_prefer_floor":
        pos = numpy.round(pos - 1e-3)
    if round_mode == "round_prefer_ceil":
        pos = numpy.round(pos + 1e-3)
    if round_mode == "ceil":
        pos = numpy.ceil(pos)
    if round_mode == "floor":
        pos = numpy.floor(pos)
    return int(numpy.argmax(pos.astype(numpy.int16) == 0)) - 64


_NEAREST_MAPS = {}


def _shifted_nearest_map(ctx, h, w, fh, fw, dr, dc):
    """Gather map of util.upsample_nearest + util.pix_offset (util.py:172-192) for one plane: output pixel (r, c) of the
    (h fh) x (w fw) map -> flat index of its source pixel in the h x w plane.  The replicated map moves by (dr, dc); the
    vacated rows take row 0 / H-1 of the UNMOVED map and the vacated columns its column 0 / W-1 (the reference assigns the
    interior, the rows and the columns one after the other), so a vacated row keeps its columns unmoved and a vacated column
    its rows.  Built once per geometry on the host (an int32 per output pixel), kept on the device."""
    key = (id(ctx), h, w, fh, fw, dr, dc)
    m = _NEAREST_MAPS.get(key)
    if m is None:
        H, W = h * fh, w * fw
        rows, cols = numpy.arange(H), numpy.arange(W)
        r_in = (rows >= dr) if dr >= 0 else (rows < H + dr)
        c_in = (cols >= dc) if dc >= 0 else (cols < W + dc)
        r_edge, c_edge = (0 if dr >= 0 else H - 1), (0 if dc >= 0 else W - 1)
        R = numpy.where(r_in[:, None], numpy.where(c_in[None, :], (rows - dr)[:, None], rows[:, None]), r_edge)
        C = numpy.where(c_in[None, :], numpy.where(r_in[:, None], (cols - dc)[None, :], cols[None, :]), c_edge)
        m = _NEAREST_MAPS[key] = asarray(((R // fh) * w + C // fw).astype(numpy.int32).reshape(-1), ctx=ctx)
    return m


def Resize(x, roi, k, size=None, mode="nearest", coordinate_transformation_mode="half_pixel",
           nearest_mode="round_prefer_floor"):
    """layer.Resize (layer.py:84-88) -> util.upsample (util.py:212-219).  Nearest: replication by the truncated factors
    (util.py:213) and the shift util.offset() derives from the two mode names -- zero for (half_pixel, round_prefer_*) and
    (asymmetric, floor), where this is UpSample's kernel; any other pair goes through a gather map that reproduces
    util.pix_offset's border rule (`_shifted_nearest_map`).  Linear: the reference ignores the two mode arguments
    (util.py:216-218), so does this."""
    if mode not in ("nearest", "linear"):
        raise NotImplementedError("resize mode %r is not on the HIP path" % mode)
    kv = _host_values(k)
    if kv.size == 0:
        sz = _host_values(size)
        kv = sz[-2:] / numpy.array(x.shape[-2:])
    fh, fw = [float(v) for v in kv[-2:].tolist()]
    if mode == "linear":
        _f32(x)
        return _upsample_linear(x, fh, fw)
    fh, fw = int(fh), int(fw)                      # util.py:213
    if fh < 1 or fw < 1:
        raise NotImplementedError("resize: nearest down-scaling (the reference returns an empty map) is not on the HIP path")
    dr = _nearest_shift(fh, coordinate_transformation_mode, nearest_mode)
    dc = _nearest_shift(fw, coordinate_transformation_mode, nearest_mode)
    if dr == 0 and dc == 0:
        return UpSample(x, numpy.array([1, 1, fh, fw], numpy.float32))
    _f32(x)
    n, c, h, w = x.shape
    y = empty((n, c, h * fh, w * fw), ctx=x.ctx)
    if y.size:
        m = _shifted_nearest_map(x.ctx, h, w, fh, fw, dr, dc)
        _lib.call("pl_gather_f32", x.ctx.handle, x.ptr, m.ptr, y.ptr, n * c, h * w, 1, h * fh * w * fw)
    return y


# ---- structural operators on the strided-map kernel ------------------------------
def _contig_strides(shape):
    st, acc = [], 1
    for d in reversed(shape):
        st.append(acc)
        acc *= d
    return st[::-1]


def _strided_map(x, out_shape, in_stride, start, step, div=None, extent=None, wrap=None, fill=0.0):
    """y[o] = x[sum_d t_d*in_stride[d]], t_d = o_d*step[d] + start[d] (see pl_strided_map_f32)."""
    _f32(x)
    n = len(out_shape)
    if n > 6:
        raise NotImplementedError("more than 6 axes are not on the HIP path")
    if n == 0:
        out_shape, in_stride, start, step = [1], [0], [0], [0]
        div, extent, wrap, n = None, [1], None, 1
        scalar = True
    else:
        scalar = False
    div = div or [1] * n
    wrap = wrap or [0] * n
    ci = _lib.c_int * n
    y = empty(tuple(int(v) for v in out_shape), ctx=x.ctx)
    _lib.call("pl_strided_map_f32", x.ctx.handle, x.ptr, y.ptr, n, ci(*[int(v) for v in out_shape]),
              (_lib.ctypes.c_longlong * n)(*[int(v) for v in in_stride]), ci(*[int(v) for v in start]),
              ci(*[int(v) for v in step]), ci(*[int(v) for v in div]), ci(*[int(v) for v in extent]),
              ci(*[int(v) for v in wrap]), float(fill))
    return y.reshape(()) if scalar else y


def Slice(x, start, end, axis=None, step=None):
    """layer.Slice (layer.py:188-196): x[tuple(slices)] with Python slice semantics (negative
    indices, clamping, negative steps); materialised contiguous."""
    start = _host_values(start).tolist()
    end = _host_values(end).tolist()
    axis = list(range(len(start))) if axis is None else _host_values(axis).tolist()
    step = [1] * len(start) if step is None else _host_values(step).tolist()
    nd = x.ndim
    sl = [slice(None, None, None)] * nd
    for s_, e_, a_, st_ in zip(start, end, axis, step):
        sl[int(a_)] = slice(int(s_), int(e_), int(st_))
    rng = [range(*sl[d].indices(x.shape[d])) for d in range(nd)]
    return _strided_map(x, [len(r) for r in rng], _contig_strides(x.shape),
                        [r.start for r in rng], [r.step for r in rng], extent=list(x.shape))


_PAD_MODES = {"constant": 0, "wrap": 1, "edge": 2, "reflect": 3, "symmetric": 4}


def Pad(x, pads, constant_value=0, mode="constant"):
    """layer.Pad (layer.py:241-245): np.pad(x, pads.reshape(2,-1).T, mode) -- constant (with its value), edge, reflect,
    symmetric and wrap are index maps of the strided-map kernel (the padding index folded back into the axis: clamped,
    mirrored without / with the border sample, modulo); np.pad's statistical modes are not on the HIP path."""
    if mode not in _PAD_MODES:
        raise NotImplementedError("pad mode %r is not on the HIP path" % mode)
    pv = _host_values(pads).reshape(2, -1).T.astype(int).tolist()
    if len(pv) != x.ndim or any(b < 0 or a < 0 for b, a in pv):
        raise ValueError("pad: need one non-negative (before, after) pair per axis")
    if mode != "constant" and any(x.shape[d] == 0 and (pv[d][0] or pv[d][1]) for d in range(x.ndim)):
        raise ValueError("can't extend empty axis using modes other than 'constant' or 'empty'")      # np.pad's refusal
    out = [x.shape[d] + pv[d][0] + pv[d][1] for d in range(x.ndim)]
    return _strided_map(x, out, _contig_strides(x.shape), [-pv[d][0] for d in range(x.ndim)], [1] * x.ndim,
                        extent=list(x.shape), wrap=[_PAD_MODES[mode]] * x.ndim, fill=float(constant_value))


def Tile(x, repeat):
    """layer.Tile (layer.py:57): np.tile(x, repeat)."""
    rep = [int(v) for v in _host_values(repeat).tolist()]
    nd = max(len(rep), x.ndim)
    rep = [1] * (nd - len(rep)) + rep
    shp = (1,) * (nd - x.ndim) + tuple(x.shape)
    return _strided_map(x, [shp[d] * rep[d] for d in range(nd)], _contig_strides(shp), [0] * nd, [1] * nd,
                        extent=list(shp), wrap=[1] * nd)


def Expand(x, shp):
    """layer.Expand (layer.py:198-200): np.ones(shp) * x, i.e. numpy broadcasting of both shapes."""
    want = [int(v) for v in _host_values(shp).tolist()]
    out = list(numpy.broadcast_shapes(tuple(want), tuple(x.shape)))
    nd = len(out)
    xs = (1,) * (nd - x.ndim) + tuple(x.shape)
    st = _contig_strides(xs)
    stride = [0 if xs[d] == 1 else st[d] for d in range(nd)]
    return _strided_map(x, out, stride, [0] * nd, [0 if xs[d] == 1 else 1 for d in range(nd)],
                        extent=[max(v, 1) for v in xs])


def Split(x, split=None, axis=0):
    """layer.Split (layer.py:170-172): np.split(x[:seg[-1]], seg[:-1], axis) -- the leading slice is
    along axis 0 whatever `axis` is, as in the reference."""
    seg = numpy.cumsum(numpy.array(split)).tolist()
    lead = min(int(seg[-1]), x.shape[0])
    shp = (lead,) + tuple(x.shape[1:])
    axis = axis + x.ndim if axis < 0 else axis
    bounds = [0] + [int(v) for v in seg[:-1]] + [shp[axis]]
    outs = []
    for lo, hi in zip(bounds, bounds[1:]):
        lo, hi = min(lo, shp[axis]), min(hi, shp[axis])
        out = list(shp)
        out[axis] = max(hi - lo, 0)
        start = [0] * x.ndim
        start[axis] = lo
        outs.append(_strided_map(x, out, _contig_strides(x.shape), start, [1] * x.ndim, extent=list(x.shape)))
    return outs


def ConvTranspose2d(x, K, B=None, strides=[2, 2], dilations=[1, 1], pads=[0, 0, 0, 0], output_padding=[0, 0],
                    group=1):
    """layer.ConvTranspose2d (layer.py:28-34): scatter x into a zero-stuffed buffer (one strided-map
    launch), flip + transpose the filter (one launch), then the stride-1 MFMA convolution."""
    _f32(x, K, B)
    if group != 1:
        raise NotImplementedError("convtranspose with group > 1: the reference's filter transpose is only "
                                  "shape-correct for group = 1 (layer.py:34)")
    n, c, h, w = x.shape
    s1, s2 = [int(v) for v in strides]
    d1, d2 = [int(v) for v in dilations]
    kh, kw = K.shape[2:]
    low_h, high_h = (kh - 1) * d1 - pads[0], (kh - 1) * d1 - pads[2] + output_padding[0]
    low_w, high_w = (kw - 1) * d2 - pads[1], (kw - 1) * d2 - pads[3] + output_padding[1]
    if min(low_h, high_h, low_w, high_w) < 0:
        raise NotImplementedError("convtranspose: pads larger than the dilated kernel reach are not on the HIP path")
    bh, bw = (h - 1) * s1 + low_h + high_h + 1, (w - 1) * s2 + low_w + high_w + 1
    buf = _strided_map(x, [n, c, bh, bw], _contig_strides(x.shape), [0, 0, -low_h, -low_w], [1, 1, 1, 1],
                       div=[1, 1, s1, s2], extent=[n, c, h, w])
    ci, co = K.shape[:2]
    kst = _contig_strides(K.shape)
    Kt = _strided_map(K, [co, ci, kh, kw], [kst[1], kst[0], kst[2], kst[3]], [0, 0, kh - 1, kw - 1], [1, 1, -1, -1],
                      extent=[co, ci, kh, kw])
    return Conv2d(buf, Kt, B, strides=[1, 1], dilations=[d1, d2])


# ---- shape-domain tensors and the operators of ONNX detection heads ---------------------------------
# ONNX exporters wrap the convolutional trunk in integer arithmetic on shapes (Shape -> Gather ->
# Unsqueeze -> Concat -> Reshape ...).  Those tensors are a handful of int64 values that steer views;
# the reference computes them with host numpy whatever the backend (layer.py:155, 202).  Here a small
# integer / bool DeviceArray carries a host mirror (`.host`), and an operator whose tensor arguments
# all have mirrors -- none of them a float32 activation -- is evaluated on the mirrors: no kernel, no
# stream synchronisation.  float32 activations never take this route.
def _mirrored(host, ctx=None):
    host = numpy.require(host, requirements="C")
    return DeviceArray(host.shape, host.dtype, ctx, host=host)      # uploaded on first use of `.ptr`, if ever


def _shape_domain(args):
    arrs = [a for a in args if isinstance(a, DeviceArray)]
    return bool(arrs) and all(a.host is not None and a.dtype != numpy.float32 for a in arrs)


def _hosts(args):
    return [a.host if isinstance(a, DeviceArray) else a for a in args]


def _wrap_host(out, ctx):
    if isinstance(out, (list, tuple)):
        return type(out)(_wrap_host(o, ctx) for o in out)
    return _mirrored(numpy.asarray(out), ctx)


def _with_shape_domain(device_fn, host_fn):
    """`device_fn` for activations; `host_fn` (numpy, the reference's own expression) when every tensor
    argument is a mirrored integer / bool tensor."""
    def op(*args, **kw):
        if _shape_domain(args):
            ctx = [a for a in args if isinstance(a, DeviceArray)][0].ctx
            return _wrap_host(host_fn(*_hosts(args), **kw), ctx)
        return device_fn(*args, **kw)
    op.__name__ = device_fn.__name__
    op.__doc__ = device_fn.__doc__
    return op


def Shape(x):
    """layer.Shape (layer.py:155): np.array(x.shape)"""
    return _mirrored(numpy.array(x.shape), x.ctx if isinstance(x, DeviceArray) else None)


def Const(value=0, dtype="float32"):
    """layer.Const (layer.py:136-139)"""
    if isinstance(value, list):
        return _mirrored(numpy.array(value, dtype=dtype))
    return value


def ConstantofShape(x, value=0, dtype="float32"):
    """layer.ConstantofShape (layer.py:167-168): np.full(x.ravel().tolist(), value, dtype)"""
    out = numpy.full(_host_values(x).ravel().tolist(), value, dtype=dtype)
    return _mirrored(out, x.ctx) if (out.dtype != numpy.float32 and out.size <= 4096) else asarray(out, ctx=x.ctx)


def Range(start, end, delta):
    """layer.Range (layer.py:202-203): np.arange(int(start), int(end), int(delta))"""
    vals = [int(numpy.asarray(_host_values(v)).reshape(-1)[0]) for v in (start, end, delta)]
    ctx = [a for a in (start, end, delta) if isinstance(a, DeviceArray)]
    return _mirrored(numpy.arange(*vals), ctx[0].ctx if ctx else None)


_CAST_CODES = {"float32": 0, "int32": 1, "int64": 2, "bool": 3}


def Cast(x, dtype="flaot32"):
    """layer.Cast (layer.py:200): x.astype(dtype); the default is the reference's own (misspelt) one."""
    if _shape_domain([x]):
        return _mirrored(x.host.astype(dtype), x.ctx)
    src, dst = str(x.dtype), str(numpy.dtype(dtype))
    if src not in _CAST_CODES or dst not in _CAST_CODES:
        raise NotImplementedError("cast %s -> %s is not on the HIP path (float32 / int32 / int64 / bool)" % (src, dst))
    y = empty(x.shape, numpy.dtype(dtype), ctx=x.ctx)
    _lib.call("pl_cast", x.ctx.handle, x.ptr, y.ptr, x.size, _CAST_CODES[src], _CAST_CODES[dst])
    return y


def _compare(op, ref):
    def cmp(a, b):
        if _shape_domain([a, b]) or not any(isinstance(t, DeviceArray) for t in (a, b)):
            ctx = [t for t in (a, b) if isinstance(t, DeviceArray)]
            return _mirrored(ref(*_hosts([a, b])), ctx[0].ctx if ctx else None)
        ctx = [t for t in (a, b) if isinstance(t, DeviceArray)][0].ctx
        a, b = [t if isinstance(t, DeviceArray) else asarray(numpy.asarray(t, numpy.float32).reshape(-1), ctx=ctx) for t in (a, b)]
        _f32(a, b)
        shape = a.shape if a.size >= b.size else b.shape
        n = max(a.size, b.size)
        if not (a.size in (1, n) and b.size in (1, n)) or (a.size == b.size and a.shape != b.shape):
            raise NotImplementedError("comparison: operands must have one shape or one of them a single value")
        y = empty(shape, numpy.bool_, ctx=ctx)
        _lib.call("pl_compare_f32", ctx.handle, a.ptr, b.ptr, y.ptr, n, op, int(a.size == 1 and n > 1), int(b.size == 1 and n > 1))
        return y
    return cmp


Equal = _compare(0, numpy.equal)                       # layer.py:204
Greater = _compare(1, numpy.greater)                   # layer.py:228
GreaterOrEqual = _compare(2, lambda a, b: a >= b)      # layer.py:232


def Where(msk, x1, x2):
    """layer.Where (layer.py:206): np.where(msk, x1, x2); bool mask of the result's shape, x1 / x2 of
    that shape or single values."""
    if _shape_domain([msk, x1, x2]):
        return _mirrored(numpy.where(*_hosts([msk, x1, x2])), msk.ctx)
    ctx = msk.ctx
    ops = [t if isinstance(t, DeviceArray) else asarray(numpy.asarray(t, numpy.float32).reshape(-1), ctx=ctx) for t in (x1, x2)]
    ops = [Cast(t, "float32") if t.dtype != numpy.float32 else t for t in ops]
    if msk.dtype != numpy.bool_:
        raise TypeError("where: the mask must be a bool tensor")
    n = msk.size
    if any(t.size not in (1, n) for t in ops):
        raise NotImplementedError("where: operands must have the mask's size or be single values")
    y = empty(msk.shape, ctx=ctx)
    _lib.call("pl_where_f32", ctx.handle, msk.ptr, ops[0].ptr, ops[1].ptr, y.ptr, n, int(ops[0].size == 1 and n > 1),
              int(ops[1].size == 1 and n > 1))
    return y


def Gather(x, idx, axis=0):
    """layer.Gather (layer.py:157): np.take(x, idx, axis=axis)"""
    if _shape_domain([x, idx]) or (_shape_domain([x]) and not isinstance(idx, DeviceArray)):
        return _mirrored(numpy.take(x.host, idx.host if isinstance(idx, DeviceArray) else idx, axis=axis), x.ctx)
    _f32(x)
    iv = numpy.asarray(_host_values(idx))
    axis = axis + x.ndim if axis < 0 else axis
    alen = x.shape[axis]
    if iv.size and (iv.min() < -alen or iv.max() >= alen):
        raise IndexError("gather: index out of bounds for axis %d with size %d" % (axis, alen))
    outer = int(numpy.prod(x.shape[:axis], dtype=numpy.int64))
    inner = int(numpy.prod(x.shape[axis + 1:], dtype=numpy.int64))
    y = empty(x.shape[:axis] + iv.shape + x.shape[axis + 1:], ctx=x.ctx)
    if y.size:
        di = asarray(iv.astype(numpy.int32).reshape(-1), ctx=x.ctx)
        _lib.call("pl_gather_f32", x.ctx.handle, x.ptr, di.ptr, y.ptr, outer, alen, max(inner, 1), iv.size)
    return y


_ERF_LUT = {}


def Erf(x):
    """layer.Erf (layer.py:253-258): the reference's table lookup -- erf(i/256 - 2) at 1025 points, index
    from x clamped to [-2, 2] by mask multiplications, which overwrite x IN PLACE like the reference."""
    _f32(x)
    key = id(x.ctx)
    if key not in _ERF_LUT:
        from math import erf
        _ERF_LUT[key] = asarray(numpy.array([erf(i / 256 - 2) for i in range(1025)], numpy.float32), ctx=x.ctx)
    y = empty(x.shape, ctx=x.ctx)
    _lib.call("pl_erf_lut_f32", x.ctx.handle, x.ptr, _ERF_LUT[key].ptr, y.ptr, x.size)
    return y


def InstanceNormalization(x, s, bias, epsilon=1e-5):
    """layer.InstanceNormalization (layer.py:214-224): normalises x IN PLACE over its spatial axes and
    returns it, like the reference."""
    _f32(x, s, bias)
    c = x.shape[1]
    if s.size != c or bias.size != c:
        raise ValueError("instancenormalization: one scale / bias value per channel")
    inner = int(numpy.prod(x.shape[2:], dtype=numpy.int64))
    _lib.call("pl_instancenorm_f32", x.ctx.handle, x.ptr, s.ptr, bias.ptr, x.shape[0] * c, c, max(inner, 1), float(epsilon))
    return x


def Scatternd(data, indices, updates):
    """layer.Scatternd (layer.py:208-212): a copy of `data` with data[tuple(indices[0, i])] = updates[0, i]
    applied for i = 0, 1, ... -- only the first batch entry of indices / updates is used, as in the
    reference.  The index tuples (a small integer tensor) are resolved on the host to row offsets, keeping
    the LAST update of a row (the reference's loop order); the rows are written by one kernel."""
    _f32(data, updates)
    iv = numpy.asarray(_host_values(indices))
    if iv.ndim < 2:
        raise IndexError("scatternd: indices need at least 2 dimensions, got shape %s" % (iv.shape,))
    if len(iv[0]) == 0:
        return data.copy()                                            # the reference's loop body never runs
    idx = iv[0].reshape(len(iv[0]), -1).astype(numpy.int64)          # (n, k)
    n, k = idx.shape
    if k > data.ndim:
        raise IndexError("too many indices for array: array is %d-dimensional, but %d were indexed" % (data.ndim, k))
    dims = numpy.array(data.shape[:k], numpy.int64).reshape(1, k)
    if n and ((idx < -dims) | (idx >= dims)).any():
        raise IndexError("scatternd: index out of bounds for shape %s" % (data.shape[:k],))
    idx = numpy.where(idx < 0, idx + dims, idx)
    row_len = int(numpy.prod(data.shape[k:], dtype=numpy.int64))
    out = data.copy()
    if not n or not row_len:
        return out
    if updates.ndim < 2 or updates.shape[1] < n or tuple(updates.shape[2:]) != tuple(data.shape[k:]):
        raise NotImplementedError("scatternd: updates must be (batch, n) + data.shape[%d:] (no broadcasting on the HIP path), "
                                  "got %s for data %s" % (k, updates.shape, data.shape))
    rows = numpy.ravel_multi_index(tuple(idx.T), tuple(data.shape[:k])).astype(numpy.int64) if k else numpy.zeros(n, numpy.int64)
    last = {}
    for i, r in enumerate(rows.tolist()):
        last[r] = i                                                   # later updates overwrite earlier ones
    dst = numpy.fromiter(last.keys(), numpy.int64, len(last))
    src = numpy.fromiter(last.values(), numpy.int32, len(last))
    d_dst, d_src = asarray(dst, ctx=data.ctx), asarray(src, ctx=data.ctx)
    _lib.call("pl_scatter_rows_f32", data.ctx.handle, out.ptr, d_dst.ptr, updates.ptr, d_src.ptr, len(dst), row_len)
    return out


def NonZero(x):
    """layer.NonZero (layer.py:230): np.array(np.nonzero(x)) -- (ndim, count) int64 coordinates in row-major
    order.  The count decides the output's shape, so this op waits for the stream once (4 + 4 bytes back)."""
    if _shape_domain([x]):
        return _mirrored(numpy.array(numpy.nonzero(x.host)), x.ctx)
    if not isinstance(x, DeviceArray):
        raise TypeError("expected DeviceArray, got %s (use planer_amd.asarray)" % type(x).__name__)
    code = _CAST_CODES.get(str(x.dtype))
    if code is None:
        raise NotImplementedError("nonzero of %s is not on the HIP path (float32 / int32 / int64 / bool)" % x.dtype)
    if not 1 <= x.ndim <= 8:
        raise NotImplementedError("nonzero: 1 to 8 dimensions on the HIP path, got %d" % x.ndim)
    n = x.size
    scratch = empty((n + _lib.NONZERO_BLOCK - 1) // _lib.NONZERO_BLOCK + 1, numpy.int64, ctx=x.ctx)
    total = _lib.c_longlong(0)
    _lib.call("pl_nonzero_count", x.ctx.handle, x.ptr, n, code, scratch.ptr, _lib.byref(total))
    out = empty((x.ndim, int(total.value)), numpy.int64, ctx=x.ctx)
    if total.value:
        shape = (_lib.c_longlong * x.ndim)(*x.shape)
        _lib.call("pl_nonzero_write", x.ctx.handle, x.ptr, n, code, scratch.ptr, shape, x.ndim, out.ptr, total.value)
    return out


def TopK(x, k, axis=-1, largest=1, sorted=1):
    """layer.TopK (layer.py:234-239): (values, int64 indices) of take(argsort(x, axis), arange(k) * -largest -
    (largest > 0), axis).  largest = 1: the k greatest, descending.  largest = 0: the reference's index list
    is k zeros, i.e. k copies of the smallest element -- reproduced.  One workgroup sorts a row in LDS (up to
    16384 elements; longer rows take k selection rounds).  Equal values: numpy's sort leaves their order
    unspecified, here the lower index sorts first."""
    _f32(x)
    kk = int(numpy.asarray(_host_values(k)).reshape(-1)[0])
    if largest not in (0, 1):
        raise NotImplementedError("topk: largest must be 0 or 1 on the HIP path")
    if not -x.ndim <= axis < x.ndim:
        raise ValueError("axis %d is out of bounds for array of dimension %d" % (axis, x.ndim))
    axis = axis + x.ndim if axis < 0 else axis
    n = x.shape[axis]
    if kk < 0 or kk > n:
        raise IndexError("topk: k = %d is out of bounds for axis %d with size %d" % (kk, axis, n))
    outer = int(numpy.prod(x.shape[:axis], dtype=numpy.int64))
    inner = int(numpy.prod(x.shape[axis + 1:], dtype=numpy.int64))
    shape = x.shape[:axis] + (kk,) + x.shape[axis + 1:]
    vals, idx = empty(shape, ctx=x.ctx), empty(shape, numpy.int64, ctx=x.ctx)
    if vals.size:
        _lib.call("pl_topk_f32", x.ctx.handle, x.ptr, outer, n, inner, kk, int(largest), vals.ptr, idx.ptr)
    return vals, idx


def LSTM(X, W, R, B=0, sequence_lens=0, initial_h=0, initial_c=0, hidden_size=None, direction="forward"):
    """layer.LSTM (layer.py:36-42) over util.lstm (util.py:102-119): X (L, N, D), W (dirs, 4H, D), R (dirs, 4H, H),
    B (dirs, 8H), initial_h / initial_c (dirs, N, H); gates in ONNX order i, o, f, c; sequence_lens is ignored
    there too.  x_t W^T for every step is ONE MFMA GEMM per direction; a step is then h R^T (GEMM) + the cell
    kernel.  Returns (Y (L, dirs, N, H), H, C) with the reference's shapes: H (N, H) and C (1, N, H) of the
    LAST direction only."""
    dirs = {"forward": [1], "reverse": [-1], "bidirectional": [1, -1]}[direction]
    _f32(X, W, R)
    for name, t in (("B", B), ("initial_h", initial_h), ("initial_c", initial_c)):
        if not isinstance(t, DeviceArray):
            raise TypeError("lstm: %s must be given (the reference indexes it per direction, layer.py:41)" % name)
    _f32(B, initial_h, initial_c)
    L, N, D = X.shape
    H = R.shape[-1]
    from .hip import zeros
    Y = zeros((L, len(dirs), N, H), ctx=X.ctx)
    X2 = X.reshape(L * N, D)
    ht = ct = None
    for i, d in enumerate(dirs):
        gx = Dense(X2, W[i], None) if L else None                   # (L*N, 4H): every step's x_t W^T
        ht, ct = initial_h[i], initial_c[i]
        for t in list(range(L))[::d]:
            gh = Dense(ht, R[i], None)
            h_new, c_new = Y[t][i], empty((1, N, H), ctx=X.ctx)
            _lib.call("pl_lstm_cell_f32", X.ctx.handle, gx.rows(t * N, (t + 1) * N).ptr, gh.ptr, B[i].ptr, ct.ptr, h_new.ptr,
                      c_new.ptr, N, H)
            ht, ct = h_new, c_new
    return Y, ht.copy(), ct


def _missing(kind):
    def op(*a, **k):
        raise NotImplementedError(
            "planer op %r has no HIP kernel in planer_amd yet (no BASELINE config uses it); "
            "there is deliberately no CPU fallback" % kind)
    op.__name__ = "missing_" + kind
    return op


# every kind of the reference's layer_map (layer.py:262-281) has a device implementation
NOT_ON_DEVICE = []

layer_map = {"dense": Dense, "conv": Conv2d, "relu": ReLU, "leakyrelu": LeakyReLU,
             "batchnorm": BatchNorm, "flatten": Flatten, "sigmoid": Sigmoid,
             "maxpool": Maxpool, "averagepool": AveragePool, "upsample": UpSample,
             "concat": Concatenate, "add": Add, "gap": GlobalAveragePool, "matmul": MatMul,
             "identity": Identity, "return": Return,
             # second wave
             "sub": Sub, "mul": Mul, "div": Div, "pow": Pow, "exp": Exp, "log": Log, "tanh": Tanh,
             "sqrt": Sqrt, "reciprocal": Reciprocal, "hardsigmoid": HardSigmoid, "clip": Clip,
             "softmax": Softmax, "logsoftmax": LogSoftmax, "reducesum": ReduceSum,
             "reducemean": ReduceMean, "reducemax": ReduceMax, "reducemin": ReduceMin,
             "transpose": Transpose, "reshape": Reshape, "squeeze": Squeeze, "unsqueeze": Unsqueeze,
             "resize": Resize, "slice": Slice, "pad": Pad, "tile": Tile, "expand": Expand, "split": Split,
             "convtranspose": ConvTranspose2d,
             # detection-head operators
             "shape": Shape, "const": Const, "constantofshape": ConstantofShape, "range": Range, "cast": Cast,
             "equal": Equal, "greater": Greater, "greaterorequal": GreaterOrEqual, "where": Where, "gather": Gather,
             "erf": Erf, "instancenormalization": InstanceNormalization,
             "scatternd": Scatternd, "nonzero": NonZero, "topk": TopK, "lstm": LSTM,
             # plan-compiler internal
             "conv_fused": ConvFused}
# integer shape arithmetic: the same kinds, evaluated on host mirrors when no activation is involved
for _k, _ref in (("add", lambda a, b: a + b), ("sub", lambda a, b: a - b), ("mul", lambda a, b: a * b),
                 ("div", lambda a, b: a / b), ("concat", lambda *xs, axis=0: numpy.concatenate(xs, axis=axis)),
                 ("slice", None), ("squeeze", lambda x, axes=[0]: numpy.squeeze(x, axis=axes[0])),
                 ("unsqueeze", lambda x, axes=None: numpy.expand_dims(x, tuple(numpy.array(axes).tolist()))),
                 ("reducesum", lambda x, axes=-1, keepdims=True: x.sum(axis=tuple(axes), keepdims=keepdims)),
                 ("transpose", lambda x, axis: x.transpose(axis)), ("identity", lambda x: x),
                 ("expand", lambda x, shp: numpy.ones(shp.tolist(), dtype=x.dtype) * x),
                 ("tile", lambda x, repeat: numpy.tile(x, repeat))):
    if _k == "slice":
        def _ref(x, start, end, axis=None, step=None):         # layer.py:188-196
            step = numpy.ones(len(start), dtype=numpy.uint32) if step is None else step
            axis = numpy.arange(len(start)) if axis is None else axis
            sl = [slice(None, None, None)] * x.ndim
            for s_, e_, a_, st_ in zip(start.tolist(), end.tolist(), axis.tolist(), step.tolist()):
                sl[a_] = slice(s_, e_, st_)
            return x[tuple(sl)]
    layer_map[_k] = _with_shape_domain(layer_map[_k], _ref)
layer_map.update({k: _missing(k) for k in NOT_ON_DEVICE})


# ---- host arrays in, host arrays out -------------------------------------------------------------
# The reference's layer callables take whatever array type its backend module produces; a script written against it with the
# numpy backend (`planer.core(numpy)`, the import-time default: __init__.py:40) calls `Conv2d(x, K, B)` with ndarrays and
# expects ndarrays back.  Every public operator therefore accepts an all-host call: the arrays are uploaded, the HIP
# operator runs, the results come back as ndarrays -- what Net.__call__ does for a whole net (net.py:94-101).  An operator that
# works in place (ReLU returns its input, layer.py:44-46) writes the result into the caller's array and returns that array.
# Calls that carry a DeviceArray are untouched (ndarrays among them are parameters the operator reads on the host).
def _host_io(f):
    import functools

    @functools.wraps(f)
    def op(*xs, **key):
        if not xs or any(isinstance(a, DeviceArray) for a in xs) or not any(isinstance(a, numpy.ndarray) for a in xs):
            return f(*xs, **key)
        dev = [asarray(a) if isinstance(a, numpy.ndarray) else a for a in xs]
        out = f(*dev, **key)

        def back(o):
            if isinstance(o, DeviceArray):
                for a, d in zip(xs, dev):
                    if o is d and isinstance(a, numpy.ndarray) and a.shape == o.shape and a.dtype == o.dtype and a.flags.writeable:
                        a[...] = o.get()
                        return a
                return o.get()
            if isinstance(o, (tuple, list)):
                return type(o)(back(i) for i in o)
            return o
        return back(out)
    op.device_op = f
    return op


_public = {}
for _k, _f in list(layer_map.items()):
    _public[id(_f)] = layer_map[_k] = _host_io(_f)
for _name, _f in list(globals().items()):
    # the module-level names of the same operators (what `from planer_amd import *` exports): kinds whose table entry carries
    # the shape-domain branch keep the plain operator under their name, as before -- wrapped on its own
    if callable(_f) and _name[:1].isupper() and not isinstance(_f, type) and getattr(_f, "__module__", None) == __name__:
        globals()[_name] = _public.get(id(_f)) or _host_io(_f)
del _public
