"""CPU oracle for planer's per-layer forward pass.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the algorithm of the reference
(Image-Py/planer v0.34, /root/reference/planer/{layer,util,net,io}.py) for the
hot path named in BASELINE.json.  It exists so that the HIP path can be checked
against something that runs anywhere (the reference itself cannot travel to
the GPU box).  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import it; the product package `planer_amd` never does.

Parity pinning: the reference ships no tests and no golden vectors
(SURVEY.md §4), so this oracle is pinned against OUTPUTS OF THE REFERENCE
ITSELF captured in the build container by tools/capture_golden.py and
committed under tests/golden/ (tests/test_oracle_golden.py re-checks every
vector on every run).

Every function cites the reference lines it follows.  Arithmetic is float32
numpy / OpenBLAS sgemm exactly like the reference, so oracle-vs-reference
differences are at most sgemm blocking noise (~1e-6 relative).
"""
import json
import os
import time
import zipfile
from concurrent.futures import ThreadPoolExecutor
from io import BytesIO

import numpy as np

__all__ = ["conv2d", "dense", "matmul", "batchnorm", "relu", "leakyrelu",
           "sigmoid", "maxpool", "avgpool", "upsample", "concat", "add",
           "gap", "flatten", "ret", "OPS", "OracleNet", "read_net"]


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
def _zero_pad_hw(x, pads):
    """util.pad (util.py:4-10): constant zero padding of H and W.

    The reference enlarges by 2*pads[0] / 2*pads[1] (util.py:8), i.e. it only
    works for symmetric pads (top==bottom, left==right); asymmetric pads give
    silently wrong results there, so they are rejected here.
    """
    pt, pl, pb, pr = [int(p) for p in pads]
    if (pt, pl) != (pb, pr):
        raise ValueError("asymmetric pads are undefined in the reference")
    if pt == 0 and pl == 0:
        return x                                   # util.py:5
    n, c, h, w = x.shape
    out = np.zeros((n, c, h + 2 * pt, w + 2 * pl), dtype=x.dtype)
    out[:, :, pt:pt + h, pl:pl + w] = x
    return out


def _out_size(size, pad_sum, k, d, s):
    """util.py:25-26: (size + pads - (k-1)*d - 1 + s) // s."""
    return (size + pad_sum - (k - 1) * d - 1 + s) // s


# --------------------------------------------------------------------------
# conv / dense
# --------------------------------------------------------------------------
def conv2d(x, K, B=None, group=1, strides=(1, 1), dilations=(1, 1),
           pads=(0, 0, 0, 0)):
    """layer.Conv2d (layer.py:22-26) via util.conv_for (util.py:17-44).

    im2col into a (Cin, kh*kw, N, Ho, Wo) scratch (one strided slab copy per
    filter tap, issued from a 9-worker thread pool like util.py:18,34-39),
    then one sgemm (Cout x Cin*kh*kw) @ (Cin*kh*kw x N*Ho*Wo) (util.py:41-43;
    batched over `group`), result viewed back as NCHW (util.py:44 -- a
    transposed, non-contiguous view), bias added in place (layer.py:26).
    """
    sh, sw = [int(v) for v in strides]
    dh, dw = [int(v) for v in dilations]
    cout, cin_g, kh, kw = K.shape
    n, cin, h, w = x.shape
    xp = _zero_pad_hw(x, pads)
    ho = _out_size(h, pads[0] + pads[2], kh, dh, sh)
    wo = _out_size(w, pads[1] + pads[3], kw, dw, sw)
    cnhw = xp.transpose(1, 0, 2, 3)                # util.py:30
    col = np.empty((cin, kh * kw, n, ho, wo), dtype=x.dtype)

    def tap(i, r, c):
        col[:, i] = cnhw[:, :, r:r + ho * sh:sh, c:c + wo * sw:sw]

    with ThreadPoolExecutor(max_workers=9) as pool:
        futs = [pool.submit(tap, a * kw + b, a * dh, b * dw)
                for a in range(kh) for b in range(kw)]
    for f in futs:
        f.result()                                 # (reference never checks)
    if group == 1:
        out = np.matmul(K.reshape(cout, -1), col.reshape(cin * kh * kw, -1))
    else:
        out = np.matmul(K.reshape(group, cout // group, -1),
                        col.reshape(group, cin_g * kh * kw, -1))
    out = out.reshape(cout, n, ho, wo).transpose(1, 0, 2, 3)
    if B is not None:
        np.add(out, B.reshape(1, -1, 1, 1), out=out)
    return out


def dense(x, K, B, shp=None):
    """layer.Dense (layer.py:15-18): x @ K.T + B ; `shp` is ignored."""
    y = np.matmul(x, K.T)
    y += B.reshape(1, -1)
    return y


def matmul(x, y):
    """layer.MatMul (layer.py:20)."""
    return np.matmul(x, y)


# --------------------------------------------------------------------------
# elementwise
# --------------------------------------------------------------------------
def batchnorm(x, K, B):
    """layer.BatchNorm (layer.py:125-127): x*K + B, K/B pre-folded (1,C,1,1)."""
    y = x * K
    y += B
    return y


def relu(x):
    """layer.ReLU (layer.py:44-46): x *= (x>0) IN PLACE; returns x itself."""
    return np.multiply(x, x > 0, out=x)


def leakyrelu(x, alpha=0.2):
    """layer.LeakyReLU (layer.py:48-51): x*((x>0)*(1-alpha)+alpha)."""
    a = np.array(alpha, x.dtype)
    b = np.array(1 - alpha, x.dtype)
    y = (x > 0) * b
    y += a
    y *= x
    return y


def sigmoid(x):
    """layer.Sigmoid (layer.py:61-64): 1/(1+exp(-x))."""
    with np.errstate(over="ignore"):
        t = np.exp(-x)
    t += 1
    return np.divide(1, t, out=t)


def add(a, b):
    """layer.Add (layer.py:93-95)."""
    return a + b


# --------------------------------------------------------------------------
# pooling / resampling / shape
# --------------------------------------------------------------------------
def _pool(x, op, win, pads, strides, init):
    """util.pool (util.py:79-92): zero-pad, accumulator filled with `init`,
    one strided pass per window tap."""
    kh, kw = [int(v) for v in win]
    sh, sw = [int(v) for v in strides]
    n, c, h, w = x.shape
    xp = _zero_pad_hw(x, pads)
    ho = (h + pads[0] + pads[2] - kh + sh) // sh   # util.py:84
    wo = (w + pads[1] + pads[3] - kw + sw) // sw   # util.py:85
    acc = np.full((n, c, ho, wo), init, dtype=x.dtype)
    for r in range(kh):
        for q in range(kw):
            op(xp[:, :, r:r + ho * sh:sh, q:q + wo * sw:sw], acc, out=acc)
    return acc


def maxpool(x, w=(2, 2), pads=(0, 0, 0, 0), strides=(2, 2)):
    """layer.Maxpool (layer.py:71-72) -> util.maxpool (util.py:94-95):
    ZERO padding (not -inf) and an accumulator initialised to -1e4."""
    return _pool(x, np.maximum, w, pads, strides, -1e4)


def avgpool(x, w=(2, 2), pads=(0, 0, 0, 0), strides=(2, 2)):
    """layer.AveragePool (layer.py:74-75) -> util.avgpool (util.py:97-100):
    sum / (kh*kw), padding counted."""
    y = _pool(x, np.add, w, pads, strides, 0)
    y /= int(w[0]) * int(w[1])
    return y


def gap(x):
    """layer.GlobalAveragePool (layer.py:77-78)."""
    return x.mean(axis=(-2, -1), keepdims=True)


def _bilinear_table(fh, fw):
    """util.make_upmat (util.py:121-132): float16 sample fractions linspace(0.5/k, 1-0.5/k, k) per axis and
    the corner weights as float16 products, rows = (left-top, right-top, left-bottom, right-bottom); with a
    factor of 1 on one axis the two weights of the other axis."""
    fy = np.linspace(0.5 / fh, 1 - 0.5 / fh, fh, dtype=np.float16)
    fx = np.linspace(0.5 / fw, 1 - 0.5 / fw, fw, dtype=np.float16)
    if fh == 1:
        return np.vstack([1 - fx, fx])
    if fw == 1:
        return np.vstack([1 - fy, fy])
    ry, rx = fy[:, None], fx[None, :]
    return np.vstack([((1 - rx) * (1 - ry)).reshape(1, -1), (rx * (1 - ry)).reshape(1, -1),
                      ((1 - rx) * ry).reshape(1, -1), (rx * ry).reshape(1, -1)])


def upsample_bilinear(x, fh, fw):
    """util.upsample_blinear (util.py:134-153): replicate the border by one pixel along every scaled axis, take
    every 2 x 2 (or 1 x 2 / 2 x 1) neighbourhood times the weight table (a float32 x float16 matmul), interleave
    the fh x fw blocks and crop fh//2, fw//2."""
    n, c, h, w = x.shape
    if fh == 1 and fw == 1:
        return x
    p = x
    if fh > 1:
        p = np.concatenate([p[:, :, :1], p, p[:, :, -1:]], axis=2)
    if fw > 1:
        p = np.concatenate([p[:, :, :, :1], p, p[:, :, :, -1:]], axis=3)
    if fh == 1:
        corners = [p[:, :, :, :-1], p[:, :, :, 1:]]
    elif fw == 1:
        corners = [p[:, :, :-1, :], p[:, :, 1:, :]]
    else:
        corners = [p[:, :, :-1, :-1], p[:, :, :-1, 1:], p[:, :, 1:, :-1], p[:, :, 1:, 1:]]
    field = np.stack(corners, axis=-1)
    hh, ww = h + (fh > 1), w + (fw > 1)
    blocks = np.matmul(field.reshape(-1, len(corners)), _bilinear_table(fh, fw))
    out = blocks.reshape(-1, ww, fh, fw).transpose(0, 2, 1, 3).reshape(n, c, hh * fh, ww * fw)
    return out[:, :, fh // 2:h * fh + fh // 2, fw // 2:w * fw + fw // 2]


def upsample_to_size(x, size):
    """util.upsample_size (util.py:194-210): separable bilinear at sample positions linspace(-0.5+0.5/k,
    n-0.5-0.5/k, size) in the image's dtype, clipped to the map; columns first, then rows."""
    lead, (h, w) = x.shape[:-2], x.shape[-2:]
    def axis(n, m):
        k = m / n
        pos = np.linspace(-0.5 + 0.5 / k, n - 0.5 - 0.5 / k, m, dtype=x.dtype)
        pos = np.clip(pos, 0, n - 1, out=pos)
        lo = np.floor(np.clip(pos, 0, n - 1.001)).astype(int)
        pos -= lo
        return lo, pos
    ra, rs = axis(h, size[0])
    ca, cs = axis(w, size[1])
    planes = x.reshape(-1, h, w)
    cols = planes[:, :, ca] * (1 - cs) + planes[:, :, ca + 1] * cs
    rs = rs.reshape(-1, 1)
    out = cols[:, ra, :] * (1 - rs) + cols[:, ra + 1, :] * rs
    return out.reshape(lead + tuple(size))


def _upsample_any(x, k, mode):
    """util.upsample (util.py:212-219), the dispatch on `mode` and on whole-number factors."""
    if mode == "linear":
        if k[0] == int(k[0]) and k[1] == int(k[1]):
            return upsample_bilinear(x, int(k[0]), int(k[1]))
        return upsample_to_size(x, (int(round(k[0] * x.shape[2])), int(round(k[1] * x.shape[3]))))
    raise NotImplementedError("oracle covers nearest and linear")


def upsample(x, k, mode="nearest"):
    """layer.UpSample (layer.py:80-82) -> util.upsample_nearest
    (util.py:184-192).  `k` is a tensor whose last two entries are the integer
    H/W factors (truncated, layer.py:82).  With UpSample's defaults util.offset() yields shift 0
    (util.py:212 passes the misspelt 'half-pixcel', so no transform applies),
    i.e. plain block replication.  mode "linear": util.upsample_blinear."""
    fh, fw = [int(v) for v in np.asarray(k)[-2:].astype(int).tolist()]
    if mode != "nearest":
        return _upsample_any(x, [fh, fw], mode)
    n, c, h, w = x.shape
    out = np.empty((n, c, h * fh, w * fw), dtype=x.dtype)
    for r in range(fh):
        for q in range(fw):
            out[:, :, r::fh, q::fw] = x
    return out


def concat(*xs, axis=0):
    """layer.Concatenate (layer.py:90-91); default axis is 0 as there."""
    return np.concatenate(xs, axis=axis)


def flatten(x):
    """layer.Flatten (layer.py:59)."""
    return x.reshape((x.shape[0], -1))


def ret(*x):
    """layer.Return (layer.py:260)."""
    return x


# --------------------------------------------------------------------------
# second-wave operators
# --------------------------------------------------------------------------
def sub(a, b):
    """layer.Sub (layer.py:97-99)"""
    return a - b


def mul(a, b):
    """layer.Mul (layer.py:101-103)"""
    return a * b


def div(a, b):
    """layer.Div (layer.py:105-107)"""
    return a / b


def power(x, p):
    """layer.Pow (layer.py:109-111)"""
    return np.power(x, p)


def hardsigmoid(x, alpha=0.2, beta=0.5):
    """layer.HardSigmoid (layer.py:66-69)"""
    y = x * alpha
    y += beta
    y = np.minimum(y, 1, out=y)
    return np.maximum(y, 0, out=y)


def clip(x, min=0, max=1):
    """layer.Clip (layer.py:247-251), numpy branch: in place, returns x."""
    x = np.minimum(x, max, out=x)
    return np.maximum(x, min, out=x)


def softmax(x, axis=-1):
    """layer.Softmax (layer.py:141-146)"""
    y = x - np.max(x, axis=axis, keepdims=True)
    s = np.sum(np.exp(y), axis=axis, keepdims=True)
    y -= np.log(s, out=s)
    return np.exp(y, out=y)


def logsoftmax(x, axis=-1):
    """layer.LogSoftmax (layer.py:148-153)"""
    y = x - np.max(x, axis=axis, keepdims=True)
    s = np.sum(np.exp(y), axis=axis, keepdims=True)
    y -= np.log(s, out=s)
    return y


def _reducer(fn):
    def op(x, axes=-1, keepdims=True):
        """layer.Reduce* (layer.py:113-123): axis=tuple(axes)"""
        return fn(x, axis=tuple(axes), keepdims=keepdims)
    return op


def reshape(x, shp):
    """layer.Reshape (layer.py:188-192): 0 keeps the input dim."""
    shp = shp.tolist()
    for i in range(len(shp)):
        shp[i] = shp[i] or x.shape[i]
    return x.reshape(shp)


def unsqueeze(x, axes=None):
    """layer.Unsqueeze (layer.py:129-131)"""
    return np.expand_dims(x, tuple(np.array(axes).tolist()))


def nearest_shift(k, trans_mode, round_mode):
    """util.offset (util.py:155-170): where output index 0 comes from under the coordinate transform and the rounding
    rule, probed on the integers -64 .. 63 -- the first one that maps to source index 0.  Unknown names apply nothing
    (a transform of its own for every name but 'half_pixel' / 'asymmetric' does not exist; an unknown rounding name
    leaves the int16 cast's truncation)."""
    pos = np.arange(-64, 64)
    if trans_mode == "half_pixel":
        pos = (pos + 0.5) / k - 0.5
    if trans_mode == "asymmetric":
        pos = pos / k
    if round_mode == "round_prefer_floor":
        pos = np.round(pos - 1e-3)
    if round_mode == "round_prefer_ceil":
        pos = np.round(pos + 1e-3)
    if round_mode == "ceil":
        pos = np.ceil(pos)
    if round_mode == "floor":
        pos = np.floor(pos)
    return int(np.argmax(pos.astype(np.int16) == 0)) - 64


def shift_with_border(img, dr, dc):
    """util.pix_offset (util.py:172-182) out of place: the map moved by (dr, dc); what moves in from outside is the border
    row / column of the UNMOVED map (three slice assignments in a row there: the interior, then the vacated rows from row 0
    or h-1 as it then stands, then the vacated columns likewise -- so a vacated row keeps its columns unmoved and a vacated
    column its rows)."""
    h, w = img.shape[-2:]
    if dr == 0 and dc == 0:
        return img
    rows, cols = np.arange(h), np.arange(w)
    r_in = (rows >= dr) if dr >= 0 else (rows < h + dr)
    c_in = (cols >= dc) if dc >= 0 else (cols < w + dc)
    r_edge, c_edge = (0 if dr >= 0 else h - 1), (0 if dc >= 0 else w - 1)
    R = np.where(r_in[:, None], np.where(c_in[None, :], (rows - dr)[:, None], rows[:, None]), r_edge)
    C = np.where(c_in[None, :], np.where(r_in[:, None], (cols - dc)[None, :], cols[None, :]), c_edge)
    return img[..., R, C]


def resize(x, roi, k, size=None, mode="nearest", coordinate_transformation_mode="half_pixel",
           nearest_mode="round_prefer_floor"):
    """layer.Resize (layer.py:84-88) -> util.upsample (util.py:212-219).  Nearest: block replication by the TRUNCATED
    factors (util.py:213), then the shift util.offset() derives from the two mode names (util.py:184-192).  Linear: the
    two mode arguments are not looked at (util.py:216-218)."""
    if k.size == 0:
        k = size[-2:] / np.array(x.shape[-2:])
    k = np.asarray(k)[-2:].tolist()
    if mode == "linear":
        return _upsample_any(x, k, mode)
    if mode != "nearest":
        raise NotImplementedError("oracle covers nearest and linear")
    kint = [int(k[0]), int(k[1])]
    out = upsample(x, np.array(kint), "nearest")
    return shift_with_border(out, nearest_shift(kint[0], coordinate_transformation_mode, nearest_mode),
                             nearest_shift(kint[1], coordinate_transformation_mode, nearest_mode))


def slice_(x, start, end, axis=None, step=None):
    """layer.Slice (layer.py:188-196)"""
    if step is None:
        step = np.ones(len(start), dtype=np.uint32)
    if axis is None:
        axis = np.arange(len(start))
    start, end, axis, step = [i.tolist() for i in (start, end, axis, step)]
    slis = [slice(None, None, None)] * x.ndim
    for s, e, a, st in zip(start, end, axis, step):
        slis[a] = slice(s, e, st)
    return x[tuple(slis)]


def pad(x, pads, constant_value=0, mode="constant"):
    """layer.Pad (layer.py:241-245)"""
    pads = pads.reshape(2, -1).T.tolist()
    para = {"mode": mode}
    if mode == "constant":
        para["constant_values"] = constant_value
    return np.pad(x, pads, **para)


def expand(x, shp):
    """layer.Expand (layer.py:198-200)"""
    return np.ones(shp.tolist(), dtype=x.dtype) * x


def split(x, split=None, axis=0):
    """layer.Split (layer.py:170-172): the leading slice is along axis 0 whatever `axis` is"""
    seg = np.cumsum(np.array(split)).tolist()
    return np.split(x[:seg[-1]], seg[:-1], axis)


def convtranspose2d(x, K, B=None, strides=[2, 2], dilations=[1, 1], pads=[0, 0, 0, 0], output_padding=[0, 0],
                    group=1):
    """layer.ConvTranspose2d (layer.py:28-34): zero-stuffed input, flipped + transposed filter,
    stride-1 Conv2d."""
    (n, c, h, w), (s1, s2), (d1, d2), (H, W) = x.shape, strides, dilations, K.shape[2:]
    low_h, high_h = ((H - 1) * d1 - pads[0]), ((H - 1) * d1 - pads[2] + output_padding[0])
    low_w, high_w = ((W - 1) * d2 - pads[1]), ((W - 1) * d2 - pads[3] + output_padding[1])
    buf = np.zeros((n, c, (h - 1) * s1 + low_h + high_h + 1, (w - 1) * s2 + low_w + high_w + 1), dtype=x.dtype)
    buf[:, :, low_h:buf.shape[2] - high_h:s1, low_w:buf.shape[3] - high_w:s2] = x
    return conv2d(buf, K.transpose(1, 0, 2, 3)[:, :, ::-1, ::-1], B, strides=[1, 1], dilations=dilations, group=group)


# ---- operators of ONNX-exported detection heads ---------------------------------------------------
def const(value=0, dtype="float32"):
    """layer.Const (layer.py:136-139)"""
    return np.array(value, dtype=dtype) if isinstance(value, list) else value


def instancenorm(x, s, bias, epsilon=1e-5):
    """layer.InstanceNormalization (layer.py:214-224): in place on x; s / bias are reshaped in place"""
    axis = tuple(range(2, x.ndim))
    mean = np.mean(x, axis=axis, keepdims=True)
    var = x - mean
    var **= 2
    var = np.mean(var, axis=axis, keepdims=True)
    s.shape = bias.shape = (-1,) + (1,) * (x.ndim - 2)
    var = (var + epsilon) ** 0.5
    x *= s / var
    x += bias - s * mean / var
    return x


_ERF_LUT = None


def erf(x):
    """layer.Erf (layer.py:253-258): table lookup at 1025 points of [-2, 2]; clobbers x like the reference"""
    global _ERF_LUT
    if _ERF_LUT is None:
        from math import erf as _erf
        _ERF_LUT = [_erf(i / 256 - 2) for i in range(1025)]
    x -= 2
    x *= x < 0
    x += 4
    x *= x > 0
    x *= 256
    return np.array(_ERF_LUT, x.dtype)[x.astype("int16")]


def lstm(X, W, R, B=0, sequence_lens=0, initial_h=0, initial_c=0, hidden_size=None, direction="forward"):
    """layer.LSTM (layer.py:36-42) + util.lstm (util.py:102-119).  X (L, N, D); per direction W (4H, D),
    R (4H, H), B (8H) = Wb | Rb, gates in ONNX order i, o, f, c.  sequence_lens is ignored; Y is
    (L, dirs, N, H); the returned H is (N, H) and C is (1, N, H) (the split keeps a leading 1), both of the
    LAST direction only -- the reference's shapes."""
    order = {"forward": [1], "reverse": [-1], "bidirectional": [1, -1]}[direction]
    L, N, _ = X.shape
    H = R.shape[-1]
    Y = np.zeros((L, len(order), N, H), dtype=X.dtype)

    def sigm(v):
        return 1 / (np.exp(-v) + 1)

    h = c = None
    for k, step in enumerate(order):
        h, c = initial_h[k], initial_c[k]
        wb, rb = B[k][:4 * H], B[k][4 * H:]
        for t in range(L)[::step]:
            g = X[t] @ W[k].T
            g = g + h @ R[k].T
            g = g + wb
            g = (g + rb)[None]
            gi, go, gf, gc = g[..., :H], g[..., H:2 * H], g[..., 2 * H:3 * H], g[..., 3 * H:]
            c = sigm(gf) * c + sigm(gi) * np.tanh(gc)
            h = (sigm(go) * np.tanh(c))[0]
            Y[t, k] = h
    return Y, h, c


def scatternd(data, indices, updates):
    """layer.Scatternd (layer.py:208-212)"""
    data = data.copy()
    for i in range(len(indices[0])):
        data[tuple(indices[0, i])] = updates[0, i]
    return data


def topk(x, k, axis=-1, largest=1, sorted=1):
    """layer.TopK (layer.py:234-239)"""
    idk = np.arange(int(np.asarray(k).reshape(-1)[0])) * -largest - (largest > 0)     # (k arrives as a 1-element tensor)
    idx = np.take(np.argsort(x, axis=axis), idk, axis=axis)
    return np.take_along_axis(x, idx, axis=axis), idx


OPS = {"shape": lambda x: np.array(x.shape), "gather": lambda x, idx, axis=0: np.take(x, idx, axis=axis),
       "const": const, "constantofshape": lambda x, value=0, dtype="float32": np.full(x.ravel().tolist(), value, dtype=dtype),
       "cast": lambda x, dtype="flaot32": x.astype(dtype),
       "range": lambda start, end, delta: np.arange(int(start), int(end), int(delta)),
       "equal": lambda a, b: np.equal(a, b), "greater": lambda a, b: np.greater(a, b), "greaterorequal": lambda a, b: a >= b,
       "where": lambda m, a, b: np.where(m, a, b), "nonzero": lambda x: np.array(np.nonzero(x)),
       "scatternd": scatternd, "topk": topk, "lstm": lstm, "erf": erf, "instancenormalization": instancenorm,
       "slice": slice_, "pad": pad, "tile": lambda x, repeat: np.tile(x, repeat.tolist()), "expand": expand,
       "split": split, "convtranspose": convtranspose2d,
       "sub": sub, "mul": mul, "div": div, "pow": power, "exp": lambda x: np.exp(x),
       "log": lambda x: np.log(x), "tanh": lambda x: np.tanh(x), "sqrt": lambda x: np.sqrt(x),
       "reciprocal": lambda x: 1 / x, "hardsigmoid": hardsigmoid, "clip": clip,
       "softmax": softmax, "logsoftmax": logsoftmax, "reducesum": _reducer(np.sum),
       "reducemean": _reducer(np.mean), "reducemax": _reducer(np.max), "reducemin": _reducer(np.min),
       "transpose": lambda x, axis: x.transpose(axis), "reshape": reshape,
       "squeeze": lambda x, axes=[0]: np.squeeze(x, axis=axes[0]), "unsqueeze": unsqueeze,
       "resize": resize, "identity": lambda x: x,
       "conv": conv2d, "dense": dense, "matmul": matmul,
       "batchnorm": batchnorm, "relu": relu, "leakyrelu": leakyrelu,
       "sigmoid": sigmoid, "add": add, "maxpool": maxpool,
       "averagepool": avgpool, "gap": gap, "upsample": upsample,
       "concat": concat, "flatten": flatten, "return": ret}


# --------------------------------------------------------------------------
# tiled large-image inference (util.py:236-348), SURVEY section 8(f) row F4
# --------------------------------------------------------------------------
def image_resize(img, size):
    """util.resize (util.py:253-269): separable bilinear on an H x W (x C) image."""
    d, (h, w) = img.ndim, img.shape[:2]
    kh, kw = size[0] / h, size[1] / w
    rs = np.linspace(-0.5 + 0.5 / kh, h - 0.5 - 0.5 / kh, size[0], dtype=np.float32)
    cs = np.linspace(-0.5 + 0.5 / kw, w - 0.5 - 0.5 / kw, size[1], dtype=np.float32)
    rs = np.clip(rs, 0, h - 1, out=rs)
    cs = np.clip(cs, 0, w - 1, out=cs)
    ra = np.floor(np.clip(rs, 0, h - 1.001)).astype(int)
    ca = np.floor(np.clip(cs, 0, w - 1.001)).astype(int)
    rs -= ra
    cs -= ca
    rb, cb = ra + 1, ca + 1
    rs.shape, cs.shape = (-1, 1, 1)[:d], (1, -1, 1)[:d]
    buf = img[:, ca] * (1 - cs) + img[:, cb] * cs
    return buf[ra, :] * (1 - rs) + buf[rb, :] * rs


def make_slice(l, w, mar):
    """util.make_slice (util.py:236-238)"""
    import math
    r = np.linspace(0, l - w, math.ceil((l - mar) / (w - mar)))
    return [slice(i, i + w) for i in r.astype(int).tolist()]


def grid_slice(H, W, h, w, mar):
    """util.grid_slice (util.py:240-242)"""
    import itertools
    return list(itertools.product(make_slice(H, h, mar), make_slice(W, w, mar)))


def tile(f, img, sample=1, glob=1, window=1024, margin=0.1):
    """util.tile (util.py:291-348) applied to `f` and one image (the decorator's body)."""
    from math import ceil
    h, w = img.shape[:2]
    img = img.astype("float32")
    ssz = list(sample) if isinstance(sample, tuple) else [int(h * sample), int(w * sample)]
    wsz = wsh = wsw = window
    if wsh > ssz[0]:
        wsh = ssz[0] = ceil(ssz[0] / glob) * glob
    if wsw > ssz[1]:
        wsw = ssz[1] = ceil(ssz[1] / glob) * glob
    if ssz != [h, w]:
        img = image_resize(img, ssz)
    mar = int(wsz * margin) if isinstance(margin, float) else margin
    rcs = grid_slice(*ssz, wsh, wsw, mar)
    rst = f(img[rcs[0]])
    k = rst.shape[0] / (rcs[0][0].stop - rcs[0][0].start)
    if len(rcs) == 1:
        return image_resize(rst, (int(h * k), int(w * k))) if ssz != [h, w] else rst

    def sk(ss):
        return (slice(int(ss[0].start * k), int(ss[0].stop * k)), slice(int(ss[1].start * k), int(ss[1].stop * k)))
    outshp = (int(img.shape[0] * k), int(img.shape[1] * k)) + rst.shape[2:]
    weights = np.zeros(rst.shape[:2], dtype="uint16")
    if rst.ndim == 3:
        weights = weights[:, :, None]
    weights += int(mar * k) + 1
    for i in range(int(mar * k), 0, -1):
        weights[i - 1, :] = weights[-i, :] = i
        weights[:, i - 1] = weights[:, -i] = i
    buf = np.zeros(outshp, dtype=np.float32)
    count = np.zeros(outshp[:2], dtype="uint16")
    if rst.ndim == 3:
        count = count[:, :, None]
    buf[sk(rcs[0])] = rst * weights
    count[sk(rcs[0])] += weights
    for i in range(1, len(rcs)):
        rst = f(img[rcs[i]])
        buf[sk(rcs[i])] += rst * weights
        count[sk(rcs[i])] += weights
    np.divide(buf, count, out=buf, casting="unsafe")
    if ssz != [h, w]:
        buf = image_resize(buf, (int(h * k), int(w * k)))
    return buf.astype(rst.dtype)


# --------------------------------------------------------------------------
# graph interpreter (net.py) and loader (io.py)
# --------------------------------------------------------------------------
class OracleNet:
    """net.Net (net.py:5-101) restated: same IR, same evaluation order, same
    liveness-based freeing and the same `__call__` unwrapping rule."""

    def __init__(self):
        self.weights, self.inits, self.input = [], [], []
        self.layer, self.flow, self.life, self.timer = [], [], {}, {}
        self._ops = {}

    def load_json(self, inputs, inits, body, flow, debug=False):
        # net.py:10-24
        self._ops = {name: (kind, OPS[kind], dict(para))
                     for name, kind, para in body}
        self.life = {}
        for i, (src, _, _) in enumerate(flow):
            for key in ([src] if isinstance(src, str) else src):
                self.life[key] = i                  # last reader, net.py:16-19
        self.weights = [np.zeros(shape, dtype=dt) for _, shape, dt in inits]
        self.input, self.inits = inputs, [i[0] for i in inits]
        self.layer, self.flow = body, flow

    def load_weights(self, blob):
        # net.py:83-88: consecutive raw bytes, in `inits` order
        raw, pos = np.asarray(blob).view(np.uint8).ravel(), 0
        for wt in self.weights:
            dst = wt.reshape(-1).view(np.uint8)
            dst[:] = raw[pos:pos + dst.size]
            pos += dst.size

    def forward(self, *xs):
        # net.py:37-72
        env = {"None": None}
        env.update(zip(self.inits, self.weights))
        env.update(zip(self.input, xs))
        out_key = None
        for i, (src, names, dst) in enumerate(self.flow):
            names = names if isinstance(names, list) else [names]
            for pos, name in enumerate(names):
                keys = src if pos == 0 else dst     # chained layers, net.py:46
                args = ([env[keys]] if isinstance(keys, str)
                        else [env.get(k) for k in keys])
                for k in set(src if isinstance(src, list) else [src]):
                    if k in env and self.life[k] <= i:
                        del env[k]                  # net.py:51-53
                kind, fn, para = self._ops[name]
                t0 = time.time()
                val = fn(*args, **para)
                if isinstance(dst, str):
                    env[dst] = val
                else:
                    env.update(zip(dst, val))       # net.py:61-62
                self.timer[kind] = self.timer.get(kind, 0) + time.time() - t0
            out_key = dst
        return env[out_key]

    def __call__(self, *xs):
        # net.py:94-101 (numpy backend: no conversions)
        if isinstance(xs[0], dict):
            xs = [xs[0][k] for k in self.input]
        rst = self.forward(*xs)
        return rst[0] if len(rst) == 1 else rst


def read_net(path):
    """io.read_net (io.py:8-34) for the .pla zip and .json+.npy forms."""
    path = path.replace(".onnx", "")
    if os.path.exists(path + ".pla"):
        with zipfile.ZipFile(path + ".pla") as z:
            base = os.path.split(path)[1]
            graph = json.loads(z.read(base + ".json"))
            blob = np.load(BytesIO(z.read(base + ".npy")))
    elif os.path.exists(path + ".json"):
        with open(path + ".json") as f:
            graph = json.load(f)
        blob = np.load(path + ".npy")
    else:
        print("model %s not found!" % path)         # io.py:30-31
        return None
    net = OracleNet()
    net.load_json(graph["input"], graph["inits"], graph["layers"],
                  graph["flow"])
    net.load_weights(blob)
    return net
