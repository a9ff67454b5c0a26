"""Tiled large-image inference around `net(x)` -- the reference's `util.tile` decorator and the
helpers it needs (`util.resize`, `make_slice`, `grid_slice`; util.py:253-269, 236-243, 291-348),
device-resident on MI355X.

Same contract as the reference: `@tile(sample, glob, window, margin)` wraps `f(img, *args)` where
`img` is an H x W (x C) image; the image is resampled, cut into overlapping windows, `f` is applied
to every window and the results are blended with border-distance weights and resampled back.
Here the image, the windows, `f`'s results and the blend buffers live in HBM: resampling, window
cuts, the weighted accumulation and the final division are HIP kernels (include/planer_hip.h:
pl_resize_hwc_f32, pl_strided_map_f32, pl_tile_accumulate_f32, pl_tile_normalise_f32); the host
only does the window geometry.  `f` receives and returns `DeviceArray`s.

One extension, because windows are independent and all of one size: with `batched=True`, `f` is
called ONCE with the stack of all windows (n, h, w[, c]) and returns the stack of results, so the
windows go through the net as one batch (the same batch-shard machinery as the forward pass).
"""
import itertools
import math

import numpy

from . import _lib
from . import hip
from .hip import DeviceArray
from .layer import _strided_map, _contig_strides, _f32


def make_slice(l, w, mar):
    """util.make_slice (util.py:236-238)"""
    r = numpy.linspace(0, l - w, math.ceil((l - mar) / (w - mar)))
    return [slice(i, i + w) for i in r.astype(int).tolist()]


def grid_slice(H, W, h, w, mar):
    """util.grid_slice (util.py:240-242)"""
    a, b = make_slice(H, h, mar), make_slice(W, w, mar)
    return list(itertools.product(a, b))


def _axis_samples(n, size):
    """Sample rows (or columns) of util.resize (util.py:256-266): float32 linspace, clip, floor."""
    k = size / n
    pos = numpy.linspace(-0.5 + 0.5 / k, n - 0.5 - 0.5 / k, size, dtype=numpy.float32)
    pos = numpy.clip(pos, 0, n - 1, out=pos)
    lo = numpy.floor(numpy.clip(pos, 0, n - 1.001)).astype(int)
    pos -= lo
    return lo.astype(numpy.int32), pos


def resize(img, size):
    """util.resize (util.py:253-269): separable bilinear resampling of an H x W (x C) device image."""
    _f32(img)
    h, w = img.shape[:2]
    c = img.size // (h * w) if img.size else 1
    oh, ow = int(size[0]), int(size[1])
    ra, rs = _axis_samples(h, oh)
    ca, cs = _axis_samples(w, ow)
    dev = [hip.asarray(a, ctx=img.ctx) for a in (ra, rs, ca, cs)]
    y = hip.empty((oh, ow) + tuple(img.shape[2:]), ctx=img.ctx)
    _lib.call("pl_resize_hwc_f32", img.ctx.handle, img.ptr, y.ptr, h, w, c, oh, ow, *[d.ptr for d in dev])
    return y


def _window(img, rc):
    """img[rows, cols] as a contiguous device array."""
    r, c = rc
    nd = img.ndim
    start = [r.start, c.start] + [0] * (nd - 2)
    out = [r.stop - r.start, c.stop - c.start] + list(img.shape[2:])
    return _strided_map(img, out, _contig_strides(img.shape), start, [1] * nd, extent=list(img.shape))


def _stack(arrays):
    out = hip.empty((len(arrays),) + arrays[0].shape, ctx=arrays[0].ctx)
    for i, a in enumerate(arrays):
        out[i].copy_from(a)
    return out


def tile(sample=1, glob=1, window=1024, margin=0.1, astype="float32", progress=print, batched=False):
    """util.tile (util.py:291-348).  sample: float factor or (h, w) size; glob: images smaller than
    the window are grown to a multiple of it; window: tile size after resampling; margin: overlap
    between windows (float = fraction of the window, int = pixels)."""
    def wrapf(f):
        def wrap(*p, **key):
            ori = p[0]
            on_host = isinstance(ori, numpy.ndarray)
            img = hip.asarray(numpy.ascontiguousarray(ori, dtype=numpy.float32)) if on_host else ori
            _f32(img)
            h, w = img.shape[:2]
            tps = {"sample", "window", "glob", "margin", "progress", "batched"}
            fp = {k: v for k, v in key.items() if k not in tps}
            tp = {k: v for k, v in key.items() if k in tps}
            ssz = tp.get("sample", sample)
            wsz = wsh = wsw = tp.get("window", window)
            gsz = tp.get("glob", glob)
            mar = tp.get("margin", margin)
            info = tp.get("progress", progress)
            stacked = tp.get("batched", batched)
            ssz = list(ssz) if isinstance(ssz, tuple) else [int(h * ssz), int(w * ssz)]
            if wsh > ssz[0]:
                wsh = ssz[0] = math.ceil(ssz[0] / gsz) * gsz
            if wsw > ssz[1]:
                wsw = ssz[1] = math.ceil(ssz[1] / gsz) * gsz
            if ssz != [h, w]:
                img = resize(img, ssz)
            if isinstance(mar, float):
                mar = int(wsz * mar)
            rcs = grid_slice(*ssz, wsh, wsw, mar)
            if len(rcs) > 1:
                info(1, len(rcs))
            if stacked:
                results = f(_stack([_window(img, rc) for rc in rcs]), *p[1:], **fp)
                rst = results[0]
            else:
                results = None
                rst = f(_window(img, rcs[0]), *p[1:], **fp)
            k = rst.shape[0] / (rcs[0][0].stop - rcs[0][0].start)
            if len(rcs) == 1:
                if ssz != [h, w]:
                    rst = resize(rst, (int(h * k), int(w * k)))
                return rst.get() if on_host else rst
            oh, ow = int(img.shape[0] * k), int(img.shape[1] * k)
            ch = rst.size // (rst.shape[0] * rst.shape[1])
            m = int(mar * k)
            buf = hip.zeros((oh, ow) + tuple(rst.shape[2:]), ctx=img.ctx)
            count = hip.zeros((oh, ow), ctx=img.ctx)
            for i, rc in enumerate(rcs):
                if i > 0:
                    info(i + 1, len(rcs))
                    rst = results[i] if stacked else f(_window(img, rc), *p[1:], **fp)
                _lib.call("pl_tile_accumulate_f32", img.ctx.handle, rst.ptr, buf.ptr, count.ptr, rst.shape[0],
                          rst.shape[1], ch, int(rc[0].start * k), int(rc[1].start * k), oh, ow, m)
            _lib.call("pl_tile_normalise_f32", img.ctx.handle, buf.ptr, count.ptr, oh, ow, ch)
            if ssz != [h, w]:
                buf = resize(buf, (int(h * k), int(w * k)))
            return buf.get() if on_host else buf
        return wrap
    return wrapf
