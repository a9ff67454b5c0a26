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


class _Tiling:
    """Geometry of one tiled run (host arithmetic only), following util.tile's rules
    (util.py:300-316): the working size after `sample`, growth of images smaller than the window to
    a multiple of `glob`, the window extents, the margin in pixels and the window grid."""

    def __init__(self, height, width, sample, glob, window, margin):
        self.src = [height, width]
        work = list(sample) if isinstance(sample, tuple) else [int(height * sample), int(width * sample)]
        extent = [window, window]
        for axis in (0, 1):
            if window > work[axis]:                      # smaller than one window: a single, glob-aligned one
                extent[axis] = work[axis] = math.ceil(work[axis] / glob) * glob
        self.work, self.win_h, self.win_w = work, extent[0], extent[1]
        self.margin = int(window * margin) if isinstance(margin, float) else margin
        self.windows = grid_slice(work[0], work[1], self.win_h, self.win_w, self.margin)
        self.resampled = work != self.src

    def back_size(self, k):
        return int(self.src[0] * k), int(self.src[1] * k)


_TILE_KEYS = ("sample", "window", "glob", "margin", "progress", "batched")


def tile(sample=1, glob=1, window=1024, margin=0.1, astype="float32", progress=print, batched=False):
    """util.tile (util.py:291-348).  sample: float factor or (h, w) size; glob: images smaller than
    the window are grown to a multiple of it; window: tile size after resampling; margin: overlap
    between windows (float = fraction of the window, int = pixels).  Every knob can be overridden
    per call by keyword, as in the reference."""
    defaults = dict(sample=sample, glob=glob, window=window, margin=margin, progress=progress, batched=batched)

    def decorate(f):
        def run(image, *rest, **kw):
            opt = dict(defaults, **{k: kw.pop(k) for k in list(kw) if k in _TILE_KEYS})
            host_in = isinstance(image, numpy.ndarray)
            dev = hip.asarray(numpy.ascontiguousarray(image, dtype=numpy.float32)) if host_in else image
            _f32(dev)
            geo = _Tiling(dev.shape[0], dev.shape[1], opt["sample"], opt["glob"], opt["window"], opt["margin"])
            if geo.resampled:
                dev = resize(dev, geo.work)
            n = len(geo.windows)
            report = opt["progress"]
            if n > 1:
                report(1, n)
            if opt["batched"]:                                 # all windows through f as ONE batch
                outs = f(_stack([_window(dev, rc) for rc in geo.windows]), *rest, **kw)
                produce = lambda i: outs[i]
            else:
                produce = lambda i: f(_window(dev, geo.windows[i]), *rest, **kw)
            first = produce(0)
            k = first.shape[0] / geo.win_h                     # output pixels per input pixel
            if n == 1:
                result = resize(first, geo.back_size(k)) if geo.resampled else first
                return result.get() if host_in else result
            # blend buffers in HBM: weighted sum and weight total (util.py:327-344)
            oh, ow = int(dev.shape[0] * k), int(dev.shape[1] * k)
            chans = first.size // (first.shape[0] * first.shape[1])
            total = hip.zeros((oh, ow) + tuple(first.shape[2:]), ctx=dev.ctx)
            weight = hip.zeros((oh, ow), ctx=dev.ctx)
            ramp = int(geo.margin * k)
            for i, (rows, cols) in enumerate(geo.windows):
                if i:
                    report(i + 1, n)
                piece = first if i == 0 else produce(i)
                _lib.call("pl_tile_accumulate_f32", dev.ctx.handle, piece.ptr, total.ptr, weight.ptr, piece.shape[0],
                          piece.shape[1], chans, int(rows.start * k), int(cols.start * k), oh, ow, ramp)
            _lib.call("pl_tile_normalise_f32", dev.ctx.handle, total.ptr, weight.ptr, oh, ow, chans)
            result = resize(total, geo.back_size(k)) if geo.resampled else total
            return result.get() if host_in else result
        return run
    return decorate
