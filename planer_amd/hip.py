"""The array backend handed to `core()`: device arrays on one MI355X.

The reference's backend is "any module that quacks like numpy"
(__init__.py:22-38).  This module is the HIP counterpart for planer_amd: it
offers what Net/layers need from a backend -- `asarray` (host -> device,
net.py:96-98), `asnumpy` / `DeviceArray.get()` (device -> host, net.py:100)
-- and nothing more; it is not a numpy clone.
"""
import ctypes
import os

import numpy

from . import _lib
from ._lib import byref, c_int, c_size_t, c_void_p

float32 = numpy.float32

_default_ctx = None
_free_hook = None         # planer_amd.export: told about every block a DeviceArray gives back while a plan is being recorded


class Context:
    """One device + one HIP stream + one caching pool (pl_ctx)."""

    def __init__(self, device=0):
        lib = _lib.load()
        h = c_void_p()
        _lib.check(lib.pl_ctx_create(int(device), byref(h)))
        self.handle = h
        self.device = int(device)
        dev, cus, hbm = c_int(), c_int(), c_size_t()
        name = ctypes.create_string_buffer(64)
        _lib.check(lib.pl_ctx_info(h, byref(dev), byref(cus), byref(hbm), name, 64))
        self.cu_count, self.hbm_bytes, self.arch = cus.value, hbm.value, name.value.decode()
        self.comm = None
        # Launch plans (tile configuration, split-K, occupancy pin per conv shape) are found by timing on first use
        # (pl_conv2d_*: MIOpen-find style).  Two sources spare a run that search and make its kernels repeatable:
        #   * the database shipped with the package for this (architecture, CU count) -- planer_amd/tuned/,
        #     loaded by default (PLANER_HIP_TUNED=0 turns it off);
        #   * PLANER_HIP_TUNE_CACHE=<file>: plans found by earlier runs of this user, loaded on top and saved back.
        self.tuned_db = tuned_db_path(self.arch, self.cu_count) if os.environ.get("PLANER_HIP_TUNED", "1") != "0" else None
        self.tuned_entries = 0
        if self.tuned_db and os.path.exists(self.tuned_db + ".plans"):
            self.tuned_entries = self.load_tune_cache(self.tuned_db + ".plans")
        else:
            self.tuned_db = None
        self.tune_cache = os.environ.get("PLANER_HIP_TUNE_CACHE")
        if self.tune_cache:
            self.load_tune_cache(self.tune_cache)

    def load_tune_cache(self, path):
        n = c_int()
        _lib.call("pl_tune_cache_load", self.handle, path.encode(), byref(n))
        return n.value

    def save_tune_cache(self, path=None):
        path = path or self.tune_cache
        if path:
            _lib.call("pl_tune_cache_save", self.handle, path.encode())

    def pci_bus_id(self):
        """'0000:c1:00.0'-style id of this context's GPU (the key of /sys/bus/pci/devices/)."""
        buf = ctypes.create_string_buffer(32)
        _lib.call("pl_ctx_pci_bus_id", self.handle, buf, 32)
        return buf.value.decode().lower()

    def synchronize(self):
        _lib.call("pl_sync", self.handle)

    def wait_for(self, other):
        """Order this context's stream after everything already enqueued on `other`'s."""
        _lib.call("pl_stream_wait", self.handle, other.handle)

    def wait_event(self, event):
        """Order this context's stream after the point `event` recorded (on any stream of this device)."""
        _lib.call("pl_stream_wait_event", self.handle, event.handle)

    def pool_stats(self):
        r, u = c_size_t(), c_size_t()
        _lib.call("pl_pool_stats", self.handle, byref(r), byref(u))
        return r.value, u.value

    def trim(self):
        _lib.call("pl_pool_trim", self.handle)

    def set_conv_config(self, cfg=-1, split_k=0):
        _lib.call("pl_conv2d_set_config", self.handle, int(cfg), int(split_k))

    def set_conv_plan(self, cfg, dp_tiles, split_k, occupancy=0):
        """Force a full launch plan: `dp_tiles` data-parallel tiles + the rest split `split_k` ways."""
        _lib.call("pl_conv2d_set_plan", self.handle, int(cfg), int(dp_tiles), int(split_k), int(occupancy))

    def tune_stats(self):
        """(launch plans held for this device, conv shapes this context had to time itself)."""
        n, m = c_int(), c_int()
        _lib.call("pl_tune_stats", self.handle, byref(n), byref(m))
        return n.value, m.value

    def last_conv_extents(self):
        """(groups, rows, columns, K) of the GEMM the last conv / dense launch executed, tile and chunk padding included."""
        ext = (ctypes.c_longlong * 4)()
        _lib.call("pl_conv2d_last_extents", self.handle, ext)
        return tuple(int(v) for v in ext)

    def last_conv_plan(self):
        """How the last convolution on this context was launched (kernel family, tile plan)."""
        buf = ctypes.create_string_buffer(160)
        _lib.call("pl_conv2d_last_plan", self.handle, buf, 160)
        return buf.value.decode()

    def close(self):
        if self.handle is not None:
            _lib.load().pl_ctx_destroy(self.handle)
            self.handle = None


def tuned_db_path(arch, cu_count):
    """Stem of the shipped tuning database for a device: planer_amd/tuned/<arch>_cu<CUs> (+ ".plans": launch plans in
    pl_tune_cache_save's format, + ".algo.json": conv algorithm and stream-plan picks of planer_amd.net)."""
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "%s_cu%d" % (arch.split(":")[0], cu_count))


def device_count():
    n = c_int()
    _lib.call("pl_device_count", byref(n))
    return n.value


def context():
    """Process-wide default context.  One process drives one GPU: the device
    is PLANER_HIP_DEVICE, else LOCAL_RANK (torchrun-style launch), else 0."""
    global _default_ctx
    if _default_ctx is None:
        dev = os.environ.get("PLANER_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0"))
        _default_ctx = Context(int(dev))
    return _default_ctx


def set_context(ctx):
    global _default_ctx
    _default_ctx = ctx


# Side streams (contexts) of a device, shared by every Net of the process and created in index order.  The runtime maps
# streams onto its four hardware queues round robin in CREATION order, and a pipelined plan's rate depends on which queue
# each replica lands on (the same seven replicas 53.1 k or 50.4 k img/s): one pool, filled in order,
# keeps side stream i in the same slot whichever Net asks first; `reserve_side_contexts` fills it before something else
# (RCCL's own streams at world > 1) can take slots in between.
_side_pool = {}


def side_context(device, i):
    """Side stream number i >= 1 of `device` (0 is the caller's own context)."""
    pool = _side_pool.setdefault(int(device), [])
    while len(pool) < i:
        pool.append(Context(int(device)))
    return pool[i - 1]


_side_perm = {}       # device -> [creation index of the stream side context i currently holds]


def set_side_stream_perm(device, want):
    """Side context i (i < len(want)) of `device` takes the stream that was CREATED as number want[i]; the side contexts behind
    them share out the remaining streams in creation order (pl_ctx_swap_streams: pools, graphs and events stay with their
    contexts).  `want` holds distinct creation indices; the pool grows to cover the largest."""
    want = [int(w) for w in want]
    if len(set(want)) != len(want) or min(want, default=0) < 0:
        raise ValueError("set_side_stream_perm: distinct creation indices expected, got %r" % (want,))
    side_context(device, max(len(want), max(want, default=-1) + 1))
    pool = _side_pool[int(device)]
    cur = _side_perm.setdefault(int(device), [])
    cur.extend(range(len(cur), len(pool)))
    full = want + [i for i in range(len(pool)) if i not in want]
    for i in range(len(pool)):
        if cur[i] != full[i]:
            j = cur.index(full[i])
            _lib.call("pl_ctx_swap_streams", pool[i].handle, pool[j].handle)
            cur[i], cur[j] = cur[j], cur[i]


def side_stream_perm(device):
    """Creation index of the stream each side context of `device` currently holds (identity until something permutes it)."""
    pool = _side_pool.get(int(device), [])
    cur = _side_perm.get(int(device), [])
    return list(cur) + list(range(len(cur), len(pool)))


def set_side_stream_shift(device, n, shift):
    """Side context i (i < n) of `device` takes the stream created as number (i + shift) % n.  The hardware queue of a stream
    follows its creation order, so a shift moves a pipeline's replicas onto other queues without re-capturing anything
    (Net._probe_streams)."""
    set_side_stream_perm(device, [(i + int(shift)) % n for i in range(n)])


def reserve_side_contexts(device, n):
    if n > 0:
        side_context(device, n)


def trim_side_contexts(device=None):
    """Hand the blocks cached by the shared side contexts back to the driver (`pl_pool_trim` on each; waits for their
    streams).  The side contexts themselves -- and so the hardware-queue slots of their streams -- stay: every Net of
    the process shares them, which also means that two Nets pipelining on one device take turns on the same side
    streams.  Call it after dropping a Net whose throughput plan held R replicas' worth of activations."""
    for dev, pool in _side_pool.items():
        if device is None or int(device) == dev:
            for c in pool:
                c.trim()


def synchronize():
    context().synchronize()


class Event:
    def __init__(self, ctx=None):
        self.ctx = ctx or context()
        h = c_void_p()
        _lib.call("pl_event_create", self.ctx.handle, byref(h))
        self.handle = h

    def record(self):
        _lib.call("pl_event_record", self.ctx.handle, self.handle)
        return self

    def synchronize(self):
        """Host waits for the recorded point (and nothing after it)."""
        _lib.call("pl_event_sync", self.handle)
        return self

    def elapsed_ms(self, stop):
        ms = ctypes.c_float()
        _lib.call("pl_event_elapsed_ms", self.handle, stop.handle, byref(ms))
        return ms.value

    def __del__(self):
        try:
            if self.handle is not None:
                _lib.load().pl_event_destroy(self.handle)
        except Exception:
            pass


class DeviceArray:
    """Dense, C-contiguous array in HBM.  Views (reshape, a[i]) share the
    owner's allocation and keep it alive; the owner returns its block to the
    context pool when it is garbage collected, which is what implements the
    reference's liveness-based freeing of intermediates (net.py:51-53)."""

    __slots__ = ("shape", "dtype", "_p", "ctx", "base", "host", "chan", "meta", "packed", "prefed", "_owned", "__weakref__")

    def __init__(self, shape, dtype=numpy.float32, ctx=None, ptr=None, base=None, host=None):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = numpy.dtype(dtype)
        self.ctx = ctx or (base.ctx if base is not None else context())
        self.base, self.host, self._owned = base, host, False
        self.chan = None      # channel-quad (Q4) tensors: logical channel count (planer_amd/q4.py)
        self.meta = None      # Winograd-domain tensors: the (N, C, H, W) of the activation they stand for
        self.packed = None    # static plan inputs: ((kw, sw, pt, pl), row-packed image) kept beside the NCHW tensor (q4.pack_rows)
        self.prefed = None    # static plan inputs: (feed(src_ptr, ctx), pooled tensor) -- the stem + max-pool kernel runs when the plan is fed
        if ptr is None:
            self._p = None
            if host is None or base is not None:
                self._allocate()
            # else: a host-mirrored tensor (shape-domain values, layer._mirrored) gets its device copy on first use of
            # `.ptr` -- values that only steer views never touch the stream
        else:
            self._p = int(ptr)

    def _allocate(self):
        p = c_void_p()
        _lib.call("pl_alloc", self.ctx.handle, max(self.nbytes, 1), byref(p))
        self._p, self._owned = p.value, True

    @property
    def ptr(self):
        if self._p is None:
            self._allocate()
            if self.host is not None and self.nbytes:
                h = numpy.require(self.host, dtype=self.dtype, requirements="C")
                try:
                    _lib.call("pl_h2d", self.ctx.handle, self._p, h.ctypes.data, h.nbytes)
                except Exception:
                    # the upload did not happen (e.g. NotCapturable during a stream capture): back to the lazy state, or
                    # every later `.ptr` / `.get()` would hand out an allocated but never written block
                    p, self._p, self._owned = self._p, None, False
                    _lib.load().pl_free(self.ctx.handle, c_void_p(p))
                    raise
        return self._p

    # -- numpy-like metadata ------------------------------------------------
    @property
    def size(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def ndim(self):
        return len(self.shape)

    def __len__(self):
        if not self.shape:
            raise TypeError("len() of unsized object")
        return self.shape[0]

    def __repr__(self):
        return "DeviceArray(shape=%s, dtype=%s, dev=%d)" % (self.shape, self.dtype, self.ctx.device)

    # -- views ----------------------------------------------------------------
    def _view(self, shape, offset_bytes=0):
        owner = self.base if self.base is not None else self
        return DeviceArray(shape, self.dtype, self.ctx, self.ptr + offset_bytes, owner)

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        shape = [int(s) for s in shape]
        if -1 in shape:
            known = 1
            for s in shape:
                if s != -1:
                    known *= s
            shape[shape.index(-1)] = self.size // known if known else 0
        n = 1
        for s in shape:
            n *= s
        if n != self.size:
            raise ValueError("cannot reshape array of size %d into shape %s" % (self.size, tuple(shape)))
        if self._p is None:                        # host-mirrored, not on the device yet: stay lazy
            return DeviceArray(shape, self.dtype, self.ctx, host=self.host.reshape(shape))
        v = self._view(shape)
        if self.host is not None:
            v.host = self.host.reshape(shape)
        return v

    def rows(self, lo, hi):
        """a[lo:hi] along axis 0 as a view (contiguous)."""
        n = self.shape[0]
        if not 0 <= lo <= hi <= n:
            raise IndexError((lo, hi))
        step = self.nbytes // n if n else 0
        v = self._view((hi - lo,) + self.shape[1:], lo * step)
        v.chan = self.chan
        return v

    def __getitem__(self, i):
        """a[i] for an integer i: the i-th slab along axis 0 (net.py:101)."""
        if not isinstance(i, (int, numpy.integer)):
            raise TypeError("DeviceArray supports integer indexing only")
        n = self.shape[0]
        i = i + n if i < 0 else i
        if not 0 <= i < n:
            raise IndexError(i)
        step = self.nbytes // n if n else 0
        return self._view(self.shape[1:], i * step)

    # -- transfers ----------------------------------------------------------------
    def get(self):
        """Device -> host copy (synchronises the stream), cupy's `.get()`."""
        if self._p is None:                        # host-mirrored tensor that never went to the device
            return numpy.array(self.host, dtype=self.dtype).reshape(self.shape)
        out = numpy.empty(self.shape, self.dtype)
        if out.nbytes:
            _lib.call("pl_d2h", self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes)
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a if dtype is None else a.astype(dtype)

    def set(self, host):
        host = numpy.require(host, dtype=self.dtype, requirements="C")      # (ascontiguousarray lifts 0-d to 1-d)
        if host.shape != self.shape:
            raise ValueError("shape mismatch %s vs %s" % (host.shape, self.shape))
        if self._p is None:
            self._allocate()
        if host.nbytes:
            _lib.call("pl_h2d", self.ctx.handle, self._p, host.ctypes.data, host.nbytes)
        if self.host is not None:                  # keep a host mirror in step with the device copy
            self.host = host.copy()
        return self

    def set_staged(self, host, consumer=None):
        """Asynchronous `set` (pl_h2d_staged): the bytes travel through the library's pinned ring on a copy stream; `consumer`'s
        stream (a Context; default: this array's) waits for them, the host does not.  `host` may be overwritten on return."""
        host = numpy.require(host, dtype=self.dtype, requirements="C")
        if host.shape != self.shape:
            raise ValueError("shape mismatch %s vs %s" % (host.shape, self.shape))
        if self._p is None:
            self._allocate()
        if host.nbytes:
            _lib.call("pl_h2d_staged", self.ctx.handle, (consumer or self.ctx).handle, self._p, host.ctypes.data, host.nbytes)
        self.host = None
        return self

    def get_begin(self, producer=None):
        """First half of an asynchronous `.get()`: the device -> pinned-host copy is enqueued behind `producer`'s stream
        (default: this array's context).  Returns a ticket for `get_finish` (None: no pinned buffer free -- `get()` then)."""
        if self._p is None or not self.nbytes:
            return None
        t = c_int(-1)
        _lib.call("pl_d2h_begin", self.ctx.handle, (producer or self.ctx).handle, self._p, self.nbytes, byref(t))
        return t.value if t.value >= 0 else None

    def get_finish(self, ticket):
        if ticket is None:
            return self.get()
        out = numpy.empty(self.shape, self.dtype)
        _lib.call("pl_d2h_finish", self.ctx.handle, int(ticket), out.ctypes.data)
        return out

    def get_cancel(self, ticket):
        if ticket is not None and self.ctx.handle is not None:
            _lib.load().pl_d2h_finish(self.ctx.handle, int(ticket), None)

    def copy_from(self, other):
        if other.nbytes != self.nbytes:
            raise ValueError("size mismatch")
        _lib.call("pl_d2d", self.ctx.handle, self.ptr, other.ptr, self.nbytes)
        self.host = None                           # a mirror would be stale now
        return self

    def copy(self):
        return DeviceArray(self.shape, self.dtype, self.ctx).copy_from(self)

    def __del__(self):
        try:
            if self._owned and self._p and self.ctx.handle is not None:
                if _free_hook is not None:
                    _free_hook(self._p)
                _lib.load().pl_free(self.ctx.handle, self._p)
        except Exception:
            pass


ndarray = DeviceArray


def empty(shape, dtype=numpy.float32, ctx=None):
    if isinstance(shape, (int, numpy.integer)):
        shape = (shape,)
    return DeviceArray(shape, dtype, ctx)


def zeros(shape, dtype=numpy.float32, ctx=None):
    a = empty(shape, dtype, ctx)
    if a.nbytes:
        _lib.call("pl_memset", a.ctx.handle, a.ptr, 0, a.nbytes)
    return a


def pinned_empty(shape, dtype=numpy.float32):
    """A numpy array in pinned host memory (pl_host_alloc): batches built in it go to the device by DMA straight out of it,
    no staging copy (`net.submit`, `net(x)`, `DeviceArray.set` / `set_staged`).  The asynchronous routes read it IN PLACE, when
    the consumer's stream gets there: refill it only after the pass that took it is over (`Pending.done()` / `.get()`).
    An ordinary numpy array has no such rule (it is staged before the call returns).  The memory is released with the array."""
    import weakref
    dtype = numpy.dtype(dtype)
    shape = (shape,) if isinstance(shape, (int, numpy.integer)) else tuple(int(v) for v in shape)
    n = dtype.itemsize
    for v in shape:
        n *= v
    p = c_void_p()
    _lib.call("pl_host_alloc", max(n, 1), byref(p))
    buf = (ctypes.c_char * max(n, 1)).from_address(p.value)
    a = numpy.frombuffer(buf, dtype=dtype, count=n // dtype.itemsize).reshape(shape)
    weakref.finalize(buf, _lib.load().pl_host_free, c_void_p(p.value))      # (numpy keeps `buf` alive through .base)
    return a


def asarray(a, dtype=None, ctx=None, consumer=None):
    """Host ndarray -> DeviceArray (net.py:96-98); DeviceArrays pass through.  `consumer`: a Context whose stream waits for
    the upload instead of the host (pl_h2d_staged); the array is read before the call returns either way."""
    if isinstance(a, DeviceArray):
        return a
    host = numpy.require(a, dtype=dtype, requirements="C")
    d = DeviceArray(host.shape, host.dtype, ctx)
    d = d.set_staged(host, consumer) if consumer is not None and host.nbytes >= (128 << 10) else d.set(host)
    if host.dtype != numpy.float32 and host.size <= 4096:
        d.host = host.copy()         # small integer / bool tensors (shapes, indices) stay readable on the host
    return d


def asnumpy(a, **key):
    """DeviceArray -> host ndarray (what core() installs as np.asnumpy,
    __init__.py:36); host arrays pass through."""
    return a.get() if isinstance(a, DeviceArray) else numpy.asarray(a)
