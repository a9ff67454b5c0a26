"""Plan files: a compiled forward pass for hosts without Python (`pl_plan_build` / `pl_plan_run` / `pl_plan_destroy`,
include/planer_hip.h, csrc/plan_exec.hip).

    blob = export_plan(net, x)            # net: a loaded planer_amd.Net, x: one input batch (host or device array)
    open("resnet18_b32.plplan", "wb").write(blob)

The reference executes its flow layer by layer from Python (net.Net.forward, net.py:37-72); this library's plan compiler turns
the flow into a fused program first.  `export_plan` runs that program ONCE with every C-ABI call recorded -- name and arguments,
each device pointer replaced by where it points: an activation block (placed in an arena by first-fit over the blocks' life
times), a constant block (weights, prepared filters, tables: copied into the file), or nothing -- and writes the sequence in the
layout plan_exec.hip documents.  The file is specific to the input shape, like a captured plan.
"""
import ctypes
import struct

import numpy

from . import _lib, hip
from .hip import DeviceArray

_ALIGN = 256
_DTYPES = {"float32": 0, "int32": 1, "int64": 2, "uint8": 3, "bool": 3}
_QUERIES = ("_elems", "_supported")
_IGNORED = {"pl_conv2d_last_plan", "pl_conv2d_last_extents", "pl_tune_stats", "pl_pool_stats", "pl_pool_block", "pl_event_create",
            "pl_event_record", "pl_event_elapsed_ms", "pl_event_sync"}


class _Recorder:
    def __init__(self, ctx):
        self.ctx = ctx
        self.events = []            # ("alloc", id, bytes) | ("free", id) | ("call", name, [arg, ...])
        self.live = {}              # base pointer -> (id, bytes) of blocks allocated while recording
        self.nblocks = 0
        self.consts = {}            # base pointer -> (const id, bytes)
        self.const_order = []

    # -- hooks -------------------------------------------------------------------------------------
    def on_free(self, ptr):
        ptr = int(ptr.value if hasattr(ptr, "value") else ptr)
        ent = self.live.pop(ptr, None)
        if ent is not None:
            self.events.append(("free", ent[0]))

    def on_call(self, name, args):
        if name == "pl_alloc":
            ptr, nbytes = int(args[2]._obj.value), int(args[1])
            self.live[ptr] = (self.nblocks, nbytes)
            self.events.append(("alloc", self.nblocks, nbytes))
            self.nblocks += 1
            return
        if name.endswith(_QUERIES) or name in _IGNORED:
            return
        sig = _lib.SIGNATURES[name]
        if not sig or sig[0] is not ctypes.c_void_p or name in ("pl_h2d", "pl_d2h", "pl_sync") or name.startswith(
                ("pl_ctx_", "pl_comm_", "pl_graph_", "pl_capture_", "pl_stream_", "pl_plan_", "pl_nonzero")):
            raise NotImplementedError("export_plan: the forward pass calls %s, which needs the host between kernels -- this net "
                                      "cannot be replayed from a plan file" % name)
        out = []
        for k, (t, a) in enumerate(zip(sig, args)):
            if k == 0:
                h = a.value if hasattr(a, "value") else a
                if int(h) != int(self.ctx.handle.value):
                    raise NotImplementedError("export_plan: %s runs on another context (side streams are not exported)" % name)
                out.append(("ctx",))
            elif t is ctypes.c_double:
                out.append(("f", float(a)))
            elif t in (ctypes.c_int, ctypes.c_size_t, ctypes.c_longlong):
                out.append(("i", int(a)))
            elif t is ctypes.c_void_p:
                v = a.value if hasattr(a, "value") else a
                out.append(("null",) if not v else self.pointer(int(v), name))
            else:                                   # POINTER(...): a host table (shape / stride arrays) or NULL
                if a is None:
                    out.append(("null",))
                elif isinstance(a, ctypes.Array):
                    out.append(("bytes", bytes(a)))
                else:
                    raise NotImplementedError("export_plan: %s takes a host out-parameter" % name)
        self.events.append(("call", name, out))

    def pointer(self, p, name):
        for base, (bid, nbytes) in self.live.items():
            if base <= p < base + max(nbytes, 1):
                return ("arena", bid, p - base)
        base, size = ctypes.c_void_p(), ctypes.c_size_t()
        try:
            _lib.check(_lib.load().pl_pool_block(self.ctx.handle, ctypes.c_void_p(p), ctypes.byref(base), ctypes.byref(size)))
        except ValueError:
            raise NotImplementedError("export_plan: %s reads %#x, which no block of this context's pool holds" % (name, p))
        b = int(base.value)
        if b not in self.consts:
            self.consts[b] = (len(self.const_order), int(size.value))
            self.const_order.append(b)
        return ("const", self.consts[b][0], p - b)


def _place(events, pinned):
    """First-fit arena offsets for the recorded blocks (life time = alloc .. free in event order; `pinned` ids are never freed).
    -> ({id: offset}, arena bytes)."""
    free, offsets, size_of, top = [], {}, {}, 0          # free: sorted list of (offset, bytes) holes below `top`
    for ev in events:
        if ev[0] == "alloc":
            _, bid, nbytes = ev
            need = (max(nbytes, 1) + _ALIGN - 1) // _ALIGN * _ALIGN
            size_of[bid] = need
            for i, (off, sz) in enumerate(free):
                if sz >= need:
                    offsets[bid] = off
                    if sz > need:
                        free[i] = (off + need, sz - need)
                    else:
                        del free[i]
                    break
            else:
                offsets[bid] = top
                top += need
        elif ev[0] == "free" and ev[1] not in pinned:
            off, sz = offsets[ev[1]], size_of[ev[1]]
            free.append((off, sz))
            free.sort()
            merged = []
            for o, s in free:                             # coalesce neighbours
                if merged and merged[-1][0] + merged[-1][1] == o:
                    merged[-1] = (merged[-1][0], merged[-1][1] + s)
                else:
                    merged.append((o, s))
            free = merged
    return offsets, top


def export_plan(net, *xs, path=None, mode="latency"):
    """-> bytes of the plan file for `net` on inputs shaped like `xs` (host arrays or DeviceArrays).  `mode="throughput"`: the conv
    algorithms a pipelined host (several plan instances on several streams) should run -- the pipeline-judged picks first."""
    ctx = net.ctx
    xs = [hip.asarray(numpy.asarray(a) if not isinstance(a, DeviceArray) else a, ctx=ctx) for a in xs]
    shapes = {k: a.shape for k, a in zip(net.input, xs)}
    shapes.update({k: w.shape for k, w in zip(net.inits, net.weights)})
    net._interpret(net._program, [a.copy() for a in xs], shapes=shapes)          # validates the graph, records every shape
    with net.picking(mode):
        prog, _ = net._fuse(shapes, net.use_fusion)
    net._interpret(prog, [a.copy() for a in xs])                                 # warm: tuning, lazy uploads, pool sizes
    ctx.synchronize()
    kinds = {name: obj.name for name, obj in prog.objs.items()}
    inplace = set()
    for src, names, dst in prog.flow:
        if kinds.get(names[0] if isinstance(names, list) else names) in ("relu", "relu_q4", "flatten", "identity", "return"):
            inplace.update(src if isinstance(src, list) else [src])
    rec = _Recorder(ctx)
    _lib._recorder, hip._free_hook = rec.on_call, rec.on_free
    try:
        statics = [DeviceArray(a.shape, a.dtype, ctx) for a in xs]               # recorded allocations: the plan's inputs
        in_ids = [rec.live[s.ptr][0] for s in statics]
        work = [s.copy() if k in inplace else s for k, s in zip(net.input, statics)]
        out = net._interpret(prog, work)
        del work
        outs = list(out) if isinstance(out, tuple) else [out]
        out_recs = []
        for o in outs:
            if not isinstance(o, DeviceArray):
                raise NotImplementedError("export_plan: the net returns a host value")
            hit = rec.pointer(o.ptr, "the result")
            if hit[0] != "arena" or hit[2] != 0:
                raise NotImplementedError("export_plan: a result that is a view of a constant or of the middle of a block")
            out_recs.append((hit[1], o))
    finally:
        _lib._recorder, hip._free_hook = None, None
    ctx.synchronize()
    pinned = set(in_ids) | {bid for bid, _ in out_recs}
    offsets, arena_bytes = _place(rec.events, pinned)
    # constants: every pool block the pass read that it did not allocate itself
    const_off, blobs, pos = {}, [], 0
    for cid, base in enumerate(rec.const_order):
        size = rec.consts[base][1]
        host = numpy.empty(size, numpy.uint8)
        _lib.call("pl_d2h", ctx.handle, host.ctypes.data, base, size)
        const_off[cid] = pos
        blobs.append(host.tobytes())
        pad = (-size) % _ALIGN
        blobs.append(b"\0" * pad)
        pos += size + pad

    def tensor(bid, arr):
        dims = list(arr.shape)[:8] + [0] * (8 - min(len(arr.shape), 8))
        return struct.pack("<QQII8I", offsets[bid], arr.nbytes, _DTYPES[str(arr.dtype)], len(arr.shape), *dims)
    body = []
    ncalls = 0
    for ev in rec.events:
        if ev[0] != "call":
            continue
        ncalls += 1
        _, name, args = ev
        nm = name.encode()
        body.append(struct.pack("<I", len(nm)) + nm + b"\0" * ((-len(nm)) % 4) + struct.pack("<I", len(args)))
        for a in args:
            if a[0] == "i":
                body.append(struct.pack("<IIq", 0, 0, a[1]))
            elif a[0] == "f":
                body.append(struct.pack("<IId", 1, 0, a[1]))
            elif a[0] == "null":
                body.append(struct.pack("<IIQ", 2, 0, 0))
            elif a[0] == "arena":
                body.append(struct.pack("<IIQ", 3, 0, offsets[a[1]] + a[2]))
            elif a[0] == "const":
                body.append(struct.pack("<IIQ", 4, 0, const_off[a[1]] + a[2]))
            elif a[0] == "bytes":
                body.append(struct.pack("<IIQ", 5, 0, len(a[1])) + a[1] + b"\0" * ((-len(a[1])) % 8))
            else:
                body.append(struct.pack("<IIQ", 6, 0, 0))
    head = b"PLPLAN1\0" + struct.pack("<QQIIII", pos, arena_bytes, len(statics), len(out_recs), ncalls, 0)
    head += b"".join(tensor(bid, s) for bid, s in zip(in_ids, statics))
    head += b"".join(tensor(bid, o) for bid, o in out_recs)
    blob = head + b"".join(body) + b"".join(blobs)
    if path:
        with open(path, "wb") as f:
            f.write(blob)
    return blob
