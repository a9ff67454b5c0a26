"""planer's graph executor (`Net`) on MI355X.

Public surface and IR semantics are the reference's (net.py:5-101):
`load_json(inputs, inits, body, flow)`, `load_weights(uint8 blob)`,
`forward`, `__call__` (dict or positional inputs, host ndarray in -> host
ndarray out, the `rst[0] if len(rst)==1` unwrapping), `run`, `timeit`,
`timer`, liveness-based freeing of intermediates, chained layers and
multi-output steps.

What is new underneath (MI355X-first, no counterpart in the reference):
  * weights live in ONE device allocation (256-byte aligned slots) so the
    multi-GPU weight exchange is a single RCCL broadcast;
  * `forward` enqueues one hand-written HIP kernel per layer on the context
    stream and never synchronises;
  * `__call__` on device-resident inputs runs a *compiled plan*: the flow is
    peephole-fused (conv -> batchnorm -> [add] -> [relu|leakyrelu] become one
    MFMA kernel with a fused epilogue), executed once to warm the memory
    pool, then captured into a hipGraph that is replayed with a single C call
    per forward pass.
"""
import os
import time

import numpy

from . import _lib
from . import hip
from .hip import DeviceArray
from . import q4 as _q4
from .layer import layer_map, wrap
from .plan import assign_layouts, chain_winograd, fuse_conv1x1_wino_in, fuse_flow, pair_sibling_convs

_q4.register(layer_map)

_ALIGN = 256

# ConvQ4 / ConvFused w_layout codes -> what runs (for run reports)
W_LAYOUT_NAMES = {0: "igemm-nchw", 1: "tap-nchw", 2: "direct-q4 (conv_q4_kernel)", 3: "wino2x2-nchw",
                  4: "wino2x2-q4 (transforms + grouped conv_q4_kernel)",
                  6: "rowpack-q4 (nchw_to_rowpack + conv_q4_kernel)",
                  7: "wino4x4-q4 (transforms + grouped conv_q4_kernel)", 8: "w1d4 F(4,3) (conv_w1d4_kernel)",
                  9: "wf4 fused F(4x4,3x3) (conv_wf4_kernel)", 10: "stem + maxpool (conv_stem_pool_kernel)",
                  11: "wino43-q4 (mixed F(4,3) x F(3,3) tiles: transforms + 121 grouped conv_q4_kernel)",
                  12: "stem + maxpool (conv_stem_pool_kernel)"}         # (the kernel reads the NCHW batch itself)


def _as_list(v):
    return list(v) if isinstance(v, (list, tuple)) else [v]


class _Program:
    """A flow ready to interpret: layer objects, steps and last-use table."""

    def __init__(self, body, flow):
        self.objs = {name: wrap(layer_map[kind], kind)(**para) for name, kind, para in body}
        self.flow = flow
        self.life = {}
        for i, step in enumerate(flow):
            for k in _as_list(step[0]):
                self.life[k] = i                     # net.py:16-19


def prog_body(prog):
    return [(name, obj.name) for name, obj in prog.objs.items()]


def graph_tag_of(layers, flow, init_shapes):
    """crc32 over layer kinds and parameters, flow wiring and init shapes (Net.graph_tag)."""
    import json
    import zlib

    def plain(v):
        # numpy scalars / arrays print differently across numpy versions (np.float32(0.1) vs 0.1): hash plain Python values
        if isinstance(v, numpy.ndarray):
            return v.tolist()
        if isinstance(v, numpy.generic):
            return v.item()
        if isinstance(v, dict):
            return {str(k): plain(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [plain(x) for x in v]
        return v
    text = json.dumps(plain([layers, flow, [list(map(int, s)) for s in init_shapes]]), sort_keys=True, default=repr)
    return "%08x" % (zlib.crc32(text.encode()) & 0xffffffff)


def _feed_static(ctx, dst, a, always=False):
    """A new batch `a` (a device tensor) into a captured plan's static input `dst`, on `ctx`'s stream.  Where the plan's
    only reader of that input is the row-packed stem conv (Net._capture, `dst.packed`), the batch is re-laid straight
    into the packed image the graph reads -- that pass replaces the copy -- and the NCHW tensor is left alone.  Where that
    reader is the stem + max-pool kernel that takes NCHW itself (`dst.prefed`), the kernel runs HERE, from the caller's batch
    into the pooled tensor the captured pass starts from: no copy and no re-layout at all."""
    if dst.prefed is not None:
        if a.ptr % 16:                                   # (a view that starts off a 16-byte boundary: through the static tensor)
            _lib.call("pl_d2d", ctx.handle, dst.ptr, a.ptr, dst.nbytes)
            a = dst
        dst.prefed[0](a.ptr, ctx)
    elif dst.packed is not None:
        _q4.pack_rows(dst, src_ptr=a.ptr, ctx=ctx)
    elif always or a is not dst:
        _lib.call("pl_d2d", ctx.handle, dst.ptr, a.ptr, dst.nbytes)


class _Plan:
    """One captured forward pass for one input signature."""

    def __init__(self, inputs, outputs, graph, ctx, fused_steps):
        self.inputs, self.outputs, self.graph, self.ctx = inputs, outputs, graph, ctx
        self.fused_steps, self.ms, self.streams = fused_steps, None, "1x1"
        self.algos = []              # per conv step: kernel family + launch plan (filled by Net._capture)
        # bounded host run-ahead: launch k waits (on the host) for launch k - max_in_flight of this plan's stream.
        # A free-running loop otherwise queues every batch it is asked for at once; eight batches keep the GPU fed
        # for milliseconds, and tools that intercept dispatches (rocprofv3) crash with ~2000 of them outstanding.
        self.max_in_flight, self._ring = 8, []
        self._held = []              # per ring slot: what that launch's feed read (Net.submit), released with the slot
        self.host_in, self.fed = None, None      # Net.submit with host batches: this replica's own upload buffers + "feed has read them"
        self._fed = False            # a feed has run since the last launch (else launch() runs the feed-time part itself)

    def feed(self, xs):
        """Copy new inputs into the plan's static input buffers (on the plan's stream)."""
        for s, a in zip(self.inputs, xs):
            _feed_static(self.ctx, s, a)
        self._fed = True

    def launch(self, join=True, hold=None):
        """`hold`: objects the feed of THIS launch read on this plan's stream (the caller's input arrays).  They stay
        referenced until the host has seen a marker behind this launch complete: a block the caller drops right after
        submitting must not go back to its pool -- and be overwritten through another stream -- before the asynchronous
        feed has read it.
        A launch that no `feed` preceded takes what the caller wrote into `plan.inputs[i]` itself: where part of the pass
        runs at feed time (the stem + max-pool kernel in front of the captured graph, `DeviceArray.prefed`; the row-packed
        copy, `.packed`) that part runs here from the static tensor -- the graph alone would read the previous batch's
        pooled tensor (round-5 advisor)."""
        if not self._fed:
            for s in self.inputs:
                if s.prefed is not None or s.packed is not None:
                    _feed_static(self.ctx, s, s)
        self._fed = False
        _lib.call("pl_graph_launch", self.graph)
        if self.max_in_flight:
            if len(self._ring) >= self.max_in_flight:
                ev = self._ring.pop(0)
                ev.synchronize()
                self._held.pop(0)
            else:
                ev = hip.Event(self.ctx)
            self._ring.append(ev.record())
            self._held.append(hold)
        elif hold is not None:       # no ring to tie the references to: wait for the launch instead
            self.ctx.synchronize()

    def join(self):
        pass

    def __del__(self):
        try:
            self.outputs = self.inputs = None      # drop buffers before their graph
            if self.graph is not None:
                _lib.load().pl_graph_destroy(self.graph)
        except Exception:
            pass


class _NotSplittable(Exception):
    """An output does not carry the batch on axis 0, so sub-batch streams cannot be used."""


class _MultiPlan:
    """A forward pass fanned out over several streams: sub-batch i is its own captured graph on
    side stream i (kernels of different streams run concurrently and fill CUs that one small
    grid leaves idle); the net's own stream forks before and joins after."""

    def __init__(self, inputs, outputs, subs, ctx, fused_steps):
        self.inputs, self.outputs, self.subs, self.ctx = inputs, outputs, subs, ctx
        self.fused_steps, self.ms = fused_steps, None
        self.streams = "%dx%d" % (len(subs), len(subs))
        self.algos = subs[0].algos

    def feed(self, xs):
        """Per-stream input copy: stream i copies ITS rows, ordered behind its own previous
        sub-graph, so consecutive forward passes may overlap across streams without a join."""
        n = self.inputs[0].shape[0] // len(self.subs)
        for i, sp in enumerate(self.subs):
            for dst, a in zip(sp.inputs, xs):
                _feed_static(sp.ctx, dst, a.rows(i * n, (i + 1) * n), always=True)
            sp._fed = True

    def launch(self, join=True):
        """join=True: fork from / join into the net's own stream (what Net.__call__ needs: inputs
        written and outputs read on the main stream).  join=False: throughput mode -- the streams
        free-run and pipeline across passes; call join() before reading outputs."""
        if join:
            for c in {id(sp.ctx): sp.ctx for sp in self.subs}.values():
                c.wait_for(self.ctx)           # fork FIRST: inputs were written on the main stream
        for sp in sorted(self.subs, key=lambda q: q.ctx is self.ctx):   # main-stream graphs last
            sp.launch()
        if join:
            self.join()

    def join(self):
        """Gather every sub-batch's outputs into the full output (on its own stream, outside the
        graphs: memcpy nodes inside the sub-graphs serialised the streams), then join."""
        for sp, rows in zip(self.subs, self.out_rows):
            outs = sp.outputs if isinstance(sp.outputs, tuple) else (sp.outputs,)
            for o, d in zip(outs, rows):
                _lib.call("pl_d2d", sp.ctx.handle, d.ptr, o.ptr, o.nbytes)
        for c in {id(sp.ctx): sp.ctx for sp in self.subs}.values():
            self.ctx.wait_for(c)


class _PipelinePlan:
    """Throughput mode only: R full-batch graphs ("replicas"), each on its own stream, used round
    robin -- consecutive forward passes run on different streams, so one batch's kernel tails,
    launch ramps and small kernels hide under the next batch's convs while every kernel keeps the
    full batch (better tile efficiency than sub-batch graphs).  `outputs` are those of the replica
    launched last; a replica's buffers are rewritten R launches later."""

    def __init__(self, replicas, ctx, fused_steps):
        self.replicas, self.ctx, self.fused_steps, self.ms = replicas, ctx, fused_steps, None
        self.streams = "pipe%d" % len(replicas)
        for rp in replicas:                  # the host's run-ahead is bounded over the whole pipeline, not per replica
            rp.max_in_flight = max(2, 24 // len(replicas))
        self.algos = replicas[0].algos
        self.turn, self.last = 0, replicas[0]
        self.stream_probe = None             # {"shift": s, "ms_per_pass": [...]} once Net._probe_streams has placed the replicas

    @property
    def inputs(self):
        return self.last.inputs

    @property
    def outputs(self):
        return self.last.outputs

    def feed(self, xs):
        rp = self.replicas[self.turn]
        for dst, a in zip(rp.inputs, xs):
            _feed_static(rp.ctx, dst, a, always=True)                           # on the replica's stream
        rp._fed = True

    def launch(self, join=True, hold=None):
        rp = self.replicas[self.turn]
        self.turn = (self.turn + 1) % len(self.replicas)
        self.last = rp
        if join:
            rp.ctx.wait_for(self.ctx)
        rp.launch(hold=hold)
        if join:
            self.join()

    def join(self):
        for rp in self.replicas:
            if rp.ctx is not self.ctx:
                self.ctx.wait_for(rp.ctx)


class Pending:
    """Handle of a forward pass submitted with `Net.submit`: the pass runs on one of the net's replica streams while the
    caller goes on (submits the next batch, prepares inputs).  `result()` hands the outputs to the net's own stream --
    device arrays, private to this handle, ordered behind the pass (no host wait); `get()` returns host arrays.  Both
    unwrap like `Net.__call__` (net.py:101)."""

    def __init__(self, net, outs, event, to_host, final=False, tickets=None):
        self.net, self._outs, self._event, self._to_host, self._final = net, outs, event, to_host, final
        # host-array submits: the device -> pinned-host copies of the outputs were enqueued at submit time (one ticket per
        # output, None where no pinned buffer was free); get() only waits for them
        self._tickets = tickets

    def done(self):
        """Host-side wait for this pass alone (not for passes submitted after it)."""
        if self._event is not None:
            self._event.synchronize()
        return self

    def result(self):
        if self._event is not None:
            self.net.ctx.wait_event(self._event)
            self.net._events.append(self._event)
            self._event = None
        rst = self._outs
        if self._final:                               # Net.__call__ ran the pass (flows that cannot be captured): as it returned
            return rst
        if self._to_host:
            tk, self._tickets = self._tickets, None
            if tk is not None:
                seq = rst if isinstance(rst, tuple) else (rst,)
                got = tuple(o.get_finish(t) if isinstance(o, DeviceArray) else o for o, t in zip(seq, tk))
                rst = got if isinstance(rst, tuple) else got[0]
            else:
                rst = tuple(i.get() for i in rst) if isinstance(rst, tuple) else rst.get()
        return rst[0] if len(rst) == 1 else rst

    def __del__(self):
        try:
            tk, self._tickets = self._tickets, None
            if tk is not None:                        # dropped without get(): hand the pinned buffers back
                seq = self._outs if isinstance(self._outs, tuple) else (self._outs,)
                for o, t in zip(seq, tk):
                    if isinstance(o, DeviceArray):
                        o.get_cancel(t)
        except Exception:
            pass

    def get(self):
        if self._final:
            rst = self._outs
            if isinstance(rst, tuple):
                return tuple(i.get() if isinstance(i, DeviceArray) else i for i in rst)
            return rst.get() if isinstance(rst, DeviceArray) else rst
        self._to_host = True
        return self.result()


class Net:
    def __init__(self, ctx=None):
        self.weights, self.body, self.flow = [], [], []
        self.life, self.timer = {}, {}
        self.input, self.inits, self.layer = [], [], []
        self.ctx = ctx
        self.use_graph = os.environ.get("PLANER_HIP_GRAPH", "1") != "0"
        self.use_fusion = os.environ.get("PLANER_HIP_FUSE", "1") != "0"
        self.profile = os.environ.get("PLANER_HIP_PROFILE", "0") == "1"
        # compiled plans keep activations channel-quad (Q4) between layers that have Q4 kernels
        # "force": channel-quad layout wherever a Q4 kernel exists, whatever the conversion-cost estimate says (tests)
        self.use_q4 = {"0": False, "force": "force"}.get(os.environ.get("PLANER_HIP_Q4", "1"), True)
        # streams: how many sub-batch graphs a forward pass is fanned out to ("auto" measures 1/2/4)
        self.streams = os.environ.get("PLANER_HIP_STREAMS", "auto")
        self._side = []
        self._events = []            # recycled stream markers of finished Pending handles (Net.submit)
        self.device_timer = {}       # kind -> ms of device time (profile mode)
        self.last_events = []        # [(layer name, kind, ms)] of the last profiled forward
        self._blob = None
        self._slots = []             # (offset, nbytes) per weight inside the blob
        self._program = None
        self._eager_prog = {}        # input signature -> fused program, for nets whose flow cannot be captured (Net.__call__)
        self._plans = {}
        self._extra = {}             # derived constant tensors (tap-major / Winograd filters)
        self._algo = {}              # conv shape signature -> chosen w_layout
        self._algo_tp = {}           # the same for THROUGHPUT plans where the pipeline's judge differs from the isolated one
        self._pick_mode = None       # "throughput" while a throughput plan's program is being fused (Net.picking)
        self.wino_chains = 0         # F(4x4,3x3) output / input transform pairs the last plan runs as one kernel
        self.conv_pairs = 0          # sibling conv pairs the last plan runs as one launch
        self.conv_wino_fused = 0     # 1x1 convs the last plan runs inside the next conv's Winograd input transform
        # force_algo: w_layout (int) every eligible 3x3/s1/p1 conv must use, or None = pick by timing
        fa = os.environ.get("PLANER_HIP_CONV_ALGO")
        self.force_algo = int(fa) if fa else None
        # choices persist next to the C library's launch-plan cache, so a second run (profiling!)
        # launches no trial kernels and reproduces the first run's kernels exactly; the database shipped in
        # planer_amd/tuned/ (hip.Context.tuned_db) holds the picks for the BASELINE workloads on this device
        tc = os.environ.get("PLANER_HIP_TUNE_CACHE")
        self.algo_cache = tc + ".algo.json" if tc else None
        self._algo_loaded = False
        self._streams_pick = {}      # compile key -> stream plan ("pipe3", "2x2", ...) chosen by measurement
        self.algo_misses = 0         # conv algorithm picks that had to be measured (no cached choice)
        self.stream_misses = 0       # stream-plan picks that had to be measured

    # ---- loading ----------------------------------------------------------------
    def load_json(self, inputs, inits, body, flow, debug=False):
        """net.py:10-24.  Unknown layer kinds raise KeyError like the reference."""
        self.ctx = self.ctx or hip.context()
        if debug:
            for i in body:
                print(i)
        self._program = _Program(body, flow)
        self.body = list(self._program.objs.items())
        self.life = self._program.life
        self.input, self.inits = inputs, [i[0] for i in inits]
        self.layer, self.flow = body, flow
        # one device allocation for every weight; slots padded to 256 B
        self._slots, off = [], 0
        for _, shape, dt in inits:
            nbytes = int(numpy.prod(shape, dtype=numpy.int64)) * numpy.dtype(dt).itemsize
            self._slots.append((off, nbytes))
            off += (nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
        self._blob = hip.zeros((max(off, 1),), numpy.uint8, self.ctx)
        self._host = numpy.zeros(max(off, 1), numpy.uint8)
        self.weights = []
        for (o, nbytes), (_, shape, dt) in zip(self._slots, inits):
            host = self._host[o:o + nbytes].view(dt).reshape(shape)
            self.weights.append(DeviceArray(shape, dt, self.ctx, self._blob.ptr + o, self._blob, host))
        self._plans = {}

    def load_weights(self, data):
        """net.py:83-88: `data` holds the raw bytes of every init back to back."""
        raw = numpy.ascontiguousarray(data).view(numpy.uint8).ravel()
        need = sum(n for _, n in self._slots)
        if raw.size < need:
            raise ValueError("weight blob has %d bytes, the graph needs %d" % (raw.size, need))
        pos = 0
        for o, n in self._slots:
            self._host[o:o + n] = raw[pos:pos + n]
            pos += n
        self._blob.set(self._host)
        self._plans, self._extra, self._eager_prog = {}, {}, {}

    def weight_blob(self):
        """The single device allocation holding all weights (RCCL broadcast unit)."""
        return self._blob

    def refresh_host_mirror(self):
        """Re-read the host mirror after the device blob was written by a broadcast."""
        self._host[:] = self._blob.get()
        self._plans, self._extra = {}, {}

    def half(self):
        raise NotImplementedError("planer_amd computes the hot path in float32 only (BASELINE metric)")

    def show(self):
        raise NotImplementedError("plot_net is missing from the reference snapshot too (net.py:90-92)")

    def info(self, obj):
        if isinstance(obj, (list, tuple)):
            return [self.info(i) for i in obj]
        return obj.shape if hasattr(obj, "shape") else obj

    # ---- eager interpreter ---------------------------------------------------------
    def _interpret(self, prog, xs, debug=False, shapes=None, profile=False, record=None, repeat=1):
        """net.py:37-72: one kernel launch (or view) per layer, in flow order.  `record` (a list)
        receives one dict per convolution / dense step: which kernel family and launch plan ran.  `profile` brackets the
        steps with stream markers; with `repeat` = R > 1 every step is launched R times between its markers (the time per
        launch is what gets recorded: a marker costs about as much as a small kernel, R launches dilute it)."""
        env = {"None": None}
        env.update(zip(self.inits, self.weights))
        env.update(self._extra)
        env.update(zip(self.input, xs))
        events, out_key = [], None
        for i, (src, names, dst) in enumerate(prog.flow):
            for pos, name in enumerate(_as_list(names)):
                keys = src if pos == 0 else dst                  # chained layers
                args = [env[keys]] if isinstance(keys, str) else [env.get(k) for k in keys]
                for k in set(_as_list(src)):
                    if k in env and prog.life[k] <= i:
                        del env[k]                               # release dead inputs
                obj = prog.objs[name]
                if debug:
                    print(name, obj.name, ":", obj.para())
                    print("\t--> ", keys, ":", self.info(args))
                t0 = time.time()
                if profile and not events:
                    events.append((None, None, hip.Event(self.ctx).record()))
                val = obj(*args)
                for _ in range(repeat - 1 if profile else 0):
                    val = obj(*args)
                if profile:                                  # ONE marker per step boundary: step time = marker to marker
                    events.append((name, obj.name, hip.Event(self.ctx).record()))
                if record is not None and obj.name in ("conv", "conv_fused", "conv_q4", "dense", "matmul", "wino4_gemm", "wino43_gemm",
                                                       "conv_q4_pair", "conv_pool_q4", "conv1x1_wino_in"):
                    lay = obj.para().get("w_layout", 2 if obj.name in ("conv_q4_pair", "conv1x1_wino_in") else 0) if obj.name != "conv" else 0
                    lname = name
                    if obj.name in ("wino4_gemm", "wino43_gemm"):       # the GEMM stage of a staged Winograd conv
                        lay, lname = (7 if obj.name == "wino4_gemm" else 11), name[:-len("@gemm")]
                    ctx_ = args[0].ctx if isinstance(args[0], DeviceArray) else self.ctx
                    xshape = (args[0].meta if args[0].meta is not None else
                              _q4.logical_shape(args[0]) if _q4.is_q4(args[0]) else args[0].shape)
                    record.append({"layer": lname, "kind": obj.name, "w_layout": lay,
                                   "algo": W_LAYOUT_NAMES.get(lay, str(lay)), "plan": ctx_.last_conv_plan(),
                                   "extents": list(ctx_.last_conv_extents()), "x": list(xshape)})
                del args
                if isinstance(dst, str):
                    env[dst] = val
                else:
                    env.update(zip(dst, val))
                if debug:
                    for k in _as_list(dst):
                        print("\t<-- ", k, ":", self.info(env[k]))
                if shapes is not None:
                    for k in _as_list(dst):
                        shapes[k] = getattr(env[k], "shape", None)
                self.timer[obj.name] = self.timer.get(obj.name, 0) + time.time() - t0
                del val
            out_key = dst
        if profile:
            self.last_events = []
            for (_, _, e0), (name, kind, e1) in zip(events, events[1:]):
                ms = e0.elapsed_ms(e1) / max(1, repeat)
                self.last_events.append((name, kind, ms))
                self.device_timer[kind] = self.device_timer.get(kind, 0.0) + ms
        return env[out_key]

    def forward(self, *x, debug=False):
        """Per-layer execution of the flow exactly as written (no fusion)."""
        return self._interpret(self._program, x, debug=debug, profile=self.profile)

    def timeit(self, status="start"):
        """net.py:74-77"""
        if status == "start":
            self.timer, self.device_timer = {}, {}
        if status == "end":
            for k in self.timer:
                print(k, self.timer[k])

    def run(self, output=None, input={}):
        """onnxruntime-style entry (net.py:79-81)"""
        rst = self(input)
        return rst if isinstance(rst, tuple) else (rst,)

    # ---- plan compiler -----------------------------------------------------------------
    def _fuse(self, shapes, fuse=True):
        """Plan program: fused epilogues + tap-major filter copies for the fast conv kernel."""
        if fuse:
            body, flow, nfused = fuse_flow(self.layer, self.flow, self.inits, shapes)
        else:
            body, flow, nfused = [list(b) for b in self.layer], [list(f) for f in self.flow], 0
        if self.use_q4:
            body, flow, _ = assign_layouts(body, flow, self.inits, shapes, force=self.use_q4 == "force")
        if os.environ.get("PLANER_HIP_TAPMAJOR", "1") != "0":
            body, flow = self._prepare_filters(body, flow, shapes)
        return _Program(body, flow), nfused

    def _prepare_filters(self, body, flow, shapes):
        """Give every eligible conv a prepared copy of its (constant) filter: tap-major for the
        fast implicit-GEMM kernel, or Winograd-domain where that measures faster for this shape."""
        from .layer import prepare_conv_weights, prepare_winograd_weights, winograd_eligible, ConvFused
        wmap = dict(zip(self.inits, self.weights))
        kinds = {b[0]: b for b in body}
        out_body = {b[0]: list(b) for b in body}
        out_flow = []
        use_wino = os.environ.get("PLANER_HIP_WINOGRAD", "1") != "0"
        for src, names, dst in flow:
            name = names[0] if isinstance(names, list) else names
            entry = kinds[name]
            srcs = list(src) if isinstance(src, list) else [src]
            if entry[1] == "conv_q4":
                K = wmap[srcs[1]]
                group = int(entry[2].get("group", 1))
                para = {k: v for k, v in entry[2].items() if k in ("group", "strides", "dilations", "pads")}
                lay = 2
                if entry[2].get("rowpack") and _q4.rowpack_eligible(K.shape, **para):
                    lay = 6
                elif (use_wino and _q4.w1d_q4_eligible(K.shape, **para)
                        and shapes.get(srcs[0].split("@")[0]) is not None):
                    lay = self._pick_conv_algo(_q4.ConvQ4, K, srcs, entry[2], shapes, wmap, q4=True)
                key = {2: "%s@q4g%d" % (srcs[1], group), 4: srcs[1] + "@winoq4", 6: srcs[1] + "@rowpack", 7: srcs[1] + "@wino4q4",
                       8: srcs[1] + "@w1d4q4", 9: srcs[1] + "@wf4q4", 11: srcs[1] + "@wino43q4"}[lay]
                if key not in self._extra:
                    self._extra[key] = {2: lambda: _q4.prepare_q4_weights(K, group),
                                        4: lambda: _q4.prepare_winograd_q4_weights(K),
                                        6: lambda: _q4.prepare_rowpack_weights(K),
                                        7: lambda: _q4.prepare_winograd4_q4_weights(K),
                                        8: lambda: _q4.prepare_w1d4_q4_weights(K),
                                        9: lambda: _q4.prepare_wf4_q4_weights(K),
                                        11: lambda: _q4.prepare_winograd43_q4_weights(K)}[lay]()
                srcs[1] = key
                out_body[name] = [name, "conv_q4", dict(entry[2], w_layout=lay)]
            elif entry[1] in ("conv", "conv_fused") and len(srcs) >= 2 and srcs[1] in wmap:
                K = wmap[srcs[1]]
                if K.ndim == 4 and K.dtype == numpy.float32 and K.shape[1] % 16 == 0:
                    para = {k: v for k, v in entry[2].items() if k in ("group", "strides", "dilations", "pads")}
                    lay = 1
                    if use_wino and winograd_eligible(K.shape, **para) and shapes.get(srcs[0]) is not None:
                        lay = self._pick_conv_algo(ConvFused, K, srcs, entry[2], shapes, wmap)
                    key = srcs[1] + ("@tap" if lay == 1 else "@wino")
                    if key not in self._extra:
                        self._extra[key] = prepare_conv_weights(K) if lay == 1 else prepare_winograd_weights(K)
                    srcs[1] = key
                    out_body[name] = [name, "conv_fused", dict(entry[2], w_layout=lay)]
            out_flow.append([srcs, [name], dst])
        # the row-packed stem conv whose only reader is maxpool(3x3 / s2 / p1): one kernel that writes the pooled tensor only
        # (csrc/conv_stem_pool_kernel.h); PLANER_HIP_STEM_POOL=0 keeps the two kernels
        if os.environ.get("PLANER_HIP_STEM_POOL", "1") != "0":
            out_flow = self._fuse_stem_pool(out_body, out_flow, shapes)
        out_flow = self._fuse_upsample_concat(out_body, out_flow)
        used = {n for _, names, _ in out_flow for n in names}
        out_list = [out_body[b[0]] for b in body if b[0] in used]
        # two direct convs on one tensor (a resolution-changing ResNet block) -> one launch; PLANER_HIP_PAIR=0 turns it off
        if os.environ.get("PLANER_HIP_PAIR", "1") != "0":
            out_list, out_flow, self.conv_pairs = pair_sibling_convs(
                out_list, out_flow, lambda key: shapes.get(key.split("@")[0]))
        # F(4x4,3x3) convs as explicit stages, consecutive ones sharing a transform kernel (plan.chain_winograd);
        # PLANER_HIP_WINO_CHAIN: "1" chain (default), "stages" explicit stages without chaining, "0" one call per conv
        mode = os.environ.get("PLANER_HIP_WINO_CHAIN", "1")
        if mode != "0":
            def fits(key):
                # whole planes per workgroup: the plane must fit the LDS kernel, and there must be enough (image,
                # channel quad) planes to fill the chip (a batch-1 detection net has 32-256 of them: not worth it)
                shp = shapes.get(key.split("@")[0])
                return (shp is not None and len(shp) == 4 and shp[0] * (shp[1] // 4) >= self.ctx.cu_count // 2
                        and _q4.wino4_chain_supported(tuple(shp), self.ctx))
            out_list, out_flow, self.wino_chains = chain_winograd(out_list, out_flow, fits, chain=mode != "stages")
            # a 1x1 conv whose only reader is such a conv's input transform writes V itself (detection nets at small maps: one
            # launch less per 3x3 conv); PLANER_HIP_CONV1X1_WINO=0 turns it off, =<tiles> moves the size limit
            lim = os.environ.get("PLANER_HIP_CONV1X1_WINO", "4096")
            if lim != "0":
                def small(key):
                    shp = shapes.get(key.split("@")[0])
                    return shp is not None and len(shp) == 4 and shp[0] * (-(-shp[2] // 4)) * (-(-shp[3] // 4)) <= int(lim)
                out_list, out_flow, self.conv_wino_fused = fuse_conv1x1_wino_in(out_list, out_flow, self._shape_of_init, small)
        return out_list, out_flow

    @staticmethod
    def _fuse_upsample_concat(body, flow):
        """upsample_q4 whose only reader is a two-input concat_q4 (axis 1) that takes it FIRST -> one upconcat_q4 step
        (detection-net routes: the upsampled tensor is written once, into its place in the concatenation)."""
        readers = {}
        for i, (src, names, dst) in enumerate(flow):
            for k in (src if isinstance(src, list) else [src]):
                readers.setdefault(k, []).append(i)
        drop, out = set(), {}
        for i, (src, names, dst) in enumerate(flow):
            entry = body[names[0]]
            if (entry[1] == "upsample_q4" and isinstance(dst, str) and isinstance(src, list) and len(src) == 2
                    and entry[2].get("mode", "nearest") == "nearest" and len(readers.get(dst, [])) == 1):
                j = readers[dst][0]
                csrc, cnames, cdst = flow[j]
                ce = body[cnames[0]]
                if (ce[1] == "concat_q4" and len(cnames) == 1 and isinstance(csrc, list) and len(csrc) == 2 and csrc[0] == dst
                        and int(ce[2].get("axis", 0)) == 1 and i < j
                        # the upsample now reads its source at step j: nobody may touch it in between (in-place ReLU)
                        and not any(src[0] in (f[0] if isinstance(f[0], list) else [f[0]]) for f in flow[i + 1:j])):
                    body[cnames[0]] = [ce[0], "upconcat_q4", {"mode": "nearest", "axis": 1}]
                    out[j] = [[src[0], src[1], csrc[1]], cnames, cdst]
                    drop.add(i)
        return [out.get(i, f) for i, f in enumerate(flow) if i not in drop]

    def _fuse_stem_pool(self, body, flow, shapes):
        """conv_q4 (row-packed stem, w_layout 6, no residual) whose only reader is maxpool_q4(w=3x3, strides 2, pads 1) -> one
        conv_pool_q4 step: the full-resolution tensor is never written."""
        readers = {}
        for i, (src, names, dst) in enumerate(flow):
            for k in (src if isinstance(src, list) else [src]):
                readers.setdefault(k, []).append(i)
        drop, out = set(), []
        for i, (src, names, dst) in enumerate(flow):
            entry = body[names[0]]
            if (entry[1] == "conv_q4" and entry[2].get("w_layout") == 6 and isinstance(dst, str)
                    and (len(src) < 6 or src[5] == "None") and len(readers.get(dst, [])) == 1 and i != len(flow) - 1):
                j = readers[dst][0]
                psrc, pnames, pdst = flow[j]
                pe = body[pnames[0]]
                xs, ks = shapes.get(src[0].split("@")[0]), self._shape_of_init(src[1])
                para = {k: v for k, v in entry[2].items() if k in ("group", "strides", "dilations", "pads")}
                if (pe[1] == "maxpool_q4" and len(pnames) == 1 and (psrc == dst or psrc == [dst]) and xs is not None
                        and [int(v) for v in pe[2].get("w", (2, 2))] == [3, 3]
                        and [int(v) for v in pe[2].get("strides", (2, 2))] == [2, 2]
                        and [int(v) for v in pe[2].get("pads", (0, 0, 0, 0))] == [1, 1, 1, 1]
                        and int(entry[2].get("act", 0)) in (0, 1, 2) and _q4.stem_pool_eligible(tuple(xs), tuple(ks), **para)):
                    lay, fsrc = 10, list(src[:5])
                    if (os.environ.get("PLANER_HIP_STEM_NCHW", "1") != "0" and src[1].endswith("@rowpack")
                            and _q4.stem_pool_nchw_eligible(tuple(xs), tuple(ks), **para)):
                        # W % 4 == 0: the kernel reads the NCHW batch itself (no row-packed copy), filter in its own k order
                        lay, key = 12, src[1][:-len("@rowpack")] + "@stemnchw"
                        if key not in self._extra:
                            self._extra[key] = _q4.prepare_stem_nchw_weights(dict(zip(self.inits, self.weights))[src[1][:-len("@rowpack")]])
                        fsrc[1] = key
                    extra = {}
                    if lay == 12 and self._pick_mode == "throughput" and os.environ.get("PLANER_HIP_STEM_ROWS14", "1") != "0":
                        # throughput plans: strips of 14 pooled rows where that still leaves half a chip of workgroups -- 15
                        # conv-row pairs instead of 2 x 8, on half the CUs for longer (160 against 91 us alone; +0.8 ... +1.2 % on
                        # seven replicas: the other replicas' kernels take the rest of the chip)
                        hq = ((xs[2] + 6 - 7) // 2 + 1 + 1) // 2
                        if xs[0] * (-(-hq // 14)) * (-(-ks[0] // 64)) * 2 >= self.ctx.cu_count:
                            extra["strip_rows"] = 14
                    body[names[0]] = [entry[0], "conv_pool_q4", dict(entry[2], w_layout=lay, **extra)]
                    out.append([fsrc, names, pdst])
                    drop.add(j)
                    continue
            if i not in drop:
                out.append([src, names, dst])
        return out

    def _pick_conv_algo(self, ConvFused, K, srcs, para, shapes, wmap, q4=False):
        """Time the direct implicit GEMM and the Winograd variants for this conv's real shape and
        epilogue; -> w_layout 1 or 3 (NCHW), 2 / 8 / 4 / 7 / 9 (channel-quad).  Cached per shape
        signature (and persisted, see `algo_cache`); `force_algo` bypasses the measurement."""
        from .layer import prepare_conv_weights, prepare_winograd_weights
        xs = tuple(shapes[srcs[0].split("@")[0]])
        has = [i < len(srcs) and srcs[i] != "None" for i in range(2, 6)]       # B, scale, shift, res
        sig = (q4, xs, tuple(K.shape), tuple(has), para.get("act", 0))
        cands = [(1, prepare_conv_weights), (3, prepare_winograd_weights)]
        if q4:
            # direct, fused 1-D Winograd F(4,3) along W, 2-D Winograd pipelines with separate transform kernels, fully fused F(4x4,3x3)
            cands = [(2, _q4.prepare_q4_weights), (8, _q4.prepare_w1d4_q4_weights)]
            if _q4.winograd_q4_eligible(K.shape, **{k: v for k, v in para.items()
                                                   if k in ("group", "strides", "dilations", "pads")}):
                cands.append((4, _q4.prepare_winograd_q4_weights))
                if os.environ.get("PLANER_HIP_WINOGRAD4", "1") != "0":
                    cands.append((7, _q4.prepare_winograd4_q4_weights))      # F(4x4,3x3), staged
                if os.environ.get("PLANER_HIP_WF4", "1") != "0":
                    cands.append((9, _q4.prepare_wf4_q4_weights))            # F(4x4,3x3), one fused kernel
                if os.environ.get("PLANER_HIP_WINOGRAD43", "1") != "0" and _q4.winograd43_eligible(xs, K.shape, 64, **{
                        k: v for k, v in para.items() if k in ("group", "strides", "dilations", "pads")}):
                    cands.append((11, _q4.prepare_winograd43_q4_weights))    # mixed F(4,3) x F(3,3) tiles, staged
        if self.force_algo is not None:
            if self.force_algo not in [c[0] for c in cands]:
                raise ValueError("force_algo=%r does not apply to conv %s k%s" % (self.force_algo, xs, tuple(K.shape)))
            return self.force_algo
        self._load_algo_cache()
        # Throughput plans first ask the table of picks made UNDER the pipeline (tools/pipeline_search.py): a kernel that holds
        # a fraction of the CUs for longer loses an isolated timing and can win there -- the other replicas' kernels take the rest
        # of the chip (ResNet-18 layer2 at batch 32: the fused F(4x4,3x3) kernel on 128 workgroups, 63 us against 43 us staged, and
        # +1.9 % on seven replicas).  Latency plans (net(x) one call at a time) keep the isolated picks.
        if self._pick_mode == "throughput" and self._algo_tp.get(sig) in [c[0] for c in cands]:
            return self._algo_tp[sig]
        if sig in self._algo and self._algo[sig] in [c[0] for c in cands]:      # (a stored pick this run's switches exclude is ignored)
            return self._algo[sig]
        self.algo_misses += 1
        ctx = self.ctx
        # random operands: zero tensors clock ~19 % higher (DVFS) and would bias the pick towards
        # the MFMA-heavy candidates (MI355X guide, "DVFS give-back")
        rng = numpy.random.default_rng(1234)
        x = hip.asarray(rng.standard_normal(xs).astype(numpy.float32), ctx=ctx)
        cout = K.shape[0]
        chan = hip.asarray(rng.uniform(0.5, 1.5, (1, cout, 1, 1)).astype(numpy.float32), ctx=ctx)
        out_shape = (xs[0], cout, xs[2], xs[3])
        res = hip.asarray(rng.standard_normal(out_shape).astype(numpy.float32), ctx=ctx) if has[3] else None
        if q4:
            x, res = _q4.to_q4(x), (_q4.to_q4(res) if res is not None else None)
        args = [hip.zeros((cout,), numpy.float32, ctx) if has[0] else None, chan if has[1] else None,
                chan if has[2] else None, res]
        kw = {k: v for k, v in para.items() if k != "w_layout"}
        best, best_ms = cands[0][0], None
        for lay, prep in cands:
            try:
                Kp = prep(K)
                run = lambda: ConvFused(x, Kp, *args, w_layout=lay, **kw)
                for _ in range(3):
                    run()                              # first call autotunes the MFMA plan(s)
                ms = None
                for _ in range(3):                     # best of three bursts of 8
                    e0 = hip.Event(ctx).record()
                    for _ in range(8):
                        run()
                    e1 = hip.Event(ctx).record()
                    t = e0.elapsed_ms(e1) / 8
                    ms = t if ms is None else min(ms, t)
            except (NotImplementedError, ValueError, MemoryError):
                continue
            if os.environ.get("PLANER_CONV_TUNE_LOG"):
                import sys
                print("[planer_amd] conv %s k%s tail %s: w_layout %d = %.1f us" % (xs, tuple(K.shape), has, lay, ms * 1e3),
                      file=sys.stderr)
            if best_ms is None or ms < best_ms:
                best, best_ms = lay, ms
        self._algo[sig] = best
        self._algo_dirty = True
        return best

    @staticmethod
    def _sig_key(sig):
        return repr(sig)

    def _device_tag(self):
        ctx = self.ctx or hip.context()
        return "%s cu%d lib%d" % (ctx.arch.split(":")[0], ctx.cu_count, _lib.load().pl_version())

    def _load_algo_cache(self):
        """Shipped database first (planer_amd/tuned/<arch>_cu<N>.algo.json), then the user's cache on top.  A file
        written for another device or library version is ignored."""
        if self._algo_loaded:
            return
        self._algo_loaded = True
        import ast
        import json
        ctx = self.ctx or hip.context()
        shipped = ctx.tuned_db + ".algo.json" if getattr(ctx, "tuned_db", None) else None
        for path in (shipped, self.algo_cache):
            if not path:
                continue
            try:
                with open(path) as f:
                    stored = json.load(f)
            except (OSError, ValueError):
                continue
            if "algo" not in stored:                       # round-2 files: a flat {signature: layout} table
                stored = {"device": None, "algo": stored, "streams": {}}
            if stored.get("device") not in (None, self._device_tag()):
                continue
            for table, dst in (("algo", self._algo), ("algo_throughput", self._algo_tp)):
                for k, v in stored.get(table, {}).items():
                    try:
                        dst[ast.literal_eval(k)] = int(v)
                    except (ValueError, SyntaxError):
                        pass
            self._streams_pick.update({str(k): str(v) for k, v in stored.get("streams", {}).items()})

    def save_algo_cache(self, path=None):
        """Persist the per-shape conv algorithm choices and the stream-plan choices (next to PLANER_HIP_TUNE_CACHE
        by default): written to a temporary file and renamed, by rank 0 only."""
        path = path or self.algo_cache
        if not path or os.environ.get("RANK", "0") != "0":
            return
        import json
        data = {"device": self._device_tag(),
                "algo": {self._sig_key(k): v for k, v in sorted(self._algo.items(), key=repr)},
                "algo_throughput": {self._sig_key(k): v for k, v in sorted(self._algo_tp.items(), key=repr)},
                "streams": dict(sorted(self._streams_pick.items()))}
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with open(tmp, "w") as f:
            json.dump(data, f, indent=1)
        os.replace(tmp, path)

    def graph_tag(self):
        """A short structural hash of the loaded graph: stored stream-plan picks are keyed by it, so that another net with
        the same input shape and step count cannot take them."""
        return graph_tag_of(self.layer, self.flow, [w.shape for w in self.weights])

    def tune_source(self):
        """Where this net's kernel choices came from, for run reports: "shipped" (every launch plan, algorithm and
        stream plan was in the database shipped for this device), "cache", or what had to be measured."""
        ctx = self.ctx
        _, plan_misses = ctx.tune_stats()
        for c in self._side:
            plan_misses += c.tune_stats()[1]
        base = "shipped" if getattr(ctx, "tuned_db", None) else None
        if ctx.tune_cache and os.path.exists(ctx.tune_cache):
            base = (base + " + cache") if base else "cache"
        miss = plan_misses + self.algo_misses + self.stream_misses
        if not miss:
            return base or "static heuristics (autotune off)"
        what = "autotuned: %d launch plans, %d conv algorithms, %d stream plans" % (plan_misses, self.algo_misses, self.stream_misses)
        return "%s; %s" % (base, what) if base else what

    def picking(self, mode):
        """Context: programs fused inside take the conv algorithms of `mode` ("throughput": the pipeline-judged table first)."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            prev, self._pick_mode = self._pick_mode, mode
            try:
                yield self
            finally:
                self._pick_mode = prev
        return cm()

    def compile(self, *xs, mode="latency"):
        """(see _compile: the plan's program is fused under this mode's conv algorithm tables)"""
        with self.picking(mode):
            return self._compile(*xs, mode=mode)

    def _compile(self, *xs, mode="latency"):
        """Build (or fetch) the captured plan for these device inputs.  `mode` only affects how
        the number of sub-batch streams is chosen: "latency" measures fork/join passes (what
        __call__ does), "throughput" measures free-running back-to-back passes."""
        key = (mode,) + tuple((a.shape, str(a.dtype)) for a in xs)
        plan = self._plans.get(key)
        if plan is not None:
            self._ensure_streams(plan)
            return plan
        ctx = self.ctx
        shapes = {k: a.shape for k, a in zip(self.input, xs)}
        shapes.update({k: w.shape for k, w in zip(self.inits, self.weights)})
        # unfused eager pass: validates the graph and records every shape.
        # ReLU works in place, so feed it copies, never the caller's arrays.
        timer = dict(self.timer)
        self._interpret(self._program, [a.copy() for a in xs], shapes=shapes)
        batch = xs[0].shape[0] if xs and xs[0].ndim else 0
        # A plan is (Q, P): the batch is cut into P sub-batches, each its own captured graph,
        # dealt round-robin onto Q streams.  Q > 1 lets two kernels overlap (one's tail and ramp
        # under the other's body); P > Q keeps per-kernel grids small enough to interleave well.
        splittable = batch > 0 and all(a.ndim and a.shape[0] == batch for a in xs)
        want = str(self.streams)
        if want == "auto":
            cands = [(1, 1)] + [(q, p_) for q, p_ in ((2, 2), (2, 4), (4, 4))
                                if splittable and batch % p_ == 0 and batch // p_ >= 4]
        elif want.startswith("pipe"):
            cands = []
        else:
            q, _, p_ = want.partition("x")
            q, p_ = int(q), int(p_ or q)
            cands = [(q, p_) if splittable and p_ >= q >= 1 and batch % p_ == 0 else (1, 1)]
        # throughput mode pipelines whole batches over R streams (R full-batch graphs); sub-batch plans
        # are dominated there (28.4 k vs 34.8 k img/s on ResNet-18), so they are not even tried
        # (depths: 3, 7 and 15 replicas interleave best -- depths that are multiples of the four hardware queues the runtime
        #  maps streams onto lose 3-8 %: ResNet-18 at batch 32 in steady state 52.3 / 53.1 / 53.6 k img/s at 3 / 7 / 15 against
        #  48.4 / 50.8 k at 4 / 8; YOLO-v3 at batch 1 1.54 / 1.60 / 1.64 k; a lone write-bound conv is best at 3)
        pipes = [2, 3, 7, 15] if (mode == "throughput" and want == "auto") else []
        if pipes:
            cands = [(1, 1)]
        if want.startswith("pipe"):
            cands, pipes = [], [int(want[4:] or 2)]
        # a stream plan chosen for this (mode, inputs, graph) before -- shipped database or the user's cache -- is taken
        # without measuring, so that consecutive runs time the same thing
        self._load_algo_cache()
        pick_key = repr((mode, [tuple(a.shape) for a in xs], "%d:%s" % (len(self.flow), self.graph_tag()), want))
        stored = self._streams_pick.get(pick_key) if want == "auto" else None
        if stored and stored.startswith("pipe") and mode == "throughput":
            cands, pipes = [], [int(stored[4:])]
        elif stored and "x" in stored:
            q, _, p_ = stored.partition("x")
            if (int(q), int(p_)) in cands:
                cands, pipes = [(int(q), int(p_))], []
        progs = {}

        def program_for(P):
            """The fused program with conv algorithms (direct vs Winograd) picked for the shapes a
            sub-batch of batch/P images really has."""
            if P not in progs:
                sub = dict(shapes)
                if P > 1:
                    for k, shp in shapes.items():
                        if k not in self.inits and shp and len(shp) >= 1 and shp[0] == batch:
                            sub[k] = (batch // P,) + tuple(shp[1:])
                progs[P] = self._fuse(sub, self.use_fusion)
            return progs[P]
        best = None
        todo = [("sub", c) for c in cands] + [("pipe", r) for r in pipes]
        for how, c in todo:
            if how == "sub":
                Q, P = c
                prog, nfused = program_for(P)
                try:
                    cand = self._build_plan(prog, xs, Q, P, nfused)
                except _NotSplittable:
                    continue
                except _lib.NotCapturable:
                    self._eager_prog[key[1:]] = program_for(1)[0]      # Net.__call__ runs this one eagerly
                    self.timer = timer
                    raise
            else:
                prog, nfused = program_for(1)
                reps = []
                try:
                    for r in range(c):
                        cx = self._side_context(r)
                        reps.append(self._capture(prog, [DeviceArray(a.shape, a.dtype, cx).copy_from(a) for a in xs],
                                                  cx, nfused))
                        ctx.synchronize()
                except MemoryError:
                    # a deep pipeline is R full copies of the activations: a candidate the device has no room for is
                    # skipped (a depth that was asked for by name is not)
                    if len(todo) == 1 or best is None:
                        raise
                    del reps
                    continue
                cand = _PipelinePlan(reps, ctx, nfused)
                self._probe_streams(cand, xs)
            if len(todo) == 1:
                best = cand
                break
            def burst(k):
                for _ in range(k):
                    if mode == "throughput":
                        cand.feed(xs)
                    cand.launch(join=mode != "throughput")
                cand.join()
                ctx.synchronize()
            burst(5)
            cand.ms = None
            # latency plans: best of five bursts of 10 passes.  Pipelined (throughput) plans are judged in steady
            # state -- best of three runs of 100 passes -- because a 10-pass burst is mostly pipeline fill and drain
            # and under-rates the deeper pipeline (host run-ahead is bounded by _Plan.max_in_flight)
            reps, depth = (3, 100) if how == "pipe" else (5, 10)
            for _ in range(reps):
                t0 = time.perf_counter()
                burst(depth)
                ms = (time.perf_counter() - t0) / depth * 1e3
                cand.ms = ms if cand.ms is None else min(cand.ms, ms)
            if os.environ.get("PLANER_PLAN_LOG"):
                import sys
                print("[planer_amd] plan candidate %s: %.4f ms/pass" % (cand.streams, cand.ms), file=sys.stderr)
            if best is None or cand.ms < best.ms:
                best = cand
        if len(todo) > 1 and want == "auto":
            self.stream_misses += 1
            self._streams_pick[pick_key] = best.streams
            self._algo_dirty = True
        self.timer = timer
        self._plans[key] = best
        # every pipeline candidate's probe rotated the process-wide side-stream pool for ITS depth: what is left is the last
        # candidate's assignment, not the winner's -- re-apply the one the winner was measured (and picked) under
        self._ensure_streams(best)
        if getattr(self, "_algo_dirty", False):
            self.save_algo_cache()
            self._algo_dirty = False
        return best

    def _probe_streams(self, plan, xs):
        """Which hardware queue the runtime put a stream on follows the order in which the process created its streams, and a
        pipeline's rate depends on where its replicas land (depth 7: 53.1 k img/s or 50.4 k) -- a host that
        created streams of its own first, or RCCL at world > 1, shifts the map.  So the replicas are not tied to the streams they
        were captured on: with three spare side streams behind the ones the pipeline needs, every rotation of the assignment
        (hip.set_side_stream_shift, 0..3 = one period of the four queues) runs the plan for a few milliseconds and the fastest
        stays.  PLANER_HIP_STREAM_PROBE=0 keeps creation order, =<s> forces a shift."""
        want = os.environ.get("PLANER_HIP_STREAM_PROBE", "auto")
        nrep = len(plan.replicas)
        if want == "0" or nrep < 2:
            return
        nside = nrep - 1 + 3
        dev, ctx = self.ctx.device, self.ctx

        def run(rounds):
            for _ in range(rounds * nrep):
                plan.feed(xs)
                plan.launch(join=False)
            plan.join()
            ctx.synchronize()
        if want != "auto":
            hip.set_side_stream_shift(dev, nside, int(want))
            plan.stream_probe = {"shift": int(want), "forced": True}
            return
        ms = []
        for s_ in range(4):
            hip.set_side_stream_shift(dev, nside, s_)
            run(1)
            best = None
            for _ in range(2):
                t0 = time.perf_counter()
                run(3)
                t = (time.perf_counter() - t0) / (3 * nrep) * 1e3
                best = t if best is None else min(best, t)
            ms.append(round(best, 5))
        pick = min(range(4), key=lambda i: ms[i])
        if ms[0] <= ms[pick] * 1.01:             # creation order unless another assignment is clearly (> 1 %) faster
            pick = 0
        hip.set_side_stream_shift(dev, nside, pick)
        plan.stream_probe = {"shift": pick, "ms_per_pass": ms}

    def _ensure_streams(self, plan):
        """The side-stream assignment is process-wide state (hip.set_side_stream_perm): another candidate's probe, or another
        Net's throughput plan compiled since, may have rotated it.  A pipeline plan carries the assignment it was measured
        under (`stream_probe`: shift over nrep + 2 side streams) and puts it back before it runs -- a few stream syncs, and only
        when something else changed it."""
        sp = getattr(plan, "stream_probe", None)
        if not isinstance(plan, _PipelinePlan) or not sp or "shift" not in sp:
            return
        n = len(plan.replicas) - 1 + 3
        want = [(i + int(sp["shift"])) % n for i in range(n)]
        if hip.side_stream_perm(self.ctx.device)[:n] != want:
            hip.set_side_stream_shift(self.ctx.device, n, sp["shift"])

    def _side_context(self, i):
        """Extra stream (context) number i of this net's device; 0 is the net's own."""
        while len(self._side) < i:              # the device's shared pool, in index order (hip.side_context)
            self._side.append(hip.side_context(self.ctx.device, len(self._side) + 1))
        return self.ctx if i == 0 else self._side[i - 1]

    def _build_plan(self, prog, xs, Q, S, nfused):
        """Q streams, S sub-batch graphs (S >= Q)."""
        if S == 1:
            return self._capture(prog, [DeviceArray(a.shape, a.dtype, self.ctx).copy_from(a) for a in xs],
                                 self.ctx, nfused)
        # S sub-batches on S streams: full-size static inputs/outputs live on the net's own
        # context; every sub-graph copies its input rows in and its output rows out itself.
        ctx0 = self.ctx
        full_in = [DeviceArray(a.shape, a.dtype, ctx0).copy_from(a) for a in xs]
        ctx0.synchronize()
        n = xs[0].shape[0] // S
        state = {"full_out": None}

        def dst_views(i, warm):
            outs = warm if isinstance(warm, tuple) else (warm,)
            if any(not isinstance(o, DeviceArray) or not o.ndim or o.shape[0] != n for o in outs):
                raise _NotSplittable()
            if state["full_out"] is None:
                state["full_out"] = [DeviceArray((n * S,) + o.shape[1:], o.dtype, ctx0) for o in outs]
            return [f.rows(i * n, (i + 1) * n) for f in state["full_out"]]

        subs = []
        for i in range(S):
            c = self._side_context(i % Q)
            # the sub-plan reads its rows of the full input in place: an alias view that belongs
            # to the side context, so every op that consumes it runs on the side stream
            statics = []
            for f in full_in:
                v = f.rows(i * n, (i + 1) * n)
                statics.append(DeviceArray(v.shape, v.dtype, c, v.ptr, f))
            subs.append(self._capture(prog, statics, c, nfused))
            dst_views(i, subs[-1].outputs)
        outs = state["full_out"]
        was_tuple = isinstance(subs[0].outputs, tuple)
        plan = _MultiPlan(full_in, tuple(outs) if was_tuple else outs[0], subs, ctx0, nfused)
        plan.streams = "%dx%d" % (Q, S)
        plan.out_rows = [[f.rows(i * n, (i + 1) * n) for f in outs] for i in range(S)]
        return plan

    def _capture(self, prog, statics, ctx, nfused, srcs=None, make_dsts=None):
        """Warm the pool / tuner with one eager pass of `prog` on `ctx`, then capture it."""
        def fill():
            if srcs is not None:
                for s_, v in zip(statics, srcs):
                    _lib.call("pl_d2d", ctx.handle, s_.ptr, v.ptr, s_.nbytes)
        fill()
        algos = []
        warm = self._interpret(prog, [s_.copy() for s_ in statics], record=algos)
        dsts = make_dsts(warm) if make_dsts else None
        del warm
        ctx.synchronize()
        # A static input is only copied first if a step could overwrite it in place.
        kinds = {b[0]: b[1] for b in prog_body(prog)}
        inplace = set()
        for src, names, dst in prog.flow:
            if kinds.get(_as_list(names)[0]) in ("relu", "relu_q4", "flatten", "identity", "return"):
                inplace.update(_as_list(src))
        self._pack_static_inputs(prog, statics, inplace)
        _lib.call("pl_capture_begin", ctx.handle)
        try:
            fill()
            work = []
            for k, s_ in zip(self.input, statics):
                if k in inplace:
                    s_ = DeviceArray(s_.shape, s_.dtype, ctx).copy_from(s_)
                work.append(s_)
            out = self._interpret(prog, work)
            del work
            if dsts is not None:
                for o, d in zip(out if isinstance(out, tuple) else (out,), dsts):
                    _lib.call("pl_d2d", ctx.handle, d.ptr, o.ptr, o.nbytes)
        except Exception:
            g = _lib.c_void_p()
            _lib.load().pl_capture_end(ctx.handle, _lib.byref(g))
            if g.value:
                _lib.load().pl_graph_destroy(g)
            raise
        g = _lib.c_void_p()
        _lib.call("pl_capture_end", ctx.handle, _lib.byref(g))
        plan = _Plan(statics, out, g, ctx, nfused)
        plan.algos = algos
        return plan

    def _pack_static_inputs(self, prog, statics, inplace):
        """A graph input whose ONLY reader is the row-packed stem conv (w_layout 6) gets its row-packed image
        as a persistent buffer beside it (`static.packed`): the captured conv reads that image, the re-layout kernel stays
        out of the graph and runs when the plan is fed (`_feed_static`) -- as the copy that brings the batch in.
        PLANER_HIP_FEED_PACK=0 keeps the re-layout inside the graph."""
        if os.environ.get("PLANER_HIP_FEED_PACK", "1") == "0":
            return
        for k, s_ in zip(self.input, statics):
            readers = [(src, names) for src, names, dst in prog.flow if k in _as_list(src)]
            if len(readers) != 1 or k in inplace or s_.packed is not None or len(s_.shape) != 4 or s_.base is not None:
                continue
            src, names = readers[0]
            obj = prog.objs[_as_list(names)[0]]
            para = obj.para()
            if (obj.name == "conv_pool_q4" and para.get("w_layout") == 12 and _as_list(src)[0] == k and _as_list(src).count(k) == 1
                    and s_.prefed is None and s_.ptr % 16 == 0 and os.environ.get("PLANER_HIP_FEED_STEM", "1") != "0"):
                # the stem + max-pool kernel reads NCHW itself: it leaves the graph and runs when the plan is fed, straight
                # from the caller's batch; the captured pass starts from its (persistent) pooled tensor
                env = {"None": None}
                env.update(zip(self.inits, self.weights))
                env.update(self._extra)
                args = [env.get(a) for a in (_as_list(src)[1:5] + ["None"] * 4)[:4]]
                fpara = {a: b for a, b in para.items() if a != "w_layout"}
                pooled = _q4.ConvPoolQ4(s_, *args, w_layout=12, **fpara)
                s_.prefed = (_q4.stem_pool_feeder(s_.shape, args[0], args[1], args[2], args[3], fpara.get("act", 0),
                                                  fpara.get("alpha", 0.0), pooled, fpara.get("strip_rows", 0)), pooled)
                continue
            if (obj.name not in ("conv_q4", "conv_pool_q4") or para.get("w_layout") not in (6, 10) or _as_list(src)[0] != k
                    or _as_list(src).count(k) != 1):
                continue
            kw = self._shape_of_init(_as_list(src)[1])[3]
            strides, pads = para.get("strides", (1, 1)), para.get("pads", (0, 0, 0, 0))
            _q4.pack_rows(s_, geom=(int(kw), int(strides[1]), int(pads[0]), int(pads[1])))

    def _shape_of_init(self, key):
        v = dict(zip(self.inits, self.weights)).get(key)
        if v is None:
            v = self._extra[key]
        return v.shape

    def _replay(self, xs, private=True):
        plan = self.compile(*xs)
        if isinstance(plan, _PipelinePlan):
            # the replica whose turn it is gets the inputs (plan.inputs are those of the replica launched LAST);
            # its stream first waits for the main stream, where the caller produced xs
            plan.replicas[plan.turn].ctx.wait_for(self.ctx)
            plan.feed(xs)
        else:
            self._feed_on_main(plan, xs)
        plan.launch()
        out = plan.outputs
        if not private:                            # the caller copies to the host right away
            return out
        # hand back private copies: the plan's buffers are rewritten by the next replay
        if isinstance(out, tuple):
            return tuple(o.copy() if isinstance(o, DeviceArray) else o for o in out)
        return out.copy() if isinstance(out, DeviceArray) else out

    def _feed_on_main(self, plan, xs):
        """Inputs of a one-graph or sub-batch plan, copied on the net's OWN stream (where the caller produced them);
        `launch()` forks the side streams behind that point."""
        for s, a in zip(plan.inputs, xs):
            if isinstance(plan, _MultiPlan) or not isinstance(a, DeviceArray):
                if a is not s:
                    s.copy_from(a)
                if s.prefed is not None or s.packed is not None:        # (a host array: through the static tensor, then the feed pass)
                    _feed_static(self.ctx, s, s)
            else:
                _feed_static(self.ctx, s, a)
        for sp in (plan.subs if isinstance(plan, _MultiPlan) else (plan,)):
            sp._fed = True

    def submit(self, *x):
        """Asynchronous form of `net(x)` (net.py:94-101): enqueue one forward pass and return a `Pending` handle at once.
        Consecutive submits rotate over the replicas of the throughput plan (R full-batch graphs on R streams,
        `compile(mode="throughput")`), so one batch's tails and memory-bound layers run under the next one's convs --
        a loop of submits reaches the rate `bench.py` reports for the plan API.  Each handle owns private copies of its
        outputs (the replica's buffers are rewritten R submits later)."""
        if type(x[0]) is dict:
            x = [x[0][i] for i in self.input]
        host = [isinstance(i, numpy.ndarray) for i in x]
        plan = None
        route = os.environ.get("PLANER_HIP_HOST_ROUTE", "direct")          # direct | staged | plain (A/B: tools/host_submit_probe.py)
        direct = False
        if any(host) and self.use_graph and not self.profile and route != "plain":
            # a pipeline compiled for this signature: a host batch goes straight into an input buffer that belongs to the
            # replica whose turn it is -- a synchronous DMA on NO stream (pl_h2d_direct: the host waits 0.35 ms for a 19 MB
            # batch while the other replicas compute; a copy enqueued on a stream would hold that stream's hardware queue, and
            # the replica that shares it, for as long).  The buffer is free once the replica's previous feed has read it.
            hx = [numpy.require(i, requirements="C") if b else i for i, b in zip(x, host)]
            plan = self._plans.get(("throughput",) + tuple((a.shape, str(a.dtype)) for a in hx))
            if isinstance(plan, _PipelinePlan) and route == "staged":       # pinned ring + DMA on the replica's stream (pl_h2d_staged)
                self._ensure_streams(plan)
                xs = [hip.asarray(i, ctx=self.ctx, consumer=plan.replicas[plan.turn].ctx) if b else i for i, b in zip(hx, host)]
            elif isinstance(plan, _PipelinePlan):
                self._ensure_streams(plan)
                rp, direct = plan.replicas[plan.turn], True
                if rp.host_in is None:
                    rp.host_in, rp.fed = [None] * len(hx), hip.Event(rp.ctx)
                else:
                    rp.fed.synchronize()              # (recorded R submits ago)
                xs = []
                for i, (a, b) in enumerate(zip(hx, host)):
                    if b:
                        d = rp.host_in[i]
                        if d is None or d.shape != a.shape or d.dtype != a.dtype:
                            d = rp.host_in[i] = DeviceArray(a.shape, a.dtype, rp.ctx)
                            rp.ctx.synchronize()      # (a fresh pool block: its previous reader was enqueued on this stream)
                        if a.nbytes:
                            _lib.call("pl_h2d_direct", rp.ctx.handle, d.ptr, a.ctypes.data, a.nbytes)
                        a = d
                    xs.append(a)
            else:
                plan = None
        if plan is None:
            xs = [hip.asarray(i, ctx=self.ctx) if b else i for i, b in zip(x, host)]
            if self.use_graph and not self.profile:
                try:
                    plan = self.compile(*xs, mode="throughput")
                except _lib.NotCapturable:
                    self.use_graph = False
        if plan is None:                              # flows that need the host between kernels: run now, hand back a finished handle
            return Pending(self, self(*x), None, any(host), final=True)
        if isinstance(plan, _PipelinePlan):
            rp = plan.replicas[plan.turn]
            if not (direct and all(host)):
                rp.ctx.wait_for(self.ctx)             # the caller produced xs on the net's own stream
            plan.feed(xs)
            if direct:
                rp.fed.record()                       # the replica's upload buffers are free again behind this point
            # what the feed read stays referenced until the replica has read it (the replica's own upload buffers live on anyway)
            keep = [a for a, b in zip(xs, host) if not (direct and b)]
            plan.launch(join=False, hold=keep or None)
            cx, out = rp.ctx, rp.outputs
        else:
            # sub-batch plans ("QxP"): the copy runs on the net's own stream, the sub-streams fork behind it and join
            # back into it (the feed of _MultiPlan.feed on the side streams could read xs before the main stream wrote them)
            self._feed_on_main(plan, xs)
            plan.launch()
            cx, out = self.ctx, plan.outputs
        # private copies on the replica's stream, then the marker result() waits for
        outs = tuple(o.copy() if isinstance(o, DeviceArray) else o for o in out) if isinstance(out, tuple) else out.copy()
        ev = self._events.pop() if self._events else hip.Event(cx)      # a marker can be recorded on any stream of its device
        _lib.call("pl_event_record", cx.handle, ev.handle)
        tickets = None
        if any(host):                                 # host in -> host out: the copies back start now, behind this pass
            seq = outs if isinstance(outs, tuple) else (outs,)
            tickets = [o.get_begin(producer=cx) if isinstance(o, DeviceArray) else None for o in seq]
        return Pending(self, outs, ev, any(host), tickets=tickets)

    def _eager_program(self, xs):
        """The fused program for THESE input shapes (conv algorithms, Winograd chaining, row packing and the fusions are
        all decided per shape, so a program fused for one image size must not run another): built on first sight of a
        signature like a plan is, from an unfused pass that records every shape."""
        sig = tuple((a.shape, str(a.dtype)) for a in xs)
        prog = self._eager_prog.get(sig)
        if prog is None:
            shapes = {k: a.shape for k, a in zip(self.input, xs)}
            shapes.update({k: w.shape for k, w in zip(self.inits, self.weights)})
            timer = dict(self.timer)
            self._interpret(self._program, [a.copy() for a in xs], shapes=shapes)     # ReLU works in place: copies
            self.timer = timer
            prog = self._eager_prog[sig] = self._fuse(shapes, self.use_fusion)[0]
        return prog

    # ---- entry point ---------------------------------------------------------------------
    def __call__(self, *x, **key):
        """net.py:94-101."""
        if type(x[0]) is dict:
            x = [x[0][i] for i in self.input]
        host = [isinstance(i, numpy.ndarray) for i in x]
        need = any(host)
        if need:
            x = [hip.asarray(i, ctx=self.ctx) if b else i for i, b in zip(x, host)]
        graphable = (self.use_graph and not key.get("debug") and not self.profile
                     and all(isinstance(i, DeviceArray) for i in x))
        if graphable:
            try:
                rst = self._replay(list(x), private=not need)          # host in -> host out: .get() reads the plan's buffers
            except _lib.NotCapturable:
                # post-processing graphs (NonZero, uploads of host-computed index lists) need the host between
                # kernels: the FUSED program (same kernels, layouts and algorithm picks as the plan) is launched
                # step by step from here on
                self.use_graph = graphable = False
        if not graphable:
            if self._eager_prog and not key.get("debug") and not self.profile and all(isinstance(i, DeviceArray) for i in x):
                rst = self._interpret(self._eager_program(list(x)), list(x))
            else:
                rst = self.forward(*x, **key)
        if need:
            rst = tuple(i.get() for i in rst) if isinstance(rst, tuple) else rst.get()
        return rst[0] if len(rst) == 1 else rst
