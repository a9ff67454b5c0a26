"""Batch-sharded multi-GPU execution: one process per GPU, RCCL over xGMI.

The reference is a single-process library with no distributed code.  Its
forward pass shards naturally by batch -- every op is per image and BatchNorm
is pre-folded to a per-channel affine (io.py:76-91), so shards never talk.
The only exchange is at load time: rank 0 reads the weight file and ONE RCCL
broadcast of the uint8 blob (net.load_weights, net.py:83-88) fills every
other rank's copy.  No collective runs in the forward pass.

Launch model: `python -m planer_amd.launch --nproc N script.py ...` (the
package's own spawner, planer_amd/launch.py: no torch anywhere) or any launcher
that sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT (`python -m
torch.distributed.run --nproc-per-node N ...` is what the bench driver uses);
the launcher is only a process spawner -- this module does not import torch.
The 128-byte RCCL unique id travels through a file (PLANER_RDZV_FILE, else a
name under /tmp derived from the launcher's port and pid: all ranks share one node).
"""
import os
import time

import numpy

from . import _lib


def shard_range(total, world, rank):
    """Contiguous batch slice of `rank`: the first `total % world` ranks take
    one extra image."""
    base, extra = divmod(int(total), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))))


def _rendezvous_path():
    explicit = os.environ.get("PLANER_RDZV_FILE")
    if explicit:
        return explicit
    tag = "%s_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                        os.getppid())
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), "planer_amd_rdzv_" + tag)


def exchange_bytes(payload, rank, path=None, timeout=300.0, fresh_after=None):
    """Rank 0 publishes `payload` (bytes) atomically; the others poll for it.
    A file older than `fresh_after` (epoch seconds) is a leftover of an
    earlier job that reused the pid/port and is ignored."""
    path = path or _rendezvous_path()
    if rank == 0:
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with open(tmp, "wb") as f:
            f.write(payload)
        os.replace(tmp, path)
        return payload
    fresh_after = time.time() - 600.0 if fresh_after is None else fresh_after
    deadline = time.time() + timeout
    while time.time() < deadline:
        try:
            if os.path.getmtime(path) >= fresh_after:
                with open(path, "rb") as f:
                    data = f.read()
                if data:
                    return data
        except OSError:
            pass
        time.sleep(0.02)
    raise TimeoutError("rank %d: no rendezvous file %s" % (rank, path))


class Communicator:
    """What the sharded path needs from a transport.  `RcclCommunicator` is
    the product; tests drive the same host logic over torch.distributed/gloo."""
    rank, world = 0, 1
    device_transport = True      # can move device memory between ranks (weight broadcast)

    def bcast_device(self, arr, root=0):
        raise NotImplementedError

    def barrier(self):
        raise NotImplementedError

    def max_over_ranks(self, value):
        raise NotImplementedError

    def min_over_ranks(self, value):
        return -self.max_over_ranks(-float(value))

    bcast_ms = None      # device time of the last weight broadcast, max over ranks (load_weights)

    def load_weights(self, net, blob, root=0):
        """Rank `root` uploads the blob; everyone else receives the device
        copy by broadcast, then refreshes the host mirror of the small
        parameter tensors (UpSample scales are read on the host).  The
        broadcast alone is timed: barrier + device sync on both sides, the
        maximum over ranks lands in `bcast_ms`."""
        sync = getattr(getattr(net, "ctx", None), "synchronize", lambda: None)
        if self.rank == root:
            net.load_weights(blob)
        sync()
        self.barrier()
        t0 = time.perf_counter()
        self.bcast_device(net.weight_blob(), root)
        sync()
        self.bcast_ms = self.max_over_ranks((time.perf_counter() - t0) * 1e3)
        if self.rank != root:
            net.refresh_host_mirror()
        self.barrier()


class SingleProcess(Communicator):
    def bcast_device(self, arr, root=0):
        return arr

    def min_over_ranks(self, value):
        return float(value)

    def transport_ranks(self):
        return 1

    def barrier(self):
        pass

    def max_over_ranks(self, value):
        return float(value)


class RcclCommunicator(Communicator):
    def __init__(self, ctx, rank, world, rdzv_path=None):
        from . import hip
        self.ctx, self.rank, self.world = ctx, rank, world
        lib = _lib.load()
        uid = (_lib.ctypes.c_char * _lib.UNIQUE_ID_BYTES)()
        if rank == 0:
            _lib.check(lib.pl_comm_unique_id(uid))
        data = exchange_bytes(bytes(uid.raw), rank, rdzv_path)
        uid = (_lib.ctypes.c_char * _lib.UNIQUE_ID_BYTES).from_buffer_copy(data[:_lib.UNIQUE_ID_BYTES])
        _lib.check(lib.pl_comm_init_rank(ctx.handle, world, rank, uid))
        self._scratch = hip.zeros((4,), numpy.float32, ctx)
        ctx.comm = self

    def bcast_device(self, arr, root=0):
        _lib.call("pl_comm_bcast", self.ctx.handle, arr.ptr, arr.nbytes, root)
        return arr

    def max_over_ranks(self, value):
        self._scratch.set(numpy.full(4, value, numpy.float32))
        _lib.call("pl_comm_allreduce_max_f32", self.ctx.handle, self._scratch.ptr, 4)
        return float(self._scratch.get()[0])

    def min_over_ranks(self, value):
        return -self.max_over_ranks(-float(value))

    def transport_ranks(self):
        """How many ranks RCCL itself counts in the communicator (ncclCommCount)."""
        n, r = _lib.c_int(), _lib.c_int()
        _lib.call("pl_comm_info", self.ctx.handle, _lib.byref(n), _lib.byref(r))
        if r.value != self.rank:
            raise RuntimeError("RCCL says this is rank %d, the launcher said %d" % (r.value, self.rank))
        return n.value

    def barrier(self):
        self.max_over_ranks(0.0)           # an all-reduce is a barrier; .get() syncs the stream

    def allgather_rows(self, local):
        """(rows, ...) per rank -> (world*rows, ...) on every rank (equal shards)."""
        from . import hip
        out = hip.empty((self.world * local.shape[0],) + local.shape[1:], local.dtype, self.ctx)
        _lib.call("pl_comm_allgather", self.ctx.handle, local.ptr, out.ptr, local.nbytes)
        return out

    def close(self):
        _lib.load().pl_comm_destroy(self.ctx.handle)


class FileCommunicator(Communicator):
    """Same-node fallback when RCCL cannot be brought up: barrier and max-over-ranks go through
    small files next to the rendezvous file.  It cannot move device memory, so `load_weights`
    needs the blob on every rank (the callers' weights are seeded and can be regenerated locally);
    nothing in the forward pass depends on the transport."""
    device_transport = False

    def __init__(self, rank, world, base=None, timeout=600.0):
        self.rank, self.world, self.timeout = rank, world, timeout
        self.base = (base or _rendezvous_path()) + ".fc"
        self.seq = 0

    def _gather(self, value):
        self.seq += 1
        mine = "%s.%d.%d" % (self.base, self.seq, self.rank)
        tmp = mine + ".tmp"
        with open(tmp, "w") as f:
            f.write(repr(float(value)))
        os.replace(tmp, mine)
        vals, deadline = [], time.time() + self.timeout
        for r in range(self.world):
            path = "%s.%d.%d" % (self.base, self.seq, r)
            while True:
                try:
                    with open(path) as f:
                        txt = f.read()
                    if txt:
                        vals.append(float(txt))
                        break
                except OSError:
                    pass
                if time.time() > deadline:
                    raise TimeoutError("rank %d: rank %d never reached collective %d" % (self.rank, r, self.seq))
                time.sleep(0.002)
        if self.seq > 2:                       # everyone has passed collective seq-2 by now
            try:
                os.remove("%s.%d.%d" % (self.base, self.seq - 2, self.rank))
            except OSError:
                pass
        return vals

    def bcast_device(self, arr, root=0):
        raise RuntimeError("FileCommunicator cannot broadcast device memory")

    def barrier(self):
        self._gather(0.0)

    def max_over_ranks(self, value):
        return max(self._gather(value))

    def min_over_ranks(self, value):
        return min(self._gather(value))

    def transport_ranks(self):
        return 0                               # no RCCL communicator behind this transport

    def load_weights(self, net, blob, root=0):
        if blob is None:
            raise ValueError("the file fallback needs the weight blob on every rank")
        net.load_weights(blob)
        self.barrier()


class VirtualWorld:
    """SURVEY 8(e) "virtual ranks on one device": `world` ranks inside ONE process on ONE GPU, for
    checking the sharded path (BASELINE config 4: batch 256 = 8 x 32) where only one device is
    visible.  Every rank is its own context (stream + memory pool) with its own `Net` and its own
    full copy of the weight blob -- rank `root` uploads it, the others receive a device-to-device
    copy (what `ncclBroadcast` does between real ranks) -- and runs the forward pass of its
    `shard_range` slice independently; no rank ever reads another rank's activations."""

    def __init__(self, world, device=None):
        from . import hip
        base = hip.context()
        dev = base.device if device is None else int(device)
        self.world = int(world)
        self.ctxs = [base if (r == 0 and dev == base.device) else hip.Context(dev) for r in range(self.world)]

    def load(self, graph, blob, root=0):
        """-> one Net per rank; only `root` touches the host blob."""
        from .net import Net
        nets = []
        for r, ctx in enumerate(self.ctxs):
            net = Net(ctx)
            net.load_json(graph["input"], graph["inits"], graph["layers"], graph["flow"])
            nets.append(net)
        nets[root].load_weights(blob)
        self.ctxs[root].synchronize()
        src = nets[root].weight_blob()
        for r, net in enumerate(nets):
            if r != root:
                dst = net.weight_blob()
                _lib.call("pl_d2d", net.ctx.handle, dst.ptr, src.ptr, dst.nbytes)      # the "broadcast"
                net.ctx.synchronize()
                net.refresh_host_mirror()
        return nets

    def forward(self, nets, x):
        """Host batch in -> assembled host result out: rank r computes rows shard_range(N, world, r)
        on its own stream (all ranks are enqueued before any is read back)."""
        from . import hip
        n = x.shape[0]
        outs = []
        for r, net in enumerate(nets):
            lo, hi = shard_range(n, self.world, r)
            outs.append(None if hi == lo else net(hip.asarray(numpy.ascontiguousarray(x[lo:hi]), ctx=net.ctx)))
        host = []
        for o in outs:
            if o is not None:
                host.append(tuple(t.get() for t in o) if isinstance(o, tuple) else o.get())
        if host and isinstance(host[0], tuple):
            return tuple(numpy.concatenate([h[i] for h in host], axis=0) for i in range(len(host[0])))
        return numpy.concatenate(host, axis=0)


def init(ctx=None, fallback=False):
    """Communicator for this process from the launcher's environment.  With `fallback`, a failure
    to bring RCCL up yields a FileCommunicator (its `.why` says what went wrong) instead of raising."""
    from . import hip
    rank, world, _ = env_world()
    ctx = ctx or hip.context()
    if world == 1:
        return SingleProcess()
    if os.environ.get("PLANER_DIST_TRANSPORT") == "file":
        comm = FileCommunicator(rank, world)
        comm.why = "PLANER_DIST_TRANSPORT=file"
        return comm
    # the pipeline's side streams take their hardware-queue slots before RCCL creates streams of its own (hip.side_context)
    hip.reserve_side_contexts(ctx.device, int(os.environ.get("PLANER_HIP_RESERVE_STREAMS", "14")))
    try:
        return RcclCommunicator(ctx, rank, world)
    except Exception as e:                    # noqa: BLE001 -- any RCCL / rendezvous failure
        if not fallback:
            raise
        comm = FileCommunicator(rank, world)
        comm.why = "RCCL unavailable: %s" % (str(e)[:200],)
        return comm


def timed_steps(comm, step, sync, steps, warmup):
    """The bench contract: W untimed steps, then exactly K timed steps
    bracketed by barrier + device sync on both sides; returns the MAX over
    ranks of the elapsed seconds."""
    for _ in range(warmup):
        step()
    sync()
    comm.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    comm.barrier()
    sync()
    return comm.max_over_ranks(elapsed)


def timed_repeats(comm, step, sync, steps, warmup, repeats=5):
    """`repeats` back-to-back timed regions of exactly `steps` steps each (timed_steps; the warm-up runs before
    the first one only) -> (per-repeat MAX-over-ranks seconds, this rank's own seconds per repeat).  The bench
    reports the median repeat and the spread (median and best)."""
    spans, own = [], []
    for r in range(max(1, int(repeats))):
        for _ in range(warmup if r == 0 else 0):
            step()
        sync()
        comm.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        mine = time.perf_counter() - t0
        comm.barrier()
        sync()
        own.append(mine)
        spans.append(comm.max_over_ranks(mine))
    return spans, own
