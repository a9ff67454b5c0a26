"""The N>1 path on CPU: two processes, torch.distributed/gloo as the transport
behind planer_amd.dist's Communicator interface.  Covers batch sharding, the
rank-0 weight broadcast protocol, the file rendezvous used for the RCCL id,
and the bench timing contract (barrier both sides, MAX over ranks)."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from planer_amd import dist
from tests.conftest import ROOT


def test_shard_range_covers_batch_without_overlap():
    for total in (0, 1, 7, 32, 255, 256):
        for world in (1, 2, 3, 8):
            spans = [dist.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


WORKER = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, %(root)r)
    import numpy as np
    import torch, torch.distributed as td
    from planer_amd import dist
    from oracle import planer_np as onp
    from planer_amd.irgen import customnet

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    td.init_process_group("gloo", rank=rank, world_size=world)

    class GlooComm(dist.Communicator):
        # same interface as RcclCommunicator, host buffers instead of HBM
        def __init__(self):
            self.rank, self.world = rank, world
        def bcast_device(self, arr, root=0):
            t = torch.from_numpy(arr)
            td.broadcast(t, root)
            return arr
        def barrier(self):
            td.barrier()
        def max_over_ranks(self, value):
            t = torch.tensor([float(value)], dtype=torch.float64)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            return float(t[0])

    class HostNet:
        # stands in for planer_amd.Net: one contiguous weight blob + a mirror refresh hook
        def __init__(self, g):
            self.net = onp.OracleNet()
            self.net.load_json(g["input"], g["inits"], g["layers"], g["flow"])
            self.blob = np.zeros(sum(w.nbytes for w in self.net.weights), np.uint8)
            self.refreshed = False
        def load_weights(self, data):
            self.blob[:] = data
        def weight_blob(self):
            return self.blob
        def refresh_host_mirror(self):
            self.refreshed = True

    comm = GlooComm()
    # 1) file rendezvous (what carries the RCCL unique id)
    token = os.urandom(128) if rank == 0 else None
    got = dist.exchange_bytes(token, rank, os.environ["PLANER_RDZV_FILE"], timeout=60)
    assert len(got) == 128
    # 2) rank-0 weight broadcast: only rank 0 holds the blob
    g, b = customnet.build()
    hn = HostNet(g)
    comm.load_weights(hn, b if rank == 0 else None)
    assert np.array_equal(hn.blob, b), "rank %%d did not receive the weights" %% rank
    assert comm.bcast_ms is not None and comm.bcast_ms >= 0.0      # the broadcast alone, max over ranks
    assert comm.min_over_ranks(rank + 1.0) == 1.0 and comm.max_over_ranks(rank + 1.0) == float(world)
    assert hn.refreshed == (rank != 0)
    hn.net.load_weights(hn.blob)
    # 3) batch sharding: each rank runs its slice, results equal the full-batch run
    x = customnet.make_input(5)
    lo, hi = dist.shard_range(5, world, rank)
    y = hn.net(x[lo:hi].copy())
    full = onp.OracleNet(); full.load_json(g["input"], g["inits"], g["layers"], g["flow"]); full.load_weights(b)
    ref = full(x.copy())
    assert np.allclose(y, ref[lo:hi], rtol=0, atol=1e-6)
    # 4) timing contract: MAX over ranks, everyone gets the same number
    calls = []
    def step():
        calls.append(1); time.sleep(0.01 * (rank + 1))
    el = dist.timed_steps(comm, step, lambda: None, steps=3, warmup=2)
    assert len(calls) == 5
    assert el >= 0.03 * world * 0.9, el          # slowest rank (rank world-1) sets the time
    t = torch.tensor([el], dtype=torch.float64); td.all_reduce(t, op=td.ReduceOp.MAX)
    assert abs(float(t[0]) - el) < 1e-12
    # 5) repeated timed regions (what bench.py reports the median of): warm-up once, K steps per repeat
    calls.clear()
    spans, own = dist.timed_repeats(comm, step, lambda: None, steps=2, warmup=1, repeats=3)
    assert len(calls) == 1 + 3 * 2 and len(spans) == len(own) == 3
    assert all(s >= o - 1e-12 for s, o in zip(spans, own)) and min(spans) >= 0.02 * world * 0.9
    print("rank", rank, "ok", got[:4].hex())
    td.destroy_process_group()
""")


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2",
               PLANER_RDZV_FILE=str(tmp_path / "rdzv"), OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
    tokens = {o.strip().split()[-1] for o in outs}
    assert len(tokens) == 1                      # both ranks saw rank 0's token


def test_stale_rendezvous_file_is_ignored(tmp_path):
    import time
    path = str(tmp_path / "old")
    with open(path, "wb") as f:
        f.write(b"x" * 128)
    old = time.time() - 3600
    os.utime(path, (old, old))
    try:
        dist.exchange_bytes(None, 1, path, timeout=0.3)
        raise AssertionError("stale file accepted")
    except TimeoutError:
        pass
    assert dist.exchange_bytes(b"y" * 128, 0, path) == b"y" * 128
    assert dist.exchange_bytes(None, 1, path, timeout=5) == b"y" * 128


FILE_WORKER = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, %(root)r)
    from planer_amd import dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    comm = dist.FileCommunicator(rank, world, base=os.environ["PLANER_RDZV_FILE"], timeout=60)
    assert not comm.device_transport
    # many back-to-back collectives: values are per-collective, never mixed up; old files are removed
    for i in range(40):
        assert comm.max_over_ranks(rank * 10 + i) == (world - 1) * 10 + i
        comm.barrier()
    calls = []
    def step():
        calls.append(1); time.sleep(0.01 * (rank + 1))
    el = dist.timed_steps(comm, step, lambda: None, steps=3, warmup=1)
    assert len(calls) == 4 and el >= 0.03 * world * 0.9
    class Net:
        loaded = None
        def load_weights(self, blob): self.loaded = blob
    n = Net(); comm.load_weights(n, b"weights"); assert n.loaded == b"weights"
    try:
        comm.load_weights(Net(), None); raise SystemExit("blob-less rank accepted")
    except ValueError:
        pass
    print("rank", rank, "ok", "%%.6f" %% el)
""")


def test_file_communicator_two_ranks(tmp_path):
    """The same-node fallback used when RCCL cannot be initialised: barrier + MAX over ranks."""
    script = tmp_path / "fworker.py"
    script.write_text(FILE_WORKER % {"root": ROOT})
    env = dict(os.environ, WORLD_SIZE="3", PLANER_RDZV_FILE=str(tmp_path / "rdzv"))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(3)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o)
    assert len({o.strip().split()[-1] for o in outs}) == 1          # everyone got the same MAX
    left = [f for f in os.listdir(tmp_path) if ".fc." in f]
    assert len(left) <= 3 * 2 + 3                                    # only the last two collectives remain
