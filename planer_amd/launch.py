"""One process per GPU without torch: `python -m planer_amd.launch --nproc N script.py [args ...]`.

The reference has no launcher (a single-process library: net.py has no notion of a rank); north_star asks for a batch-sharded
run on the GPUs of one node with no PyTorch dependency, so this is the spawner the product ships.  It does what
`planer_amd.dist` needs from one and nothing else: N children of this interpreter, each with

    RANK / LOCAL_RANK = 0 .. N-1, WORLD_SIZE = N, MASTER_ADDR = 127.0.0.1, MASTER_PORT = <free port or --master-port>,
    PLANER_RDZV_FILE = a fresh file all ranks agree on (the 128-byte RCCL unique id travels through it: dist.exchange_bytes),
    HSA_ENABLE_IPC_MODE_LEGACY = 0 (RCCL across processes needs dmabuf IPC on this driver stack),

the children's stdout / stderr passed through, the first failure's exit code returned after the other ranks have been stopped
(SIGTERM to the exact pids this launcher started, SIGKILL after --grace seconds), SIGINT / SIGTERM forwarded.
`python -m torch.distributed.run --nproc-per-node N script.py ...` keeps working (it sets the same variables)."""
import argparse
import os
import signal
import socket
import subprocess
import sys
import tempfile
import time


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def child_env(rank, world, port, rdzv, base=None):
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "PLANER_RDZV_FILE": rdzv})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def launch(nproc, argv, master_port=0, module=False, grace=10.0, poll=0.05):
    """Run `argv` (a script path + its arguments, or a module name with module=True) as `nproc` ranks; returns the exit code."""
    if nproc < 1:
        raise ValueError("--nproc must be >= 1")
    port = master_port or free_port()
    tmp = tempfile.mkdtemp(prefix="planer_amd_launch_")
    rdzv = os.path.join(tmp, "rdzv")
    cmd = [sys.executable, "-u"] + (["-m"] if module else []) + list(argv)
    procs = []
    try:
        for r in range(nproc):
            procs.append(subprocess.Popen(cmd, env=child_env(r, nproc, port, rdzv)))
    except Exception:
        _stop(procs, grace)
        raise

    def forward(signum, _frame):
        for p in procs:
            if p.poll() is None:
                try:
                    p.send_signal(signum)
                except OSError:
                    pass
    old = {s: signal.signal(s, forward) for s in (signal.SIGINT, signal.SIGTERM)}
    code = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0 and code == 0:
                    code = rc if rc > 0 else 128 - rc          # killed by a signal: the shell's convention
                    print("[planer_amd.launch] rank %d exited with %d: stopping the other ranks" % (procs.index(p), rc),
                          file=sys.stderr, flush=True)
                    _stop(live, grace)
                    live = []
                    break
            time.sleep(poll)
    finally:
        for s, h in old.items():
            signal.signal(s, h)
        _stop([p for p in procs if p.poll() is None], grace)
        for f in (rdzv,):
            try:
                os.unlink(f)
            except OSError:
                pass
        try:
            os.rmdir(tmp)
        except OSError:
            pass
    return code


def _stop(procs, grace):
    for p in procs:
        if p.poll() is None:
            try:
                p.terminate()
            except OSError:
                pass
    deadline = time.time() + grace
    for p in procs:
        try:
            p.wait(max(0.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            try:
                p.kill()
            except OSError:
                pass
            p.wait()


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m planer_amd.launch", description=__doc__.split("\n\n")[0])
    ap.add_argument("--nproc", "--nproc-per-node", dest="nproc", type=int, required=True, help="ranks = GPUs of this node to use")
    ap.add_argument("--master-port", type=int, default=0, help="MASTER_PORT for the ranks (default: a free port)")
    ap.add_argument("-m", dest="module", action="store_true", help="run a module (python -m) instead of a script path")
    ap.add_argument("--grace", type=float, default=10.0, help="seconds between SIGTERM and SIGKILL when stopping ranks")
    ap.add_argument("script", help="script path (or module name with -m)")
    ap.add_argument("args", nargs=argparse.REMAINDER)
    a = ap.parse_args(argv)
    return launch(a.nproc, [a.script] + a.args, a.master_port, a.module, a.grace)


if __name__ == "__main__":
    sys.exit(main())
