"""`python -m planer_amd.launch`: the torch-free process spawner (one rank per GPU).  CPU only: the ranks talk through
`dist.FileCommunicator` over the rendezvous file the launcher hands them; nothing here touches a device."""
import os
import subprocess
import sys
import textwrap
import time

from tests.conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    assert "torch" not in sys.modules
    from planer_amd import dist
    rank, world, local = dist.env_world()
    assert (rank, world, local) == (int(os.environ["RANK"]), 2, rank)
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    got = dist.exchange_bytes(b"id-from-rank-0" if rank == 0 else None, rank)      # how the RCCL unique id travels
    assert got == b"id-from-rank-0"
    comm = dist.FileCommunicator(rank, world, timeout=60)
    assert comm.max_over_ranks(10 + rank) == 11 and comm.min_over_ranks(10 + rank) == 10
    lo, hi = dist.shard_range(65, world, rank)
    comm.barrier()
    assert "torch" not in sys.modules
    print("rank %%d of %%d rows %%d:%%d %%s" %% (rank, world, lo, hi, " ".join(sys.argv[1:])), flush=True)
""")

FAILING = textwrap.dedent("""
    import os, sys, time
    if os.environ["RANK"] == "1":
        sys.exit(7)
    time.sleep(60)          # rank 0 would hang on its peer: the launcher has to stop it
""")


def run_launcher(args, timeout=120):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "PLANER_RDZV_FILE"):
        env.pop(k, None)
    return subprocess.run([sys.executable, "-m", "planer_amd.launch"] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_two_ranks_over_the_file_transport_without_torch(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    r = run_launcher(["--nproc", "2", str(script), "--flag", "value"])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = sorted(ln for ln in r.stdout.splitlines() if ln.startswith("rank "))
    assert lines == ["rank 0 of 2 rows 0:33 --flag value", "rank 1 of 2 rows 33:65 --flag value"], r.stdout


def test_first_failure_stops_the_other_ranks_and_is_the_exit_code(tmp_path):
    script = tmp_path / "failing.py"
    script.write_text(FAILING)
    t0 = time.time()
    r = run_launcher(["--nproc", "2", "--grace", "2", str(script)])
    assert r.returncode == 7, (r.returncode, r.stderr)
    assert time.time() - t0 < 30 and "rank 1 exited with 7" in r.stderr


def test_module_form_and_explicit_port(tmp_path):
    r = run_launcher(["--nproc", "1", "--master-port", "29611", "-m", "planer_amd.launch", "--nproc", "1", "-m", "json.tool", "--help"])
    assert r.returncode == 0 and "usage" in r.stdout.lower(), r.stdout + r.stderr
