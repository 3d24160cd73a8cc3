"""bench.py --gpus N (N > 1) without a launcher around it starts its own ranks: fresh child processes through
torch.distributed.run, before torch is imported or HIP is touched in the parent.  Runs on the CPU with a stub worker
(the ranks rendezvous over gloo on 127.0.0.1 and rank 0 prints the one JSON line)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = textwrap.dedent("""
    import json, os, sys
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1"
    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)  # the backend's connection banner goes to stderr, as bench.py does for RCCL's (_StdoutToStderr)
    dist.init_process_group("gloo")
    ones = torch.ones(1, dtype=torch.int32)
    dist.all_reduce(ones)
    dist.barrier()
    sys.stdout.flush()
    os.dup2(saved, 1)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "n_ranks_seen": int(ones.item()), "argv": sys.argv[1:]}), flush=True)
    else:
        print(f"rank {rank} alive", file=sys.stderr)
    dist.destroy_process_group()
    sys.exit(int(os.environ.get("STUB_RC", "0")) if rank == world - 1 else 0)
""")


def _run(tmp_path, n, rc_env=None):
    stub = tmp_path / "stub_worker.py"
    stub.write_text(STUB)
    env = dict(os.environ, STITCH_BENCH_WORKER=str(stub))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    if rc_env is not None:
        env["STUB_RC"] = str(rc_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                          env=env, capture_output=True, text=True, timeout=300)


@pytest.mark.timeout(400)
def test_gpus_n_starts_its_own_ranks(tmp_path):
    r = _run(tmp_path, 2)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # ONE JSON line on stdout: rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2
    assert d["argv"] == ["--gpus", "2", "--steps", "2", "--warmup", "1"]  # the ranks get the caller's own flags
    assert "launching" in r.stderr and "torch.distributed.run" in r.stderr


@pytest.mark.timeout(400)
def test_child_failure_is_the_launchers_exit_status(tmp_path):
    r = _run(tmp_path, 2, rc_env=3)  # bench.py exits 3 when an output check fails
    assert r.returncode != 0


def test_importing_bench_does_not_import_torch():
    # the launcher decision is taken before torch (and with it the HIP runtime) is loaded
    code = "import sys; sys.path.insert(0, %r); import bench; assert 'torch' not in sys.modules; print('ok')" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr
