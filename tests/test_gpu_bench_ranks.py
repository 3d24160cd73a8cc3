"""bench.py's own main() with MORE THAN ONE RANK, before the driver's 8-GPU node meets it: two (and three) fresh processes
share cuda:0, the collectives run over gloo on host tensors and the mosaics are staged through host memory
(STITCH_BENCH_BACKEND=gloo, a rehearsal switch the driver never sets).  Everything else is the code the RCCL run executes:
ragged shards and n_max gather blocks, launch sequences that hold the shares of several steps (different numbers of them
on different ranks), the gather ring, verify()'s checksum collective over all ranks, the MAX / MIN reductions, the
single-batch figure.  The ranks are started by bench.py's own launcher (fresh children; nothing is exec'ed from a process
that has touched the GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(n, *flags, env_extra=None):
    env = dict(os.environ, STITCH_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "STITCH_BENCH_WORKER"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--frame", "256", "--no-cpu-baseline"] + list(flags),
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line: rank 0's
    return json.loads(lines[0]), r.stderr


@pytest.mark.timeout(700)
def test_two_ranks_ragged_shards_and_coalesced_sequences():
    d, err = _bench(2, "--pairs-per-step", "5", "--steps", "6", "--warmup", "1")
    c = d["config"]
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["outputs_verified"] is True
    assert d["scaling"] == "strong" and "REHEARSAL" in c["backend"]
    # 5 pairs over 2 ranks: 3 + 2, so the gather blocks are padded to n_max = 3 and the ranks coalesce differently
    assert [r["pairs_per_step"] for r in c["per_rank"]] == [3, 2]
    assert [r["steps_per_sequence"] for r in c["per_rank"]] == [5, 6]  # min(16 // 3, 6), min(16 // 2, 6)
    assert c["output_check"]["failures"] == 0 and c["output_check"]["timed_regions_checked"] >= 5
    assert c["no_exchange_mpix_s"] and c["single_batch_ms"] > 0
    assert d["value"] > 0 and d["roofline"]["frac"] > 0


@pytest.mark.timeout(700)
def test_three_ranks_one_pair_short_and_no_coalescing():
    d, _ = _bench(3, "--pairs-per-step", "7", "--steps", "3", "--warmup", "2", "--no-coalesce")
    c = d["config"]
    assert d["n_ranks_seen"] == 3 and d["outputs_verified"] is True
    assert [r["pairs_per_step"] for r in c["per_rank"]] == [3, 2, 2]
    assert all(r["steps_per_sequence"] == 1 for r in c["per_rank"])


@pytest.mark.timeout(700)
def test_two_ranks_unsigned_char_frames_and_no_gather():
    d, _ = _bench(2, "--pairs-per-step", "4", "--steps", "2", "--warmup", "1", "--pixel", "u8")
    assert d["n_ranks_seen"] == 2 and d["outputs_verified"] is True and "u8" in d["metric"]
    d, _ = _bench(2, "--pairs-per-step", "4", "--steps", "2", "--warmup", "1", "--no-gather")
    assert d["n_ranks_seen"] == 2 and d["outputs_verified"] is True and d["config"]["exchange"] == "none (--no-gather)"
