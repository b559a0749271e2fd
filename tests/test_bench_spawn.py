"""`python bench.py --gpus N` with no WORLD_SIZE must start N ranks itself (VERDICT r2 missing #3).  Runs the real bench.py rank logic —
self-spawn through torch.distributed.run, process group, per-rank shards, barrier + max-over-ranks timing, one JSON line from rank 0 — on the
test harness tests/bench_emu_harness.py (which calls bench.main with CPU tensors, `gloo`, the host emulator build of the kernel sources, a two-block toy stack)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv, timeout=900):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(HIPEMU_THREADS="2", OMP_NUM_THREADS="1")
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_emu_harness.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


@pytest.mark.timeout(1200)
def test_gpus_2_spawns_two_ranks_and_reports_n_gpus_2():
    from tests import emu
    emu.build()   # (not inside the two ranks at once)
    r = _run({}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "1")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout          # rank 0 only
    j = lines[0]
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 2 and j["config"]["parallelism"] == "dp2"
    assert j["scaling"] == "weak" and j["steps"] == 2 and j["value"] > 0
    assert "EMULATOR TEST RUN" in j["data"]
    assert "spawning 2 ranks" in r.stderr
    # both all-reduce schedules exist; with two stages the toy stack can take the overlapped one
    assert isinstance(j["config"]["allreduce_overlap"], bool)


@pytest.mark.timeout(600)
def test_world_size_mismatch_is_an_error_not_a_silent_single_rank():
    r = _run({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1")
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in (r.stderr + r.stdout)
    assert not _json_lines(r.stdout)


def test_more_ranks_than_gpus_is_refused_before_spawning():
    # without the test hook: this container has no GPU, so --gpus 2 must refuse instead of measuring one rank and calling it two
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)
