"""bench.py --gpus N starts its N ranks itself (the reference fans out itself, Camera.java:129-182).  On a box without
GPUs: the launch path is rehearsed with --dry-run-gloo (N processes, gloo, library-shaped blocks exchanged and
unpacked), and asking for more GPUs than there are fails loudly instead of printing a 1-GPU number."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + [str(a) for a in args], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=300)


def test_gpus_n_starts_n_ranks_that_exchange_blocks():
    r = _run("--gpus", 3, "--dry-run-gloo", "--width", 150, "--height", 97, "--steps", 5)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["backend"] == "gloo"
    assert out["n_gpus"] == 3 and out["ranks"] == 3          # what the collective layer saw, not what the flag said
    assert out["gathered_frame_equals_single_gpu_frame"] is True
    assert out["frames_in_flight"] == 2 and out["value"] is None


def test_more_gpus_than_present_fails_loudly():
    import torch
    have = torch.cuda.device_count()
    r = _run("--gpus", have + 2)
    assert r.returncode != 0
    assert "refusing" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_must_agree_with_gpus():
    r = _run("--gpus", 1, env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_a_failing_rank_fails_the_run():
    # rank processes die at once (unknown model): the parent must report failure, not hang or print a line
    r = _run("--gpus", 2, "--dry-run-gloo", "--width", 0, "--height", 0)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


import pytest


@pytest.mark.gpu
def test_n_rank_path_rehearsed_on_one_gpu():
    """The whole N-rank bench path on the one GPU a test box has: 3 ranks share GPU 0 (tile shares rendered as compact blocks,
    one gather -- over gloo through host memory here, RCCL on a node --, the library's unpack kernel, frames in flight on
    scene views, and rank 0's check that the gathered frame is the single-GPU frame, bit for bit)."""
    r = _run("--gpus", 3, "--rehearse-one-gpu", "--steps", 4, "--warmup", 1, "--no-cpu-baseline", "--width", 640, "--height", 360, "--faces", 60000)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["ranks"] == 3 and "rehearsal" in out
    assert out["gathered_frame_equals_single_gpu_frame"] is True
    assert out["pipelined"]["frames_identical"] is True
    assert len(out["rank_render_ms"]["all"]) == 3
