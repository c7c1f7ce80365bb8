"""`python bench.py --gpus N` without a launcher must start its own N ranks (one process per GPU) before any GPU
call and have rank 0 print exactly one JSON line (the driver's contract).  `--dry-run` swaps RCCL for gloo and the
GPU step for an empty one, so the launch / rendezvous / barrier / gather plumbing of every mode runs here on CPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--dry-run", *extra],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # ONE line, from rank 0 only
    return json.loads(lines[0])


@pytest.mark.parametrize("mode,scaling", [("iter", "weak"), ("problems", "strong"), ("rhs", "strong")])
def test_bench_spawns_its_own_ranks(mode, scaling):
    out = _run("--gpus", "2", "--mode", mode, "--nrhs", "7", "--problems", "5")
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["dry_run"] is True
    assert out["scaling"] == scaling and out["value"] > 0 and out["higher_is_better"] is True
    if mode == "iter":
        # the driver's one command measures both sharded axes of SURVEY.md 8(e) behind the headline pass: the rows of a
        # short `problems` pass and a short `rhs` pass ride in the same JSON line (and exist at N = 1 as anchors)
        sm = out["scale_modes"]
        assert set(sm) == {"problems", "rhs"}, sm
        assert sm["problems"]["scaling"] == "strong" and sm["problems"]["config"]["problems_per_rank"] == 3
        assert sm["rhs"]["scaling"] == "strong" and sm["rhs"]["config"]["columns_per_rank"] == 4
        assert sm["problems"]["value"] > 0 and sm["rhs"]["value"] > 0
    if mode == "problems":
        assert out["config"]["problems_per_rank"] == 3       # rank 0 of 2 holds problems 0, 2, 4
    if mode == "rhs":
        assert out["config"]["columns_per_rank"] == 4        # columns 0, 2, 4, 6


def test_bench_single_rank_needs_no_launcher():
    out = _run()
    assert out["n_gpus"] == 1 and out["dry_run"] is True
    assert set(out["scale_modes"]) == {"problems", "rhs"}        # the N = 1 anchors of the two sharded axes
    assert _run("--no-scale-modes")["scale_modes"] is None


def test_bench_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stdout + r.stderr)
