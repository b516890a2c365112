"""`python bench.py --gpus N` must start its own ranks: an unattended N-GPU lease runs that very line.

CPU part (here): the launcher flow alone -- `MISLAM_BENCH_DRYRUN=1` makes every rank stop after the gloo bootstrap and one collective
(no context is created: the product has no CPU path, and the line says `dry_run`).  GPU part (`-m gpu`): the whole bench with two real
ranks sharing the box's one GPU over the gloo exchange context (RCCL refuses two ranks on one device)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, extra_env, timeout):
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):      # no launcher environment: that is the point
        env.pop(v, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-4000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 3])
def test_bench_starts_its_own_ranks(world):
    out = run_bench(["--gpus", str(world), "--steps", "3", "--warmup", "1"], {"MISLAM_BENCH_DRYRUN": "1"}, 600)
    assert out["n_gpus"] == world and out["rccl"]["ranks_seen"] == world and out["steps"] == 3
    assert "dry_run" in out and out["value"] is None


def test_bench_launcher_passes_a_failing_rank_on():
    env = dict(os.environ, MISLAM_BENCH_DRYRUN="1", OMP_NUM_THREADS="1")
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(v, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--nn", "no-such-search"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "{" not in r.stdout


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_without_a_launcher(capi):
    out = run_bench(["--gpus", "2", "--steps", "4", "--warmup", "2", "--points", "60000", "--no-sizes", "--brute-ref-steps", "1"],
                    {"MISLAM_BENCH_TRANSPORT": "gloo", "MISLAM_BENCH_DEVICE": "0"}, 900)
    assert out["n_gpus"] == 2 and out["rccl"]["ranks_seen"] == 2 and out["rccl"]["nranks"] == 2
    assert out["value"] > 0 and out["steps"] == 4 and "rehearsal" in out
    assert out["target_sharded"]["allreduce_u64_min"]["launches"] >= 2
