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


def _clean_env(**extra):
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MISLAM_BENCH_STAGE_DIR", "MISLAM_BENCH_PARENT"):
        env.pop(v, None)
    return env


def _error_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, stdout[-4000:]
    return json.loads(lines[0])


def test_bench_wall_budget_turns_a_hung_rank_into_a_diagnostic_line():
    """one rank never arrives at the first collective (what an RCCL initialisation hang looks like from outside).  The parent
    must not read the child's stdout past the budget: ONE JSON line with "error", every rank's last stage, `ranks_seen`, a non-zero exit -- and
    no process of the tree left behind."""
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=_clean_env(MISLAM_BENCH_DRYRUN="1", MISLAM_BENCH_DRYRUN_HANG_RANK="1", MISLAM_BENCH_WALL_BUDGET_S="40"),
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    took = time.time() - t0
    assert r.returncode not in (0, None), r.stdout[-2000:] + r.stderr[-4000:]
    out = _error_line(r.stdout)
    assert "error" in out and out["value"] is None and out["n_gpus"] == 2 and out["wall_budget_s"] == 40
    assert out["ranks_seen"] == 2, out
    # the hung rank stopped where it entered the communicator's creation; the live one is waiting inside the first collective
    assert out["stages"]["1"]["stage"] == "comm_init" and out["stages"]["0"]["stage"] == "first_allreduce", out
    assert took < 40 + 60, took
    # nothing of the tree survives (the sleeping rank would, for an hour, if only the launcher had been ended)
    import psutil

    def survivors():
        out = []
        for pr in psutil.process_iter(["pid", "cmdline"]):
            cl = pr.info["cmdline"] or []
            if any(c.endswith("bench.py") for c in cl) and cl[-4:] == ["--steps", "3", "--warmup", "1"]:
                out.append((pr.info["pid"], pr.status()))
        return out
    deadline = time.time() + 15.0
    while survivors() and time.time() < deadline:      # (a rank that was sent SIGKILL a moment ago may still be on its way out)
        time.sleep(0.5)
    assert not survivors(), survivors()


def test_bench_rank_watchdog_under_a_foreign_launcher():
    """The driver starts N > 1 with its OWN torchrun line: no parent of ours above the ranks.  Every rank's watchdog thread ends its process at the
    budget and rank 0's prints the same diagnostic line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node=2",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=_clean_env(MISLAM_BENCH_DRYRUN="1", MISLAM_BENCH_DRYRUN_HANG_RANK="1", MISLAM_BENCH_WALL_BUDGET_S="25"),
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    out = _error_line(r.stdout)
    assert "rank watchdog" in out["error"] and out["ranks_seen"] == 2
    assert out["stages"]["1"]["stage"] == "comm_init" and out["stages"]["0"]["stage"] == "first_allreduce", out
