#!/usr/bin/env python3
"""ICP iterations/s on a synthetic uniform cloud (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--points P]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step is ONE ICP iteration of the hot path on clouds already resident in HBM: three launches on the default path -- the fused
search kernel (transform, previous iteration's error sums, exact nearest-neighbour search through the cell grid with the
box-hierarchy fallback, moments of the new pairs), the rows reduction, and the solve kernel (previous iteration's stop rule,
3x3 SVD Kabsch, compose).  The stop rule runs on the device every step but never fires (eps = 0).  With N > 1 the problem is
fixed ("scaling": "strong"): the moving cloud is sharded over the ranks for the indexed searches (one 64 x 18-double RCCL sum
all-reduce per iteration), the fixed cloud for the every-pair search (plus a ncclMin all-reduce of the packed keys), all inside
libmislam.so over xGMI; torch.distributed (gloo) is only the bootstrap / barrier / max.

Rank 0 prints one JSON line.
  roofline        the search kernel of the timed steps: algorithmic bytes (20*N + 12*M_local) over its HIP-event-timed
                  average launch; `issue` beside it is the vector pipe's share of that kernel, from the committed counter profile of
                  this command, CALIBRATED by the same counters read off a kernel of known instruction count (tools/valu_probe)
  whole_call      one registration as the reference times it (host buffers in, allocation / upload / index builds / 50 iterations /
                  result included: testrunner.cpp:54-56) at 1e5 and 1e6 points and on the bunny clouds (cfg 1), with the load split into its stages
  late_iterations the same number of steps LATE in the same registration (the headline's window is a transient: the search gets cheaper as the clouds close in)
  sizes           the same measurement at N = M = 1e4, 1e5, 1e7 (BASELINE.json: "N = 10^4 ... 10^7"), a few steps each
  bruteforce_nn   the every-pair kernel on the same clouds with `valu`, its launch against the fp32 vector-issue rate
  target_sharded  N > 1 only: cfg 3's split -- fixed cloud sharded, every-pair search, ncclAllReduce(u64, min) of the keys
  rccl            N > 1 only: what the communicator itself reports (ranks, bit mask of ranks seen in one all-reduce)
  cpd_bunny       the CPD leg of the metric (cfg 4), with its own roofline object
  cpu_baseline    the REFERENCE's own GetCorrespondingPoints (oracle/_ref, all host threads) on a bounded sample of source rows,
                  scaled to a full iteration; sizes['10000'] / ['100000'] carry their own, timed DIRECTLY (one whole search each);
                  cpd_bunny carries one too (one ComputePMatrix + one MStep of cpu-slam on the bunny clouds, single thread as the reference runs it)
  cpd_bunny.published_size   the size the reference publishes CPD times for (N = 49 000, doc/plots/ms-cpd-3.png): exact and hybrid, ms per EM iteration
  runtime         which libamdhip64 / librccl objects libmislam.so's calls bind to in this process, their versions, every copy mapped; a rank that
                  finds two different RCCL (HIP) runtimes of different versions mapped refuses to start (exit 3, before any device call)
  allreduce_f64_sum_floor    N = 1: the headline path's one collective (64 x 18 doubles) through a ONE-rank RCCL communicator, event-timed
  expected_scaling           N > 1: what the one-GPU emulation of a rank's share predicts, so that the first multi-GPU numbers are read against it
  (wall budget)   N > 1 only: MISLAM_BENCH_WALL_BUDGET_S (default 420).  Every rank leaves a one-line stage file (start / gloo_init / comm_init /
                  first_allreduce / load / warmup / timed / ...) and carries a watchdog thread; `--gpus N` without a launcher also watches its child
                  tree from outside.  On expiry: ONE line with "error", "stages" (every rank's last stage) and "ranks_seen", exit code 124
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                      # MI355X HBM3E spec (MI355X_MICROARCH.md)
VALU_LANE_OPS_PEAK = 256 * 4 * 32 * 2.4e9  # 256 CU x 4 SIMD-32 x 2.4 GHz: fp32 lane-instructions/s (= 157.3 TFLOP/s / 2)
VALU_WAVE_INSTR_PEAK = 256 * 4 * 2.4e9 / 2  # a wave64 instruction occupies a SIMD-32 for two cycles (MI355X_MICROARCH.md constants)
VALU_BUSY_CEILING = 1024 / 4 / 8           # what SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE can read at most: 1 024 SIMDs, one wave instruction per four
                                           # cycles, GUI cycles summed over the 8 XCDs (tools/valu_probe reads 29.2 of it, the every-pair kernel 31.0)
OPS_PER_PAIR = {0: 8.5, 1: 6.5}            # 3 sub + 3 mul + 2 add (or 1 mul + 2 fma) + 1/2 min3, per candidate pair
SWEEP_SIZES = (10000, 100000, 10000000)


def host_cpu_budget():
    """CPUs this process may actually use: the container's CFS quota (cgroup v2 cpu.max / v1 cfs_quota_us) when there is one, else
    the cores it sees.  The GPU boxes show 256 cores and grant 16."""
    cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1, min(cores, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    try:
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0:
            return max(1, min(cores, int(round(quota / period))))
    except (OSError, ValueError):
        pass
    return cores


def quiet_host_pools():
    """Before numpy is imported: one BLAS / OpenMP thread.  numpy's BLAS pool starts a spinning thread per VISIBLE core (256 on the
    GPU boxes) for the 3x3 product in synth_cloud; under the container's 16-CPU quota that exhausts the quota of the 100 ms period
    and the kernel freezes every thread of the process until the next one -- measured as 20-60 ms 'stalls' of whichever call came
    next (hipStreamSynchronize, a copy, a launch; `MISLAM_DEV_STALL_MS` shows them as involuntary context switches)."""
    for var in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(var, "1")


def synth_cloud(np, n, seed=666):
    """SURVEY 8d / BASELINE.md section 3: uniform [-5,5]^3 (spread 10), after = R(0.2 rad about (1,2,3)/sqrt14) * before
    + 10*(1,1,1)/sqrt3, target independently permuted."""
    rng = np.random.default_rng(seed)
    before = rng.uniform(-5.0, 5.0, size=(n, 3)).astype(np.float32)
    axis = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    Rm = np.eye(3) + np.sin(0.2) * K + (1 - np.cos(0.2)) * (K @ K)
    t = 10.0 * np.ones(3) / np.sqrt(3.0)
    after = (before[rng.permutation(n)].astype(np.float64) @ Rm.T + t).astype(np.float32)
    return before, after


def search_source_hash():
    """Hash of the search kernel's sources: a committed counter profile speaks for THIS code only if it carries the same one.
    Of the CODE: `//` comments and white space do not count (a reworded comment is not another kernel)."""
    import re
    h = hashlib.sha256()
    for f in ("nn_grid.hip", "nn_grid.h", "nn_walk.hpp", "icp_rows.hpp", "nn_tree.h"):
        text = open(os.path.join(ROOT, "cuda-slam_amd", "csrc", f), "r").read()
        text = re.sub(r"//[^\n]*", "", text)
        h.update("".join(text.split()).encode())
    return h.hexdigest()[:16]


def committed_profile(workload, kernel, steps, warmup):
    """The committed rocprofv3 --pmc summary (profiles/*_counters.json) for this workload and kernel, and whether it was taken
    with THIS run's --steps / --warmup (the search's cost depends on which iterations are timed) on THIS code (source hash of
    the search kernel).  bench.py cannot run counter passes itself: they need their own rocprofv3 runs (tools/gpu_profiles.sh)."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload") == workload and d.get("kernel") == kernel:
            same = d.get("steps") == steps and d.get("warmup") == warmup
            if "source_hash" in d and kernel == "nn_grid_kernel":
                same = same and d["source_hash"] == search_source_hash()
            return d, os.path.relpath(path, ROOT), same
    return None, None, False


def valu_calibration():
    """profiles/*_valu_calibration.json: the SQ counters of tools/valu_probe (v_fma_f32 only, 8 waves per SIMD) -- the practical vector
    issue rate of this chip and what the busy counters read when the pipe is saturated."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_valu_calibration.json")), reverse=True):
        try:
            d = json.load(open(path))
            k = d["kernels"]["valu_probe<0>"]
            if k.get("wave_instructions_per_s") and k.get("valu_busy_quadcycles_per_gui_cycle"):
                return d, os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def cpu_baseline(np, before, after, target_seconds=15.0):
    """The reference's cpu-slam correspondence search (>99 % of its iteration) on a bounded row sample: a
    short probe measures this host's pair rate, then the sample is sized for ~15 s of CPU work."""
    from oracle import refbind, oraclebind
    n, m = len(before), len(after)
    cores = os.cpu_count() or 1            # what the reference starts: std::thread::hardware_concurrency() threads (common.cpp:443)
    if refbind.available():
        kind = "reference"
        search = lambda rows: refbind.corresponding_points(before[:rows], after, 1000.0, True)
    else:
        kind = "port"
        search = lambda rows: oraclebind.nn_search(before[:rows], after, threads=0)
    probe_rows = int(max(cores, min(n, 2.0e9 // m)))
    t0 = time.perf_counter()
    search(probe_rows)
    rate = probe_rows * float(m) / (time.perf_counter() - t0)
    rows = int(max(probe_rows, min(n, rate * target_seconds // m)))
    t0 = time.perf_counter()
    search(rows)
    dt = time.perf_counter() - t0
    full_iter_s = dt * (n / rows)
    return {"value": 1.0 / full_iter_s, "unit": "iterations/s", "cores": cores, "host_cpu_quota": host_cpu_budget(), "kind": kind,
            "sample": "GetCorrespondingPoints (common.cpp:441-507, %d threads) on %d of %d source rows x all %d targets: "
                      "%.2f s, scaled x%.1f to one iteration (the search is >99%% of a cpu-slam iteration)"
                      % (cores, rows, n, m, dt, n / rows),
            "pairs_per_s": rows * m / dt}


def cpu_baseline_direct(np, before, after):
    """"time N = 10^4 and 10^5 directly" -- ONE whole correspondence search of the reference's cpu-slam (all source rows, all
    targets: 1e8 / 1e10 pairs, ~0.01 / ~1 s on this box's pair rate), twice, the better one."""
    from oracle import refbind, oraclebind
    cores = os.cpu_count() or 1
    if refbind.available():
        kind, search = "reference", lambda: refbind.corresponding_points(before, after, 1000.0, True)
    else:
        kind, search = "port", lambda: oraclebind.nn_search(before, after, threads=0)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        search()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    n, m = len(before), len(after)
    return {"value": 1.0 / best, "unit": "iterations/s", "cores": cores, "host_cpu_quota": host_cpu_budget(), "kind": kind,
            "sample": "GetCorrespondingPoints (common.cpp:441-507, %d threads) on ALL %d source rows x %d targets, timed directly: %.4f s "
                      "(the search is >99%% of a cpu-slam iteration)" % (cores, n, m, best),
            "pairs_per_s": n * float(m) / best}


def cpu_baseline_cpd(np, before, after, sigma2, weight=0.3):
    """One EM iteration of the reference's cpu-slam CPD on the same clouds: ComputePMatrix (coherentpointdrift.cpp:168-221, one thread --
    the reference's E-step is sequential) + MStep (:223-277)."""
    from oracle import refbind, oraclebind
    m, n = len(before), len(after)
    c = oraclebind.cpd_constant(sigma2, weight, m, n)
    kind, mod = ("reference", refbind) if refbind.available() else ("port", oraclebind)
    t0 = time.perf_counter()
    p1, pt1, px, L = mod.cpd_estep(before, after, c, sigma2)
    t1 = time.perf_counter()
    mod.cpd_mstep(before, after, p1, pt1, px, False)
    t2 = time.perf_counter()
    return {"value": 1.0 / (t2 - t0), "unit": "EM iterations/s", "cores": 1, "host_cpu_quota": host_cpu_budget(), "kind": kind,
            "sample": "one ComputePMatrix (%d x %d affinities, %.2f s) + one MStep (%.4f s) of cpu-slam on the same clouds, single thread as the reference runs them"
                      % (m, n, t1 - t0, t2 - t1),
            "ms_per_em_iteration": (t2 - t0) * 1e3}


def estep_roofline(workload, pairs):
    """Roofline of the exact CPD E-step (K7a + K7b: the affinity is evaluated twice, P is never stored) for ONE workload: vector-pipe busy time of its
    two kernels from the committed counter profile's entry for THAT workload (profiles/*_cpd_estep_counters.json `workloads`: only the dispatches whose
    grid is the workload's -- VERDICT r05 item 1), over what the same counters read on a saturated pipe (tools/valu_probe)."""
    prof_c, src, _ = committed_profile("cpd_bunny_14904", "cpd_estep", None, None)
    cal, cal_src = valu_calibration()
    if prof_c is None or cal is None:
        return None
    entry = prof_c.get("workloads", {}).get(workload)
    if entry is None and workload == prof_c.get("workload") and "workloads" not in prof_c:
        entry = prof_c                                   # (a profile of round 5 or earlier: one entry, mixed launches)
    if entry is None or "kernels" not in entry:
        return None
    sat = cal["kernels"]["valu_probe<0>"]["valu_busy_quadcycles_per_gui_cycle"]
    ceiling = cal.get("valu_busy_ceiling", VALU_BUSY_CEILING)
    # the kernels of the EXACT mode only: the `<.., true>` instantiations are the hybrid mode's truncated E-step
    ks = {k: v for k, v in entry["kernels"].items()
          if v.get("valu_busy_quadcycles_per_gui_cycle") and v.get("launch_ms") and not k.rstrip().endswith("true>") and "cpd_trunc_" not in k}
    if not ks:
        return None
    t_all = sum(v["launch_ms"] for v in ks.values())
    busy = sum(v["valu_busy_quadcycles_per_gui_cycle"] * v["launch_ms"] for v in ks.values()) / t_all
    every_pair = cal["kernels"].get("nn_bruteforce_kernel", {}).get("valu_busy_quadcycles_per_gui_cycle")
    roof = {"bound": "fp32-valu-issue", "achieved": busy, "peak": ceiling, "unit": "vector-pipe busy quad-cycles per GPU cycle",
            "frac": busy / ceiling, "workload": workload, "pairs_per_launch": pairs,
            "peak_is": "the counter's ceiling: 1 024 SIMDs / 4 cycles per wave instruction / 8 XCDs summed in GRBM_GUI_ACTIVE",
            "probe_reads": sat, "every_pair_kernel_reads": every_pair, "frac_of_probe": busy / sat,
            "calibration_source": cal_src, "source": src,
            "kernels": {k: {"launch_ms": v["launch_ms"], "launches_in_profile": v.get("launches"), "frac": v["valu_busy_quadcycles_per_gui_cycle"] / ceiling,
                            "frac_of_probe": v["valu_busy_quadcycles_per_gui_cycle"] / sat,
                            "valu_instructions_per_pair": v["SQ_INSTS_VALU"] * 64.0 / pairs if v.get("SQ_INSTS_VALU") else None}
                        for k, v in ks.items()},
            "note": "time-weighted over the two E-step kernels of the exact mode, dispatches of this workload only; packed instructions count by the time they hold the pipe, "
                    "not as one; the contraction's 4 FMAs per pair run on the matrix pipe (MFMA 4x4x1) when that form is selected"}
    mf = [v["mfma"] for k, v in ks.items() if "mfma" in k and v.get("mfma")]
    if mf:      # north_star: "MFMA utilisation against the chip's peak" -- the contraction kernel's matrix pipe, from its own counter pass
        m0 = mf[0]
        roof["mfma_util"] = m0["mfma_util"]
        roof["mfma"] = {"busy_cycles_per_launch": m0["SQ_VALU_MFMA_BUSY_CYCLES"], "gpu_cycles_per_launch": m0["gpu_cycles"],
                        "flops_on_matrix_pipe_per_launch": m0["mfma_flops_per_launch"],
                        "contraction_flops_per_launch": m0["contraction_flops_per_launch"],
                        "share_of_contraction_on_matrix_pipe": m0["mfma_flops_per_launch"] / m0["contraction_flops_per_launch"],
                        "source": src,
                        "note": "fp32 MFMA peak = fp32 vector peak on this chip; a 4-FMA-per-pair contraction beside ~9 vector "
                                "instructions per pair of affinity arithmetic cannot fill the matrix pipe -- what it buys is those FMAs off the binding pipe"}
    return roof


def cpd_published_size(np, capi, ctx, n=49000, iterations=8):
    """The size the reference publishes CPD times for (N = 49 000: doc/plots/ms-cpd-3.png, BASELINE.md section 1 -- exact P ~15 500 ms per
    iteration on its GPU build, hybrid ~3 000 ms), on the synthetic recipe of the headline: `iterations` EM iterations each (no stop rule),
    exact sigma^2_0, time per iteration as a caller sees it."""
    before, after = synth_cloud(np, n)
    out = {"workload": "cpd_synthetic_uniform_n%d" % n, "points": n,
           "published_reference_ms_per_iteration": {"exact_gpu_rtx2060s": 15500, "exact_cpu": 31000, "hybrid_gpu_rtx2060s": 3000, "hybrid_cpu": 2100,
                                                    "source": "doc/plots/ms-cpd-3.png (read off a log plot, +-10 %), BASELINE.md section 1; other hardware, allocation included"}}
    for label, approx in (("exact", capi.CPD_APPROX_NONE), ("hybrid", capi.CPD_APPROX_HYBRID)):
        p = capi.cpd_params(max_iterations=iterations, eps=0.0, tolerance=0.0, approximation=approx)
        ctx.cpd_register(before, after, p)
        walls = []
        for _ in range(2):
            t0 = time.perf_counter()
            sR, t, scale, it, err = ctx.cpd_register(before, after, p)
            walls.append((time.perf_counter() - t0) * 1e3)
        out[label] = {"iterations": it, "ms_total": min(walls), "ms_per_em_iteration": min(walls) / max(it, 1), "final_sigma2": err}
    # the kernels' own times at this size (events around every kernel, one more exact run) and the workload's own roofline entry
    ctx.profile_enable(True)
    ctx.profile_select(None)
    ctx.profile_reset()
    ctx.cpd_register(before, after, capi.cpd_params(max_iterations=iterations, eps=0.0, tolerance=0.0, approximation=capi.CPD_APPROX_NONE))
    prof = {capi.KERNEL_NAMES[k]: ctx.profile_get(k) for k in range(len(capi.KERNEL_NAMES))}
    ctx.profile_enable(False)
    out["exact"]["kernels_ms_per_launch"] = {k: v[0] / v[1] for k, v in prof.items() if v[1] > 0}
    roof = estep_roofline(out["workload"], float(n) * float(n))
    if roof is not None:
        out["exact"]["roofline"] = roof
    return out


def whole_call(np, capi, ctx, n, repeats=3, clouds=None, params=None):
    """One registration as the reference times it (testrunner.cpp:54-56, doc/documentation.tex:397): mi_icp_register on HOST buffers,
    upload / index builds / 50 iterations / result included, GPU-reference driver rules and sweep settings (testset.cpp:82-117).
    `ms` = best of `repeats` calls on a context that has seen the size (the first call of a size also pays its device allocations:
    `first_call_ms`); `breakdown` = one more call split into load stages (stream drained after each, so they add up) + iterations."""
    before, after = clouds if clouds is not None else synth_cloud(np, n)
    p = params if params is not None else capi.icp_params(cuda_slam=True, max_iterations=50, eps=1e-3, max_distance_squared=10000.0)
    t0 = time.perf_counter()
    R, t, it, err = ctx.icp_register(before, after, p)
    first = (time.perf_counter() - t0) * 1e3
    times = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        R, t, it, err = ctx.icp_register(before, after, p)
        times.append((time.perf_counter() - t0) * 1e3)
    ctx.profile_enable(True)
    ctx.profile_select([])
    t0 = time.perf_counter()
    ctx.icp_load(before, after, p)
    t1 = time.perf_counter()
    done = ctx.icp_run(-1)
    ctx.icp_result()
    t2 = time.perf_counter()
    stages = ctx.icp_load_times()
    ctx.profile_enable(False)
    ctx.profile_select(None)
    passes = max(done, 1)
    return {"points": n, "ms": min(times), "ms_all": times, "first_call_ms": first, "iterations": it, "loop_passes": done,
            "ms_per_iteration_whole_call": min(times) / passes,
            "breakdown": {"load_stages_ms": stages, "load_ms": (t1 - t0) * 1e3, "iterations_ms": (t2 - t1) * 1e3,
                          "note": "stages drained one by one (profiling on): their sum exceeds what the same load costs inside `ms`"}}


def cpd_bunny(np, capi, ctx, world):
    """cfg 4 (bunny 14 904 x 14 904, cpd-weight .3, scale free) through mi_cpd_register: exact P (fixed cloud sharded over the ranks) and
    the parser's default hybrid mode (replicated on every rank of a multi-GPU context)."""
    gold = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(gold, "bunny_clouds.npz"))
    g = json.load(open(os.path.join(gold, "bunny_cpd.json")))
    before, after = z["before"], z["after"]
    pairs = float(len(before)) * len(after)
    out = {"workload": "cpd_bunny_14904", "n_gpus": world}
    if world == 1:
        # the initial sigma^2 on its own, both ways (the legs below start from cpu-slam's value handed in, so neither is inside them)
        for label, mode in (("sigma2_exact", capi.SIGMA2_EXACT), ("sigma2_cpu_sequential", capi.SIGMA2_CPU_SEQUENTIAL)):
            try:
                ctx.cpd_sigma_squared(before, after, mode)
                t0 = time.perf_counter()
                val = ctx.cpd_sigma_squared(before, after, mode)
                out[label] = {"value": float(val), "ms": (time.perf_counter() - t0) * 1e3}
            except capi.MiSlamError as e:                        # (a one-rank multi-GPU context -- the rehearsal -- has no sequential sum)
                out[label] = {"value": None, "ms": None, "skipped": str(e)}
        out["sigma2_cpu_sequential"]["note"] = ("cpu-slam's saturating sequential fp32 sum over all %.3g pairs, bit for bit (coherentpointdrift.cpp:126-139); "
                                                "what a registration with sigma2_mode = MI_SIGMA2_CPU_SEQUENTIAL pays once, before its first E-step" % pairs)
    modes = [("exact", capi.CPD_APPROX_NONE), ("hybrid", capi.CPD_APPROX_HYBRID)]
    for label, approx in modes:
        p = capi.cpd_params(max_iterations=50, sigma2_init=g["sigma2_init"], approximation=approx)
        ctx.cpd_register(before, after, p)                       # warm-up: allocations, code load
        walls = []
        for _ in range(3):                                       # the call as a caller makes it: no events on the stream
            t0 = time.perf_counter()
            sR, t, scale, it, err = ctx.cpd_register(before, after, p)
            walls.append(time.perf_counter() - t0)               # host buffers in, result out: upload included
        wall = min(walls)
        ctx.profile_enable(True)                                 # once more with HIP events around every kernel, for the breakdown only
        ctx.profile_select(None)
        ctx.profile_reset()
        # (ONE iteration per host check, capped at the iteration count just measured: a registration's iterations are otherwise enqueued in batches, and what
        # is still queued when the device-side stop rule fires returns at the `done` flag in 3-4 us -- launches of the kernels, not E-steps: ~11 % of the
        # bunny leg's launches, which diluted every per-launch figure of rounds 1-5 by as much.  The kernels' own times do not depend on the host's checks.)
        p_capped = capi.cpd_params(max_iterations=it, sigma2_init=g["sigma2_init"], approximation=approx, sync_every=1)
        t0 = time.perf_counter()
        ctx.cpd_register(before, after, p_capped)
        wall_profiled = time.perf_counter() - t0
        prof = {capi.KERNEL_NAMES[k]: ctx.profile_get(k) for k in range(len(capi.KERNEL_NAMES))}
        ctx.profile_enable(False)
        leg = {"iterations": it, "ms_total": wall * 1e3, "ms_per_em_iteration": wall * 1e3 / max(it, 1), "ms_total_all": [w * 1e3 for w in walls],
               "ms_total_with_events_around_every_kernel": wall_profiled * 1e3,
               "kernels_ms_per_launch": {k: v[0] / v[1] for k, v in prof.items() if v[1] > 0}}
        if label == "exact":
            f = g["final_scale_free"]
            leg["frobenius_vs_cpu_slam"] = float(np.sqrt(((sR - np.array(f["sR"])) ** 2).sum() + ((t - np.array(f["t"])) ** 2).sum()))
            leg["iterations_cpu_slam"] = f["iterations"]
            den, con = prof["cpd_denom"], prof["cpd_contract"]
            if den[1] > 0:
                leg["estep_pairs_per_s"] = pairs / world / (den[0] / den[1] * 1e-3)   # K7a: this rank's share of the N*M affinities
            # roofline of the E-step (K7a + K7b: the affinity is evaluated twice, P is never stored): vector-pipe busy time of the two
            # kernels from the committed counter profile, over what the same counters read on a saturated pipe (tools/valu_probe)
            if world == 1:
                roof = estep_roofline("cpd_bunny_14904", pairs)
                if roof is not None:
                    leg["roofline"] = roof
        out[label] = leg
    return out


def loaded_runtimes():
    """Every libamdhip64 / librccl object mapped into this process (/proc/self/maps): more than one of a kind means two ROCm runtimes."""
    import re
    found = {"hip": set(), "rccl": set()}
    try:
        for line in open("/proc/self/maps"):
            m = re.search(r"(/\S*/(libamdhip64|librccl)[^/\s]*)", line)
            if m:
                found["hip" if m.group(2) == "libamdhip64" else "rccl"].add(os.path.realpath(m.group(1)))
    except OSError:
        pass
    return {k: sorted(v) for k, v in found.items()}


def runtime_report(capi):
    """What the process actually resolved: the objects libmislam.so's HIP / RCCL calls bind to, their versions, and
    every copy of those libraries that is mapped.  `conflict`: a reason to refuse to start, or None.

    The decision: under a launcher the ranks import torch FIRST (gloo bootstrap); torch/lib carries libamdhip64.so and
    librccl.so under the SONAMEs libmislam.so asks for (libamdhip64.so.7, librccl.so.1), so the dynamic loader hands libmislam.so those
    already-mapped objects -- ONE runtime in the process, torch's.  That order is what the world-1 RCCL rehearsals ran.  What must never
    happen is two different RCCL (or HIP) objects of different versions mapped at once: then libmislam.so's communicator and the other
    copy each run their own runtime state on the same devices."""
    import ctypes
    info = capi.runtime_info()
    maps = loaded_runtimes()
    rep = {"libmislam_binds": info, "mapped": maps, "torch_imported_first": "torch" in sys.modules}
    conflict = None
    for kind, sym in (("rccl", "ncclGetVersion"), ("hip", "hipRuntimeGetVersion")):
        if len(maps[kind]) > 1:
            versions = {}
            for path in maps[kind]:
                try:
                    v = ctypes.c_int(0)
                    getattr(ctypes.CDLL(path), sym)(ctypes.byref(v))
                    versions[path] = v.value
                except Exception as e:      # noqa: BLE001
                    versions[path] = "unreadable: %s" % e
            rep[kind + "_versions"] = versions
            if len(set(versions.values())) > 1:
                conflict = "two different %s runtimes are mapped: %s" % (kind.upper(), versions)
    rep["conflict"] = conflict
    return rep


WALL_BUDGET_DEFAULT_S = 420.0


def wall_budget_s():
    """Seconds a `--gpus N` (N > 1) run may take before it is declared hung: MISLAM_BENCH_WALL_BUDGET_S, default 420."""
    try:
        return max(1.0, float(os.environ.get("MISLAM_BENCH_WALL_BUDGET_S", WALL_BUDGET_DEFAULT_S)))
    except ValueError:
        return WALL_BUDGET_DEFAULT_S


def stage_dir():
    """Where the ranks of ONE run leave their one-line stage files: handed down by the parent that started them, else derived from what every
    rank of a launcher's tree shares (its run id, or the launcher's pid: torchrun's agent is every worker's parent)."""
    d = os.environ.get("MISLAM_BENCH_STAGE_DIR")
    if not d:
        import tempfile
        d = os.path.join(tempfile.gettempdir(), "mislam_bench_%s_%d" % (os.environ.get("TORCHELASTIC_RUN_ID", "run"), os.getppid()))
    os.makedirs(d, exist_ok=True)
    return d


_STAGE = {"dir": None, "rank": 0, "t0": time.time()}


def stage(name):
    """One line per rank: the last stage it ENTERED and when (comm_init / first_allreduce / load / timed ... -- an RCCL initialisation that
    hangs and a kernel that hangs leave different last lines).  Written atomically; costs a rename, never inside a timed region."""
    d = _STAGE["dir"]
    if d is None:
        return
    path = os.path.join(d, "rank%d.stage" % _STAGE["rank"])
    try:
        with open(path + ".tmp", "w") as f:
            f.write("%s %.3f\n" % (name, time.time() - _STAGE["t0"]))
        os.replace(path + ".tmp", path)
    except OSError:
        pass


def read_stages(d):
    """{rank: {"stage": ..., "t": seconds since that rank started}} from the stage files under d."""
    out = {}
    for path in sorted(glob.glob(os.path.join(d, "rank*.stage"))):
        try:
            name, t = open(path).read().split()
            out[int(os.path.basename(path)[4:-6])] = {"stage": name, "t": float(t)}
        except (OSError, ValueError):
            continue
    return out


def budget_error_line(n_ranks, steps, warmup, budget, d, who):
    stages = read_stages(d) if d else {}
    return json.dumps({"metric": "icp_iterations_per_s", "value": None, "unit": "iterations/s", "n_gpus": n_ranks, "steps": steps, "warmup": warmup,
                       "error": "wall budget of %.0f s exceeded (%s); the ranks' last stages are in `stages`" % (budget, who),
                       "wall_budget_s": budget, "ranks_seen": len(stages), "stages": {str(r): v for r, v in sorted(stages.items())}})


def start_rank_watchdog(rank, world, steps, warmup):
    """Ranks started by a launcher other than self_launch (the driver's own torchrun line) have no parent of ours above them: every rank
    carries a daemon thread that ends ITS process when the budget runs out -- a thread blocked inside ncclCommInitRank or a
    hipStreamSynchronize holds no Python lock, so the watchdog still runs -- and rank 0's prints the diagnostic line first."""
    import threading
    budget = wall_budget_s()
    d = _STAGE["dir"]
    # under self_launch the parent's budget is the same number: fire a little earlier so that a live rank 0 gets to print
    fire_after = budget - (10.0 if os.environ.get("MISLAM_BENCH_PARENT") == "1" and budget > 30.0 else 0.0)
    done = threading.Event()

    def watch():
        if done.wait(fire_after):
            return
        if rank == 0:
            sys.stdout.write(budget_error_line(world, steps, warmup, budget, d, "rank watchdog") + "\n")
            sys.stdout.flush()
        else:
            time.sleep(1.0)                    # (rank 0's line first)
        sys.stderr.write("bench.py rank %d: wall budget of %.0f s exceeded at stage %r\n" % (rank, budget, read_stages(d).get(rank, {}).get("stage")))
        sys.stderr.flush()
        os._exit(124)
    threading.Thread(target=watch, name="bench-wall-budget", daemon=True).start()
    return done


def self_launch(n_ranks, steps, warmup):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks as a fresh child process tree -- `python -m
    torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`, the driver's own launch line -- relay rank 0's
    JSON line and return the child's exit code.  A child, not an exec; called before this process has imported the package or
    touched the GPU.  The child tree gets a WALL BUDGET (wall_budget_s): its stdout is read without ever blocking past it, and on expiry
    the tree is terminated, ONE JSON line with "error", every rank's last stage and `ranks_seen` is printed and the exit code is 124 --
    an RCCL initialisation hang on a first unattended multi-GPU run becomes a diagnostic instead of the driver's kill."""
    import selectors
    import signal
    import subprocess
    import tempfile
    # --standalone: the launcher binds its own rendezvous to a free port (no pre-picked port another process could take between a probe
    # and the launcher's bind -- ADVICE r04); --local-addr: the workers' MASTER_ADDR, an address that resolves everywhere
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node=%d" % n_ranks,
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    budget = wall_budget_s()
    d = tempfile.mkdtemp(prefix="mislam_bench_")
    env["MISLAM_BENCH_STAGE_DIR"] = d
    env["MISLAM_BENCH_PARENT"] = "1"
    child = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, start_new_session=True)   # own process group: the whole tree can be ended
    os.set_blocking(child.stdout.fileno(), False)
    sel = selectors.DefaultSelector()
    sel.register(child.stdout, selectors.EVENT_READ)
    deadline = time.time() + budget
    lines, errors, pending, eof = 0, 0, b"", False

    def relay(raw):
        nonlocal lines, errors
        line = raw.decode("utf-8", "replace")
        if line.lstrip().startswith("{") and '"metric"' in line:      # rank 0 prints ONE JSON line; anything else a rank writes to stdout goes to stderr
            sys.stdout.write(line if line.endswith("\n") else line + "\n")
            sys.stdout.flush()
            lines += 1
            errors += '"error"' in line
        else:
            sys.stderr.write(line)

    while not eof and time.time() < deadline:
        if not sel.select(timeout=min(1.0, max(0.0, deadline - time.time()))):
            if child.poll() is not None and not pending:
                # (the launcher is gone; whatever a straggler still holds the pipe open for is not waited for beyond the budget)
                pass
            continue
        try:
            chunk = os.read(child.stdout.fileno(), 65536)
        except BlockingIOError:
            continue
        if not chunk:
            eof = True
            break
        pending += chunk
        while b"\n" in pending:
            raw, pending = pending.split(b"\n", 1)
            relay(raw + b"\n")
    if pending:
        relay(pending)
    if not eof:
        # budget spent: end the tree (TERM, then KILL), say where every rank was
        for sig, grace in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 5.0)):
            try:
                os.killpg(child.pid, sig)
            except (ProcessLookupError, PermissionError):
                break
            try:
                child.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        if errors == 0:
            sys.stdout.write(budget_error_line(n_ranks, steps, warmup, budget, d, "parent of the launcher") + "\n")
            sys.stdout.flush()
        sys.stderr.write("bench.py: the %d-rank child exceeded its wall budget of %.0f s and was terminated\n" % (n_ranks, budget))
        rc = 124
    else:
        try:
            rc = child.wait(timeout=max(5.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
            rc = 124
        if rc == 0 and (lines != 1 or errors):
            sys.stderr.write("bench.py: the %d-rank child printed %d result lines (%d with an error)\n" % (n_ranks, lines, errors))
            rc = 1
    import shutil
    shutil.rmtree(d, ignore_errors=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # default: iterations 6..25 of one registration (what the driver's own command line asks for); the search is data-dependent:
    # in early iterations a third of the moving cloud still lies outside the fixed one and walks the box hierarchy
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=1000000, help="N = M, BASELINE.json: 10^6")
    ap.add_argument("--dist-mode", type=int, default=0, help="0 = cpu-slam rounding (parity default), 1 = fma")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sizes", action="store_true", help="skip the N = 1e4 / 1e5 / 1e7 legs")
    ap.add_argument("--no-whole-call", action="store_true", help="skip the whole-registration legs (host buffers in, 50 iterations)")
    ap.add_argument("--no-cpd", action="store_true", help="skip the CPD leg (cfg 4 on the bunny clouds)")
    ap.add_argument("--shard", choices=["auto", "target", "source"], default="auto",
                    help="what N > 1 GPUs split: auto = moving cloud for the indexed searches, fixed cloud for the every-pair search")
    ap.add_argument("--brute-ref-steps", type=int, default=2,
                    help="untimed every-pair steps measured after the timed region when an indexed search was used (0 = skip)")
    ap.add_argument("--nn", choices=["auto", "brute", "tree", "grid"], default="auto",
                    help="search strategy (identical results): auto = the library default (cell grid at this size)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, args.steps, args.warmup))       # (nothing has touched the GPU or imported the package yet)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world                          # under a launcher the launcher's world size is the truth

    # MISLAM_BENCH_FORCE_DIST=1 takes the multi-process path (gloo bootstrap, RCCL communicator) even with one rank:
    # the rehearsal a single-GPU box allows
    use_dist = world > 1 or os.environ.get("MISLAM_BENCH_FORCE_DIST") == "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL between processes on this driver); before anything loads the HIP runtime
    quiet_host_pools()           # (before numpy / torch are imported)
    watchdog_done = None
    if use_dist:
        # the wall budget of a multi-rank run: stage files + a watchdog per rank, before anything can hang
        _STAGE.update(dir=stage_dir(), rank=rank, t0=time.time())
        stage("start")
        watchdog_done = start_rank_watchdog(rank, world, args.steps, args.warmup)
    dist = None
    if use_dist:
        stage("gloo_init")
        # torch first: its bundled HIP/RCCL runtime must be the one in the process before libmislam.so is loaded
        import torch  # noqa: F401
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")

    if os.environ.get("MISLAM_BENCH_DRYRUN") == "1":
        # launcher check on a box WITHOUT a GPU (tests/test_bench_launch.py): bootstrap, one collective, one line, clean exit --
        # no context is created, nothing is measured, and the line says so
        import torch
        seen = torch.tensor([float(1 << rank)], dtype=torch.float64)
        stage("comm_init")
        # (tests/test_bench_launch.py: one rank that never arrives -- what an RCCL initialisation hang looks like from outside)
        if os.environ.get("MISLAM_BENCH_DRYRUN_HANG_RANK") == str(rank):
            time.sleep(3600.0)
        stage("first_allreduce")
        if dist is not None:
            dist.all_reduce(seen, op=dist.ReduceOp.SUM)
            dist.barrier()
        if rank == 0:
            mask = int(seen.item())
            print(json.dumps({"metric": "icp_iterations_per_s", "value": None, "unit": "iterations/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "dry_run": "launcher flow only: no device touched, nothing measured",
                              "rccl": {"nranks": world, "ranks_seen_mask": mask, "ranks_seen": bin(mask).count("1"), "transport": "gloo (dry run)"}}), flush=True)
        stage("done")
        if watchdog_done is not None:
            watchdog_done.set()
        if dist is not None:
            dist.destroy_process_group()
        return

    stage("package_load")
    import numpy as np
    from __graft_entry__ import load_package
    capi = load_package().capi
    # which ROCm runtime(s) this process holds, BEFORE any device call: two different RCCL / HIP objects mapped at once = refuse to start
    runtime = runtime_report(capi)
    if use_dist and runtime["conflict"]:
        sys.stderr.write("bench.py rank %d: %s -- refusing to start (see runtime_report in bench.py)\n" % (rank, runtime["conflict"]))
        sys.exit(3)

    # Rehearsal of the N > 1 flow on a box with ONE GPU (tools/gpu_dist_rehearsal.sh): MISLAM_BENCH_DEVICE pins every rank to
    # that device and MISLAM_BENCH_TRANSPORT=gloo swaps RCCL (which refuses two ranks on one device) for the caller's-transport
    # context over gloo.  It validates the launch line, sharding, barriers and the report -- its numbers mean nothing.
    rehearsal_transport = os.environ.get("MISLAM_BENCH_TRANSPORT", "rccl")
    if "MISLAM_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["MISLAM_BENCH_DEVICE"])
    if use_dist and rehearsal_transport == "gloo":
        import torch
        sign = np.uint64(1 << 63)

        def exchange(arr, kind):
            if kind == capi.EXCHANGE_MIN_U64:      # gloo has no unsigned MIN: flip the top bit and take the signed one
                tk = torch.from_numpy((arr ^ sign).view(np.int64))
                dist.all_reduce(tk, op=dist.ReduceOp.MIN)
                arr[:] = tk.numpy().view(np.uint64) ^ sign
            else:
                dist.all_reduce(torch.from_numpy(arr), op=dist.ReduceOp.SUM)
        stage("comm_init")
        ctx = capi.Context(local_rank, rank, world, exchange=exchange)
    elif use_dist:
        uid = [capi.dist_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        stage("comm_init")                     # ncclCommInitRank: every rank must arrive
        ctx = capi.Context(local_rank, rank, world, uid[0])
    else:
        ctx = capi.Context(local_rank)
    if use_dist:
        # a 1-element all-reduce on the context's stream right behind the communicator's creation: a hang HERE is the transport's
        # (initialisation finished, the first collective did not), a hang later is a kernel's or the bench's
        stage("first_allreduce")
        nr0, rk0, seen0 = ctx.dist_info()
        if bin(int(seen0)).count("1") != world:
            sys.stderr.write("bench.py rank %d: the first all-reduce saw ranks %s of %d\n" % (rank, bin(int(seen0)), world))
            sys.exit(4)
    stage("load")

    def barrier():
        ctx.synchronize()
        if dist is not None:
            dist.barrier()
        ctx.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        import torch
        tt = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    nn_mode = {"auto": capi.NN_AUTO, "brute": capi.NN_BRUTEFORCE, "tree": capi.NN_TREE, "grid": capi.NN_GRID}[args.nn]
    shard_mode = {"auto": capi.SHARD_AUTO, "target": capi.SHARD_TARGET, "source": capi.SHARD_SOURCE}[args.shard]

    def plan(n, m, nn_choice, shard_choice):
        """What the library will do with these settings (mi_slam.h MI_NN_AUTO / MI_SHARD_AUTO): local sizes, kernel, split."""
        indexed_if = lambda mm: nn_choice in ("tree", "grid") or (nn_choice == "auto" and mm >= capi.NN_INDEX_MIN_POINTS)
        source_sharded = use_dist and (shard_choice == "source" or (shard_choice == "auto" and indexed_if(m)))
        if source_sharded:
            n_local, m_local = capi.source_share(n, rank, world), m
        else:
            lo, hi = capi.shard_range(m, rank, world)
            n_local, m_local = n, hi - lo
        indexed = indexed_if(m_local)
        mode = {"auto": capi.NN_AUTO, "brute": capi.NN_BRUTEFORCE, "tree": capi.NN_TREE, "grid": capi.NN_GRID}[nn_choice]
        return {"n_local": n_local, "m_local": m_local, "indexed": indexed, "source_sharded": source_sharded,
                "kernel": ctx.nn_kernel_name(n_local, m_local, mode)}

    def nn_figures(pl, nn_ms, nn_n):
        nn_avg_s = nn_ms / max(nn_n, 1) * 1e-3
        alg_bytes = 20.0 * pl["n_local"] + 12.0 * pl["m_local"]   # this rank: 12 B source xyz + 8 B packed key per moving point, 12 B per fixed point
        achieved_gbs = alg_bytes / nn_avg_s / 1e9
        return {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": None, "kernel": pl["kernel"], "avg_launch_ms": nn_avg_s * 1e3, "launches": nn_n,
                "algorithmic_bytes_per_launch": alg_bytes}

    def timed_run(before, after, params, warmup, steps, events=True):
        """load (untimed) -> warm-up iterations -> `steps` iterations between barriers, HIP events around the search kernel only
        (events=False: none at all -- the steps as a caller's registration runs them)."""
        stage("load")
        ctx.icp_load(before, after, params)            # H2D upload, SoA conversion, index build: outside the timed region
        stage("warmup")
        if warmup > 0:
            ctx.icp_run(warmup)
        ctx.profile_enable(events)
        ctx.profile_select([capi.KERNEL_NN])
        ctx.profile_reset()
        barrier()
        stage("timed")                                 # (a rename before the clock starts, nothing inside the timed region)
        t0 = time.perf_counter()
        done = ctx.icp_run(steps)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        assert done == steps, "ran %d of %d steps" % (done, steps)
        nn = ctx.profile_get(capi.KERNEL_NN)
        ctx.profile_enable(False)
        return elapsed, nn

    # ------------------------------------------------------------------------------------------------ headline: N = M = --points
    before, after = synth_cloud(np, args.points)
    n, m = len(before), len(after)
    pl = plan(n, m, args.nn, args.shard)
    # eps = 0: the device-side stop rule is evaluated every step and never fires -> every step is a full iteration
    # sync_every = steps: the host looks at the state once, after the timed steps (the stop rule itself runs on the device every
    # step) -- the headline does not depend on the library's batch heuristic
    params = capi.icp_params(eps=0.0, max_iterations=-1, dist_mode=args.dist_mode, nn_mode=nn_mode, shard_mode=shard_mode, sync_every=max(args.steps, 1))
    elapsed, nn_prof = timed_run(before, after, params, args.warmup, args.steps)
    R, t, iters, err, why = ctx.icp_result()
    # the same steps with the library's OWN host-check interval (sync_every = 0 -> mi_icp_auto_batch: 16 at this size on one GPU): what a
    # registration pays per iteration when the host looks at the state as often as mi_icp_run does by default (a flush + a read-back per batch)
    params_default = capi.icp_params(eps=0.0, max_iterations=-1, dist_mode=args.dist_mode, nn_mode=nn_mode, shard_mode=shard_mode, sync_every=0)
    elapsed_default, _ = timed_run(before, after, params_default, args.warmup, args.steps)
    auto_batch = capi.icp_auto_batch(n, m, world, pl["source_sharded"], not pl["indexed"])
    elapsed_plain, _ = timed_run(before, after, params, args.warmup, args.steps, events=False)
    # The headline's window (iterations warmup .. warmup + steps of one registration) is a TRANSIENT: the search gets cheaper as the clouds close
    # in.  The same number of steps late in the same registration, for the record: what an iteration costs once a third of
    # the moving cloud no longer hangs out of the fixed one.
    steady = None
    if not args.no_sizes:
        late = max(40, 2 * (args.warmup + args.steps))
        el_late, nn_late = timed_run(before, after, params, late, args.steps)
        steady = {"warmup": late, "steps": args.steps, "iterations_per_s": args.steps / el_late, "ms_per_step": el_late / args.steps * 1e3,
                  "nn_avg_launch_ms": nn_late[0] / max(nn_late[1], 1), "error_after_steps": ctx.icp_result()[3],
                  "note": "iterations %d..%d of the same registration (the headline times iterations %d..%d)" % (late + 1, late + args.steps, args.warmup + 1, args.warmup + args.steps)}
    # (the legs below continue from the headline's registration state: reload it)
    timed_run(before, after, params, args.warmup, args.steps)
    headline_allreduce = None
    # per-kernel breakdown of a step: a few more (untimed) iterations with events around every kernel
    stage("breakdown")
    ctx.profile_enable(True)
    ctx.profile_select(None)
    ctx.profile_reset()
    extra = ctx.icp_run(min(args.steps, 5))
    breakdown = {capi.KERNEL_NAMES[k]: ctx.profile_get(k) for k in range(len(capi.KERNEL_NAMES))}
    if use_dist:
        ar = breakdown.get("allreduce", (0.0, 0))
        if ar[1] > 0:        # the headline path's own collective: one 64 x 18-double sum per iteration (+ a 2-double one per host batch), event-timed
            headline_allreduce = {"launches": ar[1], "ms_per_launch": ar[0] / ar[1], "payload_bytes": 8 * 18 * 64,
                                  "collective": "ncclAllReduce(ncclDouble, ncclSum) of the 64 reduced rows x (16 moments + 2 error sums) on the context's stream (mislam_api.hip allreduce_sum_f64)"
                                                if rehearsal_transport != "gloo" else "caller's transport (gloo rehearsal)"}
    breakdown = {k: v[0] / v[1] for k, v in breakdown.items() if v[1] > 0 and extra > 0}

    # Outside the timed region: the same steps with the every-pair search (K1), for the brute-force roofline figures the
    # north star asks for.  Same keys, same registration -- only the number of evaluated pairs differs.
    brute_fig = None
    if pl["indexed"] and args.brute_ref_steps > 0:
        bpl = plan(n, m, "brute", "source" if pl["source_sharded"] else "target")
        ctx.icp_load(before, after, capi.icp_params(eps=0.0, max_iterations=-1, dist_mode=args.dist_mode, nn_mode=capi.NN_BRUTEFORCE,
                                                    shard_mode=capi.SHARD_SOURCE if pl["source_sharded"] else capi.SHARD_TARGET))
        ctx.icp_run(1)
        ctx.profile_reset()
        ctx.icp_run(args.brute_ref_steps)
        bp = ctx.profile_get(capi.KERNEL_NN)
        brute_fig = nn_figures(bpl, bp[0], bp[1])
        pairs_per_s = bpl["n_local"] * float(bpl["m_local"]) / (brute_fig["avg_launch_ms"] * 1e-3)
        lane_ops = pairs_per_s * OPS_PER_PAIR[args.dist_mode]
        brute_fig["valu"] = {"bound": "fp32-valu-issue", "achieved": lane_ops, "peak": VALU_LANE_OPS_PEAK, "unit": "lane-ops/s",
                             "frac": lane_ops / VALU_LANE_OPS_PEAK, "pairs_per_s": pairs_per_s,
                             "ops_per_pair": OPS_PER_PAIR[args.dist_mode]}
        brute_fig["note"] = "every-pair search: fp32-VALU-bound (see valu), its compulsory HBM bytes are ~4 us of bandwidth"
    ctx.profile_enable(False)

    # cfg 3 as BASELINE.json words it: fixed cloud sharded across the ranks, every-pair search, ONE ncclAllReduce(ncclUint64,
    # ncclMin) of the N packed (min-dist, argmin) keys per iteration.  Event-timed, outside the headline (the indexed search with
    # a sharded MOVING cloud is what the headline measures, because it is two orders of magnitude faster).
    target_leg, rccl = None, None
    if use_dist:
        stage("target_sharded_leg")
        ctx.icp_load(before, after, capi.icp_params(eps=0.0, max_iterations=-1, dist_mode=args.dist_mode, nn_mode=capi.NN_BRUTEFORCE,
                                                    shard_mode=capi.SHARD_TARGET))
        ctx.icp_run(1)
        ctx.profile_enable(True)
        ctx.profile_select(None)
        ctx.profile_reset()
        barrier()
        t0 = time.perf_counter()
        ctx.icp_run(2)
        barrier()
        t_leg = max_over_ranks(time.perf_counter() - t0)
        ar, nnp = ctx.profile_get(capi.KERNEL_ALLREDUCE), ctx.profile_get(capi.KERNEL_NN)
        ctx.profile_enable(False)
        lo, hi = capi.shard_range(m, rank, world)
        target_leg = {"nn": "bruteforce", "shard": "fixed cloud, rank 0 holds [%d, %d)" % (lo, hi), "steps": 2,
                      "ms_per_step": t_leg / 2 * 1e3, "nn_kernel_ms": nnp[0] / max(nnp[1], 1),
                      "allreduce_u64_min": {"launches": ar[1], "ms_per_launch": ar[0] / max(ar[1], 1), "payload_bytes": 8 * n,
                                            "collective": "ncclAllReduce(ncclUint64, ncclMin) on the context's stream (mislam_api.hip allreduce_min_u64)"
                                                          if rehearsal_transport != "gloo" else "caller's transport (gloo rehearsal)"}}
        nr, rk, seen = ctx.dist_info()
        rccl = {"nranks": nr, "rank0_sees_rank": rk, "ranks_seen_mask": seen, "ranks_seen": bin(seen).count("1"),
                "transport": "rccl" if rehearsal_transport != "gloo" else "gloo exchange context (rehearsal)"}

    # The other sizes BASELINE.json names, a few steps each (same recipe, same rules; 1e7 costs ~2 s of host-side generation)
    sizes = None
    if not args.no_sizes:
        sizes = {}
        for sn in SWEEP_SIZES:
            if sn == n:
                continue
            sb, sa = synth_cloud(np, sn)
            spl = plan(sn, sn, args.nn, args.shard)
            s_el, s_nn = timed_run(sb, sa, params, 5, 10)
            fig = nn_figures(spl, s_nn[0], s_nn[1])
            sizes[str(sn)] = {"iterations_per_s": 10 / s_el, "ms_per_step": s_el / 10 * 1e3, "steps": 10, "warmup": 5,
                              "nn_kernel": fig["kernel"], "nn_avg_launch_ms": fig["avg_launch_ms"], "hbm_achieved_GBs": fig["achieved"],
                              "hbm_frac": fig["frac"]}
            if rank == 0 and world == 1 and not args.no_cpu_baseline and sn <= 100000:
                sizes[str(sn)]["cpu_baseline"] = cpu_baseline_direct(np, sb, sa)      # 1e4 and 1e5 timed directly
            del sb, sa

    # Outside the timed region as well: the CPD leg of BASELINE.json's metric ("CPD E-step on bunny"), cfg 4 on the committed
    # bunny clouds from cpu-slam's own sigma^2_0 -- exact P and the parser's default hybrid mode.  One GPU by default; with
    # MISLAM_BENCH_CPD=1 also on N > 1 (fixed cloud sharded, one 24-double all-reduce per EM iteration).
    cpd = None
    if not args.no_cpd and (world == 1 or os.environ.get("MISLAM_BENCH_CPD") == "1"):
        stage("cpd_leg")
        cpd = cpd_bunny(np, capi, ctx, world)
        if world == 1 and rank == 0 and not args.no_cpu_baseline:
            z = np.load(os.path.join(ROOT, "tests", "golden", "bunny_clouds.npz"))
            s2 = json.load(open(os.path.join(ROOT, "tests", "golden", "bunny_cpd.json")))["sigma2_init"]
            cpd["cpu_baseline"] = cpu_baseline_cpd(np, z["before"], z["after"], s2)
        if world == 1:
            cpd["published_size"] = cpd_published_size(np, capi, ctx)

    # Outside the timed region too: whole registrations on host buffers, as the reference times a SlamFunc
    whole = None
    if world == 1 and not args.no_whole_call:
        whole = {str(wn): whole_call(np, capi, ctx, wn) for wn in (100000, 1000000)}
        # cfg 1 the same way: the bunny clouds with config/default.json's rules (cpu-slam's 39 iterations), host buffers in, result out
        gold = os.path.join(ROOT, "tests", "golden")
        bz = np.load(os.path.join(gold, "bunny_clouds.npz"))
        bp = json.load(open(os.path.join(gold, "bunny_icp.json")))["params"]
        whole["bunny_14904"] = whole_call(np, capi, ctx, 14904, clouds=(bz["before"], bz["after"]),
                                          params=capi.icp_params(eps=bp["eps"], max_iterations=bp["max_iterations"], max_distance_squared=bp["max_distance_squared"]))

    # The floor of the headline path's collective: the 64 x 18-double sum of one iteration through a ONE-rank RCCL communicator, event-timed
    # (what ncclAllReduce costs on this stream before a single byte crosses xGMI).  N = 1 only, outside the timed region, never fatal.
    allreduce_floor = None
    if world == 1 and not use_dist and os.environ.get("MISLAM_BENCH_NO_RCCL_FLOOR") != "1":
        try:
            dctx = capi.Context(local_rank, 0, 1, capi.dist_unique_id())
            fb, fa = synth_cloud(np, 100000)
            dctx.icp_load(fb, fa, capi.icp_params(eps=0.0, max_iterations=-1, shard_mode=capi.SHARD_SOURCE, sync_every=8))
            dctx.icp_run(4)
            dctx.profile_enable(True)
            dctx.profile_select([capi.KERNEL_ALLREDUCE])
            dctx.profile_reset()
            dctx.icp_run(16)
            ar = dctx.profile_get(capi.KERNEL_ALLREDUCE)
            dctx.profile_enable(False)
            dctx.close()
            allreduce_floor = {"ranks": 1, "launches": ar[1], "ms_per_launch": ar[0] / max(ar[1], 1), "payload_bytes": 8 * 18 * 64,
                               "note": "ncclAllReduce(ncclDouble, ncclSum) of the 64 reduced rows on a one-rank communicator, HIP events around it: the latency floor of "
                                       "the one collective an iteration of the headline path issues; every rank beyond one adds xGMI hops to it"}
        except Exception as e:      # noqa: BLE001
            allreduce_floor = {"error": str(e)}

    if rank == 0:
        workload = "icp_synthetic_uniform_n%d" % n
        roof = nn_figures(pl, nn_prof[0], nn_prof[1])
        if pl["indexed"]:
            roof["note"] = ("exact search through the cell grid (box-hierarchy walk for the lanes whose neighbour is farther than two cells): "
                            "bound by vector-instruction issue, not by HBM -- see `issue`; algorithmic bytes are the same 20N+12M as for "
                            "the every-pair kernel it replaces")
        if world == 1 and args.dist_mode == 0:
            prof_c, src, same_cmd = committed_profile(workload, roof["kernel"], args.steps, args.warmup)
            if prof_c is not None:
                # HBM bytes per launch from the committed counter passes.  Attached as `traffic` only when that profile was taken
                # with this run's --steps / --warmup; otherwise it is kept apart, labelled as what it is.
                key = "traffic" if same_cmd else "traffic_from_committed_profile"
                roof[key] = prof_c.get("traffic_bytes_per_launch")
                roof["traffic_source"] = src + (" (rocprofv3 --pmc passes of this same command, means per launch; not measured by this run)"
                                                if same_cmd else " (taken with --steps %s --warmup %s)" % (prof_c.get("steps"), prof_c.get("warmup")))
                cal, cal_src = valu_calibration()
                if cal is not None and prof_c.get("valu_busy_quadcycles_per_gui_cycle"):
                    # ONE number: the vector pipe's busy time in this kernel over what the same counters read on a kernel that only
                    # issues vector instructions at 8 waves per SIMD (packed instructions weigh by the time they hold the pipe)
                    probe = cal["kernels"]["valu_probe<0>"]
                    sat = probe["valu_busy_quadcycles_per_gui_cycle"]
                    ceiling = cal.get("valu_busy_ceiling", VALU_BUSY_CEILING)
                    rate = prof_c["valu_wave_instructions_per_launch"] / (prof_c["avg_launch_ms"] * 1e-3)
                    roof["issue"] = {"bound": "valu-issue", "achieved": prof_c["valu_busy_quadcycles_per_gui_cycle"], "peak": ceiling,
                                     "unit": "vector-pipe busy quad-cycles per GPU cycle", "frac": prof_c["valu_busy_quadcycles_per_gui_cycle"] / ceiling,
                                     "peak_is": "the counter's ceiling: 1 024 SIMDs / 4 cycles per wave instruction / 8 XCDs summed in GRBM_GUI_ACTIVE",
                                     "probe_reads": sat, "every_pair_kernel_reads": cal["kernels"].get("nn_bruteforce_kernel", {}).get("valu_busy_quadcycles_per_gui_cycle"),
                                     "frac_of_probe": prof_c["valu_busy_quadcycles_per_gui_cycle"] / sat,
                                     "calibration_source": cal_src, "same_code_as_profile": same_cmd,
                                     "wave_instructions_per_s": rate, "probe_wave_instructions_per_s": probe["wave_instructions_per_s"],
                                     "instruction_rate_over_probe": rate / probe["wave_instructions_per_s"],
                                     "valu_instructions_per_wave": prof_c.get("valu_instructions_per_wave"),
                                     "lanes_active_of_64": prof_c.get("lanes_active_of_64"),
                                     "l1_line_accesses_per_launch": prof_c.get("tcp_total_cache_accesses_per_launch"),
                                     "source": src}
        out = {
            "metric": "icp_iterations_per_s", "value": args.steps / elapsed, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "n_before": n, "n_after": m,
                       "nn": ({"nn_tree_kernel": "box-hierarchy (exact)", "nn_grid_kernel": "cell-grid + box-hierarchy fallback (exact)"}
                              .get(pl["kernel"], "bruteforce")),
                       "dist_arithmetic": "cpu_rounding" if args.dist_mode == 0 else "fma",
                       "compose": "cpu_additive",
                       "parallelism": ("single GPU" if world == 1 else
                                       "moving cloud sharded x%d, fixed cloud replicated, one 64 x 18-double RCCL sum all-reduce per iteration" % world
                                       if pl["source_sharded"] else
                                       "fixed cloud sharded x%d, RCCL u64-min all-reduce of the packed keys + one 64 x 18-double sum per iteration" % world),
                       "error_after_steps": err},
            "roofline": roof,
            "ms_per_step_without_events": elapsed_plain / args.steps * 1e3,     # the headline's timed region carries two HIP event records per step (the roofline's live kernel time)
            "library_default_batch": {"sync_every": auto_batch, "iterations_per_s": args.steps / elapsed_default, "ms_per_step": elapsed_default / args.steps * 1e3,
                                      "note": "the same timed steps with mi_icp_run's default host-check interval (mi_icp_auto_batch) instead of ONE check "
                                              "behind all of them: every batch ends in a flush (transform + error of the last iteration) and a 256-byte read-back"},
            "kernels_ms_per_step": breakdown,     # ms per launch, from the untimed follow-up iterations (every kernel event-timed)
        }
        out["runtime"] = runtime
        if allreduce_floor is not None:
            out["allreduce_f64_sum_floor"] = allreduce_floor
        if world > 1 or use_dist:
            # what to expect BEFORE reading the first multi-GPU numbers: from the one-GPU emulation of a rank's share
            # (tools/search_vs_queries.py, profiles/r04_search_vs_queries.log): the search of the indexed path shrinks with the rank's share of the
            # moving cloud, the other ~0.02 ms of a step (rows reduce + solve + launch boundaries) and the all-reduce do not
            emu = {1000000: {1: 0.0819 + 0.0187, 2: 0.0609 + 0.0230, 4: 0.0488 + 0.0221, 8: 0.0418 + 0.0214},
                   10000000: {1: 0.8159 + 0.0898, 2: 0.4230 + 0.0638, 4: 0.2218 + 0.0296, 8: 0.1294 + 0.0282}}
            out["expected_scaling"] = {
                "source": "profiles/r04_search_vs_queries.log (one GPU running one rank's share; all-reduce latency NOT included)",
                "speedup_upper_bound": {str(pts): {str(w): round(t[1] / t[w], 2) for w in t} for pts, t in emu.items()},
                "note": "strong scaling of an indexed search at 1e6 points is latency-limited (<= 1.6x on 8 GPUs before the all-reduce is paid): a launch cannot be shorter "
                        "than its longest wave; 1e7 points (sizes['10000000'] in this line, the same --gpus) is the size where a curve can show (~5.8x at 8); cfg 3 as "
                        "BASELINE.json words it (target-sharded every-pair search: `target_sharded`) scales ~linearly but is two orders of magnitude slower than one GPU's grid"}
        if use_dist and rehearsal_transport == "gloo":
            out["rehearsal"] = "ranks share device %d over the gloo exchange context: flow check only, not a measurement" % local_rank
        if steady is not None:
            out["late_iterations"] = steady
        if whole is not None:
            out["whole_call"] = whole
        if sizes is not None:
            out["sizes"] = sizes
        if brute_fig is not None:
            out["bruteforce_nn"] = brute_fig
        if headline_allreduce is not None:
            out["allreduce_f64_sum"] = headline_allreduce
        if target_leg is not None:
            out["target_sharded"] = target_leg
            out["rccl"] = rccl
        if cpd is not None:
            out["cpd_bunny"] = cpd
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(np, before, after)
        print(json.dumps(out), flush=True)

    stage("report_done")
    ctx.close()
    stage("done")
    if watchdog_done is not None:
        watchdog_done.set()
    if dist is not None:
        dist.barrier()
        destroy = getattr(dist, "destroy_process_group", None)
        if destroy:
            destroy()
        if rank == 0 and _STAGE["dir"] and os.environ.get("MISLAM_BENCH_PARENT") != "1":     # (a launcher of someone else's: nobody above us removes the stage files)
            import shutil
            shutil.rmtree(_STAGE["dir"], ignore_errors=True)


if __name__ == "__main__":
    main()
