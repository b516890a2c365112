#!/usr/bin/env python3
"""Reduces the rocprofv3 outputs of tools/gpu_profiles.sh (gpurun_out/<R>_<pass>/, R = round tag) to the summaries bench.py reads from profiles/:
    <R>_bench_n1e6_kernel_stats.csv             rocprofv3 --stats per-kernel summary of the whole command
    <R>_bench_n1e6_nn_grid_timed_dispatches.csv the kernel-trace rows of the TIMED launches of the search kernel (avg_launch_ms recomputable)
    <R>_bench_n1e6_nn_grid_counters.json        per-launch means of the counters over those launches + what produced them (git head, source hash)
    <R>_valu_calibration.json                   the same SQ counters on tools/valu_probe and on the every-pair kernel of the same run
    <R>_cpd_estep_counters.json                 the CPD E-step kernels of the same run, one entry per workload (bunny; the published 49 000 points)
    <R>_cpd_estep_timed_dispatches.csv          the kernel-trace rows those entries are means over, labelled by workload
"""
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

S, W = int(sys.argv[1]), int(sys.argv[2])
R = os.environ.get("R", "r06")
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out")


def rows(d, pat):
    out = []
    for f in glob.glob(os.path.join(OUT, "%s_%s" % (R, d), "**", "*%s" % pat), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def source_hash():
    sys.path.insert(0, ROOT)
    from bench import search_source_hash           # (one definition: the code of the search kernel, comments and white space aside)
    return search_source_hash()


def counter_series(d, counter, kernel):
    rs = [r for r in rows(d, "counter_collection.csv") if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    rs.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rs]


def mean(v):
    return sum(v) / len(v) if v else None


SQ1 = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVES", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"]
SQ2 = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_SCA", "GRBM_GUI_ACTIVE"]
SETS = (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("sq1", SQ1), ("sq2", SQ2),
        ("tcp", ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"]))

# ---- kernel trace: per-kernel stats + the timed launches of the search kernel
for f in glob.glob(os.path.join(OUT, R + "_stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(OUT, R + "_bench_n1e6_kernel_stats.csv"))
tr = [r for r in rows("stats", "kernel_trace.csv") if "nn_grid_kernel" in r["Kernel_Name"]]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
timed_rows = tr[W:W + S]
with open(os.path.join(OUT, R + "_bench_n1e6_nn_grid_timed_dispatches.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["launch_of_this_kernel", "Dispatch_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "duration_ns"])
    for k, r in enumerate(timed_rows):
        w.writerow([W + k, r.get("Dispatch_Id", ""), r["Kernel_Name"][:60], r["Start_Timestamp"], r["End_Timestamp"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
timed = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in timed_rows]


def timed_mean(d, counter):
    return mean(counter_series(d, counter, "nn_grid_kernel")[W:W + S])


c = {k: timed_mean(d, k) for d, ks in SETS for k in ks}
try:
    head = subprocess.check_output(["git", "rev-parse", "HEAD"], cwd=ROOT, stderr=subprocess.DEVNULL).decode().strip()
except Exception:
    head = None
doc = {
    "workload": "icp_synthetic_uniform_n1000000", "kernel": "nn_grid_kernel", "steps": S, "warmup": W,
    "git_head": head, "source_hash": source_hash(),
    "command": "rocprofv3 --pmc <one pass per counter set> -- python3 bench.py --steps %d --warmup %d --no-cpu-baseline --no-sizes --no-whole-call; "
               "means over the %d TIMED launches of the search kernel (launches %d..%d of it)" % (S, W, S, W, W + S - 1),
    "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE / WRITE_SIZE are in KB; gfx950 FETCH_SIZE counts 64 B per 128 B request, so x2 on the read side",
    "avg_launch_ms": mean(timed), "avg_launch_ms_source": R + "_bench_n1e6_nn_grid_timed_dispatches.csv (rocprofv3 --kernel-trace of the same command, same launches)",
    "traffic_bytes_per_launch": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
    "algorithmic_bytes_per_launch": 32000000,
    "valu_wave_instructions_per_launch": c["SQ_INSTS_VALU"],
    "valu_instructions_per_wave": c["SQ_INSTS_VALU"] / c["SQ_WAVES"],
    "valu_busy_quadcycles_per_gui_cycle": c["SQ_ACTIVE_INST_VALU"] / c["GRBM_GUI_ACTIVE"],
    "lanes_active_of_64": c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"],
    "tcp_total_cache_accesses_per_launch": c["TCP_TOTAL_CACHE_ACCESSES_sum"],
    "wait_any_over_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
    "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
    "per_launch_mean": c,
}
json.dump(doc, open(os.path.join(OUT, R + "_bench_n1e6_nn_grid_counters.json"), "w"), indent=1)
print(json.dumps({k: doc[k] for k in ("avg_launch_ms", "traffic_bytes_per_launch", "valu_wave_instructions_per_launch", "valu_instructions_per_wave",
                                      "lanes_active_of_64", "wait_any_over_wave_cycles", "l2_hit_rate")}))

# ---- calibration: the probe kernels (known instruction count, 8 waves per SIMD) and the every-pair kernel under the same counters
# what the busy ratio can read at most: one quad-cycle per vector instruction, 1 024 SIMDs issuing one every four cycles, GRBM_GUI_ACTIVE summed
# over the 8 XCDs -> 1024 / 4 / 8 = 32 (state every `issue` fraction against this, with the probe's own reading beside it)
VALU_BUSY_CEILING = 1024 / 4 / 8
cal = {"valu_busy_ceiling": VALU_BUSY_CEILING, "command": "rocprofv3 --pmc <SQ sets> -- tools/valu_probe (and the nn_bruteforce_kernel launches of the bench command above)",
       "note": "SQ_ACTIVE_INST_VALU counts quad-cycles in which a wave has a vector instruction executing, summed over waves; GRBM_GUI_ACTIVE is "
               "summed over the 8 XCDs.  Their ratio on a kernel that only issues vector instructions at 8 waves per SIMD is what `saturated` "
               "means for that ratio; a kernel's `issue.frac` is its own ratio over that one.", "kernels": {}}
ptr = rows("probe_trace", "kernel_trace.csv")
for name, key in (("valu_probe<0>", "valu_probeILi0"), ("valu_probe<1>", "valu_probeILi1")):
    dur = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ptr if key in r["Kernel_Name"] or name in r["Kernel_Name"])
    e = {"what": "v_fma_f32 x 8 chains" if name.endswith("<0>") else "v_pk_fma_f32 x 8 chains", "launch_ms": dur[-1] * 1e-6 if dur else None}
    for d, ks in (("probe_sq1", SQ1), ("probe_sq2", SQ2)):
        for k in ks:
            v = counter_series(d, k, "valu_probe") if False else [float(r["Counter_Value"]) for r in rows(d, "counter_collection.csv")
                                                                   if (key in r["Kernel_Name"] or name in r["Kernel_Name"]) and r["Counter_Name"] == k]
            e[k] = max(v) if v else None                      # the long launch of the two (the short one is the warm-up)
    if e.get("SQ_INSTS_VALU") and e["launch_ms"]:
        e["wave_instructions_per_s"] = e["SQ_INSTS_VALU"] / (e["launch_ms"] * 1e-3)
        e["valu_busy_quadcycles_per_gui_cycle"] = e["SQ_ACTIVE_INST_VALU"] / e["GRBM_GUI_ACTIVE"]
    cal["kernels"][name] = e
bf = {k: mean(counter_series(d, k, "nn_bruteforce_kernel")) for d, ks in (("sq1", SQ1), ("sq2", SQ2)) for k in ks}
bft = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows("stats", "kernel_trace.csv") if "nn_bruteforce_kernel" in r["Kernel_Name"]]
if bf.get("SQ_INSTS_VALU") and bft:
    bf["launch_ms"] = mean(bft) * 1e-6
    bf["wave_instructions_per_s"] = bf["SQ_INSTS_VALU"] / (bf["launch_ms"] * 1e-3)
    bf["valu_busy_quadcycles_per_gui_cycle"] = bf["SQ_ACTIVE_INST_VALU"] / bf["GRBM_GUI_ACTIVE"]
cal["kernels"]["nn_bruteforce_kernel"] = bf
json.dump(cal, open(os.path.join(OUT, R + "_valu_calibration.json"), "w"), indent=1)
print(json.dumps({k: {kk: v.get(kk) for kk in ("wave_instructions_per_s", "valu_busy_quadcycles_per_gui_cycle")} for k, v in cal["kernels"].items()}))

# ---- CPD E-step kernels of the same run, ONE ENTRY PER WORKLOAD
# The bench command runs the exact E-step kernels on two workloads -- the bunny clouds (cfg 4: 14 904 x 14 904; the hybrid leg's exact iterations are the
# same kernels on the same clouds) and the reference's published size (49 000 x 49 000) -- and round 5 averaged every launch of a kernel in the process
# under the bunny label.  A launch's workload is read off its GRID: the grid is a function of (n, m) alone (cpd_kernels.hip cpd_denominators /
# cpd_contract), so the dispatches of one kernel fall into one grid size per workload; a workload's entry holds only ITS dispatches -- trace rows and
# counter rows alike (both files carry the grid) -- and the selected trace rows are committed as <R>_cpd_estep_timed_dispatches.csv.
CPD_WORKLOADS = (("cpd_bunny_14904", 14904, 14904), ("cpd_synthetic_uniform_n49000", 49000, 49000))


def is_estep_kernel(name):
    return "cpd_denominator_kernel" in name or "cpd_contract" in name


def short(name):
    return name.split("(")[0][-60:]


def exact_form(name):            # the `<.., true>` instantiations are the hybrid mode's truncated E-step: other arithmetic, not this roofline's
    return not short(name).rstrip().endswith("true>")


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if v else 0


def dur(r):
    return int(r["End_Timestamp"]) - int(r["Start_Timestamp"])


def live(rs):
    """Without the NO-OP launches: a registration's iterations are enqueued in batches, and what is still in the queue when the device-side stop rule
    fires returns at the `done` flag (3-4 us, same grid).  They are launches of the kernel, not E-steps: a dispatch counts if it lasts at least half
    the median of its (kernel, grid) group."""
    m = median([dur(r) for r in rs])
    return [r for r in rs if 2 * dur(r) >= m]


def grid_of_trace(r):
    return int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)


trace = [r for r in rows("stats", "kernel_trace.csv") if is_estep_kernel(r["Kernel_Name"]) and exact_form(r["Kernel_Name"])]
trace.sort(key=lambda r: int(r["Start_Timestamp"]))
by_kernel_grid = {}
for r in trace:
    by_kernel_grid.setdefault((r["Kernel_Name"], grid_of_trace(r)), []).append(r)
noop_launches = {k: len(rs) - len(live(rs)) for k, rs in by_kernel_grid.items()}
by_kernel_grid = {k: live(rs) for k, rs in by_kernel_grid.items()}
# per kernel: its grid sizes in ascending order of work; the workloads in ascending order of pairs -- the k-th largest grid that carries more than a
# handful of launches belongs to the k-th largest workload (tiny grids: primitives called on small inputs, e.g. a self-test -- listed under "other")
grids_of = {}
for (kn, g), rs in by_kernel_grid.items():
    grids_of.setdefault(kn, []).append((g, len(rs)))
assign = {}                                                       # (kernel, grid) -> workload
for kn, gl in grids_of.items():
    big = sorted([g for g, cnt in gl if cnt >= 8], reverse=True)[:len(CPD_WORKLOADS)]
    for g, wl in zip(big, sorted(CPD_WORKLOADS, key=lambda w: -w[1] * w[2])):
        assign[(kn, g)] = wl[0]


def counter_rows_of(d, kn, grid):
    rs = [r for r in rows(d, "counter_collection.csv") if r["Kernel_Name"] == kn and int(r["Grid_Size"]) == grid]
    # (one row per counter and dispatch, each with the dispatch's own timestamps under that pass: the same rule, per pass)
    per_dispatch = {}
    for r in rs:
        per_dispatch.setdefault(r["Dispatch_Id"], r)
    keep = {r["Dispatch_Id"] for r in live(list(per_dispatch.values()))}
    return [r for r in rs if r["Dispatch_Id"] in keep]


def counter_mean(d, counter, kn, grid):
    v = [float(r["Counter_Value"]) for r in counter_rows_of(d, kn, grid) if r["Counter_Name"] == counter]
    return mean(v)


cpd = {"kernel": "cpd_estep", "steps": None, "warmup": None, "git_head": head,
       "command": doc["command"].split(";")[0] + " (its cpd_bunny leg: exact P on the bunny clouds and at the reference's published size)",
       "selection": "per kernel, the dispatches whose grid is the workload's (the grid is a function of (n, m) alone), without the no-op launches that return at a "
                    "finished registration's `done` flag (shorter than half the group's median); means over ALL remaining dispatches of the process; the rows themselves: " + R + "_cpd_estep_timed_dispatches.csv",
       "workloads": {}, "other_dispatches": {}}
with open(os.path.join(OUT, R + "_cpd_estep_timed_dispatches.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["workload", "kernel", "Dispatch_Id", "grid_work_items", "Start_Timestamp", "End_Timestamp", "duration_ns"])
    for (kn, g), rs in sorted(by_kernel_grid.items(), key=lambda kv: (assign.get(kv[0], "~"), kv[0][0], kv[0][1])):
        wl = assign.get((kn, g))
        if wl is None:
            cpd["other_dispatches"]["%s grid %d" % (short(kn), g)] = {"launches": len(rs), "mean_ms": mean([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rs])}
            continue
        for r in rs:
            w.writerow([wl, short(kn), r.get("Dispatch_Id", ""), g, r["Start_Timestamp"], r["End_Timestamp"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
for wl, n_, m_ in CPD_WORKLOADS:
    pairs = float(n_) * float(m_)
    entry = {"points": [n_, m_], "pairs_per_launch": pairs, "kernels": {}}
    for (kn, g), rs in by_kernel_grid.items():
        if assign.get((kn, g)) != wl:
            continue
        e = {k: counter_mean(d, k, kn, g) for d, ks in (("sq1", SQ1), ("sq2", SQ2)) for k in ks}
        t = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs]
        e["grid_work_items"] = g
        e["launches"] = len(t)
        e["noop_launches_excluded"] = noop_launches.get((kn, g), 0)      # (returned at the registration's `done` flag: see live())
        e["launch_ms"] = mean(t) * 1e-6
        e["launch_ms_min_max"] = [min(t) * 1e-6, max(t) * 1e-6]
        e["counter_launches"] = len([r for r in counter_rows_of("sq1", kn, g) if r["Counter_Name"] == "SQ_INSTS_VALU"])
        if e.get("SQ_INSTS_VALU"):
            e["valu_instructions_per_pair"] = e["SQ_INSTS_VALU"] * 64.0 / pairs
        if e.get("SQ_ACTIVE_INST_VALU") and e.get("GRBM_GUI_ACTIVE"):
            e["valu_busy_quadcycles_per_gui_cycle"] = e["SQ_ACTIVE_INST_VALU"] / e["GRBM_GUI_ACTIVE"]
        # the matrix pipe: its own pass (SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which a SIMD's matrix pipe is busy, summed over the SIMDs;
        # GRBM_GUI_ACTIVE of that pass is summed over the 8 XCDs)
        mm = {k: counter_mean("mfma", k, kn, g) for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_MFMA", "GRBM_GUI_ACTIVE")}
        if "mfma" in kn and mm.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None and mm.get("GRBM_GUI_ACTIVE"):
            cycles = mm["GRBM_GUI_ACTIVE"] / 8.0
            e["mfma"] = {"SQ_VALU_MFMA_BUSY_CYCLES": mm["SQ_VALU_MFMA_BUSY_CYCLES"], "SQ_INSTS_VALU_MFMA_MOPS_F32": mm["SQ_INSTS_VALU_MFMA_MOPS_F32"],
                         "SQ_INSTS_MFMA": mm.get("SQ_INSTS_MFMA"), "gpu_cycles": cycles,
                         "mfma_util": mm["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0),
                         "mfma_flops_per_launch": (mm["SQ_INSTS_VALU_MFMA_MOPS_F32"] or 0.0) * 512.0,
                         "contraction_flops_per_launch": pairs * 8.0,
                         "share_of_contraction_on_matrix_pipe": (mm["SQ_INSTS_VALU_MFMA_MOPS_F32"] or 0.0) * 512.0 / (pairs * 8.0),
                         "note": "mfma_util = busy cycles / (GPU cycles x 1 024 SIMDs); MOPS_F32 counts 512 flops each; the contraction P~ [X|1] is "
                                 "pairs x 4 multiply-adds = pairs x 8 flops, all of it on the matrix pipe when the MFMA form runs"}
        entry["kernels"][short(kn)] = e
    cpd["workloads"][wl] = entry
# (bench.py's committed_profile() looks a file up by its top-level workload / kernel: the bunny entry is also the file's own)
cpd["workload"] = CPD_WORKLOADS[0][0]
cpd["pairs_per_launch"] = cpd["workloads"][CPD_WORKLOADS[0][0]]["pairs_per_launch"]
cpd["kernels"] = cpd["workloads"][CPD_WORKLOADS[0][0]]["kernels"]
json.dump(cpd, open(os.path.join(OUT, R + "_cpd_estep_counters.json"), "w"), indent=1)
print(json.dumps({wl: {k: {kk: v.get(kk) for kk in ("launches", "launch_ms", "valu_instructions_per_pair", "valu_busy_quadcycles_per_gui_cycle")} for k, v in e["kernels"].items()}
                  for wl, e in cpd["workloads"].items()}))
print(json.dumps({"other": cpd["other_dispatches"]}))
