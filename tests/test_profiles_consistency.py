"""The committed profile summaries of the latest round agree with each other (no GPU needed): the search kernel's average launch in the counters file
is the mean of the committed timed-dispatch rows, the algorithmic bytes are 20 N + 12 M, and bench.py's calibrated `issue` fraction can be
recomputed from the two committed files."""
import csv
import json
import os

import pytest

from conftest import ROOT

PROF = os.path.join(ROOT, "profiles")


def _load():
    import glob
    tag = sorted(os.path.basename(f).split("_")[0] for f in glob.glob(os.path.join(PROF, "r*_bench_n1e6_nn_grid_counters.json")))[-1]   # the latest round's
    c = json.load(open(os.path.join(PROF, tag + "_bench_n1e6_nn_grid_counters.json")))
    rows = list(csv.DictReader(open(os.path.join(PROF, tag + "_bench_n1e6_nn_grid_timed_dispatches.csv"))))
    cal = json.load(open(os.path.join(PROF, tag + "_valu_calibration.json")))
    return c, rows, cal


def test_average_launch_is_the_mean_of_the_committed_timed_dispatches():
    c, rows, _ = _load()
    assert len(rows) == c["steps"] == 20
    mean_ms = sum(int(r["duration_ns"]) for r in rows) / len(rows) * 1e-6
    assert abs(mean_ms - c["avg_launch_ms"]) < 1e-9 * max(1.0, mean_ms) + 1e-9
    assert all("nn_grid_kernel" in r["Kernel_Name"] for r in rows)
    starts = [int(r["Start_Timestamp"]) for r in rows]
    assert starts == sorted(starts)


def test_algorithmic_bytes_and_traffic():
    c, _, _ = _load()
    n = m = 1000000
    assert c["algorithmic_bytes_per_launch"] == 20 * n + 12 * m           # 20 B per moving point + 12 B per fixed point
    assert c["traffic_bytes_per_launch"] > c["algorithmic_bytes_per_launch"]   # measured HBM bytes (FETCH + WRITE, corrected) per launch
    achieved = c["algorithmic_bytes_per_launch"] / (c["avg_launch_ms"] * 1e-3) / 1e9
    assert 100.0 < achieved < 8000.0


def test_issue_fraction_is_the_calibrated_busy_ratio():
    c, _, cal = _load()
    probe = cal["kernels"]["valu_probe<0>"]["valu_busy_quadcycles_per_gui_cycle"]
    brute = cal["kernels"]["nn_bruteforce_kernel"]["valu_busy_quadcycles_per_gui_cycle"]
    assert 25.0 < probe <= 32.0 and 25.0 < brute <= 32.0                   # a saturated vector pipe reads close to the counter's ceiling
    frac = c["valu_busy_quadcycles_per_gui_cycle"] / probe
    assert 0.5 < frac < 1.0
    m = c["per_launch_mean"]
    assert abs(c["valu_busy_quadcycles_per_gui_cycle"] - m["SQ_ACTIVE_INST_VALU"] / m["GRBM_GUI_ACTIVE"]) < 1e-6 * c["valu_busy_quadcycles_per_gui_cycle"]


def test_profile_speaks_for_the_committed_search_kernel():
    import sys
    sys.path.insert(0, ROOT)
    from bench import search_source_hash
    c, _, _ = _load()
    # a stale profile fails the suite: the counters bench.py attaches to its roofline must have been taken on the code it runs
    assert c.get("source_hash") == search_source_hash(), ("the committed counter profile was taken on other search-kernel code: run "
                                                          "tools/gpu_profiles.sh again and commit its summaries with the kernel change")


# ---- the CPD leg: one profile entry per workload, each recomputable from the committed rows --------------------------------
def _load_cpd():
    import glob
    files = sorted(glob.glob(os.path.join(PROF, "r*_cpd_estep_counters.json")))
    c = json.load(open(files[-1]))
    tag = os.path.basename(files[-1]).split("_")[0]
    if "workloads" not in c:
        pytest.skip("the latest CPD profile predates the per-workload form")
    rows = list(csv.DictReader(open(os.path.join(PROF, tag + "_cpd_estep_timed_dispatches.csv"))))
    return tag, c, rows


def _bench_lines(tag):
    path = os.path.join(PROF, tag + "_bench_default.log")
    if not os.path.exists(path):
        pytest.skip("no committed default bench log for " + tag)
    return [json.loads(ln) for ln in open(path) if ln.lstrip().startswith("{") and '"metric"' in ln]


CPD_WORKLOADS = {"cpd_bunny_14904": 14904.0 * 14904.0, "cpd_synthetic_uniform_n49000": 49000.0 * 49000.0}
# vector instructions per pair of the two exact E-step kernels (denominators, contraction) -- the per-workload profile must say the same
DESIGN_VALU_PER_PAIR = {"cpd_denominator_kernel": 9.5, "cpd_contract_mfma_kernel": 9.9}


def test_cpd_profile_has_one_entry_per_workload_recomputable_from_the_committed_rows():
    tag, c, rows = _load_cpd()
    assert set(c["workloads"]) == set(CPD_WORKLOADS)
    for wl, pairs in CPD_WORKLOADS.items():
        e = c["workloads"][wl]
        assert e["pairs_per_launch"] == pairs
        assert len(e["kernels"]) == 2                                       # the exact mode's two kernels, nothing of the hybrid mode's truncated ones
        for kn, k in e["kernels"].items():
            mine = [r for r in rows if r["workload"] == wl and r["kernel"].strip() == kn.strip()]
            assert len(mine) == k["launches"] > 0
            mean_ms = sum(int(r["duration_ns"]) for r in mine) / len(mine) * 1e-6
            assert abs(mean_ms - k["launch_ms"]) < 1e-9 + 1e-9 * mean_ms
            assert len({r["grid_work_items"] for r in mine}) == 1           # one grid per (kernel, workload): what the selection rests on
            # no no-op launch among them (a launch that returns at a finished registration's `done` flag lasts 3-4 us)
            assert min(int(r["duration_ns"]) for r in mine) * 2 >= sorted(int(r["duration_ns"]) for r in mine)[len(mine) // 2]
            assert k["counter_launches"] == k["launches"]                   # the counter passes saw the same dispatches
            name = "cpd_denominator_kernel" if "denominator" in kn else "cpd_contract_mfma_kernel"
            assert abs(k["valu_instructions_per_pair"] - DESIGN_VALU_PER_PAIR[name]) <= 0.1 * DESIGN_VALU_PER_PAIR[name], (wl, kn, k["valu_instructions_per_pair"])
            assert abs(k["SQ_INSTS_VALU"] * 64.0 / pairs - k["valu_instructions_per_pair"]) < 1e-9 * k["valu_instructions_per_pair"]


def test_cpd_profile_matrix_pipe_carries_the_contraction_once():
    _, c, _ = _load_cpd()
    for wl, e in c["workloads"].items():
        mf = [k["mfma"] for kn, k in e["kernels"].items() if "mfma" in kn]
        assert len(mf) == 1
        share = mf[0]["share_of_contraction_on_matrix_pipe"]
        assert 0.9 <= share <= 1.1, (wl, share)                             # round 5's mixed average read 2.23
        assert abs(mf[0]["contraction_flops_per_launch"] - 8.0 * e["pairs_per_launch"]) < 1.0
        assert 0.05 < mf[0]["mfma_util"] < 0.5


def test_cpd_profile_agrees_with_the_bench_lines_live_kernel_times():
    tag, c, _ = _load_cpd()
    lines = _bench_lines(tag)
    assert lines
    for d in lines:
        live = d["cpd_bunny"]["exact"]["kernels_ms_per_launch"]
        prof = c["workloads"]["cpd_bunny_14904"]["kernels"]
        for live_name, key in (("cpd_denom", "denominator"), ("cpd_contract", "contract")):
            p = [k["launch_ms"] for kn, k in prof.items() if key in kn][0]
            assert abs(p - live[live_name]) <= 0.1 * live[live_name], (live_name, p, live[live_name])
        # ... and the roofline in the line is the bunny entry's, recomputable by hand: time-weighted busy ratio over the counter's ceiling
        roof = d["cpd_bunny"]["exact"]["roofline"]
        ks = list(prof.values())
        busy = sum(k["valu_busy_quadcycles_per_gui_cycle"] * k["launch_ms"] for k in ks) / sum(k["launch_ms"] for k in ks)
        assert roof["workload"] == "cpd_bunny_14904" and abs(roof["achieved"] - busy) < 1e-6 * busy and abs(roof["frac"] - busy / 32.0) < 1e-6
        pub = d["cpd_bunny"]["published_size"]["exact"]
        assert pub["roofline"]["workload"] == "cpd_synthetic_uniform_n49000"
        prof49 = c["workloads"]["cpd_synthetic_uniform_n49000"]["kernels"]
        for live_name, key in (("cpd_denom", "denominator"), ("cpd_contract", "contract")):
            p = [k["launch_ms"] for kn, k in prof49.items() if key in kn][0]
            assert abs(p - pub["kernels_ms_per_launch"][live_name]) <= 0.1 * pub["kernels_ms_per_launch"][live_name]


# ---- the search kernel's measured instruction budget speaks for the committed kernel ---------------------------------------
def test_search_budget_is_recomputable_and_current(tmp_path):
    """profiles/r06_search_budget.md = static instruction counts of the warm kernel's ISA per phase (profiles/r06_search_budget_spec.json) x the counting
    build's trip counts (profiles/r06_phase_counts.json), against SQ_INSTS_VALU / SQ_WAVES of the same launches.  Recomputed here from the sources (hipcc
    cross-compiles without a GPU): a kernel change that moves the block layout makes the composer refuse the spec, and the rows must still add up to the
    counter within the model's stated error."""
    import shutil
    import subprocess
    import sys
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "isa_dump.sh"), "nn_grid"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    isa = r.stdout.strip().splitlines()[-1]
    c, _, _ = _load()
    per_wave = c["valu_instructions_per_wave"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_budget.py"), isa, "x", "--compose", os.path.join(PROF, "r06_search_budget_spec.json"),
                        os.path.join(PROF, "r06_phase_counts.json"), "%.1f" % per_wave], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    import re
    m = re.search(r"rows sum to (\d+) = ([0-9.]+) of it", r.stdout)
    assert m, r.stdout[-800:]
    assert 0.95 <= float(m.group(2)) <= 1.15            # (conditional code inside a block is counted as always run: the model errs high)
    counts = json.load(open(os.path.join(PROF, "r06_phase_counts.json")))
    ph = counts["phases"]
    L = counts["launches"]
    assert ph["waves"] == L * ((counts["points"] + 63) // 64)                      # one wave per 64 moving points
    assert ph["scan_waves"] + ph["walk_only_waves"] == ph["waves"] and ph["block_batches"] == ph["scan_waves"]
    assert ph["block_dealt"] <= ph["block_batches"] and ph["rest_dealt"] <= ph["rest_rounds"] and ph["walk_leaf_offers"] <= ph["walk_leaf_hits"] <= counts["stats"]["walk_leaves"]
