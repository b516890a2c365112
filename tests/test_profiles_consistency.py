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
