#!/usr/bin/env python3
"""Developer tool: DESIGN.md's table of the reference's CPD evaluation corpus (doc/noise/configs) from the fixture (tests/golden/noise_configs.json,
oracle/make_golden_noise.py) and the GPU suite's measurements (gpurun_out/noise_corpus_results.json, written by tests/test_gpu_noise_corpus.py).
    python tools/noise_table.py [results.json] > profiles/r05_noise_corpus.md"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    res_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "noise_corpus_results.json")
    res = {r["config"]: r for r in json.load(open(res_path))["configs"]}
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "noise_configs.json")))
    print("| config | before -> after | points | const-scale | cpu-slam: iterations, final sigma^2 | cpu-slam vs ITSELF, points reordered (2 runs): iterations, |d(sR\\|t)|_F | restatement (fp64 M-step sums) vs cpu-slam | MI355X vs restatement | MI355X vs cpu-slam | MI355X iterations |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for c in doc["configs"]:
        o, r = c["options"], res.get(c["config"], {})
        ref = c["cpu_slam"]
        if r.get("diverged") or ref["error"] != ref["error"]:
            print("| %d | %s -> %s | %d x %d | %s | %d, NaN (sigma^2_0 saturates at %.3g: Np = 0) | %s | NaN alike | NaN alike (R = identity, t = NaN) | same entries finite, same values | %s |"
                  % (c["config"], o["before"][:-4], o["after"][:-4], c["n_before"], c["n_after"], "yes" if o["cpd_const_scale"] else "no", ref["iterations"], c["sigma2_init"],
                     ", ".join("%d, NaN" % q["iterations"] for q in c["cpu_slam_reordered"]), r.get("iterations", "-")))
            continue
        fmt = lambda v: "-" if v is None else "%.1e" % v
        print("| %d | %s -> %s | %d x %d | %s | %d, %.3g | %s | %d it, %s | %s | %s | %s%s |"
              % (c["config"], o["before"][:-4], o["after"][:-4], c["n_before"], c["n_after"], "yes" if o["cpd_const_scale"] else "no", ref["iterations"], ref["error"],
                 "; ".join("%d it, %.1e" % (q["iterations"], q["distance"]) for q in c["cpu_slam_reordered"]), c["oracle"]["iterations"], fmt(c["oracle_vs_cpu_slam"]),
                 fmt(r.get("vs_oracle")), fmt(r.get("vs_cpu_slam")), r.get("iterations", "-"), " **(asserted < 1e-4 vs restatement)**" if r.get("reproducible") else ""))
    print()
    print("Skipped (their .obj files are missing blobs of the reference checkout): configs %s." % ", ".join(str(s["config"]) for s in doc["skipped"]))


if __name__ == "__main__":
    main()
