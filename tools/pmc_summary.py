#!/usr/bin/env python3
"""Developer tool: turns the two rocprofv3 --pmc passes of tools/gpu_suite_and_profiles.sh (FETCH_SIZE and WRITE_SIZE, separate
runs of the same bench.py command) into the per-kernel HBM-traffic summary bench.py reads (profiles/*_hbm_counters.json).

    python tools/pmc_summary.py gpurun_out/prof_fetch gpurun_out/prof_write profiles/r01_bench_n1e6_hbm_counters.json

Units and correction as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are in KB; on gfx950 FETCH_SIZE counts
64 B per 128 B request, hence x2 on the read side."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(directory, counter):
    sums, counts = defaultdict(float), defaultdict(int)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                sums[name] += float(row["Counter_Value"])
                counts[name] += 1
    return {k: (sums[k] / counts[k], counts[k]) for k in sums}


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fetch, write = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    n = 1000000
    doc = {
        "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 50 --warmup 2 --no-cpu-baseline --brute-ref-steps 1",
        "workload": "icp_synthetic_uniform_n%d" % n,
        "correction": "MI355X_MICROARCH.md HBM section: counters are in KB; gfx950 FETCH_SIZE counts 64 B per 128 B request, so x2 on "
                      "the read side (calibrated there for 16 B/lane streams; other widths uncalibrated)",
    }
    tree_note = ("exact box-hierarchy search (default path): compact hierarchy (12 MB leaf coordinates + 3 MB sibling "
                 "records) + 12 MB sources + 8 MB keys read, 8 MB keys written; the per-lane walks are served by L2/MALL")
    notes = {
        "nn_tree_lane_dynamic_kernel": tree_note,
        "nn_tree_lane_compact_kernel": tree_note,
        "nn_bruteforce_kernel": "every-pair search, 8 XCD-pinned target chunks: each XCD reads all sources once (8 x 12 MB) and posts one "
                                "8-byte atomicMin per source and chunk",
    }
    for short, note in notes.items():
        keys = [k for k in fetch if short in k]
        if not keys:
            continue
        k = keys[0]
        f_kb, w_kb = fetch[k][0], write.get(k, (0.0, 0))[0]
        doc[short] = {"FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb, "traffic_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0,
                      "algorithmic_bytes_per_launch": 20 * n + 12 * n, "note": note}
    doc["all_kernels"] = {k: {"FETCH_SIZE_KB_mean_per_launch": fetch[k][0], "FETCH_SIZE_launches": fetch[k][1],
                              "WRITE_SIZE_KB_mean_per_launch": write.get(k, (0.0, 0))[0], "WRITE_SIZE_launches": write.get(k, (0.0, 0))[1]}
                          for k in sorted(fetch)}
    json.dump(doc, open(out, "w"), indent=1)
    print("wrote", out, {k: round(v["traffic_bytes_per_launch"] / 1e6, 1) for k, v in doc.items() if isinstance(v, dict) and "traffic_bytes_per_launch" in v})


if __name__ == "__main__":
    main()
