#!/bin/bash
# Round-2 profiles of the driver's default bench command (python bench.py --steps 20 --warmup 5): rocprofv3 kernel-trace stats and
# the --pmc passes (one per counter set, never mixed with tracing), reduced to the summaries bench.py reads from profiles/.
#   gpurun -- 'bash tools/gpu_profiles_r02.sh'   ->   gpurun_out/r02_*  (copy the summaries into profiles/)
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
S=${STEPS:-20}; W=${WARMUP:-5}
B="python3 bench.py --steps $S --warmup $W --no-cpu-baseline --no-sizes --brute-ref-steps 0"
run() { d=gpurun_out/r02_$1; shift; rm -rf $d; timeout -k 10 400 rocprofv3 "$@" -d $d --output-format csv -- $B > $d.log 2>&1 || { tail -5 $d.log; exit 1; }; }
run stats --kernel-trace --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq1 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
run sq2 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
run tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
python3 - $S $W <<'PY'
import csv, glob, json, sys, collections, shutil
S, W = int(sys.argv[1]), int(sys.argv[2])
def rows(d, pat):
    out = []
    for f in glob.glob("gpurun_out/r02_%s/**/*%s" % (d, pat), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
# --- kernel trace: per-kernel stats file + the timed launches of the search kernel
for f in glob.glob("gpurun_out/r02_stats/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, "gpurun_out/r02_bench_n1e6_kernel_stats.csv")
tr = [r for r in rows("stats", "kernel_trace.csv") if "nn_grid_kernel" in r["Kernel_Name"]]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in tr]
timed = dur[W:W + S]
def timed_mean(d, counter, kernel="nn_grid_kernel"):
    rs = [r for r in rows(d, "counter_collection.csv") if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    rs.sort(key=lambda r: int(r["Dispatch_Id"]))
    v = [float(r["Counter_Value"]) for r in rs][W:W + S]
    return sum(v) / len(v) if v else None
c = {k: timed_mean(d, k) for d, ks in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
     ("sq1", ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVES", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"]),
     ("sq2", ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_SCA", "GRBM_GUI_ACTIVE"]),
     ("tcp", ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"])) for k in ks}
doc = {
    "workload": "icp_synthetic_uniform_n1000000", "kernel": "nn_grid_kernel", "steps": S, "warmup": W,
    "command": "rocprofv3 --pmc <one pass per counter set> -- python3 bench.py --steps %d --warmup %d --no-cpu-baseline --no-sizes --brute-ref-steps 0; "
               "means over the %d TIMED launches of the search kernel (dispatches %d..%d of it)" % (S, W, S, W, W + S - 1),
    "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE / WRITE_SIZE are in KB; gfx950 FETCH_SIZE counts 64 B per 128 B request, so x2 on the read side",
    "avg_launch_ms": sum(timed) / len(timed), "avg_launch_ms_source": "rocprofv3 --kernel-trace of the same command, same launches",
    "traffic_bytes_per_launch": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
    "algorithmic_bytes_per_launch": 32000000,
    "valu_wave_instructions_per_launch": c["SQ_INSTS_VALU"],
    "lanes_active_of_64": c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"],
    "tcp_total_cache_accesses_per_launch": c["TCP_TOTAL_CACHE_ACCESSES_sum"],
    "wait_any_over_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
    "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
    "per_launch_mean": c,
}
json.dump(doc, open("gpurun_out/r02_bench_n1e6_nn_grid_counters.json", "w"), indent=1)
print(json.dumps({k: doc[k] for k in ("avg_launch_ms", "traffic_bytes_per_launch", "valu_wave_instructions_per_launch", "lanes_active_of_64",
                                      "tcp_total_cache_accesses_per_launch", "wait_any_over_wave_cycles", "l2_hit_rate")}))
# --- CPD E-step: vector lane operations per pair, both passes (exact P kernels of the bunny leg)
pairs = 14904.0 * 14904.0
den = timed_mean.__globals__["rows"]("sq1", "counter_collection.csv")
def mean_k(kernel_sub):
    v = [float(r["Counter_Value"]) for r in den if kernel_sub in r["Kernel_Name"] and r["Counter_Name"] == "SQ_INSTS_VALU"]
    return sum(v) / len(v), len(v)
d_i, d_n = mean_k("cpd_denominator_kernel<4, false>")
k_i, k_n = mean_k("cpd_contract_mfma_kernel<1, false>")
cpd = {"workload": "cpd_bunny_14904", "kernel": "cpd_estep", "steps": None, "warmup": None,
       "command": doc["command"].split(";")[0] + " (its cpd_bunny leg: exact P, 27 EM iterations, twice)",
       "SQ_INSTS_VALU_per_launch": {"cpd_denominator_kernel<4,false>": d_i, "cpd_contract_mfma_kernel<1,false>": k_i}, "launches": [d_n, k_n],
       "pairs_per_launch": pairs, "valu_lane_ops_per_pair_both_passes": (d_i + k_i) * 64.0 / pairs}
json.dump(cpd, open("gpurun_out/r02_cpd_estep_counters.json", "w"), indent=1)
print(json.dumps({"cpd_valu_lane_ops_per_pair": cpd["valu_lane_ops_per_pair_both_passes"]}))
PY
find gpurun_out/r02_stats gpurun_out/r02_fetch gpurun_out/r02_write gpurun_out/r02_sq1 gpurun_out/r02_sq2 gpurun_out/r02_tcp -type f -delete 2>/dev/null
exit 0
