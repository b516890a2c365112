"""Developer A/B (round 6): whole registrations and loads alone on host buffers, for MISLAM_PIN=0 / 1 (profiles/r06_upload_paths.log).
    python tools/upload_paths_ab.py [points]"""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np
from __graft_entry__ import load_package
from bench import synth_cloud
capi = load_package().capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
before, after = synth_cloud(np, n)
p = capi.icp_params(cuda_slam=True, max_iterations=50, eps=1e-3, max_distance_squared=10000.0)
with capi.Context(0) as ctx:
    ctx.icp_register(before, after, p)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); ctx.icp_register(before, after, p); ts.append((time.perf_counter() - t0) * 1e3)
    tl = []
    for _ in range(20):
        t0 = time.perf_counter(); ctx.icp_load(before, after, p); ctx.synchronize(); tl.append((time.perf_counter() - t0) * 1e3)
print(n, "PIN", os.environ.get("MISLAM_PIN", "0"), "whole call ms: min %.3f median %.3f max %.3f | load alone: min %.3f median %.3f max %.3f" % (min(ts), sorted(ts)[10], max(ts), min(tl), sorted(tl)[10], max(tl)))
