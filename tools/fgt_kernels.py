#!/usr/bin/env python3
"""Developer tool: four registrations of the bunny clouds in the reference's approximation-type "full" (17 EM iterations, every E-step a Fast Gauss
Transform) -- the command to put under `rocprofv3 --kernel-trace --stats` for the per-kernel breakdown of an FGT iteration (K9):
    rocprofv3 --kernel-trace --stats -d gpurun_out/fgt_kernels --output-format csv -- python3 tools/fgt_kernels.py"""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from __graft_entry__ import load_package
capi = load_package().capi
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
z = np.load(os.path.join(ROOT, "tests", "golden", "bunny_clouds.npz"))
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "bunny_cpd.json")))
ctx = capi.Context(0)
p = capi.cpd_params(max_iterations=17, const_scale=0, sigma2_init=gold["sigma2_init"], approximation=capi.CPD_APPROX_FULL)
for _ in range(4):
    out = ctx.cpd_register(z["before"], z["after"], p)
print(out[3])
