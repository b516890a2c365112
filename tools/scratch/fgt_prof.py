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
