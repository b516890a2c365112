"""Developer tool (round 6): one configuration of the reference's convergence set (20 000 points of bird.obj, spread 10) capped at 1 .. 60 iterations -- the CPU
restatement against the device in MI_SUM_CPU_SEQUENTIAL (fast and IEEE K3) and with the default fp64 sums: where a trajectory parts from cpu-slam's.
    python tools/conv_retrace.py SEED ROTATION TRANSLATION        (e.g. 1208 0.6 30: the rank-1 first iteration of DESIGN section 2)"""
import os, sys, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np
from __graft_entry__ import load_package
from oracle import oraclebind as O
from oracle import refbind as ref
capi = load_package().capi
seed, rot, trans, size = int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3]), 20000
z = np.load(os.path.join(ROOT, "tests", "golden", "noise_meshes.npz"))
raw = np.ascontiguousarray(z["bird_v"][z["bird_f"].astype(np.int64)])
before, after, Rk, tk = ref.clouds_from_config_random(raw, None, seed, rot, trans, resize_before=size, resize_after=size, spread=10.0)
def frob(R, t, R2, t2): return float(np.sqrt(((np.asarray(R, np.float64) - R2) ** 2).sum() + ((np.asarray(t, np.float64) - t2) ** 2).sum()))
os.environ["MISLAM_SVD_IEEE"] = "1"
ieee = capi.Context(0)
del os.environ["MISLAM_SVD_IEEE"]
with capi.Context(0) as ctx:
    for cap in list(range(1, 16)) + [20, 30, 47, 60]:
        Ro, to, ito, eo = O.icp(before, after, 1e-3, 10000.0, cap)[:4]
        p = capi.icp_params(eps=1e-3, max_iterations=cap, max_distance_squared=10000.0, sum_mode=capi.SUM_CPU_SEQUENTIAL)
        R, t, it, err = ctx.icp_register(before, after, p)[:4]
        Ri, ti, iti, erri = ieee.icp_register(before, after, p)[:4]
        pd = capi.icp_params(eps=1e-3, max_iterations=cap, max_distance_squared=10000.0)
        Rd, td, itd, errd = ctx.icp_register(before, after, pd)[:4]
        print("cap %2d: oracle it %d err %.5g | device(seq sums) it %d err %.5g d %.3e | IEEE K3 d %.3e | fp64 sums it %d err %.5g d %.3e" % (cap, ito, eo, it, err, frob(R, t, Ro, to), frob(Ri, ti, Ro, to), itd, errd, frob(Rd, td, Ro, to)), flush=True)
ieee.close()
