"""Developer tool (round 6): the FIRST Kabsch solve of one configuration of the reference's convergence set -- how many distinct fixed points the moving cloud is
matched to, the cross-covariance and its singular values, the device's R against the restatement's (seed 1208, two distinct targets, rank 1).
    python tools/conv_first_solve.py SEED ROTATION TRANSLATION"""
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
np.set_printoptions(precision=6, suppress=False, linewidth=200)
with capi.Context(0) as ctx:
    idx, d2 = ctx.nn_search(before, after)
    io, do = O.nn_search(before, after)
    print("nn equal:", np.array_equal(idx, io), "distinct targets:", len(np.unique(idx)), "d2 max", d2.max(), "kept (<1e4):", int((d2 < 1e4).sum()))
    keep = (d2 < 10000.0).astype(np.uint8)
    R, t = ctx.kabsch(before, after, idx, keep)[:2]
    mom = ctx.cross_moments(before, after, idx, keep)
    bsel, asel = before[keep.astype(bool)], after[idx[keep.astype(bool)]]
    Ro, to = O.least_squares_svd(bsel, asel)[:2]
    print("device R\n", np.asarray(R), "\n t", np.asarray(t)); print("oracle R\n", np.asarray(Ro), "\n t", np.asarray(to))
    # H in fp64
    cb, ca = bsel.astype(np.float64).mean(0), asel.astype(np.float64).mean(0)
    H = (asel.astype(np.float64) - ca).T @ (bsel.astype(np.float64) - cb)
    print("H (a b^T centered, fp64)\n", H, "\n singular values", np.linalg.svd(H)[1])
    U, S, Vt = np.linalg.svd(H.astype(np.float32)); print("numpy float32 svd: S", S, "det(U Vt)", np.linalg.det(U @ Vt))
    u, s, v = O.jacobi_svd3(H.astype(np.float32)); print("oracle jacobi: s", s, "\n u\n", u, "\n v\n", v)
