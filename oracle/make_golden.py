"""TEST INFRASTRUCTURE ONLY.  Regenerates tests/golden/ by RUNNING THE REFERENCE's own cpu-slam code
(oracle/_ref/libref_cpuslam.so, built by oracle/Makefile from /root/reference) -- run in the build container only:

    make -C oracle ref && python -m oracle.make_golden

The fixtures are data (inputs + the reference's outputs), never reference source text:

  bunny_clouds.npz      post-GetCloudsFromConfig clouds for config/default.json + "random-seed": 666
                        (before, after: float32 [14904,3]); raw points from data/bunny.obj via oracle/objio.py
  bunny_icp.json        cfg 1: final (R, t, iterations, error) of BasicICP::GetBasicICPTransformationMatrix and the
                        (R, t, error) it returns when capped at k = 1, 2, 3, 5, 10, 20 iterations
  bunny_icp_iter0.npz   first-iteration correspondences (kept source / target indices) and the first Kabsch solve
  bunny_cpd.json        cfg 4: sigma^2 init, constant c, first E-step scalars, first M-step, final result of
                        CoherentPointDrift::GetRigidCPDTransformationMatrix (approximation none; const-scale false and true)
  bunny_cpd_estep0.npz  first E-step arrays P1, Pt1, PX
  synth2k_*.npz/json    a 2 000-point synthetic cloud (SURVEY 8d recipe) through the same ICP entry point
"""
import json
import os
import sys

import numpy as np

from . import objio, refbind as R

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = os.environ.get("REF", "/root/reference")

DEFAULT_ROT = [0.36, 0.47, -0.8, -0.8, 0.6, 0, 0.48, 0.64, 0.6]   # config/default.json:8-12 (row-major)
DEFAULT_T = [1.0, 1.0, 1.0]                                       # config/default.json:7


def synth_cloud(n, seed=666):
    """SURVEY 8d synthetic recipe: uniform [-5,5]^3, after = R(0.2 rad about (1,2,3)/sqrt14) * before_perm + 10*(1,1,1)/sqrt3."""
    rng = np.random.default_rng(seed)
    before = rng.uniform(-5.0, 5.0, size=(n, 3)).astype(np.float32)
    axis = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
    ang = 0.2
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    Rm = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    t = 10.0 * np.ones(3) / np.sqrt(3.0)
    perm = rng.permutation(n)
    after = (before[perm].astype(np.float64) @ Rm.T + t).astype(np.float32)
    return before, after, Rm.astype(np.float32), t.astype(np.float32)


def main():
    os.makedirs(GOLD, exist_ok=True)
    raw = objio.load_obj_points(os.path.join(REF, "data", "bunny.obj"))
    assert raw.shape == (14904, 3)
    before, after = R.clouds_from_config(raw, 10.0, 666, DEFAULT_ROT, DEFAULT_T)
    np.savez_compressed(os.path.join(GOLD, "bunny_clouds.npz"), before=before, after=after)

    # ---- cfg 1: ICP, config/default.json (max-distance-squared 400, max-iterations 50, eps default 1e-3, parallel)
    icp_params = dict(eps=1e-3, max_distance_squared=400.0, max_iterations=50, parallel=True)
    Rf, tf, it, err = R.icp(before, after, **icp_params)
    capped = {}
    for k in (1, 2, 3, 5, 10, 20):
        Rk, tk, itk, ek = R.icp(before, after, 1e-3, 400.0, k, True)
        capped[str(k)] = dict(R=Rk.tolist(), t=tk.tolist(), iterations=itk, error=ek)
    json.dump(dict(params=icp_params, R=Rf.tolist(), t=tf.tolist(), iterations=it, error=err, capped=capped,
                   source="BasicICP::GetBasicICPTransformationMatrix via oracle/_ref"),
              open(os.path.join(GOLD, "bunny_icp.json"), "w"), indent=1)
    ib, ia = R.corresponding_points(before, after, 400.0, True)
    R0, t0 = R.least_squares_svd(before[ib], after[ia])
    np.savez_compressed(os.path.join(GOLD, "bunny_icp_iter0.npz"), idx_before=ib, idx_after=ia, R0=R0, t0=t0)

    # ---- cfg 4: CPD exact P; parser defaults cpd-weight .3, cpd-const-scale false, tolerance 1e-3, eps 1e-3
    s2 = R.cpd_sigma_squared(before, after)
    from . import oraclebind as O
    const = O.cpd_constant(s2, 0.3, len(before), len(after))     # formula check against the E-step below is in tests
    p1, pt1, px, L = R.cpd_estep(before, after, const, s2)
    np.savez_compressed(os.path.join(GOLD, "bunny_cpd_estep0.npz"), p1=p1, pt1=pt1, px=px)
    m_false = R.cpd_mstep(before, after, p1, pt1, px, False)
    m_true = R.cpd_mstep(before, after, p1, pt1, px, True)
    out = dict(sigma2_init=s2, constant=const, L0=L,
               mstep0_scale_free=dict(R=m_false[0].tolist(), t=m_false[1].tolist(), scale=m_false[2], sigma2=m_false[3]),
               mstep0_const_scale=dict(R=m_true[0].tolist(), t=m_true[1].tolist(), scale=m_true[2], sigma2=m_true[3]),
               params=dict(eps=1e-3, weight=0.3, max_iterations=50, tolerance=1e-3, fgt=0))
    for name, cs in (("final_scale_free", False), ("final_const_scale", True)):
        Rc, tc, itc, ec = R.cpd(before, after, 1e-3, 0.3, cs, 50, 1e-3, 0)
        out[name] = dict(sR=Rc.tolist(), t=tc.tolist(), iterations=itc, error=ec)
        print(name, itc, ec, tc, file=sys.stderr)
    json.dump(out, open(os.path.join(GOLD, "bunny_cpd.json"), "w"), indent=1)

    # ---- synthetic 2 000-point cloud through the same entry points
    sb, sa, Rm, tm = synth_cloud(2000)
    np.savez_compressed(os.path.join(GOLD, "synth2k_clouds.npz"), before=sb, after=sa, R_true=Rm, t_true=tm)
    Rs, ts, its, es = R.icp(sb, sa, 1e-3, 1000.0, 60, True)
    sib, sia = R.corresponding_points(sb, sa, 1000.0, True)
    np.savez_compressed(os.path.join(GOLD, "synth2k_icp_iter0.npz"), idx_before=sib, idx_after=sia)
    json.dump(dict(params=dict(eps=1e-3, max_distance_squared=1000.0, max_iterations=60, parallel=True),
                   R=Rs.tolist(), t=ts.tolist(), iterations=its, error=es),
              open(os.path.join(GOLD, "synth2k_icp.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
