"""GPU suite: the ICP and CPD legs of the reference's convergence test set (GetConvergenceTestSet, source/common/testset.cpp:119-187) -- FIVE repetitions of nine
random known transformations (0.2 / 0.4 / 0.6 rad x 10 / 20 / 30 units, drawn by the reference's generators; the set has no seeds: a repetition is another draw) at the
set's first size (20 000 points of bird.obj for ICP; 4 000 of bunny.obj for CPD, and the nine pairs at its next two sizes), cloud-spread 10, max-iterations 100 -- and the
ICP leg of the SIZES set at its own stride up to the largest object in data/ (testset.cpp:48-80) -- through `mi-slam` (configuration -> OBJ -> input stage on the device
incl. the random draw -> adapter with cpu-slam's rules -> C ABI) and through the ABI, against what the reference's own cpu-slam produced (tests/golden/convergence_*.json,
sizes_icp.json; oracle/make_golden_convergence.py), with cpu-slam run AGAIN on the same points in another order beside every result.

Round 6.  ICP, 54 configurations: cpu-slam converges on 31 of the 45 convergence-set draws (its additive translation update, basicicp.cpp:43-44, sends the others away) and
on 5 of the 9 sizes; reordered it keeps its iteration count on 26 and stays 4.6e-5 ... 1.5e-2 from itself where it converges.  Bars:
  * prepared clouds: the reference's, bit for bit (the random transformation included);
  * **the parity mode** -- cpu-slam's sequential fp32 sums, its cross-covariance of fp32-centred pairs, the 3 x 3 SVD in IEEE arithmetic -- **retraces the CPU restatement on
    all 54: the same iteration count, R|t bit for bit (measured 0.0)**, converging, capped and diverging runs alike.  (Found with these fixtures: the first iteration of a
    registration whose clouds start 30 units apart matches 20 000 points to TWO; the exact cross-covariance is then rank 1 and leaves R undetermined, cpu-slam's is decided by
    the rounding of its centring, and the device -- which used the exact one -- landed in another basin on a configuration cpu-slam converges on.)
  * the product's fast K3 in the same mode, and `mi-slam` with the default fp64 sums: recorded everywhere; asserted -- the restatement's iteration count, no farther from the
    restatement or from cpu-slam than 1.5 x cpu-slam's own spread under reordering (+ 1e-4) -- where cpu-slam's answer is defined: it converges before the cap, keeps its
    iteration count when reordered and stays within 5e-3 of itself.
CPD, 65 configurations (hybrid, cpd-weight 0.1, cpd-tolerance 1e-4): see test_convergence_set_cpd.
"""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import Golden, check_measured, frob, noise_corpus, write_noise_meshes
from test_host_cpp import EXE, read_dump

pytestmark = pytest.mark.gpu

CONV = Golden().json("convergence_icp.json")["configs"] + Golden().json("sizes_icp.json")["configs"]     # (+ the ICP leg of GetSizesTestSet, testset.cpp:48-80)
CONV_CPD = Golden().json("convergence_cpd.json")["configs"]
NICP_SETS = Golden().json("nicp_sets.json")["configs"]


@pytest.fixture(scope="module")
def corpus_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("convergence_set")
    _, meshes = noise_corpus(Golden())
    write_noise_meshes({"bird.obj": meshes["bird.obj"], "bunny.obj": meshes["bunny.obj"]}, str(d))
    return d


@pytest.fixture(scope="module")
def ieee_ctx(capi):
    """A context whose 3 x 3 SVDs run in IEEE divisions and roots (MISLAM_SVD_IEEE=1; switches are read at context creation)."""
    os.environ["MISLAM_SVD_IEEE"] = "1"
    try:
        c = capi.Context(0)
    finally:
        del os.environ["MISLAM_SVD_IEEE"]
    yield c
    c.close()


@pytest.mark.parametrize("k", range(len(CONV)))
def test_convergence_set_icp(corpus_dir, ctx, ieee_ctx, capi, k):
    if not os.path.exists(EXE):
        pytest.skip("mi-slam not built")
    c = CONV[k]
    cfg = corpus_dir / ("conv%d.json" % k)
    cfg.write_text(json.dumps(c["config_json"]))
    dump, res_path = corpus_dir / ("clouds%d.bin" % k), corpus_dir / ("result%d.json" % k)
    r = subprocess.run([EXE, str(cfg), "--rules", "cpu", "--dump-clouds", str(dump), "--result-json", str(res_path)], capture_output=True, text=True,
                       cwd=str(corpus_dir), timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    before, after = read_dump(dump)
    assert hashlib.sha256(before.tobytes()).hexdigest() == c["sha256_before"]       # input stage on the device, random transformation drawn on the host
    assert hashlib.sha256(after.tobytes()).hexdigest() == c["sha256_after"]
    res = json.loads(res_path.read_text().replace("-nan", "NaN").replace("nan", "NaN"))
    R_prog = np.array(res["R_colmajor"], np.float64).reshape(3, 3).T
    t_prog = np.array(res["t"], np.float64)
    ref, orc = c["cpu_slam"], c["oracle"]
    diverged = ref["error"] > 1.0
    cap = c["config_json"]["max-iterations"]
    p = capi.icp_params(eps=1e-3, max_iterations=cap, max_distance_squared=10000.0, sum_mode=capi.SUM_CPU_SEQUENTIAL)
    # (1) The PARITY mode proper (round 6): cpu-slam's sequential fp32 sums, ITS cross-covariance (the pairs centred in fp32 with its own centroids: what decides R
    # when the first iteration matches 20 000 points to two -- icp_kernels.hip icp_seq_cross_kernel) and the 3 x 3 SVD in IEEE arithmetic: the restatement's
    # arithmetic operation for operation.  Every configuration of the set -- converging, capped, diverging -- retraces the restatement: its iteration count, R|t to
    # rounding (measured: bit for bit on most).
    Ri, ti, iti, erri = ieee_ctx.icp_register(before, after, p)[:4]
    d_ieee = frob(Ri, ti, orc["R"], orc["t"])
    # (2) the same with the product's fast K3 (refined hardware reciprocals and roots: 1e-5 per iteration of rounding, which a loop of 50-100 iterations that
    # stops while still moving amplifies as it amplifies a reordering of cpu-slam's own input), and (3) the product's default sums through mi-slam
    R, t, it, err = ctx.icp_register(before, after, p)[:4]
    d_orc = frob(R, t, orc["R"], orc["t"])
    d_cpu = frob(R, t, ref["R"], ref["t"])
    spread = max(c["oracle_vs_cpu_slam"], c["cpu_slam_reordered"]["distance"])
    print("%d points, rot %.1f trans %2.0f (seed %d): restatement %d iterations, cpu-slam %d (reordered %d, %.1e from itself; restatement vs cpu-slam %.1e) | IEEE K3 + cpu-slam's sums: %d iterations, "
          "%.3e from the restatement | fast K3: %d iterations, %.3e from the restatement, %.3e from cpu-slam | mi-slam (fp64 sums) %d iterations, %.3e from cpu-slam"
          % (c["n_before"], c["rotation_range"], c["translation_range"], c["seed"], orc["iterations"], ref["iterations"], c["cpu_slam_reordered"]["iterations"],
             c["cpu_slam_reordered"]["distance"], c["oracle_vs_cpu_slam"], iti, d_ieee, it, d_orc, d_cpu, res["iterations"], frob(R_prog, t_prog, ref["R"], ref["t"])))
    assert np.isfinite(R).all() and np.isfinite(t).all() and np.isfinite(Ri).all() and np.isfinite(R_prog).all() and 1 <= res["iterations"] <= cap
    scale = max(1.0, float(np.abs(np.asarray(orc["t"])).max()))
    assert iti == orc["iterations"], (iti, orc["iterations"])
    check_measured("convergence_set_seed%d_ieee_vs_restatement" % c["seed"], d_ieee / scale, 1e-5, floor=1e-6)      # (measured: 0.0 on all 54 -- the restatement's bits)
    # cpu-slam's own spread decides what else carries an assertion: where cpu-slam, handed the same points in another order, keeps its
    # iteration count and stays within 5e-3 of itself, and stops before the cap, the fast-K3 run must keep the restatement's count and stay as close
    well_posed = (not diverged and ref["iterations"] < cap and c["cpu_slam_reordered"]["iterations"] == ref["iterations"] == orc["iterations"] and spread < 5e-3)
    if well_posed:
        assert it == orc["iterations"], (it, orc["iterations"])
        assert d_orc <= 1.5 * spread + 1e-4, (d_orc, spread)
        assert d_cpu <= 1.5 * spread + 1e-4 + d_orc, (d_cpu, spread)


@pytest.mark.parametrize("k", range(len(CONV_CPD)))
def test_convergence_set_cpd(corpus_dir, k):
    """The CPD leg (testset.cpp:122-151): hybrid approximation, cpd-weight 0.1, cpd-tolerance 1e-4, through `mi-slam` -- round 6: the set's five repetitions of the nine random
    transformations at its first size (4 000 points of bunny.obj), the nine at 8 000 and 12 000 points, two at 20 000 points of bird.obj: 65 configurations.  Unlike the noise corpus
    this set is well posed (the same cloud before and after, no noise): the restatement keeps cpu-slam's iteration count almost everywhere, cpu-slam reordered stays 2e-5 ... 3e-3 from
    itself; at translation 30 cpu-slam stops after 4-5 iterations at sigma^2 = 5.8, 1.7 from the known motion (its tolerance rule, coherentpointdrift.cpp:113-117) -- and so must the
    device.  Bars: clouds bit for bit; an iteration count cpu-slam itself shows (its own, a reordered run's, the restatement's) or one next to it (see below); where the count is the
    restatement's, the distance to it recorded and held inside max(3e-4, 1.5 x cpu-slam's own spread); no farther from cpu-slam than 1.5 x that spread (+ 1e-4)."""
    if not os.path.exists(EXE):
        pytest.skip("mi-slam not built")
    c = CONV_CPD[k]
    cfg = corpus_dir / ("conv_cpd%d.json" % k)
    cfg.write_text(json.dumps(c["config_json"]))
    dump, res_path = corpus_dir / ("cpd_clouds%d.bin" % k), corpus_dir / ("cpd_result%d.json" % k)
    r = subprocess.run([EXE, str(cfg), "--dump-clouds", str(dump), "--result-json", str(res_path)], capture_output=True, text=True, cwd=str(corpus_dir), timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    before, after = read_dump(dump)
    assert hashlib.sha256(before.tobytes()).hexdigest() == c["sha256_before"]
    assert hashlib.sha256(after.tobytes()).hexdigest() == c["sha256_after"]
    res = json.loads(res_path.read_text().replace("-nan", "NaN").replace("nan", "NaN"))
    sR = np.array(res["R_colmajor"], np.float64).reshape(3, 3).T
    t = np.array(res["t"], np.float64)
    ref, orc = c["cpu_slam"], c["oracle"]
    d_orc, d_cpu = frob(sR, t, orc["sR"], orc["t"]), frob(sR, t, ref["sR"], ref["t"])
    spread = max([q["distance"] for q in c["cpu_slam_reordered"]] + [c["oracle_vs_cpu_slam"]])
    print("CPD rot %.1f trans %2.0f: iterations %d (restatement %d, cpu-slam %d), |d(sR|t)|_F vs restatement %.3e, vs cpu-slam %.3e (restatement vs cpu-slam %.3e, "
          "cpu-slam vs itself reordered %s), sigma^2 %.4g (cpu-slam %.4g)" % (c["rotation_range"], c["translation_range"], res["iterations"], orc["iterations"], ref["iterations"],
                                                                             d_orc, d_cpu, c["oracle_vs_cpu_slam"], ["%.1e" % q["distance"] for q in c["cpu_slam_reordered"]],
                                                                             res["error"], ref["error"]))
    assert np.isfinite(sR).all() and np.isfinite(t).all() and 1 <= res["iterations"] <= 100
    if spread > 1e-2:
        # the two configurations at 20 000 points of bird.obj: cpu-slam handed the same points in another order ends 0.16 ... 4 away from itself after
        # 11 ... 55 iterations (the noise corpus's finding) -- recorded, not asserted
        return
    seen = {orc["iterations"], ref["iterations"]} | {q["iterations"] for q in c["cpu_slam_reordered"]}
    # Round 6 (81 configurations: the set's five repetitions and three sizes): these runs end on the tolerance rule while sigma^2 still falls by a factor of
    # four per iteration (coherentpointdrift.cpp:113-117), so a likelihood that differs in its last digits can grant ONE more EM step (seed 2105: 35 iterations
    # where cpu-slam, its reordered runs and the restatement take 34 -- sigma^2 1.3e-4 instead of 5.6e-4, 2.6e-3 of s R|t: that step's own motion).  Within one
    # iteration of a count cpu-slam itself shows is accepted, and recorded; the distances are asserted where the count is one of cpu-slam's own.
    assert min(abs(res["iterations"] - q) for q in seen) <= 1, (res["iterations"], seen)
    if res["iterations"] not in seen:
        return
    if res["iterations"] == orc["iterations"]:
        check_measured("convergence_set_cpd_seed%d_vs_restatement" % c["seed"], d_orc, max(3e-4, 1.5 * spread), floor=1e-5)
    assert d_cpu <= 1.5 * spread + 1e-4, (d_cpu, spread)


@pytest.mark.parametrize("k", range(len(NICP_SETS)))
def test_nicp_sets(corpus_dir, k):
    """The NICP legs of the reference's sizes set (testset.cpp:48-80: approximation none, parser defaults: 32 repetitions, subcloud 1 000; 1 000 / 5 000 /
    13 000 points of bunny.obj, 33 000 of bird.obj) and performance set (:82-116: hybrid, cloud-spread 10, 64 repetitions; 10 000 / 20 000 / 30 000 points),
    sequential policy, the random known transformation -- through `mi-slam`: configuration -> OBJ -> input stage -> the adapter, which draws the comparison
    subcloud and one permutation per repetition from the generator AS THE INPUT STAGE LEFT IT (noniterative.cpp:213-222), exactly what the reference's program
    does (oracle/_ref: ref_clouds_from_config_random, then ref_nicp_continue without reseeding).  Bars: clouds bit for bit; cpu-slam's repetition count (on the
    un-spread bunny clouds, ~0.1 units across, the first candidate's error is already below eps = 1e-3 and cpu-slam stops there, 2 from the known motion:
    so must the device); R|t within 1e-4 x max(1, |t|) (north_star) of cpu-slam's -- measured 1.6e-6 ... 3.3e-4 at |t| = 10 ... 17: the same candidates chosen, the
    documented fp64-moments-vs-fp32-SVD deviation -- and the error as close as that distance allows."""
    if not os.path.exists(EXE):
        pytest.skip("mi-slam not built")
    c = NICP_SETS[k]
    cfg = corpus_dir / ("nicp_set%d.json" % k)
    cfg.write_text(json.dumps(c["config_json"]))
    dump, res_path = corpus_dir / ("nicp_clouds%d.bin" % k), corpus_dir / ("nicp_result%d.json" % k)
    r = subprocess.run([EXE, str(cfg), "--dump-clouds", str(dump), "--result-json", str(res_path)], capture_output=True, text=True, cwd=str(corpus_dir), timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    before, after = read_dump(dump)
    assert hashlib.sha256(before.tobytes()).hexdigest() == c["sha256_before"]
    assert hashlib.sha256(after.tobytes()).hexdigest() == c["sha256_after"]
    res = json.loads(res_path.read_text().replace("-nan", "NaN").replace("nan", "NaN"))
    R = np.array(res["R_colmajor"], np.float64).reshape(3, 3).T
    t = np.array(res["t"], np.float64)
    ref = c["cpu_slam"]
    d = frob(R, t, ref["R"], ref["t"])
    print("NICP %s set, %d points: repetitions %d (cpu-slam %d), |d(R|t)|_F vs cpu-slam %.3e, error %.6g (cpu-slam %.6g)"
          % (c["set"], c["n_before"], res["iterations"], ref["repetitions"], d, res["error"], ref["error"]))
    assert res["iterations"] == ref["repetitions"]
    check_measured("nicp_set_%d_vs_cpu_slam" % k, d, 1e-4 * max(1.0, float(np.abs(ref["t"]).max())), floor=2e-6)
    # the error is the mean squared distance of the comparison subcloud under R|t: a transformation d away moves it by up to 2 sqrt(error) d
    assert abs(res["error"] - ref["error"]) <= 3.0 * np.sqrt(ref["error"]) * d + 1e-4 * ref["error"] + 1e-7
