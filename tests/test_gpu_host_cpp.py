"""GPU suite: the C++ host side end to end -- `mi-slam config.json` (JSON config -> OBJ -> clouds -> SlamFunc adapter -> C ABI
-> HIP kernels -> printed result), compared with the same registration driven through the Python binding on the clouds the
program dumped.  This is the drop-in path a user of the reference's `cuda-slam` binary takes."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from test_host_cpp import EXE, read_dump, write_obj

pytestmark = pytest.mark.gpu


def make_obj(path, n_vertices=3000, seed=4):
    rng = np.random.default_rng(seed)
    # a bumpy sphere: structured enough for ICP to converge
    u = rng.normal(size=(n_vertices, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = u * (1.0 + 0.3 * np.sin(3 * u[:, :1]) * np.cos(2 * u[:, 1:2]))
    faces = [(int(a) + 1, int(b) + 1, int(c) + 1) for a, b, c in rng.integers(0, n_vertices, (4000, 3))]
    write_obj(path, v, faces)


def run_mi_slam(cfg, tmp_path, *extra):
    if not os.path.exists(EXE):
        pytest.skip("mi-slam not built")
    p = tmp_path / "cfg.json"
    p.write_text(json.dumps(cfg))
    args = [EXE, str(p), "--dump-clouds", str(tmp_path / "clouds.bin"), "--result-json", str(tmp_path / "result.json")] + list(extra)
    return subprocess.run(args, capture_output=True, text=True, cwd=str(tmp_path), timeout=300)


@pytest.mark.parametrize("rules", ["cpu", "cuda"])
def test_mi_slam_icp_matches_the_abi(tmp_path, capi, ctx, rules):
    make_obj(tmp_path / "model.obj")
    cfg = {"before-path": "model.obj", "after-path": "model.obj", "method": "icp", "translation": [0.3, -0.2, 0.1],
           "rotation": [0.9553365, -0.2955202, 0.0, 0.2955202, 0.9553365, 0.0, 0.0, 0.0, 1.0], "cloud-spread": 10.0,
           "max-iterations": 60, "random-seed": 7, "max-distance-squared": 400}
    r = run_mi_slam(cfg, tmp_path, "--rules", rules)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Results:" in r.stdout and "Rotation matrix:" in r.stdout and "Translation vector:" in r.stdout   # mainwrapper.cpp:27-34
    res = json.loads((tmp_path / "result.json").read_text())
    before, after = read_dump(tmp_path / "clouds.bin")
    assert before.shape == (12000, 3)                       # 4 000 faces x 3 corners (loader.cpp:58-66)
    p = capi.icp_params(cuda_slam=(rules == "cuda"), eps=1e-3, max_iterations=60, max_distance_squared=400.0)
    R, t, it, err = ctx.icp_register(before, after, p)
    Rc = np.array(res["R_colmajor"], np.float32).reshape(3, 3).T
    assert res["iterations"] == it
    assert np.array_equal(Rc, R) and np.array_equal(np.array(res["t"], np.float32), t)   # same library, same inputs: bitwise
    # the printed matrix is the same numbers in the reference's "%1.8f " row format
    rows = re.findall(r"^(-?\d\.\d{8}) (-?\d\.\d{8}) (-?\d\.\d{8}) $", r.stdout, flags=re.M)
    assert len(rows) >= 3 and np.allclose(np.array(rows[-3:], float), R, atol=5e-8)
    # and it actually registered the clouds
    assert err < 1e-3 or it == 60


def test_mi_slam_cpd(tmp_path, capi, ctx):
    make_obj(tmp_path / "model.obj", n_vertices=1500)
    cfg = {"before-path": "model.obj", "after-path": "model.obj", "method": "cpd", "translation": [0.3, -0.2, 0.1],
           "rotation": [0.9553365, -0.2955202, 0.0, 0.2955202, 0.9553365, 0.0, 0.0, 0.0, 1.0], "cloud-spread": 10.0,
           "max-iterations": 40, "random-seed": 7, "approximation-type": "none", "cloud-before-resize": 2000,
           "cloud-after-resize": 2000}
    r = run_mi_slam(cfg, tmp_path)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads((tmp_path / "result.json").read_text())
    before, after = read_dump(tmp_path / "clouds.bin")
    assert before.shape == (2000, 3) and after.shape == (2000, 3)
    sR, t, sc, it, err = ctx.cpd_register(before, after, capi.cpd_params(max_iterations=40, sigma2_mode=capi.SIGMA2_CPU_SEQUENTIAL))
    Rc = np.array(res["R_colmajor"], np.float32).reshape(3, 3).T
    assert res["iterations"] == it and np.array_equal(Rc, sR) and np.array_equal(np.array(res["t"], np.float32), t)


def test_mi_slam_cpd_default_approximation_is_hybrid(tmp_path, capi, ctx):
    # no "approximation-type" key: the reference's parser falls back to hybrid (configparser.cpp:221-230) and so does mi-slam
    make_obj(tmp_path / "model.obj", n_vertices=1500)
    cfg = {"before-path": "model.obj", "after-path": "model.obj", "method": "cpd", "translation": [0.3, -0.2, 0.1],
           "rotation": [0.9553365, -0.2955202, 0.0, 0.2955202, 0.9553365, 0.0, 0.0, 0.0, 1.0], "cloud-spread": 10.0,
           "max-iterations": 40, "random-seed": 7, "cloud-before-resize": 2000, "cloud-after-resize": 2000}
    r = run_mi_slam(cfg, tmp_path)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Approximation type: hybrid" in r.stdout
    res = json.loads((tmp_path / "result.json").read_text())
    before, after = read_dump(tmp_path / "clouds.bin")
    p = capi.cpd_params(max_iterations=40, approximation=capi.CPD_APPROX_HYBRID, sigma2_mode=capi.SIGMA2_CPU_SEQUENTIAL)
    sR, t, sc, it, err = ctx.cpd_register(before, after, p)
    Rc = np.array(res["R_colmajor"], np.float32).reshape(3, 3).T
    assert res["iterations"] == it and np.array_equal(Rc, sR) and np.array_equal(np.array(res["t"], np.float32), t)
    exact = ctx.cpd_register(before, after, capi.cpd_params(max_iterations=40, sigma2_mode=capi.SIGMA2_CPU_SEQUENTIAL))
    assert not np.array_equal(exact[0], sR)                # the approximation really ran


@pytest.mark.parametrize("approx", ["none", "hybrid"])
def test_mi_slam_nicp(tmp_path, capi, ctx, ref, approx):
    # "method": "nicp": the program draws the comparison subcloud and one permutation per repetition from the generator that
    # "random-seed" seeded and the cloud stage already shuffled both clouds with -- the reference's draw order
    # (common.cpp:166-167, noniterative.cpp:213-222).  The same draws come out of the reference's own generator here.
    make_obj(tmp_path / "model.obj")
    seed, reps, sub_n = 11, 6, 500
    cfg = {"before-path": "model.obj", "after-path": "model.obj", "method": "nicp", "translation": [0.3, -0.2, 0.1],
           "rotation": [0.9553365, -0.2955202, 0.0, 0.2955202, 0.9553365, 0.0, 0.0, 0.0, 1.0], "cloud-spread": 10.0,
           "random-seed": seed, "nicp-iterations": reps, "nicp-subcloud-size": sub_n, "approximation-type": approx,
           "convergence-epsilon": 1e-7}
    r = run_mi_slam(cfg, tmp_path)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads((tmp_path / "result.json").read_text())
    before, after = read_dump(tmp_path / "clouds.bin")
    n = len(before)
    sub = ref.random_permutation(seed, n, 2)[:sub_n]                     # draws 0 and 1 shuffled the two clouds
    heads = np.stack([ref.random_permutation(seed, n, 3 + k)[:3] for k in range(reps)])
    p = capi.nicp_params(eps=1e-7, max_repetitions=reps, approximation={"none": 0, "hybrid": 2}[approx])
    R, t, it, err = ctx.nicp_register(before, after, p, heads, sub)
    Rc = np.array(res["R_colmajor"], np.float32).reshape(3, 3).T
    assert res["iterations"] == it == reps
    assert np.array_equal(Rc, R) and np.array_equal(np.array(res["t"], np.float32), t)


def test_mi_slam_rejects_unknown_method_like_the_reference(tmp_path):
    make_obj(tmp_path / "model.obj", n_vertices=300)
    cfg = {"before-path": "model.obj", "after-path": "model.obj", "method": "banana", "translation": [0, 0, 0],
           "rotation": [1, 0, 0, 0, 1, 0, 0, 0, 1]}
    r = run_mi_slam(cfg, tmp_path)
    assert r.returncode != 0


@pytest.mark.parametrize("variant", ["explicit", "random-transform", "two-files"])
def test_device_input_stage_equals_the_host_one(tmp_path, variant):
    # --prepare device (mi_prepare_cloud, the default) and --prepare host (cloud_io.cpp's mirror of GetCloudsFromConfig) draw the
    # same random outcomes and must dump identical clouds: resize, normalisation, shuffle, noise, outliers, transformation
    make_obj(tmp_path / "model.obj", n_vertices=2000, seed=11)
    make_obj(tmp_path / "other.obj", n_vertices=1500, seed=12)
    cfg = {"before-path": "model.obj", "after-path": "model.obj", "method": "icp", "cloud-spread": 7.5, "max-iterations": 3,
           "random-seed": 31, "cloud-before-resize": 5000, "cloud-after-resize": 7000, "noise-affected-points-before": 0.2,
           "noise-intensity-before": 0.05, "noise-affected-points-after": 0.6, "noise-intensity-after": 0.01,
           "additional-outliers-before": 17, "additional-outliers-after": 3}
    if variant == "random-transform":
        cfg.update({"translation-range": 2.0, "rotation-range": 0.4})
    else:
        cfg.update({"translation": [0.3, -0.2, 0.1], "rotation": [0.9553365, -0.2955202, 0.0, 0.2955202, 0.9553365, 0.0, 0.0, 0.0, 1.0]})
    if variant == "two-files":
        cfg["after-path"] = "other.obj"
        cfg["cloud-after-resize"] = 100000          # beyond the cloud: no subcloud, nothing drawn
    dumps = {}
    for where in ("device", "host"):
        r = run_mi_slam(cfg, tmp_path, "--prepare", where)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        dumps[where] = read_dump(tmp_path / "clouds.bin")
    for k in (0, 1):
        assert dumps["device"][k].shape == dumps["host"][k].shape
        assert np.array_equal(dumps["device"][k].view(np.uint32), dumps["host"][k].view(np.uint32)), (variant, k)
    assert dumps["device"][0].shape == (5017, 3)
