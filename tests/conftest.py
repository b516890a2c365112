import importlib.util
import json
import os
import sys

for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):   # one BLAS thread: a pool spinning on every visible core
    os.environ.setdefault(_v, "1")                                          # exhausts the GPU box's 16-CPU quota (bench.quiet_host_pools)

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_package():
    """The product package lives in `cuda-slam_amd/` (hyphen), so it is imported by path as `cuda_slam_amd`."""
    if "cuda_slam_amd" in sys.modules:
        return sys.modules["cuda_slam_amd"]
    root = os.path.join(ROOT, "cuda-slam_amd")
    spec = importlib.util.spec_from_file_location("cuda_slam_amd", os.path.join(root, "__init__.py"),
                                                  submodule_search_locations=[root])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cuda_slam_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def capi():
    capi = load_package().capi
    # built artefacts stay out of history: a fresh checkout builds them once (hipcc cross-compiles without a GPU)
    if not os.path.exists(capi.LIB_PATH) or not os.path.exists(os.path.join(ROOT, "cuda-slam_amd", "mi-slam")):
        import __graft_entry__
        __graft_entry__.build()
    return capi


@pytest.fixture(scope="session")
def oracle():
    from oracle import oraclebind
    oraclebind.lib()
    return oraclebind


@pytest.fixture(scope="session")
def ref():
    """The reference's own cpu-slam code (oracle/_ref, built in the build container and shipped prebuilt)."""
    from oracle import refbind
    if not refbind.available():
        pytest.skip("oracle/_ref/libref_cpuslam.so not built (needs /root/reference: make -C oracle ref)")
    refbind.lib()
    return refbind


@pytest.fixture(scope="session")
def ctx(capi):
    """One device context for the whole GPU session (the HIP path; there is no fallback -- this raises without a GPU)."""
    c = capi.Context(0)
    yield c
    c.close()


class Golden:
    def npz(self, name):
        return np.load(os.path.join(GOLD, name))   # allow_pickle stays False

    def json(self, name):
        with open(os.path.join(GOLD, name)) as f:
            return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return Golden()


@pytest.fixture(scope="session")
def bunny(golden):
    z = golden.npz("bunny_clouds.npz")
    return z["before"], z["after"]


def synth_cloud(n, seed=666, m=None):
    """SURVEY 8d synthetic recipe (same as oracle/make_golden.py): uniform [-5,5]^3, 0.2 rad about (1,2,3)/sqrt14,
    translation 10*(1,1,1)/sqrt3, independently permuted target."""
    rng = np.random.default_rng(seed)
    before = rng.uniform(-5.0, 5.0, size=(n, 3)).astype(np.float32)
    axis = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
    ang = 0.2
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    Rm = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    t = 10.0 * np.ones(3) / np.sqrt(3.0)
    perm = rng.permutation(n)
    after = (before[perm].astype(np.float64) @ Rm.T + t).astype(np.float32)
    if m is not None:
        after = after[:m]
    return before, after, Rm.astype(np.float32), t.astype(np.float32)


def frob(R1, t1, R2, t2):
    return float(np.sqrt(((np.asarray(R1, np.float64) - R2) ** 2).sum() + ((np.asarray(t1, np.float64) - t2) ** 2).sum()))


def prepare_cases(golden):
    """The input-stage fixture (tests/golden/bunny_prepare.npz, oracle/make_golden_prepare.py) as a list of
    (options, raw cloud, {side: draws}, {side: the reference's prepared cloud})."""
    z = golden.npz("bunny_prepare.npz")
    raw_all = golden.npz("bunny_clouds.npz")["before"]
    cases = []
    k = 0
    while "c%d_options" % k in z.files:
        opt = json.loads(str(z["c%d_options" % k]))
        draws = {}
        for side in "ba":
            d = {}
            for name in ("subcloud_idx", "shuffle_idx", "noise_rows", "noise_unit", "outlier_unit"):
                key = "c%d_%s_%s" % (k, side, name)
                d[name] = (z[key].astype(np.int32) if name.endswith(("idx", "rows")) else z[key]) if key in z.files else None
            draws[side] = d
        cases.append((opt, raw_all[:opt["raw_rows"]], draws, {"b": z["c%d_before" % k], "a": z["c%d_after" % k]}))
        k += 1
    return cases


def prepare_kwargs(opt, draws, side):
    """Arguments of oraclebind.prepare_cloud / Context.prepare_cloud for one side ('b' = before, 'a' = after) of a case."""
    d = draws[side]
    noise = opt["noise_before" if side == "b" else "noise_after"]
    kw = dict(subcloud_idx=d["subcloud_idx"], shuffle_idx=d["shuffle_idx"], noise_rows=d["noise_rows"], noise_unit=d["noise_unit"],
              noise_intensity=0.0 if noise is None else noise[1], outlier_unit=d["outlier_unit"], spread=opt["spread"])
    if side == "a":
        kw.update(R=np.array(opt["R"], np.float32), t=np.array(opt["t"], np.float32))
    return kw


# ---- the reference's CPD evaluation corpus (doc/noise/configs, oracle/make_golden_noise.py) ----
def noise_corpus(golden):
    """(fixture document, {file name: raw cloud}) -- the raw clouds are the meshes' face corners in face order (loader.cpp:58-66)."""
    doc = golden.json("noise_configs.json")
    z = golden.npz("noise_meshes.npz")
    meshes = {}
    for key in z.files:
        if key.endswith("_v"):
            name = key[:-2].replace("_", "-") + ".obj"
            meshes[name] = (z[key], z[key[:-2] + "_f"].astype(np.int64))
    return doc, meshes


def write_noise_meshes(meshes, directory):
    """The corpus' .obj files rebuilt from the fixture (vertices with nine significant digits: every fp32 value survives the round
    trip; faces: the corner list cut into triangles -- a loader that yields one point per face corner reads the same cloud from any cut)."""
    os.makedirs(os.path.join(directory, "data"), exist_ok=True)
    for name, (v, f) in meshes.items():
        with open(os.path.join(directory, "data", name), "w") as out:
            for p in v:
                out.write("v %.9g %.9g %.9g\n" % (float(p[0]), float(p[1]), float(p[2])))
            cuts = list(range(0, len(f) - len(f) % 3, 3))
            for k, c in enumerate(cuts):
                end = len(f) if k == len(cuts) - 1 else c + 3
                out.write("f " + " ".join(str(int(i) + 1) for i in f[c:end]) + "\n")


# ---- measured values beside generous bars ----
# A bar that is 10x what is measured lets a silent 10x regression through.  check_measured(key, value, bar) asserts the bar AND, where
# tests/golden/measured_bounds.json holds this quantity's value as measured on MI355X (tools/collect_measured.py writes it from the "MEASURED"
# lines of a GPU run's log), value <= factor x that measurement (+ an absolute floor for quantities that sit at rounding level).
def _measured_bounds():
    path = os.path.join(GOLD, "measured_bounds.json")
    try:
        with open(path) as f:
            return json.load(f)["values"]
    except (OSError, ValueError, KeyError):
        return {}


MEASURED = _measured_bounds()


def check_measured(key, value, bar, factor=2.0, floor=0.0):
    value = float(value)
    print("MEASURED %s %.6e" % (key, value))
    assert value <= bar, (key, value, bar)
    m = MEASURED.get(key)
    if m is not None:
        assert value <= factor * float(m) + floor, "%s: %.3e against %.3e measured when the fixture was written (x %.1f allowed): a regression inside the bar %.3e" % (key, value, float(m), factor, bar)
