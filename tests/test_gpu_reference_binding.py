"""GPU suite: the reference-side binding of INTEGRATION.md, run for real.

integration/mi355x_adapters.cpp holds the three function bodies a maintainer of the reference adds behind
GetCudaIcpTransformationMatrix / GetCudaCpdTransformationMatrix / GetCudaNicpTransformationMatrix.  oracle/Makefile compiles
that file against the reference's OWN headers (source/common/common.h, glm) and links the reference's common.cpp object and
libmislam.so into oracle/_ref/binding_check (built in the build container, shipped prebuilt like the rest of oracle/_ref).
Here it runs on the bunny clouds; its results must be the library's, bit for bit, and -- for the method whose random draws the
reference's generator provides -- the reference's own."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, frob

pytestmark = pytest.mark.gpu

EXE = os.path.join(ROOT, "oracle", "_ref", "binding_check")


def run(tmp_path, before, after, *args):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/binding_check not built (needs /root/reference: make -C oracle ref)")
    path = tmp_path / "clouds.bin"
    with open(path, "wb") as f:
        np.array([len(before), len(after)], np.int32).tofile(f)
        np.ascontiguousarray(before, np.float32).tofile(f)
        np.ascontiguousarray(after, np.float32).tofile(f)
    r = subprocess.run([EXE, str(path)] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    d = json.loads(line[len("RESULT "):])
    R = np.array(d["R_colmajor"], np.float32).reshape(3, 3).T
    return R, np.array(d["t"], np.float32), d["iterations"], np.float32(d["error"]), r.stdout


def test_icp_entry_point(tmp_path, ctx, capi, bunny):
    before, after = bunny
    R, t, it, err, out = run(tmp_path, before, after, "icp", 1e-3, 50)
    Rl, tl, itl, errl = ctx.icp_register(before, after, capi.icp_params(cuda_slam=True, eps=1e-3, max_iterations=50))
    assert it == itl and np.array_equal(R, Rl) and np.array_equal(t, tl) and err == np.float32(errl)
    Rcfg = np.array([[0.36, 0.47, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]])      # config/default.json
    assert np.abs(R - Rcfg).max() < 2e-2 and np.abs(t - 1.0).max() < 2e-2


def test_cpd_entry_point_default_approximation(tmp_path, ctx, capi, bunny):
    before, after = bunny
    R, t, it, err, out = run(tmp_path, before, after, "cpd", 1e-3, 50, 0.3, 1e-3, 2)
    p = capi.cpd_params(eps=1e-3, max_iterations=50, weight=0.3, tolerance=1e-3, approximation=capi.CPD_APPROX_HYBRID,
                        sigma2_mode=capi.SIGMA2_CPU_SEQUENTIAL)   # the adapter's default at this size: cpu-slam's own sigma^2_0
    sR, tl, sc, itl, errl = ctx.cpd_register(before, after, p)
    assert it == itl and np.array_equal(R, sR) and np.array_equal(t, tl)


@pytest.mark.parametrize("name,approx", [("none", 0), ("hybrid", 2)])
def test_nicp_entry_point_draws_like_the_reference(tmp_path, ctx, capi, golden, bunny, name, approx):
    # the adapter draws the subcloud and the permutations from Common::mtRandom through the reference's own
    # GetRandomPermutationVector; seeded like the fixture generator, it must land on the reference's cpu-slam result
    before, after = bunny
    g = golden.json("bunny_nicp.json")
    R, t, it, err, out = run(tmp_path, before, after, "nicp", g["eps"], g["repetitions"], approx, g["subcloud_size"], g["seed"])
    r = g["runs"][name]
    assert it == r["repetitions"]
    assert frob(R, t, np.array(r["R"]), np.array(r["t"])) < 1e-4
    p = capi.nicp_params(eps=g["eps"], max_repetitions=g["repetitions"], approximation=approx)
    Rl, tl, itl, errl = ctx.nicp_register(before, after, p, np.array(g["order_heads"], np.int32), np.array(g["subcloud_idx"], np.int32))
    assert it == itl and np.array_equal(R, Rl) and np.array_equal(t, tl)
