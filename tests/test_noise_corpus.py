"""CPU suite: the reference's CPD evaluation corpus (doc/noise/configs/config*.json, 25 of 39 runnable with the .obj files the reference
ships) through the C++ host side's input stage -- configuration reader, OBJ reader, GetCloudsFromConfig mirror -- against the clouds the
reference's own code prepared (tests/golden/noise_configs.json holds their sizes and sha256; oracle/make_golden_noise.py).  No GPU call:
MISLAM_DUMP_ONLY stops mi-slam before the registration; the GPU suite (test_gpu_noise_corpus.py) runs the registrations."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import Golden, noise_corpus, write_noise_meshes
from test_host_cpp import EXE, read_dump

DOC, MESHES = noise_corpus(Golden())
CONFIGS = {c["config"]: c for c in DOC["configs"]}


@pytest.fixture(scope="module")
def corpus_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("noise_corpus")
    write_noise_meshes(MESHES, str(d))
    return d


def test_the_corpus_is_complete():
    # 39 configurations; 14 name .obj files that are missing blobs of the reference checkout (.MISSING_LARGE_BLOBS)
    assert len(CONFIGS) == 25 and len(DOC["skipped"]) == 14
    assert sorted(CONFIGS) + sorted(s["config"] for s in DOC["skipped"]) != [] and set(CONFIGS) | {s["config"] for s in DOC["skipped"]} == set(range(1, 40))
    assert all(c["options"]["method"] == "cpd" and c["options"]["approximation"] == "hybrid" for c in CONFIGS.values())
    assert sum(1 for c in CONFIGS.values() if c["options"]["cpd_const_scale"]) == 19
    # bird.obj: 8 752 quads = 35 008 points (testset.cpp:25-26): one point per face corner as written
    assert len(MESHES["bird.obj"][1]) == 35008 and len(MESHES["bunny.obj"][1]) == 14904


@pytest.mark.parametrize("number", sorted(CONFIGS))
def test_host_input_stage_reproduces_the_reference_clouds(corpus_dir, number):
    if not os.path.exists(EXE):
        pytest.skip("mi-slam not built (run __graft_entry__.build())")
    c = CONFIGS[number]
    cfg = corpus_dir / ("config%d.json" % number)
    cfg.write_text(json.dumps(c["config_json"]))           # the reference's file as it is: paths relative to the working directory
    dump = corpus_dir / ("clouds%d.bin" % number)
    r = subprocess.run([EXE, str(cfg), "--prepare", "host", "--dump-clouds", str(dump)], capture_output=True, text=True,
                       env=dict(os.environ, MISLAM_DUMP_ONLY="1"), cwd=str(corpus_dir), timeout=120)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    before, after = read_dump(dump)
    assert (len(before), len(after)) == (c["n_before"], c["n_after"])
    assert hashlib.sha256(before.tobytes()).hexdigest() == c["sha256_before"]
    assert hashlib.sha256(after.tobytes()).hexdigest() == c["sha256_after"]


# ---- the ICP leg of the reference's convergence test set (testset.cpp:119-187; oracle/make_golden_convergence.py) ----
CONV = Golden().json("convergence_icp.json")["configs"] + Golden().json("sizes_icp.json")["configs"]     # (+ the ICP leg of GetSizesTestSet, testset.cpp:48-80)


@pytest.mark.parametrize("k", range(len(CONV)))
def test_host_input_stage_draws_the_reference_random_transformation(corpus_dir, k):
    # "rotation-range" / "translation-range": the known transformation is DRAWN (Tests::GetRandomRotationMatrix / GetRandomTranslationVector,
    # testutils.cpp:43-55) from rand() behind the shuffles -- the host mirror must draw the same numbers in the same order: the reference's
    # prepared clouds, bit for bit (sha256), for all nine (rotation, translation) pairs of the set
    if not os.path.exists(EXE):
        pytest.skip("mi-slam not built (run __graft_entry__.build())")
    c = CONV[k]
    cfg = corpus_dir / ("conv%d.json" % k)
    cfg.write_text(json.dumps(c["config_json"]))
    dump = corpus_dir / ("conv_clouds%d.bin" % k)
    r = subprocess.run([EXE, str(cfg), "--prepare", "host", "--dump-clouds", str(dump)], capture_output=True, text=True,
                       env=dict(os.environ, MISLAM_DUMP_ONLY="1"), cwd=str(corpus_dir), timeout=120)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    before, after = read_dump(dump)
    assert (len(before), len(after)) == (c["n_before"], c["n_after"])
    assert hashlib.sha256(before.tobytes()).hexdigest() == c["sha256_before"]
    assert hashlib.sha256(after.tobytes()).hexdigest() == c["sha256_after"]


# ---- the CPD leg of the same set (hybrid, cpd-weight 0.1, cpd-tolerance 1e-4; 4 000 points of bunny.obj; oracle/make_golden_convergence.py --cpd) ----
CONV_CPD = Golden().json("convergence_cpd.json")["configs"]


@pytest.mark.parametrize("k", range(len(CONV_CPD)))
def test_host_input_stage_of_the_cpd_convergence_set(corpus_dir, k):
    if not os.path.exists(EXE):
        pytest.skip("mi-slam not built (run __graft_entry__.build())")
    c = CONV_CPD[k]
    cfg = corpus_dir / ("conv_cpd%d.json" % k)
    cfg.write_text(json.dumps(c["config_json"]))
    dump = corpus_dir / ("conv_cpd_clouds%d.bin" % k)
    r = subprocess.run([EXE, str(cfg), "--prepare", "host", "--dump-clouds", str(dump)], capture_output=True, text=True,
                       env=dict(os.environ, MISLAM_DUMP_ONLY="1"), cwd=str(corpus_dir), timeout=120)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    before, after = read_dump(dump)
    assert (len(before), len(after)) == (c["n_before"], c["n_after"])
    assert hashlib.sha256(before.tobytes()).hexdigest() == c["sha256_before"]
    assert hashlib.sha256(after.tobytes()).hexdigest() == c["sha256_after"]


# ---- the NICP legs of the reference's sizes and performance sets (testset.cpp:48-116; oracle/make_golden_nicp_sets.py) ----
NICP_SETS = Golden().json("nicp_sets.json")["configs"]


@pytest.mark.parametrize("k", range(len(NICP_SETS)))
def test_host_input_stage_of_the_nicp_sets(corpus_dir, k):
    if not os.path.exists(EXE):
        pytest.skip("mi-slam not built (run __graft_entry__.build())")
    c = NICP_SETS[k]
    cfg = corpus_dir / ("nicp_set%d.json" % k)
    cfg.write_text(json.dumps(c["config_json"]))
    dump = corpus_dir / ("nicp_set_clouds%d.bin" % k)
    r = subprocess.run([EXE, str(cfg), "--prepare", "host", "--dump-clouds", str(dump)], capture_output=True, text=True,
                       env=dict(os.environ, MISLAM_DUMP_ONLY="1"), cwd=str(corpus_dir), timeout=120)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    before, after = read_dump(dump)
    assert (len(before), len(after)) == (c["n_before"], c["n_after"])
    assert hashlib.sha256(before.tobytes()).hexdigest() == c["sha256_before"]
    assert hashlib.sha256(after.tobytes()).hexdigest() == c["sha256_after"]
