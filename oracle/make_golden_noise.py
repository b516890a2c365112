"""TEST INFRASTRUCTURE ONLY.  Fixtures for the reference's own CPD evaluation corpus, doc/noise/configs/config1..39.json
(doc/documentation.tex:476-574): every configuration whose .obj files are present under /root/reference/data is run through the
reference's OWN code (oracle/_ref) -- GetCloudsFromConfig's stages (common.cpp:134-210: subcloud, normalisation, shuffle, noise,
outliers, known transformation, on the reference's generators seeded with the config's seed) and then cpu-slam's
GetRigidCPDTransformationMatrix (coherentpointdrift.cpp:69-124) with the config's approximation-type / cpd-const-scale / cpd-weight /
cpd-tolerance / convergence-epsilon / max-iterations.  Run in the build container (minutes of CPU per configuration):

    python oracle/make_golden_noise.py [--jobs 7] [--only 1,2,28]

Output
    tests/golden/noise_meshes.npz    the raw clouds' sources: per .obj file its vertex table and the face-corner index list (the
                                     reference's loader yields one point per face corner, loader.cpp:58-66): data files, not code
    tests/golden/noise_configs.json  per configuration: the options as the reference's parser resolves them (configparser.cpp:186-257),
                                     sizes and sha256 of the two prepared clouds, cpu-slam's iterations / s*R / t / final sigma^2, and the
                                     same run by the fp64-summing restatement (oracle/slam_oracle.c) with its distance to cpu-slam --
                                     which says how reproducible cpu-slam's own number is (a run that ends in cancellation noise is not)
The configurations whose files are missing blobs (.MISSING_LARGE_BLOBS: mustang, rose, airbus, plane-*) are listed as skipped.
"""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = os.environ.get("REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, ROOT)


def load_mesh(path):
    """(vertex table float32 [V,3], face-corner indices int32 [C]) -- oracle/objio.py's reading of the file, kept factored."""
    verts, corners = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                ids = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    ids.append(i - 1 if i > 0 else len(verts) + i)
                corners.extend(ids)
    return np.asarray(verts, np.float32).reshape(-1, 3), np.asarray(corners, np.int32)


def resolve(cfg):
    """The options the reference's parser hands on (configparser.cpp:186-257), JSON-serialisable."""
    rot = [float(np.float32(v)) for v in cfg["rotation"]]
    scale = float(np.float32(cfg.get("scale", 1.0)))
    return {
        "before": os.path.basename(cfg["before-path"]), "after": os.path.basename(cfg["after-path"]),
        "method": cfg["method"], "approximation": cfg.get("approximation-type", "hybrid"),
        "max_iterations": cfg.get("max-iterations", -1), "rotation_rowmajor": rot, "translation": [float(v) for v in cfg["translation"]],
        "scale": scale, "resize_before": cfg.get("cloud-before-resize"), "resize_after": cfg.get("cloud-after-resize"),
        "spread": cfg.get("cloud-spread"), "seed": cfg.get("random-seed"),
        "noise_before": None if "noise-affected-points-before" not in cfg else [float(cfg["noise-affected-points-before"]), float(cfg.get("noise-intensity-before", 0.1))],
        "noise_after": None if "noise-affected-points-after" not in cfg else [float(cfg["noise-affected-points-after"]), float(cfg.get("noise-intensity-after", 0.1))],
        "outliers_before": int(cfg.get("additional-outliers-before", 0)), "outliers_after": int(cfg.get("additional-outliers-after", 0)),
        "cpd_weight": float(cfg.get("cpd-weight", 0.3)), "cpd_const_scale": bool(cfg.get("cpd-const-scale", False)),
        "cpd_tolerance": float(cfg.get("cpd-tolerance", 1e-3)), "eps": float(cfg.get("convergence-epsilon", 1e-3)),
        "ratio_of_far_field": float(cfg.get("fgt-ratio-of-far-field", 10.0)), "order_of_truncation": int(cfg.get("fgt-order-of-truncation", 8)),
        "json": cfg,
    }


def known_transform(opt):
    # configparser.cpp:139-147: rotationMatrix[y][x] = rotation[x*3+y] (the JSON is row-major), then scale * rotationMatrix
    R = (np.float32(opt["scale"]) * np.array(opt["rotation_rowmajor"], np.float32).reshape(3, 3)).astype(np.float32)
    return R, np.array(opt["translation"], np.float32)


def prepared_clouds(opt, meshes):
    from oracle import refbind as ref
    vb, fb = meshes[opt["before"]]
    raw_b = np.ascontiguousarray(vb[fb])
    raw_a = None
    if opt["after"] != opt["before"]:
        va, fa = meshes[opt["after"]]
        raw_a = np.ascontiguousarray(va[fa])
    R, t = known_transform(opt)
    return ref.clouds_from_config_full(raw_b, raw_a, opt["seed"], R, t, resize_before=opt["resize_before"], resize_after=opt["resize_after"],
                                       spread=opt["spread"], noise_before=None if opt["noise_before"] is None else tuple(opt["noise_before"]),
                                       noise_after=None if opt["noise_after"] is None else tuple(opt["noise_after"]),
                                       outliers_before=opt["outliers_before"], outliers_after=opt["outliers_after"])


APPROX = {"none": 0, "full": 1, "hybrid": 2}
SPREAD_RUNS = 2


def frob(R1, t1, R2, t2):
    return float(np.sqrt(((np.asarray(R1, np.float64) - R2) ** 2).sum() + ((np.asarray(t1, np.float64) - t2) ** 2).sum()))


def run_one(args):
    number, opt, meshes = args
    from oracle import oraclebind as O
    from oracle import refbind as ref
    # the reference prints one line per EM iteration: send them to a log beside the fixture's scratch, not to the terminal
    devnull = os.open(os.devnull, os.O_WRONLY)
    keep = os.dup(1)
    os.dup2(devnull, 1)
    try:
        before, after = prepared_clouds(opt, meshes)
        kw = dict(eps=opt["eps"], weight=opt["cpd_weight"], const_scale=opt["cpd_const_scale"], max_iterations=opt["max_iterations"],
                  tolerance=opt["cpd_tolerance"], ratio_of_far_field=opt["ratio_of_far_field"], order_of_truncation=float(opt["order_of_truncation"]))
        t0 = time.time()
        sR, t, it, err = ref.cpd(before, after, fgt=APPROX[opt["approximation"]], **kw)
        t_ref = time.time() - t0
        s20 = ref.cpd_sigma_squared(before, after)
        t0 = time.time()
        oR, ot, oit, oerr = O.cpd_approx(before, after, APPROX[opt["approximation"]], **kw)
        t_orc = time.time() - t0
        # cpu-slam's own reproducibility: the same two point SETS in another order (what another "random-seed" of the reference's own shuffle
        # would hand it, common.cpp:166-167) -- in exact arithmetic the same registration problem; in cpu-slam's fp32 sums, and in the K-centre
        # clustering of its Fast Gauss Transform, another rounding of it.  How far its answer moves is how far ANY faithful implementation may
        # sit from the number above.
        spread = []
        if np.isfinite(sR).all():
            for k in range(SPREAD_RUNS):
                rng = np.random.default_rng(1000 * number + k)
                pR, pt, pit, perr = ref.cpd(before[rng.permutation(len(before))], after[rng.permutation(len(after))], fgt=APPROX[opt["approximation"]], **kw)
                spread.append({"iterations": int(pit), "distance": frob(pR, pt, sR, t), "error": float(perr)})
    finally:
        os.dup2(keep, 1)
        os.close(keep)
        os.close(devnull)
    out = {
        "config": number, "options": {k: v for k, v in opt.items() if k != "json"}, "config_json": opt["json"],
        "n_before": int(len(before)), "n_after": int(len(after)),
        "sha256_before": hashlib.sha256(before.tobytes()).hexdigest(), "sha256_after": hashlib.sha256(after.tobytes()).hexdigest(),
        "sigma2_init": float(s20),
        "cpu_slam": {"iterations": int(it), "sR": np.asarray(sR, np.float64).tolist(), "t": np.asarray(t, np.float64).tolist(), "error": float(err), "seconds": round(t_ref, 1)},
        "oracle": {"iterations": int(oit), "sR": np.asarray(oR, np.float64).tolist(), "t": np.asarray(ot, np.float64).tolist(), "error": float(oerr), "seconds": round(t_orc, 1)},
        "oracle_vs_cpu_slam": frob(oR, ot, sR, t),
        "cpu_slam_reordered": spread,
    }
    if not np.isfinite(sR).all():       # (config 7: cpu-slam's saturated sigma^2_0 leaves every P below fp32 -- Np = 0, NaN after one iteration)
        out["cpu_slam"].update(sR=None, t=None, error=None, diverged=True)
        out["oracle"].update(sR=None, t=None, error=None, diverged=bool(not np.isfinite(oR).all()))
        out["oracle_vs_cpu_slam"] = None
    sys.stderr.write("config%-2d %5d x %5d  cpu-slam %3d it %.3g s  sigma2 %.4g | oracle %3d it, |d|_F %.3e | reordered: %s\n"
                     % (number, len(before), len(after), it, t_ref, err, oit, frob(oR, ot, sR, t), ", ".join("%d it %.2e" % (q["iterations"], q["distance"]) for q in spread)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=7)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    only = {int(v) for v in a.only.split(",") if v}
    cfg_dir = os.path.join(REF, "doc", "noise", "configs")
    opts, skipped, names = {}, [], set()
    for number in range(1, 40):
        cfg = json.load(open(os.path.join(cfg_dir, "config%d.json" % number)))
        files = [os.path.join(REF, cfg["before-path"]), os.path.join(REF, cfg["after-path"])]
        if not all(os.path.exists(f) for f in files):
            skipped.append({"config": number, "missing": [os.path.basename(f) for f in files if not os.path.exists(f)]})
            continue
        opts[number] = resolve(cfg)
        names.update((opts[number]["before"], opts[number]["after"]))
    meshes = {name: load_mesh(os.path.join(REF, "data", name)) for name in sorted(names)}
    packed = {}
    for name, (v, f) in meshes.items():
        key = name[:-4].replace("-", "_")
        packed[key + "_v"] = v
        packed[key + "_f"] = f.astype(np.uint16) if len(v) < 65536 else f
    np.savez_compressed(os.path.join(GOLD, "noise_meshes.npz"), **packed)
    todo = [(n, o, meshes) for n, o in sorted(opts.items(), key=lambda kv: -(kv[1]["resize_before"] or 15000) * (kv[1]["resize_after"] or 15000) * kv[1]["max_iterations"])
            if not only or n in only]
    with mp.Pool(a.jobs) as pool:
        results = pool.map(run_one, todo, chunksize=1)
    path = os.path.join(GOLD, "noise_configs.json")
    doc = {"source": "doc/noise/configs/config*.json run by oracle/_ref (cpu-slam), see oracle/make_golden_noise.py", "skipped": skipped, "configs": []}
    if only and os.path.exists(path):
        doc = json.load(open(path))
        doc["configs"] = [c for c in doc["configs"] if c["config"] not in only]
    doc["configs"] = sorted(doc["configs"] + results, key=lambda c: c["config"])
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", path, "with", len(doc["configs"]), "configurations;", len(skipped), "skipped (missing .obj blobs)")


if __name__ == "__main__":
    main()
