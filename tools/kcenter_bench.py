#!/usr/bin/env python3
"""Developer tool: the K-centre sweep (fgt_cluster through mi_fgt_kcenter: upload, sweep, member lists, means, download) at sizes beyond one workgroup's
registers -- the cooperative several-workgroup sweep (cpd_fgt.hip: fgt_kcenter_coop_kernel) against the paths it replaces (MISLAM_FGT_COOP_SWEEP=0:
one workgroup with its distances in memory up to 65 536 points, two launches per centre beyond), one process per setting; labels compared.
    python tools/kcenter_bench.py"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import hashlib
    import numpy as np
    from __graft_entry__ import load_package
    capi = load_package().capi
    ctx = capi.Context(0)
    out = {}
    for n in (20000, 49000, 100000, 300000, 1000000):
        rng = np.random.default_rng(n)
        cloud = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
        for K in (51, 200):
            ctx.fgt_kcenter(cloud, K)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                centers, cluster = ctx.fgt_kcenter(cloud, K)
                best = min(best, (time.perf_counter() - t0) * 1e3)
            out["%d/%d" % (n, K)] = [round(best, 3), hashlib.sha256(cluster.tobytes() + centers.tobytes()).hexdigest()[:12]]
    print(json.dumps(out))


def main():
    if "--child" in sys.argv:
        return child()
    res = {}
    for coop in ("1", "0"):
        env = dict(os.environ, MISLAM_FGT_COOP_SWEEP=coop)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=280)
        if r.returncode != 0:
            print("coop=%s FAILED: %s" % (coop, r.stderr[-600:]))
            continue
        res[coop] = json.loads(r.stdout.strip().splitlines()[-1])
    for key in res.get("1", {}):
        a, b = res["1"][key], res.get("0", {}).get(key, [float("nan"), "?"])
        print("n/K %-12s cooperative %8.3f ms   before %8.3f ms   same labels and means: %s" % (key, a[0], b[0], a[1] == b[1]), flush=True)


if __name__ == "__main__":
    main()
