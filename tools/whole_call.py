#!/usr/bin/env python3
"""Whole mi_icp_register calls on host buffers, the way the reference times a SlamFunc (testrunner.cpp:54-56: allocation, upload,
index builds and release included; sweep rules of testset.cpp:82-117: max 50 iterations, GPU-reference driver rules), and where
the time of one call goes (mi_icp_load_times with the stream drained after every stage + the iterations).
    python tools/whole_call.py [points ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):   # see bench.quiet_host_pools: BLAS pools vs the CPU quota
    os.environ.setdefault(_v, "1")
import numpy as np  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
from bench import synth_cloud, whole_call  # noqa: E402

capi = load_package().capi
sizes = [int(a) for a in sys.argv[1:]] or [100000, 1000000]
with capi.Context(0) as ctx:
    for n in sizes:
        print(json.dumps(whole_call(np, capi, ctx, n)), flush=True)
