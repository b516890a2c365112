"""GPU suite: the multi-rank HIP path with MORE THAN ONE rank, on the one GPU of the box (see tests/_dist_gpu_worker.py):
2 and 3 processes, each with its own context on device 0, the caller's-transport context (mi_ctx_create_exchange) over gloo."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_icp_and_cpd_with_real_ranks(capi, world):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", "_dist_gpu_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "DIST_GPU_OK world=%d" % world in r.stdout


def test_exchange_context_argument_checks(capi):
    with pytest.raises(capi.MiSlamError):
        capi.Context(0, 2, 2, exchange=lambda arr, kind: None)      # rank out of range
    # one rank over the caller's transport: the callback is still called (and may be the identity)
    seen = []
    with capi.Context(0, 0, 1, exchange=lambda arr, kind: seen.append((kind, len(arr)))) as c:
        import numpy as np
        rng = np.random.default_rng(5)
        a = rng.uniform(-1, 1, (500, 3)).astype(np.float32)
        R, t, it, err = c.icp_register(a, a + np.float32(0.05), capi.icp_params(eps=0.0, max_iterations=3))
        assert it == 3
    assert any(k == capi.EXCHANGE_SUM_F64 for k, _ in seen)
