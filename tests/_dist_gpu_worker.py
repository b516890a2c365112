"""Worker of tests/test_gpu_dist_ranks.py: one rank of a REAL multi-rank run of the HIP path, all ranks on the one GPU of the box.

RCCL refuses two ranks on one device, so the ranks use mi_ctx_create_exchange with gloo as the transport: every kernel, shard
range, index base and the order of the collectives are the ones the RCCL context runs (the two contexts differ in the one
function that moves the operand); the result is checked against a single-GPU context in the same process.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import load_package, synth_cloud  # noqa: E402

SIGN = np.uint64(1 << 63)


def frob(Ra, ta, Rb, tb):
    return float(np.sqrt(((np.asarray(Ra, np.float64) - Rb) ** 2).sum() + ((np.asarray(ta, np.float64) - tb) ** 2).sum()))


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    capi = load_package().capi
    calls = {capi.EXCHANGE_MIN_U64: 0, capi.EXCHANGE_SUM_F64: 0}

    def exchange(arr, kind):
        calls[kind] += 1
        if kind == capi.EXCHANGE_MIN_U64:     # gloo has no unsigned MIN: flipping the top bit maps unsigned order onto signed
            t = torch.from_numpy((arr ^ SIGN).view(np.int64))
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            arr[:] = t.numpy().view(np.uint64) ^ SIGN
        else:
            t = torch.from_numpy(arr)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bunny_clouds.npz"))
    before, after = z["before"], z["after"]
    with capi.Context(0, rank, world, exchange=exchange) as dctx, capi.Context(0) as sctx:
        assert dctx.rank_world() == (rank, world)

        # the sharded correspondence search: bit-exact keys, duplicates across the shard boundary resolved to the lowest index
        tgt = after.copy()
        tgt[len(tgt) // 2: len(tgt) // 2 + 200] = tgt[100:300]
        i1, d1 = dctx.nn_search(before[:3000], tgt)
        i2, d2 = sctx.nn_search(before[:3000], tgt)
        assert np.array_equal(i1, i2) and np.array_equal(d1.view(np.uint32), d2.view(np.uint32))
        assert calls[capi.EXCHANGE_MIN_U64] >= 1

        # ICP, every way of splitting it, both searches: the same iteration count and stop reason, R|t equal to the single-GPU
        # run up to the order of the fp64 partial sums (1e-6; the correspondences themselves are identical)
        for nn_mode in (capi.NN_BRUTEFORCE, capi.NN_TREE, capi.NN_GRID):
            for shard_mode in (capi.SHARD_TARGET, capi.SHARD_SOURCE, capi.SHARD_AUTO):
                for kw in (dict(eps=1e-3, max_iterations=50, max_distance_squared=400.0), dict(max_iterations=7),
                           dict(eps=1e-3, max_iterations=50, max_distance_squared=400.0, sync_every=3)):
                    p = capi.icp_params(nn_mode=nn_mode, shard_mode=shard_mode, **kw)
                    a = dctx.icp_register(before, after, p)
                    b = sctx.icp_register(before, after, p)
                    assert a[2] == b[2], (nn_mode, shard_mode, kw, a[2], b[2])
                    assert frob(a[0], a[1], b[0], b[1]) < 1e-6 and abs(a[3] - b[3]) < 1e-7, (nn_mode, shard_mode, kw)
        # the GPU reference's rules (abort + rollback when the error rises)
        a = dctx.icp_register(before, after, capi.icp_params(cuda_slam=True, max_iterations=60))
        b = sctx.icp_register(before, after, capi.icp_params(cuda_slam=True, max_iterations=60))
        assert a[2] == b[2] and frob(a[0], a[1], b[0], b[1]) < 1e-6

        # a run enqueued in pieces
        for c in (dctx, sctx):
            c.icp_load(before, after, capi.icp_params(eps=1e-3, max_iterations=50, max_distance_squared=400.0))
            assert c.icp_run(5) == 5
            c.icp_run(-1)
        ra, rb = dctx.icp_result(), sctx.icp_result()
        assert ra[2] == rb[2] and ra[4] == rb[4] == capi.STOP_CONVERGED and frob(ra[0], ra[1], rb[0], rb[1]) < 1e-6

        # a larger synthetic problem with ragged shares (sizes not divisible by the rank count)
        sb, sa = synth_cloud(200003, m=150001)[:2]
        for shard_mode in (capi.SHARD_TARGET, capi.SHARD_SOURCE):
            p = capi.icp_params(eps=0.0, max_iterations=6, shard_mode=shard_mode)
            a = dctx.icp_register(sb, sa, p)
            b = sctx.icp_register(sb, sa, p)
            assert a[2] == b[2] == 6 and frob(a[0], a[1], b[0], b[1]) < 1e-6, shard_mode

        # shares whose REDUCED-ROW counts differ between the ranks (33 chunks of 64 points reduce to 2 rows, 32 chunks to 1): the
        # in-place all-reduce of the 64 reduced rows must not carry a neighbour's sums of the previous iteration into the next
        # one (it did until round 4: wrong moments from the second iteration on)
        n_odd = 64 * 65 if world == 2 else 64 * 97
        ob, oa = synth_cloud(n_odd, m=n_odd + 37, seed=11)[:2]
        for nn_mode in (capi.NN_BRUTEFORCE, capi.NN_GRID):
            for its in (2, 5):
                p = capi.icp_params(eps=0.0, max_iterations=its, shard_mode=capi.SHARD_SOURCE, nn_mode=nn_mode)
                a = dctx.icp_register(ob, oa, p)
                b = sctx.icp_register(ob, oa, p)
                assert a[2] == b[2] == its and frob(a[0], a[1], b[0], b[1]) < 1e-6 and abs(a[3] - b[3]) < 1e-7, (nn_mode, its, frob(a[0], a[1], b[0], b[1]))

        # rigid CPD with the fixed cloud sharded: the per-point sums are fp32 in a different order, hence 1e-4 like the
        # bar against cpu-slam itself
        for kw in (dict(max_iterations=50), dict(max_iterations=9, const_scale=1)):
            cp = capi.cpd_params(approximation=capi.CPD_APPROX_NONE, **kw)
            ca = dctx.cpd_register(before, after, cp)          # (sR, t, scale, iterations, error)
            cb = sctx.cpd_register(before, after, cp)
            assert ca[3] == cb[3], (kw, ca[3], cb[3])
            assert frob(ca[0], ca[1], cb[0], cb[1]) < 1e-4 and abs(ca[2] - cb[2]) < 1e-5, kw
        assert calls[capi.EXCHANGE_SUM_F64] > 50
        # The Fast Gauss Transform modes (hybrid is the reference parser's default, configparser.cpp:217).  Rounds 4-5 ran them REPLICATED on a multi-rank
        # context; round 6 splits the E-step's QUERIES over the ranks -- the fixed points of the first transform, the moving points of
        # the second, the truncated E-step's fixed-cloud tiles -- with clusterings and coefficient tables replicated: every per-point value is the
        # single-GPU run's to the bit (the weights travel through an unsigned-minimum all-reduce against all-ones), only the M-step's 24 sums are added in
        # another grouping.  Same iteration counts, R|t within 1e-6.
        n_sum, n_min = calls[capi.EXCHANGE_SUM_F64], calls[capi.EXCHANGE_MIN_U64]
        for approx, kw in ((capi.CPD_APPROX_HYBRID, dict(max_iterations=50)), (capi.CPD_APPROX_FULL, dict(max_iterations=5)),
                           (capi.CPD_APPROX_FULL, dict(max_iterations=17)),
                           (capi.CPD_APPROX_HYBRID, dict(max_iterations=50, sigma2_mode=capi.SIGMA2_CPU_SEQUENTIAL))):
            cp = capi.cpd_params(approximation=approx, **kw)
            ca = dctx.cpd_register(before, after, cp)
            cb = sctx.cpd_register(before, after, cp)
            assert ca[3] == cb[3], (approx, kw, ca[3], cb[3])
            d = frob(ca[0], ca[1], cb[0], cb[1])
            assert d < 1e-6 and abs(ca[2] - cb[2]) < 1e-6 and abs(ca[4] - cb[4]) <= 1e-6 * max(1.0, abs(cb[4])), (approx, kw, d, ca[2], cb[2], ca[4], cb[4])
        assert calls[capi.EXCHANGE_SUM_F64] > n_sum and calls[capi.EXCHANGE_MIN_U64] > n_min, "the sharded E-step combines its sums and its weights through the transport"
        # ragged shares: sizes the rank count does not divide, fewer fixed-cloud tiles than a whole number per rank
        rb_, ra_ = synth_cloud(5003, m=4099, seed=5)[:2]
        for approx, kw in ((capi.CPD_APPROX_HYBRID, dict(max_iterations=30)), (capi.CPD_APPROX_FULL, dict(max_iterations=6))):
            cp = capi.cpd_params(approximation=approx, **kw)
            ca, cb = dctx.cpd_register(rb_, ra_, cp), sctx.cpd_register(rb_, ra_, cp)
            assert ca[3] == cb[3] and frob(ca[0], ca[1], cb[0], cb[1]) < 1e-6, (approx, kw, ca[3], cb[3], frob(ca[0], ca[1], cb[0], cb[1]))

    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("DIST_GPU_OK world=%d exchanges=%d+%d" % (world, calls[capi.EXCHANGE_MIN_U64], calls[capi.EXCHANGE_SUM_F64]))


if __name__ == "__main__":
    main()
