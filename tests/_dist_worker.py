"""Worker of tests/test_dist_protocol.py: one rank of the target-sharded correspondence search, on CPU over gloo.

Per-shard compute is the oracle (test infrastructure); what is under test is the PROTOCOL the HIP path uses across GPUs --
the product's own shard ranges (mi_shard_range) and key packing (mi_pack_key), a MIN all-reduce of the packed keys, and the
owner-accumulates rule for the moments and error sums followed by a SUM all-reduce -- checked against the unsharded result.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import load_package  # noqa: E402

from oracle import oraclebind as O  # noqa: E402


def moments(src, tgt, idx, mask):
    b = src[mask].astype(np.float64)
    a = tgt[idx[mask]].astype(np.float64)
    return np.concatenate([[mask.sum()], b.sum(0), a.sum(0), (a[:, :, None] * b[:, None, :]).sum(0).reshape(9)])


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    capi = load_package().capi

    rng = np.random.default_rng(1234)          # same data on every rank
    n, m = 700, 1003
    src = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    tgt = rng.uniform(-5, 5, (m, 3)).astype(np.float32)
    tgt[600:900] = tgt[100:400]                # duplicates straddling the shard boundaries: lowest GLOBAL index must win
    src[:50] = tgt[100:150]

    lo, hi = capi.shard_range(m, rank, world)
    lidx, ld2 = O.nn_search(src, tgt[lo:hi], threads=1)
    keys = np.array([capi.pack_key(d, i + lo) for d, i in zip(ld2, lidx)], dtype=np.uint64)
    assert np.all(keys < (1 << 63))            # d2 >= 0: sign bit clear, so int64 MIN orders like uint64 MIN
    t = torch.from_numpy(keys.astype(np.int64))
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    merged = t.numpy().astype(np.uint64)
    gidx = (merged & np.uint64(0xFFFFFFFF)).astype(np.int32)
    gd2 = (merged >> np.uint64(32)).astype(np.uint32).view(np.float32)

    ridx, rd2 = O.nn_search(src, tgt, threads=1)
    assert np.array_equal(gidx, ridx), "merged argmin differs from the unsharded search"
    assert np.array_equal(gd2.view(np.uint32), rd2.view(np.uint32))

    # owner-accumulates: the rank whose shard holds the winner adds the pair
    kept = gd2 < np.float32(30.0)
    mine = (gidx >= lo) & (gidx < hi)
    local = moments(src, tgt, gidx, kept & mine)
    tm = torch.from_numpy(local.copy())
    dist.all_reduce(tm, op=dist.ReduceOp.SUM)
    full = moments(src, tgt, ridx, kept)
    assert tm[0].item() == full[0] == kept.sum()          # every kept pair counted exactly once
    assert np.allclose(tm.numpy(), full, rtol=1e-12, atol=1e-9)

    # error sums of the kept pairs against a transformed source
    R = np.array([[0.36, 0.47, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]], np.float32)
    cur = O.transform_cloud(src, R, np.array([0.1, 0.2, 0.3], np.float32))
    sel = kept & mine
    diff = tgt[gidx[sel]].astype(np.float64) - cur[sel]
    te = torch.tensor([float((diff ** 2).sum()), float(sel.sum())], dtype=torch.float64)
    dist.all_reduce(te, op=dist.ReduceOp.SUM)
    dfull = tgt[ridx[kept]].astype(np.float64) - cur[kept]
    assert abs(te[0].item() - (dfull ** 2).sum()) < 1e-9 * max(1.0, (dfull ** 2).sum())
    assert te[1].item() == kept.sum()

    # MI_SHARD_SOURCE as the product splits it (mislam_api.hip mi_icp_load / deal_chunks_kernel): every rank orders the WHOLE moving
    # cloud the same way (the product: along the Hilbert curve, on the device; here any order all ranks agree on) and keeps the
    # 64-point chunks rank, rank + W, rank + 2W ... of that order -- the last chunk may be ragged -- or, below 4 * 64 * W points, a
    # contiguous slice of the caller's order.  Nothing per-point is exchanged; the shares partition the cloud (mi_source_share
    # says how many points each rank holds) and the moments / error sums of the shares add up to the unsharded ones.
    def source_share(points, r):
        count = capi.source_share(points, r, world)
        if points < 4 * 64 * world:
            slo, shi = capi.shard_range(points, r, world)
            assert shi - slo == count
            return np.arange(slo, shi)
        il = np.arange(count)
        g = ((il // 64) * world + r) * 64 + il % 64                       # deal_chunks_kernel's index map
        assert g.max() < points
        return g

    for n2 in (n, 2500, 64 * 4 * world, 64 * 4 * world + 1):               # slices or dealt chunks by world; ragged; the threshold itself
        src2 = src if n2 == n else rng.uniform(-5, 5, (n2, 3)).astype(np.float32)
        order = np.lexsort((src2[:, 2], src2[:, 1], src2[:, 0]))            # stand-in for the Hilbert order
        shares = [source_share(n2, r) for r in range(world)]
        allpts = np.sort(np.concatenate(shares))
        assert np.array_equal(allpts, np.arange(n2)), "the shares do not partition the moving cloud"
        dealt = n2 >= 4 * 64 * world
        mine_idx = order[shares[rank]] if dealt else shares[rank]
        ridx2, rd22 = O.nn_search(src2, tgt, threads=1)
        sidx, sd2 = O.nn_search(src2[mine_idx], tgt, threads=1)
        assert np.array_equal(sidx, ridx2[mine_idx]) and np.array_equal(sd2.view(np.uint32), rd22[mine_idx].view(np.uint32))
        skept, kept2 = sd2 < np.float32(30.0), rd22 < np.float32(30.0)
        full2 = moments(src2, tgt, ridx2, kept2)
        ts = torch.from_numpy(moments(src2[mine_idx], tgt, sidx, skept).copy())
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        assert ts[0].item() == full2[0]
        assert np.allclose(ts.numpy(), full2, rtol=1e-12, atol=1e-9)
        cur2 = O.transform_cloud(src2, R, np.array([0.1, 0.2, 0.3], np.float32))
        sdiff = tgt[sidx[skept]].astype(np.float64) - cur2[mine_idx][skept]
        # ONE all-reduce per iteration in the product: the 16 moments and the 2 error sums travel together (IcpState mom | err)
        tse = torch.from_numpy(np.concatenate([moments(src2[mine_idx], tgt, sidx, skept), [float((sdiff ** 2).sum()), float(skept.sum())]]))
        dist.all_reduce(tse, op=dist.ReduceOp.SUM)
        dfull2 = tgt[ridx2[kept2]].astype(np.float64) - cur2[kept2]
        assert abs(tse[16].item() - (dfull2 ** 2).sum()) < 1e-9 * max(1.0, (dfull2 ** 2).sum()) and tse[17].item() == kept2.sum()

    # C2 -- rigid CPD with the FIXED cloud sharded (mi_cpd_register on a multi-GPU context): rank r owns fixed points
    # [N*r/W, N*(r+1)/W) and the whole moving cloud.  The denominators and Pt1 of its fixed points are local; its P1/PX hold only
    # its fixed points' share, but the M-step moments are linear in them, so ONE sum all-reduce of 8 + 16 doubles (the x-sums
    # and the k-sums, same layout as CpdState::xs/ks) reproduces the unsharded M-step.  Sigma^2_0 needs the fixed cloud's four
    # sums added over the ranks (the moving cloud's are replicated).
    y = (src * 0.9 + 0.2).astype(np.float32)        # "transformed" moving cloud
    sigma2, weight = 1.7, 0.3
    constant = O.cpd_constant(sigma2, weight, n, m)
    lo, hi = capi.shard_range(m, rank, world)
    p1, pt1, px, L = O.cpd_estep(y, tgt[lo:hi], constant, sigma2)       # this rank's share: sums over x in [lo, hi)
    den_log = -(L - np.float32(3 * (hi - lo)) * np.log(np.float32(sigma2)) / np.float32(2.0))   # sum of log den over the shard
    a64, b64 = tgt[lo:hi].astype(np.float64), src.astype(np.float64)
    xs = np.zeros(8)
    xs[0] = den_log
    xs[1:4] = (a64 * pt1[:, None]).sum(0)
    xs[4] = ((a64 ** 2) * pt1[:, None]).sum()
    ks = np.zeros(16)
    ks[0] = p1.astype(np.float64).sum()
    ks[1:4] = (b64 * p1[:, None]).sum(0)
    ks[4:13] = (b64[:, :, None] * px.astype(np.float64)[:, None, :]).sum(0).reshape(9)
    ks[13] = ((b64 ** 2) * p1[:, None]).sum()
    tc = torch.from_numpy(np.concatenate([xs, ks]))
    dist.all_reduce(tc, op=dist.ReduceOp.SUM)
    f1, ft1, fx, fL = O.cpd_estep(y, tgt, constant, sigma2)             # unsharded
    A64 = tgt.astype(np.float64)
    want = np.zeros(24)
    want[0] = -(fL - np.float32(3 * m) * np.log(np.float32(sigma2)) / np.float32(2.0))
    want[1:4] = (A64 * ft1[:, None]).sum(0)
    want[4] = ((A64 ** 2) * ft1[:, None]).sum()
    want[8] = f1.astype(np.float64).sum()
    want[9:12] = (b64 * f1[:, None]).sum(0)
    want[12:21] = (b64[:, :, None] * fx.astype(np.float64)[:, None, :]).sum(0).reshape(9)
    want[21] = ((b64 ** 2) * f1[:, None]).sum()
    # the per-point P1/PX are fp32 sums in a different order when sharded: 1e-5 relative on the moments is their rounding
    assert np.allclose(tc.numpy(), want, rtol=2e-5, atol=1e-4), (tc.numpy(), want)
    assert np.array_equal(pt1, ft1[lo:hi])                              # Pt1 is shard-local, bit for bit
    init = torch.tensor([*a64.sum(0), (a64 ** 2).sum()], dtype=torch.float64)
    dist.all_reduce(init, op=dist.ReduceOp.SUM)
    sb, sbb = b64.sum(0), (b64 ** 2).sum()
    total = m * sbb + n * init[3].item() - 2.0 * float(init[:3].numpy() @ sb)
    exact = ((b64[:, None, :] - A64[None, :, :]) ** 2).sum()
    assert abs(total - exact) < 1e-9 * exact

    # C3 (round 6) -- the FGT / hybrid CPD modes with their E-step's QUERIES sharded (cpd_api.hip cpd_estep_fgt_enqueue): both clouds whole on every rank, the
    # coefficient tables replicated; rank r evaluates the first transform at fixed points mi_shard_range(n), writes THEIR weights (1/den, x/den: a float4 per point),
    # fills the rest with all-ones and the ranks take an element-wise UNSIGNED minimum -- which must hand every rank every weight bit for bit, whatever the
    # pattern (NaN payloads, -0, denormals, infinities: a sum with zeros would not) -- then evaluates the second transform at moving points mi_shard_range(m);
    # the M-step's sums of the shares add up to the unsharded ones.  Per-shard compute is the oracle's FGT E-step (test infrastructure): a query's value does
    # not depend on which other queries are evaluated with it.
    wts = rng.uniform(-3, 3, (m, 4)).astype(np.float32)
    wbits = wts.view(np.uint32)
    wbits[5] = [0x7fc00001, 0xffc12345, 0x7f800000, 0xff800000]         # NaNs with payloads, +-inf
    wbits[6] = [0x80000000, 0x00000001, 0x807fffff, 0x00000000]         # -0, denormals, +0
    wbits[m - 1] = [0xffffffff, 0xfffffffe, 0x7fffffff, 0x80000001]     # a NaN that IS the fill pattern, and its neighbours
    mine = np.full((m, 4), 0xffffffff, np.uint32)
    mine[lo:hi] = wbits[lo:hi]
    sign = np.uint64(1) << np.uint64(63)
    packed = mine.reshape(-1).view(np.uint64) ^ sign                     # gloo has no unsigned MIN: flip the top bit, take the signed one (bench.py does the same)
    tw = torch.from_numpy(packed.view(np.int64).copy())
    dist.all_reduce(tw, op=dist.ReduceOp.MIN)
    got = (tw.numpy().view(np.uint64) ^ sign).view(np.uint32).reshape(m, 4)
    assert np.array_equal(got, wbits), "the unsigned-minimum exchange did not reproduce every rank's bit patterns"
    # (ranges: every fixed point, every moving point and every tile of the truncated E-step belongs to exactly one rank)
    for count in (m, n, (m + 63) // 64, (n + 63) // 64):
        cover = np.zeros(count, np.int32)
        for r in range(world):
            rl, rh = capi.shard_range(count, r, world)
            cover[rl:rh] += 1
        assert np.all(cover == 1)
    # (the shares are cut from the whole run's arrays, as the device cuts them: tables replicated, queries split -- an oracle E-step on a SLICE of the fixed
    # cloud would cluster another cloud)
    f1, ft1, fx, fL = O.cpd_estep_fgt(y, tgt, weight, sigma2, 4.0)
    mlo, mhi = capi.shard_range(n, rank, world)
    a64 = tgt.astype(np.float64)
    xs = np.zeros(8); ks = np.zeros(16)
    xs[1:4] = (a64[lo:hi] * ft1[lo:hi, None]).sum(0); xs[4] = ((a64[lo:hi] ** 2) * ft1[lo:hi, None]).sum()
    ks[0] = f1[mlo:mhi].astype(np.float64).sum(); ks[1:4] = (b64[mlo:mhi] * f1[mlo:mhi, None]).sum(0)
    ks[4:13] = (b64[mlo:mhi, :, None] * fx[mlo:mhi].astype(np.float64)[:, None, :]).sum(0).reshape(9); ks[13] = ((b64[mlo:mhi] ** 2) * f1[mlo:mhi, None]).sum()
    tf = torch.from_numpy(np.concatenate([xs, ks]))
    dist.all_reduce(tf, op=dist.ReduceOp.SUM)
    wantf = np.zeros(24)
    wantf[1:4] = (a64 * ft1[:, None]).sum(0); wantf[4] = ((a64 ** 2) * ft1[:, None]).sum()
    wantf[8] = f1.astype(np.float64).sum(); wantf[9:12] = (b64 * f1[:, None]).sum(0)
    wantf[12:21] = (b64[:, :, None] * fx.astype(np.float64)[:, None, :]).sum(0).reshape(9); wantf[21] = ((b64 ** 2) * f1[:, None]).sum()
    assert np.allclose(tf.numpy(), wantf, rtol=1e-12, atol=1e-9), "the shares' M-step sums do not add up to the whole run's"

    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("DIST_PROTOCOL_OK world=%d" % world)


if __name__ == "__main__":
    main()
