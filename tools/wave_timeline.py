#!/usr/bin/env python3
"""Developer tool (needs the MISLAM_DEV_WAVE_TIMELINE variant build, run with MISLAM_LIB=.../libmislam_timeline.so): when the waves of
ONE search launch start, end their grid scan and end, against the launch's span -- where a launch's time goes.
    python tools/wave_timeline.py [iteration ...]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):   # see bench.quiet_host_pools: BLAS pools vs the CPU quota
    os.environ.setdefault(_v, "1")
import numpy as np  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
from bench import synth_cloud  # noqa: E402

capi = load_package().capi
its = [int(a) for a in sys.argv[1:]] or [6, 12, 20]
n = int(os.environ.get("POINTS", "1000000"))
path = "/tmp/mislam_timeline.bin"
os.environ["MISLAM_DEV_TIMELINE_FILE"] = path
before, after = synth_cloud(np, n)
with capi.Context(0) as ctx:
    ctx.icp_load(before, after, capi.icp_params(eps=0.0, max_iterations=-1, sync_every=1))
    done = 0
    for it in its:
        if it > done:
            ctx.icp_run(it - done)
        done = it + 1
        ctx.search_stats(True)
        ctx.icp_run(1)
        ctx.search_stats(False)
        tl = np.fromfile(path, dtype=np.uint64).reshape(-1, 16)[: (n + 63) // 64].astype(np.int64)
        t0 = tl[:, 0].min()
        start, scan, end = (tl[:, 0] - t0) * 0.01, (tl[:, 1] - t0) * 0.01, (tl[:, 2] - t0) * 0.01      # microseconds (100 MHz ticks)
        steps, far_lanes, walk_only = tl[:, 3] & 0xffff, (tl[:, 3] >> 16) & 0xffff, (tl[:, 3] >> 32) & 1
        walked = far_lanes > 0
        span = end.max()
        print("it %d: launch span %.1f us, %d waves, %d walked (%d of them walk-only)" % (it, span, len(tl), walked.sum(), walk_only.sum()))
        for name, m in (("scan-only waves", ~walked), ("walking waves", walked)):
            if m.sum() == 0:
                continue
            print("  %-16s start p50 %.1f p99 %.1f max %.1f | scan time mean %.1f p99 %.1f | after-scan (walk) mean %.1f p99 %.1f max %.1f | end p50 %.1f p99 %.1f max %.1f"
                  % (name, np.percentile(start[m], 50), np.percentile(start[m], 99), start[m].max(), (scan - start)[m].mean(),
                     np.percentile((scan - start)[m], 99), (end - scan)[m].mean(), np.percentile((end - scan)[m], 99), (end - scan)[m].max(),
                     np.percentile(end[m], 50), np.percentile(end[m], 99), end[m].max()))
        for name, mm in (("walk-only waves", walk_only == 1), ("scan-then-walk waves", walked & (walk_only == 0))):
            if mm.sum():
                print("  %-20s %5d: steps mean %.1f p99 %d max %d | start->end mean %.1f p99 %.1f max %.1f | end p50 %.1f p99 %.1f max %.1f"
                      % (name, mm.sum(), steps[mm].mean(), np.percentile(steps[mm], 99), steps[mm].max(), (end - start)[mm].mean(),
                         np.percentile((end - start)[mm], 99), (end - start)[mm].max(), np.percentile(end[mm], 50), np.percentile(end[mm], 99), end[mm].max()))
        m = ~walked
        t_in, t_blk = (tl[:, 4] - t0) * 0.01, (tl[:, 5] - t0) * 0.01
        trips_b, trips_r, batches_r = tl[:, 6] & 0xffff, (tl[:, 6] >> 16) & 0xffff, (tl[:, 6] >> 32) & 0xffff
        item_passes, batches_r = batches_r >> 8, batches_r & 0xff
        print("  scan-only waves, us: prologue %.1f | first batch %.1f (%.1f trips) | rest %.1f (%.2f item passes, %.1f lockstep batches, %.1f trips) | epilogue %.1f"
              % ((t_in - start)[m].mean(), (t_blk - t_in)[m].mean(), trips_b[m].mean(), (scan - t_blk)[m].mean(), item_passes[m].mean(), batches_r[m].mean(), trips_r[m].mean(), (end - scan)[m].mean()))
        p1, p2, p3 = (tl[:, 8] - t0) * 0.01, (tl[:, 9] - t0) * 0.01, (tl[:, 10] - t0) * 0.01
        print("  scan-only waves, prologue us: work order -> %.1f | state + moving point -> %.1f | key, slot, previous match -> %.1f | near flag -> %.1f"
              % ((p1 - start)[m].mean(), (p2 - p1)[m].mean(), (p3 - p2)[m].mean(), (t_in - p3)[m].mean()))
        if walked.sum():
            w = np.where(walked)[0]
            per_step = (end - scan)[w] / np.maximum(steps[w], 1)
            print("  walk: steps mean %.1f p99 %d max %d; us per step mean %.2f (p10 %.2f p90 %.2f); far lanes per walking wave mean %.1f"
                  % (steps[w].mean(), np.percentile(steps[w], 99), steps[w].max(), per_step.mean(), np.percentile(per_step, 10), np.percentile(per_step, 90), far_lanes[w].mean()))
        # where the waves ran: XCD (XCC_ID), and the slot (SE, CU, SIMD, wave) inside it -- per XCD the end of its last wave, and per slot the
        # gaps between one wave's end and the next one's start
        hw, xcc = tl[:, 7] & 0xffffffff, (tl[:, 7] >> 32) & 0xf
        slot = (xcc << 32) | (hw & 0xffff)
        print("  per XCD: waves / walkers / last start / last end: " + "  ".join("%d: %d/%d/%.0f/%.0f" % (x, (xcc == x).sum(), (walked & (xcc == x)).sum(), start[xcc == x].max(), end[xcc == x].max()) for x in sorted(set(xcc.tolist()))))
        b = np.arange(len(tl))
        print("  XCC_ID of workgroup b is a function of b %% 8: %s; walkers by b %% 8: %s; walkers among the first / second / later thirds of the launch order: %d / %d / %d"
              % (all(len(set(xcc[b % 8 == k].tolist())) == 1 for k in range(8)), [int((walked & (b % 8 == k)).sum()) for k in range(8)],
                 walked[: len(tl) // 3].sum(), walked[len(tl) // 3: 2 * len(tl) // 3].sum(), walked[2 * len(tl) // 3:].sum()))
        wpos = np.where(walked)[0]
        print("  launch position of the walkers: p50 %d p90 %d p99 %d max %d; steps by XCD: %s" % (np.percentile(wpos, 50), np.percentile(wpos, 90), np.percentile(wpos, 99), wpos.max(),
              [int(steps[walked & (xcc == x)].sum()) for x in sorted(set(xcc.tolist()))]))
        order = np.lexsort((start, slot))
        same = slot[order][1:] == slot[order][:-1]
        gaps = (start[order][1:] - end[order][:-1])[same]
        if len(gaps) == 0:
            gaps = np.zeros(1)                # (every wave had a slot of its own)
        print("  slots used %d; waves per slot mean %.1f; gap between a slot's waves: mean %.2f us p50 %.2f p90 %.2f p99 %.2f (negative: HW_ID wave ids reused) ; sum of gaps / (slots x span) = %.3f"
              % (len(set(slot.tolist())), len(tl) / max(len(set(slot.tolist())), 1), gaps.mean(), np.percentile(gaps, 50), np.percentile(gaps, 90), np.percentile(gaps, 99),
                 gaps[gaps > 0].sum() / (len(set(slot.tolist())) * span)))
        # waves resident over time (10 us bins)
        edges = np.arange(0, span + 10, 10)
        res = [(int(((start <= t) & (end > t)).sum()), int(((start <= t) & (end > t) & walked).sum())) for t in edges]
        print("  resident waves (all/walking) every 10 us:", " ".join("%d/%d" % r for r in res))
