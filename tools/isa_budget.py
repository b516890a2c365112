#!/usr/bin/env python3
"""Static instruction budget of one kernel from its ISA ("static VALU / SALU counts per phase from the ISA").

    tools/isa_budget.py <file.s> <mangled kernel name prefix> [--blocks]

The .s comes from `hipcc ... -gline-tables-only --save-temps` (tools/gpu_isa_budget.sh): every instruction carries the source line it was
generated for (.loc; with inlining: the innermost one), so instructions are attributed to PHASES by source file + line range (PHASES below),
per basic block.  Prints per phase: vector / scalar / scalar-memory / vector-memory / LDS instruction counts (static), and with --blocks the
basic blocks (label, phase mix, whether it is a loop body: a backward branch to its own or an earlier label).
"""
import re
import sys
from collections import Counter, defaultdict


def classify(op):
    if op.startswith(("v_", "v_mfma")):
        return "valu"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache")):
        return "smem"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_sleep", "s_setprio", "s_branch", "s_cbranch")):
        return "sctl"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def parse(path, prefix):
    files = {}
    blocks = []          # (label, [(op, file, line)])
    cur = None
    inside = False
    loc = (None, 0)
    for raw in open(path):
        line = raw.rstrip("\n")
        m = re.match(r"\s+\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", line)
        if m:
            files[int(m.group(1))] = m.group(3) or m.group(2)
            continue
        if not inside:
            if line.startswith(prefix) and line.split(":")[0].startswith(prefix) and ":" in line:
                inside = True
                cur = ("entry", [])
                blocks.append(cur)
            continue
        if line.startswith(".Lfunc_end") or ".end_amdhsa_kernel" in line:
            break
        m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"(\.LBB[0-9_]+):", line)
        if m:
            cur = (m.group(1), [])
            blocks.append(cur)
            continue
        m = re.match(r"\s+([a-z][a-z0-9_]+)\b(.*)", line)
        if m and not m.group(1).startswith(("amdhsa", "p2align", "section", "type", "size", "globl", "protected", "text")):
            op = m.group(1)
            if line.lstrip().startswith("."):
                continue
            cur[1].append((op, loc[0], loc[1], m.group(2)))
    return blocks


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    blocks = parse(path, prefix)
    labels = {b[0]: k for k, b in enumerate(blocks)}
    tot = Counter()
    by_line = defaultdict(Counter)
    for label, ins in blocks:
        for op, f, ln, rest in ins:
            c = classify(op)
            tot[c] += 1
            by_line[(f, ln)][c] += 1
    print("total static:", dict(tot))
    if "--lines" in sys.argv:
        for (f, ln), c in sorted(by_line.items(), key=lambda kv: (str(kv[0][0]), kv[0][1])):
            print("%-14s %5d  valu %4d salu %4d smem %3d vmem %3d lds %3d" % (f, ln, c["valu"], c["salu"] + c["sctl"], c["smem"], c["vmem"], c["lds"]))
    if "--blocks" in sys.argv:
        for k, (label, ins) in enumerate(blocks):
            c = Counter(classify(op) for op, *_ in ins)
            back = [rest.strip() for op, f, ln, rest in ins if op.startswith(("s_cbranch", "s_branch")) and rest.strip() in labels and labels[rest.strip()] <= k]
            lines = Counter((f, ln) for op, f, ln, rest in ins if classify(op) == "valu")
            top = ", ".join("%s:%d x%d" % (f, ln, n) for (f, ln), n in lines.most_common(4))
            print("%-12s valu %4d salu %4d smem %3d vmem %3d lds %3d %s | %s" % (label, c["valu"], c["salu"] + c["sctl"], c["smem"], c["vmem"], c["lds"],
                                                                              ("LOOP->" + back[0]) if back else "", top))


def compose(path, spec_path, counts_path, measured_valu_per_wave=None):
    """The measured budget: static counts of the spec's phases (block index ranges of THIS build) x the counting build's trip counts."""
    import json
    spec = json.load(open(spec_path))
    cnt = json.load(open(counts_path))
    blocks = parse(path, spec["kernel"])
    static_total = sum(1 for _, ins in blocks for op, *_ in ins if classify(op) == "valu")
    if static_total != spec["static_valu_total"]:
        sys.exit("the spec speaks for a kernel of %d static vector instructions, this one has %d: map its blocks again (--blocks)" % (spec["static_valu_total"], static_total))
    L = float(cnt["launches"])
    ph, st = cnt["phases"], cnt["stats"]
    env = {"W": ph["waves"] / L, "S": ph["scan_waves"] / L, "WW": st["walking_waves"] / L,
           "Bb": ph["block_batches"] / L, "Bd": ph["block_dealt"] / L, "Bp": ph["block_deal_passes"] / L, "Bw": ph["block_deal_writes"] / L, "Bl": ph["block_lockstep_trips"] / L,
           "Rw": ph["rest_waves"] / L, "Rr": ph["rest_rounds"] / L, "Rd": ph["rest_dealt"] / L, "Rp": ph["rest_deal_passes"] / L, "Rwr": ph["rest_deal_writes"] / L,
           "Rl": ph["rest_lockstep_trips"] / L, "st": st["walk_steps"] / L, "lc": ph["walk_leaf_children"] / L, "lv": st["walk_leaves"] / L,
           "lh": ph["walk_leaf_hits"] / L, "lo": ph["walk_leaf_offers"] / L}
    per_block = []
    for label, ins in blocks:
        c = Counter(classify(op) for op, *_ in ins)
        per_block.append((c["valu"], c["salu"] + c["sctl"], c["smem"] + c["vmem"] + c["lds"]))
    rows = []
    tot_v = tot_s = 0.0
    for entry in spec["phases"]:
        name, lo, hi, expr = entry[0], entry[1], entry[2], entry[3]
        over = entry[4] if len(entry) > 4 else {}
        mult = [eval(expr, {}, env)] * (hi - lo + 1)
        for rng, e in over.items():
            a, _, b = rng.partition("-")
            for k in range(int(a), int(b or a) + 1):
                mult[k - lo] = eval(e, {}, env)
        sv = sum(per_block[k][0] for k in range(lo, hi + 1))
        dv = sum(per_block[k][0] * max(mult[k - lo], 0.0) for k in range(lo, hi + 1))
        ds = sum(per_block[k][1] * max(mult[k - lo], 0.0) for k in range(lo, hi + 1))
        rows.append((name, sv, expr, dv, ds))
        tot_v += dv
        tot_s += ds
    W = env["W"]
    print("| phase | static VALU | runs per launch | VALU per launch | per wave (of %d) | share | SALU per wave |" % round(W))
    print("|---|---|---|---|---|---|---|")
    for name, sv, expr, dv, ds in rows:
        print("| %s | %d | `%s` | %.3g | %.0f | %.1f %% | %.0f |" % (name, sv, expr, dv, dv / W, 100.0 * dv / tot_v, ds / W))
    print("| **sum** | %d | | **%.4g** | **%.0f** | | %.0f |" % (static_total, tot_v, tot_v / W, tot_s / W))
    if measured_valu_per_wave:
        print("\nmeasured (SQ_INSTS_VALU / SQ_WAVES over the same 20 launches): %.0f per wave; the table's rows sum to %.0f = %.3f of it" % (measured_valu_per_wave, tot_v / W, tot_v / W / measured_valu_per_wave))
    print("\ntrip counts per launch: " + ", ".join("%s %.0f" % kv for kv in sorted(env.items())))


if __name__ == "__main__":
    if "--compose" in sys.argv:
        k = sys.argv.index("--compose")
        compose(sys.argv[1], sys.argv[k + 1], sys.argv[k + 2], float(sys.argv[k + 3]) if len(sys.argv) > k + 3 else None)
    else:
        main()
