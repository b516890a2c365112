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


if __name__ == "__main__":
    main()
