#!/usr/bin/env python3
"""Developer tool: the "MEASURED key value" lines of a GPU suite's log (pytest -s, or the captured output of failing tests; tests/conftest.py
check_measured prints them) -> tests/golden/measured_bounds.json, the values the suite then holds later runs to (value <= factor x measured).
    python -m pytest tests -m gpu -q -s > gpurun_out/gputest.log;  python tools/collect_measured.py gpurun_out/gputest.log"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    values = {}
    for path in [a for a in sys.argv[1:] if not a.startswith("--")]:
        for line in open(path, errors="replace"):
            for m in re.finditer(r"MEASURED (\S+) ([-+0-9.eE]+|nan)", line):
                v = float(m.group(2))
                values[m.group(1)] = max(values.get(m.group(1), 0.0), v)       # several contexts measure the same quantity: keep the largest
    out = os.path.join(ROOT, "tests", "golden", "measured_bounds.json")
    # An existing bound is never RAISED silently (regenerating the fixture from a run that regressed would have moved the bars with it):
    # a larger measurement keeps the committed value and is reported, unless --allow-raise says the new one is wanted.
    allow_raise = "--allow-raise" in sys.argv
    refused = []
    if os.path.exists(out):
        old = json.load(open(out)).get("values", {})
        for k, v in old.items():
            if k in values and values[k] > v and not allow_raise:
                refused.append((k, v, values[k]))
                values[k] = v
            elif k not in values:
                values[k] = v                                   # (a quantity this log does not hold keeps its bound)
    for k, v, new in refused:
        print("kept %s = %.6g (this log measured %.6g: pass --allow-raise to take it)" % (k, v, new))
    with open(out, "w") as f:
        json.dump({"source": "MI355X, " + ", ".join(os.path.basename(p) for p in sys.argv[1:] if not p.startswith("--")), "note": "tests/conftest.py check_measured: value <= factor x these", "values": dict(sorted(values.items()))}, f, indent=1)
    print("wrote %s: %d quantities" % (out, len(values)))


if __name__ == "__main__":
    main()
