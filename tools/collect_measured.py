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
    for path in sys.argv[1:]:
        for line in open(path, errors="replace"):
            for m in re.finditer(r"MEASURED (\S+) ([-+0-9.eE]+|nan)", line):
                v = float(m.group(2))
                values[m.group(1)] = max(values.get(m.group(1), 0.0), v)       # several contexts measure the same quantity: keep the largest
    out = os.path.join(ROOT, "tests", "golden", "measured_bounds.json")
    with open(out, "w") as f:
        json.dump({"source": "MI355X, " + ", ".join(os.path.basename(p) for p in sys.argv[1:]), "note": "tests/conftest.py check_measured: value <= factor x these", "values": dict(sorted(values.items()))}, f, indent=1)
    print("wrote %s: %d quantities" % (out, len(values)))


if __name__ == "__main__":
    main()
