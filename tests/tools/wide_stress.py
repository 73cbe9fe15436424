#!/usr/bin/env python3
"""Long randomised cross-check of the panels-inside-a-tile kernels (generator and yardstick: tests/cross_check.py; a seeded slice of it runs in
`pytest -m gpu`).  usage: wide_stress.py [cases] [seed] [f64]"""
import os
import sys

TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.dirname(TESTS), TESTS]
import cross_check  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
f64 = len(sys.argv) > 3 and sys.argv[3] == "f64"
worst, flags = 0.0, 0
for i in range(cases):
    case = cross_check.wide_case(seed, i, f64)
    res = cross_check.run_case(case)
    worst = max(worst, res["err"])
    flags += 0 if res["ok"] else 1
    print(f"case {i:3d}: {cross_check.describe(case)} -> plane mode {res['gram_mode']}: {res['err']:7.2f} eps from float64 (generic kernel: {res['err_generic']:7.2f})"
          f"{'' if res['ok'] else '   <-- CHECK'}", flush=True)
print(f"worst: {worst:.2f} eps of the row's summands from the float64 product; {flags} case(s) flagged")
sys.exit(1 if flags else 0)
