#!/usr/bin/env python3
"""Long randomised cross-check of the resident-row-panel tile kernels (generator and yardstick: tests/cross_check.py; a seeded slice of it runs
in `pytest -m gpu`).  usage: narrow_stress.py [cases] [seed] [pair]   (pair: the 256-row workgroups on block pairs only)"""
import os
import sys

TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.dirname(TESTS), TESTS]
import cross_check  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
make_case = cross_check.pair_case if len(sys.argv) > 3 and sys.argv[3] == "pair" else cross_check.narrow_case
worst, flags = 0.0, 0
for i in range(cases):
    case = make_case(seed, i)
    res = cross_check.run_case(case)
    worst = max(worst, res["err"])
    flags += 0 if res["ok"] else 1
    print(f"case {i:3d}: {cross_check.describe(case)} -> (gram mode, symmetric) ({res['gram_mode']}, {res['symmetric']}): {res['err']:7.2f} eps from float64"
          f" (generic kernel: {res['err_generic']:7.2f}){'' if res['ok'] else '   <-- CHECK'}", flush=True)
print(f"worst: {worst:.2f} eps of the row's summands from the float64 product; {flags} case(s) flagged")
sys.exit(1 if flags else 0)
