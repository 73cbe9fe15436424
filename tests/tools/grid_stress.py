#!/usr/bin/env python3
"""Long run of the randomised cross-checks of rbf with a large exponent scale (tests/cross_check.py: grid_case) with another seed:
    python tests/tools/grid_stress.py [--cases 300] [--seed 31]
prints every flagged case, the worst error per path (gram mode 3 = grid planes, 0 with rbf_direct = direct kernel) and how many cases took which."""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import cross_check  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=31)
    ap.add_argument("--family", default="grid", choices=["grid", "grid_pair", "rect"], help="grid: rbf on grid planes (round 5); grid_pair: its 256-row form; rect: the rectangular predict kernel (round 6)")
    args = ap.parse_args()
    worst = collections.defaultdict(float)
    count = collections.Counter()
    flagged = 0
    top = []
    for i in range(args.cases):
        case = {"grid": cross_check.grid_case, "grid_pair": cross_check.grid_pair_case, "rect": cross_check.rect_case}[args.family](args.seed, i)
        res = cross_check.run_rect_case(case) if args.family == "rect" else cross_check.run_case(case)
        key = f"gram mode {res['gram_mode']}"
        count[key] += 1
        worst[key] = max(worst[key], res["err"])
        top = sorted(top + [(res["err"], key, cross_check.describe(case))], reverse=True)[:5]
        if not res["ok"]:
            flagged += 1
            print("FLAGGED", cross_check.describe(case), res, flush=True)
    for err, key, text in top:
        print(f"  {err:6.2f} eps  {key}  {text}")
    print(f"{args.family}: {args.cases} cases, seed {args.seed}: {flagged} flagged; " + "; ".join(f"{k}: {count[k]} cases, worst {worst[k]:.2f} eps (generic kernel as the yardstick inside run_case)" for k in sorted(count)))


if __name__ == "__main__":
    main()
