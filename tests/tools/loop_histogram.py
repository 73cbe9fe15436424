#!/usr/bin/env python3
"""Instruction histogram of the steady-state tile loop of one kernel in a `make asm` dump:
   tests/tools/loop_histogram.py <file.s> <mangled-name prefix>   (the innermost loop with the most MFMAs is taken)"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and l.rstrip().split(";")[0].rstrip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
best = None
for i, l in enumerate(body):
    m = re.match(r"\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
    if m and labels.get(m.group(1), i) < i:
        lo = labels[m.group(1)]
        n = sum("v_mfma" in x for x in body[lo:i])
        if best is None or n > best[0]:
            best = (n, lo, i)
n, lo, hi = best
hist = collections.Counter(l.split()[0] for l in body[lo:hi + 1] if l.startswith("\t") and not l.strip().startswith((";", ".")))
valu = sum(c for k, c in hist.items() if k.startswith("v_") and "mfma" not in k)
print(f"loop lines {lo}..{hi}: {n} MFMA, {valu} other vector-ALU, {sum(c for k, c in hist.items() if k.startswith('ds_'))} LDS, "
      f"{sum(c for k, c in hist.items() if k.startswith('global_'))} VMEM, {sum(c for k, c in hist.items() if k.startswith('s_'))} scalar/wait")
for k, c in hist.most_common():
    print(f"{c:6d} {k}")
