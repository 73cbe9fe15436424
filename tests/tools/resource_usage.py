#!/usr/bin/env python3
"""Compact table of a `make asm` resource-usage dump: kernel, SGPRs, VGPRs, AGPRs, spills, scratch, occupancy.
usage: resource_usage.py plssvm_amd/lib/asm/resource_usage_<tu>.txt [name filter]"""
import re
import subprocess
import sys

rows, cur = [], None
for line in open(sys.argv[1]):
    m = re.search(r"remark: .*?:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r":\d+:\d+: +(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    if flt and flt not in name:
        continue
    print(f"{name:70s} sgpr {r.get('TotalSGPRs', '?'):>3s} vgpr {r.get('VGPRs', '?'):>3s} agpr {r.get('AGPRs', '?'):>3s} spill v{r.get('VGPRs Spill', '?')} s{r.get('SGPRs Spill', '?')} scratch {r.get('ScratchSize [bytes/lane]', '?'):>4s} occ {r.get('Occupancy [waves/SIMD]', '?')}")
