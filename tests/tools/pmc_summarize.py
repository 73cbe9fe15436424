#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tests/tools/pmc_passes.sh for the implicit-matvec tile kernel.

usage: pmc_summarize.py <label> <pmc_outdir> [--json profiles/hbm_traffic.json --key c5_n1]

Per counter: mean/min/max over the launches of the tile kernel (the kernel whose name contains ``tile_matvec``), then the
derived figures DESIGN.md section 5 quotes.  Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM /
rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so the
fabric-side read bytes are 2 x 1024 x FETCH_SIZE.
"""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(outdir):
    per = defaultdict(lambda: defaultdict(float))  # counter -> dispatch -> value (summed over the rows of one dispatch)
    names = set()
    for f in glob.glob(os.path.join(outdir, "*", "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if "tile_matvec" not in row["Kernel_Name"]:
                    continue
                names.add(row["Kernel_Name"])
                per[row["Counter_Name"]][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
    return per, names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("label")
    ap.add_argument("outdir")
    ap.add_argument("--json")
    ap.add_argument("--key")
    ap.add_argument("--profile", default=None, help="name of the committed summary file the numbers come from")
    args = ap.parse_args()
    per, names = collect(args.outdir)
    if not per:
        print(f"{args.label}: no tile_matvec dispatches found under {args.outdir}", file=sys.stderr)
        return 1
    mean = {}
    for c in sorted(per):
        v = list(per[c].values())
        mean[c] = sum(v) / len(v)
        print(f"{args.label} {c:<28} launches={len(v):2d} mean={mean[c]:.6g} min={min(v):.6g} max={max(v):.6g}")
    for nm in sorted(names):
        print(f"{args.label} kernel {nm}")
    rd = 2.0 * 1024.0 * mean.get("FETCH_SIZE", float("nan"))
    wr = 1024.0 * mean.get("WRITE_SIZE", float("nan"))
    hit, miss = mean.get("TCC_HIT_sum", float("nan")), mean.get("TCC_MISS_sum", float("nan"))
    # SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs, SQ_BUSY_CYCLES over the shader engines x4 quad-cycles:
    # busy fraction = MFMA_BUSY / (GRBM_GUI_ACTIVE x 1024 SIMDs)
    clk = mean.get("GRBM_GUI_ACTIVE", float("nan"))
    mfma = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan"))
    # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs
    busy = mfma / (clk / 8.0 * 1024.0) if clk == clk and clk > 0 else float("nan")
    print(f"{args.label} => fabric-side read bytes/launch {rd:.4g} (corrected x2), write bytes/launch {wr:.4g}; L2 hit rate {hit / (hit + miss):.3f}; "
          f"MFMA busy fraction {busy:.3f} of SIMD-cycles; clock {clk / 8.0:.3g} cycles/launch; LDS bank conflict cycles {mean.get('SQ_LDS_BANK_CONFLICT', float('nan')):.3g}")
    if args.json and args.key and rd == rd:
        data = {}
        if os.path.isfile(args.json):
            with open(args.json) as fh:
                data = json.load(fh)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
        from bench import kernel_source_hash  # the stamp bench.py checks before it reports `traffic`

        data[args.key] = {"bytes": rd + wr, "read_bytes": rd, "write_bytes": wr, "kernel_sha": kernel_source_hash(code_only=False), "code_sha": kernel_source_hash(), "profile": args.profile or args.outdir,
                          "launches_per_matvec_note": "bytes are per tile-kernel LAUNCH as rocprofv3 counts them; bench.py multiplies by tile_launches_per_matvec"}
        with open(args.json, "w") as fh:
            json.dump(data, fh, indent=1, sort_keys=True)
            fh.write("\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
