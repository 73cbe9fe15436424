#!/usr/bin/env python3
"""DESIGN.md = docs/DESIGN.template.md with its R06_* placeholders filled from the committed measurement files of the round (profiles/r06_*), so that every number in
DESIGN.md traces to a file under profiles/.   usage: python tests/tools/fill_design.py [profiles-dir]   (writes DESIGN.md at the repository root)"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def last_json_line(path):
    with open(path) as f:
        lines = [ln for ln in f if ln.startswith("{")]
    return json.loads(lines[-1])


def main():
    prof = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles")
    b = last_json_line(os.path.join(prof, "r06_bench_default.json"))
    ow, roof = b["other_workloads"], b["roofline"]
    pr, pl = ow["predict"]["rbf"], ow["predict"]["linear"]
    e2e = b["e2e"]["train"]
    e2p = b["e2e"]["predict"]
    rep = {
        "R06_C5_MS": f"{b['ms_per_step']:.1f} ms", "R06_C5_KERN": f"{roof['avg_launch_ms']:.1f} ms = {roof['tile_launches_per_matvec']} band launches", "R06_C5_FRAC": f"{roof['frac']:.3f}",
        "R06_C5_VALUE": f"{b['value'] / 1e3:.0f}", "R06_C5_SETUP": f"{b['setup_ms']:.0f} ms",
        "R06_CPU": f"{b['cpu_baseline']['value']:.1f}", "R06_CPUR": f"{b['cpu_baseline_release']['value']:.1f}" if "cpu_baseline_release" in b else "n/a",
        "R06_PR_CALL": f"{pr['call_ms']:.1f} ms", "R06_PR_KERN": f"{pr['kernel_ms']:.2f} ms", "R06_PR_FIRST": f"{pr['first_launch_kernel_ms']:.2f} ms", "R06_PR_FRAC": f"{pr['frac']:.3f}",
        "R06_PR_FFRAC": f"{pr['first_launch_frac']:.3f}", "R06_PR_SETUP": f"{pr['setup_ms']:.1f} ms",
        "R06_PL_CALL": f"{pl['call_ms']:.1f} ms", "R06_PL_KERN": f"{pl['kernel_ms'] * 1e3:.0f} µs", "R06_PL_FRAC": f"{pl['frac']:.2f}", "R06_PL_TBPS": f"{pl['achieved'] / 1e3:.1f}",
        "R06_E2E_READ": f"{e2e['read_s'] * 1e3:.0f} ms", "R06_E2E_SETUP": f"{e2e['setup_ms']:.1f} ms", "R06_E2E_SOLVE": f"{e2e['solve_s'] * 1e3:.1f} ms", "R06_E2E_ITS": str(e2e["iterations"]),
        "R06_E2E_WRITE": f"{e2e['write_s'] * 1e3:.0f} ms",
        "R06_E2P_READ": f"{e2p['read_s'] * 1e3:.0f} ms", "R06_E2P_MODEL": f"{e2p['model_read_s'] * 1e3:.0f} ms", "R06_E2P_PREDICT": f"{e2p['predict_s'] * 1e3:.1f} ms",
        "R06_E2P_WRITE": f"{e2p['write_s'] * 1e3:.0f} ms",
    }
    res = ow["predict"].get("resident_predictor_rbf")
    if res:
        rep.update({"R06_RES_ALL": f"{res['batch_all_points']['resident_call_ms']:.1f} ms", "R06_ONE_ALL": f"{res['batch_all_points']['one_shot_call_ms']:.1f} ms",
                    "R06_RES_1K": f"{res['batch_1000_points']['resident_call_ms']:.2f} ms", "R06_ONE_1K": f"{res['batch_1000_points']['one_shot_call_ms']:.2f} ms",
                    "R06_HBM_ALL": (f"{res['batch_all_points']['batch_in_hbm_call_ms']:.1f} ms" if "batch_in_hbm_call_ms" in res["batch_all_points"] else "n/a")})
    else:
        rep.update({"R06_RES_ALL": "n/a", "R06_ONE_ALL": "n/a", "R06_RES_1K": "n/a", "R06_ONE_1K": "n/a", "R06_HBM_ALL": "n/a"})
    for name, unit in (("c2", "TFLOP/s"), ("c3", "TFLOP/s"), ("c4", "TFLOP/s")):
        w = ow[name]
        up = name.upper()
        rep[f"R06_{up}_MS"] = f"{w['ms_per_step']:.3f} ms" if w["ms_per_step"] < 1 else f"{w['ms_per_step']:.2f} ms"
        rep[f"R06_{up}_KERN"] = f"{w['avg_launch_ms']:.3f} ms" if w["avg_launch_ms"] < 1 else f"{w['avg_launch_ms']:.2f} ms"
        rep[f"R06_{up}_FRAC"] = f"{w['frac']:.3f}"
        rep[f"R06_{up}_VALUE"] = f"{w['value'] / 1e3:.0f} {unit}"
    rep["R06_C2_SETUP"] = f"{ow['c2']['setup_ms']:.1f} ms"
    # the c2 chain, launch by launch, from the rocprofv3 kernel statistics of the c2 bench command
    rows = {}
    with open(os.path.join(prof, "r06_rocprofv3_kernel_stats_bench_c2.csv"), newline="") as f:
        for row in csv.DictReader(f):
            rows[row["Name"]] = row
    chain = []
    total = 0.0
    for key, what in (("tile_matvec_f32_pair", "the tile kernel"), ("k_reduce_colslab", "mirrored column sums → K·v"), ("k_reduce_partials_sym", "row slabs → K·v"), ("k_Ad_and_dAd", "Ad, dᵀAd"),
                      ("k_update_x_r", "α; x += αd, r −= αAd; rᵀr partials"), ("k_finish_delta", "δ → mapped host word"), ("k_update_d", "d = βd + r; next records; clear K·v")):
        name = next((n for n in rows if key in n), None)
        if name is None:
            continue
        us = float(rows[name]["AverageNs"]) / 1e3
        if key != "tile_matvec_f32_pair":
            total += us
        chain.append(f"| `{key}` | {what} | {us:.1f} µs |")
    rep["R06_C2_CHAIN"] = "| launch | what | average duration |\n|---|---|---|\n" + "\n".join(chain) + f"\n\nThe six launches beside the tile kernel sum to {total:.1f} µs; with the ≈ 1.2 µs that separates dependent launches on one stream, ≈ {total + 7:.0f} µs per iteration."
    with open(os.path.join(prof, "r06_gram_mode_by_data.log")) as f:
        rep["R06_GRAM_TABLE"] = "```\n" + "".join(ln if len(ln) <= 160 else ln[:157] + "...\n" for ln in f if not ln.startswith("#")).rstrip() + "\n```"
    with open(os.path.join(ROOT, "docs", "DESIGN.template.md"), encoding="utf-8") as f:
        text = f.read()
    for key in sorted(rep, key=len, reverse=True):
        text = text.replace(key, rep[key])
    left = sorted({w for w in text.replace("\n", " ").split(" ") if w.startswith("R06_")})
    if left:
        raise SystemExit(f"placeholders left: {left}")
    with open(os.path.join(ROOT, "DESIGN.md"), "w", encoding="utf-8") as f:
        f.write("<!-- generated by tests/tools/fill_design.py from docs/DESIGN.template.md and profiles/r06_*: edit the template -->\n" + text)
    long = [i + 1 for i, ln in enumerate(text.split("\n")) if len(ln) > 160]
    print("DESIGN.md written;", "lines longer than 160 columns:", long if long else "none")


if __name__ == "__main__":
    main()
