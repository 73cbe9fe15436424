#!/usr/bin/env python3
"""Copies the scaling-factor DATA files the reference's own tests hold (tests/data/scaling_factors/**) into
tests/golden/scaling_factors/ and writes expected.json: what the reference's tests expect for each of them
(tests/detail/io/scaling_factors_parsing.cpp:44-183: the parsed interval / factors of the valid files, the error text of
every invalid one; "{real}" stands for the name of the real type).  Run in the build container only: the GPU box has no
/root/reference, it uses the committed copies.
usage: python tests/golden/make_scaling_fixture.py"""
import json
import os
import shutil

REF = "/root/reference/tests/data/scaling_factors"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scaling_factors")

EXPECTED = {
    "scaling_factors.txt": {"interval": [-1.4, 2.6], "factors": [[0, 0.0, 1.0], [1, 1.1, 2.1], [3, 3.3, 4.3], [4, 4.4, 5.4]]},
    "no_scaling_factors.txt": {"interval": [-1.4, 2.6], "factors": []},
    "invalid/too_many_scaling_interval_values.txt": {"error": "The interval to which the data points should be scaled must exactly have two values, but 3 were given!"},
    "invalid/too_few_scaling_interval_values.txt": {"error": "The interval to which the data points should be scaled must exactly have two values, but 1 were given!"},
    "invalid/inconsistent_scaling_interval_values.txt": {"error": "Inconsistent scaling interval specification: lower (1.4) must be less than upper (-2.6)!"},
    "invalid/no_header.txt": {"error": "The first line must only contain an 'x', but is \"-1.4 2.6\"!"},
    "invalid/too_few_lines.txt": {"error": "At least two lines must be present, but only 1 were given!"},
    "../empty.txt": {"error": "At least two lines must be present, but only 0 were given!"},
    "invalid/too_few_scaling_factor_values.txt": {"error": "Each line must contain exactly three values, but 2 were given!"},
    "invalid/too_many_scaling_factor_values.txt": {"error": "Each line must contain exactly three values, but 4 were given!"},
    "invalid/zero_based_scaling_factors.txt": {"error": "The scaling factors must be provided one-based, but are zero-based!"},
    "invalid/invalid_number.txt": {"error": "Can't convert 'a' to a value of type {real}!"},
}

if __name__ == "__main__":
    os.makedirs(os.path.join(OUT, "invalid"), exist_ok=True)
    table = {}
    for rel, exp in EXPECTED.items():
        name = "empty.txt" if rel == "../empty.txt" else rel
        shutil.copyfile(os.path.normpath(os.path.join(REF, rel)), os.path.join(OUT, name))
        table[name] = exp
    with open(os.path.join(OUT, "expected.json"), "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    print(f"wrote {len(table)} files + expected.json to {OUT}")
