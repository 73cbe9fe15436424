#!/usr/bin/env python3
"""The predict leg of bench.py on its own (other_workloads.predict): the program the rocprofv3 passes of tests/tools/profile_round.sh wrap.
usage: predict_bench.py [num_points [num_sv [calls]]]   -> one JSON line"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    npts = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    nsv = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
    calls = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    print(json.dumps(bench.predict_leg(42, 0, num_sv=nsv, num_points=npts, calls=calls)))
