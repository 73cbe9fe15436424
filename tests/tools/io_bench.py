#!/usr/bin/env python3
"""File-format throughput of the native readers / writers behind the C ABI (csrc/model_io.hpp, libsvm_reader.hpp) at the size of BASELINE.json's configs[4]:
a 1 000 000 x 128 fp32 model and data file written and read through the public Python entry points (Model.save / Model.load, write_libsvm_data /
parse_libsvm_data), host code only -- no GPU needed.  VERDICT r05 item 1: each in < 10 s on 8 cores (the per-element Python loops took ~90 s / ~100 s).
usage: io_bench.py [num_points [num_features [directory]]]    (needs ~6 GB of free disk or tmpfs and ~4 GB of memory)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from plssvm_amd import _capi  # noqa: E402,F401  (loads the library: its start-up is not what is timed below)
from plssvm_amd.data_set import DataSet  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.io_libsvm import parse_libsvm_data, write_libsvm_data  # noqa: E402
from plssvm_amd.model import Model  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    out = sys.argv[3] if len(sys.argv) > 3 else "/tmp"
    print(f"{n} x {d} fp32, {os.cpu_count()} hardware threads, files under {out}")
    X, y = make_blobs_pm1(n, d, seed=1, dtype=np.float32)
    labels = [int(v) for v in y]
    t = time.perf_counter()
    ds = DataSet(X, labels, real_type=np.float32)
    print(f"DataSet(X, labels)        {time.perf_counter() - t:6.2f} s")
    m = Model(Parameter(kernel_type="rbf", gamma=1.0 / d), ds, alpha=np.random.default_rng(0).standard_normal(n).astype(np.float32), rho=0.5)
    mf, df = os.path.join(out, "io_bench.model"), os.path.join(out, "io_bench.libsvm")
    for rep in range(2):
        t = time.perf_counter()
        m.save(mf)
        tw = time.perf_counter() - t
        size = os.path.getsize(mf)
        print(f"Model.save                {tw:6.2f} s   {size / 1e9:.2f} GB  {size / tw / 1e9:.2f} GB/s")
    t = time.perf_counter()
    m2 = Model.load(mf, real_type=np.float32)
    print(f"Model.load                {time.perf_counter() - t:6.2f} s")
    order = m.class_order().astype(np.int64)
    assert m2.data.data().shape == (n, d) and np.allclose(m2.data.data()[::997], X[order][::997], rtol=1e-9) and np.allclose(m2.alpha[::997], m.alpha[order][::997], rtol=1e-9)
    t = time.perf_counter()
    write_libsvm_data(df, X, labels=labels)
    print(f"write_libsvm_data         {time.perf_counter() - t:6.2f} s   {os.path.getsize(df) / 1e9:.2f} GB")
    t = time.perf_counter()
    X2, l2 = parse_libsvm_data(df, dtype=np.float32)
    print(f"parse_libsvm_data         {time.perf_counter() - t:6.2f} s")
    assert X2.shape == (n, d) and np.allclose(X2[::997], X[::997], rtol=1e-9) and l2[:5] == [float(v) for v in labels[:5]]
    os.remove(mf)
    os.remove(df)


if __name__ == "__main__":
    main()
