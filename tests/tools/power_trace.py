#!/usr/bin/env python3
"""Developer tool (not a test): board power and shader clock while the tile kernels run, read from the amdgpu hwmon files of the device
(an ordinary user can read them; no counters, no profiler).  Is the chip at its power cap while the c5 kernel runs, and while the bare MFMA
loop runs?  Then time IS energy: a vector instruction costs what it costs wherever it stands, and the fraction of the nominal matrix-core
peak a kernel can reach on a box is set by what the box's power budget lets the MFMAs alone sustain.

    python tests/tools/power_trace.py [--points 1000000] [--steps 12]

Phases (each sampled at ~50 Hz, the first 30 % of every phase dropped as settling time):
  idle | f16x3 (default) | bf16x6 | native v_mfma_f32 | bare f16 MFMA loop from registers | the same with B fragments from LDS | fp64 (c4 shape)"""

import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from plssvm_amd import _capi, backend  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.hwmon import PowerSampler  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402


def fmt(a, unit):
    return "n/a" if a.size == 0 else f"median {np.median(a):7.1f}  min {a.min():7.1f}  max {a.max():7.1f} {unit} ({a.size} samples)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1000000)
    ap.add_argument("--features", type=int, default=128)
    ap.add_argument("--steps", type=int, default=12)
    a = ap.parse_args()

    smp = PowerSampler(0)
    if not smp.available:
        print(f"no hwmon power reading for HIP device 0 (PCI {smp.bus}) is visible to this user")
        return 1
    cap = smp.cap_watts()
    print(f"device: {_capi.device_name(0)}   PCI {smp.bus}   hwmon: {smp.hwmon}")
    print(f"power cap: {'n/a' if cap is None else f'{cap:.0f} W'}")
    smp.start()
    phases = []

    def phase(name, fn):
        t0 = time.time()
        extra = fn()
        t1 = time.time()
        phases.append((name, t0, t1, extra or ""))

    phase("idle", lambda: time.sleep(2.0))

    X, y = make_blobs_pm1(a.points, a.features, seed=42, dtype=np.float32)

    def cg(gram_mode, steps):
        def run():
            _capi.set_option("gram_mode", gram_mode)
            prob = backend.ResidentProblem(Parameter(kernel_type="rbf", cost=1.0), X)
            prob.cg_begin(y, 1e-30)
            prob.cg_step(1)
            prob.synchronize()
            t0 = time.time()
            prob.cg_step(steps)
            prob.synchronize()
            dt = (time.time() - t0) / steps
            info = prob.info()
            prob.close()
            # the phase window should hold the stepping only: move its start behind the set-up
            phases_fix.append(t0)
            return f"{dt * 1e3:8.2f} ms per CG iteration, tile kernels {info['matvec_kernel_ms']:.2f} ms per matvec"
        return run

    phases_fix = []

    def timed_phase(name, fn):
        phase(name, fn)
        n, _, t1, extra = phases[-1]
        phases[-1] = (n, phases_fix[-1], t1, extra)

    timed_phase("f16x3 (default)", cg(2, a.steps))
    timed_phase("bf16x6", cg(1, max(2, a.steps * 2 // 3)))
    timed_phase("native v_mfma_f32", cg(0, max(2, a.steps // 3)))
    _capi.set_option("gram_mode", 2)

    def bare(bits, label):
        def run():
            tf, ghz, nominal = C.c_double(), C.c_double(), C.c_double()
            t0 = time.time()
            _capi.check(_capi.lib.lssvm_mi355_measure_bf16_mfma_ceiling(C.c_int(0), C.c_int(bits), C.c_double(4000.0), C.byref(tf), C.byref(ghz), C.byref(nominal)))
            phases_fix.append(t0)
            return f"{tf.value:7.1f} TFLOP/s = {tf.value / nominal.value:.3f} of nominal, in-kernel clock {ghz.value:.2f} GHz"
        return run

    f16_bit = 2
    timed_phase("bare f16 MFMA loop, operands in registers", bare(0 | f16_bit, ""))
    timed_phase("bare f16 MFMA loop, B fragments from LDS", bare(1 | f16_bit, ""))

    X64, y64 = make_blobs_pm1(100000, 64, seed=42, dtype=np.float64)

    def fp64():
        prob = backend.ResidentProblem(Parameter(kernel_type="polynomial", degree=3, cost=1.0), X64)
        prob.cg_begin(y64, 1e-30)
        prob.cg_step(20)
        prob.synchronize()
        t0 = time.time()
        prob.cg_step(300)
        prob.synchronize()
        dt = (time.time() - t0) / 300
        prob.close()
        phases_fix.append(t0)
        return f"{dt * 1e3:8.2f} ms per CG iteration"

    timed_phase("fp64 polynomial 100 000 x 64 (v_mfma_f64)", fp64)
    phase("idle again", lambda: time.sleep(1.0))

    smp.stop()
    for name, t0, t1, extra in phases:
        pw, fq = (np.array(x) for x in smp.window(t0, t1))
        print(f"{name:46s} {t1 - t0:6.1f} s | power {fmt(pw, 'W')} | sclk {fmt(fq, 'GHz')} | {extra}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
