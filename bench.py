#!/usr/bin/env python3
"""bench.py -- CG-iteration throughput of the MI355X LS-SVM backend on BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is ONE CG iteration (one implicit kernel-matrix--vector product + the CG vector updates, plus the residual
refresh every 50th iteration, src/plssvm/backends/OpenMP/csvm.cpp:125-166) on synthetic data that is already resident in
HBM when the timed region starts.  With N > 1 the implicit matrix is row-block sharded over the GPUs and every step contains
ONE exchange of the partial K*d vectors (default symmetric variant: an RCCL all-reduce of n reals; full-square variant: an
all-gather of the slices); the problem size is fixed, i.e. STRONG scaling.

Three ways to run N > 1:
  * under torchrun (the driver's way): one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE from the environment;
  * `python bench.py --gpus N` typed as is: the parent starts N child ranks (one process per GPU) BEFORE it touches the GPU
    itself, waits for them and exits with their worst code; rank 0 prints the JSON line;
  * `python bench.py --gpus N --single-process`: ONE process drives all N devices through lssvm_mi355_problem_create_multi
    (the mode behind plssvm::csvm); `--devices 0,0` lists the ordinals explicitly (repeats allowed: shards share a device,
    which is how the sharded path runs on a one-GPU box -- n_gpus then still reports the distinct devices).

Prints ONE JSON line on rank 0.  `value` = effective K*d GFLOP/s = 2 * n^2 * d * K / t (n = N_points - 1; full square, no
symmetry credit); `cg_iters_per_s` is the other half of BASELINE.json's metric.
"""

from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")  # idle OpenMP workers of the cpu_baseline leg sleep instead of spinning

# BASELINE.json configs[1..4]
WORKLOADS = {
    "c2": dict(n=50_000, d=128, kernel="rbf", dtype="float32", desc="configs[1]: 50000x128 rbf gamma=1/128 fp32"),
    "c3": dict(n=200_000, d=256, kernel="linear", dtype="float32", desc="configs[2]: 200000x256 linear fp32"),
    "c4": dict(n=100_000, d=64, kernel="polynomial", dtype="float64", desc="configs[3]: 100000x64 polynomial degree=3 fp64"),
    "c5": dict(n=1_000_000, d=128, kernel="rbf", dtype="float32", desc="configs[4]: 1000000x128 rbf gamma=1/128 fp32"),
}
# dense matrix-core peaks of the arithmetic type (MI355X_MICROARCH.md, Chip-level parameters / Matrix cores)
PEAK_TFLOPS = {"float32": 157.3, "float64": 78.6, "bf16": 16 * 157.3}  # dense MFMA peaks, MI355X_MICROARCH.md (bf16 = 16 x the f32 MFMA rate, ~2.5 PF)

# the sources that decide how many bytes the tile kernel moves: profiles/hbm_traffic.json carries their hash, a stale entry is dropped
TRAFFIC_SOURCES = ["plssvm_amd/csrc/lssvm_tile_f32_split.hip.hpp", "plssvm_amd/csrc/lssvm_tile_f32_pair.hip.hpp", "plssvm_amd/csrc/lssvm_s6w_groups.inc", "plssvm_amd/csrc/lssvm_tile_f32.hip.hpp",
                   "plssvm_amd/csrc/lssvm_tile_f32_wide.hip.hpp", "plssvm_amd/csrc/lssvm_tile_f64.hip.hpp", "plssvm_amd/csrc/lssvm_tile_f64_wide.hip.hpp", "plssvm_amd/csrc/lssvm_device_common.hip.hpp",
                   "plssvm_amd/csrc/lssvm_kernels.hip.hpp", "plssvm_amd/csrc/lssvm_problem.hip", "plssvm_amd/csrc/lssvm_predict.hip", "plssvm_amd/csrc/lssvm_solver.hip", "plssvm_amd/csrc/lssvm_exchange.hip", "plssvm_amd/csrc/lssvm_problem.hip.hpp", "plssvm_amd/csrc/lssvm_types.hpp",
                   "plssvm_amd/csrc/tile_launch_f32.hip", "plssvm_amd/csrc/tile_launch_f32h.hip", "plssvm_amd/csrc/tile_launch_f32s.hip", "plssvm_amd/csrc/tile_launch_f32d.hip",
                   "plssvm_amd/csrc/tile_launch_f32x.hip", "plssvm_amd/csrc/tile_launch_f64.hip", "plssvm_amd/csrc/tile_launch_f64x.hip"]  # (the launch and split sources belong here too)


_C_TOKENS = None


def strip_comments(text: str) -> str:
    """C / C++ source without comments and with every run of white space collapsed (string and character literals are kept as they are)."""
    import re

    global _C_TOKENS
    if _C_TOKENS is None:
        _C_TOKENS = re.compile(r'//[^\n]*|/\*.*?\*/|"(?:\\.|[^"\\])*"|\'(?:\\.|[^\'\\])*\'', re.S)
    code = _C_TOKENS.sub(lambda m: " " if m.group(0).startswith("/") else m.group(0), text)
    return " ".join(code.split())


def kernel_source_hash(code_only: bool = True) -> str:
    """sha256 (16 hex digits) over the kernel and launch sources: of the CODE (comments and white space stripped), so that a stamp in
    profiles/hbm_traffic.json stays valid across comment-only edits and goes stale with the first changed token; code_only=False: of the raw bytes
    (the stamp of the entries written before this distinction existed, kept beside the new one)."""
    h = hashlib.sha256()
    for rel in TRAFFIC_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            raw = f.read()
        h.update(strip_comments(raw.decode("utf-8", errors="replace")).encode() if code_only else raw)
    return h.hexdigest()[:16]


def measured_traffic(key: str):
    """HBM-side bytes per tile-kernel launch from the committed rocprofv3 PMC passes (tests/tools/profile_round.sh writes
    profiles/hbm_traffic.json).  PMC counters cannot be read from inside an un-profiled run, so the number is taken from the
    file -- but ONLY while the kernel sources still hash to what was profiled; otherwise `traffic` is null."""
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(tfile) as f:
            table = json.load(f)
    except Exception:
        return None, "profiles/hbm_traffic.json missing"
    entry = table.get(key)
    if entry is None:
        return None, f"no PMC pass committed for {key}"
    if isinstance(entry, dict):
        if "code_sha" in entry:  # (comments and white space do not count)
            if entry["code_sha"] != kernel_source_hash():
                return None, f"stale: PMC pass was taken on kernel code {entry['code_sha']}, this is {kernel_source_hash()}"
            return entry.get("bytes"), f"{entry.get('profile', 'profiles/')} (kernel code {entry['code_sha']}, comments and white space stripped)"
        if entry.get("kernel_sha") != kernel_source_hash(code_only=False):
            return None, f"stale: PMC pass was taken on kernel sources {entry.get('kernel_sha')}, these are {kernel_source_hash(code_only=False)}"
        return entry.get("bytes"), f"{entry.get('profile', 'profiles/')} (kernel sources {entry.get('kernel_sha')})"
    return None, "unstamped entry (taken before the kernel-source stamp existed)"


def physical_cores() -> int:
    """Physical cores of the box (distinct (package, core) pairs of /proc/cpuinfo) -- `cpu_baseline.cores`; os.cpu_count() counts hardware threads
    (256 on the 2 x 64-core host of the GPU box)."""
    try:
        pairs, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
        if phys is not None and core is not None:
            pairs.add((phys, core))
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return os.cpu_count() or 1


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(X, y, kernel, sample_rows, iters):
    """The reference's own OpenMP kernels (where oracle/_ref was built), else the CPU oracle, on a bounded sample of the workload.
    Line A = the reference's default build type (RelWithDebInfo: -O2, CMakeLists.txt:23); line B = its Release flags
    (-O3 -ffast-math, CMakeLists.txt:100-104, with -march=x86-64-v3 standing in for -march=native: the library is built off-box)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    ns = min(sample_rows, X.shape[0])
    Xs, ys = X[:ns], y[:ns]
    threads = os.cpu_count() if oracle_lib.have_ref() else oracle_lib.oracle().num_threads()  # OpenMP's default: one thread per hardware thread
    cores = min(physical_cores(), threads)

    def run(impl, kind, flags):
        t0 = time.perf_counter()
        _, _, info = impl.solve(kernel, Xs, ys, 1e-30, iters, gamma=1.0 / X.shape[1], degree=3, coef0=0.0, cost=1.0)
        wall = time.perf_counter() - t0
        its = int(info["iterations"])
        # the solve runs its + 1 implicit matvecs (one for the initial residual); price an iteration as wall / (its + 1)
        t_iter = wall / (its + 1)
        flop = 2.0 * (ns - 1) ** 2 * X.shape[1]
        return {"value": flop / t_iter / 1e9, "unit": "GFLOP/s", "cores": cores, "threads": threads, "kind": kind, "cpu_model": cpu_model(), "build_flags": flags,
                "sample": f"first {ns} rows of the workload, {its} CG iterations + initial residual ({its + 1} implicit matvecs), "
                          f"{t_iter * 1e3:.1f} ms per matvec; effective GFLOP/s = 2*n^2*d / t is size independent",
                "ms_per_step_at_sample": t_iter * 1e3}

    if oracle_lib.have_ref():
        a = run(oracle_lib.ref(), "reference", "-O2 -fopenmp (the reference's default build type RelWithDebInfo, CMakeLists.txt:23)")
    else:
        a = run(oracle_lib.oracle(), "port", "-O2 -fopenmp -ffp-contract=off (oracle/Makefile)")
    b = None
    if oracle_lib.have_ref_release():
        b = run(oracle_lib.ref_release(), "reference", "-O3 -ffast-math -march=x86-64-v3 -fopenmp (the reference's Release flags, CMakeLists.txt:100-104; "
                                                       "x86-64-v3 instead of -march=native because the library is built off-box)")
    return a, b


def kernel_ms_between(i0, i1):
    """(matvecs enqueued, matvecs timed, average tile-kernel time of ONE implicit matvec in ms) between two lssvm_mi355_problem_info snapshots.  Short
    matvecs are event-bracketed by sampling (every 8th); the average is over the TIMED ones only -- their summed time over their count (the
    earlier form weighted the samples with the count of all matvecs)."""
    launches = int(i1["matvec_launches"] - i0["matvec_launches"])
    timed = int(i1["matvec_timed"] - i0["matvec_timed"])
    total = float(i1["matvec_kernel_ms_total"] - i0["matvec_kernel_ms_total"])
    return launches, timed, (total / timed if timed > 0 else 0.0)


def plane_products_of(gram_mode: int) -> float:
    return {0: 1.0, 1: 6.0, 2: 3.0, 3: 6.0}[gram_mode]  # (3: rbf on f16 grid planes, six plane products)


def short_leg(name, steps, warmup, seed, device):
    """One short leg of another BASELINE workload on the same device (reported as other_workloads.<name>): the same timed region as the
    headline -- `steps` CG iterations after `warmup`, data resident -- and the same pricing of the tile kernel (`frac` = algorithmic flop of the kernel as
    executed on the matrix cores / its peak)."""
    import numpy as np

    from plssvm_amd import backend
    from plssvm_amd.datagen import make_blobs_pm1
    from plssvm_amd.parameter import Parameter
    from plssvm_amd.sharding import triangle_share, work_share

    wl = WORKLOADS[name]
    N, d = wl["n"], wl["d"]
    dt = np.dtype(wl["dtype"])
    X, y = make_blobs_pm1(N, d, seed=seed, dtype=dt)
    prm = Parameter(kernel_type=wl["kernel"], degree=3, gamma=None, coef0=0.0, cost=1.0)
    with backend.ResidentProblem(prm, X, device=device) as prob:
        prob.cg_begin(y, 1e-30)
        prob.cg_step(warmup)
        prob.synchronize()
        i0 = prob.info()
        t0 = time.perf_counter()
        prob.cg_step(steps)
        prob.synchronize()
        elapsed = time.perf_counter() - t0
        i1 = prob.info()
    done = int(i1["iterations"] - i0["iterations"])
    launches, timed, kern_ms = kernel_ms_between(i0, i1)
    n = N - 1
    symmetric = bool(i1.get("symmetric", 0))
    sq_mac, exe_mac = work_share(n, 1, 0, symmetric)
    use_mac = triangle_share(n, 1, 0) if symmetric else sq_mac
    gram_mode = int(i1.get("gram_mode", 0))
    pp = plane_products_of(gram_mode)
    peak = PEAK_TFLOPS["bf16"] if gram_mode != 0 else PEAK_TFLOPS[wl["dtype"]]
    kern_s = kern_ms * 1e-3
    return {"workload": wl["desc"], "steps": done, "warmup": warmup, "setup_ms": float(i1.get("setup_ms", 0.0)), "ms_per_step": elapsed / max(done, 1) * 1e3, "value": 2.0 * n * n * d * done / elapsed / 1e9, "unit": "GFLOP/s",
            "avg_launch_ms": kern_ms, "launches": launches, "launches_timed": timed, "tile_launches_per_matvec": int(i1.get("tile_launches_per_matvec", 1)),
            "frac": (2.0 * use_mac * d * pp / kern_s / 1e12 / peak) if kern_ms > 0 else None,
            "executed_frac": (2.0 * exe_mac * d * pp / kern_s / 1e12 / peak) if kern_ms > 0 else None, "peak": peak,
            "gram_mode": {0: "native", 1: "bf16x6", 2: "f16x3", 3: "f16 grid planes"}[gram_mode], "dtype": "f32" if wl["dtype"] == "float32" else "f64", "symmetric": symmetric}


def predict_leg(seed, device, num_sv=50_000, num_points=200_000, d=128, calls=3):
    """other_workloads.predict: csvm::predict_values (include/plssvm/csvm.hpp:204-208; GPU recipe gpu_csvm.hpp:656-730,
    HIP/predict_kernel.hip.hpp:63-117) on `num_points` points against `num_sv` support vectors, rbf fp32 -- a RECTANGULAR instance of the tile kernel -- and the linear
    kernel through w.  The entry point is one-shot like the reference's (it uploads both point sets and prepares them on every call), so the line carries two rates:
    `kernel` (HIP events around the product kernel alone; `frac` prices it like the solve's tile kernel: 2 * num_sv * num_points * d multiply-adds-as-flop x plane
    products over the 16-bit matrix-core peak) and `call` (host wall clock of the whole call from host buffers: PCIe and set-up included)."""
    import numpy as np

    from plssvm_amd import backend
    from plssvm_amd.datagen import make_blobs_pm1
    from plssvm_amd.parameter import Parameter

    X, _ = make_blobs_pm1(num_sv + num_points, d, seed=seed + 1, dtype=np.float32)
    sv, pts = np.ascontiguousarray(X[:num_sv]), np.ascontiguousarray(X[num_sv:])
    alpha = np.random.default_rng(seed).standard_normal(num_sv).astype(np.float32)
    out = {"workload": f"predict_values: {num_points} points x {num_sv} support vectors x {d} features, fp32", "calls": calls,
           "timing": "kernel_ms / frac: the product kernel once the chip's clocks have settled -- every call launches it 4 times (LSSVM_MI355_PREDICT_REPEAT, a measurement aid of "
                     "the library: same slabs, same result) and HIP events bracket the LAST launch, like the W warm-up steps in front of the K timed steps of the headline; "
                     "first_launch_*: the single launch of an ordinary call, which comes behind the set-up's host-side gaps and runs while the firmware is still raising the "
                     "clock (5-10 % slower); call_ms: host wall clock of an ordinary call from host buffers (uploads and set-up included)"}
    for kernel in ("rbf", "linear"):
        prm = Parameter(kernel_type=kernel, gamma=None, cost=1.0)
        infos, settled, w = [], [], None
        for k in range(2 * calls + 1):  # the first call is the warm-up (code-object load, first allocations); then ordinary calls and repeated-launch calls in turn
            info = {}
            repeated = k > 0 and k % 2 == 0
            if repeated:
                os.environ["LSSVM_MI355_PREDICT_REPEAT"] = "4"
            try:
                values, w = backend.predict_values(prm, sv, alpha, 0.25, w, pts, info_out=info)
            finally:
                os.environ.pop("LSSVM_MI355_PREDICT_REPEAT", None)
            if k > 0:
                (settled if repeated else infos).append(info)
        first_ms = sum(i["kernel_ms"] for i in infos) / len(infos)
        kern_ms = sum(i["kernel_ms"] for i in settled) / len(settled) if kernel == "rbf" else first_ms  # (the linear kernel's w.x pass is not repeated by the aid)
        call_ms = sum(i["total_ms"] for i in infos) / len(infos)
        leg = {"kernel_ms": kern_ms, "first_launch_kernel_ms": first_ms, "call_ms": call_ms, "setup_ms": sum(i["setup_ms"] for i in infos) / len(infos),
               "points_per_s_call": num_points / (call_ms * 1e-3), "finite": bool(np.all(np.isfinite(values)))}
        if kernel == "rbf":
            gm = int(infos[-1]["gram_mode"])
            pp = plane_products_of(gm)
            peak = PEAK_TFLOPS["bf16"] if gm != 0 else PEAK_TFLOPS["float32"]
            flop = 2.0 * num_sv * num_points * d
            leg.update({"gram_mode": {0: "native", 1: "bf16x6", 2: "f16x3", 3: "f16 grid planes"}[gm], "bound": "mfma", "peak": peak, "algorithmic_flop_per_launch": flop * pp,
                        "achieved": flop * pp / (kern_ms * 1e-3) / 1e12, "frac": flop * pp / (kern_ms * 1e-3) / 1e12 / peak, "avg_launch_ms": kern_ms,
                        "first_launch_frac": flop * pp / (first_ms * 1e-3) / 1e12 / peak,
                        "value": flop / (call_ms * 1e-3) / 1e9, "value_kernel_only": flop / (kern_ms * 1e-3) / 1e9, "unit": "GFLOP/s (2 * num_sv * num_points * d per call)",
                        "f16_row_rel_error": infos[-1]["f16_row_rel_error"]})
        else:
            # w.x per point: one pass over the points, HBM bound -- algorithmic bytes = the points once + the outputs
            nbytes = float(num_points) * d * 4 + num_points * 4
            leg.update({"bound": "hbm", "peak": 8000.0, "unit": "GB/s", "achieved": nbytes / (kern_ms * 1e-3) / 1e9, "frac": nbytes / (kern_ms * 1e-3) / 1e9 / 8000.0,
                        "avg_launch_ms": kern_ms, "algorithmic_bytes_per_launch": nbytes, "note": "w (calculate_w) is computed by the first call and handed back to the later ones"})
        out[kernel] = leg
    # the model RESIDENT in HBM (lssvm_mi355_predictor_*, round 6): what a call costs when the support vectors are not uploaded and prepared again -- the whole batch, and a
    # small batch of 1 000 points, whose one-shot call is almost entirely the support vectors' set-up
    prm = Parameter(kernel_type="rbf", gamma=None, cost=1.0)
    res = {}
    with backend.Predictor(prm, sv, alpha, 0.25) as pred:
        for label, batch in (("batch_all_points", pts), ("batch_1000_points", pts[:1000])):
            calls_ms, resident = [], None
            for k in range(calls + 1):
                info = {}
                pred.predict(batch, info_out=info)
                if k > 0:
                    calls_ms.append(info["total_ms"])
                    resident = int(info["resident"])
            one = {}
            backend.predict_values(prm, sv, alpha, 0.25, None, batch, info_out=one)
            res[label] = {"points": int(batch.shape[0]), "resident_call_ms": sum(calls_ms) / len(calls_ms), "one_shot_call_ms": one["total_ms"], "ran_resident": resident,
                          "points_per_s_resident": batch.shape[0] / (sum(calls_ms) / len(calls_ms) * 1e-3)}
            try:  # the batch and the values in HBM as well (LSSVM_MEM_DEVICE): no PCIe in the call
                import torch

                pd = torch.from_numpy(np.ascontiguousarray(batch)).cuda()
                od = torch.zeros(batch.shape[0], dtype=torch.float32, device="cuda")
                torch.cuda.synchronize()
                hbm_ms = []
                for k in range(calls + 1):
                    info = {}
                    pred.predict_device(pd.data_ptr(), batch.shape[0], od.data_ptr(), info_out=info)
                    if k > 0:
                        hbm_ms.append(info["total_ms"])
                res[label].update({"batch_in_hbm_call_ms": sum(hbm_ms) / len(hbm_ms), "points_per_s_batch_in_hbm": batch.shape[0] / (sum(hbm_ms) / len(hbm_ms) * 1e-3)})
                del pd, od
            except Exception as e:  # noqa: BLE001  (torch is plumbing here: the leg stands without it)
                res[label]["batch_in_hbm_error"] = f"{type(e).__name__}: {e}"
    out["resident_predictor_rbf"] = res
    return out


def e2e_leg(seed, workdir=None):
    """The reference's tracker reports the file -> model wall clock beside cg/total_runtime (main_train.cpp:24-70, csvm.cpp:167-176): `python -m plssvm_amd.train` on a
    configs[1]-shaped LIBSVM file (50 000 x 128, rbf, fp32, eps 1e-3 -- the reference's default), then `python -m plssvm_amd.predict` of the same file with the model --
    run in this process through the command lines' own entry points, phases from plssvm_amd.cli.LAST_TIMINGS.
    This process has run every other leg by now, and a full collection of Python's cyclic garbage collector over what they left behind took 51 ms when it fell into the
    6 ms predict phase: the command lines in their own processes have no such heap, so it is set aside first (gc.freeze) and the collector's pauses during the two
    commands are reported beside the phases."""
    import gc
    import tempfile

    from plssvm_amd import cli
    from plssvm_amd.datagen import generate_libsvm_file

    out = {}
    pauses, started = [], [0.0]

    def on_gc(phase, info):
        if phase == "start":
            started[0] = time.perf_counter()
        else:
            pauses.append({"generation": int(info.get("generation", -1)), "s": time.perf_counter() - started[0]})

    with tempfile.TemporaryDirectory(dir=workdir) as tmp:
        data_file, model_file, pred_file = os.path.join(tmp, "c2.libsvm"), os.path.join(tmp, "c2.libsvm.model"), os.path.join(tmp, "c2.libsvm.predict")
        t0 = time.perf_counter()
        generate_libsvm_file(data_file, 50_000, 128, seed=seed)
        out["generate_and_write_data_s"] = time.perf_counter() - t0
        out["data_bytes"] = os.path.getsize(data_file)
        gc.collect()
        gc.freeze()
        gc.callbacks.append(on_gc)
        try:
            rc = cli.train_main(["-t", "2", "--use_float_as_real_type", "-e", "0.001", "-q", data_file, model_file])
            if rc != 0:
                return {"error": f"plssvm_amd.train returned {rc}"}
            out["train"] = {k: v for k, v in cli.LAST_TIMINGS.items() if k != "task"}
            rc = cli.predict_main(["--use_float_as_real_type", "-q", data_file, model_file, pred_file])
            if rc != 0:
                return {"error": f"plssvm_amd.predict returned {rc}"}
            out["predict"] = {k: v for k, v in cli.LAST_TIMINGS.items() if k != "task"}
        finally:
            gc.callbacks.remove(on_gc)
            gc.unfreeze()
    out["python_gc"] = {"pauses_s": sum(p["s"] for p in pauses), "full_collections": sum(1 for p in pauses if p["generation"] == 2)}
    out["command"] = "python -m plssvm_amd.train -t 2 --use_float_as_real_type -e 0.001 c2.libsvm ; python -m plssvm_amd.predict --use_float_as_real_type c2.libsvm c2.libsvm.model"
    return out


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N child ranks, one process per GPU.  Nothing in THIS process has touched
    the GPU (no HIP call, no torch.cuda call), and no process is ever replaced by another: the children are ordinary child
    processes and the parent only waits.  Rank 0 inherits stdout and prints the JSON line."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=(_RESULT_FD if r == 0 else subprocess.DEVNULL)))  # (rank 0 writes the result line: it gets the ORIGINAL stdout)
    # poll ALL children: a rank that dies leaves its siblings blocked in the exchange (RCCL, or the IPC wait with its 600 s default), so
    # on the first non-zero exit the others get SIGTERM, a short grace period, then SIGKILL -- these are children this parent started itself
    worst = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and worst == 0:
                worst = rc
                for q in live:
                    q.terminate()
                deadline = time.time() + 10.0
                for q in live:
                    try:
                        q.wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
                        q.wait()
                live = []
                break
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c5", choices=sorted(WORKLOADS))
    ap.add_argument("--single-process", action="store_true", help="N > 1: ONE process drives all devices (lssvm_mi355_problem_create_multi) instead of one process per GPU")
    ap.add_argument("--devices", default=None, help="--single-process: comma separated HIP ordinals (repeats allowed), default 0..N-1")
    ap.add_argument("--exchange", type=int, default=None, choices=[0, 1, 2],
                    help="how the partial K*v vectors meet.  --single-process: 0 automatic, 1 RCCL, 2 peer kernels over xGMI.  One process per GPU: 1 (default) the "
                         "library's RCCL communicator, 2 HIP IPC + the peer kernel (no RCCL inside the library)")
    ap.add_argument("--dist-backend", default="gloo", choices=["gloo", "nccl"],
                    help="torch.distributed backend of the ranks' SIDE channel (barrier, id exchange, max over ranks).  gloo (default): the data path -- one "
                         "all-reduce per implicit matvec -- is the library's own RCCL communicator either way, and a side channel that does not depend on RCCL is "
                         "what lets the ranks agree on a fall-back when the RCCL bootstrap fails or hangs")
    ap.add_argument("--rccl-timeout-s", type=float, default=120.0,
                    help="one process per GPU: how long the library's ncclCommInitRank may take before the ranks give up on RCCL (bootstrap hangs of minutes were "
                         "seen on single-GPU boxes of this pool)")
    ap.add_argument("--balance-shares", action="store_true",
                    help="one process per GPU, symmetric variant: after the warm-up the ranks compare the tile-kernel time of their (equal) shares, set shard weights "
                         "proportional to their pace on the live problems (lssvm_mi355_problem_rebalance; the data is replicated, nothing moves) and warm up again -- "
                         "the devices of one node do not run at one pace (the boxes of this pool: 252 ... 277 ms for the same kernel)")
    ap.add_argument("--rank-devices", default=None,
                    help="one process per GPU: comma separated HIP ordinal per local rank (default: the local rank).  Repeats put several ranks on one device "
                         "-- a functional check of the rank path on a one-GPU box (needs --exchange 2, or --rccl-stand-in: the real RCCL refuses two ranks on one device)")
    ap.add_argument("--rccl-stand-in", default=None, metavar="PATH",
                    help="TESTING AID for one-GPU boxes: load this library (tests/tools/fake_rccl/librccl.so.1, a stand-in with RCCL's SONAME that accepts several "
                         "ranks per device) into the process before torch and the product library, so that the product's own dlopen(\"librccl.so.1\") resolves to it.  "
                         "The line then names it in config.rccl_library; such a line is a functional check of the RCCL exchange path, never a scaling number")
    ap.add_argument("--no-other-workloads", action="store_true", help="default run (c5, one GPU): skip the short legs of c2 / c3 / c4 reported as other_workloads")
    ap.add_argument("--gram-mode", type=int, default=None, choices=[0, 1, 2, 3],
                    help="fp32 only: 3 (library default) = f16x3 where the data passes the representability check, else bf16x6; 2 = f16x3 unchecked; "
                         "1 = bf16x6; 0 = native v_mfma_f32 chains")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE", help="set a library tuning knob (lssvm_mi355_set_option) before the problem is created; repeatable")
    ap.add_argument("--cpu-sample-rows", type=int, default=20480)  # 4 implicit matvecs of ~3 s each on 128 host cores: about 12 s of CPU work per line
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-native-reference", action="store_true", help="skip the extra native v_mfma_f32 run reported beside a bf16x6 headline")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the bare bf16 MFMA loop (about 4 s) whose rate on this device is reported beside roofline.frac")
    ap.add_argument("--seed", type=int, default=42)
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ
    if args.devices is not None:
        args.single_process = True
    if args.gpus > 1 and not launched and not args.single_process:
        sys.exit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0")) if not args.single_process else 0
    world = int(os.environ.get("WORLD_SIZE", "1")) if not args.single_process else 1
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if not args.single_process else 0
    if not args.single_process and world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    if world > 1:
        # all ranks live on ONE node (the contract of this script): RCCL's bootstrap need not scan the network interfaces (it took
        # 100-600 s on some boxes of the pool), and the host driver only supports dmabuf IPC
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # (the side channel: a container's hostname need not resolve)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    stand_in = None
    if args.rccl_stand_in:
        import importlib.util

        spec = importlib.util.spec_from_file_location("fake_rccl_preload", os.path.join(os.path.dirname(os.path.abspath(args.rccl_stand_in)), "preload.py"))
        preload = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(preload)
        stand_in = preload.load(args.rccl_stand_in)  # ahead of torch (which keeps its own RCCL): the product's dlopen("librccl.so.1") finds this one by SONAME

    import numpy as np
    import torch  # plumbing only: device selection, synchronisation, torch.distributed (RCCL) for the barrier / id exchange

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible (the HIP path has no CPU fallback)")
    rank_devices = [int(t) for t in args.rank_devices.split(",")] if args.rank_devices else None
    if rank_devices is not None and not args.single_process:
        if len(rank_devices) != world:
            raise SystemExit(f"--rank-devices lists {len(rank_devices)} devices for {world} ranks")
        local_rank = rank_devices[local_rank]  # from here on: this rank's HIP ordinal
    torch.cuda.set_device(local_rank)

    from plssvm_amd import _capi, backend
    from plssvm_amd.datagen import make_blobs_pm1
    from plssvm_amd.parameter import Parameter

    devices = None
    if args.single_process:
        devices = [int(t) for t in args.devices.split(",")] if args.devices else list(range(args.gpus))
    shards = len(devices) if devices is not None else world
    n_gpus = len(set(devices)) if devices is not None else (len(set(rank_devices)) if rank_devices is not None else world)

    dist = None
    exchange_note = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        side_device = "cuda" if args.dist_backend == "nccl" else "cpu"
        if args.exchange != 2:
            # Hand rank 0's RCCL unique id to every rank, then build the library's own communicator (one per process) -- in a worker thread, so
            # that a bootstrap that hangs cannot take the launcher's whole timeout: after --rccl-timeout-s the ranks compare notes over the
            # side channel.  Every rank built it: RCCL exchange.  Some rank FAILED (an error came back): every rank drops its communicator and
            # all use the HIP IPC exchange (same partition, same sum order); the JSON line says so.  Some rank is still STUCK inside RCCL: that
            # process cannot be trusted to continue (the call cannot be cancelled), so every rank exits non-zero with a clear message.
            import threading

            from plssvm_amd.sharding import init_library_communicator

            result = {}

            def build():
                try:
                    init_library_communicator(dist, local_rank, device="cuda" if args.dist_backend == "nccl" else None)
                    result["ok"] = True
                except Exception as e:  # noqa: BLE001  (PlssvmError from the library, or a torch.distributed error)
                    result["error"] = f"{type(e).__name__}: {e}"

            worker = threading.Thread(target=build, daemon=True)
            t_boot = time.perf_counter()
            worker.start()
            worker.join(args.rccl_timeout_s)
            stuck = worker.is_alive()
            ok = 1 if (not stuck and result.get("ok")) else 0
            why = "still inside ncclCommInitRank after %.0f s" % args.rccl_timeout_s if stuck else result.get("error", "")
            if dist.get_backend() == "nccl" and stuck:
                raise SystemExit(f"[bench rank {rank}] the library's RCCL communicator: {why} (and the side channel is RCCL too: nothing to agree over)")
            flags = torch.tensor([ok, 0 if stuck else 1], dtype=torch.int32, device=side_device)
            dist.all_reduce(flags, op=dist.ReduceOp.MIN)
            all_ok, none_stuck = int(flags[0].item()), int(flags[1].item())
            if not none_stuck:
                print(f"[bench rank {rank}] giving up: a rank is stuck inside the RCCL bootstrap" + (f" (this one: {why})" if stuck else ""), file=sys.stderr, flush=True)
                os._exit(3)  # (not sys.exit: a worker thread may be blocked inside RCCL)
            if not all_ok:
                if ok:
                    backend.comm_destroy()
                args.exchange = 2
                exchange_note = "the library's RCCL communicator could not be built on every rank" + (f" ({why})" if why else "") + ": HIP IPC exchange instead"
                print(f"[bench rank {rank}] {exchange_note}", file=sys.stderr, flush=True)
            elif rank == 0:
                print(f"[bench] RCCL communicator of {world} ranks built in {time.perf_counter() - t_boot:.1f} s", file=sys.stderr, flush=True)

    wl = WORKLOADS[args.workload]
    N, d = wl["n"], wl["d"]
    dt = np.dtype(wl["dtype"])
    X, y = make_blobs_pm1(N, d, seed=args.seed, dtype=dt)  # identical on every rank (seeded)
    params = Parameter(kernel_type=wl["kernel"], degree=3, gamma=None, coef0=0.0, cost=1.0)

    def device_sync():
        if devices is not None:
            for dev in sorted(set(devices)):
                torch.cuda.synchronize(dev)
        else:
            torch.cuda.synchronize()

    def barrier():
        device_sync()
        if dist is not None:
            dist.barrier()
        device_sync()

    if args.gram_mode is not None:
        _capi.set_option("gram_mode", args.gram_mode)
    if args.exchange is not None:
        _capi.set_option("exchange", args.exchange)
    for kv in args.option:
        name, value = kv.split("=", 1)
        _capi.set_option(name.strip(), int(value))
    if devices is not None:
        prob = backend.ResidentProblem(params, X, devices=devices)
    else:
        prob = backend.ResidentProblem(params, X, device=local_rank, rank=rank, world=world)
        if world > 1 and args.exchange == 2:
            from plssvm_amd.sharding import connect_peers

            connect_peers(dist, prob)  # HIP IPC: every rank maps every rank's partial vector
    prob.cg_begin(y, 1e-30)  # eps^2 underflows: the loop only stops early on delta == 0 (fixed iteration count)
    if args.warmup > 0:
        prob.cg_step(args.warmup)
    prob.synchronize()
    i0 = prob.info()
    shard_weights = shares_before = None
    if args.balance_shares and dist is not None and world > 1 and devices is None and int(i0.get("symmetric", 0)) == 1:
        # shares by measured pace: every rank's tile-kernel time per matvec over the warm-up (its share of the triangle was 1 / world), gathered over the side channel
        from plssvm_amd.sharding import triangle_share

        mine = (float(i0["matvec_kernel_ms"]), triangle_share(N - 1, world, rank))
        both = [None] * world
        dist.all_gather_object(both, mine)
        shares_before = [round(ms, 4) for ms, _ in both]
        if all(ms > 0 for ms, _ in both) and max(shares_before) > 1.01 * min(shares_before):
            pace = [area / ms for ms, area in both]
            shard_weights = [v * world / sum(pace) for v in pace]
            prob.rebalance(shard_weights)  # (the same list on every rank: all_gather_object hands every rank the same values; the live problem is resharded, nothing is uploaded again)
            if args.warmup > 0:
                prob.cg_step(args.warmup)
            prob.synchronize()
            i0 = prob.info()
    # board power and shader clock of this rank's device while the timed steps run (a host thread reading two sysfs files every 20 ms: the
    # 16-bit tile kernels sit at the board's power cap, DESIGN.md 4.1.0 -- the line should say so for the box it was taken on)
    sampler = None
    if rank == 0 and devices is None:
        from plssvm_amd.hwmon import PowerSampler

        sampler = PowerSampler(local_rank)
        if sampler.available:
            sampler.start()
    # throttle residency of the timed region ("power-bound" as a counter, not an inference): the firmware's accumulators through amd-smi,
    # read just outside the region, and the violation flags once from a side thread in its middle (host processes only; nothing touches the device's queues)
    smi0 = smi_mid = None
    mid_thread = None
    if sampler is not None:
        from plssvm_amd.hwmon import smi_snapshot, throttle_between

        smi0 = smi_snapshot()
        if smi0 is not None:
            import threading

            mid_box = {}

            def mid():
                time.sleep(max(0.5, 0.25 * args.steps * max(float(i0["avg_iteration_ms"]), 1.0) * 1e-3))
                mid_box["snap"] = smi_snapshot()

            mid_thread = threading.Thread(target=mid, daemon=True)
    barrier()
    t0 = time.perf_counter()
    w0 = time.time()
    if mid_thread is not None:
        mid_thread.start()
    prob.cg_step(args.steps)
    prob.synchronize()
    barrier()
    t1 = time.perf_counter()
    w1 = time.time()
    i1 = prob.info()
    throttle = None
    if smi0 is not None:
        mid_thread.join(40.0)
        smi1 = smi_snapshot()
        throttle = throttle_between(smi0, smi1, w0, w1)
        smi_mid = mid_box.get("snap")
        if throttle is not None and smi_mid is not None and w0 <= smi_mid[0] <= w1:
            throttle["flags_in_mid_region"] = smi_mid[1]["status"]
            throttle["socket_power_w_in_mid_region"] = smi_mid[1]["socket_power_w"]
            throttle["gfx_clocks_mhz_in_mid_region"] = smi_mid[1]["gfx_clocks_mhz"]
        if throttle is not None:
            throttle["source"] = "amd-smi metric --json: throttle.*_accumulated over accumulation_counter (firmware residency counters), read before and after the timed region; flags from one reading inside it"
    board_power = None
    if sampler is not None and sampler.available:
        from plssvm_amd.hwmon import median

        sampler.stop()
        watts, ghz = sampler.window(w0, w1)
        if watts:
            # energy of the WHOLE timed region (the settled samples alone leave the ramp at its start out): trapezoid rule over every sample in [w0, w1]
            allw, _ = sampler.window(w0, w1, settle=0.0)
            energy = sum(allw) / len(allw) * (w1 - w0) if allw else None
            board_power = {"median_w": median(watts), "max_w": max(watts), "cap_w": sampler.cap_watts(), "shader_clock_ghz_median": median(ghz), "samples": len(watts),
                           "source": f"amdgpu hwmon of PCI {sampler.bus}, every 20 ms: median / max over the last 70 % of the timed region, energy over all of it",
                           "energy_j_per_step": (energy / max(args.steps, 1)) if energy is not None else None, "energy_samples": len(allw),
                           "throttle": throttle if throttle is not None else {"unavailable": "amd-smi metric --json gave no reading (tool missing, profiled run, or no amdsmi module for this interpreter)"}}

    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")  # (the side channel: gloo by default)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    steps_done = int(i1["iterations"] - i0["iterations"])
    # every rank runs the identical O(n) kernels on the identical exchanged vector: the CG scalars must be the SAME BITS on all ranks (the library checks that
    # across the shards of one process at cg_finish; across processes only the application can): delta after the timed steps, compared over the side channel
    ranks_agree = None
    if dist is not None:
        import struct

        mine = struct.pack("<dQ", float(i1["residuum"]), int(i1["iterations"]))
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        ranks_agree = all(e == everyone[0] for e in everyone)
        if not ranks_agree:
            print(f"[bench rank {rank}] the ranks DISAGREE on the residuum after the timed steps: {[struct.unpack('<dQ', e) for e in everyone]}", file=sys.stderr, flush=True)

    n = N - 1
    flop_step = 2.0 * n * n * d
    value = flop_step * steps_done / elapsed / 1e9
    # every rank's tile-kernel time per matvec (rank order): on a real node the spread of the devices' pace, beside rank 0's figure that prices the roofline
    kernel_ms_per_rank = None
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, float(kernel_ms_between(i0, i1)[2]))
        kernel_ms_per_rank = gathered

    # roofline of the dominant kernel (the tile kernel of the implicit matvec), from HIP events on the solver stream(s); a process
    # that drives several shards reports the slowest shard's average launch
    launches, timed, kern_ms = kernel_ms_between(i0, i1)
    from plssvm_amd.sharding import triangle_share, work_share

    # Three flop counts of one tile-kernel launch of one shard:
    #   square   = its share of the full n x n square, 2 n^2 d in total: the convention of `value` ("no symmetry credit")
    #   executed = what the matrix cores really do (symmetric variant: tiles on/below the diagonal, whole tiles incl. padding)
    #   useful   = the algorithm's own count: the full square for the full-square variant; entries j <= i for the symmetric one
    #              (the reference's count, svm_kernel.cpp:36-39).  `roofline.achieved` = useful / kernel time, so frac <= 1.
    # The shares are dealt by equal area, so every shard's counts agree within a row block; priced here for shard `rank` (0 for one process).
    symmetric = bool(i1.get("symmetric", 0))
    sq_mac, exe_mac = work_share(n, shards, rank, symmetric, shard_weights)
    use_mac = triangle_share(n, shards, rank, shard_weights) if symmetric else sq_mac
    kern_s = kern_ms * 1e-3
    square_launch, exec_launch, useful_launch = 2.0 * sq_mac * d, 2.0 * exe_mac * d, 2.0 * use_mac * d
    achieved = useful_launch / kern_s / 1e12 if kern_ms > 0 else 0.0
    executed = exec_launch / kern_s / 1e12 if kern_ms > 0 else 0.0
    effective = square_launch / kern_s / 1e12 if kern_ms > 0 else 0.0
    peak = PEAK_TFLOPS[wl["dtype"]]
    # fp32 split modes: every fp32 multiply-add runs as THREE f16 plane products ("f16x3": two f16 planes per operand, the default where
    # the data allows) or SIX bf16 plane products ("bf16x6": exact 3-way split) on the 16-bit MFMA with fp32 accumulation, at
    # fp32-equivalent accuracy.  The roofline of those kernels is the dense 16-bit MFMA peak (16 x the f32 MFMA rate,
    # MI355X_MICROARCH.md "Matrix cores": f16 and bf16 take the same cycles), and the kernel's own flop count is 3 x / 6 x the fp32 count.
    gram_mode = int(i1.get("gram_mode", 0))
    gram_name = {0: "native", 1: "bf16x6", 2: "f16x3", 3: "f16 grid planes"}[gram_mode]
    bf16x6 = gram_mode != 0  # (a split mode on the 16-bit matrix cores)
    plane_products = plane_products_of(gram_mode)
    fp32_equivalent = achieved
    if bf16x6:
        peak = PEAK_TFLOPS["bf16"]
        achieved *= plane_products
        executed *= plane_products
    traffic, traffic_source = measured_traffic(f"{args.workload}_n{shards}")
    bands = int(i1.get("tile_launches_per_matvec", 1))
    if traffic is not None:
        traffic *= bands  # the PMC passes count per tile-kernel launch (= per row-block band); `achieved` is priced per matvec

    if rank == 0:
        exchange_names = {0: "none", 1: "RCCL all-reduce" if symmetric else "RCCL all-gather", 2: "peer kernels over xGMI (fixed-order sum)"}
        if shards == 1:
            parallelism = "single GPU"
        elif devices is not None:
            parallelism = f"row-block sharding x{shards}, one process driving devices {devices}"
        else:
            parallelism = f"row-block sharding x{world}, one process per GPU" if rank_devices is None else f"row-block sharding x{world}, one process per rank on devices {rank_devices}"
            parallelism += ", partial vectors over HIP IPC" if args.exchange == 2 else ", partial vectors over RCCL"
            if exchange_note:
                parallelism += f" [{exchange_note}]"
        out = {
            "metric": "effective K*d GFLOP/s of the CG iteration (2*n^2*d per iteration / time), RBF fp32 N x d" if wl["kernel"] == "rbf" and wl["dtype"] == "float32"
            else f"effective K*d GFLOP/s of the CG iteration, {wl['kernel']} {wl['dtype']}",
            "value": value, "unit": "GFLOP/s", "cg_iters_per_s": steps_done / elapsed,
            "n_gpus": n_gpus, "steps": steps_done, "warmup": args.warmup, "ms_per_step": elapsed / max(steps_done, 1) * 1e3,
            "setup_ms": float(i1.get("setup_ms", 0.0)),  # upload of the data matrix + q, norms, operand planes, work-item lists (outside the timed region; lssvm_cg_info.setup_ms)
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if wl["dtype"] == "float32" else "f64", "data": "synthetic",
            "arithmetic": {1: "fp32 operands split exactly into 3 bf16 planes, 6 plane products per multiply-add accumulated in fp32 on the bf16 matrix cores",
                           2: "fp32 operands as 2 f16 planes (hi + mid, representation checked on the data at set-up), 3 plane products per multiply-add "
                              "accumulated in fp32 on the f16 matrix cores",
                           3: "rbf with a large exponent scale: fp32 operands as 3 f16 planes (a grid plane + 2 rest planes), 6 plane products per multiply-add accumulated in fp32 "
                              "on the f16 matrix cores, the grid-plane products first so that the large terms cancel exactly"}.get(gram_mode, "native " + wl["dtype"] + " matrix-core fma chains"),
            "config": {"workload": wl["desc"], "num_points": N, "num_features": d, "kernel": wl["kernel"], "gamma": 1.0 / d, "cost": 1.0,
                       "seed": args.seed, "library_options": args.option, "parallelism": parallelism, "shards": shards, "ranks": world, "exchange": exchange_names.get(int(i1.get("exchange", 0)), "?"),
                       "residuum_after_timed_steps": i1["residuum"], "residuum_bit_equal_on_all_ranks": ranks_agree,
                       # what RCCL itself reports for the communicator the partial vectors travelled over (ncclCommCount / ncclCommCuDevice on rank 0), and the
                       # file its entry points were resolved from -- null / 0 when no RCCL exchange ran
                       "persistent_launches": int(i1.get("persistent_launches", 0)), "shard_weights": ([round(w, 5) for w in shard_weights] if shard_weights else None), "share_kernel_ms_at_equal_shares": shares_before, "rccl_nranks": int(i1.get("rccl_nranks", 0)), "rccl_rank0_device": (int(i1["rccl_device"]) if int(i1.get("rccl_nranks", 0)) > 0 else None),
                       "rccl_library": (backend.comm_library_path() if int(i1.get("rccl_nranks", 0)) > 0 else None),
                       "rccl_is_stand_in": stand_in is not None},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "traffic_source": traffic_source,
                         # the HBM side of the same launch (SURVEY.md 8d asks for both fractions): algorithmic bytes n*d*s + 4*n*s, the measured
                         # fabric traffic over the kernel time against the 8 TB/s peak -- far from binding, the launch is matrix-core bound
                         "hbm": {"algorithmic_bytes_per_launch": float(n) * d * dt.itemsize + 4.0 * n * dt.itemsize,
                                 "traffic_rate_TBps": (traffic / kern_s / 1e12) if (traffic is not None and kern_ms > 0) else None,
                                 "traffic_frac_of_hbm_peak": (traffic / kern_s / 1e12 / 8.0) if (traffic is not None and kern_ms > 0) else None},
                         "board_power": board_power,
                         "kernel": "lssvm::tile_matvec (implicit K*d tile kernel)", "launches": launches, "launches_timed": timed, "avg_launch_ms": kern_ms,
                         "kernel_ms_per_rank": kernel_ms_per_rank, "tile_launches_per_matvec": bands,
                         "launch_note": "a 'launch' here is ONE implicit matvec = the sum of its row-block band launches of the tile kernel (rocprofv3 lists the bands one by one); "
                                        "where a matvec is short (< 1 ms by the library's shape rule) the HIP events bracket every 8th matvec only (never the first after cg_begin) -- avg_launch_ms is the summed time of the launches_timed bracketed matvecs over their count",
                         "algorithmic_flop_per_launch": useful_launch * plane_products, "symmetric": symmetric,
                         "executed_flop_per_launch": exec_launch * plane_products, "executed": executed, "executed_frac": executed / peak,
                         "gram_mode": gram_name, "plane_products": plane_products, "fp32_equivalent": fp32_equivalent, "fp32_mfma_peak": PEAK_TFLOPS["float32"],
                         # the same launch priced with the full-square convention of `value` (can exceed the peak when symmetric)
                         "full_square_flop_per_launch": square_launch, "effective_full_square": effective},
        }
        if bf16x6 and shards == 1 and not args.no_native_reference:
            # the same workload on native v_mfma_f32 chains (option gram_mode = 0), reported beside the headline -- never as `value`
            prob.close()
            _capi.set_option("gram_mode", 0)
            nat = backend.ResidentProblem(params, X, device=local_rank, rank=rank, world=world)
            nat.cg_begin(y, 1e-30)
            nat.cg_step(1)
            nat.synchronize()
            j0 = nat.info()
            tn = time.perf_counter()
            nat.cg_step(min(args.steps, 5))
            nat.synchronize()
            tn = time.perf_counter() - tn
            j1 = nat.info()
            nsteps = int(j1["iterations"] - j0["iterations"])
            _, _, nk_ms = kernel_ms_between(j0, j1)
            out["native_f32_path"] = {"ms_per_step": tn / max(nsteps, 1) * 1e3, "value": flop_step * nsteps / tn / 1e9, "unit": "GFLOP/s", "steps": nsteps,
                                      "tile_kernel_ms": nk_ms, "useful_tflops": useful_launch / (nk_ms * 1e-3) / 1e12 if nk_ms > 0 else 0.0,
                                      "frac_of_f32_mfma_peak": useful_launch / (nk_ms * 1e-3) / 1e12 / PEAK_TFLOPS["float32"] if nk_ms > 0 else 0.0}
            nat.close()
            _capi.set_option("gram_mode", 3)
            prob = None
        if bf16x6 and shards == 1 and not args.no_ceiling:
            # what a loop of NOTHING BUT the Gram kernel's MFMA instruction sustains on this device, now (the chip lowers its clock under
            # matrix-core load): reported beside roofline.frac, never instead of it -- `peak` stays the nominal figure
            if prob is not None:
                prob.close()
                prob = None
            import ctypes as C

            f16_bit = 2 if gram_mode in (2, 3) else 0  # the bare loop runs the instruction of the kernel it is compared with

            def ceiling(b_from_lds):
                tf, ghz, nominal = C.c_double(0), C.c_double(0), C.c_double(0)
                _capi.check(_capi.lib.lssvm_mi355_measure_bf16_mfma_ceiling(C.c_int(local_rank), C.c_int(b_from_lds | f16_bit), C.c_double(1500.0), C.byref(tf), C.byref(ghz), C.byref(nominal)))
                return tf.value, ghz.value, nominal.value

            regs_tf, regs_ghz, nominal = ceiling(0)
            lds_tf, lds_ghz, _ = ceiling(1)
            out["roofline"]["bare_mfma_loop"] = {
                "what": ("v_mfma_f32_16x16x32_f16" if gram_mode == 2 else "v_mfma_f32_16x16x32_bf16") + " only, 64x64 wave tiles, two waves per SIMD, normal(0,1) operands, 1.5 s of back-to-back launches before timing "
                        "(lssvm_mi355_measure_bf16_mfma_ceiling); same device, same process, after the timed region",
                "operands_in_registers": {"tflops": regs_tf, "frac_of_peak": regs_tf / peak, "clock_ghz": regs_ghz},
                "b_fragments_from_lds": {"tflops": lds_tf, "frac_of_peak": lds_tf / peak, "clock_ghz": lds_ghz},
                "nominal_peak_tflops": nominal,
                "kernel_vs_bare_register_loop": achieved / regs_tf if regs_tf > 0 else None,
                "kernel_vs_bare_lds_fed_loop": achieved / lds_tf if lds_tf > 0 else None,
            }
        if args.workload == "c5" and shards == 1 and not args.no_other_workloads:
            # BASELINE's other single-GPU configs, driver-observed: short legs on the same device after the headline's legs
            if prob is not None:
                prob.close()
                prob = None
            legs = {}
            for name, k, w in (("c2", 200, 20), ("c3", 20, 3), ("c4", 20, 3)):
                try:
                    legs[name] = short_leg(name, k, w, args.seed, local_rank)
                except Exception as e:  # noqa: BLE001  (the headline line must still be printed)
                    legs[name] = {"error": f"{type(e).__name__}: {e}"}
            try:
                legs["predict"] = predict_leg(args.seed, local_rank)
            except Exception as e:  # noqa: BLE001
                legs["predict"] = {"error": f"{type(e).__name__}: {e}"}
            out["other_workloads"] = legs
            try:
                out["e2e"] = e2e_leg(args.seed)
            except Exception as e:  # noqa: BLE001
                out["e2e"] = {"error": f"{type(e).__name__}: {e}"}
        # the CPU legs come LAST: the OpenMP runtime's workers keep spinning after a parallel region and would slow the host side of
        # the GPU legs down (measured: 16 ms instead of 2.6 ms per c2 iteration in a native leg that followed them)
        if not args.no_cpu_baseline and shards == 1:
            line_a, line_b = cpu_baseline(X, y, wl["kernel"], args.cpu_sample_rows, 3)
            out["cpu_baseline"] = line_a
            if line_b is not None:
                out["cpu_baseline_release"] = line_b
        emit_result_line(json.dumps(out))

    if dist is not None:
        dist.barrier()  # (HIP IPC: a rank's partial vector stays mapped by its peers until every rank is done)
    if prob is not None:
        prob.close()
    if dist is not None:
        if args.exchange != 2:
            backend.comm_destroy()
        dist.destroy_process_group()


_RESULT_FD = None


def keep_stdout_for_the_result_line():
    """The contract is ONE JSON line on stdout.  Libraries underneath write there too (gloo: "[Gloo] Rank 0 is connected to 3 peer ranks ..." in front of
    the line of a multi-rank run), so file descriptor 1 is pointed at stderr for the whole run and the original is kept for the result line alone."""
    global _RESULT_FD
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)


def emit_result_line(line: str):
    sys.stdout.flush()
    if _RESULT_FD is None:
        print(line, flush=True)
    else:
        os.write(_RESULT_FD, (line + "\n").encode())


if __name__ == "__main__":
    keep_stdout_for_the_result_line()
    main()
