"""GPU (-m gpu): the library next to PyTorch in one process (bench.py imports torch first for torch.distributed), a device
pointer handed over the C ABI, and bench.py's JSON contract on a small workload."""

import json
import os

os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # single node: RCCL's bootstrap need not scan the interfaces (it took 100-600 s on some boxes)
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_torch_first_then_library_and_device_pointer_input():
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')
import torch
assert torch.cuda.is_available()
from plssvm_amd import backend
from plssvm_amd.parameter import Parameter
from plssvm_amd.datagen import make_blobs_pm1
X, y = make_blobs_pm1(600, 48, seed=1, dtype=np.float32)
p = Parameter(kernel_type='rbf')
a, rho, _ = backend.solve_system_of_linear_equations(p, X, y, 1e-30, 5)
Xd = torch.from_numpy(X).cuda()
torch.cuda.synchronize()
prob = backend.ResidentProblem(p, None, device_ptr=Xd.data_ptr(), shape=X.shape, dtype=np.float32)
prob.cg_begin(y, 1e-30); prob.cg_step(5); a2, rho2, info = prob.cg_finish(); prob.close()
assert np.array_equal(a, a2) and rho == rho2, 'device-pointer input must give the same bits as host input'
print('OK')
""" % (ROOT, ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_resident_predictor_with_the_batch_and_the_values_in_hbm():
    """lssvm_mi355_predictor_predict with LSSVM_MEM_DEVICE: the points and the values are torch tensors on device 0 -- the same bits as the call from host buffers, for a
    batch on the resident path (rbf, linear), for fp64 (the one-shot path inside the predictor: a round trip through the host) and for a batch the resident form declines."""
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')
import torch
from plssvm_amd import backend
from plssvm_amd.parameter import Parameter
from plssvm_amd.datagen import make_blobs_pm1
rng = np.random.default_rng(5)
for kernel, dt in (('rbf', np.float32), ('linear', np.float32), ('polynomial', np.float32), ('rbf', np.float64)):
    X, _ = make_blobs_pm1(2500 + 9000, 96, seed=3, dtype=dt)
    sv, pts = X[:2500], X[2500:]
    alpha = rng.standard_normal(2500).astype(dt)
    with backend.Predictor(Parameter(kernel_type=kernel, degree=2, gamma=1.0 / 96, coef0=1.0), sv, alpha, 0.25) as pred:
        for batch in (pts, pts[:77], (pts * 12.0).astype(dt)):
            info_h, info_d = {}, {}
            want = pred.predict(batch, info_out=info_h)
            Pd = torch.from_numpy(np.ascontiguousarray(batch)).cuda()
            Od = torch.full((batch.shape[0],), float('nan'), dtype=Pd.dtype, device='cuda')
            torch.cuda.synchronize()
            pred.predict_device(Pd.data_ptr(), batch.shape[0], Od.data_ptr(), info_out=info_d)
            got = Od.cpu().numpy()
            assert np.array_equal(got, want), (kernel, dt, batch.shape, float(np.max(np.abs(got - want))))
            assert info_d['resident'] == info_h['resident'], (kernel, dt, info_d, info_h)
            assert np.array_equal(Pd.cpu().numpy(), batch)  # the caller's tensor is read only
print('OK')
""" % (ROOT, ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_bench_json_contract_small_workload():
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c2", "--steps", "3", "--warmup", "1", "--cpu-sample-rows", "2048"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["dtype"] == "f32" and j["data"] == "synthetic" and j["vs_baseline"] is None
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # `achieved` counts the algorithm's own flops (entries j <= i in the default symmetric variant), `executed` whole tiles,
    # `effective_full_square` prices the launch like `value` does (2 n^2 d)
    assert 0.05 < r["frac"] <= r["executed_frac"] < 1.0 and r["symmetric"] in (True, False)
    # fp32 default: the Gram tiles run as six bf16 plane products per multiply-add; the roofline is the bf16 MFMA peak
    planes = {"f16x3": 3.0, "bf16x6": 6.0, "native": 1.0}[r["gram_mode"]]
    if r["gram_mode"] != "native":  # the native v_mfma_f32 path of the same workload is reported beside the headline
        nat = j["native_f32_path"]
        assert nat["ms_per_step"] > j["ms_per_step"] and 0.05 < nat["frac_of_f32_mfma_peak"] < 1.0
        # ... and so is what a bare loop of the kernel's MFMA instruction sustains on this device (the practical ceiling under the nominal peak)
        bare = r["bare_mfma_loop"]
        assert 0.4 < bare["b_fragments_from_lds"]["frac_of_peak"] < 1.0 and 0.4 < bare["operands_in_registers"]["frac_of_peak"] < 1.0
        assert 1.0 < bare["operands_in_registers"]["clock_ghz"] < 2.6 and abs(bare["nominal_peak_tflops"] - r["peak"]) < 0.01 * r["peak"]
        assert 0.3 < bare["kernel_vs_bare_register_loop"] < 1.0
    assert r["gram_mode"] in ("f16x3", "bf16x6", "native") and abs(r["achieved"] - planes * r["fp32_equivalent"]) < 1e-9 * r["achieved"]
    assert r["effective_full_square"] >= r["fp32_equivalent"] and planes * r["full_square_flop_per_launch"] >= r["executed_flop_per_launch"] * 0.99
    c = j["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    # value is whole-job: 2 n^2 d per step / ms_per_step
    n, d = 49_999, 128
    assert abs(j["value"] - 2.0 * n * n * d / (j["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * j["value"]


def test_bench_rank_path_two_ranks_on_one_device():
    """bench.py's one-process-per-GPU path as the driver runs it -- the parent starts its ranks as child processes, torch.distributed side
    channel, row-block partition, exchange of the partial vectors, max-over-ranks timing, ONE JSON line from rank 0 -- on a one-GPU box: two
    ranks share device 0 and exchange over HIP IPC (RCCL refuses two ranks on one device; with distinct devices the same path runs the
    library's RCCL communicator)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--exchange", "2", "--dist-backend", "gloo", "--rank-devices", "0,0",
                          "--workload", "c2", "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=900, env=dict(os.environ))
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["scaling"] == "strong" and j["config"]["shards"] == 2
    assert "HIP IPC" in j["config"]["parallelism"] and j["config"]["exchange"].startswith("peer kernels")
    assert j["roofline"]["symmetric"] is True and 0.05 < j["roofline"]["frac"] < 1.0 and j["ms_per_step"] > 0
    # value is whole-job: 2 n^2 d per step / ms_per_step
    n, d = 49_999, 128
    assert abs(j["value"] - 2.0 * n * n * d / (j["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * j["value"]


def test_rccl_all_gather_path_with_a_world_of_one():
    """One GPU per box here, so the multi-rank exchange is exercised with world = 1: the library dlopens RCCL, builds its own
    communicator from a unique id and runs ncclAllGather (in place, on the solver stream) after every implicit matvec."""
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from plssvm_amd import _capi, backend
from plssvm_amd.parameter import Parameter
from plssvm_amd.datagen import make_blobs_pm1
X, y = make_blobs_pm1(700, 30, seed=2, dtype=np.float32)
p = Parameter(kernel_type='rbf')
a, rho, _ = backend.solve_system_of_linear_equations(p, X, y, 1e-30, 8)
uid = backend.comm_get_unique_id()
assert len(uid) == 128 and any(uid)
backend.comm_init(0, 0, 1, uid)
_capi.set_option('force_collective', 1)
prob = backend.ResidentProblem(p, X, device=0, rank=0, world=1)
prob.cg_begin(y, 1e-30); prob.cg_step(8); a2, rho2, info = prob.cg_finish(); prob.close()
backend.comm_destroy()
assert np.array_equal(a, a2) and rho == rho2
print('OK')
""" % ROOT
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180)
    except subprocess.TimeoutExpired:
        # seen on some boxes of the pool: RCCL's bootstrap (ncclGetUniqueId / ncclCommInitRank) does not return for 10+ minutes even for a world
        # of one with NCCL_SOCKET_IFNAME=lo -- an environment property, nothing this library controls; the exchange code itself is covered by the
        # multi-shard tests of tests/test_gpu_multi_device.py
        pytest.skip("RCCL bootstrap did not complete within 180 s on this box")
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_cpp_adaptor_reference_style_tests():
    """tests/cpp/test_csvm: the reference's GenericCSVM known-answer tests + fit/predict/score through the C++ adaptor."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_csvm")
    assert os.path.isfile(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
