"""GPU (-m gpu): the HIP path, called through the C ABI, against (1) the golden vectors captured from the reference's own
OpenMP kernels, (2) the CPU oracle on seeded inputs, (3) size-independent properties at BASELINE.json's full sizes, and
the edge cases the reference tests (ragged sizes, tiny inputs, preconditions).

Tolerances (stated per test):
  kernel level (q, one implicit matvec): the reference's own GPU-vs-CPU criterion |a-b| < 128 eps (|a|+|b|) per element
      (tests/custom_test_macros.hpp:114-137), asserted here in the stricter rel-inf form < 32 eps;
  CG level fp64: alpha rel-inf < 1e-6, iterations within +-1 of the reference (the stop test is a float comparison);
  CG level fp32: the reference's fp32 CG does not reproduce ITSELF to 1e-4 across thread counts (atomics reorder the sums,
      DESIGN.md section 5); asserted instead: our distance to the fp64 solution <= max(2 x the reference's, 1e-4).
"""

import os

import numpy as np
import pytest

import oracle_lib as ol
from conftest import DATASETS, DTYPES, KERNELS, PARAM_SETS, resolved_kw
from plssvm_amd import _capi, backend
from plssvm_amd.datagen import make_blobs_pm1
from plssvm_amd.exceptions import InvalidParameterError
from plssvm_amd.parameter import Parameter

pytestmark = pytest.mark.gpu


def prm(kernel, P):
    return Parameter(kernel_type=kernel, degree=P["degree"], gamma=P["gamma"], coef0=P["coef0"], cost=P["cost"])


def test_device_is_gfx950():
    assert _capi.device_count() >= 1
    assert "gfx950" in _capi.device_name(0)


@pytest.mark.parametrize("name", DATASETS)
@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("pname", ["ref", "def"])
@pytest.mark.parametrize("kernel", KERNELS)
def test_generate_q_and_run_device_kernel_vs_golden(golden, inputs, name, tag, pname, kernel):
    """generate_q / run_device_kernel (add = +-1) as in GenericGPUCSVM (generic_csvm_tests.hpp:372-493)."""
    dt = DTYPES[tag]
    eps = np.finfo(dt).eps
    X = inputs[name + "_X"].astype(dt)
    N, d = X.shape
    P = PARAM_SETS[pname]
    key = f"{name}/{kernel}/{tag}/{pname}"
    q = backend.generate_q(prm(kernel, P), X)
    if kernel == "linear":
        assert np.array_equal(q, golden[key + "/q"])  # same fma chain in the same order: bit identical
    else:
        assert ol.rel_inf(q, golden[key + "/q"]) < 8 * eps
    # the reference's own criterion, ELEMENT-wise |a - b| < 128 eps (|a| + |b|) (EXPECT_FLOATING_POINT_VECTOR_NEAR,
    # tests/custom_test_macros.hpp:114-137), held at 16 eps here (measured: q <= 0.5, matvec <= 4.8, profiles/archive/r02_golden_elementwise.log)
    assert ol.float_near(q, golden[key + "/q"], 16)
    rhs = golden[key + "/rhs"]
    QA = float(golden[key + "/QA_cost"])
    for add, atag in ((1.0, "p1"), (-1.0, "m1")):
        got = backend.run_device_kernel(prm(kernel, P), q, np.zeros(N - 1, dt), rhs, X, QA, add)
        assert ol.rel_inf(got, golden[f"{key}/matvec_{atag}"]) < 32 * eps
        assert ol.float_near(got, golden[f"{key}/matvec_{atag}"], 16)
    # ret is accumulated into, not overwritten (svm_kernel.cpp:50-51)
    base = np.linspace(-1, 1, N - 1).astype(dt)
    got = backend.run_device_kernel(prm(kernel, P), q, base, rhs, X, QA, 1.0)
    assert ol.rel_inf(got - base, golden[f"{key}/matvec_p1"]) < 64 * eps


@pytest.mark.parametrize("name", DATASETS)
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("case", ["cg_tight", "cg_refresh", "cg_default"])
def test_solve_f64_vs_golden(golden, inputs, name, kernel, case):
    X, y = inputs[name + "_X"], inputs[name + "_y"]
    P = PARAM_SETS["def"]
    key = f"{name}/{kernel}/f64/def/{case}"
    a, rho, info = backend.solve_system_of_linear_equations(prm(kernel, P), X, y, float(golden[key + "/eps"]), int(golden[key + "/max_iter"]))
    assert abs(int(info["iterations"]) - int(golden[key + "/iterations"])) <= 3  # the stop test is a float comparison on a noisy plateau
    assert ol.rel_inf(a, golden[key + "/alpha"]) < 1e-6
    assert abs(float(rho) - float(golden[key + "/rho"])) < 1e-6 * max(1.0, abs(float(golden[key + "/rho"])))
    assert a.shape == (X.shape[0],) and abs(a.sum()) < 1e-9 * np.abs(a).sum()  # alpha_N = -sum(alpha) (csvm.cpp:180)


@pytest.mark.parametrize("name", DATASETS)
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("case", ["cg_default", "cg_refresh"])
def test_solve_f32_distance_to_fp64_truth(golden, inputs, name, kernel, case):
    X = inputs[name + "_X"].astype(np.float32)
    y = inputs[name + "_y"].astype(np.float32)
    P = PARAM_SETS["def"]
    key32, key64 = f"{name}/{kernel}/f32/def/{case}", f"{name}/{kernel}/f64/def/{case}"
    a, rho, info = backend.solve_system_of_linear_equations(prm(kernel, P), X, y, float(golden[key32 + "/eps"]), int(golden[key32 + "/max_iter"]))
    assert np.all(np.isfinite(a)) and np.isfinite(rho)
    truth = golden[key64 + "/alpha"]
    err_ours, err_ref = ol.rel_inf(a, truth), ol.rel_inf(golden[key32 + "/alpha"], truth)
    assert err_ours <= max(2 * err_ref, 1e-4), (err_ours, err_ref)
    if case == "cg_default":  # same stopping iteration as the reference at the default eps (+-1: float stop test)
        assert abs(int(info["iterations"]) - int(golden[key32 + "/iterations"])) <= 1


def _rho_from_alpha(kernel, X, y, alpha, P, oracle):
    """rho = -(y_N + QA_cost * sum(x) - q.x) (csvm.cpp:179-182) evaluated in float64 from a returned alpha; also returns the natural
    error scale of that expression, eps * (|QA_cost| sum|x| + sum|q x|)."""
    X64 = np.asarray(X, np.float64)
    kw = resolved_kw(P, X.shape[1])
    q = oracle.q(kernel, X64, **kw)
    QA = float(oracle.kernel_function(kernel, X64[-1], X64[-1], **kw)) + 1.0 / P["cost"]
    x = np.asarray(alpha, np.float64)[:-1]
    rho = -(float(y[-1]) + QA * x.sum() - q @ x)
    return rho, abs(QA) * np.abs(x).sum() + np.abs(q * x).sum()


@pytest.mark.parametrize("name", DATASETS)
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("mode", [3, 2, 1, 0])
def test_solve_f32_tight_goldens_traces_and_rho(golden, inputs, oracle, name, kernel, mode):
    """The fp32 `cg_tight` goldens (eps = 1e-5) and the per-iteration delta traces of the reference, for both Gram modes.

    What can be held tightly in fp32, measured on MI355X (profiles/archive/r02_golden_elementwise.log): delta_0 agrees with the reference to
    <= 2e-6 and delta_1 to <= 6e-5 relative; from delta_2 on the trajectories of ANY two fp32 implementations separate (up to 0.23 at
    delta_2, order 1 at delta_3: x0 = 1 makes delta_0 ~ 1e8, every step amplifies rounding noise -- the reference's own 8-thread run
    leaves its 1-thread run the same way).  Asserted: delta_0 < 1e-5, delta_1 < 1e-3; the iteration count within 2 of the
    reference's; alpha no farther from the float64 golden than max(2x the reference's fp32 alpha, 1e-4); rho CONSISTENT with the returned
    alpha (the bias formula evaluated in float64 from alpha) within 64 eps of its natural scale, and rho's distance to the float64 golden
    within what the alpha error explains (the same bound) plus the reference's own rho error."""
    X = inputs[name + "_X"].astype(np.float32)
    y = inputs[name + "_y"].astype(np.float32)
    P = PARAM_SETS["def"]
    key32, key64 = f"{name}/{kernel}/f32/def/cg_tight", f"{name}/{kernel}/f64/def/cg_tight"
    eps_cg, max_iter = float(golden[key32 + "/eps"]), int(golden[key32 + "/max_iter"])
    trace = golden[key32 + "/trace"]
    _capi.set_option("gram_mode", mode)
    with backend.ResidentProblem(prm(kernel, P), X) as prob:
        prob.cg_begin(y, eps_cg)
        d0 = prob.info()["residuum"]
        assert abs(d0 - float(golden[key32 + "/delta0"])) < 1e-5 * float(golden[key32 + "/delta0"])
        done = prob.cg_step(1)
        assert abs(prob.info()["residuum"] - trace[0]) < 1e-3 * trace[0]
        if not done:
            prob.cg_step(max_iter - 1)
        a, rho, info = prob.cg_finish()
    assert abs(int(info["iterations"]) - int(golden[key32 + "/iterations"])) <= 2
    truth, rho_truth = golden[key64 + "/alpha"], float(golden[key64 + "/rho"])
    err_ours, err_ref = ol.rel_inf(a, truth), ol.rel_inf(golden[key32 + "/alpha"], truth)
    assert err_ours <= max(2 * err_ref, 1e-4), (err_ours, err_ref)
    rho_from_alpha, scale = _rho_from_alpha(kernel, X, y, a, P, oracle)
    eps32 = np.finfo(np.float32).eps
    assert abs(float(rho) - rho_from_alpha) <= 64 * eps32 * scale, (float(rho), rho_from_alpha, scale)
    # the alpha error bounds the rho error through the same linear functional
    _, scale_err = _rho_from_alpha(kernel, X, y, np.asarray(a, np.float64) - truth, P, oracle)
    rho_err_ref = abs(float(golden[key32 + "/rho"]) - rho_truth)
    assert abs(float(rho) - rho_truth) <= scale_err + 64 * eps32 * scale + rho_err_ref


@pytest.mark.parametrize("kernel", ["linear", "rbf"])
def test_rho_of_both_gram_modes_at_4096x128(oracle, kernel):
    """at 16384 x 128 (linear) the bf16x6 mode's rho looked 20x worse than the native mode's.  The number was a RELATIVE error
    of a rho that happens to be ~1e-2 there; over seeds neither mode is systematically better (profiles/archive/r02_rho_study_8192.log,
    r02_rho_study_16384.log: |drho| 7.6e-2 vs 8.9e-3, 9.3e-3 vs 6.9e-3, 9.4e-2 vs 1.3e-1, 4.0e-2 vs 1.1e-3, 1.3e-1 vs 1.2e-1 ...; the
    reference's own fp32 solve: 4e-2 ... 7e-2).  What IS deterministic is asserted here for both modes at 4096 x 128: rho is consistent
    with the returned alpha, and its distance to the float64 solve is explained by alpha's distance (same linear functional)."""
    X, y = make_blobs_pm1(4096, 128, seed=5, dtype=np.float32)
    P = dict(degree=3, gamma=None, coef0=0.0, cost=1.0)
    p = prm(kernel, P)
    a64, rho64, _ = backend.solve_system_of_linear_equations(p, X.astype(np.float64), y.astype(np.float64), 1e-6, 400)
    eps32 = np.finfo(np.float32).eps
    errs = {}
    for mode in (2, 1, 0):
        _capi.set_option("gram_mode", mode)
        a, rho, info = backend.solve_system_of_linear_equations(p, X, y, 1e-6, 400)
        assert info["gram_mode"] == mode
        rho_from_alpha, scale = _rho_from_alpha(kernel, X, y, a, P, oracle)
        assert abs(float(rho) - rho_from_alpha) <= 64 * eps32 * scale
        _, scale_err = _rho_from_alpha(kernel, X, y, np.asarray(a, np.float64) - a64, P, oracle)
        assert abs(float(rho) - float(rho64)) <= scale_err + 64 * eps32 * scale
        errs[mode] = ol.rel_inf(a, a64)
    # the yardstick for an fp32 solve of this system is the fp32 CPU solve of the same recipe (the oracle: the reference's OpenMP kernels
    # restated), whose distance to the float64 solution both Gram modes must not exceed by more than 2x
    a_cpu, _, _ = oracle.solve(kernel, X, y, 1e-6, 400, degree=3, gamma=1.0 / 128, coef0=0.0, cost=1.0)
    err_cpu = ol.rel_inf(a_cpu, a64)
    assert all(errs[m] <= max(2 * err_cpu, 1e-3) for m in (2, 1, 0)), (errs, err_cpu)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_trivial_system_known_answer(dt):
    """GenericCSVM.solve_system_of_linear_equations_trivial (generic_csvm_tests.hpp:99-137): alpha == y, rho ~ 0."""
    A = (np.sqrt(dt(1.0) - dt(1.0) / dt(2.0)) * np.eye(4)).astype(dt)
    rhs = np.array([1, -1, 1, -1], dtype=dt)
    for p in (Parameter(kernel_type="linear", cost=2.0), Parameter(kernel_type="polynomial", degree=1, gamma=1.0, coef0=0.0, cost=2.0)):
        x, rho, info = backend.solve_system_of_linear_equations(p, A, rhs, 1e-5, 4)
        assert ol.float_near(x, rhs)
        assert abs(float(rho)) < 8 * np.finfo(dt).eps


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_predict_values_known_answer(dt):
    """GenericCSVM.predict_values (generic_csvm_tests.hpp:149-195)."""
    sv = np.eye(4, dtype=dt)
    w8 = np.array([1, -1, 1, -1], dtype=dt)
    pts = np.array([[1, 1, 1, 1], [1, -1, 1, -1]], dtype=dt)
    out, w = backend.predict_values(Parameter(kernel_type="linear", cost=2.0), sv, w8, 0.0, None, pts)
    assert ol.float_near(out, np.array([0, 4], dtype=dt)) and ol.float_near(w, w8)
    out, w = backend.predict_values(Parameter(kernel_type="polynomial", degree=1, gamma=1.0, coef0=0.0, cost=2.0), sv, w8, 0.0, None, pts)
    assert ol.float_near(out, np.array([0, 4], dtype=dt)) and w is None


@pytest.mark.parametrize("kernel", ["polynomial", "rbf"])
@pytest.mark.parametrize("nsv, npts, d", [(1300, 700, 128), (777, 130, 40), (520, 1025, 384), (300, 200, 500), (700, 520, 1100)])
def test_predict_values_on_the_bf16_matrix_cores(oracle, kernel, nsv, npts, d):
    """predict_values is the rectangular instance of the tile kernel (HIP/predict_kernel.hip.hpp:63-117: rows = points to predict,
    columns = support vectors).  In fp32 it runs on the f16x3 kernels (gram_mode 2 / 3, up to 512 features in one pass, rbf 384) or the
    bf16x6 kernels (gram_mode 1, up to 384 features in one pass), beyond that on the kernel that walks feature panels inside a tile (either
    plane kind); every mode must match the float64 oracle on the scale of each point's summands."""
    rng = np.random.default_rng(nsv + d)
    sv = rng.uniform(-1, 1, size=(nsv, d)).astype(np.float32)
    alpha = rng.uniform(-1, 1, size=nsv).astype(np.float32)
    pts = rng.uniform(-1, 1, size=(npts, d)).astype(np.float32)
    kw = dict(degree=3, gamma=1.0 / d, coef0=0.5)
    p = Parameter(kernel_type=kernel, degree=3, gamma=1.0 / d, coef0=0.5)
    want, _ = oracle.predict_values(kernel, sv.astype(np.float64), alpha.astype(np.float64), 0.125, pts.astype(np.float64), **kw)
    G = pts.astype(np.float64) @ sv.astype(np.float64).T
    if kernel == "rbf":
        sq_s, sq_p = np.einsum("ij,ij->i", sv, sv, dtype=np.float64), np.einsum("ij,ij->i", pts, pts, dtype=np.float64)
        K = np.exp(-(1.0 / d) * np.maximum(sq_p[:, None] + sq_s[None, :] - 2.0 * G, 0.0))
    else:
        K = (G / d + 0.5) ** 3
    scale = np.abs(K) @ np.abs(alpha.astype(np.float64)) + 0.125
    out = {}
    for mode in (3, 2, 1, 0):
        _capi.set_option("gram_mode", mode)
        out[mode], _ = backend.predict_values(p, sv, alpha, 0.125, None, pts)
        assert np.max(np.abs(out[mode] - want) / scale) < 16 * np.finfo(np.float32).eps, mode
    assert np.array_equal(out[3], out[2])          # uniform [-1, 1] data passes the f16 representability check: the default is f16x3
    assert not np.array_equal(out[0], out[2]) and not np.array_equal(out[0], out[1]) and not np.array_equal(out[1], out[2])  # three different kernels ran


@pytest.mark.parametrize("d", [37, 300])
@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("kernel", KERNELS)
def test_predict_values_and_calculate_w_vs_oracle(oracle, kernel, dt, d):
    """(300 features: beyond the one-pass fp64 kernels -- rbf / polynomial take the full-square instance of the panel kernel)"""
    rng = np.random.default_rng(3)
    sv = rng.uniform(-1, 1, size=(301, d)).astype(dt)
    alpha = rng.uniform(-1, 1, size=301).astype(dt)
    pts = rng.uniform(-1, 1, size=(157, d)).astype(dt)
    kw = dict(degree=3, gamma=1.0 / d, coef0=0.5)
    p = Parameter(kernel_type=kernel, degree=3, gamma=1.0 / d, coef0=0.5)
    want, w_want = oracle.predict_values(kernel, sv, alpha, 0.125, pts, **kw)
    got, w = backend.predict_values(p, sv, alpha, 0.125, None, pts)
    scale = np.abs(alpha).sum()  # the sums cancel: compare on the scale of the summands
    assert np.max(np.abs(got - want)) < 16 * np.finfo(dt).eps * scale
    if kernel == "linear":
        assert ol.rel_inf(w, w_want) < 8 * np.finfo(dt).eps
        assert ol.rel_inf(backend.calculate_w(sv, alpha), w_want) < 8 * np.finfo(dt).eps
        got2, _ = backend.predict_values(p, sv, alpha, 0.125, w, pts)  # cached w is used (csvm.cpp:204-207)
        assert np.array_equal(got, got2)


@pytest.mark.parametrize("N, d", [(2, 1), (3, 5), (129, 33), (130, 32), (257, 3), (384, 64), (385, 200), (1000, 130)])
@pytest.mark.parametrize("kernel", KERNELS)
def test_ragged_sizes_vs_oracle(oracle, kernel, N, d):
    """Tile (128) and k-chunk (32 / 16) boundaries, N-1 a multiple of the tile, a single row, a single feature."""
    for dt in (np.float32, np.float64):
        eps = np.finfo(dt).eps
        rng = np.random.default_rng(N * 1000 + d)
        X = rng.uniform(-1, 1, size=(N, d)).astype(dt)
        kw = dict(degree=2, gamma=0.5 / d, coef0=1.0)
        p = Parameter(kernel_type=kernel, degree=2, gamma=0.5 / d, coef0=1.0, cost=0.5)
        q = backend.generate_q(p, X)
        q_ref = oracle.q(kernel, X, **kw)
        assert ol.rel_inf(q, q_ref) < 8 * eps
        rhs = rng.uniform(1, 2, size=N - 1).astype(dt)
        QA = float(dt(oracle.kernel_function(kernel, X[-1], X[-1], **kw)) + dt(2.0))
        got = backend.run_device_kernel(p, q, np.zeros(N - 1, dt), rhs, X, QA, -1.0)
        want = oracle.matvec(kernel, X, q_ref, rhs, np.zeros(N - 1, dt), QA, 2.0, -1.0, **kw)
        assert ol.rel_inf(got, want) < 32 * eps


def test_negative_and_zero_polynomial_degree(oracle):
    """(gamma*0 + 0)^degree on zero-padded columns must not leak inf/nan into valid rows."""
    rng = np.random.default_rng(9)
    X = rng.uniform(0.5, 1.5, size=(200, 7))
    for degree in (-2, 0, 1, 5):
        kw = dict(degree=degree, gamma=0.25, coef0=0.0)
        p = Parameter(kernel_type="polynomial", degree=degree, gamma=0.25, coef0=0.0)
        q = backend.generate_q(p, X)
        assert ol.rel_inf(q, oracle.q("polynomial", X, **kw)) < 1e-13
        rhs = rng.uniform(1, 2, size=199)
        QA = float(oracle.kernel_function("polynomial", X[-1], X[-1], **kw)) + 1.0
        got = backend.run_device_kernel(p, q, np.zeros(199), rhs, X, QA, 1.0)
        want = oracle.matvec("polynomial", X, q, rhs, np.zeros(199), QA, 1.0, 1.0, **kw)
        assert np.all(np.isfinite(got)) and ol.rel_inf(got, want) < 1e-12


def test_rbf_direct_form_agrees_with_matrix_core_form(oracle):
    """The formula-exact (x_i - x_j)^2 vector-ALU kernel and the norm-expansion matrix-core kernel are two routes to the
    same numbers; both must match the oracle."""
    X, _ = make_blobs_pm1(700, 100, seed=5, dtype=np.float32)
    p = Parameter(kernel_type="rbf")
    rhs = np.random.default_rng(1).uniform(1, 2, size=699).astype(np.float32)
    q = backend.generate_q(p, X)
    want = oracle.matvec("rbf", X, q, rhs, np.zeros(699, np.float32), 2.0, 1.0, 1.0, gamma=0.01)
    a = backend.run_device_kernel(p, q, np.zeros(699, np.float32), rhs, X, 2.0, 1.0)
    _capi.set_option("rbf_form", 1)
    try:
        b = backend.run_device_kernel(p, q, np.zeros(699, np.float32), rhs, X, 2.0, 1.0)
        assert _capi.get_option("rbf_form") == 1
    finally:
        _capi.set_option("rbf_form", 0)
    eps = np.finfo(np.float32).eps
    assert ol.rel_inf(a, want) < 32 * eps and ol.rel_inf(b, want) < 32 * eps


@pytest.mark.parametrize("case", ["gamma1", "gamma10", "unscaled"])
def test_rbf_large_exponent_scale_switches_to_the_direct_kernel(oracle, case):
    """the matrix-core rbf form evaluates the exponent as c_i + c_j + x_i'.x_j'; its absolute error is ~2^-24 R2 with
    R2 = 2 gamma log2(e) max|x - mean|^2, which pairs of NEARBY points (K ~ 1) see as a relative error of K.  With rbf_form = 0 the
    library measures R2 and runs the formula-exact (x_i - x_j)^2 kernel above RBF_DIRECT_ABOVE = 32 (lssvm_problem.hip.hpp).  Cases: gamma = 1 and 10 at
    128 features on [-1,1]-scaled data (gamma = 1 lands just BELOW the threshold and stays on the matrix cores), and unscaled data (features of order 30) at gamma = 1/d.  Every data set carries near-duplicate
    points, the pairs that lose digits.  Asserted against the float64 oracle on the scale of each row's summands: the automatic choice
    stays below 16 eps; the forced matrix-core form (rbf_form = 2) is measurably worse by roughly R2 / 32 -- the reason for the switch --
    yet bounded by 2^-22 R2.  predict_values follows the same rule."""
    rng = np.random.default_rng(12)
    N, d = 1500, 128
    X, y = make_blobs_pm1(N, d, seed=3, dtype=np.float32)
    if case == "unscaled":
        X = (X * 30.0).astype(np.float32)
        gamma = 1.0 / d
    else:
        gamma = 1.0 if case == "gamma1" else 10.0
    X[1::2] = (X[0::2] + rng.normal(0, 1e-3, size=X[0::2].shape) * (30.0 if case == "unscaled" else 1.0)).astype(np.float32)  # near-duplicate pairs
    p = Parameter(kernel_type="rbf", gamma=gamma)
    n = N - 1
    v = rng.uniform(-1, 1, size=n).astype(np.float32)
    X64, v64 = X.astype(np.float64), v.astype(np.float64)
    q64 = oracle.q("rbf", X64, gamma=gamma)
    want = oracle.matvec("rbf", X64, q64, v64, np.zeros(n), 2.0, 1.0, 1.0, gamma=gamma)
    sq = np.einsum("ij,ij->i", X64, X64)
    K = np.exp(-gamma * np.maximum(sq[:n, None] + sq[None, :n] - 2.0 * (X64[:n] @ X64[:n].T), 0.0))
    absv = np.abs(v64)
    scale = K @ absv + (2.0 + np.abs(q64)) * absv.sum() + np.abs(q64) @ absv + absv
    eps = np.finfo(np.float32).eps
    errs = {}
    for form in (0, 2, 3, 1):
        _capi.set_option("rbf_form", form)
        with backend.ResidentProblem(p, X) as prob:
            info = prob.info()
            got = prob.matvec(v, np.zeros(n, np.float32), 1.0)
        if form == 0:
            r2 = info["rbf_exponent_scale"]
            grid = info["gram_mode"] == 3
            # round 5: above the threshold the automatic choice is the matrix cores on GRID planes (KT_RBFG), no longer the direct kernel (128 features, R2 <= 4 096)
            assert info["rbf_direct"] == 0 and grid == (r2 > 32) and (case == "gamma1" or grid)  # gamma = 1 sits just below the threshold (R2 ~ 26)
        elif form == 1:
            assert info["rbf_direct"] == 1
        else:
            assert info["rbf_direct"] == 0 and info["gram_mode"] == (3 if form == 3 else 2)
        # the GPU's own q (fp32) enters the rank-1 terms: compare K*v + rank-1 with the float64 product of the SAME fp32 data
        errs[form] = float(np.max(np.abs(got - want) / scale))
    print(f"\n{case}: exponent scale {r2:.1f}: automatic {errs[0] / eps:.2f} eps, grid planes {errs[3] / eps:.2f} eps, direct kernel {errs[1] / eps:.2f} eps, norm expansion {errs[2] / eps:.1f} eps")
    assert errs[0] < (16 if grid else 32) * eps, (case, errs, r2)  # below the threshold the norm expansion keeps 32 eps: the threshold's promise
    assert errs[3] < 16 * eps and errs[1] < 16 * eps, (case, errs, r2)  # the grid planes hold the direct form's bar at every scale
    assert errs[2] < 2.0 ** -22 * max(r2, 32.0), (case, errs, r2)
    # predict_values takes the same decision
    alpha = rng.uniform(-1, 1, size=N).astype(np.float32)
    pts = X[:64] + np.float32(1e-3)
    want_p, _ = oracle.predict_values("rbf", X64, alpha.astype(np.float64), 0.25, pts.astype(np.float64), gamma=gamma)
    Kp = np.exp(-gamma * np.maximum(np.einsum("ij,ij->i", pts.astype(np.float64), pts.astype(np.float64))[:, None] + sq[None, :] - 2.0 * (pts.astype(np.float64) @ X64.T), 0.0))
    scale_p = Kp @ np.abs(alpha.astype(np.float64)) + 0.25
    _capi.set_option("rbf_form", 0)
    got_p, _ = backend.predict_values(p, X, alpha, 0.25, None, pts)
    assert np.max(np.abs(got_p - want_p) / scale_p) < (16 if r2 > 32 else 32) * eps  # (predict_values keeps the direct kernel above the threshold)


@pytest.mark.parametrize("d, gamma, spread, devices", [(128, 2.0, 1.0, None), (128, 30.0, 1.0, None), (64, 8.0, 1.0, None), (100, 1.0 / 100, 25.0, None), (40, 60.0, 1.0, None),
                                                       (128, 4.0, 1.0, [0, 0, 0]), (96, 100.0, 1.0, None), (96, 250.0, 1.0, None), (256, 2.0, 1.0, None), (200, 6.0, 1.0, [0, 0]), (384, 1.5, 1.0, None), (330, 20.0, 1.0, None), (640, 1.0, 1.0, None), (1000, 0.5, 1.0, [0, 0]), (513, 3.0, 1.0, None)])
def test_rbf_on_grid_planes_keeps_the_direct_forms_accuracy_on_the_matrix_cores(oracle, d, gamma, spread, devices):
    """Round 5.  Exponent scales R2 = 2 gamma log2(e) max|x - mean|^2 from 40 to 4 000 on 40 ... 1 000 features, data with
    near-duplicate points (the pairs that lose digits in the norm expansion) and far ones.  The automatic choice is the matrix cores on GRID planes (tile_matvec_f32_g6h:
    x = h + s1 + s2 with h on a grid, the accumulators started from sigma^2 (ch_i + ch_j) and fed the h.h products first, so that the large terms cancel EXACTLY);
    asserted against the float64 oracle on the scale of each row's summands, symmetric and full square, one device and three shards: below 16 eps like the direct
    kernel -- which the same data also runs (rbf_form = 1) -- where the norm expansion (rbf_form = 2) is off by tens to thousands of eps; and a CG solve on the grid
    planes ends where the direct kernel's ends."""
    rng = np.random.default_rng(5)
    N = 1700
    X, y = make_blobs_pm1(N, d, seed=6, dtype=np.float32)
    X = (X * spread).astype(np.float32)
    X[1::2] = (X[0::2] + rng.normal(0, 2e-3 * spread, size=X[0::2].shape)).astype(np.float32)  # near-duplicate pairs
    p = Parameter(kernel_type="rbf", gamma=gamma)
    n = N - 1
    X64 = X.astype(np.float64)
    q64 = oracle.q("rbf", X64, gamma=gamma)
    # a right-hand side orthogonal to 1 and to q: the rank-1 terms of Abar v (QA S - q.v, S q_i; three orders of magnitude above K v on such data) vanish, and the
    # comparison sees the kernel matrix itself
    v64 = rng.uniform(-1, 1, size=n)
    basis = np.linalg.qr(np.stack([np.ones(n), q64], axis=1))[0]
    for _ in range(2):
        v64 = v64 - basis @ (basis.T @ v64)
    v = v64.astype(np.float32)
    v64 = v.astype(np.float64)
    want = oracle.matvec("rbf", X64, q64, v64, np.zeros(n), 2.0, 1.0, 1.0, gamma=gamma)
    sq = np.einsum("ij,ij->i", X64, X64)
    K = np.exp(-gamma * np.maximum(sq[:n, None] + sq[None, :n] - 2.0 * (X64[:n] @ X64[:n].T), 0.0))
    absv = np.abs(v64)
    scale = K @ absv + (2.0 + np.abs(q64)) * absv.sum() + np.abs(q64) @ absv + absv
    scale_k = K @ absv + absv + (2.0 + np.abs(q64)) * abs(v64.sum()) + abs(q64 @ v64)  # (what is left of the rank-1 terms after the fp32 rounding of v)
    eps = np.finfo(np.float32).eps
    errs, errs_k = {}, {}
    for name, form, sym in (("automatic", 0, 1), ("automatic, full square", 0, 0), ("direct", 1, 1), ("norm expansion", 2, 1)):
        _capi.set_option("rbf_form", form)
        _capi.set_option("symmetric", sym)
        with backend.ResidentProblem(p, X, devices=devices) as prob:
            info = prob.info()
            got = prob.matvec(v, np.zeros(n, np.float32), 1.0)
            again = prob.matvec(v, np.zeros(n, np.float32), 1.0)
        assert np.array_equal(got, again)
        r2 = info["rbf_exponent_scale"]
        if form == 0:
            assert 32 < r2 <= 4096 and info["gram_mode"] == 3 and info["rbf_direct"] == 0 and info["symmetric"] == sym, (r2, info)
        errs[name] = float(np.max(np.abs(got - want) / scale))
        errs_k[name] = float(np.max(np.abs(got - want) / scale_k))
    print(f"\n{N} x {d}, gamma {gamma:g}, exponent scale {r2:.0f}: on the scale of all summands: " + ", ".join(f"{k} {e / eps:.2f} eps" for k, e in errs.items())
          + "; of the K v summands alone: " + ", ".join(f"{k} {e / eps:.1f} eps" for k, e in errs_k.items()))
    assert errs["automatic"] < 16 * eps and errs["automatic, full square"] < 16 * eps and errs["direct"] < 16 * eps, errs
    # the sharper yardstick: sum_j K_ij |v_j| alone (the rank-1 terms, evaluated in double by the library, are 10 ... 1000 x larger and would hide the kernel's error)
    # (measured: 1.3 ... 5.8 eps up to an exponent scale of 1 600, 12.8 eps at 7 950 -- the error grows with the cross terms |h||s| ~ R2 sqrt(d) 2^-12 -- which is why the
    # automatic choice ends at 4 096; the direct kernel: 0.8 ... 2.5 eps; the norm expansion: 33 ... 6 600 eps)
    assert errs_k["automatic"] < 16 * eps and errs_k["automatic, full square"] < 16 * eps and errs_k["direct"] < 16 * eps, errs_k
    assert errs_k["norm expansion"] > 4 * errs_k["automatic"], errs_k  # what the grid planes are for
    if devices is None:
        # predict_values takes the same decision (grid planes of the support vectors and of the points, one grid and one scale for both)
        alpha = rng.uniform(-1, 1, size=N).astype(np.float32)
        pts = (X[:200] + rng.normal(0, 3e-3 * spread, size=(200, d))).astype(np.float32)
        want_p, _ = oracle.predict_values("rbf", X64, alpha.astype(np.float64), 0.25, pts.astype(np.float64), gamma=gamma)
        p64 = pts.astype(np.float64)
        Kp = np.exp(-gamma * np.maximum(np.einsum("ij,ij->i", p64, p64)[:, None] + sq[None, :] - 2.0 * (p64 @ X64.T), 0.0))
        scale_p = Kp @ np.abs(alpha.astype(np.float64)) + 0.25
        perr = {}
        for form in (0, 1, 2):
            _capi.set_option("rbf_form", form)
            got_p, _ = backend.predict_values(p, X, alpha, 0.25, None, pts)
            perr[form] = float(np.max(np.abs(got_p - want_p) / scale_p))
        print(f"predict_values: automatic {perr[0] / eps:.2f} eps, direct {perr[1] / eps:.2f} eps, norm expansion {perr[2] / eps:.1f} eps")
        assert perr[0] < 16 * eps and perr[1] < 16 * eps and perr[2] > 4 * perr[0], perr
        sol = {}
        for form in (0, 1):
            _capi.set_option("rbf_form", form)
            _capi.set_option("symmetric", 1)
            sol[form] = backend.solve_system_of_linear_equations(p, X, y, 1e-6, 300)
        assert abs(int(sol[0][2]["iterations"]) - int(sol[1][2]["iterations"])) <= 2
        assert ol.rel_inf(sol[0][0], sol[1][0]) < 1e-3 and abs(float(sol[0][1]) - float(sol[1][1])) < 1e-3 * max(1.0, abs(float(sol[1][1])))


def test_rbf_folded_and_unfolded_column_records_agree(oracle):
    """fp32 rbf on the 16x16x32 bf16x6 kernels: option rbf_fold (default 1) starts the accumulators from c_i as the first MFMA's C operand and
    carries 2^c_j in the column record; rbf_fold = 0 starts them at c_i + c_j with vector adds.  Two evaluations of the same numbers: both
    within 16 eps of the float64 product on the scale of each row's summands, for the symmetric variant, the full square and predict_values."""
    N, d = 2500, 128
    X, _ = make_blobs_pm1(N, d, seed=9, dtype=np.float32)
    p = Parameter(kernel_type="rbf", gamma=0.5)  # an exponent scale well above the default gamma's
    _capi.set_option("rbf_form", 2)               # ... kept on the matrix cores whatever it is
    rows = np.array([0, 1, 127, 128, 129, 1000, 2047, N - 2])
    eps = np.finfo(np.float32).eps
    alpha = np.random.default_rng(3).uniform(-1, 1, size=N).astype(np.float32)
    pts = X[:200] + np.float32(0.01)
    want_p, _ = oracle.predict_values("rbf", X.astype(np.float64), alpha.astype(np.float64), 0.5, pts.astype(np.float64), gamma=0.5)
    for sym in (1, 0):
        _capi.set_option("symmetric", sym)
        for fold in (1, 0):
            _capi.set_option("rbf_fold", fold)
            with backend.ResidentProblem(p, X) as prob:
                info = prob.info()
                assert info["rbf_direct"] == 0 and info["gram_mode"] in (1, 2) and 4 < info["rbf_exponent_scale"] < 200
                err, _, _ = _sampled_rows_vs_oracle_gamma(oracle, prob, X, rows, 0.5)
            assert err < 2.0 ** -22 * max(info["rbf_exponent_scale"], 32.0), (sym, fold, err / eps)  # the matrix-core bound of INTEGRATION.md section 6
    for fold in (1, 0):
        _capi.set_option("rbf_fold", fold)
        got_p, _ = backend.predict_values(p, X, alpha, 0.5, None, pts)
        assert np.max(np.abs(got_p - want_p)) < 16 * eps * np.abs(alpha).sum()


def _sampled_rows_vs_oracle_gamma(oracle, prob, X, rows, gamma):
    """_sampled_rows_vs_oracle for the rbf kernel with an explicit gamma"""
    N, d = X.shape
    n = N - 1
    rng = np.random.default_rng(0)
    rhs = rng.uniform(-1, 1, size=n).astype(X.dtype)
    q, QA = prob.q()
    got = prob.matvec(rhs, np.zeros(n, X.dtype), 1.0)
    X64, q64, rhs64 = X.astype(np.float64), q.astype(np.float64), rhs.astype(np.float64)
    want = np.zeros(n)
    for r in rows:
        want = oracle.matvec_rows("rbf", X64, q64, rhs64, want, float(QA), 1.0, 1.0, int(r), int(r) + 1, gamma=gamma)
    sq = np.einsum("ij,ij->i", X64, X64)
    K = np.exp(-gamma * np.maximum(sq[rows, None] + sq[None, :n] - 2.0 * (X64[rows] @ X64[:n].T), 0.0))
    absd = np.abs(rhs64)
    scale = K @ absd + (abs(float(QA)) + np.abs(q64[rows])) * absd.sum() + np.abs(q64) @ absd + absd[rows]
    return float(np.max(np.abs(got[rows] - want[rows]) / scale)), got, rhs


def test_rbf_uncentred_data_with_large_offset(oracle):
    """Features with a common offset far larger than their spread: the centring keeps the norm expansion accurate."""
    rng = np.random.default_rng(2)
    X = (100.0 + rng.uniform(-1, 1, size=(300, 16))).astype(np.float32)
    p = Parameter(kernel_type="rbf", gamma=0.05)
    q = backend.generate_q(p, X)
    rhs = rng.uniform(1, 2, size=299).astype(np.float32)
    got = backend.run_device_kernel(p, q, np.zeros(299, np.float32), rhs, X, 2.0, 1.0)
    q64 = oracle.q("rbf", X.astype(np.float64), gamma=0.05)
    want = oracle.matvec("rbf", X.astype(np.float64), q64, rhs.astype(np.float64), np.zeros(299), 2.0, 1.0, 1.0, gamma=0.05)
    assert ol.rel_inf(got, want) < 2e-4  # limited by rounding x - mean to fp32 at |x| ~ 100


def test_preconditions_raise_like_the_reference():
    X = np.ones((4, 3), dtype=np.float32)
    y = np.array([1, -1, 1, -1], dtype=np.float32)
    with pytest.raises(InvalidParameterError, match="stopping criterion"):
        backend.solve_system_of_linear_equations(Parameter(), X, y, -1.0, 4)
    with pytest.raises(InvalidParameterError, match="add must either be"):  # svm_kernel.cpp:28
        backend.run_device_kernel(Parameter(), np.zeros(3, np.float32), np.zeros(3, np.float32), np.ones(3, np.float32), X, 1.0, 2.0)
    with pytest.raises(InvalidParameterError, match="Sizes mismatch"):       # svm_kernel.cpp:24
        backend.run_device_kernel(Parameter(), np.zeros(2, np.float32), np.zeros(3, np.float32), np.ones(3, np.float32), X, 1.0, 1.0)


@pytest.mark.parametrize("kernel", KERNELS)
def test_results_are_bitwise_reproducible_and_independent_of_the_work_split(kernel):
    """No atomics anywhere: two runs agree bit for bit, and so do different column-chunk sizes per work item only through
    the fixed slab order (the chunking changes the association, hence allclose; equal chunking => equal bits)."""
    X, y = make_blobs_pm1(1500, 40, seed=8, dtype=np.float32)
    p = Parameter(kernel_type=kernel)
    a1, r1, _ = backend.solve_system_of_linear_equations(p, X, y, 1e-30, 7)  # default (symmetric) variant: fixed-order slabs, no atomics
    a2, r2, _ = backend.solve_system_of_linear_equations(p, X, y, 1e-30, 7)
    assert np.array_equal(a1, a2) and r1 == r2
    _capi.set_option("j_chunk_tiles", 3)
    try:
        q = backend.generate_q(p, X)
        rhs = np.linspace(1, 2, 1499).astype(np.float32)
        m3 = backend.run_device_kernel(p, q, np.zeros(1499, np.float32), rhs, X, 2.0, 1.0)
    finally:
        _capi.set_option("j_chunk_tiles", 0)
    m16 = backend.run_device_kernel(p, q, np.zeros(1499, np.float32), rhs, X, 2.0, 1.0)
    assert ol.rel_inf(m3, m16) < 16 * np.finfo(np.float32).eps


@pytest.mark.parametrize("kernel, d, shards", [("rbf", 128, 1), ("linear", 200, 1), ("polynomial", 64, 1), ("rbf", 96, 3)])
def test_a_head_of_short_column_chunks_changes_only_the_association(oracle, kernel, d, shards):
    """Round 5: the 256-row workgroups may cut the FIRST column chunks of every row pair short (option j_chunk_head = 1024 count + tiles; chosen by the replayed
    dispatch for launches of a few rounds): their items fill the last dispatch round.  A chunk map is a work split -- same tiles, same fixed-order slabs -- so
    every head gives the float64 oracle's rows within the kernel-level bar, agrees with the uniform split to the association, and repeats bit for bit; and
    the automatic choice (j_chunk_tiles = 0, whatever it picks for this shape) does too."""
    N = 9100  # 72 row blocks: the 256-row workgroups run from 64 on
    X, _ = make_blobs_pm1(N, d, seed=13, dtype=np.float32)
    n = N - 1
    p = Parameter(kernel_type=kernel, degree=3)
    eps = np.finfo(np.float32).eps
    rows = np.array([0, 1, 127, 128, 255, 256, 257, 640, 4000, 8191, 8192, 9000, n - 1])
    devices = [0] * shards if shards > 1 else None
    results = {}
    try:
        for name, tiles, head in (("uniform", 8, 0), ("head 3 x 2", 8, 3 * 1024 + 2), ("head 1 x 5", 16, 1 * 1024 + 5), ("head 6 x 3", 24, 6 * 1024 + 3), ("head as long as the triangle allows", 40, 2 * 1024 + 30),
                                  ("automatic without a head", 0, 0), ("automatic", 0, 1)):
            _capi.set_option("j_chunk_tiles", tiles)
            _capi.set_option("j_chunk_head", head)
            with backend.ResidentProblem(p, X, devices=devices) as prob:
                err, got, rhs = _sampled_rows_vs_oracle(oracle, prob, kernel, X, rows)
                again = prob.matvec(rhs, np.zeros(n, np.float32), 1.0)
            assert err < 16 * eps, (name, err / eps)
            assert np.array_equal(got, again), name
            results[name] = got
    finally:
        _capi.set_option("j_chunk_tiles", 0)
        _capi.set_option("j_chunk_head", 1)
    base = results["uniform"]
    for name, got in results.items():
        assert ol.rel_inf(got, base) < 64 * eps, name  # (on the scale of the RESULT, whose rank-1 terms cancel: the bar of the sharded runs; measured 18 eps)


@pytest.mark.parametrize("kernel, d, shards, band_mb", [("rbf", 128, 1, 0), ("linear", 200, 1, 0), ("polynomial", 64, 1, 1), ("rbf", 96, 3, 0), ("rbf", 128, 1, 1)])
def test_persistent_launches_give_the_bits_of_one_workgroup_per_item(monkeypatch, kernel, d, shards, band_mb):
    """Round 5: launches of the 256-row kernel with more work items than CUs are PERSISTENT -- one workgroup per CU, the items drawn from per-XCD counters (own lane
    first, then the fullest; two counter sets per problem, a launch draws from one and zeroes the other) instead of one workgroup per item dealt by the hardware.  Which
    CU evaluates an item must not change a bit: every item owns its slab rows and records.  Compared with the former launches (LSSVM_MI355_PAIR_QUEUE=0, read when a
    problem is created) over a fixed-length solve that crosses several matvecs -- every launch the counters must have been re-armed by its predecessor --, with several
    band launches per matvec (1 MiB bands), feature-panel passes (200 features) and three shards on one device (three problems, three streams, three counter pairs)."""
    N = 16_500  # 129 row blocks -> 65 block pairs; with 2-tile items some 2 100 items per launch (256 CUs)
    X, y = make_blobs_pm1(N, d, seed=21, dtype=np.float32)
    p = Parameter(kernel_type=kernel, degree=3)
    devices = [0] * shards if shards > 1 else None
    _capi.set_option("j_chunk_tiles", 2)
    if band_mb:
        _capi.set_option("colslab_band_mb", band_mb)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("LSSVM_MI355_PAIR_QUEUE", mode)
        with backend.ResidentProblem(p, X, devices=devices) as prob:
            info = prob.info()
            prob.cg_begin(y, 1e-30)
            prob.cg_step(7)
            alpha, rho, fin = prob.cg_finish()
        assert info["symmetric"] == 1
        assert (info["persistent_launches"] > 0) == (mode == "1"), info
        if mode == "1":
            assert info["persistent_launches"] <= info["tile_launches_per_matvec"] and (band_mb == 0 or info["tile_launches_per_matvec"] > 1)
        out[mode] = (alpha, rho, fin["residuum"])
    assert np.array_equal(out["0"][0], out["1"][0]) and out["0"][1] == out["1"][1] and out["0"][2] == out["1"][2]


def test_resident_problem_stepping_equals_one_shot():
    X, y = make_blobs_pm1(900, 24, seed=4, dtype=np.float64)
    p = Parameter(kernel_type="rbf")
    a, rho, info = backend.solve_system_of_linear_equations(p, X, y, 1e-30, 60)
    with backend.ResidentProblem(p, X) as prob:
        prob.cg_begin(y, 1e-30)
        for k in (1, 9, 39, 11):  # crosses the iteration-49 refresh inside a cg_step call
            prob.cg_step(k)
        a2, rho2, info2 = prob.cg_finish()
        assert info2["iterations"] == 60 and info2["matvec_launches"] == 62  # 60 + initial residual + one refresh
    assert np.array_equal(a, a2) and rho == rho2


@pytest.mark.parametrize("kernel, dtype, devices", [("rbf", np.float32, None), ("linear", np.float64, None), ("polynomial", np.float32, [0, 0, 0])])
def test_matvecs_enqueued_ahead_of_the_stop_test_change_nothing(kernel, dtype, devices):
    """Option enqueue_ahead_below_us: the direction update and the next implicit matvec enter the queue before the host has read the
    stop test.  Same bits, same iteration count, same stop as the fully synchronous loop -- in a solve that converges (one discarded
    matvec), in one that runs out of iterations, stepped in pieces across the residual refresh, and sharded."""
    X, y = make_blobs_pm1(1500, 40, seed=21, dtype=dtype)
    p = Parameter(kernel_type=kernel, cost=3.0)
    runs = {}
    for ahead in (0, 5000):
        _capi.set_option("enqueue_ahead_below_us", ahead)
        conv = backend.solve_system_of_linear_equations(p, X, y, 1e-3, 1500, devices=devices)
        fixed = backend.solve_system_of_linear_equations(p, X, y, 1e-30, 57, devices=devices)
        with backend.ResidentProblem(p, X, **({"devices": devices} if devices else {})) as prob:
            prob.cg_begin(y, 1e-30)
            for k in (3, 1, 44, 9):
                prob.cg_step(k)
            stepped = prob.cg_finish()
        runs[ahead] = (conv, fixed, stepped)
    (c0, f0, s0), (c1, f1, s1) = runs[0], runs[5000]
    assert c0[2]["converged"] == 1 and c1[2]["converged"] == 1 and c0[2]["iterations"] == c1[2]["iterations"] < 1500
    assert c1[2]["matvec_launches"] == c0[2]["matvec_launches"] + 1  # the matvec that was in the queue when the stop test fired
    for (a, rho, info), (b, rho_b, info_b) in ((c0, c1), (f0, f1), (s0, s1), (f0, s1)):
        assert np.array_equal(a, b) and rho == rho_b and info["iterations"] == info_b["iterations"] and info["residuum"] == info_b["residuum"]
    assert f1[2]["matvec_launches"] == f0[2]["matvec_launches"] == 57 + 2


def test_sub_sampled_rows_of_a_large_matvec_vs_oracle(oracle):
    """BASELINE configs[1] shape (50 000 x 128 rbf fp32): the full CPU product takes minutes, so 256 seeded rows of one
    implicit matvec are checked against the oracle's row-owned product."""
    X, _ = make_blobs_pm1(50_000, 128, seed=42, dtype=np.float32)
    p = Parameter(kernel_type="rbf")
    rng = np.random.default_rng(0)
    rhs = rng.uniform(-1, 1, size=49_999).astype(np.float32)
    with backend.ResidentProblem(p, X) as prob:
        q, QA = prob.q()
        got = prob.matvec(rhs, np.zeros(49_999, np.float32), 1.0)
    rows = np.sort(rng.choice(49_999, size=256, replace=False))
    X64, q64, rhs64 = X.astype(np.float64), q.astype(np.float64), rhs.astype(np.float64)
    want = np.zeros(49_999)
    for r in rows:
        want = oracle.matvec_rows("rbf", X64, q64, rhs64, want, 2.0, 1.0, 1.0, int(r), int(r) + 1, gamma=1.0 / 128)
    scale = np.abs(rhs64).sum() * 2.0  # the row sums cancel heavily: compare on the scale of the summands
    assert np.max(np.abs(got[rows] - want[rows])) < 16 * np.finfo(np.float32).eps * scale


def _sampled_rows_vs_oracle(oracle, prob, kernel, X, rows, seed=0, with_abs=False):
    """one implicit matvec of a resident problem against the float64 oracle on `rows` (row-owned sums, svm_kernel.cpp:33-54)"""
    N, d = X.shape
    n = N - 1
    dt = X.dtype
    rng = np.random.default_rng(seed)
    rhs = rng.uniform(-1, 1, size=n).astype(dt)
    q, QA = prob.q()
    got = prob.matvec(rhs, np.zeros(n, dt), 1.0)
    X64, q64, rhs64 = X.astype(np.float64), q.astype(np.float64), rhs.astype(np.float64)
    want = np.zeros(n)
    for r in rows:
        want = oracle.matvec_rows(kernel, X64, q64, rhs64, want, float(QA), 1.0, 1.0, int(r), int(r) + 1, degree=3, gamma=1.0 / d, coef0=0.0)
    # the row sums cancel heavily: compare every sampled row on the scale of ITS summands, sum_j |Abar_ij| |d_j| (numpy, float64)
    G = X64[rows] @ X64[:n].T
    if kernel == "rbf":
        sq = np.einsum("ij,ij->i", X64, X64)
        K = np.exp(-(1.0 / d) * np.maximum(sq[rows, None] + sq[None, :n] - 2.0 * G, 0.0))
    elif kernel == "polynomial":
        K = (G / d) ** 3
    else:
        K = G
    absd = np.abs(rhs64)
    scale = np.abs(K) @ absd + (abs(float(QA)) + np.abs(q64[rows])) * absd.sum() + np.abs(q64) @ absd + absd[rows]
    err = float(np.max(np.abs(got[rows] - want[rows]) / scale))
    if with_abs:  # also the largest absolute deviation of a sampled row
        return err, got, rhs, float(np.max(np.abs(got[rows] - want[rows])))
    return err, got, rhs


@pytest.mark.parametrize("cfg, kernel, dt, N, d", [("configs[2]", "linear", np.float32, 200_000, 256), ("configs[3]", "polynomial", np.float64, 100_000, 64),
                                                   ("configs[4]", "rbf", np.float32, 1_000_000, 128)])
def test_baseline_configs_at_full_size(oracle, cfg, kernel, dt, N, d):
    """BASELINE.json configs[2..4] at their FULL sizes (configs[1] is test_sub_sampled_rows_of_a_large_matvec_vs_oracle): 64 seeded rows
    of one implicit matvec against the float64 oracle, plus the size-independent properties of the operator (linearity, symmetry).
    The full CPU product would take hours (O(n^2 d)); a sampled row costs n * d multiply-adds."""
    X, _ = make_blobs_pm1(N, d, seed=42, dtype=dt)
    p = Parameter(kernel_type=kernel, degree=3)
    n = N - 1
    eps = np.finfo(dt).eps
    rows = np.sort(np.random.default_rng(3).choice(n, size=64, replace=False))
    rows[0], rows[-1] = 0, n - 1  # first row (longest mirrored column) and last row (longest row of the triangle)
    with backend.ResidentProblem(p, X) as prob:
        assert prob.info()["symmetric"] == 1 and prob.info()["gram_mode"] == (2 if dt == np.float32 else 0)  # fp32: f16x3 (the data passes the check)
        err, Au, u, abs_err = _sampled_rows_vs_oracle(oracle, prob, kernel, X, rows, with_abs=True)
        # on the scale of a row's SUMMANDS (the rank-1 terms are 1e3 x the result at configs[4]): measured 0.01 ... 0.03 eps (profiles/r04_scale_check_*.log); the bar
        # was 16 eps until round 4 -- and on the scale of the RESULT itself
        assert err < 1 * eps, (cfg, err / eps)
        assert abs_err <= (4e-6 if dt == np.float32 else 4e-14) * np.max(np.abs(Au)), (cfg, abs_err / np.max(np.abs(Au)))
        v = np.random.default_rng(1).uniform(-1, 1, size=n).astype(dt)
        z = np.zeros(n, dt)
        Av = prob.matvec(v, z)
        Aw = prob.matvec((dt(2) * u - dt(3) * v).astype(dt), z)
    scale = np.max(np.abs(Au)) + np.max(np.abs(Av))
    assert np.max(np.abs(Aw - (2 * Au.astype(np.float64) - 3 * Av.astype(np.float64)))) < 2e3 * eps * scale
    uAv, vAu = float(u.astype(np.float64) @ Av.astype(np.float64)), float(v.astype(np.float64) @ Au.astype(np.float64))
    assert abs(uAv - vAu) < 2e3 * eps * (np.abs(u) @ np.abs(Av) + np.abs(v) @ np.abs(Au))


def test_baseline_config4_row_sharded_over_eight_shards(oracle):
    """BASELINE.json configs[4] as it is meant to run: 1 000 000 x 128 rbf fp32, row-block sharded 8 ways with one exchange of the partial
    K*d vectors per matvec -- eight shards behind ONE call; on a one-GPU box they share the device and exchange through the peer kernels,
    on an eight-GPU box they take devices 0..7 and RCCL."""
    N, d = 1_000_000, 128
    X, _ = make_blobs_pm1(N, d, seed=42, dtype=np.float32)
    devices = list(range(8)) if _capi.device_count() >= 8 else [0] * 8
    rows = np.sort(np.random.default_rng(4).choice(N - 1, size=32, replace=False))
    with backend.ResidentProblem(Parameter(kernel_type="rbf"), X, devices=devices) as prob:
        info = prob.info()
        assert info["devices_used"] == 8 and info["local_devices"] == 8 and info["symmetric"] == 1
        err, Au, _, abs_err = _sampled_rows_vs_oracle(oracle, prob, "rbf", X, rows, with_abs=True)
    assert err < 1 * np.finfo(np.float32).eps, err / np.finfo(np.float32).eps  # (16 eps until round 4; measured 0.01 eps)
    assert abs_err <= 4e-6 * np.max(np.abs(Au)), abs_err / np.max(np.abs(Au))


def test_baseline_config0_500x4_linear_fp64_through_the_libsvm_files(tmp_path):
    """BASELINE.json configs[0]: "generate_data.py 500x4, linear kernel, fp64" -- the plumbing case, end to end through the file formats:
    the data set is written as a LIBSVM file, read back (bit-equal to the fixture an independent parser produced), trained with the
    reference's defaults (eps = 1e-3) and at eps = 1e-10, and compared with what the reference's own OpenMP kernels computed on the same
    file (tests/golden/c1_500x4.npz, generator tests/golden/make_c1_fixture.py); the model file round trip keeps alpha and rho."""
    from conftest import HERE
    from plssvm_amd.csvm import make_csvm
    from plssvm_amd.data_set import DataSet
    from plssvm_amd.model import Model

    fx = np.load(os.path.join(HERE, "golden", "c1_500x4.npz"))
    f = tmp_path / "500x4.libsvm"
    f.write_bytes(fx["libsvm_text"].tobytes())
    ds = DataSet(filename=str(f), real_type=np.float64)
    assert ds.num_data_points() == 500 and ds.num_features() == 4
    assert np.array_equal(ds.data(), fx["X"]) and np.array_equal(np.asarray(ds.mapped_labels(), np.float64), fx["y"])
    svm = make_csvm("mi355", params=Parameter(kernel_type="linear"))
    for tag in ("default", "tight"):
        model = svm.fit(ds, epsilon=float(fx[f"{tag}/eps"]))
        info = svm.last_cg_info
        assert int(info["iterations"]) == int(fx[f"{tag}/iterations"])
        # the first N-1 entries are the CG iterate; alpha_N = -sum(alpha) and rho live in the direction of the all-ones vector, which the
        # converged solve determines worst: the reference's own 256-thread run on the GPU box leaves its 1-thread golden by 9e-8 there
        # (and by 2e-13 at the default eps) -- measured in round 2 with a throw-away script; ours: 8.9e-8 / 2.9e-12
        g = fx[f"{tag}/alpha"]
        tol, tol_sum = (1e-11, 1e-10) if tag == "default" else (1e-8, 1e-6)
        assert np.max(np.abs(model.alpha[:-1] - g[:-1])) < tol * np.max(np.abs(g))
        assert abs(model.alpha[-1] - g[-1]) < tol_sum and abs(model.alpha.sum()) < 1e-12
        assert abs(float(model.rho) - float(fx[f"{tag}/rho"])) < tol_sum
        if tag == "default":
            assert abs(float(info["residuum"]) - float(fx[f"{tag}/delta"])) <= 1e-6 * float(fx[f"{tag}/delta"])
        else:  # at delta / delta0 ~ 1e-22 the last residual is rounding noise: it only has to meet the stop criterion, as the reference's does
            assert float(info["residuum"]) <= float(info["target_residuum"]) and float(fx[f"{tag}/delta"]) <= float(info["target_residuum"])
    mf = tmp_path / "500x4.model"
    model.save(mf)
    m2 = Model.load(mf, real_type=np.float64)
    assert abs(float(m2.rho) - float(model.rho)) < 1e-9 * max(1.0, abs(float(model.rho)))
    assert svm.score(m2, ds) == svm.score(model, ds)


@pytest.mark.parametrize("kernel, dt, N, d", [("rbf", np.float32, 50_000, 128), ("polynomial", np.float64, 20_000, 64), ("linear", np.float32, 30_000, 256)])
def test_linearity_and_symmetry_at_scale(kernel, dt, N, d):
    """Size-independent properties of the implicit operator at BASELINE-like sizes: A(a u + b v) = a A u + b A v and
    u^T (A v) = v^T (A u) (the reduced matrix is symmetric)."""
    X, _ = make_blobs_pm1(N, d, seed=42, dtype=dt)
    p = Parameter(kernel_type=kernel)
    rng = np.random.default_rng(1)
    u = rng.uniform(-1, 1, size=N - 1).astype(dt)
    v = rng.uniform(-1, 1, size=N - 1).astype(dt)
    z = np.zeros(N - 1, dt)
    with backend.ResidentProblem(p, X) as prob:
        Au, Av = prob.matvec(u, z), prob.matvec(v, z)
        Aw = prob.matvec((dt(2) * u - dt(3) * v).astype(dt), z)
    eps = np.finfo(dt).eps
    scale = np.max(np.abs(Au)) + np.max(np.abs(Av))
    assert np.max(np.abs(Aw - (2 * Au.astype(np.float64) - 3 * Av.astype(np.float64)))) < 2e3 * eps * scale
    uAv, vAu = float(u.astype(np.float64) @ Av.astype(np.float64)), float(v.astype(np.float64) @ Au.astype(np.float64))
    assert abs(uAv - vAu) < 2e3 * eps * (np.abs(u) @ np.abs(Av) + np.abs(v) @ np.abs(Au))


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_predict_and_score_on_the_reference_fixture(kernel, dt):
    """GenericCSVM.predict / score (generic_csvm_tests.hpp:197-247) on the reference's LIBSVM-trained 500x200 models:
    predicted labels equal tests/data/predict/500x200.libsvm.predict exactly, score == 1.0."""
    import os
    from conftest import HERE
    fx = np.load(os.path.join(HERE, "golden", "predict_500x200.npz"))
    gamma = float(fx[f"{kernel}_gamma"])
    p = Parameter(kernel_type=kernel, degree=int(fx[f"{kernel}_degree"]), gamma=(gamma if gamma == gamma else None), coef0=float(fx[f"{kernel}_coef0"]))
    out, w = backend.predict_values(p, fx[f"{kernel}_sv"].astype(dt), fx[f"{kernel}_alpha"].astype(dt), float(fx[f"{kernel}_rho"]), None, fx["test_X"].astype(dt))
    labels = np.where(out > 0, 1, -1)
    assert np.array_equal(labels, fx["expected"])
    assert float(np.mean(labels == fx["test_y"].astype(int))) == 1.0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("N, d", [(300, 7), (1500, 40), (4097, 128), (2300, 200), (1100, 300), (900, 512), (700, 600)])
def test_symmetric_and_full_square_variants_agree(oracle, kernel, N, d, dtype):
    """The default path evaluates only the tiles on/below the diagonal and mirrors them (as the reference does); the
    full-square variant (option symmetric=0) sums every row independently.  Both must match the oracle and each other.
    (Both types are symmetric at any width: the linear kernel by passes over feature panels, rbf / polynomial by panels inside a tile.)"""
    X, y = make_blobs_pm1(N, d, seed=21, dtype=dtype)
    p = Parameter(kernel_type=kernel)
    rhs = np.random.default_rng(5).uniform(-1, 1, size=N - 1).astype(dtype)
    out = {}
    for sym in (1, 0):
        _capi.set_option("symmetric", sym)
        try:
            with backend.ResidentProblem(p, X) as prob:
                q, QA = prob.q()
                out[sym] = prob.matvec(rhs, np.zeros(N - 1, dtype), 1.0)
                prob.cg_begin(y, 1e-30)
                prob.cg_step(5)
                out[("a", sym)] = prob.cg_finish()[0]
                out[("sym", sym)] = prob.info()["symmetric"]
        finally:
            _capi.set_option("symmetric", 1)
    assert out[("sym", 0)] == 0
    assert out[("sym", 1)] == 1
    kw = dict(degree=3, gamma=1.0 / d, coef0=0.0)
    want = oracle.matvec(kernel, X, q, rhs, np.zeros(N - 1, dtype), QA, 1.0, 1.0, **kw)
    scale = np.max(np.abs(want))
    eps = np.finfo(dtype).eps
    # yardstick: the distance of the oracle (run in `dtype`) from the same product in float64 with the same q -- a few thousand
    # terms with cancellation make every fp32 evaluation order deviate by more than a fixed multiple of eps * max|result|
    if dtype == np.float32:
        truth = oracle.matvec(kernel, X.astype(np.float64), q.astype(np.float64), rhs.astype(np.float64), np.zeros(N - 1), float(QA), 1.0, 1.0, **kw)
        tol = max(4.0 * np.max(np.abs(want - truth)), 64 * eps * scale)
    else:
        truth = want
        tol = 64 * eps * scale
    assert np.max(np.abs(out[1] - truth)) < tol and np.max(np.abs(out[0] - truth)) < tol
    assert np.max(np.abs(out[1] - out[0])) < 2 * tol
    assert np.all(np.isfinite(out[("a", 1)]))


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("sym", [1, 0])
@pytest.mark.parametrize("kernel, dtype, N, d", [("rbf", np.float32, 3000, 128), ("linear", np.float32, 2100, 200), ("polynomial", np.float64, 2500, 64),
                                                 ("rbf", np.float64, 1300, 40), ("polynomial", np.float32, 700, 9)])
def test_every_ranks_share_on_one_gpu(kernel, dtype, N, d, sym, world):
    """The sharded path (row blocks of rank r of `world`, offsets into the work-item list, the packed column slab, zero-initialised
    partial vectors) evaluated rank by rank on ONE GPU with the exchange switched off (option skip_collective): the partial
    products must add up to the unsharded product.  Every rank adds the rank-1 terms of A-bar itself, so the sum holds them
    `world` times."""
    X, _ = make_blobs_pm1(N, d, seed=33, dtype=dtype)
    p = Parameter(kernel_type=kernel, cost=2.0)
    n = N - 1
    v = np.random.default_rng(9).uniform(-1, 1, size=n).astype(dtype)
    zero = np.zeros(n, dtype)
    _capi.set_option("symmetric", sym)
    try:
        with backend.ResidentProblem(p, X) as prob:
            q, QA = prob.q()
            single = prob.matvec(v, zero, 1.0)
        _capi.set_option("skip_collective", 1)
        total = np.zeros(n, np.float64)
        shares = []
        for rank in range(world):
            with backend.ResidentProblem(p, X, rank=rank, world=world) as prob:
                part = prob.matvec(v, zero, 1.0)
                shares.append(float(np.max(np.abs(part))))
                total += part
    finally:
        _capi.set_option("skip_collective", 0)
        _capi.set_option("symmetric", 1)
    q64, v64 = q.astype(np.float64), v.astype(np.float64)
    S = v64.sum()
    rank1 = v64 / 2.0 + (float(QA) * S - q64 @ v64) - S * q64  # v / C + (QA_cost * S - q.v) * 1 - S * q
    got = total - (world - 1) * rank1
    scale = np.max(np.abs(single)) + np.max(np.abs(rank1))
    assert np.max(np.abs(got - single)) < 256 * np.finfo(dtype).eps * scale
    assert all(s > 0 for s in shares[: min(world, (n + 127) // 128)])


@pytest.mark.parametrize("kernel, dtype, N, d, weights", [("rbf", np.float32, 9100, 128, [3.0, 1.0, 2.0]), ("polynomial", np.float64, 2500, 64, [1.0, 1.0, 0.25, 4.0]),
                                                          ("linear", np.float32, 3000, 200, [0.97, 1.0, 1.03, 1.0, 0.9, 1.1, 1.0, 1.0])])
def test_weighted_shares_add_up_to_the_unsharded_product(kernel, dtype, N, d, weights):
    """Round 5: devices of unequal pace get unequal shares of the triangle (lssvm_mi355_set_shard_weights: rank r evaluates weights[r] / sum of its area; the boundaries stay
    even block indices).  Every rank's share evaluated in turn on ONE GPU with the exchange switched off: the partial products add up to the unsharded product
    (every rank adds the rank-1 terms itself), the shares' sizes follow the weights, and the solve over all shards of one process -- peer exchange on one device --
    gives the unsharded product (and in fp64 the unsharded solve's alpha)."""
    from plssvm_amd import sharding

    X, y = make_blobs_pm1(N, d, seed=33, dtype=dtype)
    p = Parameter(kernel_type=kernel, cost=2.0)
    n, world = N - 1, len(weights)
    v = np.random.default_rng(9).uniform(-1, 1, size=n).astype(dtype)
    zero = np.zeros(n, dtype)
    with backend.ResidentProblem(p, X) as prob:
        q, QA = prob.q()
        single = prob.matvec(v, zero, 1.0)
    a1, rho1, _ = backend.solve_system_of_linear_equations(p, X, y, 1e-30, 6)
    _capi.set_shard_weights(weights)
    try:
        parts = sharding.sym_block_partition(n, world, weights)
        assert [_capi.shard_blocks(N, world, r, True) for r in range(world)] == parts and parts[-1][1] == (n + 127) // 128
        _capi.set_option("skip_collective", 1)
        total = np.zeros(n, np.float64)
        for rank in range(world):
            with backend.ResidentProblem(p, X, rank=rank, world=world) as prob:
                total += prob.matvec(v, zero, 1.0)
        _capi.set_option("skip_collective", 0)
        with backend.ResidentProblem(p, X, devices=[0] * world) as prob:
            exchanged = prob.matvec(v, zero, 1.0)
            prob.cg_begin(y, 1e-30)
            prob.cg_step(6)
            a2, rho2, _ = prob.cg_finish()
    finally:
        _capi.set_option("skip_collective", 0)
        _capi.set_shard_weights(None)
    q64, v64 = q.astype(np.float64), v.astype(np.float64)
    S = v64.sum()
    rank1 = v64 / 2.0 + (float(QA) * S - q64 @ v64) - S * q64
    got = total - (world - 1) * rank1
    eps = np.finfo(dtype).eps
    assert np.max(np.abs(got - single)) < 256 * eps * (np.max(np.abs(single)) + np.max(np.abs(rank1)))
    areas = [sharding.triangle_share(n, world, r, weights) for r in range(world)]
    want = [w / sum(weights) * n * (n + 1) / 2 for w in weights]
    assert all(abs(a - w) <= 2.5 * 256 * (n + 256) for a, w in zip(areas, want)), (areas, want)  # (within a pair of row blocks of the ideal boundary)
    assert np.max(np.abs(exchanged - single)) < 256 * eps * (np.max(np.abs(single)) + np.max(np.abs(rank1)))  # all shards behind one call, partial vectors exchanged
    if dtype == np.float64:  # (six fp32 CG iterations amplify the re-association beyond any useful bar: DESIGN.md section 5)
        assert ol.rel_inf(a2, a1) < 1e-9 and abs(float(rho2) - float(rho1)) <= 1e-9 * max(1.0, abs(float(rho1)))
    else:
        assert np.all(np.isfinite(a2)) and np.isfinite(float(rho2))


@pytest.mark.parametrize("kernel, dtype, N, d, shards", [("polynomial", np.float64, 2500, 64, 3), ("rbf", np.float32, 16_500, 128, 4), ("linear", np.float64, 3000, 40, 8)])
def test_a_live_problem_takes_new_shares_between_two_cg_steps(kernel, dtype, N, d, shards):
    """lssvm_mi355_problem_rebalance (round 5): the shards of a sharded symmetric problem are rebuilt for new shares of the triangle while the solve is under way -- the
    data, the operand planes, the vectors and the CG state stay, only who evaluates which tiles changes.  All shards behind one call on one device: explicit weights
    (changed = True; the product of a scratch vector before and after agrees to the association; in fp64 the solve ends where the unsharded solve does), shares by
    measured pace (whatever the time sharing of one device makes of it: the solve must go on either way), and a single-device problem (nothing to do)."""
    X, y = make_blobs_pm1(N, d, seed=12, dtype=dtype)
    p = Parameter(kernel_type=kernel, cost=2.0)
    n = N - 1
    eps = np.finfo(dtype).eps
    v = np.random.default_rng(4).uniform(-1, 1, size=n).astype(dtype)
    zero = np.zeros(n, dtype)
    a1, rho1, i1 = backend.solve_system_of_linear_equations(p, X, y, 1e-10, 4000)  # to convergence: unconverged iterates of two associations differ by what CG amplifies
    with backend.ResidentProblem(p, X) as single:
        want = single.matvec(v, zero, 1.0)
        assert single.rebalance() is False and single.rebalance([1.0]) is False
    with backend.ResidentProblem(p, X, devices=[0] * shards) as prob:
        before = prob.matvec(v, zero, 1.0)
        prob.cg_begin(y, 1e-10)
        prob.cg_step(3)
        assert prob.rebalance([1.0 + (r % 3) for r in range(shards)]) is True
        mid = prob.matvec(v, zero, 1.0)
        prob.cg_step(3)
        measured = prob.rebalance()  # shares by measured pace
        after = prob.matvec(v, zero, 1.0)
        prob.cg_step(4000)
        a2, rho2, info = prob.cg_finish()
        with pytest.raises(InvalidParameterError):
            prob.rebalance([1.0] * (shards + 1))
    assert info["converged"] == 1 and i1["converged"] == 1 and isinstance(measured, bool)
    scale = 256 * eps * np.max(np.abs(want))
    assert np.max(np.abs(before - want)) < scale and np.max(np.abs(mid - want)) < scale and np.max(np.abs(after - want)) < scale
    if dtype == np.float64:
        # both solves converged (eps 1e-10); alpha_N and rho lie in the direction of the all-ones vector, which a converged solve determines worst (4e-7 / 2e-6 measured)
        assert ol.rel_inf(a2, a1) < 1e-5 and abs(float(rho2) - float(rho1)) <= 1e-4 * max(1.0, abs(float(rho1)))
    else:
        assert np.all(np.isfinite(a2))  # (two fp32 CG runs of different association do not reproduce each other: DESIGN.md section 5; the products above are the check)


def test_one_shot_solve_on_several_shards_rebalances_when_asked():
    """Option rebalance_after (round 5; default 0 = never): lssvm_mi355_solve_multi_* runs that many iterations, gives the shards new shares by their measured pace
    (lssvm_mi355_problem_rebalance) and finishes the solve -- the call a plssvm::csvm adaptor makes.  Four shards on one device: whatever the time sharing measures, the
    converged solve is the single-device solve's within the CG-level fp64 bar; with more iterations asked for than the solve has, nothing happens."""
    X, y = make_blobs_pm1(2600, 48, seed=19, dtype=np.float64)
    p = Parameter(kernel_type="polynomial", degree=2, cost=2.0)
    a1, rho1, i1 = backend.solve_system_of_linear_equations(p, X, y, 1e-10, 4000)
    for after in (3, 10**6):
        _capi.set_option("rebalance_after", after)
        try:
            a2, rho2, i2 = backend.solve_system_of_linear_equations(p, X, y, 1e-10, 4000, devices=[0] * 4)
        finally:
            _capi.set_option("rebalance_after", 0)
        assert i1["converged"] == 1 and i2["converged"] == 1 and i2["devices_used"] == 4
        assert ol.rel_inf(a2, a1) < 1e-5 and abs(float(rho2) - float(rho1)) <= 1e-4 * max(1.0, abs(float(rho1)))


@pytest.mark.parametrize("kernel, dtype", [("rbf", np.float32), ("polynomial", np.float64), ("linear", np.float32)])
def test_row_block_bands_of_the_column_slab(oracle, kernel, dtype):
    """The column-sum records of the symmetric variant are produced band by band into one slab (option colslab_band_mb; 1M points in fp32
    would otherwise need 15.6 GB of records): with 1 MiB bands a 20 000-point problem runs its tile kernel 5-7 times per matvec, and the
    result must equal the single-band result within the kernel-level bar (the bands only re-associate the fixed-order sums), also sharded."""
    N, d = 20_000, 24
    X, y = make_blobs_pm1(N, d, seed=17, dtype=dtype)
    p = Parameter(kernel_type=kernel)
    v = np.random.default_rng(2).uniform(-1, 1, size=N - 1).astype(dtype)
    zero = np.zeros(N - 1, dtype)
    with backend.ResidentProblem(p, X) as prob:
        assert prob.info()["tile_launches_per_matvec"] == 1
        one = prob.matvec(v, zero, 1.0)
    _capi.set_option("colslab_band_mb", 1)
    with backend.ResidentProblem(p, X) as prob:
        bands = prob.info()["tile_launches_per_matvec"]
        many = prob.matvec(v, zero, 1.0)
        err, _, _ = _sampled_rows_vs_oracle(oracle, prob, kernel, X, np.array([0, 1, 127, 128, 5000, 12345, N - 2]))
        assert err < 16 * np.finfo(dtype).eps
        prob.cg_begin(y, 1e-30)
        prob.cg_step(2)
        assert prob.info()["matvec_launches"] == 3  # a matvec counts once however many bands it takes
    with backend.ResidentProblem(p, X, devices=[0, 0, 0]) as prob:
        sharded = prob.matvec(v, zero, 1.0)
    rec_mib = (157 * 156 / 2) * 128 * np.dtype(dtype).itemsize / 2**20
    assert bands == int(np.ceil(rec_mib)) and bands > 1
    scale = np.max(np.abs(one))
    eps = np.finfo(dtype).eps
    assert np.max(np.abs(many - one)) < 64 * eps * scale and np.max(np.abs(sharded - one)) < 64 * eps * scale


def test_column_slab_budget_falls_back_to_the_full_square():
    """Option colslab_limit_mb = 0 switches the symmetric variant off: the solver evaluates the full square (row-owned sums)."""
    X, y = make_blobs_pm1(3000, 16, seed=4, dtype=np.float32)
    p = Parameter(kernel_type="rbf")
    with backend.ResidentProblem(p, X) as prob:
        assert prob.info()["symmetric"] == 1
        ref = prob.matvec(np.ones(2999, np.float32), np.zeros(2999, np.float32), 1.0)
    _capi.set_option("colslab_limit_mb", 0)
    try:
        with backend.ResidentProblem(p, X) as prob:
            assert prob.info()["symmetric"] == 0
            got = prob.matvec(np.ones(2999, np.float32), np.zeros(2999, np.float32), 1.0)
    finally:
        _capi.set_option("colslab_limit_mb", 98304)
    assert ol.rel_inf(got, ref) < 64 * np.finfo(np.float32).eps


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("N, d", [(300, 7), (1500, 64), (4097, 128), (2300, 200), (1100, 256), (1300, 320), (900, 384), (700, 448), (1000, 512)])
@pytest.mark.parametrize("sym", [1, 0])
def test_bf16_split_gram_mode_is_fp32_accurate(oracle, kernel, N, d, sym):
    """The split Gram modes against the native one.  gram_mode = 2 ("f16x3", the default where the data passes the representability check,
    up to 512 features in one pass): two f16 planes of the pre-scaled operands, three plane products on the f16 matrix cores; gram_mode = 1
    ("bf16x6", up to 384 features in one pass): exact three-way bf16 split, six plane products; beyond the one-pass limits the symmetric
    variant walks feature panels inside a tile (rbf / polynomial); gram_mode = 0: native v_mfma_f32 chains.  All must meet the same
    bar: no farther from the float64 product than 4x the fp32 CPU oracle (or 64 eps)."""
    X, y = make_blobs_pm1(N, d, seed=27, dtype=np.float32)
    p = Parameter(kernel_type=kernel)
    rhs = np.random.default_rng(6).uniform(-1, 1, size=N - 1).astype(np.float32)
    zero = np.zeros(N - 1, np.float32)
    out = {}
    _capi.set_option("symmetric", sym)
    try:
        for mode in (0, 1, 2, 3):
            _capi.set_option("gram_mode", mode)
            with backend.ResidentProblem(p, X) as prob:
                q, QA = prob.q()
                f16_limit = 384 if kernel == "rbf" else 512  # rbf holds three row planes in registers (the shifted planes)
                # beyond the one-pass kernels: feature panels inside a tile (either variant, either plane kind) -- the linear kernel too where it runs as the
                # polynomial kernel of degree 1 (more than 256 features, fewer than 10 000 points: every shape of this test with d > 256)
                panels = kernel != "linear" or d > 256
                assert prob.info()["gram_mode"] == {0: 0, 1: 1 if d <= 384 or panels else 0, 2: 2 if d <= f16_limit or panels else 0, 3: 2 if d <= f16_limit or panels else 0}[mode]
                out[mode] = prob.matvec(rhs, zero, 1.0)
                prob.cg_begin(y, 1e-30)
                prob.cg_step(5)
                out[("a", mode)] = prob.cg_finish()[0]
    finally:
        _capi.set_option("gram_mode", 3)  # the library default
        _capi.set_option("symmetric", 1)
    kw = dict(degree=3, gamma=1.0 / d, coef0=0.0)
    want = oracle.matvec(kernel, X, q, rhs, zero, QA, 1.0, 1.0, **kw)
    truth = oracle.matvec(kernel, X.astype(np.float64), q.astype(np.float64), rhs.astype(np.float64), np.zeros(N - 1), float(QA), 1.0, 1.0, **kw)
    scale = np.max(np.abs(truth))
    tol = max(4.0 * np.max(np.abs(want - truth)), 64 * np.finfo(np.float32).eps * scale)
    errs = {mode: float(np.max(np.abs(out[mode] - truth))) for mode in (0, 1, 2)}
    assert all(e < tol for e in errs.values()), (errs, tol)
    assert np.array_equal(out[3], out[2])  # this data passes the check: the default mode IS f16x3
    assert np.all(np.isfinite(out[("a", 1)])) and np.all(np.isfinite(out[("a", 2)]))


def _f16x2_row_statistic(X, shift):
    """k_split_f16x2 (tile_launch_f32h.hip) in numpy for the pre-scaled planes of the linear / polynomial kernels: max over the rows of
    |y - hi - mid|_2 / |y|_2 with y = 2^shift x, hi = f16(y), mid = f16(y - hi) (numpy's float16 conversion rounds to nearest even like v_cvt_f16_f32)."""
    y = X.astype(np.float32) * np.float32(2.0 ** shift)
    hi = y.astype(np.float16).astype(np.float32)
    r1 = y - hi
    r2 = r1 - r1.astype(np.float16).astype(np.float32)
    sr, sx = np.sum(r2.astype(np.float64) ** 2, axis=1), np.sum(y.astype(np.float64) ** 2, axis=1)
    return float(np.sqrt(np.max(sr[sx > 0] / sx[sx > 0])))


def _data_at_the_f16_acceptance_edge(N, d, log2_target, seed):
    """[-1, 1] data whose largest entry fixes the pre-scale (2^14 <= 2^k max|x| < 2^15) plus 64 rows of SMALL entries, s * (+-[0.5, 1]): their mid
    plane is subnormal in f16 (absolute error ~2^-25 in the scaled domain), so their relative representation error grows as s shrinks -- s is
    searched on a fine grid for the statistic closest to 2^log2_target."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1.0, 1.0, size=(N, d)).astype(np.float32)
    X[0, 0] = 1.0
    shift = 14  # ilogb(1.0) = 0
    small = np.arange(N) % (N // 64) == 1
    U = (rng.uniform(0.5, 1.0, size=(int(small.sum()), d)) * rng.choice([-1.0, 1.0], size=(int(small.sum()), d))).astype(np.float32)
    best = None
    for e in np.arange(-21.0, -14.0, 1.0 / 64):
        X[small] = (U * np.float32(2.0 ** e)).astype(np.float32)
        st = np.log2(_f16x2_row_statistic(X, shift))
        if best is None or abs(st - log2_target) < abs(best[0] - log2_target):
            best = (st, e)
    X[small] = (U * np.float32(2.0 ** best[1])).astype(np.float32)
    return X, small, np.log2(_f16x2_row_statistic(X, shift))


@pytest.mark.parametrize("sym", [1, 0])
@pytest.mark.parametrize("kernel", ["linear", "polynomial"])
@pytest.mark.parametrize("N, d", [(1500, 96), (9000, 128)])
def test_f16x3_at_its_acceptance_edge(kernel, N, d, sym):
    """make_planes accepts two f16 planes while the largest relative representation error of a row stays <= 2^-22
    (F16_REL2_MAX, lssvm_problem.hip), and every test so far sat at 2^-24.5.  Data constructed to land at 2^-22.1 (must be accepted and run f16x3)
    and at 2^-21.9 (must be rejected and run bf16x6): on BOTH sides every row of A v stays within 16 eps of the float64 product on the scale of that
    row's own summands -- the rows made of the small entries included, whose right-hand side is 1 on the small points only, so that they sum
    small x small products and show what their planes are worth -- symmetric (9 000 points: the 256-row workgroups) and full-square variant.
    Just beyond the edge the forced f16x3 kernel (gram_mode = 2) is still inside the bar: the threshold errs on the safe side.  (Round 6: the LINEAR
    kernel answers the rejection with a scale per row and stays on f16x3 -- same bar.)"""
    eps = np.finfo(np.float32).eps
    for log2_target, want_mode in ((-22.1, 2), (-21.9, 1)):
        X, small, got_stat = _data_at_the_f16_acceptance_edge(N, d, log2_target, seed=17)
        assert abs(got_stat - log2_target) < 0.05, got_stat
        rhs = np.where(small[:N - 1], 1.0, 0.0).astype(np.float32) * np.random.default_rng(3).uniform(0.5, 1.0, N - 1).astype(np.float32)
        gamma = 1.0 / d
        p = Parameter(kernel_type=kernel, degree=3, gamma=gamma, coef0=0.0)
        X64, v64 = X.astype(np.float64)[:N - 1], rhs.astype(np.float64)
        G = X64 @ X64.T
        K = G if kernel == "linear" else (gamma * G) ** 3
        for mode in ((3, 2) if want_mode == 1 else (3,)):
            _capi.set_option("symmetric", sym)
            _capi.set_option("gram_mode", mode)
            with backend.ResidentProblem(p, X) as prob:
                used = prob.info()["gram_mode"]
                row_error = prob.info()["f16_row_rel_error"]
                q, QA = prob.q()
                got = prob.matvec(rhs, np.zeros(N - 1, np.float32), 1.0).astype(np.float64)
            if kernel == "linear" and mode == 3 and want_mode == 1:
                # round 6: rejected with ONE scale for the matrix, the linear kernel's planes get a power-of-two scale per row (K = D Xs Xs^T D) and stay f16x3;
                # the reported error is that of the planes that run
                assert used == 2 and row_error <= 2.0 ** -22, (log2_target, mode, used, row_error)
            else:
                assert used == (want_mode if mode == 3 else 2), (log2_target, mode, used)
            q64 = q.astype(np.float64)
            S, qv = float(v64.sum()), float(q64 @ v64)
            truth = K @ v64 + v64 + (float(QA) * S - qv) - S * q64
            scale = np.abs(K) @ np.abs(v64) + np.abs(v64) + abs(float(QA) * S) + abs(qv) + np.abs(S * q64)
            err = np.abs(got - truth) / scale / eps
            print(f"\n{kernel} {N} x {d} sym {sym}: statistic 2^{got_stat:.2f}, gram_mode {mode} -> ran {used}: worst row {err.max():.2f} eps (rows of the small points: {err[small[:N - 1]].max():.2f})")
            assert err.max() < 16.0, (log2_target, mode, float(err.max()))


@pytest.mark.parametrize("N, sym, mode", [(700, s, m) for s in (1, 0) for m in (3, 2, 1)] + [(10100, 1, 3), (10100, 0, 3), (10100, 1, 1)])
def test_linear_kernel_beyond_512_features_reports_the_kernel_that_ran(N, sym, mode):
    """From 10 000 points on: the linear kernel on more than 512 features runs the f16x3 panel passes in the SYMMETRIC variant only; the
    full-square variant (and bf16x6) stays on the generic native kernel -- and must then neither build planes nobody reads nor report a split Gram mode
    (bench.py prices `achieved` by it).  Below 10 000 points (round 4: the panel passes are launch-bound there) the tile kernels evaluate the linear
    kernel as the polynomial kernel of degree 1 -- one launch of the panels-inside-a-tile kernels, which exist for both variants and both plane kinds.
    Results against the float64 product either way."""
    d = 600
    X, _ = make_blobs_pm1(N, d, seed=31, dtype=np.float32)
    v = np.random.default_rng(2).uniform(-1, 1, N - 1).astype(np.float32)
    _capi.set_option("symmetric", sym)
    _capi.set_option("gram_mode", mode)
    with backend.ResidentProblem(Parameter(kernel_type="linear"), X) as prob:
        info = prob.info()
        q, QA = prob.q()
        got = prob.matvec(v, np.zeros(N - 1, np.float32), 1.0).astype(np.float64)
    if N >= 10000:
        panels = sym == 1 and mode >= 2  # (the panel passes are f16x3 kernels of the symmetric variant; everything else runs the generic full-square kernel)
        assert info["symmetric"] == (1 if panels else 0)
        assert info["gram_mode"] == (2 if panels else 0)
    else:
        assert info["symmetric"] == sym
        assert info["gram_mode"] == (2 if mode >= 2 else 1)
    X64, v64 = X.astype(np.float64), v.astype(np.float64)
    K = X64[:-1] @ X64[:-1].T
    S = float(v64.sum())
    q64 = X64[:-1] @ X64[-1]
    truth = K @ v64 + v64 + ((float(X64[-1] @ X64[-1]) + 1.0) * S - float(q64 @ v64)) - S * q64
    scale = np.abs(K) @ np.abs(v64) + np.abs(v64) + abs(QA * S) + abs(float(q64 @ v64)) + np.abs(S * q64)
    assert np.max(np.abs(got - truth) / scale) < 16 * np.finfo(np.float32).eps


@pytest.mark.parametrize("kernel", KERNELS)
def test_f16_planes_prescale_check_and_fallback(oracle, kernel):
    """The default Gram mode (3) decides ON THE DATA whether two f16 planes represent it as well as fp32 does (make_planes in
    lssvm_problem.hip): (a) data of tiny magnitude is moved into f16's normal range by an exact power-of-two pre-scale (linear /
    polynomial: undone on the finished sums / inside gamma; rbf works on the exponent's own scale and accepts an absolute bound) and stays
    on f16x3; (b) data whose dynamic range exceeds what two f16 planes carry fails the check and runs as bf16x6, which is exact for every
    fp32 input -- and forcing f16x3 onto it (gram_mode = 2) is measurably worse, i.e. the check is what keeps the default accurate."""
    rng = np.random.default_rng(11)
    N, d = 1500, 96
    y = np.where(np.arange(N) % 2 == 0, 1.0, -1.0).astype(np.float32)
    rhs = rng.uniform(-1, 1, size=N - 1).astype(np.float32)
    zero = np.zeros(N - 1, np.float32)
    eps = np.finfo(np.float32).eps

    def run(X, mode, gamma):
        p = Parameter(kernel_type=kernel, degree=3, gamma=gamma, coef0=0.25)
        _capi.set_option("gram_mode", mode)
        with backend.ResidentProblem(p, X) as prob:
            q, QA = prob.q()
            got = prob.matvec(rhs, zero, 1.0).astype(np.float64)
            used = prob.info()["gram_mode"]
        # float64 evaluation of A v = K v + v / C + (QA S - q.v) 1 - S q with the kernel matrix formed explicitly; every row's error is
        # measured on the scale of THAT row's summands (a few huge entries must not hide the rows made of small ones)
        X64, v64, q64 = X.astype(np.float64)[:N - 1], rhs.astype(np.float64), q.astype(np.float64)
        G = X64 @ X64.T
        if kernel == "linear":
            K = G
        elif kernel == "polynomial":
            K = (gamma * G + 0.25) ** 3
        else:
            sq = np.einsum("ij,ij->i", X64, X64)
            K = np.exp(-gamma * np.maximum(sq[:, None] + sq[None, :] - 2.0 * G, 0.0))
        S, qv = float(v64.sum()), float(q64 @ v64)
        truth = K @ v64 + v64 + (float(QA) * S - qv) - S * q64
        scale = np.abs(K) @ np.abs(v64) + np.abs(v64) + abs(float(QA) * S) + abs(qv) + np.abs(S * q64)
        return used, float(np.max(np.abs(got - truth) / scale))

    # (a) entries ~1e-4: the mid plane would be subnormal without the pre-scale (rbf: without the shifted planes)
    Xa = (1e-4 * rng.standard_normal((N, d))).astype(np.float32)
    ga = 1.0 / (d * 1e-8)  # gamma of the same effect as 1/d on unit-variance data
    used, err = run(Xa, 3, ga)
    assert used == 2, used
    assert err < 8 * eps, err / eps
    if kernel != "rbf":
        # (b) two populations of points, entries ~1e+4 and ~1e-4 (eight decades apart): the pre-scale that keeps the large ones below f16's
        # overflow leaves the small ones without a usable mid plane.  The right-hand side is zero on the large points, so that the rows of
        # the small points sum small * small products only and show what their planes are worth.
        big = np.arange(N) % 2 == 0
        Xb = rng.standard_normal((N, d)).astype(np.float32)
        Xb[big] *= np.float32(1e4)
        Xb[~big] *= np.float32(1e-4)
        rhs[big[:N - 1]] = 0.0
        gb = 1.0 / (d * 1e8)
        used, err = run(Xb, 3, gb)
        # polynomial: bf16x6.  linear (round 6): the planes get a power-of-two scale PER ROW (K = D Xs Xs^T D) and stay f16x3 -- inside the same bar, where the
        # one scale forced below is not
        assert used == (2 if kernel == "linear" else 1), used
        assert err < 8 * eps, err / eps
        used_forced, err_forced = run(Xb, 2, gb)
        assert used_forced == 2
        if kernel == "linear":  # (the polynomial kernel's values are dominated by coef0 here and do not show the difference)
            assert err_forced > 100 * err, (err_forced / eps, err / eps)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("d", [64, 128, 192, 256, 320, 384, 448, 512])
@pytest.mark.parametrize("mode", [2, 1])
def test_split_kernels_steady_state_at_every_chunk_count(oracle, kernel, d, mode):
    """Work items of FIVE tiles, so that the branch-free steady-state tiles of the split kernels run at every number of 64-feature chunks
    (the automatic chunk length gives these small problems two tiles per item, which never leaves the checked tail -- round 3 found the
    LDS-DMA source of the 64-feature f16x3 form two tiles ahead wrong that way), in both variants, against the float64 oracle."""
    if mode == 1 and d > 384 or (mode == 2 and kernel == "rbf" and d > 384):
        pytest.skip("no such kernel: bf16x6 and the rbf form of f16x3 hold three row planes (up to 384 features)")
    N = 1700
    X, _ = make_blobs_pm1(N, d, seed=d, dtype=np.float32)
    p = Parameter(kernel_type=kernel)
    rhs = np.random.default_rng(d).uniform(-1, 1, size=N - 1).astype(np.float32)
    zero = np.zeros(N - 1, np.float32)
    kw = dict(degree=3, gamma=1.0 / d, coef0=0.0)
    _capi.set_option("gram_mode", mode)
    _capi.set_option("j_chunk_tiles", 5)
    out = {}
    for sym in (1, 0):
        _capi.set_option("symmetric", sym)
        with backend.ResidentProblem(p, X) as prob:
            assert prob.info()["gram_mode"] == mode and prob.info()["symmetric"] == sym
            q, QA = prob.q()
            out[sym] = prob.matvec(rhs, zero, 1.0)
    truth = oracle.matvec(kernel, X.astype(np.float64), q.astype(np.float64), rhs.astype(np.float64), np.zeros(N - 1), float(QA), 1.0, 1.0, **kw)
    want = oracle.matvec(kernel, X, q, rhs, zero, QA, 1.0, 1.0, **kw)
    scale = np.max(np.abs(truth))
    tol = max(4.0 * np.max(np.abs(want - truth)), 64 * np.finfo(np.float32).eps * scale)
    for sym in (1, 0):
        assert np.max(np.abs(out[sym] - truth)) < tol, (sym, float(np.max(np.abs(out[sym] - truth))), tol)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("kernel", KERNELS)
def test_wide_data_at_the_reference_test_shape(oracle, kernel, dt):
    """The reference's own kernel tests run on 5000 x 2000 data (tests/CMakeLists.txt:36-69, generic_csvm_tests.hpp:372-493).  2000 features
    are far beyond what the resident-row-panel kernels hold in registers (512 in fp32, 256 in fp64): the linear kernel runs one pass per
    feature panel (fp32: the f16x3 kernels over 128 features, fp64: 128), rbf / polynomial walk the panels inside a tile (fp32: 128
    features, fp64: 64).  q and one implicit matvec against the float64 oracle at 1000 x 2000, with the
    reference's test parameters, on the scale of each row's summands."""
    N, d = 1000, 2000
    rng = np.random.default_rng(2000)
    X = rng.uniform(-1, 1, size=(N, d)).astype(dt)
    P = dict(degree=2, gamma=0.001, coef0=1.0, cost=0.1)  # generic_csvm_tests.hpp:372-493
    p = prm(kernel, P)
    rhs = rng.uniform(-1, 1, size=N - 1).astype(dt)
    zero = np.zeros(N - 1, dt)
    kw = dict(degree=2, gamma=0.001, coef0=1.0)
    with backend.ResidentProblem(p, X) as prob:
        info = prob.info()
        if dt == np.float32:
            # fp32: the f16x3 kernels over feature panels -- linear: K = sum over the panels of X_p X_p^T, one pass per panel; rbf / polynomial:
            # the panels walked inside a tile (lssvm_tile_f32_wide.hip.hpp)
            assert info["gram_mode"] == 2 and info["symmetric"] == 1
        else:
            assert info["gram_mode"] == 0 and info["symmetric"] == 1   # fp64: linear over feature panels of 128, rbf / polynomial over panels of 64 inside a sub-tile
        q, QA = prob.q()
        got = prob.matvec(rhs, zero, 1.0).astype(np.float64)
    X64 = X.astype(np.float64)
    q64 = oracle.q(kernel, X64, **kw)
    eps = np.finfo(dt).eps
    assert ol.rel_inf(q, q64) < 64 * eps
    truth = oracle.matvec(kernel, X64, q64, rhs.astype(np.float64), np.zeros(N - 1), float(oracle.kernel_function(kernel, X64[-1], X64[-1], **kw)) + 1.0 / 0.1, 1.0 / 0.1, 1.0, **kw)  # (the kernels take 1 / cost, svm_kernel.cpp:27)
    G = X64[:N - 1] @ X64[:N - 1].T
    if kernel == "linear":
        K = G
    elif kernel == "polynomial":
        K = (0.001 * G + 1.0) ** 2
    else:
        sq = np.einsum("ij,ij->i", X64[:N - 1], X64[:N - 1])
        K = np.exp(-0.001 * np.maximum(sq[:, None] + sq[None, :] - 2.0 * G, 0.0))
    v64 = rhs.astype(np.float64)
    S = float(v64.sum())
    scale = np.abs(K) @ np.abs(v64) + np.abs(v64) / 0.1 + abs(float(QA) * S) + abs(float(q64 @ v64)) + np.abs(S * q64)
    # fp64: the yardstick is itself a float64 evaluation -- the oracle's (= the reference's) lower-triangle kernel adds its mirrored terms with
    # `omp atomic` in an order that changes from run to run, and two float64 evaluations of this product (the oracle's, numpy's K @ v) differ by
    # 14 eps of the rows' summands.  Against such a yardstick the bitwise reproducible device result measured 15.x ... 16.4 eps: at a bar of 16 this
    # test failed once in ~15 runs.  64 eps for float64 (the randomised cross-checks allow 256); float32 is held at 16 eps of float32, where the
    # yardstick's uncertainty is nine orders of magnitude below the bar.
    bar = 16 * eps if dt == np.float32 else 64 * eps
    assert np.max(np.abs(got - truth) / scale) < bar, float(np.max(np.abs(got - truth) / scale) / eps)


@pytest.mark.parametrize("sym", [1, 0])
@pytest.mark.parametrize("N, d, jct", [(700, 320, 0), (1500, 700, 2), (900, 2049, 0), (2700, 257, 5)])
@pytest.mark.parametrize("kernel", KERNELS)
def test_fp64_on_more_than_256_features(oracle, kernel, N, d, jct, sym):
    """fp64 beyond the widest resident row panel (256 features): the linear kernel as one pass of tile_matvec_f64_v2 per feature panel of
    128, rbf / polynomial on tile_matvec_f64_wide (panels of 64 inside a sub-tile, lssvm_tile_f64_wide.hip.hpp), both variants; ragged
    feature counts (the data is padded to whole panels), 1 ... 12 tiles per work item.  One implicit matvec against the float64 product
    formed with numpy, on the scale of each row's summands, and a short solve against the same iterations in numpy."""
    X, y = make_blobs_pm1(N, d, seed=13, dtype=np.float64)
    P = dict(degree=3, gamma=1.0 / d, coef0=0.5, cost=2.0)
    p = prm(kernel, P)
    n = N - 1
    v = np.random.default_rng(5).uniform(-1, 1, n)
    zero = np.zeros(n)
    _capi.set_option("j_chunk_tiles", jct)
    _capi.set_option("symmetric", sym)
    with backend.ResidentProblem(p, X) as prob:
        # (the full-square linear fall-back is the generic kernel; everything else here is a panel path)
        assert prob.info()["symmetric"] == sym
        got = prob.matvec(v, zero, 1.0)
        prob.cg_begin(y, 1e-30)
        prob.cg_step(4)
        a_panels = prob.cg_finish()[0]
    G = X @ X.T
    if kernel == "linear":
        Ka = G
    elif kernel == "polynomial":
        Ka = (G / d + 0.5) ** 3
    else:
        sq = np.einsum("ij,ij->i", X, X)
        Ka = np.exp(-np.maximum(sq[:, None] + sq[None, :] - 2.0 * G, 0.0) / d)
    K, q, QA = Ka[:n, :n], Ka[:n, n], Ka[n, n] + 0.5
    S = float(v.sum())
    truth = K @ v + v * 0.5 + (QA * S - float(q @ v)) - S * q
    scale = np.abs(K) @ np.abs(v) + np.abs(v) * 0.5 + abs(QA * S) + abs(float(q @ v)) + np.abs(S * q)
    assert np.max(np.abs(got - truth) / scale) < 64 * np.finfo(np.float64).eps, float(np.max(np.abs(got - truth) / scale) / np.finfo(np.float64).eps)
    # four CG iterations (x0 = 1, csvm.cpp:95-163) against the same iterations in numpy on the float64 matrix: the start from x0 = 1 amplifies
    # rounding differences by ~1e8 (the generic full-square kernel is 3e-9 ... 2e-6 from it on these shapes, the panel kernels 6e-8 ... 2e-6:
    # tests/tools/cg_check_f64_wide.py); alpha_N = -sum(alpha) is left out (a cancellation of n terms)
    A = K + 0.5 * np.eye(n) + QA - q[:, None] - q[None, :]
    b = y[:n] - y[n]
    x = np.ones(n)
    res = b - A @ x
    dvec = res.copy()
    delta = res @ res
    for _ in range(4):
        Ad = A @ dvec
        alpha_cd = delta / (dvec @ Ad)
        x = x + alpha_cd * dvec
        res = res - alpha_cd * Ad
        delta_new = res @ res
        dvec = res + (delta_new / delta) * dvec
        delta = delta_new
    assert np.max(np.abs(a_panels[:n] - x)) / np.max(np.abs(x)) < 1e-5


@pytest.mark.parametrize("mode", [3, 1])
@pytest.mark.parametrize("sym", [1, 0])
@pytest.mark.parametrize("degree, coef0", [(4, 0.0), (1, 1.0), (5, 0.5), (0, 1.0)])
def test_fp32_wide_polynomial_with_the_run_time_integer_power(degree, coef0, sym, mode):
    """The generic-degree instantiations of the fp32 panel kernel (every degree but 2 and 3; both plane kinds, both variants) on the shape and
    the all-ones vector that exposed the broken fp64 counterpart: every row against the float64 product."""
    N, d = 513, 640
    X, y = make_blobs_pm1(N, d, seed=166, dtype=np.float32)
    p = Parameter(kernel_type="polynomial", gamma=0.3 / d, degree=degree, coef0=coef0, cost=1.0)
    n = N - 1
    v = np.ones(n, np.float32)
    _capi.set_option("gram_mode", mode)
    _capi.set_option("symmetric", sym)
    with backend.ResidentProblem(p, X) as prob:
        assert prob.info()["symmetric"] == sym and prob.info()["gram_mode"] == (1 if mode == 1 else 2)
        out = prob.matvec(v, np.zeros(n, np.float32), 1.0).astype(np.float64)
    X64 = X.astype(np.float64)
    Ka = (p.gamma * (X64 @ X64.T) + coef0) ** degree
    K, q, QA = Ka[:n, :n], Ka[:n, n], Ka[n, n] + 1.0
    S = float(n)
    truth = K.sum(axis=1) + 1.0 + (QA * S - float(q.sum())) - S * q
    scale = np.abs(K).sum(axis=1) + 1.0 + abs(QA * S) + abs(float(q.sum())) + np.abs(S * q)
    assert np.max(np.abs(out - truth) / scale) < 16 * np.finfo(np.float32).eps


@pytest.mark.parametrize("sym", [1, 0])
@pytest.mark.parametrize("degree, coef0", [(4, 0.0), (1, 1.0), (5, 0.5), (0, 1.0)])
def test_fp64_wide_polynomial_with_the_run_time_integer_power(degree, coef0, sym):
    """The run-time-degree instantiation of the fp64 panel kernel on the shape and the all-ones vector with which tests/tools/wide_stress.py
    found it WRONG in round 3.  Root cause (round 4, profiles/r04_miscompile_f64_wide_root_cause.log): the optimiser folded the splat start value
    into one register quad serving as the C operand of the first MFMA of all eight accumulators and loaded the next B fragments into that quad
    right behind the eighth -- on gfx950 a load into the C operand of an in-flight v_mfma_f64_16x16x4_f64 corrupts its last rows, and the
    compiler has no hazard rule for it (tests/tools/repro/dgemm_srcc_war.hip).  The kernel now keeps the start values in the accumulators' own
    registers, and the build audits every kernel for the sequence.  Every row against the float64 product."""
    N, d = 513, 320
    X, y = make_blobs_pm1(N, d, seed=166, dtype=np.float64)
    p = Parameter(kernel_type="polynomial", gamma=0.3 / d, degree=degree, coef0=coef0, cost=1.0)
    n = N - 1
    v = np.ones(n)
    _capi.set_option("symmetric", sym)
    with backend.ResidentProblem(p, X) as prob:
        assert prob.info()["symmetric"] == sym
        out = prob.matvec(v, np.zeros(n), 1.0)
    Ka = (p.gamma * (X @ X.T) + coef0) ** degree
    K, q, QA = Ka[:n, :n], Ka[:n, n], Ka[n, n] + 1.0
    S = float(v.sum())
    truth = K @ v + v + (QA * S - float(q @ v)) - S * q
    scale = np.abs(K) @ np.abs(v) + np.abs(v) + abs(QA * S) + abs(float(q @ v)) + np.abs(S * q)
    assert np.max(np.abs(out - truth) / scale) < 64 * np.finfo(np.float64).eps


@pytest.mark.parametrize("mode", [3, 1])
@pytest.mark.parametrize("N, d, jct", [(700, 520, 0), (1500, 640, 2), (1500, 1030, 0), (2700, 400, 5)])
@pytest.mark.parametrize("kernel", ["rbf", "polynomial"])
def test_wide_rbf_and_polynomial_on_the_split_kernels(oracle, kernel, N, d, jct, mode):
    """fp32 rbf / polynomial on more features than a row panel in registers holds (f16x3: 384 rbf / 512 polynomial, bf16x6: 384): the
    kernel of lssvm_tile_f32_wide.hip.hpp walks feature panels of 128 inside a tile, re-loading the row panel per panel.  Feature counts that
    are not multiples of 128 (zero-padded planes), 5 ... 9 panels, work items of 1 ... 12 tiles (items cut at the diagonal included), both
    plane kinds (mode 3: f16x3 where the data allows, mode 1: bf16x6), symmetric and full-square variant; against the float64 Gram matrix on
    the scale of each row's summands, and the same solve trajectory as the full-square generic kernel.  (400 features: rbf only -- the polynomial kernel still fits one pass.)"""
    X, y = make_blobs_pm1(N, d, seed=11, dtype=np.float32)
    P = dict(degree=3, gamma=1.0 / d, coef0=0.5, cost=2.0)
    p = prm(kernel, P)
    n = N - 1
    v = np.random.default_rng(5).uniform(-1, 1, n).astype(np.float32)
    zero = np.zeros(n, np.float32)
    _capi.set_option("gram_mode", mode)
    _capi.set_option("j_chunk_tiles", jct)
    with backend.ResidentProblem(p, X) as prob:
        info = prob.info()
        wide = d > (384 if kernel == "rbf" or mode == 1 else 512)
        assert info["symmetric"] == 1 and info["gram_mode"] == (1 if mode == 1 else 2)
        q, QA = prob.q()
        got = prob.matvec(v, zero, 1.0).astype(np.float64)
        prob.cg_begin(y, 1e-30)
        prob.cg_step(4)
        a_sym = prob.cg_finish()[0]
    assert wide or d == 400
    X64 = X.astype(np.float64)
    G = X64[:n] @ X64[:n].T
    if kernel == "polynomial":
        K = (G / d + 0.5) ** 3
    else:
        sq = np.einsum("ij,ij->i", X64[:n], X64[:n])
        K = np.exp(-np.maximum(sq[:, None] + sq[None, :] - 2.0 * G, 0.0) / d)
    kw = dict(degree=3, gamma=1.0 / d, coef0=0.5)
    q64 = oracle.q(kernel, X64, **kw)
    v64 = v.astype(np.float64)
    S = float(v64.sum())
    QA64 = float(oracle.kernel_function(kernel, X64[-1], X64[-1], **kw)) + 0.5
    truth = K @ v64 + v64 * 0.5 + (QA64 * S - float(q64 @ v64)) - S * q64
    scale = np.abs(K) @ np.abs(v64) + np.abs(v64) * 0.5 + abs(QA64 * S) + abs(float(q64 @ v64)) + np.abs(S * q64)
    eps = np.finfo(np.float32).eps
    assert np.max(np.abs(got - truth) / scale) < 16 * eps, float(np.max(np.abs(got - truth) / scale) / eps)
    # the full-square variant of the same kernels: rows only, no mirrored sums
    _capi.set_option("symmetric", 0)
    with backend.ResidentProblem(p, X) as prob:
        assert prob.info()["symmetric"] == 0 and prob.info()["gram_mode"] == (1 if mode == 1 else 2)
        got0 = prob.matvec(v, zero, 1.0).astype(np.float64)
    assert np.max(np.abs(got0 - truth) / scale) < 16 * eps, float(np.max(np.abs(got0 - truth) / scale) / eps)
    # the same four CG iterations on the full-square generic kernel (native v_mfma_f32 chains)
    _capi.set_option("tile_kernel", 1)
    with backend.ResidentProblem(p, X) as prob:
        assert prob.info()["symmetric"] == 0 and prob.info()["gram_mode"] == 0
        prob.cg_begin(y, 1e-30)
        prob.cg_step(4)
        a_full = prob.cg_finish()[0]
    with backend.ResidentProblem(p, X.astype(np.float64)) as prob:
        prob.cg_begin(y.astype(np.float64), 1e-30)
        prob.cg_step(4)
        a64 = prob.cg_finish()[0]
    assert ol.rel_inf(a_sym[:-1], a64[:-1]) < 2 * ol.rel_inf(a_full[:-1], a64[:-1]) + 1e-4




FP32_CG = None


def _fp32_cg():
    global FP32_CG
    if FP32_CG is None:
        FP32_CG = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fp32_cg.npz"))
    return FP32_CG


def _fp32_cg_inputs(inputs, name):
    if name == "blobs2000x64":
        X, y = make_blobs_pm1(2000, 64, seed=5, dtype=np.float64)
        return X.astype(np.float32), y.astype(np.float32)
    return inputs[name + "_X"].astype(np.float32), inputs[name + "_y"].astype(np.float32)


@pytest.mark.parametrize("mode", [3, 1, 0])
@pytest.mark.parametrize("pname", ["ref", "def"])
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("name", ["500x200", "blobs263x37", "blobs2000x64"])
def test_fp32_alpha_against_the_reference_as_the_north_star_writes_it(inputs, name, kernel, pname, mode):
    """BASELINE.json: "alpha within 1e-4 rel-inf of OpenMP".  In fp64 that holds at 1e-6 (test_solve_f64_vs_golden).  In fp32 the sentence cannot be met by
    ANY implementation, the reference included, and the achieved figures are recorded instead of a widened bar (profiles/r04_fp32_cg_gpu_vs_reference.log,
    printed here with pytest -s):
      * the reference against ITSELF: the same binary with one and with eight OpenMP threads (only the order of its `omp atomic` partial sums differs)
        ends 2e-4 ... 3e-1 apart in alpha on all 18 systems, its own well-conditioned test parameters (cost = 0.1, 4 iterations) included
        (tests/golden/make_golden_fp32_cg.py, profiles/r04_ref_fp32_self_reproducibility.log);
      * the reference against the float64 solve of the same system: 6e-4 ... 1.4 (its sequential fp32 sums over n terms);
      * the GPU against the float64 solve: 3e-5 ... 6e-1, closer than the reference on every system measured -- so its distance to the reference's fp32
        run IS the reference's own error (4e-4 ... 9e-1), whatever the Gram mode.
    Asserted for the three Gram modes (f16x3, bf16x6, native v_mfma_f32): the stop criterion reached, delta_0 equal to 1e-5, the iteration count within 2
    (+ the reference's own spread), alpha no farther from the float64 solve than twice the reference's fp32 solve (or 1e-4) -- and the north_star's
    sentence as written wherever the reference meets it against itself and against float64 (on these systems: nowhere)."""
    G = _fp32_cg()
    X, y = _fp32_cg_inputs(inputs, name)
    P = dict(PARAM_SETS[pname])
    if P["gamma"] is None:
        P["gamma"] = 1.0 / X.shape[1]
    key = f"{name}/{kernel}/{pname}"
    a1, a8, a64 = G[f"t1/{key}/alpha"], G[f"t8/{key}/alpha"], G[f"t1/{key}/alpha64"]
    _capi.set_option("gram_mode", mode)
    with backend.ResidentProblem(prm(kernel, P), X) as prob:
        prob.cg_begin(y, 1e-6)
        d0 = prob.info()["residuum"]
        prob.cg_step(X.shape[0])
        a, rho, info = prob.cg_finish()
    e_gpu, e_self = ol.rel_inf(a, a1), ol.rel_inf(a8, a1)
    e_gpu64, e_ref64 = ol.rel_inf(a, a64), ol.rel_inf(a1, a64)
    print(f"\n{key:30s} gram mode {mode}: iterations {info['iterations']} (reference {int(G[f't1/{key}/iterations'])} / {int(G[f't8/{key}/iterations'])})  alpha vs the reference's 1-thread run {e_gpu:.2e}"
          f"  [reference 8 threads vs 1 thread {e_self:.2e}]  vs float64 {e_gpu64:.2e} [reference {e_ref64:.2e}]  rho {abs(float(rho) - float(G[f't1/{key}/rho'])):.2e}")
    assert info["converged"] and info["residuum"] <= 1e-12 * d0 * (1 + 1e-6)
    assert abs(d0 - float(G[f"t1/{key}/delta0"])) <= 1e-5 * float(G[f"t1/{key}/delta0"])
    assert abs(int(info["iterations"]) - int(G[f"t1/{key}/iterations"])) <= 2 + abs(int(G[f"t8/{key}/iterations"]) - int(G[f"t1/{key}/iterations"]))
    assert e_gpu64 <= max(2.0 * e_ref64, 1e-4), (e_gpu64, e_ref64)
    if e_self <= 1e-4 and e_ref64 <= 1e-4:
        assert e_gpu <= 1e-4, (e_gpu, e_self, e_ref64)


FP32_CG_FIXED = None


@pytest.mark.parametrize("mode", [3, 1, 0])
@pytest.mark.parametrize("pname", ["ref", "def"])
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("name", ["500x200", "blobs263x37", "blobs2000x64"])
def test_fp32_alpha_after_one_two_and_three_iterations_against_the_reference(inputs, name, kernel, pname, mode):
    """"alpha within 1e-4 rel-inf of OpenMP" (BASELINE.json north_star) asserted where it is ATTAINABLE in fp32 -- after a FIXED number of CG
    iterations k = 1, 2, 3 (eps = 1e-30), before the recursion has amplified rounding differences -- against the reference's own kernels run with one thread
    (tests/golden/fp32_cg_fixed.npz, generator tests/golden/make_golden_r05.py; recipe src/plssvm/backends/OpenMP/csvm.cpp:125-166).  The fixture also holds
    the reference's float64 run of the same k iterations: with the reference's start vector x0 = 1 the first residual r0 = b - A 1 is a difference of row sums
    over n terms, and the reference's sequential fp32 sums miss their own float64 result by 2e-7 ... 2e-2 after ONE iteration (and by up to 27 after two: the
    generator prints the table).  So, for every k and all three Gram modes:
      * where the reference's fp32 run agrees with its float64 run to 5e-5 (9 of the 18 systems at k = 1), alpha_gpu is within 1e-4 rel-inf of the reference's
        fp32 alpha -- the north_star sentence as written;
      * wherever the reference's fp32 run still has a digit (within 0.1 of its float64 run), alpha_gpu is no farther from the float64 run than twice the
        reference's fp32 run is (or 1e-4): the distance to the reference is the reference's own rounding, not the GPU's."""
    global FP32_CG_FIXED
    if FP32_CG_FIXED is None:
        FP32_CG_FIXED = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fp32_cg_fixed.npz"))
    G = FP32_CG_FIXED
    X, y = _fp32_cg_inputs(inputs, name)
    P = dict(PARAM_SETS[pname])
    if P["gamma"] is None:
        P["gamma"] = 1.0 / X.shape[1]
    _capi.set_option("gram_mode", mode)
    asserted_as_written = 0
    with backend.ResidentProblem(prm(kernel, P), X) as prob:
        for k in (1, 2, 3):
            key = f"{name}/{kernel}/{pname}/k{k}"
            a_ref, a64 = G[key + "/alpha"], G[key + "/alpha64"]
            prob.cg_begin(y, 1e-30)
            prob.cg_step(k)
            a, rho, info = prob.cg_finish()
            assert info["iterations"] == k
            e_gpu, e_gpu64, e_ref64 = ol.rel_inf(a, a_ref), ol.rel_inf(a, a64), ol.rel_inf(a_ref, a64)
            print(f"\n{key:36s} gram mode {mode}: alpha vs the reference's fp32 run {e_gpu:.2e}   vs its float64 run {e_gpu64:.2e} [the reference's fp32 run: {e_ref64:.2e}]"
                  f"   rho {abs(float(rho) - float(G[key + '/rho64'])):.2e} [{abs(float(G[key + '/rho']) - float(G[key + '/rho64'])):.2e}]")
            if e_ref64 < 0.1:  # (beyond that the reference's fp32 run has lost every digit against its own float64 run: nothing to hold the GPU to)
                assert e_gpu64 <= max(2.0 * e_ref64, 1e-4), (key, e_gpu64, e_ref64)
            if e_ref64 <= 5e-5:
                assert e_gpu <= 1e-4, (key, e_gpu, e_ref64)
                asserted_as_written += 1
    if (name, kernel, pname) in {("500x200", "linear", "ref"), ("500x200", "rbf", "def"), ("blobs263x37", "linear", "def"), ("blobs2000x64", "linear", "def"), ("blobs2000x64", "rbf", "def")}:
        assert asserted_as_written >= 1  # (these systems meet the sentence at k = 1 in the fixture: the branch above is not vacuous)


PAIR_MATVEC = None


@pytest.mark.parametrize("mode", [3, 1, 0])
@pytest.mark.parametrize("name, kernel, N, d", [("rbf8704x128", "rbf", 8704, 128), ("linear8704x256", "linear", 8704, 256)])
def test_matvec_at_a_size_that_reaches_the_256_row_kernel_vs_the_references_own_kernels(name, kernel, N, d, mode):
    """tile_matvec_f32_pair (256-row workgroups on block pairs, the headline kernel) runs from 64 row blocks = 8 192 points on; the goldens
    generated from the reference's kernels stopped at 2 000 points, so that kernel was pinned only through the restated oracle.  tests/golden/pair_matvec.npz
    holds q, QA_cost and 512 sampled rows of ONE implicit matvec of the reference's own OpenMP kernels (liblssvm_ref.so, one thread; svm_kernel.cpp:33-54,
    q_kernel.cpp:18-55) on 8 704 x 128 rbf and 8 704 x 256 linear (two feature-panel passes of the pair kernel), fp32, plus the same rows of its float64 run.
    The reference tests its kernels on a 5 000 x 2 000 file with 128 eps element-wise (tests/CMakeLists.txt:36-69, custom_test_macros.hpp:114-137)."""
    global PAIR_MATVEC
    if PAIR_MATVEC is None:
        PAIR_MATVEC = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pair_matvec.npz"))
    G = PAIR_MATVEC
    import hashlib

    X, _ = make_blobs_pm1(N, d, seed=11, dtype=np.float32)
    assert np.array_equal(np.frombuffer(hashlib.sha256(X.tobytes()).digest(), dtype=np.uint8), G[f"{name}/X_sha256"])  # the inputs the fixture was made from
    n = N - 1
    rhs = np.random.default_rng(77).uniform(1.0, 2.0, size=n).astype(np.float32)
    rows = G[f"{name}/rows"]
    eps = np.finfo(np.float32).eps
    _capi.set_option("gram_mode", mode)
    p = Parameter(kernel_type=kernel, degree=3)
    with backend.ResidentProblem(p, X) as prob:
        info = prob.info()
        assert info["symmetric"] == 1 and info["gram_mode"] == {3: 2, 1: 1, 0: 0}[mode]
        q, QA = prob.q()
        got = prob.matvec(rhs, np.zeros(n, np.float32), 1.0)
    if kernel == "linear":
        assert np.array_equal(q, G[f"{name}/q"])
    assert ol.float_near(q, G[f"{name}/q"], 16) and abs(float(QA) - float(G[f"{name}/QA_cost"])) <= 4 * eps * abs(float(G[f"{name}/QA_cost"]))
    ref32, ref64, scale = G[f"{name}/matvec_p1_rows"], G[f"{name}/matvec_p1_rows64"], float(G[f"{name}/matvec_p1_absmax"])
    e_ref = float(np.max(np.abs(ref32 - ref64))) / scale
    e_gpu = float(np.max(np.abs(got[rows] - ref64))) / scale
    e_vs_ref = float(np.max(np.abs(got[rows].astype(np.float64) - ref32))) / scale
    print(f"\n{name} gram mode {mode}: sampled rows vs the reference's float64 run {e_gpu / eps:.2f} eps [the reference's fp32 run: {e_ref / eps:.1f} eps]   vs the reference's fp32 run {e_vs_ref / eps:.1f} eps")
    assert ol.float_near(got[rows], ref32, 128)          # the reference's own element-wise criterion against its fp32 result
    assert e_vs_ref <= e_ref + 4 * eps                   # as close to the reference's fp32 rows as the reference's own rounding allows
    assert e_gpu <= 6 * eps                              # and within 6 eps of what the reference's kernels compute in float64 (measured 1.4 rbf, 3.6 linear; the reference's fp32 run: 42, 33)
