"""CPU: the plain-C oracle (oracle/lssvm_oracle.c) against the golden vectors captured from the REFERENCE's own OpenMP
kernels (tests/golden/make_golden.py) and -- where oracle/_ref was built (this container) -- against that library live.
This is what pins the oracle."""

import numpy as np
import pytest

import oracle_lib as ol
from conftest import DATASETS, DTYPES, KERNELS, PARAM_SETS, resolved_kw


@pytest.mark.parametrize("name", DATASETS)
@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("pname", ["ref", "def"])
@pytest.mark.parametrize("kernel", KERNELS)
def test_q_and_matvec_match_reference(golden, inputs, oracle, name, tag, pname, kernel):
    dt = DTYPES[tag]
    X = inputs[name + "_X"].astype(dt)
    N, d = X.shape
    P = PARAM_SETS[pname]
    kw = resolved_kw(P, d)
    key = f"{name}/{kernel}/{tag}/{pname}"
    # q: the same sequential fma chain, same libm -> bit identical (q_kernel.cpp:18-55)
    q = oracle.q(kernel, X, **kw)
    assert np.array_equal(q, golden[key + "/q"])
    QA = golden[key + "/QA_cost"]
    qa = dt(oracle.kernel_function(kernel, X[-1], X[-1], **kw)) + dt(1.0) / dt(P["cost"])
    assert qa == QA
    rhs = golden[key + "/rhs"]
    for add, atag in ((1.0, "p1"), (-1.0, "m1")):
        got = oracle.matvec(kernel, X, q, rhs, np.zeros(N - 1, dt), QA, dt(1.0) / dt(P["cost"]), add, **kw)
        # multi-threaded atomics change the summation order only: the reference's own tolerance (custom_test_macros.hpp:114-137)
        assert ol.rel_inf(got, golden[f"{key}/matvec_{atag}"]) < 64 * np.finfo(dt).eps
        rows = oracle.matvec_rows(kernel, X, q, rhs, np.zeros(N - 1, dt), QA, dt(1.0) / dt(P["cost"]), add, 0, N - 1, **kw)
        assert ol.rel_inf(rows, golden[f"{key}/matvec_{atag}"]) < 64 * np.finfo(dt).eps


@pytest.mark.parametrize("name", DATASETS)
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("case", ["cg_tight", "cg_refresh", "cg_default"])
def test_cg_f64_matches_reference(golden, inputs, oracle, name, kernel, case):
    X = inputs[name + "_X"]
    y = inputs[name + "_y"]
    d = X.shape[1]
    key = f"{name}/{kernel}/f64/def/{case}"
    P = PARAM_SETS["def"]
    a, rho, info = oracle.solve(kernel, X, y, float(golden[key + "/eps"]), int(golden[key + "/max_iter"]), cost=P["cost"], **resolved_kw(P, d))
    assert abs(int(info["iterations"]) - int(golden[key + "/iterations"])) <= 3  # the stop test is a float comparison on a noisy plateau
    # 8 OpenMP threads push their partial sums with atomics in a varying order (the goldens are single-threaded); a +-1
    # difference in the stopping iteration of a tight solve moves alpha by up to ~1e-6
    assert ol.rel_inf(a, golden[key + "/alpha"]) < 1e-5
    assert abs(float(rho) - float(golden[key + "/rho"])) < 1e-6 * max(1.0, abs(float(golden[key + "/rho"])))


@pytest.mark.parametrize("name", ["5x4", "blobs263x37"])
@pytest.mark.parametrize("kernel", KERNELS)
def test_cg_f32_is_as_close_to_truth_as_the_reference(golden, inputs, oracle, name, kernel):
    """fp32 CG is chaotic: the reference does not reproduce ITSELF to 1e-4 across thread counts.  What
    can be pinned is the distance to the fp64 solution after the same number of iterations."""
    X = inputs[name + "_X"].astype(np.float32)
    y = inputs[name + "_y"].astype(np.float32)
    d = X.shape[1]
    P = PARAM_SETS["def"]
    key32 = f"{name}/{kernel}/f32/def/cg_default"
    key64 = f"{name}/{kernel}/f64/def/cg_default"
    a, rho, info = oracle.solve(kernel, X, y, 1e-3, X.shape[0], cost=P["cost"], **resolved_kw(P, d))
    assert int(info["iterations"]) == int(golden[key32 + "/iterations"])
    err_ours = ol.rel_inf(a, golden[key64 + "/alpha"])
    err_ref = ol.rel_inf(golden[key32 + "/alpha"], golden[key64 + "/alpha"])
    assert err_ours <= max(10 * err_ref, 1e-4)


def test_trivial_system_known_answer(oracle):
    """GenericCSVM.solve_system_of_linear_equations_trivial (tests/backends/generic_csvm_tests.hpp:99-137)."""
    for dt in (np.float32, np.float64):
        A = (np.sqrt(dt(1.0) - dt(1.0) / dt(2.0)) * np.eye(4)).astype(dt)
        rhs = np.array([1, -1, 1, -1], dtype=dt)
        for kernel, kw in (("linear", {}), ("polynomial", dict(degree=1, gamma=1.0, coef0=0.0))):
            x, rho, _ = oracle.solve(kernel, A, rhs, 1e-5, 4, cost=2.0, **kw)
            assert ol.float_near(x, rhs)
            assert abs(float(rho)) < 4 * np.finfo(dt).eps


def test_predict_values_known_answer(oracle):
    """GenericCSVM.predict_values (generic_csvm_tests.hpp:149-195)."""
    for dt in (np.float32, np.float64):
        sv = np.eye(4, dtype=dt)
        w8 = np.array([1, -1, 1, -1], dtype=dt)
        pts = np.array([[1, 1, 1, 1], [1, -1, 1, -1]], dtype=dt)
        out, w = oracle.predict_values("linear", sv, w8, 0.0, pts)
        assert ol.float_near(out, np.array([0, 4], dtype=dt)) and ol.float_near(w, w8)
        out, w = oracle.predict_values("polynomial", sv, w8, 0.0, pts, degree=1, gamma=1.0, coef0=0.0)
        assert ol.float_near(out, np.array([0, 4], dtype=dt)) and w is None


def test_preconditions(oracle):
    X = np.ones((3, 2))
    with pytest.raises(RuntimeError):
        oracle.solve("linear", X, np.ones(3), 0.0, 3)      # eps > 0  (csvm.cpp:77)
    with pytest.raises(RuntimeError):
        oracle.solve("linear", X, np.ones(3), 1e-3, 0)     # max_iter > 0 (csvm.cpp:78)


@pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_oracle_vs_live_reference(oracle, kernel, dt):
    rng = np.random.default_rng(5)
    X = rng.uniform(-1, 1, size=(151, 23)).astype(dt)
    y = np.where(rng.uniform(size=151) > 0.5, 1, -1).astype(dt)
    ref = ol.ref()
    kw = dict(degree=3, gamma=1.0 / 23, coef0=0.5)
    assert np.array_equal(oracle.q(kernel, X, **kw), ref.q(kernel, X, **kw))
    q = ref.q(kernel, X, **kw)
    rhs = rng.uniform(1, 2, size=150).astype(dt)
    a = oracle.matvec(kernel, X, q, rhs, np.zeros(150, dt), 2.0, 1.0, 1.0, **kw)
    b = ref.matvec(kernel, X, q, rhs, np.zeros(150, dt), 2.0, 1.0, 1.0, **kw)
    assert ol.rel_inf(a, b) < 64 * np.finfo(dt).eps
    ao, ro, io = oracle.solve(kernel, X, y, 1e-30, 5, cost=1.0, **kw)
    ar, rr, ir = ref.solve(kernel, X, y, 1e-30, 5, cost=1.0, **kw)
    assert io["iterations"] == ir["iterations"] == 5
    if dt == np.float64:
        assert ol.rel_inf(ao, ar) < 1e-4  # atomics reorder the sums; 5 CG steps on unscaled data amplify 1e-16 to ~1e-6
    else:
        # fp32 CG amplifies the atomics' summation-order noise: compare both against the fp64 solve of the same length
        a64, _, _ = ref.solve(kernel, X.astype(np.float64), y.astype(np.float64), 1e-30, 5, cost=1.0, **kw)
        assert ol.rel_inf(ao, a64) <= max(10 * ol.rel_inf(ar, a64), 1e-4)
    sv, al, pts = X[:50], rhs[:50], X[100:120]
    po, _ = oracle.predict_values(kernel, sv, al, 0.25, pts, **kw)
    pr, _ = ref.predict_values(kernel, sv, al, 0.25, pts, **kw)
    assert ol.rel_inf(po, pr) < 64 * np.finfo(dt).eps


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_oracle_predict_reproduces_reference_fixture(oracle, kernel, dt):
    """GenericCSVM.predict / score (generic_csvm_tests.hpp:197-247): LIBSVM-trained models, exact labels, score == 1."""
    import os
    from conftest import HERE
    fx = np.load(os.path.join(HERE, "golden", "predict_500x200.npz"))
    gamma = float(fx[f"{kernel}_gamma"])
    kw = dict(degree=int(fx[f"{kernel}_degree"]), gamma=(gamma if gamma == gamma else 1.0 / 200), coef0=float(fx[f"{kernel}_coef0"]))
    out, _ = oracle.predict_values(kernel, fx[f"{kernel}_sv"].astype(dt), fx[f"{kernel}_alpha"].astype(dt), float(fx[f"{kernel}_rho"]), fx["test_X"].astype(dt), **kw)
    labels = np.where(out > 0, 1, -1)
    assert np.array_equal(labels, fx["expected"]) and np.array_equal(labels, fx["test_y"].astype(int))
