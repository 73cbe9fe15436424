"""CPU: host-side logic around the hot path -- parameters, LIBSVM reader, synthetic data, row-block partition."""

import os

import numpy as np
import pytest

from plssvm_amd import sharding
from plssvm_amd.datagen import generate_libsvm_file, make_blobs_pm1
from plssvm_amd.exceptions import InvalidFileFormatError, InvalidParameterError, UnsupportedKernelTypeError
from plssvm_amd.io_libsvm import parse_libsvm_data, write_libsvm_data
from plssvm_amd.parameter import KernelFunctionType, Parameter, kernel_function_type_from_string


def test_parameter_defaults_and_gamma_resolution():
    p = Parameter()                                         # parameter.hpp:156-165
    assert (p.kernel_type, p.degree, p.gamma, p.coef0, p.cost) == (KernelFunctionType.LINEAR, 3, None, 0.0, 1.0)
    assert Parameter(kernel_type="rbf").resolved(128).gamma == 1.0 / 128    # csvm.hpp:303-307
    assert Parameter(kernel_type="rbf", gamma=0.5).resolved(128).gamma == 0.5
    with pytest.raises(InvalidParameterError):              # csvm.hpp:384
        Parameter(kernel_type="rbf", gamma=-1.0)
    Parameter(kernel_type="linear", gamma=-1.0)             # gamma is ignored by the linear kernel
    assert kernel_function_type_from_string("poly") == KernelFunctionType.POLYNOMIAL
    assert kernel_function_type_from_string("2") == KernelFunctionType.RBF
    with pytest.raises(UnsupportedKernelTypeError):
        kernel_function_type_from_string("sigmoid")
    assert str(KernelFunctionType.RBF) == "rbf"


def test_libsvm_dense_sparse_and_roundtrip(tmp_path, inputs):
    # the reference's 5x4.libsvm content is part of the committed inputs; write + re-read it through the LIBSVM text format
    X, y = inputs["5x4_X"], inputs["5x4_y"]
    f = tmp_path / "a.libsvm"
    write_libsvm_data(f, X, labels=[int(v) for v in y], comment="comment")
    X2, y2 = parse_libsvm_data(f)
    assert np.allclose(X2, X, rtol=1e-10) and list(y2) == list(y)
    # sparse lines, blank lines, comments, label-only line (tests/data/libsvm/5x4_sparse.libsvm has the same shapes of line)
    f.write_text("# c\n1\n1  2:0.5\n-1 1:1.25\n\n-1 1:0.25      3:-0.125\n-1  4:2.0   # trailing\n   # another\n")
    Xs, ys = parse_libsvm_data(f)
    assert Xs.shape == (5, 4) and ys == [1, 1, -1, -1, -1]
    assert Xs[1, 1] == 0.5 and Xs[3, 2] == -0.125 and Xs[4, 3] == 2.0 and Xs[0].sum() == 0
    f.write_text("1:1.0 2:2.0\n2:3.0\n")
    Xn, yn = parse_libsvm_data(f)
    assert yn is None and Xn.shape == (2, 2)


@pytest.mark.parametrize("text, msg", [
    ("1 0:1.0 1:2.0\n", "1-based"),                              # zero_based_features
    ("1 2:1.0 1:2.0\n", "strictly increasing"),                  # non_increasing_indices
    ("1 1:1.0 1:2.0\n", "strictly increasing"),                  # non_strictly_increasing_indices
    ("1 1:1.0\n2:1.0\n", "Inconsistent label"),                  # inconsistent_label_specification
    ("1 1:a1.0\n", "Can't convert"),                             # feature_with_alpha_char_at_the_beginning
    ("1 a1:1.0\n", "Can't convert"),                             # index_with_alpha_char_at_the_beginning
    ("1 1:\n", "Can't convert"),                                 # missing_feature_value
    ("", "no data points"),
])
def test_libsvm_invalid_files(tmp_path, text, msg):
    f = tmp_path / "bad.libsvm"
    f.write_text(text)
    with pytest.raises(InvalidFileFormatError, match=msg):
        parse_libsvm_data(f)


def test_synthetic_data_recipe(tmp_path):
    X, y = make_blobs_pm1(1000, 16, seed=1, dtype=np.float32)
    assert X.shape == (1000, 16) and X.dtype == np.float32
    assert set(np.unique(y)) == {-1.0, 1.0} and abs(y.sum()) == 0           # balanced, labels +-1
    assert np.allclose(X.min(axis=0), -1, atol=1e-6) and np.allclose(X.max(axis=0), 1, atol=1e-6)  # min-max scaled per feature
    X2, _ = make_blobs_pm1(1000, 16, seed=1, dtype=np.float32)
    assert np.array_equal(X, X2)                                            # seeded
    f = tmp_path / "g.libsvm"
    generate_libsvm_file(f, 50, 4, seed=2)
    Xf, yf = parse_libsvm_data(f)
    assert Xf.shape == (50, 4) and set(yf) == {-1, 1}


@pytest.mark.parametrize("n, world", [(1, 1), (127, 2), (128, 2), (129, 2), (499, 3), (49_999, 8), (999_999, 8), (300, 8)])
def test_row_block_partition_covers_all_rows_once(n, world):
    parts = sharding.row_block_partition(n, world)
    assert len(parts) == world and parts[0][0] == 0 and parts[-1][1] == n
    for (a0, a1), (b0, b1) in zip(parts, parts[1:]):
        assert a1 == b0 and a0 <= a1
    # every interior boundary is a multiple of the 128-row block; runs of blocks differ by at most one block run
    assert all(a % 128 == 0 or a == n for a, _ in parts)
    sizes = [b - a for a, b in parts]
    assert max(sizes) <= -(-(-(-n // 128)) // world) * 128
    assert sharding.padded_vector_length(n, world) % (128 * world) == 0
    assert sharding.padded_vector_length(n, world) >= n


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_native_libsvm_reader_equals_python_parser(tmp_path, dtype):
    """libplssvm_amd's multi-threaded reader (fast path for well-formed files) against the line-by-line Python parser, on the
    reference's own data files' shapes: dense, sparse with gaps, unlabelled, comments and blank lines, CRLF, skipped lines."""
    from plssvm_amd import _capi  # noqa: F401  (the library must be loadable for the fast path to be exercised)
    from plssvm_amd.io_libsvm import _parse_native

    cases = {
        "dense": "1 1:0.5 2:-1.25e-3 3:7\n-1 1:1 2:2 3:3\n",
        "sparse": "# header comment\n\n1 2:0.25 7:1e10\r\n  -1 1:3\n1 5:-0.0 9:4.5\n",
        "unlabelled": "1:1.5 4:2\n2:3 3:4\n",
        "float_labels": "0.5 1:1\n-2.25 2:1\n",
    }
    for name, text in cases.items():
        f = tmp_path / f"{name}.libsvm"
        f.write_text(text)
        for skipped in (0, 1):
            fast = _parse_native(str(f), dtype, skipped, float)
            assert fast is not None, name
            slow = parse_libsvm_data(f, dtype=dtype, skipped_lines=skipped, use_native=False)
            assert fast[0].dtype == slow[0].dtype and np.array_equal(fast[0], slow[0]), name
            assert fast[1] == slow[1], name
    # a larger generated file: many lines per thread
    g = tmp_path / "gen.libsvm"
    generate_libsvm_file(g, 5000, 17, seed=3)
    A, a = parse_libsvm_data(g, dtype=dtype)
    B, b = parse_libsvm_data(g, dtype=dtype, use_native=False)
    assert np.array_equal(A, B) and a == b


@pytest.mark.parametrize("text", ["1 1:1 1:2\n", "1 0:1\n", "1 1:abc\n", "1 1:1\n2:2\n", "1 1:1 # trailing comment\n", "1\t1:2\n", "1 2:1 1:1\n", "x 1:1\n"])
def test_native_libsvm_reader_declines_what_it_cannot_vouch_for(tmp_path, text):
    """Irregular input is never accepted by the fast path: it reports failure and the Python parser decides."""
    from plssvm_amd.io_libsvm import _parse_native

    f = tmp_path / "bad.libsvm"
    f.write_text(text)
    assert _parse_native(str(f), np.float64, 0, float) is None


def test_bench_parent_stops_when_its_ranks_fail():
    """`python bench.py --gpus 2` without a launcher starts its ranks as child processes.  Here there is no GPU: every rank exits with an error,
    and the parent must come back with a non-zero code instead of waiting on a blocked sibling (it polls all children and stops the rest on
    the first failure)."""
    import os
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "c2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0 and time.time() - t0 < 300
    assert "no HIP device is visible" in out.stderr or "needs an MI355X" in out.stderr, out.stderr[-500:]


def test_the_grid_plane_rbf_scheme_in_the_numpy_model():
    """Round 5: the arithmetic of the rbf kernel on GRID planes (tile_matvec_f32_g6h) restated in numpy (tests/tools/grid_planes_model.py) --
    float16 planes, exact products, one fp32 rounding per 32-feature MFMA.  On clustered data with an exponent scale of ~1300 (gamma = 10) the accumulator chain over the
    grid plane must be EXACT for every pair whose kernel value is not zero in fp32, the grid planes must stay within a small factor of the formula-exact direct form, and
    the norm expansion they replace must be two orders of magnitude off -- the premise of the automatic choice between exponent scales of 32 and 4 096."""
    import importlib.util
    import math

    from conftest import ROOT

    spec = importlib.util.spec_from_file_location("grid_planes_model", os.path.join(ROOT, "tests", "tools", "grid_planes_model.py"))
    gm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gm)
    X = gm.clustered(np.random.default_rng(3), 256, 128, 0.02)
    r = gm.model(X, 10.0)
    assert 500 < r["r2"] < 8192 and r["hh_exact"]
    assert r["grid planes"][1] < 4 * max(r["direct"][1], 1.0) and r["grid planes"][0] < 4 * max(r["direct"][0], 2.0)
    assert r["norm expansion"][1] > 50 * r["grid planes"][1]
    for r2 in (1.0, 33.0, 500.0, 8192.0):
        g, sigma = gm.grid_parameters(r2)
        assert (r2 + 160.0) / (g * g / 2.0) <= 2.0 ** 24 and math.sqrt(r2) / g <= 2048 and (math.sqrt(r2) + g) * sigma <= 65504
