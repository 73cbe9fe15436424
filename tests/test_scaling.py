"""CPU: the scaling side of the data formats next to the path -- ``data_set::scaling`` (include/plssvm/data_set.hpp:290-375,
:670-730), its factor files (detail/io/scaling_factors_parsing.hpp:57-150) and the ``plssvm-scale`` command line
(src/main_scale.cpp, parser_scale.cpp), pinned on the data files and expectations of the reference's own tests
(tests/golden/scaling_factors/, written by tests/golden/make_scaling_fixture.py)."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from plssvm_amd.data_set import DataSet, DataSetError, Scaling
from plssvm_amd.exceptions import InvalidFileFormatError
from plssvm_amd.io_libsvm import parse_libsvm_data, write_libsvm_data
from plssvm_amd.io_scaling_factors import parse_scaling_factors, write_scaling_factors

FIX = os.path.join(ROOT, "tests", "golden", "scaling_factors")
EXPECTED = json.load(open(os.path.join(FIX, "expected.json")))


@pytest.mark.parametrize("real_type", [np.float32, np.float64])
@pytest.mark.parametrize("name", sorted(EXPECTED))
def test_factor_files_of_the_reference_tests(name, real_type):
    exp = EXPECTED[name]
    path = os.path.join(FIX, name)
    if "error" in exp:
        message = exp["error"].replace("{real}", "float" if real_type == np.float32 else "double")
        with pytest.raises(InvalidFileFormatError) as e:
            parse_scaling_factors(path, real_type)
        assert str(e.value) == message
        return
    (lower, upper), factors = parse_scaling_factors(path, real_type)
    assert (lower, upper) == tuple(real_type(v) for v in exp["interval"])
    assert [(f, lo, hi) for f, lo, hi in factors] == [(f, real_type(lo), real_type(hi)) for f, lo, hi in exp["factors"]]  # zero-based here, one-based in the file


def test_factor_file_round_trip_and_layout(tmp_path):
    factors = [(0, -1.5, 2.25), (2, 0.0, 1e-3), (7, 3.0, 4.0)]
    path = tmp_path / "f.txt"
    write_scaling_factors(path, (-1.0, 1.0), factors)
    lines = open(path).read().splitlines()
    assert lines[0].startswith("# These scaling factors have been created at ") and lines[1] == "x" and lines[2] == "-1 1"
    assert lines[3:] == ["1 -1.5 2.25", "3 0 0.001", "8 3 4"]  # one-based, {fmt}'s shortest digits
    assert parse_scaling_factors(path) == ((-1.0, 1.0), factors)
    with pytest.raises(InvalidFileFormatError, match="Illegal interval"):
        write_scaling_factors(path, (1.0, 1.0), factors)


@pytest.mark.parametrize("real_type", [np.float32, np.float64])
def test_scale_computes_and_restores_factors_like_the_reference(tmp_path, real_type):
    rng = np.random.default_rng(3)
    X = rng.normal(size=(40, 6)).astype(real_type)
    X[:, 2] = 0  # a feature that is zero everywhere: no factor, stays zero (data_set.hpp:693-695)
    ds = DataSet(X, [1, -1] * 20, real_type=real_type, scaling=(-2.0, 3.0))
    sc = ds.scaling_factors()
    assert ds.is_scaled() and [f for f, _, _ in sc.factors] == [0, 1, 3, 4, 5]
    lo, hi = real_type(-2.0), real_type(3.0)
    for f, mn, mx in sc.factors:
        assert mn == X[:, f].min() and mx == X[:, f].max()
        assert np.array_equal(ds.data()[:, f], lo + (hi - lo) * (X[:, f] - mn) / (mx - mn))  # the reference's expression, in the real type
        assert ds.data()[:, f].min() == lo and abs(ds.data()[:, f].max() - hi) <= 4 * np.finfo(real_type).eps * 3
    assert np.all(ds.data()[:, 2] == 0)
    # a second data set (the test set) gets the factors of the first through a file
    sc.save(tmp_path / "train.factors")
    T = rng.normal(size=(7, 6)).astype(real_type)
    restored = Scaling(filename=tmp_path / "train.factors", real_type=real_type)
    dt = DataSet(T, real_type=real_type, scaling=restored)
    for f, mn, mx in sc.factors:
        assert np.allclose(dt.data()[:, f], lo + (hi - lo) * (T[:, f] - mn) / (mx - mn), rtol=8 * np.finfo(real_type).eps, atol=0)
    assert np.array_equal(dt.data()[:, 2], T[:, 2])  # no factor: untouched
    assert not DataSet(T, real_type=real_type).is_scaled()


def test_scaling_errors_of_the_reference():
    X = np.arange(12, dtype=np.float64).reshape(4, 3)
    with pytest.raises(DataSetError, match=r"Inconsistent scaling interval specification: lower \(1\) must be less than upper \(-1\)!"):
        DataSet(X, scaling=(1.0, -1.0))
    sc = Scaling(-1.0, 1.0)
    sc.factors = [(0, 0.0, 1.0), (1, 0.0, 1.0), (2, 0.0, 1.0), (2, 0.0, 2.0)]
    with pytest.raises(DataSetError, match=r"Need at most as much scaling factors as features in the data set are present \(3\), but 4 were given!"):
        DataSet(X, scaling=sc)
    sc.factors = [(0, 0.0, 1.0), (3, 0.0, 1.0)]
    with pytest.raises(DataSetError, match=r"The maximum scaling feature index most not be greater than 2, but is 3!"):
        DataSet(X, scaling=sc)
    sc.factors = [(1, 0.0, 1.0), (0, 0.0, 1.0), (1, 0.0, 2.0)]
    with pytest.raises(DataSetError, match=r"Found more than one scaling factor for the feature index 1!"):
        DataSet(X, scaling=sc)


def _scale(*argv):
    return subprocess.run([sys.executable, "-m", "plssvm_amd.scale", *map(str, argv)], capture_output=True, text=True, cwd=ROOT, timeout=120)


def test_scale_command_line(tmp_path):
    rng = np.random.default_rng(5)
    X = rng.uniform(-3, 9, size=(30, 5))
    y = [1.0 if i % 3 else -1.0 for i in range(30)]
    train, test = tmp_path / "train.libsvm", tmp_path / "test.libsvm"
    write_libsvm_data(train, X, labels=y)
    write_libsvm_data(test, X[:8] * 0.5, labels=y[:8])
    out = _scale("-l", "0", "-u", "1", "-s", tmp_path / "factors.txt", train, tmp_path / "train.scaled")
    assert out.returncode == 0 and "Scaled the data set to the range [0, 1]." in out.stdout, out.stdout + out.stderr
    Xs, ys = parse_libsvm_data(tmp_path / "train.scaled")
    assert ys == y and np.allclose(Xs, (X - X.min(0)) / (X.max(0) - X.min(0)), rtol=0, atol=2e-10)  # the file keeps {:.10e}
    # the test set with the factors of the training set, printed to stdout ("\n" first, then "{label} {index}:{value:.10e} " per non-zero)
    out = _scale("-q", "-r", tmp_path / "factors.txt", test)
    assert out.returncode == 0, out.stderr
    rows = out.stdout.splitlines()
    assert rows[0] == "" and len(rows) == 9 and rows[1].startswith("-1 1:") and rows[1].endswith(" ")
    first = np.array([float(t.split(":")[1]) for t in rows[1].split()[1:]])
    assert np.allclose(first, (X[0] * 0.5 - X.min(0)) / (X.max(0) - X.min(0)), rtol=0, atol=2e-10)
    # -f arff: the scaled set in the reference's other data format; an .arff input is read with the ARFF parser
    out_arff = _scale("-q", "-r", tmp_path / "factors.txt", "-f", "arff", test, tmp_path / "test.scaled.arff")
    assert out_arff.returncode == 0, out_arff.stderr
    from plssvm_amd.io_arff import parse_arff_data
    Xa, ya = parse_arff_data(tmp_path / "test.scaled.arff")
    assert ya == y[:8] and np.allclose(Xa, (X[:8] * 0.5 - X.min(0)) / (X.max(0) - X.min(0)), rtol=0, atol=2e-10)
    again = _scale("-q", "-l", "0", "-u", "1", tmp_path / "test.scaled.arff")
    assert again.returncode == 0 and len(again.stdout.splitlines()) == 9
    assert "the output format must be libsvm or arff" in _scale("-f", "csv", train).stderr
    # the parser's errors (parser_scale.cpp:89-93, :131-135, :122-126)
    assert "Error invalid scaling range [lower, upper] with [1, 1]!" in _scale("-l", "1", "-u", "1", train).stderr
    assert "Error cannot use -s (--save_filename) and -r (--restore_filename) simultaneously!" in _scale("-s", "a", "-r", "b", train).stderr
    assert "Error missing input file!" in _scale().stderr
    warn = _scale("-q", "-l", "-5", "-r", tmp_path / "factors.txt", test)
    assert warn.returncode == 0 and "ignoring -l/-u" in warn.stderr and warn.stdout == out.stdout
    bad = _scale("-r", os.path.join(FIX, "invalid", "no_header.txt"), test)
    assert bad.returncode == 1 and "The first line must only contain an 'x'" in bad.stderr
