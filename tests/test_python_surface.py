"""Python surface next to the hot path: DataSet label mapping, LIBSVM model files, CSVM / SVC /
CLI.  CPU part: formats and host logic; GPU part (-m gpu): fit / predict / score through the backend."""

import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from plssvm_amd.data_set import DataSet, DataSetError
from plssvm_amd.exceptions import InvalidFileFormatError, InvalidParameterError, UnsupportedBackendError
from plssvm_amd.model import Model
from plssvm_amd.parameter import KernelFunctionType, Parameter


def test_label_mapping_follows_sorted_order(inputs):
    # data_set.hpp:438-454: first label in std::set order -> -1, second -> +1
    ds = DataSet(inputs["5x4_X"], [1, 1, -1, -1, -1])
    assert ds.different_labels() == [-1, 1] and list(ds.mapped_labels()) == [1, 1, -1, -1, -1]
    ds = DataSet(inputs["5x4_X"], ["cat", "dog", "cat", "dog", "dog"], real_type=np.float32)
    assert ds.different_labels() == ["cat", "dog"] and list(ds.mapped_labels()) == [-1, 1, -1, 1, 1]
    with pytest.raises(DataSetError, match="binary classification"):
        DataSet(inputs["5x4_X"], [1, 2, 3, 1, 2])
    ds = DataSet(inputs["5x4_X"], [0, 0, 1, 1, 1], scaling=(-1.0, 1.0))
    assert np.allclose(ds.data().min(axis=0), -1) and np.allclose(ds.data().max(axis=0), 1)


@pytest.mark.parametrize("kernel", ["linear", "polynomial", "rbf"])
def test_model_file_roundtrip_and_format(tmp_path, inputs, kernel):
    X = inputs["5x4_X"]
    ds = DataSet(X, [1, 1, -1, -1, -1])
    p = Parameter(kernel_type=kernel, degree=2, gamma=0.25, coef0=1.5)
    m = Model(p, ds, alpha=[-0.1, 0.8, -0.4, 0.003, -0.2], rho=0.3733)
    f = tmp_path / "m.model"
    m.save(f)
    lines = f.read_text().splitlines()
    assert lines[0].startswith("# This model file has been created at")
    assert lines[1] == "svm_type c_svc" and lines[2] == f"kernel_type {kernel}"
    body = lines[lines.index("SV") + 1:]
    assert len(body) == 5
    # grouped by class in label order (-1 first): the three -1 points, then the two +1 points; {:.10e}
    assert body[0].startswith("-4.0000000000e-01 1:") and body[3].startswith("-1.0000000000e-01 1:-1.1178275006e+00")
    assert "nr_sv 3 2" in lines and "label -1 1" in lines and "total_sv 5" in lines
    if kernel == "polynomial":
        assert "degree 2" in lines and "gamma 0.25" in lines and "coef0 1.5" in lines
    if kernel == "linear":
        assert not any(ln.startswith("gamma") for ln in lines)
    m2 = Model.load(f)
    assert m2.params.kernel_type == KernelFunctionType(["linear", "polynomial", "rbf"].index(kernel))
    assert m2.num_support_vectors() == 5 and m2.num_features() == 4 and abs(float(m2.rho) - 0.3733) < 1e-12
    order = [2, 3, 4, 0, 1]
    assert np.allclose(m2.support_vectors(), X[order], rtol=1e-9) and np.allclose(m2.alpha, np.array([-0.1, 0.8, -0.4, 0.003, -0.2])[order])
    assert m2.labels() == [-1, -1, -1, 1, 1]


def test_model_file_written_by_libsvm_style_tools_parses(tmp_path):
    # shape of tests/data/model/5x4_rbf.libsvm.model: header keys in another order, comment line, labels 0/1
    f = tmp_path / "x.model"
    f.write_text("# c\nsvm_type c_svc\nkernel_type rbf\ngamma 0.025\nnr_class 2\ntotal_sv 3\nrho 0.37\nlabel 0 1\nnr_sv 1 2\nSV\n"
                 "-0.17 1:-1.1 2:-2.9 3:6.6e-01 4:1.0\n0.88 1:-5.2e-01 4:5.4e-01\n-0.47 2:6.0e-01\n")
    m = Model.load(f)
    assert m.params.gamma == 0.025 and m.labels() == [0, 1, 1] and m.support_vectors().shape == (3, 4) and m.support_vectors()[2, 1] == 0.6


@pytest.mark.parametrize("text, msg", [
    ("svm_type nu_svc\nkernel_type linear\n", "Can only use c_svc"),
    ("svm_type c_svc\nkernel_type sigmoid\n", "Unrecognized kernel type"),
    ("svm_type c_svc\nkernel_type linear\ngamma 0.1\nnr_class 2\ntotal_sv 1\nrho 0\nlabel 0 1\nnr_sv 1 0\nSV\n1 1:1\n", "gamma parameter which is not used in the linear kernel"),
    ("svm_type c_svc\nkernel_type linear\nnr_class 2\ntotal_sv 2\nrho 0\nlabel 0 1\nnr_sv 1 2\nSV\n1 1:1\n", "but the sum of nr_sv is 3"),
    ("svm_type c_svc\nkernel_type linear\nnr_class 2\nrho 0\nlabel 0 1\nnr_sv 1 1\nSV\n1 1:1\n", "Missing total number of support vectors"),
    ("svm_type c_svc\nkernel_type linear\nfoo 1\n", "Unrecognized header entry"),
])
def test_model_file_invalid(tmp_path, text, msg):
    f = tmp_path / "bad.model"
    f.write_text(text)
    with pytest.raises(InvalidFileFormatError, match=msg):
        Model.load(f)


def test_factory_and_argument_validation_without_gpu():
    from plssvm_amd.csvm import BackendType, CSVM, make_csvm
    with pytest.raises(UnsupportedBackendError, match="No cuda backend available!"):   # csvm_factory.hpp:74-79
        make_csvm(BackendType.CUDA)
    with pytest.raises(UnsupportedBackendError, match="No openmp backend available!"):
        make_csvm("openmp")

    class Fake(CSVM):  # the reference's mock_csvm (tests/mock_csvm.hpp:34-85): canned answers prove fit's plumbing
        calls = 0

        def solve_system_of_linear_equations(self, params, A, b, eps, max_iter):
            Fake.calls += 1
            assert params.gamma == 0.25 and max_iter == 5 and list(b) == [1, 1, -1, -1, -1]
            return np.arange(1.0, 6.0), 3.1415, {"iterations": 1}

    X = np.arange(20.0).reshape(5, 4)
    ds = DataSet(X, [1, 1, -1, -1, -1])
    svm = Fake(kernel_type="rbf")
    model = svm.fit(ds)
    assert Fake.calls == 1 and list(model.alpha) == [1, 2, 3, 4, 5] and float(model.rho) == 3.1415 and model.params.gamma == 0.25
    with pytest.raises(InvalidParameterError, match="epsilon must be less than 0.0"):
        svm.fit(ds, epsilon=0.0)
    with pytest.raises(InvalidParameterError, match="max_iter must be greater than 0"):
        svm.fit(ds, max_iter=0)
    with pytest.raises(InvalidParameterError, match="No labels given for training"):
        svm.fit(DataSet(X))
    assert Fake.calls == 1  # the boundary is not reached on invalid arguments (tests/csvm.cpp:196-340)


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["linear", "polynomial", "rbf"])
@pytest.mark.parametrize("rt", [np.float32, np.float64])
def test_fit_predict_score_and_cli_roundtrip(tmp_path, kernel, rt):
    from plssvm_amd.csvm import make_csvm
    from plssvm_amd.datagen import make_blobs_pm1
    from plssvm_amd.io_libsvm import write_libsvm_data
    from plssvm_amd.svc import SVC
    X, y = make_blobs_pm1(600, 12, seed=11, dtype=np.float64)
    train, test = slice(0, 400), slice(400, 600)
    ds = DataSet(X[train], [int(v) for v in y[train]], real_type=rt)
    svm = make_csvm("mi355", params=Parameter(kernel_type=kernel))
    model = svm.fit(ds, epsilon=1e-6)
    assert svm.score(model) > 0.99 and svm.score(model, DataSet(X[test], [int(v) for v in y[test]], real_type=rt)) > 0.98
    # model file -> load -> identical predictions (f1 row: formats either side of the path)
    mf = tmp_path / "m.model"
    model.save(mf)
    m2 = Model.load(mf, real_type=rt)
    tds = DataSet(X[test], [int(v) for v in y[test]], real_type=rt)
    assert svm.predict(m2, tds) == svm.predict(model, tds)
    clf = SVC(kernel="poly" if kernel == "polynomial" else kernel, tol=1e-6, real_type=rt).fit(X[train], y[train])
    assert clf.score(X[test], y[test]) > 0.98
    # CLI: plssvm-train / plssvm-predict flags (parser_train.cpp:41-73)
    trf, tef = tmp_path / "train.libsvm", tmp_path / "test.libsvm"
    write_libsvm_data(trf, X[train], labels=[int(v) for v in y[train]])
    write_libsvm_data(tef, X[test], labels=[int(v) for v in y[test]])
    t = {"linear": "0", "polynomial": "1", "rbf": "2"}[kernel]
    extra = ["--use_float_as_real_type"] if rt == np.float32 else []
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-m", "plssvm_amd.train", "-t", t, "-e", "1e-6", "--verbosity", "libsvm", *extra, str(trf), str(tmp_path / "cli.model")],
                         capture_output=True, text=True, env=env, cwd=tmp_path)
    assert out.returncode == 0 and "optimization finished, #iter = " in out.stdout, out.stdout + out.stderr
    out = subprocess.run([sys.executable, "-m", "plssvm_amd.predict", *extra, str(tef), str(tmp_path / "cli.model"), str(tmp_path / "out.predict")],
                         capture_output=True, text=True, env=env, cwd=tmp_path)
    assert out.returncode == 0 and "Accuracy = " in out.stdout, out.stdout + out.stderr
    pred = [float(v) for v in (tmp_path / "out.predict").read_text().split()]
    assert np.mean(np.array(pred) == y[test]) > 0.98


def test_performance_tracker_yaml_layout(tmp_path):
    """plssvm_amd/performance_tracker.py writes what performance_tracker::save writes (src/plssvm/detail/performance_tracker.cpp:139-190):
    one YAML document per run -- '---', a meta_data block, then one block per category with two-space indented `name: value` lines, string
    values quoted (:36-38), the ungrouped total_time last (main_train.cpp:57); appended to the file."""
    import yaml

    from plssvm_amd.performance_tracker import PerformanceTracker

    info = dict(iterations=7, max_iterations=500, residuum=2.5e-12, target_residuum=1.0e-11, avg_iteration_ms=0.123456, epsilon=1e-10, total_runtime_ms=12.6,
                devices_used=2)
    out = tmp_path / "track.yaml"
    for run in range(2):
        tr = PerformanceTracker()
        tr.add_parameter(Parameter(kernel_type="rbf", gamma=0.25, cost=2.0), "float32")
        tr.add_backend(info["devices_used"])
        tr.add_cg_info(info)
        tr.add("data_set_read", "filename", "train.libsvm")
        tr.add("", "total_time", "15ms")
        tr.save(str(out))
    text = out.read_text()
    assert text.count("---\n") == 2 and text.startswith("---\nmeta_data:\n  date:")
    docs = list(yaml.safe_load_all(text))
    assert len(docs) == 2
    doc = docs[1]
    assert doc["parameter"] == {"kernel_type": "rbf", "degree": 3, "gamma": 0.25, "coef0": 0.0, "cost": 2.0, "real_type": "float"}
    assert doc["backend"] == {"backend": "mi355", "target_platform": "gpu_amd", "num_devices": 2}
    assert doc["cg"]["iterations"] == 7 and doc["cg"]["max_iterations"] == 500 and float(doc["cg"]["epsilon"]) == 1e-10  # ({} formatting like the reference: "1e-10", which YAML 1.1 reads as a string)
    assert doc["cg"]["avg_iteration_time"] == "0.123456ms" and doc["cg"]["total_runtime"] == "13ms"
    assert doc["data_set_read"]["filename"] == "train.libsvm" and doc["total_time"] == "15ms"
    assert '  filename: "train.libsvm"\n' in text and "\ncg:\n  iterations: 7\n" in text
