"""the command lines against the oracle and the reference's own artefacts (-m gpu).

  * `python -m plssvm_amd.train` (src/main_train.cpp:24-70) on the committed 5x4, 500x200 and 500x4 inputs: the model FILE it writes is parsed
    by a small independent parser here and must carry the alpha / rho the reference's OpenMP kernels compute (tests/golden/*.npz), the
    support vectors in the reference's order (grouped by class, label order of the data set), the header in the order of
    write_libsvm_model_header (include/plssvm/detail/io/libsvm_model_parsing.hpp:296-340) and `{:.10e}` numbers (:374-395).
  * `python -m plssvm_amd.predict` (src/main_predict.cpp) with the reference's LIBSVM-trained 500x200 models: the output file must EQUAL the
    reference's tests/data/predict/500x200.libsvm.predict byte for byte (tests/golden/predict/, data files of the reference's tests).
  * ARFF input through both command lines (data_set.hpp:494-505 picks the parser by extension for predict as well)."""

import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import HERE, ROOT

KERNEL_FLAG = {"linear": "0", "polynomial": "1", "rbf": "2"}
E10 = r"-?\d\.\d{10}e[+-]\d{2}"  # {:.10e}


def write_libsvm(path, X, y=None):
    """independent of the package's writer: full double precision, dense"""
    with open(path, "w") as f:
        for i in range(X.shape[0]):
            head = (f"{int(y[i])} " if y is not None else "")
            f.write(head + " ".join(f"{j + 1}:{float(X[i, j])!r}" for j in range(X.shape[1])) + "\n")


def parse_model(path):
    """header as ordered (key, value) pairs + the body as (alpha, {index: value}) -- independent of plssvm_amd.model"""
    lines = open(path).read().split("\n")
    assert lines[0].startswith("# This model file has been created at ")
    sv_at = lines.index("SV")
    header = [tuple(ln.split(" ", 1)) for ln in lines[1:sv_at]]
    body = []
    for ln in lines[sv_at + 1:]:
        if not ln:
            continue
        assert ln.endswith(" "), "every value is followed by a blank (libsvm_model_parsing.hpp:374-395)"
        toks = ln.split()
        assert re.fullmatch(E10, toks[0]), toks[0]
        feats = {}
        for t in toks[1:]:
            idx, val = t.split(":")
            assert re.fullmatch(E10, val), val
            feats[int(idx)] = float(val)
        body.append((float(toks[0]), feats))
    return header, body


def run_cli(module, args, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT)
    return subprocess.run([sys.executable, "-m", module, *args], capture_output=True, text=True, env=env, cwd=cwd, timeout=600)


def check_model_against_golden(path, kernel, X, y, alpha_want, rho_want, gamma):
    header, body = parse_model(path)
    keys = [k for k, _ in header]
    want_keys = ["svm_type", "kernel_type"] + {"linear": [], "polynomial": ["degree", "gamma", "coef0"], "rbf": ["gamma"]}[kernel] + ["nr_class", "label", "total_sv", "nr_sv", "rho"]
    assert keys == want_keys, keys
    h = dict(header)
    labels = sorted(set(int(v) for v in y))
    order = [i for lab in labels for i in range(len(y)) if int(y[i]) == lab]  # grouped by class, label order of the data set (:355-372)
    assert h["svm_type"] == "c_svc" and h["kernel_type"] == kernel and h["nr_class"] == "2"
    assert h["label"] == " ".join(str(v) for v in labels) and h["total_sv"] == str(len(y))
    assert h["nr_sv"] == " ".join(str(sum(1 for v in y if int(v) == lab)) for lab in labels)
    if kernel != "linear":
        assert float(h["gamma"]) == gamma
    assert len(body) == len(y)
    alpha_file = np.array([a for a, _ in body])
    scale = np.max(np.abs(alpha_want))
    assert np.max(np.abs(alpha_file - alpha_want[order])) <= 1e-6 * scale + 1e-10 * scale  # the fp64 parity bar + what {:.10e} drops
    assert abs(float(h["rho"]) - rho_want) <= 1e-6 * max(1.0, abs(rho_want))
    assert repr(float(h["rho"])) == h["rho"] or h["rho"] == str(int(float(h["rho"])))     # fmt's {} of a double: shortest round trip
    for (_, feats), i in zip(body, order):
        for j in range(X.shape[1]):
            if X[i, j] != 0:
                assert abs(feats[j + 1] - X[i, j]) <= 1e-10 * abs(X[i, j]) + 1e-300
            else:
                assert (j + 1) not in feats  # zeros are not written


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["linear", "polynomial", "rbf"])
@pytest.mark.parametrize("name", ["5x4", "500x200"])
def test_train_cli_writes_the_reference_kernels_model(tmp_path, inputs, golden, name, kernel):
    X, y = inputs[name + "_X"], inputs[name + "_y"]
    key = f"{name}/{kernel}/f64/def/cg_tight"
    eps, max_iter = float(golden[key + "/eps"]), int(golden[key + "/max_iter"])
    data, model = tmp_path / "train.libsvm", tmp_path / "out.model"
    write_libsvm(data, X, y)
    out = run_cli("plssvm_amd.train", ["-t", KERNEL_FLAG[kernel], "-e", repr(eps), "-i", str(max_iter), str(data), str(model)], tmp_path)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"#iter = (\d+)", out.stdout + out.stderr)
    assert m is None or abs(int(m.group(1)) - int(golden[key + "/iterations"])) <= 3
    check_model_against_golden(model, kernel, X, y, golden[key + "/alpha"], float(golden[key + "/rho"]), 1.0 / X.shape[1])


@pytest.mark.gpu
def test_train_cli_on_baseline_config0_file(tmp_path):
    """BASELINE configs[0]: the 500 x 4 LIBSVM text of tests/golden/c1_500x4.npz (generate_data.py's shape), linear kernel, fp64, through
    `plssvm-train`: the written model carries the alpha / rho the reference's kernels computed from the same file."""
    fx = np.load(os.path.join(HERE, "golden", "c1_500x4.npz"))
    data, model = tmp_path / "c1.libsvm", tmp_path / "c1.model"
    data.write_bytes(fx["libsvm_text"].tobytes())
    out = run_cli("plssvm_amd.train", ["-t", "0", "-e", "1e-10", str(data), str(model)], tmp_path)
    assert out.returncode == 0, out.stdout + out.stderr
    check_model_against_golden(model, "linear", fx["X"], fx["y"], fx["tight/alpha"], float(fx["tight/rho"]), 0.25)


def write_reference_style_model(path, kernel, fx):
    """the layout of the reference's LIBSVM-trained 500x200 models (tests/data/predict/500x200_rbf.libsvm.model): total_sv before rho,
    labels in file order, plain {} numbers"""
    sv, alpha, labels = fx[kernel + "_sv"], fx[kernel + "_alpha"], fx[kernel + "_labels"]
    first = []
    for lab in labels:
        if int(lab) not in first:
            first.append(int(lab))
    with open(path, "w") as f:
        f.write(f"svm_type c_svc\nkernel_type {kernel}\n")
        if kernel == "polynomial":
            f.write(f"degree {int(fx[kernel + '_degree'])}\n")
        if kernel != "linear":
            f.write(f"gamma {float(fx[kernel + '_gamma'])!r}\n")
        if kernel == "polynomial":
            f.write(f"coef0 {float(fx[kernel + '_coef0'])!r}\n")
        f.write(f"nr_class 2\ntotal_sv {len(alpha)}\nrho {float(fx[kernel + '_rho'])!r}\nlabel {' '.join(str(v) for v in first)}\n")
        f.write("nr_sv " + " ".join(str(int(np.sum(labels == lab))) for lab in first) + "\nSV\n")
        for a, row in zip(alpha, sv):
            f.write(f"{float(a)!r} " + " ".join(f"{j + 1}:{float(v)!r}" for j, v in enumerate(row) if v != 0) + "\n")


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["linear", "polynomial", "rbf"])
@pytest.mark.parametrize("real", ["double", "float"])
def test_predict_cli_reproduces_the_reference_predict_file(tmp_path, kernel, real):
    fx = np.load(os.path.join(HERE, "golden", "predict_500x200.npz"))
    want = open(os.path.join(HERE, "golden", "predict", "500x200.libsvm.predict"), "rb").read()
    assert [int(v) for v in want.split()] == [int(v) for v in fx["expected"]]  # the fixture array IS the reference's file
    test = tmp_path / "test.libsvm"
    write_libsvm(test, fx["test_X"], fx["test_y"])
    ref_model = os.path.join(HERE, "golden", "predict", f"500x200_{kernel}.libsvm.model")
    if os.path.isfile(ref_model):   # linear, polynomial: the reference's own model file
        model = ref_model
    else:                           # rbf (600 KB as text): written again from the fixture arrays in the same layout
        model = tmp_path / "rbf.model"
        write_reference_style_model(model, kernel, fx)
    outf = tmp_path / "out.predict"
    extra = ["--use_float_as_real_type"] if real == "float" else []
    out = run_cli("plssvm_amd.predict", [*extra, str(test), str(model), str(outf)], tmp_path)
    assert out.returncode == 0, out.stdout + out.stderr
    # line for line the reference's file; main_predict.cpp:53 writes fmt::join(labels, "\n"), i.e. no newline behind the last label
    # (the reference's committed file ends with one)
    got = outf.read_bytes()
    assert not got.endswith(b"\n") and got.split(b"\n") == want.rstrip(b"\n").split(b"\n")
    assert re.search(r"Accuracy = 100(\.0+)?% \(500/500\) \(classification\)", out.stdout), out.stdout  # main_predict.cpp:75-79


@pytest.mark.gpu
def test_train_and_predict_cli_read_arff_by_extension(tmp_path):
    """ARFF files through plssvm-train AND plssvm-predict (round 2 routed only the training side through the ARFF parser)."""
    from plssvm_amd.datagen import make_blobs_pm1
    X, y = make_blobs_pm1(300, 6, seed=5, dtype=np.float64)

    def write_arff(path, Xp, yp):
        with open(path, "w") as f:
            f.write("@RELATION blobs\n" + "".join(f"@ATTRIBUTE f{j} NUMERIC\n" for j in range(Xp.shape[1])) + "@ATTRIBUTE class {-1,1}\n@DATA\n")
            for row, lab in zip(Xp, yp):
                f.write(",".join(repr(float(v)) for v in row) + f",{int(lab)}\n")

    tr, te, model, outf = tmp_path / "train.arff", tmp_path / "test.arff", tmp_path / "m.model", tmp_path / "o.predict"
    write_arff(tr, X[:200], y[:200])
    write_arff(te, X[200:], y[200:])
    out = run_cli("plssvm_amd.train", ["-t", "2", "-e", "1e-8", str(tr), str(model)], tmp_path)
    assert out.returncode == 0, out.stdout + out.stderr
    out = run_cli("plssvm_amd.predict", [str(te), str(model), str(outf)], tmp_path)
    assert out.returncode == 0 and "Accuracy = " in out.stdout, out.stdout + out.stderr
    pred = np.array([int(v) for v in outf.read_text().split()])
    assert pred.shape == (100,) and np.mean(pred == y[200:]) > 0.97
