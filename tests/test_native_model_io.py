"""The library's multi-threaded model writer / model reader / LIBSVM data writer (csrc/model_io.hpp, ``lssvm_mi355_model_*``, ``lssvm_mi355_libsvm_write_*``)
against the per-line Python implementations that restate the reference's formats (plssvm_amd/model.py, plssvm_amd/io_libsvm.py):

  * writers: the SAME BYTES as the Python writers on the reference's fixtures and on random models (both real types, all kernels, numeric and string labels,
    sparse and empty rows, extreme exponents, non-finite values), for any thread count;
  * reader: the same ``Model`` as the Python parser wherever it accepts, and NEVER WIDER than it -- whatever the fast path accepts, the reference-exact parser
    accepts with the same result (named shapes it must decline + random damage).

Formats: /root/reference/include/plssvm/detail/io/libsvm_model_parsing.hpp:64-262 (reader), :296-342 (header), :371-499 (body),
/root/reference/include/plssvm/detail/io/libsvm_parsing.hpp:244-296 (data writer).  Host code only: no GPU needed."""

import os

import numpy as np
import pytest

from plssvm_amd import _capi
from plssvm_amd.data_set import DataSet
from plssvm_amd.exceptions import PlssvmError
from plssvm_amd.io_libsvm import parse_libsvm_data, write_libsvm_data
from plssvm_amd.model import Model, _load_native
from plssvm_amd.parameter import Parameter

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "predict")


def _body(path):
    """The file without its time-stamp line."""
    data = open(path, "rb").read()
    assert data.startswith(b"# This model file has been created at ")
    return data.split(b"\n", 1)[1]


def _random_model(rng, k, dtype):
    n = int(rng.integers(2, 60)) if k % 10 else int(rng.integers(3000, 5000))  # some models span several chunks and threads
    d = int(rng.integers(1, 40)) if k % 7 else int(rng.integers(200, 400))
    X = rng.standard_normal((n, d)) * 10.0 ** rng.integers(-12, 12, size=(n, 1))
    X[rng.random((n, d)) < rng.choice([0.0, 0.3, 0.9])] = 0.0
    if k % 5 == 0:
        X[int(rng.integers(0, n))] = 0.0  # a support vector without a single feature
    if k % 11 == 0:
        X[0, 0], X[1, d - 1] = np.inf, -np.inf
    if k % 13 == 0:
        X[n - 1, 0] = np.nan
    if k % 17 == 0:
        X[0, d - 1] = 5e-324 if dtype == np.float64 else 1e-45  # denormal
        X[1, 0] = np.finfo(dtype).max
    X = X.astype(dtype)
    kind = k % 4
    if kind == 0:
        names = [-1.0, 1.0]
    elif kind == 1:
        names = [0, 1]
    elif kind == 2:
        names = [2.5, -3.25]
    else:
        names = ["cat", "dog"]
    which = rng.integers(0, 2, size=n)
    which[:2] = [0, 1]
    labels = [names[i] for i in which]
    kernel = ("linear", "polynomial", "rbf")[k % 3]
    params = Parameter(kernel_type=kernel, degree=int(rng.integers(1, 5)), gamma=float(rng.random() + 0.01) if kernel != "linear" else None,
                       coef0=float(rng.standard_normal()), cost=1.0)
    data = DataSet(X, labels, real_type=dtype, label_type=str if kind == 3 else float)
    alpha = (rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8, size=n)).astype(dtype)
    return Model(params, data, alpha=alpha, rho=float(rng.standard_normal())), (str if kind == 3 else (int if kind == 1 else float))


def _same_model(a: Model, b: Model):
    return (a.params == b.params and a.data.data().dtype == b.data.data().dtype and np.array_equal(a.data.data(), b.data.data(), equal_nan=True)
            and np.array_equal(a.alpha, b.alpha, equal_nan=True) and a.data.labels() == b.data.labels() and a.rho == b.rho
            and a.data.different_labels() == b.data.different_labels())


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_native_model_writer_writes_the_python_writers_bytes(tmp_path, dtype):
    """200 random models + the reference's LIBSVM-trained fixtures, native against Python, and the native reader against the Python reader on each."""
    rng = np.random.default_rng(11 if dtype == np.float32 else 12)
    fast_reads = 0
    for k in range(100):
        model, label_type = _random_model(rng, k, dtype)
        fn, fp = tmp_path / "native.model", tmp_path / "python.model"
        model.save(fn)
        model.save(fp, use_native=False)
        assert _body(fn) == _body(fp), k
        slow = Model.load(fn, real_type=dtype, label_type=label_type, use_native=False)
        fast = _load_native(fn, np.dtype(dtype), label_type)
        # ("inf" / "nan" features: both readers take them.  A denormal and a value whose 11 digits round beyond the largest double: std::from_chars
        # reports both as out of range where Python's float() returns the denormal / inf -- the fast path declines, which it may)
        assert fast is not None or (k % 17 == 0 and dtype == np.float64), k
        fast_reads += fast is not None
        loaded = Model.load(fn, real_type=dtype, label_type=label_type)
        assert _same_model(loaded, slow), k
        # what was read is what was written: the SVs grouped by class, {:.10e} of a value parses back to within its 11 digits
        order = model.class_order().astype(np.int64)
        dl = loaded.num_features()  # (trailing features that are zero everywhere leave no trace in the file)
        if k % 17:  # (the largest finite value reads back as inf once cut to 11 digits)
            assert np.allclose(loaded.data.data(), model.data.data()[order][:, :dl], rtol=1e-10, atol=0, equal_nan=True) and not np.any(model.data.data()[:, dl:]), k
        assert np.allclose(loaded.alpha, model.alpha[order], rtol=1e-10, atol=0)
    assert fast_reads >= 94
    for name in ("500x200_linear.libsvm.model", "500x200_polynomial.libsvm.model", "500x200_rbf.libsvm.model"):
        path = os.path.join(FIX, name)
        if not os.path.isfile(path):
            continue
        slow = Model.load(path, real_type=dtype, use_native=False)
        assert _load_native(path, np.dtype(dtype), float) is not None, name
        assert _same_model(Model.load(path, real_type=dtype), slow), name
        slow.save(tmp_path / "n.model")
        slow.save(tmp_path / "p.model", use_native=False)
        assert _body(tmp_path / "n.model") == _body(tmp_path / "p.model"), name


def test_written_files_do_not_depend_on_the_thread_count(tmp_path):
    rng = np.random.default_rng(3)
    model, _ = _random_model(rng, 10, np.float32)  # k = 10: thousands of rows -> several chunks
    X, labels = model.data.data(), model.data.labels()
    written = {}
    try:
        for threads in (1, 2, 3, 8):
            _capi.check(_capi.lib.lssvm_mi355_set_io_threads(threads))
            model.save(tmp_path / f"m{threads}.model")
            write_libsvm_data(tmp_path / f"d{threads}.libsvm", X, labels=labels, comment="c")
            written[threads] = (_body(tmp_path / f"m{threads}.model"), open(tmp_path / f"d{threads}.libsvm", "rb").read())
            assert _same_model(Model.load(tmp_path / f"m{threads}.model", real_type=np.float32), Model.load(tmp_path / "m1.model", real_type=np.float32, use_native=False))
    finally:
        _capi.check(_capi.lib.lssvm_mi355_set_io_threads(0))
    assert all(v == written[1] for v in written.values())


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_native_data_writer_writes_the_python_writers_bytes(tmp_path, dtype):
    rng = np.random.default_rng(21)
    for k in range(60):
        n, d = (int(rng.integers(1, 50)), int(rng.integers(1, 30))) if k % 10 else (4000, 17)
        X = (rng.standard_normal((n, d)) * 10.0 ** rng.integers(-6, 6, size=(n, 1))).astype(dtype)
        X[rng.random((n, d)) < 0.4] = 0
        labels = [None, [int(v) for v in rng.integers(-3, 4, size=n)], [float(v) for v in rng.integers(-3, 4, size=n)], [float(v) / 4 for v in rng.integers(-9, 9, size=n)],
                  [("a", "bb", "ccc")[int(v)] for v in rng.integers(0, 3, size=n)]][k % 5]
        comment = None if k % 2 else "made for a test"
        fn, fp = tmp_path / "n.libsvm", tmp_path / "p.libsvm"
        write_libsvm_data(fn, X, labels=labels, comment=comment)
        write_libsvm_data(fp, X, labels=labels, comment=comment, use_native=False)
        assert open(fn, "rb").read() == open(fp, "rb").read(), k
        if np.any(X != 0) and labels is not None and not isinstance(labels[0], str):  # (without labels a point without features is an empty line: gone when read)
            Xr, lr = parse_libsvm_data(fn, dtype=dtype)
            assert np.allclose(Xr, X[:, : Xr.shape[1]], rtol=1e-10, atol=0) and lr == [float(v) for v in labels]
    # the reference's line shape, literally (libsvm_parsing.hpp:268-279: "{}:{:.10e} " per feature, "{} " for the label)
    write_libsvm_data(tmp_path / "lit.libsvm", np.array([[1.5, 0.0, -2.0], [0.0, 0.0, 0.0]], dtype=dtype), labels=[1, -1])
    assert open(tmp_path / "lit.libsvm").read() == "1 1:1.5000000000e+00 3:-2.0000000000e+00 \n-1 \n"


HEADER = "svm_type c_svc\nkernel_type rbf\ngamma 0.5\nnr_class 2\ntotal_sv 3\nrho 0.25\nlabel 1 -1\nnr_sv 2 1\nSV\n"
BODY = "1.0e+00 1:1.0e+00 3:2.0e+00 \n-5.0e-01 2:1.0e+00 \n2.5e-01 1:3.0e+00 2:-1.0e+00 \n"


def _both(path, label_type=float):
    fast = _load_native(path, np.dtype(np.float64), label_type)
    try:
        slow = Model.load(path, real_type=np.float64, label_type=label_type, use_native=False)
    except (PlssvmError, ValueError, IndexError) as e:  # (the Python parser may trip over what the reference's would also reject, in its own words)
        slow = e
    return fast, slow


def _fast_equals(fast, slow: Model):
    params, X, labels, alpha, rho = fast
    return params == slow.params and np.array_equal(X, slow.data.data(), equal_nan=True) and labels == slow.data.labels() and np.array_equal(alpha, slow.alpha, equal_nan=True) and rho == float(slow.rho)


@pytest.mark.parametrize("text", [
    HEADER.replace("label 1 -1", "label 1 1.0") + BODY,             # the same label twice once converted: the reference's "unique" error
    HEADER.replace("gamma 0.5", "gamma 0.5\ngamma 0.25") + BODY,    # a key twice (the reference takes the last)
    HEADER.replace("kernel_type rbf", "kernel_type 2") + BODY,      # the LIBSVM number instead of the name
    HEADER.replace("kernel_type rbf", "kernel_type rbf\ndegree 2") + BODY,   # a parameter the kernel does not use
    HEADER.replace("total_sv 3", "total_sv 4") + BODY,
    HEADER.replace("nr_sv 2 1", "nr_sv 1 1") + BODY,
    HEADER.replace("nr_class 2", "nr_class 3") + BODY,
    HEADER.replace("rho 0.25", "rho +0.25") + BODY,                 # converts for Python, not for std::from_chars
    HEADER.replace("rho 0.25", "rho 0.25x") + BODY,
    HEADER.replace("rho 0.25\n", "") + BODY,
    HEADER.replace("svm_type c_svc", "svm_type nu_svc") + BODY,
    HEADER.replace("SV\n", "") + BODY,
    HEADER.replace("gamma 0.5", "gamma_x 0.5") + BODY,              # a key that merely STARTS like one (the Python parser goes by prefix)
    HEADER + BODY.replace("2:1.0e+00", "2:1.0e+00 1:4.0"),          # indices out of order
    HEADER + BODY.replace("-5.0e-01 ", ""),                         # a line without alpha
    HEADER + BODY.replace("3:2.0e+00", "3:2.0e+00\t"),
    HEADER + BODY + "1.0 1:1\n",
    HEADER,
])
def test_native_model_reader_declines_what_it_cannot_vouch_for(tmp_path, text):
    f = tmp_path / "case.model"
    f.write_text(text)
    fast, _ = _both(f)
    assert fast is None, text
    # ... and the plain variations ARE taken: keys in another order and case, a comment, CRLF, blank lines, blanks around the values, no gamma (the default stays open)
    g = tmp_path / "plain.model"
    g.write_bytes(("# c\r\n\r\nRHO  0.25 \r\nNR_SV 2 1\r\nlabel cat dog\r\ntotal_sv 3\r\nnr_class 2\r\nKernel_Type RBF\r\nsvm_type C_SVC\r\n  SV\r\n" + BODY.replace("\n", "\r\n")).encode())
    fast, slow = _both(g, str)
    assert fast is not None and _fast_equals(fast, slow) and fast[0].gamma is None and fast[2] == ["cat", "cat", "dog"] and fast[1].tolist() == [[1, 0, 2], [0, 1, 0], [3, -1, 0]]


def test_native_model_reader_is_never_wider_under_random_damage(tmp_path):
    """Every single-character damage of a valid model file: if the fast path still accepts it, so does the reference-exact parser, with the same model."""
    rng = np.random.default_rng(9)
    base = "# made by a test\n" + HEADER.replace("kernel_type rbf\ngamma 0.5", "kernel_type polynomial\ndegree 3\ngamma 0.5\ncoef0 1.5") + BODY
    alphabet = " \t:#+-.e0159xSV_\nlr"
    accepted = 0
    for trial in range(600):
        pos = int(rng.integers(0, len(base)))
        kind = trial % 3
        ch = alphabet[int(rng.integers(0, len(alphabet)))]
        text = base[:pos] + ch + base[pos + 1:] if kind == 0 else (base[:pos] + ch + base[pos:] if kind == 1 else base[:pos] + base[pos + 1:])
        f = tmp_path / "damaged.model"
        f.write_text(text)
        for lt in (float, str):
            fast, slow = _both(f, lt)
            if fast is not None:
                accepted += 1
                assert not isinstance(slow, Exception) and _fast_equals(fast, slow), (trial, lt, text, slow)
    assert accepted > 40  # (damage inside the comment or a number often leaves a well-formed file)


def test_writer_errors_are_reported(tmp_path):
    X = np.ones((3, 2), dtype=np.float64)
    with pytest.raises(PlssvmError, match="can't write"):
        write_libsvm_data(tmp_path / "no_such_dir" / "x.libsvm", X)
    import ctypes as C
    bad_order = np.array([0, 7], dtype=np.uint64)
    st = _capi.lib.lssvm_mi355_model_write_f64(os.fsencode(tmp_path / "m"), b"SV\n", _capi.ptr(X), C.c_uint64(3), C.c_uint64(2), C.c_uint64(2), _capi.ptr(np.ones(3)), _capi.ptr(bad_order), C.c_uint64(2))
    assert st == -1 and "beyond" in _capi.last_error()
