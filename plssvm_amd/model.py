"""``Model`` + LIBSVM-compatible model files (include/plssvm/model.hpp, include/plssvm/detail/io/libsvm_model_parsing.hpp).

Writer (libsvm_model_parsing.hpp:296-342 header, :371-499 body): a ``#`` timestamp line, ``svm_type c_svc``, ``kernel_type``,
then ``degree/gamma/coef0`` (polynomial) or ``gamma`` (rbf), ``nr_class``, ``label``, ``total_sv``, ``nr_sv``, ``rho``, ``SV`` and
one line per training point ``alpha idx:val ...`` formatted ``{:.10e}``, zeros omitted, points grouped by class in label order.
Reader (:64-262): key-prefix based and order agnostic; labels are expanded per support vector from ``label`` + ``nr_sv``."""

from __future__ import annotations

import datetime

import numpy as np

from .data_set import DataSet
from .exceptions import InvalidFileFormatError
from .io_libsvm import read_lines
from .parameter import KernelFunctionType, Parameter, kernel_function_type_from_string

__all__ = ["Model"]


def _fmt_label(lab):
    if isinstance(lab, str):
        return lab
    f = float(lab)
    return str(int(f)) if f.is_integer() else repr(f)


def _capi_or_none():
    try:
        from . import _capi
    except (ImportError, OSError):
        return None
    return _capi


def _save_native(filename, header, X, alpha, order) -> bool:
    capi = _capi_or_none()
    if capi is None or X.dtype not in (np.dtype(np.float32), np.dtype(np.float64)) or not X.flags.c_contiguous:
        return False
    import ctypes as C
    import os

    write = capi.lib.lssvm_mi355_model_write_f32 if X.dtype == np.float32 else capi.lib.lssvm_mi355_model_write_f64
    capi.check(write(os.fsencode(filename), header.encode(), capi.ptr(X), C.c_uint64(X.shape[0]), C.c_uint64(X.shape[1]), C.c_uint64(X.shape[1]),
                     capi.ptr(alpha), capi.ptr(order), C.c_uint64(order.size)))
    return True


def _load_native(filename, dtype, label_type):
    """``(params, X, per_sv_labels, alpha, rho)`` from the library's reader, or None where it declines (the Python parser then decides)."""
    capi = _capi_or_none()
    if capi is None or dtype not in (np.dtype(np.float32), np.dtype(np.float64)) or label_type not in (float, int, str):
        return None
    import ctypes as C
    import os

    lib = capi.lib
    handle, info = C.c_void_p(), capi.LssvmModelInfo()
    if lib.lssvm_mi355_model_open(os.fsencode(filename), C.byref(handle), C.byref(info)) != 0:
        return None
    try:
        text = C.create_string_buffer(int(info.label_text_bytes))
        nr_sv = (C.c_uint64 * int(info.nr_class))()
        if lib.lssvm_mi355_model_labels(handle, text, C.c_uint64(info.label_text_bytes), nr_sv) != 0:
            return None
        toks = text.value.decode("ascii").split(" ")
        try:
            labels = toks if label_type is str else [label_type(float(t)) for t in toks]
        except (ValueError, OverflowError):
            return None
        if len(set(labels)) != len(labels):  # duplicates AFTER the conversion ("1" and "1.0"): the parser below words the error
            return None
        X = np.empty((int(info.total_sv), int(info.num_features)), dtype=dtype)
        alpha = np.empty(int(info.total_sv), dtype=dtype)
        fill = lib.lssvm_mi355_model_fill_f32 if dtype == np.float32 else lib.lssvm_mi355_model_fill_f64
        if fill(handle, capi.ptr(X), C.c_uint64(X.shape[1]), capi.ptr(alpha)) != 0:
            return None
    finally:
        lib.lssvm_mi355_model_close(handle)
    try:
        params = Parameter(kernel_type=KernelFunctionType(int(info.kernel_type)), degree=int(info.degree) if info.has_degree else 3,
                           gamma=float(info.gamma) if info.has_gamma else None, coef0=float(info.coef0) if info.has_coef0 else 0.0)
    except Exception:
        return None
    per_sv_labels = [lab for lab, cnt in zip(labels, list(nr_sv)) for _ in range(int(cnt))]
    return params, X, per_sv_labels, alpha, float(info.rho)


class Model:
    def __init__(self, params: Parameter, data: DataSet, alpha=None, rho=0.0):
        self.params = params
        self.data = data
        self.alpha = np.zeros(data.num_data_points(), dtype=data.real_type) if alpha is None else np.asarray(alpha, dtype=data.real_type)
        self.rho = data.real_type.type(rho)
        self.w = None  # linear kernel: cached normal vector (model.hpp:166)

    # accessors named after the reference's (model.hpp:60-135)
    def num_support_vectors(self):
        return self.data.num_data_points()

    def num_features(self):
        return self.data.num_features()

    def get_params(self):
        return self.params

    def support_vectors(self):
        return self.data.data()

    def labels(self):
        return self.data.labels()

    def different_labels(self):
        return self.data.different_labels()

    def weights(self):
        return self.alpha

    # ------------------------------------------------------------------ save
    def header_text(self) -> str:
        """The lines from the time stamp to ``SV`` (libsvm_model_parsing.hpp:296-342, :381)."""
        p = self.params
        labels = self.data.labels()
        order = self.data.different_labels()
        lab = np.asarray(labels)
        counts = [int(np.count_nonzero(lab == o)) for o in order]
        text = f"# This model file has been created at {datetime.datetime.now():%Y-%m-%d %H:%M:%S}\nsvm_type c_svc\nkernel_type {p.kernel_type}\n"
        if p.kernel_type == KernelFunctionType.POLYNOMIAL:
            text += f"degree {p.degree}\ngamma {p.gamma!r}\ncoef0 {p.coef0!r}\n"
        elif p.kernel_type == KernelFunctionType.RBF:
            text += f"gamma {p.gamma!r}\n"
        text += (f"nr_class 2\nlabel {' '.join(_fmt_label(o) for o in order)}\ntotal_sv {self.data.num_data_points()}\n"
                 f"nr_sv {' '.join(str(c) for c in counts)}\nrho {float(self.rho)!r}\nSV\n")
        return text

    def class_order(self) -> np.ndarray:
        """Row indices grouped by class in the order of the ``label`` line, ascending inside a class (libsvm_model_parsing.hpp:416-499)."""
        lab = np.asarray(self.data.labels())
        return np.concatenate([np.flatnonzero(lab == o) for o in self.data.different_labels()]).astype(np.uint64)

    def save(self, filename, use_native: bool = True) -> None:
        """write_libsvm_model_data (libsvm_model_parsing.hpp:371-499).  The body is formatted by the library's multi-threaded writer
        (csrc/model_io.hpp, ``lssvm_mi355_model_write_*``); the Python loop below writes the same bytes and is what is left when the library is not built."""
        header, order = self.header_text(), self.class_order()
        X = self.data.data()
        alpha = np.ascontiguousarray(self.alpha, dtype=X.dtype)
        if use_native and _save_native(filename, header, X, alpha, order):
            return
        with open(filename, "w") as f:
            f.write(header)
            for i in order.tolist():
                f.write(f"{float(alpha[i]):.10e} " + "".join(f"{j + 1}:{float(v):.10e} " for j, v in enumerate(X[i]) if v != 0) + "\n")

    # ------------------------------------------------------------------ load
    @classmethod
    def load(cls, filename, real_type=np.float64, label_type=float, use_native: bool = True) -> "Model":
        """parse_libsvm_model_header + the support vectors (libsvm_model_parsing.hpp:64-262, model.hpp:170-200).  Well-formed files are read by the
        library's multi-threaded reader (``lssvm_mi355_model_open / _fill / _close``); whatever that declines is parsed -- and, where it is wrong, reported
        in the reference's words -- by the code below."""
        if use_native:
            fast = _load_native(filename, np.dtype(real_type), label_type)
            if fast is not None:
                params, X, per_sv_labels, alphas, rho = fast
                return cls(params, DataSet(X, per_sv_labels, real_type=real_type), alpha=alphas, rho=rho)
        lines = read_lines(filename)
        kernel = degree = gamma = coef0 = None
        svm_type_set = False
        nr_class = total_sv = rho = labels = nr_sv = None
        header_line = None
        for idx, raw in enumerate(lines):
            line = raw.strip()
            low = line.lower()
            value = low.split(" ", 1)[1].strip() if " " in low else ""
            if low.startswith("svm_type"):
                if value != "c_svc":
                    raise InvalidFileFormatError(f"Can only use c_svc as svm_type, but '{value}' was given!")
                svm_type_set = True
            elif low.startswith("kernel_type"):
                try:
                    kernel = kernel_function_type_from_string(value)
                except Exception:
                    raise InvalidFileFormatError(f"Unrecognized kernel type '{value}'!") from None
            elif low.startswith("gamma"):
                gamma = float(value)
            elif low.startswith("degree"):
                degree = int(value)
            elif low.startswith("coef0"):
                coef0 = float(value)
            elif low.startswith("nr_class"):
                nr_class = int(value)
            elif low.startswith("total_sv"):
                total_sv = int(value)
                if total_sv == 0:
                    raise InvalidFileFormatError("The number of support vectors must be greater than 0!")
            elif low.startswith("rho"):
                rho = float(value)
            elif low.startswith("label"):
                toks = line.split(" ", 1)[1].split()
                labels = toks if label_type is str else [label_type(float(t)) for t in toks]
                if len(labels) < 2:
                    raise InvalidFileFormatError(f"At least two labels must be set, but only {len(labels)} label ([{', '.join(toks)}]) was given!")
                if len(set(labels)) != len(labels):
                    raise InvalidFileFormatError(f"Provided {len(labels)} labels but only {len(set(labels))} of them was/where unique!")
            elif low.startswith("nr_sv"):
                nr_sv = [int(t) for t in value.split()]
                if len(nr_sv) < 2:
                    raise InvalidFileFormatError(f"At least two nr_sv must be set, but only {len(nr_sv)} ([{', '.join(map(str, nr_sv))}]) was given!")
            elif low == "sv":
                header_line = idx
                break
            else:
                raise InvalidFileFormatError(f"Unrecognized header entry '{raw}'! Maybe SV is missing?")
        if not svm_type_set:
            raise InvalidFileFormatError("Missing svm_type!")
        if kernel is None:
            raise InvalidFileFormatError("Missing kernel_type!")
        if kernel == KernelFunctionType.LINEAR:
            for name, v in (("degree", degree), ("gamma", gamma), ("coef0", coef0)):
                if v is not None:
                    raise InvalidFileFormatError(f"Explicitly provided a value for the {name} parameter which is not used in the linear kernel!")
        if kernel == KernelFunctionType.RBF:
            for name, v in (("degree", degree), ("coef0", coef0)):
                if v is not None:
                    raise InvalidFileFormatError(f"Explicitly provided a value for the {name} parameter which is not used in the radial basis function kernel!")
        if nr_class is None:
            raise InvalidFileFormatError("Missing number of different classes nr_class!")
        if total_sv is None:
            raise InvalidFileFormatError("Missing total number of support vectors total_sv!")
        if rho is None:
            raise InvalidFileFormatError("Missing rho value!")
        if labels is None:
            raise InvalidFileFormatError("Missing class label specification!")
        if nr_class != len(labels):
            raise InvalidFileFormatError(f"The number of classes (nr_class) is {nr_class}, but the provided number of different labels is {len(labels)} (label)!")
        if nr_sv is None:
            raise InvalidFileFormatError("Missing number of support vectors per class nr_sv!")
        if nr_class != len(nr_sv):
            raise InvalidFileFormatError(f"The number of classes (nr_class) is {nr_class}, but the provided number of different labels is {len(nr_sv)} (nr_sv)!")
        if sum(nr_sv) != total_sv:
            raise InvalidFileFormatError(f"The total number of support vectors is {total_sv}, but the sum of nr_sv is {sum(nr_sv)}!")
        if header_line is None or header_line + 1 >= len(lines):
            raise InvalidFileFormatError("Can't parse file: no support vectors are given or SV is missing!")

        # body: "alpha idx:val ..." == a LIBSVM data file whose "label" column is alpha (model.hpp:187)
        rows = lines[header_line + 1:]
        alphas = np.zeros(len(rows), dtype=real_type)
        parsed, nfeat = [], 0
        for i, raw in enumerate(rows):
            toks = raw.split()
            alphas[i] = float(toks[0])
            feats = []
            for t in toks[1:]:
                j, v = t.split(":", 1)
                feats.append((int(j) - 1, float(v)))
            nfeat = max(nfeat, max((j for j, _ in feats), default=-1) + 1)
            parsed.append(feats)
        if len(rows) != total_sv:
            raise InvalidFileFormatError(f"The total number of support vectors is {total_sv}, but {len(rows)} were found!")
        X = np.zeros((len(rows), nfeat), dtype=real_type)
        for i, feats in enumerate(parsed):
            for j, v in feats:
                X[i, j] = v
        per_sv_labels = [lab for lab, cnt in zip(labels, nr_sv) for _ in range(cnt)]
        params = Parameter(kernel_type=kernel, degree=degree if degree is not None else 3, gamma=gamma, coef0=coef0 if coef0 is not None else 0.0)
        data = DataSet(X, per_sv_labels, real_type=real_type)
        # the model file fixes the label ORDER (first label <-> first nr_sv block); keep it for writing, the sign mapping stays sorted
        return cls(params, data, alpha=alphas, rho=rho)
