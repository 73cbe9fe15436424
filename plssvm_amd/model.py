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
    def save(self, filename) -> None:
        p = self.params
        X = self.data.data()
        labels = self.data.labels()
        order = self.data.different_labels()
        counts = [sum(1 for lab in labels if lab == o) for o in order]
        with open(filename, "w") as f:
            f.write(f"# This model file has been created at {datetime.datetime.now():%Y-%m-%d %H:%M:%S}\n")
            f.write(f"svm_type c_svc\nkernel_type {p.kernel_type}\n")
            if p.kernel_type == KernelFunctionType.POLYNOMIAL:
                f.write(f"degree {p.degree}\ngamma {p.gamma!r}\ncoef0 {p.coef0!r}\n")
            elif p.kernel_type == KernelFunctionType.RBF:
                f.write(f"gamma {p.gamma!r}\n")
            f.write(f"nr_class 2\nlabel {' '.join(_fmt_label(o) for o in order)}\ntotal_sv {X.shape[0]}\n"
                    f"nr_sv {' '.join(str(c) for c in counts)}\nrho {float(self.rho)!r}\nSV\n")
            for o in order:
                for i in range(X.shape[0]):
                    if labels[i] == o:
                        feats = " ".join(f"{j + 1}:{float(v):.10e}" for j, v in enumerate(X[i]) if v != 0)
                        f.write(f"{float(self.alpha[i]):.10e} {feats} \n")

    # ------------------------------------------------------------------ load
    @classmethod
    def load(cls, filename, real_type=np.float64, label_type=float) -> "Model":
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
