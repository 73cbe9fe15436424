"""``DataSet``: the data / label container on the input side of the hot path, mirroring ``plssvm::data_set``
(include/plssvm/data_set.hpp) as far as the path needs it: LIBSVM file or array input, the binary label mapping
(first label in sorted order -> -1, second -> +1; data_set.hpp:438-454, :653-667) and optional min-max scaling with
factors that can be saved and restored (``data_set::scaling``, data_set.hpp:290-375, :670-730; ``plssvm-scale -s / -r``)."""

from __future__ import annotations

import numpy as np

from .exceptions import InvalidParameterError, PlssvmError
from .io_arff import parse_arff_data, write_arff_data
from .io_libsvm import parse_libsvm_data, write_libsvm_data
from .io_scaling_factors import _fmt, parse_scaling_factors, write_scaling_factors

__all__ = ["DataSet", "DataSetError", "LabelMapper", "Scaling"]


class DataSetError(PlssvmError):
    """reference: ``plssvm::data_set_exception``."""


class LabelMapper:
    """``data_set::label_mapper`` (data_set.hpp:438-470): exactly two distinct labels; std::set order decides the sign."""

    def __init__(self, labels):
        unique = sorted(set(labels))
        if len(unique) != 2:
            raise DataSetError(f"Currently only binary classification is supported, but {len(unique)} different labels were given!")
        self.labels = unique  # [label mapped to -1, label mapped to +1]

    def mapped(self, label):
        if label == self.labels[0]:
            return -1.0
        if label == self.labels[1]:
            return 1.0
        raise DataSetError(f'Label "{label}" unknown in this label mapping!')

    def label_of(self, mapped_value):
        if mapped_value == -1:
            return self.labels[0]
        if mapped_value == 1:
            return self.labels[1]
        raise DataSetError(f'Mapped value "{mapped_value}" unknown in this label mapping!')


def _file_format(name) -> str:
    """file_format_type (file_format_types.hpp): "libsvm" or "arff", case-insensitive."""
    fmt = str(name).lower()
    if fmt not in ("libsvm", "arff"):
        raise InvalidParameterError(f'The file format must be one of "libsvm" or "arff", but is "{name}"!')
    return fmt


class Scaling:
    """``data_set::scaling`` (data_set.hpp:290-375): the interval ``[lower, upper]`` plus, per feature that is not zero everywhere,
    its ``(feature, min, max)`` -- computed by the first data set that is scaled with this object, or restored from a file written
    by an earlier run (``plssvm-scale -s`` / ``-r``)."""

    def __init__(self, lower=None, upper=None, filename=None, real_type=np.float64):
        if filename is not None:
            (self.lower, self.upper), self.factors = parse_scaling_factors(filename, real_type)
        else:
            if lower >= upper:
                raise DataSetError(f"Inconsistent scaling interval specification: lower ({_fmt(lower)}) must be less than upper ({_fmt(upper)})!")
            self.lower, self.upper, self.factors = lower, upper, []

    def save(self, filename) -> None:
        write_scaling_factors(filename, (self.lower, self.upper), self.factors)


class DataSet:
    def __init__(self, data=None, labels=None, filename=None, real_type=np.float64, label_type=float, scaling=None, file_format=None):
        """Either ``filename`` or ``data`` (+ optional ``labels``).  A file is read with the ARFF parser if ``file_format`` is "arff" or,
        without ``file_format``, if its name ends with ``.arff``; otherwise with the LIBSVM parser (data_set.hpp:494-505).
        ``scaling``: ``(lower, upper)`` or a
        :class:`Scaling` (possibly restored from a file) -- every feature is scaled to that interval (data_set.hpp:670-730)."""
        self.real_type = np.dtype(real_type)
        if self.real_type not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise InvalidParameterError("real_type must be float32 or float64")
        if filename is not None:
            fmt = _file_format(file_format) if file_format is not None else ("arff" if str(filename).endswith(".arff") else "libsvm")
            if fmt == "arff":
                X, labels = parse_arff_data(filename, dtype=self.real_type, label_type=label_type)
            else:
                X, labels = parse_libsvm_data(filename, dtype=self.real_type, label_type=label_type)
        else:
            X = np.ascontiguousarray(np.asarray(data), dtype=self.real_type)
            if X.ndim != 2 or X.shape[0] == 0:
                raise DataSetError("Data vector is empty!")
            if X.shape[1] == 0:
                raise DataSetError("No features provided for the data points!")
            if labels is not None:
                labels = list(labels.tolist() if isinstance(labels, np.ndarray) else labels)
                if len(labels) != X.shape[0]:
                    raise DataSetError(f"Number of labels ({len(labels)}) must match the number of data points ({X.shape[0]})!")
        self._X = X
        self._labels = labels
        self.mapping = None
        self._y = None
        self.scaling_factors_ = None
        if labels is not None:
            self.mapping = LabelMapper(labels)
            # (vectorised: a million labels through LabelMapper.mapped one by one took a second of a train run whose solve takes 7)
            lab = np.asarray(labels)
            self._y = np.where(lab == self.mapping.labels[1], 1.0, -1.0).astype(self.real_type)
        if scaling is not None:
            self._scaling = scaling if isinstance(scaling, Scaling) else Scaling(*scaling)
            self._scale()
            self.scaling_factors_ = self._scaling

    def _scale(self):
        """data_set::scale (data_set.hpp:670-730), statement for statement in its semantics: factors are computed unless they were
        restored; a feature that is zero everywhere gets no factor and stays zero; a feature that is constant but not zero divides
        by zero exactly as the reference does."""
        sc = self._scaling
        rt = self.real_type.type
        lower, upper = rt(sc.lower), rt(sc.upper)
        nfeat = self._X.shape[1]
        if not sc.factors:
            mn, mx = self._X.min(axis=0), self._X.max(axis=0)
            sc.factors = [(f, mn[f], mx[f]) for f in range(nfeat) if not (mn[f] == 0 and mx[f] == 0)]
        else:
            if len(sc.factors) > nfeat:
                raise DataSetError(f"Need at most as much scaling factors as features in the data set are present ({nfeat}), but {len(sc.factors)} were given!")
            sc.factors = sorted(sc.factors, key=lambda t: t[0])
            if sc.factors[-1][0] >= nfeat:
                raise DataSetError(f"The maximum scaling feature index most not be greater than {nfeat - 1}, but is {sc.factors[-1][0]}!")
            for a, b in zip(sc.factors, sc.factors[1:]):
                if a[0] == b[0]:
                    raise DataSetError(f"Found more than one scaling factor for the feature index {a[0]}!")
        X = self._X.copy()
        with np.errstate(divide="ignore", invalid="ignore"):
            for f, lo, hi in sc.factors:
                lo, hi = rt(lo), rt(hi)
                X[:, f] = lower + (upper - lower) * (X[:, f] - lo) / (hi - lo)
        self._X = X

    # accessors named after the reference's (data_set.hpp:150-230)
    def data(self):
        return self._X

    def is_scaled(self):
        return self.scaling_factors_ is not None

    def scaling_factors(self):
        """The :class:`Scaling` this data set was scaled with (``data_set::scaling_factors``), or None."""
        return self.scaling_factors_

    def has_labels(self):
        return self._labels is not None

    def labels(self):
        return self._labels

    def different_labels(self):
        return None if self.mapping is None else list(self.mapping.labels)

    def num_data_points(self):
        return int(self._X.shape[0])

    def num_features(self):
        return int(self._X.shape[1])

    def num_different_labels(self):
        return 0 if self.mapping is None else 2

    def mapped_labels(self):
        """y in {-1, +1} (data_set::y_ptr_)."""
        return self._y

    def save(self, filename, file_format=None):
        """data_set::save (data_set.hpp:580-622): the format is given or follows the extension (.libsvm / .arff)."""
        name = str(filename)
        if file_format is None:
            if name.endswith(".libsvm"):
                file_format = "libsvm"
            elif name.endswith(".arff"):
                file_format = "arff"
            else:
                raise DataSetError(f'Unrecognized file extension for file "{name}" (must be one of: .libsvm or .arff)!')
        if _file_format(file_format) == "arff":
            write_arff_data(filename, self._X, labels=self._labels)
        else:
            write_libsvm_data(filename, self._X, labels=self._labels)
