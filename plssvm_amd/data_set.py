"""``DataSet``: the data / label container on the input side of the hot path, mirroring ``plssvm::data_set``
(include/plssvm/data_set.hpp) as far as the path needs it: LIBSVM file or array input, the binary label mapping
(first label in sorted order -> -1, second -> +1; data_set.hpp:438-454, :653-667) and optional min-max scaling
(data_set.hpp:670-730)."""

from __future__ import annotations

import numpy as np

from .exceptions import InvalidParameterError, PlssvmError
from .io_libsvm import parse_libsvm_data, write_libsvm_data

__all__ = ["DataSet", "DataSetError", "LabelMapper"]


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


class DataSet:
    def __init__(self, data=None, labels=None, filename=None, real_type=np.float64, label_type=float, scaling=None):
        """Either ``filename`` (LIBSVM format) or ``data`` (+ optional ``labels``).  ``scaling=(lower, upper)`` scales every
        feature to that interval (data_set.hpp:670-730)."""
        self.real_type = np.dtype(real_type)
        if self.real_type not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise InvalidParameterError("real_type must be float32 or float64")
        if filename is not None:
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
        self.scaling_factors = None
        if labels is not None:
            self.mapping = LabelMapper(labels)
            self._y = np.array([self.mapping.mapped(lab) for lab in labels], dtype=self.real_type)
        if scaling is not None:
            self._scale(*scaling)

    def _scale(self, lower, upper):
        if lower >= upper:
            raise DataSetError(f"Inconsistent scaling interval specification: lower ({lower}) must be less than upper ({upper})!")
        mn, mx = self._X.min(axis=0), self._X.max(axis=0)
        span = np.where(mx > mn, mx - mn, 1.0)
        self._X = (lower + (upper - lower) * (self._X - mn) / span).astype(self.real_type)
        self.scaling_factors = (lower, upper, mn, mx)

    # accessors named after the reference's (data_set.hpp:150-230)
    def data(self):
        return self._X

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

    def save(self, filename):
        write_libsvm_data(filename, self._X, labels=self._labels)
