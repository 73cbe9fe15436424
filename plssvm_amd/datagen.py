"""Seeded synthetic classification data -- the counterpart of the reference's ``utility_scripts/generate_data.py``.

The reference script (generate_data.py:57-59, :76, :78, :100-108) draws ``make_blobs(n, d, centers=2)``, maps the labels to
+-1, min-max scales every feature to [-1, 1] and writes a one-based LIBSVM file; it is unseeded and imports a module
(`humanize`) that is not available here.  This module reproduces the recipe with numpy only and an explicit seed:
two Gaussian blobs with centres ~ U(-10, 10)^d (sklearn's default ``center_box``), sigma = 1, alternating labels
(balanced classes), per-feature min-max scaling to [-1, 1].
"""

from __future__ import annotations

import numpy as np

__all__ = ["make_blobs_pm1", "generate_libsvm_file"]


def make_blobs_pm1(num_points: int, num_features: int, seed: int = 42, dtype=np.float32, chunk: int = 65536):
    """Return ``(X[num_points, num_features], y[num_points] in {-1,+1})``.

    Generated in row chunks so that the 1 000 000 x 128 configuration never needs more than the output plus one chunk
    of float64 temporaries.
    """
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-10.0, 10.0, size=(2, num_features))
    X = np.empty((num_points, num_features), dtype=dtype)
    y = np.empty(num_points, dtype=dtype)
    lo = np.full(num_features, np.inf)
    hi = np.full(num_features, -np.inf)
    for start in range(0, num_points, chunk):
        stop = min(start + chunk, num_points)
        cls = (np.arange(start, stop) & 1)
        block = rng.standard_normal(size=(stop - start, num_features)) + centres[cls]
        lo = np.minimum(lo, block.min(axis=0))
        hi = np.maximum(hi, block.max(axis=0))
        X[start:stop] = block.astype(dtype)
        y[start:stop] = (2 * cls - 1).astype(dtype)
    # per-feature min-max scaling to [-1, 1]  (sklearn.preprocessing.minmax_scale(feature_range=(-1, 1)))
    span = np.where(hi > lo, hi - lo, 1.0)
    scale = (2.0 / span).astype(np.float64)
    shift = (-1.0 - lo * scale).astype(np.float64)
    for start in range(0, num_points, chunk):
        stop = min(start + chunk, num_points)
        X[start:stop] = (X[start:stop].astype(np.float64) * scale + shift).astype(dtype)
    np.clip(X, -1.0, 1.0, out=X)
    return X, y


def generate_libsvm_file(filename, num_points: int, num_features: int, seed: int = 42) -> None:
    """Write the data set as a one-based LIBSVM file with a leading comment line (generate_data.py:100-108)."""
    from .io_libsvm import write_libsvm_data

    X, y = make_blobs_pm1(num_points, num_features, seed=seed, dtype=np.float64)
    write_libsvm_data(filename, X, labels=[int(v) for v in y], comment=f"synthetic blobs {num_points}x{num_features} seed={seed}")
