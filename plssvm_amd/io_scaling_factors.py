"""Scaling-factor files of ``plssvm-scale -s / -r`` (include/plssvm/detail/io/scaling_factors_parsing.hpp:57-150): what lets a
test set be scaled with the factors of the training set.

    # comment lines are ignored
    x
    <lower> <upper>                 the interval every feature is scaled to
    <feature> <min> <max>           one line per feature, ONE-based index; features that are zero everywhere have no line
"""

from __future__ import annotations

import time

import numpy as np

from .exceptions import InvalidFileFormatError
from .io_libsvm import read_lines
from .string_conversion import real_prefix

__all__ = ["parse_scaling_factors", "write_scaling_factors"]


def _type_name(dtype) -> str:
    return "float" if np.dtype(dtype) == np.dtype(np.float32) else "double"


def _split_as_real(line: str, dtype):
    """detail::split_as<real_type> (string_utility / string_conversion): whitespace separated numbers of the real type."""
    out = []
    for token in line.split():
        value = real_prefix(token)  # the reference converts the longest valid prefix (string_conversion.py)
        if value is None:
            raise InvalidFileFormatError(f"Can't convert '{token}' to a value of type {_type_name(dtype)}!")
        out.append(np.dtype(dtype).type(value))
    return out


def parse_scaling_factors(filename, dtype=np.float64):
    """Returns ``((lower, upper), [(feature, lower, upper), ...])`` with ZERO-based feature indices
    (scaling_factors_parsing.hpp:57-119; the error texts are the reference's)."""
    lines = read_lines(filename, "#")
    if len(lines) < 2:
        raise InvalidFileFormatError(f"At least two lines must be present, but only {len(lines)} were given!")
    if lines[0].strip() != "x":
        raise InvalidFileFormatError(f"The first line must only contain an 'x', but is \"{lines[0]}\"!")
    interval = _split_as_real(lines[1], dtype)
    if len(interval) != 2:
        raise InvalidFileFormatError(f"The interval to which the data points should be scaled must exactly have two values, but {len(interval)} were given!")
    if interval[0] >= interval[1]:
        raise InvalidFileFormatError(f"Inconsistent scaling interval specification: lower ({_fmt(interval[0])}) must be less than upper ({_fmt(interval[1])})!")
    factors = []
    for line in lines[2:]:
        values = _split_as_real(line, dtype)
        if len(values) != 3:
            raise InvalidFileFormatError(f"Each line must contain exactly three values, but {len(values)} were given!")
        feature = int(values[0])
        if feature == 0:
            raise InvalidFileFormatError("The scaling factors must be provided one-based, but are zero-based!")
        factors.append((feature - 1, values[1], values[2]))
    return (interval[0], interval[1]), factors  # (numpy scalars of `dtype`)


def _fmt(v) -> str:
    """{fmt}'s default ``{}`` of a floating point value: shortest round-trip digits OF ITS TYPE, no trailing ``.0``."""
    r = str(v) if isinstance(v, np.floating) else repr(float(v))
    return r[:-2] if r.endswith(".0") else r


def write_scaling_factors(filename, interval, factors) -> None:
    """scaling_factors_parsing.hpp:131-150 (``factors``: zero-based feature indices, written one-based)."""
    lower, upper = interval
    if not lower < upper:
        raise InvalidFileFormatError(f"Illegal interval specification: lower ({_fmt(lower)}) < upper ({_fmt(upper)}).")
    with open(filename, "w") as f:
        f.write(f"# These scaling factors have been created at {time.strftime('%Y-%m-%d %H:%M:%S')}\n")
        f.write("x\n")
        f.write(f"{_fmt(lower)} {_fmt(upper)}\n")
        for feature, lo, hi in factors:
            f.write(f"{int(feature) + 1} {_fmt(lo)} {_fmt(hi)}\n")
