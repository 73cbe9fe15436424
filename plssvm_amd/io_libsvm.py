"""LIBSVM data-file reader / writer (the on-disk format on the input side of the hot path).

Follows the reference's rules (citations relative to /root/reference):
  * lines are split at ``\\r`` / ``\\n``, left-trimmed, and dropped when empty or starting with ``#``
    (src/plssvm/detail/io/file_reader.cpp:179-205);
  * number of features = largest index found on any line (include/plssvm/detail/io/libsvm_parsing.hpp:47-95);
  * one-based, strictly increasing feature indices, dense zero fill, labels either on every line or on none
    (libsvm_parsing.hpp:118-229); the error messages are the reference's.
"""

from __future__ import annotations

import os

import numpy as np

from .exceptions import FileNotFoundPlssvmError, InvalidFileFormatError
from .string_conversion import real_prefix

__all__ = ["read_lines", "parse_libsvm_data", "write_libsvm_data"]


def read_lines(filename, comment: str = "#") -> list[str]:
    """file_reader::read_lines (file_reader.cpp:179-205)."""
    if not os.path.isfile(filename):
        raise FileNotFoundPlssvmError(f"Couldn't find file: '{filename}'!")
    with open(filename, "rb") as f:
        content = f.read().decode("utf-8", errors="replace")
    lines = []
    for raw in content.replace("\r", "\n").split("\n"):
        sv = raw.lstrip(" \t\v\f")
        if sv and not sv.startswith(comment):
            lines.append(sv)
    return lines


def _to_float(token: str, what: str) -> float:
    value = real_prefix(token)  # the reference converts the longest valid prefix (string_conversion.py)
    if value is None:
        raise InvalidFileFormatError(f"Can't convert '{token}' to a value of type {what}!")
    return value


def _to_index(token: str) -> int:
    t = token.strip()
    if not t or not t.isdigit():
        raise InvalidFileFormatError(f"Can't convert '{token}' to a value of type unsigned long!")
    return int(t)


def _parse_native(filename, dtype, skipped_lines, label_type):
    """Fast path: libplssvm_amd's multi-threaded reader for well-formed files (csrc/libsvm_reader.hpp).  Returns None when the
    library is not built or the file is anything but plainly well formed -- the Python parser below then decides (and words
    the error exactly like the reference)."""
    dtype = np.dtype(dtype)
    if label_type not in (float, int) or dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
        return None
    try:
        import ctypes as C

        from . import _capi
    except (ImportError, OSError):
        return None
    lib = _capi.lib
    handle = C.c_void_p()
    npts, nfeat, has_label = C.c_uint64(), C.c_uint64(), C.c_int()
    if lib.lssvm_mi355_libsvm_open(os.fsencode(filename), C.c_uint64(skipped_lines), C.byref(handle), C.byref(npts), C.byref(nfeat), C.byref(has_label)) != 0:
        return None
    try:
        X = np.empty((npts.value, nfeat.value), dtype=dtype)
        y = np.empty(npts.value, dtype=np.float64) if has_label.value else None
        fill = lib.lssvm_mi355_libsvm_fill_f32 if dtype == np.float32 else lib.lssvm_mi355_libsvm_fill_f64
        if fill(handle, _capi.ptr(X), C.c_uint64(nfeat.value), _capi.ptr(y) if y is not None else None) != 0:
            return None
    finally:
        lib.lssvm_mi355_libsvm_close(handle)
    labels = None
    if y is not None:
        labels = [label_type(v) for v in y.tolist()]  # double, like float(token) in the Python path
    return X, labels


def parse_libsvm_data(filename, dtype=np.float64, skipped_lines: int = 0, label_type=float, use_native: bool = True):
    """Parse a LIBSVM file into ``(X[num_points, num_features], labels | None)`` (libsvm_parsing.hpp:118-229)."""
    if not os.path.isfile(filename):
        raise FileNotFoundPlssvmError(f"Couldn't find file: '{filename}'!")
    if use_native:
        fast = _parse_native(filename, dtype, skipped_lines, label_type)
        if fast is not None:
            return fast
    lines = read_lines(filename)[skipped_lines:]
    if not lines:
        raise InvalidFileFormatError("Can't parse file: no data points are given!")

    # strip a trailing in-line comment exactly as the tokenizer would stop: the reference does not support in-line
    # comments on data lines with features after them; "value # comment" parses because conversion trims the token
    parsed = []
    num_features = 0
    has_label = False
    has_no_label = False
    for line in lines:
        hash_pos = line.find("#")
        if hash_pos != -1:
            line = line[:hash_pos]
        line = line.rstrip()
        first_space = len(line)
        for k, ch in enumerate(line):
            if ch in " \n":
                first_space = k
                break
        first_colon = line.find(":")
        if first_colon == -1:
            first_colon = len(line)
        if first_colon >= first_space:
            has_label = True
            label_tok = line[:first_space]
            rest = line[first_space:]
        else:
            has_no_label = True
            label_tok = None
            rest = line
        feats = []
        last_index = 0
        for tok in rest.split():
            if ":" not in tok:
                raise InvalidFileFormatError(f"Can't convert '{tok}' to a value of type unsigned long!")
            idx_tok, val_tok = tok.split(":", 1)
            index = _to_index(idx_tok)
            if index == 0:
                raise InvalidFileFormatError("LIBSVM assumes a 1-based feature indexing scheme, but 0 was given!")
            if last_index >= index:
                raise InvalidFileFormatError(
                    f"The features indices must be strictly increasing, but {index} is smaller or equal than {last_index}!")
            last_index = index
            if val_tok == "":
                raise InvalidFileFormatError("Can't convert '' to a value of type real!")
            feats.append((index - 1, _to_float(val_tok, "real")))
        num_features = max(num_features, last_index)
        parsed.append((label_tok, feats))

    if num_features == 0:
        raise InvalidFileFormatError("Can't parse file: no data points are given!")
    if has_label and has_no_label:
        raise InvalidFileFormatError("Inconsistent label specification found (some data points are labeled, others are not)!")

    X = np.zeros((len(parsed), num_features), dtype=dtype)
    for i, (_, feats) in enumerate(parsed):
        for j, v in feats:
            X[i, j] = v
    labels = None
    if has_label:
        if label_type is str:
            labels = [p[0] for p in parsed]
        else:
            labels = [label_type(_to_float(p[0], "label")) for p in parsed]
    return X, labels


def _label_text(lab) -> str:
    """fmt's ``{}`` of a label: strings as they are, whole numbers without a fraction."""
    if isinstance(lab, str):
        return lab
    if isinstance(lab, (int, np.integer)) or float(lab).is_integer():
        return str(int(lab))
    return repr(float(lab))


def _write_native(filename, header, X, labels) -> bool:
    if X.dtype not in (np.dtype(np.float32), np.dtype(np.float64)) or not X.flags.c_contiguous:
        return False
    try:
        import ctypes as C

        from . import _capi
    except (ImportError, OSError):
        return False
    ints = text = offsets = None
    if labels is not None:
        arr = np.asarray(labels)
        if arr.dtype.kind in "iu" or (arr.dtype.kind == "f" and np.all(np.isfinite(arr)) and np.all(arr == np.rint(arr)) and np.all(np.abs(arr) < 2.0**62)):
            ints = np.ascontiguousarray(arr, dtype=np.int64)
        else:
            parts = [_label_text(lab).encode() for lab in labels]
            text = b"".join(parts)
            offsets = np.zeros(len(parts) + 1, dtype=np.uint64)
            np.cumsum([len(t) for t in parts], out=offsets[1:])
    write = _capi.lib.lssvm_mi355_libsvm_write_f32 if X.dtype == np.float32 else _capi.lib.lssvm_mi355_libsvm_write_f64
    _capi.check(write(os.fsencode(filename), header.encode() if header else None, _capi.ptr(X), C.c_uint64(X.shape[0]), C.c_uint64(X.shape[1]), C.c_uint64(X.shape[1]),
                      _capi.ptr(ints) if ints is not None else None, text, _capi.ptr(offsets) if offsets is not None else None))
    return True


def write_libsvm_data(filename, X, labels=None, comment: str | None = None, use_native: bool = True) -> None:
    """Write a (sparse: zeros omitted) LIBSVM file, one-based indices, ``label idx:val idx:val `` with ``{:.10e}`` values and a blank after every token
    (libsvm_parsing.hpp:254-296).  ``comment`` becomes a leading ``#`` line (the reference writes a time stamp and the shape there).  The rows are
    formatted by the library's multi-threaded writer (csrc/model_io.hpp); the loop below writes the same bytes and is what is left without the library."""
    X = np.ascontiguousarray(X)
    if labels is not None and len(labels) != X.shape[0]:
        raise InvalidFileFormatError(f"Number of data points ({X.shape[0]}) and number of labels ({len(labels)}) mismatch!")
    header = f"# {comment}\n" if comment is not None else ""
    if use_native and _write_native(filename, header, X, labels):
        return
    with open(filename, "w") as f:
        f.write(header)
        for i in range(X.shape[0]):
            label = _label_text(labels[i]) + " " if labels is not None else ""
            f.write(label + "".join(f"{j + 1}:{float(v):.10e} " for j, v in enumerate(X[i]) if v != 0) + "\n")
