"""ARFF data files, the second data format of ``plssvm::data_set`` (include/plssvm/detail/io/arff_parsing.hpp): header rules and
error texts of ``parse_arff_header`` (:57-170), dense and sparse data rows of ``parse_arff_data`` (:196-372), the writer (:389-470).

    % comment
    @RELATION name
    @ATTRIBUTE first NUMERIC
    @ATTRIBUTE class {-1,1}            the (single, optional) nominal attribute with the reserved name "class" holds the labels
    @DATA
    0.1,0.2,-1                         dense: one value per attribute, in header order
    {0 0.1,2 -1}                       sparse: "index value" pairs, zero-based attribute indices, missing features are 0
"""

from __future__ import annotations

import time

import numpy as np

from .exceptions import InvalidFileFormatError
from .string_conversion import int_prefix, real_prefix
from .io_libsvm import read_lines

__all__ = ["parse_arff_header", "parse_arff_data", "write_arff_data"]

_WS = " \t\v\f\n\r"


def _type_name(t) -> str:
    if t is str:
        return "std::string"
    if t is int:
        return "int"
    if t is float:
        return "double"
    return "float" if np.dtype(t) == np.dtype(np.float32) else "double"


def _convert(token: str, t):
    """detail::convert_to<T, invalid_file_format_exception> (string_conversion.hpp): the whole trimmed token must be a value of T."""
    tok = token.strip(_WS)
    if t is str:
        return tok
    value = int_prefix(tok) if t is int else real_prefix(tok)  # the reference converts the longest valid prefix (string_conversion.py)
    if value is None:
        raise InvalidFileFormatError(f"Can't convert '{token}' to a value of type {_type_name(t)}!")
    return value if t in (int, float) else np.dtype(t).type(value)


def _show(label) -> str:
    """{fmt}'s ``{}`` of a label"""
    if isinstance(label, str):
        return label
    f = float(label)
    return str(int(f)) if f.is_integer() else repr(f)


def parse_arff_header(lines, label_type=float):
    """Returns ``(num_features, first_data_line, labels (sorted, [] without a class attribute), label_idx)`` -- arff_parsing.hpp:57-170."""
    num_features, label_idx, has_label, labels = 0, 0, False, []

    def check_for_name(line, prefix, suffix):
        sv = line[prefix:len(line) - suffix].strip(_WS)
        if not sv:
            raise InvalidFileFormatError(f'The "{line}" field must contain a name!')
        if " " in sv and not sv.startswith('"') and not sv.endswith('"'):
            raise InvalidFileFormatError(f'A "{line}" name that contains a whitespace must be quoted!')
        return sv

    header_line = 0
    while header_line < len(lines):
        line = lines[header_line]
        upper = line.upper()
        if upper.startswith("@RELATION"):
            if header_line != 0:
                raise InvalidFileFormatError("The @RELATION attribute must be set before any other @ATTRIBUTE!")
            check_for_name(line, 9, 0)
            header_line += 1
            continue
        if upper.startswith("@ATTRIBUTE"):
            if "NUMERIC" in upper:
                name = check_for_name(line, 10, 7)
                if name.upper() == "CLASS":
                    raise InvalidFileFormatError('May not use the combination of the reserved name "class" and attribute type NUMERIC!')
                num_features += 1
                if not has_label:
                    label_idx += 1
                header_line += 1
                continue
            sv = line[len("@ATTRIBUTE"):].lstrip(_WS)
            if sv.upper().startswith("CLASS"):
                if has_label:
                    raise InvalidFileFormatError("A nominal attribute with the name CLASS may only be provided once!")
                sv = sv[len("CLASS"):].strip(_WS)
                if not sv:
                    raise InvalidFileFormatError(f'The "{line}" field must contain class labels!')
                if not sv.startswith("{") and not sv.endswith("}"):
                    raise InvalidFileFormatError(f'The "{line}" nominal attribute must be enclosed with {{}}!')
                split = sv[1:-1].split(",")
                if len(split) == 1:
                    raise InvalidFileFormatError("Only a single label has been provided!")
                unique = sorted({_convert(tok, label_type) for tok in split})
                if len(split) != len(unique):
                    raise InvalidFileFormatError(f"Provided {len(split)} labels but only {len(unique)} of them was/where unique!")
                if label_type is str:
                    for tok in split:
                        if " " in tok.strip(_WS):
                            raise InvalidFileFormatError(f'String labels may not contain whitespaces, but "{tok.strip(_WS)}" has at least one!')
                labels = unique
                has_label = True
                header_line += 1
                continue
        if upper.startswith("@DATA"):
            break
        if upper.startswith("@"):
            raise InvalidFileFormatError(f'Read an invalid header entry: "{line}"!')
        header_line += 1
    if num_features == 0:
        raise InvalidFileFormatError("Can't parse file: no feature ATTRIBUTES are defined!")
    if header_line + 1 >= len(lines):
        raise InvalidFileFormatError("Can't parse file: @DATA is missing!")
    return num_features, header_line + 1, labels, (label_idx if has_label else 0)


def _parse_native(filename, dtype, label_type):
    """Fast path: libplssvm_amd's multi-threaded reader for well-formed files (csrc/arff_reader.hpp).  Returns None when the library is not built or the
    file is anything but plainly well formed -- the Python parser below then decides (and words the error exactly like the reference)."""
    dtype = np.dtype(dtype)
    if label_type not in (float, int) or dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
        return None
    try:
        import ctypes as C
        import os

        from . import _capi
    except (ImportError, OSError):
        return None
    lib = _capi.lib
    handle = C.c_void_p()
    npts, nfeat, has_label = C.c_uint64(), C.c_uint64(), C.c_int()
    if lib.lssvm_mi355_arff_open(os.fsencode(filename), C.c_int(1 if label_type is int else 0), C.byref(handle), C.byref(npts), C.byref(nfeat), C.byref(has_label)) != 0:
        return None
    try:
        X = np.empty((npts.value, nfeat.value), dtype=dtype)
        y = np.empty(npts.value, dtype=np.float64) if has_label.value else None
        fill = lib.lssvm_mi355_arff_fill_f32 if dtype == np.float32 else lib.lssvm_mi355_arff_fill_f64
        if fill(handle, _capi.ptr(X), C.c_uint64(nfeat.value), _capi.ptr(y) if y is not None else None) != 0:
            return None
    finally:
        lib.lssvm_mi355_arff_close(handle)
    return X, ([label_type(v) for v in y.tolist()] if y is not None else None)


def parse_arff_data(filename, dtype=np.float64, label_type=float, use_native: bool = True):
    """Returns ``(X[num_points, num_features], labels or None)`` -- arff_parsing.hpp:196-372."""
    if use_native:
        fast = _parse_native(filename, dtype, label_type)
        if fast is not None:
            return fast
    lines = read_lines(filename, "%")
    num_features, first, unique, label_idx = parse_arff_header(lines, label_type)
    has_label = bool(unique)
    num_attributes = num_features + (1 if has_label else 0)
    rows = lines[first:]
    X = np.zeros((len(rows), num_features), dtype=dtype)
    labels = [None] * len(rows)
    for i, full in enumerate(rows):
        line = full
        if line.startswith("@"):
            raise InvalidFileFormatError(f'Read @ inside data section!: "{line}"!')
        if line.startswith("{"):
            if not line.endswith("}"):
                raise InvalidFileFormatError(f"Missing closing '}}' for sparse data point \"{line}\" description!")
            class_set = False
            pos = 1
            while True:
                nxt = line.find(" ", pos)
                if nxt < 0:
                    break
                tok = line[pos:nxt]
                if not tok.strip(_WS).isdigit():
                    raise InvalidFileFormatError(f"Can't convert '{tok}' to a value of type unsigned long!")
                index = int(tok)
                if index >= num_attributes:
                    raise InvalidFileFormatError(f"Trying to add feature/label at index {index} but the maximum index is {num_attributes - 1}!")
                pos = nxt + 1
                ends = [p for p in (line.find(",", pos), line.find("}", pos)) if p >= 0]
                nxt = min(ends) if ends else len(line)
                if has_label and index == label_idx:
                    class_set = True
                    labels[i] = _convert(line[pos:nxt], label_type)
                else:
                    if has_label and index > label_idx:
                        index -= 1
                    X[i, index] = _convert(line[pos:nxt], dtype)
                line = line[nxt + 1:].lstrip(_WS)
                pos = 0
            if has_label and not class_set:
                raise InvalidFileFormatError(f'Missing label for data point "{full}"!')
        else:
            if line.endswith("}"):
                raise InvalidFileFormatError(f"Missing opening '{{' for sparse data point \"{line}\" description!")
            split = line.split(",")
            if len(split) != num_attributes:
                raise InvalidFileFormatError(f"Invalid number of features and labels! Found {len(split)} but should be {num_attributes}!")
            col = 0
            for j, tok in enumerate(split):
                if has_label and j == label_idx:
                    labels[i] = _convert(tok, label_type)
                else:
                    X[i, col] = _convert(tok, dtype)
                    col += 1
        if has_label and labels[i] not in unique:
            raise InvalidFileFormatError(f'Found the label "{_show(labels[i])}" which was not specified in the header ({{{",".join(_show(u) for u in unique)}}})!')
    return X, (labels if has_label else None)


def write_arff_data(filename, X, labels=None) -> None:
    """arff_parsing.hpp:389-470: dense rows, ``{:.10e}`` values, the class attribute lists the labels that occur (sorted)."""
    X = np.asarray(X)
    with open(filename, "w") as f:
        f.write(f"% This data set has been created at {time.strftime('%Y-%m-%d %H:%M:%S')}\n")
        if X.shape[0] == 0:
            return
        f.write(f"% {X.shape[0]}x{X.shape[1]}\n")
        f.write("@RELATION data_set\n")
        for j in range(X.shape[1]):
            f.write(f"@ATTRIBUTE feature_{j} NUMERIC\n")
        if labels is not None:
            f.write("@ATTRIBUTE class {" + ",".join(_show(u) for u in sorted(set(labels))) + "}\n")
        f.write("@DATA\n")
        for i in range(X.shape[0]):
            row = ",".join(f"{float(v):.10e}" for v in X[i])
            f.write(row + (f",{_show(labels[i])}" if labels is not None else "") + "\n")
