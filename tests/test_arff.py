"""CPU: ARFF, the second file format of ``plssvm::data_set`` (include/plssvm/detail/io/arff_parsing.hpp), pinned on the data files
and expectations of the reference's own tests (tests/golden/arff/, written by tests/golden/make_arff_fixture.py)."""

import json
import os

import numpy as np
import pytest

from conftest import ROOT
from plssvm_amd.data_set import DataSet, DataSetError
from plssvm_amd.exceptions import InvalidFileFormatError
from plssvm_amd.io_arff import parse_arff_data, parse_arff_header, write_arff_data
from plssvm_amd.io_libsvm import read_lines

FIX = os.path.join(ROOT, "tests", "golden", "arff")
EXPECTED = json.load(open(os.path.join(FIX, "expected.json")))
LABEL_TYPES = {"int": int, "float": float, "str": str}


@pytest.mark.parametrize("real_type", [np.float32, np.float64])
@pytest.mark.parametrize("name", sorted(EXPECTED["valid"]))
def test_valid_files_of_the_reference_tests(name, real_type):
    exp = EXPECTED["valid"][name]
    lt = LABEL_TYPES[exp["label_type"]]
    X, labels = parse_arff_data(os.path.join(FIX, name), dtype=real_type, label_type=lt)
    assert X.dtype == real_type and np.array_equal(X, np.array(exp["data"], dtype=np.float64).astype(real_type))
    assert labels == (None if exp["labels"] is None else [lt(v) for v in exp["labels"]])


@pytest.mark.parametrize("name", sorted(EXPECTED["invalid"]))
def test_invalid_files_of_the_reference_tests(name):
    for lt in (int, str):
        with pytest.raises(InvalidFileFormatError) as e:
            parse_arff_data(os.path.join(FIX, "invalid", name), label_type=lt)
        assert str(e.value) == EXPECTED["invalid"][name]


def test_invalid_files_whose_error_depends_on_the_label_type():
    for name, by_type in EXPECTED["invalid_by_label_type"].items():
        for kind, message in by_type.items():
            with pytest.raises(InvalidFileFormatError) as e:
                parse_arff_data(os.path.join(FIX, "invalid", name), label_type=LABEL_TYPES[kind])
            assert str(e.value) == message
    with pytest.raises(InvalidFileFormatError, match="no feature ATTRIBUTES"):  # arff_parsing.cpp:177-184
        parse_arff_header(read_lines(os.path.join(FIX, "empty.txt"), "%"), int)


def test_writer_layout_and_round_trip(tmp_path):
    rng = np.random.default_rng(11)
    X = rng.normal(size=(6, 3))
    X[2, 1] = 0.0
    labels = [1, -1, 1, 1, -1, -1]
    path = tmp_path / "out.arff"
    write_arff_data(path, X, labels)
    text = open(path).read().splitlines()
    assert text[0].startswith("% This data set has been created at ") and text[1] == "% 6x3" and text[2] == "@RELATION data_set"
    assert text[3:6] == [f"@ATTRIBUTE feature_{j} NUMERIC" for j in range(3)] and text[6] == "@ATTRIBUTE class {-1,1}" and text[7] == "@DATA"
    assert text[8] == ",".join(f"{v:.10e}" for v in X[0]) + ",1" and len(text) == 8 + 6
    X2, l2 = parse_arff_data(path, label_type=int)
    assert l2 == labels and np.allclose(X2, X, rtol=0, atol=1e-10 * np.max(np.abs(X)))
    write_arff_data(path, X)  # without labels: no class attribute
    X3, l3 = parse_arff_data(path)
    assert l3 is None and np.array_equal(X3, X2)


def test_data_set_reads_and_saves_both_formats_by_extension(tmp_path):
    ds = DataSet(filename=os.path.join(FIX, "5x4_int.arff"), label_type=int)  # ".arff" -> the ARFF parser (data_set.hpp:496-497)
    assert ds.num_data_points() == 5 and ds.num_features() == 4 and ds.different_labels() == [-1, 1] and list(ds.mapped_labels()) == [-1, -1, 1, 1, 1]
    ds.save(tmp_path / "copy.libsvm")
    ds.save(tmp_path / "copy.arff")
    a = DataSet(filename=tmp_path / "copy.libsvm", label_type=int)
    b = DataSet(filename=tmp_path / "copy.arff", label_type=int)
    assert np.allclose(a.data(), ds.data(), atol=1e-10) and np.allclose(b.data(), ds.data(), atol=1e-10) and a.labels() == b.labels() == ds.labels()
    with pytest.raises(DataSetError, match=r'Unrecognized file extension for file ".*copy.txt" \(must be one of: .libsvm or .arff\)!'):
        ds.save(tmp_path / "copy.txt")
    ds.save(tmp_path / "copy.txt", file_format="arff")  # an explicit format overrides the extension (data_set.hpp:580-600)
    assert DataSet(filename=tmp_path / "copy.txt", file_format="arff", label_type=int).labels() == ds.labels()


# ---------------------------------------------------------------------------------------------------------------------------------------------------
# the native fast path (plssvm_amd/csrc/arff_reader.hpp through the C ABI): accepts a NARROWER language than the format, never a wider one
# ---------------------------------------------------------------------------------------------------------------------------------------------------
def _both(path, dtype, lt):
    """(native result or None, Python parser's result or its exception)"""
    from plssvm_amd.io_arff import _parse_native

    fast = _parse_native(str(path), dtype, lt)
    try:
        slow = parse_arff_data(path, dtype=dtype, label_type=lt, use_native=False)
    except InvalidFileFormatError as e:
        slow = e
    return fast, slow


def _same(fast, slow):
    return not isinstance(slow, Exception) and fast[0].dtype == slow[0].dtype and np.array_equal(fast[0], slow[0]) and fast[1] == slow[1]


def _random_arff(rng, rows, features, sparse_share, with_label, label_pos):
    out = ["% generated", "@RELATION generated"]
    attrs = [f"@ATTRIBUTE f{j} NUMERIC" for j in range(features)]
    if with_label:
        attrs.insert(label_pos, "@attribute CLASS {-1, 1,3}")
    out += attrs + ["@DATA"]
    for _ in range(rows):
        vals = [repr(float(v)) if rng.random() < 0.8 else str(int(v * 10)) for v in rng.normal(size=features)]
        label = str(rng.choice([-1, 1, 3]))
        if rng.random() < sparse_share:
            cols = sorted(rng.choice(features, size=rng.integers(0, features + 1), replace=False).tolist())
            entries = [(c + (1 if with_label and c >= label_pos else 0), vals[c]) for c in cols]
            if with_label:
                entries.append((label_pos, label))
                entries.sort()
            out.append("{" + ",".join(f"{i} {v}" for i, v in entries) + "}")
        else:
            row = list(vals)
            if with_label:
                row.insert(label_pos, label)
            out.append(",".join(row))
    return "\n".join(out) + "\n"


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_native_arff_reader_equals_python_parser(tmp_path, dtype):
    """The multi-threaded reader against the line-by-line Python parser: the reference's own well-formed fixtures (dense, sparse, without labels; int and float
    label types) and generated files (labels in every attribute position, mixed dense / sparse rows, CRLF, comments; 6 000 rows: several threads)."""
    from plssvm_amd import _capi  # noqa: F401  (the library must be loadable for the fast path to be exercised)

    for name in ("3x2_without_label.arff", "5x4_float.arff", "5x4_int.arff", "5x4_sparse_float.arff", "5x4_sparse_int.arff"):
        for lt in (float, int):
            fast, slow = _both(os.path.join(FIX, name), dtype, lt)
            if isinstance(slow, Exception):  # (a float-labelled file read with int labels: the reference's conversion decides, not the fast path)
                assert fast is None or lt is int, name
                continue
            assert fast is None or _same(fast, slow), (name, lt)
            if lt is float:
                assert fast is not None, name  # every well-formed fixture takes the fast path with float labels
    rng = np.random.default_rng(5)
    for k, (rows, features, sparse_share, with_label) in enumerate([(40, 5, 0.5, True), (40, 1, 0.3, True), (30, 7, 1.0, False), (6000, 9, 0.4, True), (25, 3, 0.0, False)]):
        for label_pos in ({0, features // 2, features} if with_label else {0}):
            text = _random_arff(rng, rows, features, sparse_share, with_label, label_pos)
            if k % 2:
                text = text.replace("\n", "\r\n")
            f = tmp_path / f"gen_{k}_{label_pos}.arff"
            f.write_bytes(text.encode())
            for lt in (float, int):
                fast, slow = _both(f, dtype, lt)
                assert fast is not None and _same(fast, slow), (k, label_pos, lt)
            X, labels = parse_arff_data(f, dtype=dtype, label_type=int)  # the public entry point takes the fast path by itself
            assert np.array_equal(X, slow[0]) and labels == [int(v) for v in slow[1]] if with_label else labels is None


HEADER = "@RELATION r\n@ATTRIBUTE a NUMERIC\n@ATTRIBUTE b NUMERIC\n@ATTRIBUTE class {0,1}\n@DATA\n"


@pytest.mark.parametrize("text", [
    HEADER + "1,2,0 \n{0 1,2 1} \n",                    # a blank behind a closing brace: the reference then misses the brace
    HEADER.replace("a NUMERIC", "a NUMERIC ") + "1,2,0\n",   # a blank behind NUMERIC: the reference cuts the name differently
    HEADER.replace("@DATA", "@DATA ") + "1,2,0\n",
    HEADER + "{ 0 1,2 1}\n",                             # a blank behind the opening brace: not an index for the reference
    HEADER + "{0\t1,2 1}\n",                             # a tab between index and value: the reference looks for a space
    HEADER + "1,+2,0\n",                                 # "+2": converts for the reference, not for std::from_chars
    HEADER + "1,2abc,0\n",                               # the reference converts the longest valid prefix
    HEADER + "{0 1 5,2 1}\n",
    HEADER + "{0 1,2 1,}\n",                             # a trailing comma: the reference lets it pass
    HEADER + "1,nan,0\n",
    HEADER.replace("@RELATION r", '@RELATION "r s"') + "1,2,0\n",
    HEADER.replace("class {0,1}", "class{0,1}") + "1,2,0\n",
    HEADER.replace("class {0,1}", "classes {0,1}") + "1,2,0\n",
    HEADER.replace("class {0,1}", "class {0,1.5}") + "1,2,1.5\n",   # with int labels the reference reads the prefix 1
    "some text\n" + HEADER + "1,2,0\n",                  # the reference skips lines without '@' in the header
    HEADER + "1,2,2\n",                                  # a label the header does not list
    HEADER + "1,2\n",
    HEADER,                                              # no data rows
])
def test_native_arff_reader_declines_what_it_cannot_vouch_for(tmp_path, text):
    """Never wider than the reference: whatever the fast path accepts, the reference-exact parser accepts with the same result -- and the shapes listed here, which
    the reference treats in ways of its own (or rejects), are not accepted at all."""
    f = tmp_path / "case.arff"
    f.write_text(text)
    for lt in (int, float):
        fast, slow = _both(f, np.float64, lt)
        if "1.5" in text and lt is float:  # (a float label is fine for a float label type)
            assert fast is not None and _same(fast, slow)
        else:
            assert fast is None, (text, lt)
    # and the plainest variations ARE taken: no @RELATION, a comment first, keywords in lower case, blanks around the values
    g = tmp_path / "plain.arff"
    g.write_text("% c\n" + HEADER.replace("@RELATION r\n", "").replace("@ATTRIBUTE a", "@attribute  a").replace("@DATA", "@data") + " 1 , 2 ,0\n{0 3,  2 1}\n{2 0}\n")
    fast, slow = _both(g, np.float64, int)
    assert fast is not None and _same(fast, slow) and fast[0].tolist() == [[1.0, 2.0], [3.0, 0.0], [0.0, 0.0]] and fast[1] == [0, 1, 0]


def test_native_arff_reader_is_never_wider_under_random_damage(tmp_path):
    """Every single-character damage of a valid file: if the fast path still accepts it, so does the reference-exact parser, with the same matrix and labels."""
    rng = np.random.default_rng(9)
    base = _random_arff(rng, 12, 4, 0.5, True, 2)
    alphabet = " \t,{}@%+-.e0159xN\n\"'"
    accepted = 0
    for trial in range(400):
        pos = int(rng.integers(0, len(base)))
        kind = trial % 3
        ch = alphabet[int(rng.integers(0, len(alphabet)))]
        text = base[:pos] + ch + base[pos + 1:] if kind == 0 else (base[:pos] + ch + base[pos:] if kind == 1 else base[:pos] + base[pos + 1:])
        f = tmp_path / "damaged.arff"
        f.write_text(text)
        for lt in (int, float):
            fast, slow = _both(f, np.float64, lt)
            if fast is not None:
                accepted += 1
                assert _same(fast, slow), (trial, lt, text)
    assert accepted > 20  # (damage inside a comment, a name or a number often leaves a well-formed file)
