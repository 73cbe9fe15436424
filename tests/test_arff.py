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
