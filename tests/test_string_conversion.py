"""detail::convert_to of the reference (include/plssvm/detail/string_conversion.hpp:36-108) converts the LONGEST VALID PREFIX after skipping
leading white space (fast_float / std::from_chars, error code checked, end pointer not): the edge cases where Python's float() / int() differ."""

import numpy as np
import pytest

from plssvm_amd.exceptions import InvalidFileFormatError
from plssvm_amd.io_arff import _convert
from plssvm_amd.string_conversion import int_prefix, real_prefix


@pytest.mark.parametrize("token, want", [("1.5", 1.5), ("  -2e3", -2000.0), ("1_000", 1.0), ("3.25abc", 3.25), ("1e", 1.0), (".5", 0.5), ("5.", 5.0),
                                         ("-inf", -np.inf), ("Infinity", np.inf), ("1e+2x", 100.0)])
def test_real_prefix(token, want):
    assert real_prefix(token) == want


def test_real_prefix_nan_and_rejections():
    assert np.isnan(real_prefix("nan")) and np.isnan(real_prefix("NaN(1)"))
    for bad in ("+5", "abc", "", "  ", "-", ".", "e5", "--1"):
        assert real_prefix(bad) is None, bad


def test_int_prefix():
    assert int_prefix("42") == 42 and int_prefix(" -7x") == -7 and int_prefix("1_000") == 1 and int_prefix("12.9") == 12
    assert int_prefix("+5") is None and int_prefix("x1") is None
    assert int_prefix("-3", signed=False) is None and int_prefix("3", signed=False) == 3


def test_arff_convert_uses_the_reference_rules():
    assert _convert(" 1_000", float) == 1.0 and _convert("7;", int) == 7
    assert isinstance(_convert("2.5", np.float32), np.float32)
    with pytest.raises(InvalidFileFormatError, match="Can't convert '\\+5' to a value of type double!"):
        _convert("+5", np.float64)
