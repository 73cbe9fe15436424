"""detail::convert_to<T> of the reference (include/plssvm/detail/string_conversion.hpp:36-108), restated: leading white space is skipped, then the
LONGEST VALID PREFIX of the token is converted -- fast_float::from_chars for the real types, std::from_chars for the integral ones -- and only a
token WITHOUT such a prefix is an error (the reference checks the error code, not the end pointer).  Python's float()/int() differ at the
edges: they accept a leading '+', digit-group underscores ("1_000") and surrounding white space, and reject trailing characters."""

from __future__ import annotations

import re

_WS = " \t\v\f\n\r"
# fast_float (default format: general): [-] digits [. digits] [e[+-]digits] | [-] inf | infinity | nan, case-insensitive; no leading '+'
_REAL = re.compile(r"-?(?:(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?|inf(?:inity)?|nan)", re.IGNORECASE)
_INT = re.compile(r"-?\d+")      # std::from_chars, base 10: no leading '+', no white space
_UINT = re.compile(r"\d+")


def real_prefix(token: str):
    """the value of the longest valid real-number prefix of `token` (left-trimmed), or None"""
    m = _REAL.match(token.lstrip(_WS))
    return float(m.group(0)) if m else None


def int_prefix(token: str, signed: bool = True):
    m = (_INT if signed else _UINT).match(token.lstrip(_WS))
    return int(m.group(0)) if m else None
