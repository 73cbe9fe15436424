"""SVM parameters, mirroring ``plssvm::parameter`` (include/plssvm/parameter.hpp:105-266) and the kernel enumeration
(include/plssvm/kernel_function_types.hpp:31-38).

Defaults: kernel_type = linear, degree = 3, gamma = "default" (resolved to 1 / num_features when a model is fitted,
csvm.hpp:303-307), coef0 = 0, cost = 1 (parameter.hpp:156-165).
"""

from __future__ import annotations

import enum
from dataclasses import dataclass, replace

from .exceptions import InvalidParameterError, UnsupportedKernelTypeError

__all__ = ["KernelFunctionType", "Parameter", "kernel_function_type_from_string"]


class KernelFunctionType(enum.IntEnum):
    """plssvm::kernel_function_type (kernel_function_types.hpp:31-38)."""
    LINEAR = 0
    POLYNOMIAL = 1
    RBF = 2

    def __str__(self):  # kernel_function_types.cpp:25-35 prints lower-case names
        return self.name.lower()


def kernel_function_type_from_string(text) -> KernelFunctionType:
    """operator>> of the reference (kernel_function_types.cpp:49-66): accepts names, "poly" and the LIBSVM numbers 0, 1, 2."""
    if isinstance(text, KernelFunctionType):
        return text
    if isinstance(text, int):
        try:
            return KernelFunctionType(text)
        except ValueError:
            raise UnsupportedKernelTypeError(f"Invalid kernel function {text} given!") from None
    t = str(text).strip().lower()
    table = {"linear": 0, "0": 0, "polynomial": 1, "poly": 1, "1": 1, "rbf": 2, "2": 2}
    if t not in table:
        raise UnsupportedKernelTypeError(f"Invalid kernel function {text} given!")
    return KernelFunctionType(table[t])


@dataclass
class Parameter:
    kernel_type: KernelFunctionType = KernelFunctionType.LINEAR
    degree: int = 3
    gamma: float | None = None  # None == "default value" (default_value<T>::is_default(), default_value.hpp:62-197)
    coef0: float = 0.0
    cost: float = 1.0

    def __post_init__(self):
        self.kernel_type = kernel_function_type_from_string(self.kernel_type)
        self.degree = int(self.degree)
        self.sanity_check()

    def sanity_check(self) -> None:
        """csvm::sanity_check_parameter (csvm.hpp:377-390)."""
        if self.kernel_type in (KernelFunctionType.POLYNOMIAL, KernelFunctionType.RBF) and self.gamma is not None and self.gamma <= 0.0:
            raise InvalidParameterError(f"gamma must be greater than 0.0, but is {self.gamma}!")

    def resolved(self, num_features: int) -> "Parameter":
        """Copy with gamma = 1 / num_features if it is still the default (csvm.hpp:303-307)."""
        if self.gamma is None:
            return replace(self, gamma=1.0 / float(num_features))
        return replace(self)

    def equivalent(self, other: "Parameter") -> bool:
        """plssvm::detail::equivalent (parameter.hpp:270-300): only the fields the kernel uses are compared."""
        if self.kernel_type != other.kernel_type or self.cost != other.cost:
            return False
        if self.kernel_type == KernelFunctionType.LINEAR:
            return True
        if self.kernel_type == KernelFunctionType.POLYNOMIAL:
            return self.degree == other.degree and self.gamma == other.gamma and self.coef0 == other.coef0
        return self.gamma == other.gamma
