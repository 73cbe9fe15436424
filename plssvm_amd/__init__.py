"""plssvm_amd -- MI355X-native LS-SVM Conjugate-Gradient backend behind the PLSSVM backend boundary.

The compute path is the HIP library ``plssvm_amd/lib/libplssvm_amd.so`` (C ABI: ``include/plssvm_amd.h``); importing a module
that needs it (``backend``, ``csvm``, ``svc``, ``cli``) fails loudly when the library has not been built.  Pure host-side
modules (``parameter``, ``io_libsvm``, ``io_arff``, ``io_scaling_factors``, ``data_set``, ``model``, ``datagen``, ``sharding``) import without it.
"""

__version__ = "0.1.0"

from .exceptions import (BackendError, FileNotFoundPlssvmError, InvalidFileFormatError, InvalidParameterError, PlssvmError,  # noqa: F401
                         UnsupportedBackendError, UnsupportedKernelTypeError)
from .parameter import KernelFunctionType, Parameter  # noqa: F401


def __getattr__(name):
    # lazy: these pull in the native library
    if name in ("CSVM", "MI355CSVM", "make_csvm", "BackendType", "TargetPlatform"):
        from . import csvm
        return getattr(csvm, name)
    if name == "SVC":
        from .svc import SVC
        return SVC
    if name in ("DataSet", "Scaling"):
        from . import data_set
        return getattr(data_set, name)
    if name in ("Model",):
        from .model import Model
        return Model
    raise AttributeError(name)
