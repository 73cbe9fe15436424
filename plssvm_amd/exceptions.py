"""Exception hierarchy mirroring the reference's (include/plssvm/exceptions/exceptions.hpp:29-153).

Every exception derives from :class:`PlssvmError` (reference: ``plssvm::exception``); the names keep the reference's
vocabulary so that a test written against the reference reads the same here.
"""


class PlssvmError(RuntimeError):
    """Base class (reference: ``plssvm::exception : std::runtime_error``)."""


class InvalidParameterError(PlssvmError):
    """reference: ``plssvm::invalid_parameter_exception`` (csvm.hpp:283, :291, :298, :379, :384)."""


class FileReaderError(PlssvmError):
    """reference: ``plssvm::file_reader_exception``."""


class FileNotFoundPlssvmError(PlssvmError):
    """reference: ``plssvm::file_not_found_exception``."""


class InvalidFileFormatError(PlssvmError):
    """reference: ``plssvm::invalid_file_format_exception`` (io/libsvm_parsing.hpp:118-229)."""


class UnsupportedBackendError(PlssvmError):
    """reference: ``plssvm::unsupported_backend_exception`` (csvm_factory.hpp:74-79)."""


class UnsupportedKernelTypeError(PlssvmError):
    """reference: ``plssvm::unsupported_kernel_type_exception`` (kernel_function_types.cpp:81)."""


class BackendError(PlssvmError):
    """reference: ``plssvm::hip::backend_exception`` (backends/HIP/exceptions.hpp); raised for every non-zero status
    returned through the C ABI (include/plssvm_amd.h)."""
