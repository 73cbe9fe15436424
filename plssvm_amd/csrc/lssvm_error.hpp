/*
 * lssvm_error.hpp -- the error type of the library and the guard every extern "C" entry point runs under (host code only, no HIP): no exception crosses
 * the C ABI, every entry point maps lssvm::Error / std::exception to a negative lssvm_status and stores the message in a thread-local string
 * (lssvm_mi355_last_error; the C++ adaptor rethrows it as plssvm::mi355::backend_exception, the counterpart of include/plssvm/backends/HIP/exceptions.hpp
 * in the reference tree).
 */
#pragma once

#include "../../include/plssvm_amd.h"

#include <new>
#include <stdexcept>
#include <string>

namespace lssvm {

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string &msg) : std::runtime_error(msg), status(st) {}
};

#define LSSVM_REQUIRE(cond, msg)                                              \
    do {                                                                      \
        if (!(cond)) throw ::lssvm::Error(LSSVM_ERR_INVALID_ARGUMENT, (msg)); \
    } while (0)

/* one per thread and process: the translation units of the C ABI (capi.hip, capi_io.cpp) share it */
inline std::string &last_error_message() {
    static thread_local std::string message;
    return message;
}

template <typename F>
int guarded(F &&f) {
    try {
        f();
        return LSSVM_SUCCESS;
    } catch (const Error &e) {
        last_error_message() = e.what();
        return e.status;
    } catch (const std::bad_alloc &) {
        last_error_message() = "host allocation failed";
        return LSSVM_ERR_OUT_OF_MEMORY;
    } catch (const std::exception &e) {
        last_error_message() = e.what();
        return LSSVM_ERR_INTERNAL;
    } catch (...) {
        last_error_message() = "unknown error";
        return LSSVM_ERR_INTERNAL;
    }
}

}  // namespace lssvm
