/*
 * plssvm_amd/csvm.hpp -- C++17 host-side adaptor above the C ABI (include/plssvm_amd.h).
 *
 * It restates, with the SAME names, argument meaning and error behaviour, the part of the reference's public C++ API
 * that sits directly on the hot path (citations relative to the reference tree SC-SGS/PLSSVM v2.0.0):
 *
 *   plssvm::csvm                     include/plssvm/csvm.hpp:50-222        abstract base, 4 protected pure virtuals (:188-208)
 *   plssvm::hip::csvm                include/plssvm/backends/HIP/csvm.hpp:39-99   -> plssvm_amd::mi355::csvm (constructors, target check)
 *   plssvm::make_csvm                include/plssvm/csvm_factory.hpp:123-171
 *   plssvm::backend_type             include/plssvm/backend_types.hpp:30-43 (+ the new enumerator `mi355`)
 *   plssvm::detail::parameter<T>     include/plssvm/parameter.hpp:105-266 (plain members; "is default" tracked for gamma only)
 *   plssvm::exception hierarchy      include/plssvm/exceptions/exceptions.hpp:29-153
 *
 * The types live in namespace `plssvm_amd` so that this header can be compiled next to the reference's own headers;
 * INTEGRATION.md shows the ~40-line subclass of the real `plssvm::csvm` a maintainer adds inside the reference tree, which
 * forwards to the same C entry points.
 *
 * Header only.  Link with -lplssvm_amd.  No arithmetic happens here: every virtual flattens its
 * std::vector<std::vector<T>> arguments (one heap block per row in the reference, csvm.hpp:188) into one row-major
 * buffer and calls the C ABI; a non-zero status is rethrown as the exception the reference would have thrown.
 */
#ifndef PLSSVM_AMD_CSVM_HPP
#define PLSSVM_AMD_CSVM_HPP

#include "../plssvm_amd.h"

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <iostream>
#include <memory>
#include <ostream>
#include <stdexcept>
#include <mutex>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace plssvm_amd {

/* ------------------------------------------------------------ exceptions (exceptions.hpp:29-153) ------------------------------------------------------------ */
class exception : public std::runtime_error {
  public:
    explicit exception(const std::string &msg, std::string class_name = "exception") : std::runtime_error{ msg }, class_name_{ std::move(class_name) } {}
    [[nodiscard]] std::string what_with_loc() const { return std::string{ what() } + "\nException type: plssvm_amd::" + class_name_; }

  private:
    std::string class_name_;
};
class invalid_parameter_exception : public exception {
  public:
    explicit invalid_parameter_exception(const std::string &msg) : exception{ msg, "invalid_parameter_exception" } {}
};
class unsupported_backend_exception : public exception {
  public:
    explicit unsupported_backend_exception(const std::string &msg) : exception{ msg, "unsupported_backend_exception" } {}
};
class unsupported_kernel_type_exception : public exception {
  public:
    explicit unsupported_kernel_type_exception(const std::string &msg) : exception{ msg, "unsupported_kernel_type_exception" } {}
};
namespace mi355 {
/* counterpart of plssvm::hip::backend_exception (include/plssvm/backends/HIP/exceptions.hpp) */
class backend_exception : public exception {
  public:
    explicit backend_exception(const std::string &msg) : exception{ msg, "mi355::backend_exception" } {}
};
}  // namespace mi355

/* ------------------------------------------------------------ enumerations ------------------------------------------------------------ */
enum class kernel_function_type { linear = 0, polynomial = 1, rbf = 2 };                  // kernel_function_types.hpp:31-38
enum class backend_type { automatic, openmp, cuda, hip, opencl, sycl, mi355 };            // backend_types.hpp:30-43 + mi355
enum class target_platform { automatic, cpu, gpu_nvidia, gpu_amd, gpu_intel };            // target_platforms.hpp

/* ------------------------------------------------------------ parameter (parameter.hpp:156-165) ------------------------------------------------------------ */
namespace detail {
template <typename T>
struct parameter {
    kernel_function_type kernel_type{ kernel_function_type::linear };
    int degree{ 3 };
    T gamma{ 0 };
    bool gamma_is_default{ true };  // default_value<T>::is_default(): gamma = 1 / num_features is filled in by fit (csvm.hpp:303-307)
    T coef0{ 0 };
    T cost{ 1 };

    void set_gamma(T g) {
        gamma = g;
        gamma_is_default = false;
    }
    template <typename U>
    explicit operator parameter<U>() const {  // parameter.hpp:208-210: conversion between real types
        parameter<U> p;
        p.kernel_type = kernel_type;
        p.degree = degree;
        p.gamma = static_cast<U>(gamma);
        p.gamma_is_default = gamma_is_default;
        p.coef0 = static_cast<U>(coef0);
        p.cost = static_cast<U>(cost);
        return p;
    }
};
}  // namespace detail
using parameter = detail::parameter<double>;  // parameter.hpp:328

/* ------------------------------------------------------------ named arguments (parameter.hpp:35-51, :209-262) ------------------------------------------------------------ */
/* The reference builds its named parameters with the un-vendored `igor` library: `plssvm::kernel_type = ..., plssvm::gamma = ...`.
 * The same call syntax, restated in ~40 lines: a name is an empty tag object whose operator= wraps the value. */
namespace named {
template <typename Tag, typename V>
struct argument {
    V value;
};
template <typename Tag>
struct name {
    template <typename V>
    constexpr argument<Tag, std::decay_t<V>> operator=(V &&v) const {
        return { std::forward<V>(v) };
    }
};
struct kernel_type_tag {};
struct gamma_tag {};
struct degree_tag {};
struct coef0_tag {};
struct cost_tag {};
struct num_devices_tag {};  // mi355 backend only: devices used by one solve (default 1; 0 = automatic: every visible device)

template <typename T>
struct is_argument : std::false_type {};
template <typename Tag, typename V>
struct is_argument<argument<Tag, V>> : std::true_type {};
template <typename T, typename Tag>
struct has_tag : std::false_type {};
template <typename Tag, typename V>
struct has_tag<argument<Tag, V>, Tag> : std::true_type {};
template <typename Tag, typename... Args>
constexpr int count_tag = (0 + ... + (has_tag<std::decay_t<Args>, Tag>::value ? 1 : 0));
/* "Can only use named parameter!" / "Can only use each named parameter once!" (parameter.hpp:219-222) */
template <typename... Args>
constexpr bool only_named_v = (true && ... && is_argument<std::decay_t<Args>>::value);
template <typename... Args>
constexpr bool no_duplicates_v = count_tag<kernel_type_tag, Args...> <= 1 && count_tag<gamma_tag, Args...> <= 1 && count_tag<degree_tag, Args...> <= 1
                                 && count_tag<coef0_tag, Args...> <= 1 && count_tag<cost_tag, Args...> <= 1 && count_tag<num_devices_tag, Args...> <= 1;
}  // namespace named
inline constexpr named::name<named::kernel_type_tag> kernel_type{};
inline constexpr named::name<named::gamma_tag> gamma{};
inline constexpr named::name<named::degree_tag> degree{};
inline constexpr named::name<named::coef0_tag> coef0{};
inline constexpr named::name<named::cost_tag> cost{};
inline constexpr named::name<named::num_devices_tag> num_devices{};

namespace detail {
inline const char *kernel_name(kernel_function_type k) { return k == kernel_function_type::linear ? "linear" : (k == kernel_function_type::polynomial ? "polynomial" : "rbf"); }
/* parameter<T>::set_named_arguments (parameter.hpp:216-262): kernel_type first, then the values, with the reference's warnings for
 * values the chosen kernel ignores */
template <typename... Args>
inline void set_named_arguments(::plssvm_amd::parameter &p, int *num_devices_out, const Args &...args) {
    static_assert(named::only_named_v<Args...>, "Can only use named parameter!");
    static_assert(named::no_duplicates_v<Args...>, "Can only use each named parameter once!");
    const auto warn = [&](const char *what) {
        std::clog << what << " parameter provided, which is not used in the " << kernel_name(p.kernel_type) << " kernel!" << std::endl;
    };
    const auto first = [&](const auto &a) {
        using A = std::decay_t<decltype(a)>;
        if constexpr (named::has_tag<A, named::kernel_type_tag>::value) p.kernel_type = static_cast<kernel_function_type>(a.value);
    };
    const auto second = [&](const auto &a) {
        using A = std::decay_t<decltype(a)>;
        if constexpr (named::has_tag<A, named::gamma_tag>::value) {
            p.set_gamma(static_cast<double>(a.value));
            if (p.kernel_type == kernel_function_type::linear) warn("gamma");
        } else if constexpr (named::has_tag<A, named::degree_tag>::value) {
            p.degree = static_cast<int>(a.value);
            if (p.kernel_type != kernel_function_type::polynomial) warn("degree");
        } else if constexpr (named::has_tag<A, named::coef0_tag>::value) {
            p.coef0 = static_cast<double>(a.value);
            if (p.kernel_type != kernel_function_type::polynomial) warn("coef0");
        } else if constexpr (named::has_tag<A, named::cost_tag>::value) {
            p.cost = static_cast<double>(a.value);
        } else if constexpr (named::has_tag<A, named::num_devices_tag>::value) {
            if (num_devices_out != nullptr) *num_devices_out = static_cast<int>(a.value);
        }
    };
    (first(args), ...);
    (second(args), ...);
}
}  // namespace detail

/* ------------------------------------------------------------ model (model.hpp, the members fit/predict touch) ------------------------------------------------------------ */
template <typename T>
struct model {
    parameter params{};
    std::vector<std::vector<T>> support_vectors{};  // every training point is a support vector
    std::vector<T> alpha{};
    T rho{ 0 };
    std::vector<T> w{};  // linear kernel: filled lazily by predict (model.hpp:166)
    [[nodiscard]] std::size_t num_support_vectors() const noexcept { return support_vectors.size(); }
    [[nodiscard]] std::size_t num_features() const noexcept { return support_vectors.empty() ? 0 : support_vectors.front().size(); }
};

/* ------------------------------------------------------------ csvm (csvm.hpp:50-222) ------------------------------------------------------------ */
class csvm {
  public:
    explicit csvm(parameter params = {}) : params_{ params } { sanity_check_parameter(); }
    /* csvm::csvm(Args &&...named_args) (csvm.hpp:77-83) */
    template <typename... Args, std::enable_if_t<(sizeof...(Args) > 0) && named::only_named_v<Args...>, bool> = true>
    explicit csvm(Args &&...named_args) {
        detail::set_named_arguments(params_, nullptr, named_args...);
        sanity_check_parameter();
    }
    csvm(const csvm &) = delete;
    csvm(csvm &&) noexcept = default;
    csvm &operator=(const csvm &) = delete;
    csvm &operator=(csvm &&) noexcept = default;
    virtual ~csvm() = default;

    [[nodiscard]] target_platform get_target_platform() const noexcept { return target_; }
    [[nodiscard]] parameter get_params() const noexcept { return params_; }
    void set_params(parameter params) {
        params_ = params;
        sanity_check_parameter();
    }

    /* csvm::fit (csvm.hpp:263-323): `y` holds the labels already mapped to -1 / +1 (data_set.hpp:438-454 does the mapping in
     * the reference); epsilon default 0.001, max_iter default = number of data points (csvm.hpp:268-269). */
    template <typename T>
    [[nodiscard]] model<T> fit(const std::vector<std::vector<T>> &data, const std::vector<T> &y, T epsilon = T(0.001), unsigned long long max_iter = 0) const {
        if (epsilon <= T(0)) throw invalid_parameter_exception{ "epsilon must be less than 0.0, but is " + std::to_string(epsilon) + "!" };  // csvm.hpp:283 (message verbatim)
        if (data.empty()) throw invalid_parameter_exception{ "Data vector is empty!" };
        if (y.size() != data.size()) throw invalid_parameter_exception{ "No labels given for training! Maybe the data is only usable for prediction?" };  // csvm.hpp:298
        if (max_iter == 0) max_iter = data.size();
        parameter params{ params_ };
        if (params.gamma_is_default) params.set_gamma(1.0 / static_cast<double>(data.front().size()));  // csvm.hpp:303-307
        model<T> m;
        m.params = params;
        m.support_vectors = data;
        auto res = solve_system_of_linear_equations(static_cast<detail::parameter<T>>(params), data, y, epsilon, max_iter);  // csvm.hpp:315
        m.alpha = std::move(res.first);
        m.rho = res.second;
        return m;
    }

    /* csvm::predict (csvm.hpp:325-343): sign of the decision value, +1 if > 0 else -1 (operators.hpp:180-182) */
    template <typename T>
    [[nodiscard]] std::vector<T> predict(model<T> &m, const std::vector<std::vector<T>> &points) const {
        if (!points.empty() && m.num_features() != points.front().size()) {
            throw invalid_parameter_exception{ "Number of features per data point (" + std::to_string(points.front().size())
                                               + ") must match the number of features per support vector of the provided model (" + std::to_string(m.num_features()) + ")!" };
        }
        const std::vector<T> values = predict_values(static_cast<detail::parameter<T>>(m.params), m.support_vectors, m.alpha, m.rho, m.w, points);
        std::vector<T> labels(values.size());
        for (std::size_t i = 0; i < values.size(); ++i) labels[i] = values[i] > T(0) ? T(1) : T(-1);
        return labels;
    }

    /* csvm::score (csvm.hpp:345-375) */
    template <typename T>
    [[nodiscard]] T score(model<T> &m, const std::vector<std::vector<T>> &points, const std::vector<T> &y) const {
        if (y.size() != points.size()) throw invalid_parameter_exception{ "The data set to score must have labels!" };
        const std::vector<T> labels = predict(m, points);
        std::size_t correct = 0;
        for (std::size_t i = 0; i < labels.size(); ++i) correct += (labels[i] == y[i]) ? 1u : 0u;
        return static_cast<T>(correct) / static_cast<T>(labels.size());
    }

  protected:
    /* the four pure virtuals of the backend boundary, csvm.hpp:188-208 (signatures verbatim) */
    [[nodiscard]] virtual std::pair<std::vector<float>, float> solve_system_of_linear_equations(const detail::parameter<float> &params, const std::vector<std::vector<float>> &A,
                                                                                                 std::vector<float> b, float eps, unsigned long long max_iter) const = 0;
    [[nodiscard]] virtual std::pair<std::vector<double>, double> solve_system_of_linear_equations(const detail::parameter<double> &params, const std::vector<std::vector<double>> &A,
                                                                                                   std::vector<double> b, double eps, unsigned long long max_iter) const = 0;
    [[nodiscard]] virtual std::vector<float> predict_values(const detail::parameter<float> &params, const std::vector<std::vector<float>> &support_vectors,
                                                            const std::vector<float> &alpha, float rho, std::vector<float> &w, const std::vector<std::vector<float>> &predict_points) const = 0;
    [[nodiscard]] virtual std::vector<double> predict_values(const detail::parameter<double> &params, const std::vector<std::vector<double>> &support_vectors,
                                                             const std::vector<double> &alpha, double rho, std::vector<double> &w, const std::vector<std::vector<double>> &predict_points) const = 0;

    target_platform target_{ target_platform::automatic };

  private:
    void sanity_check_parameter() const {  // csvm.hpp:377-390
        if (params_.kernel_type != kernel_function_type::linear && params_.kernel_type != kernel_function_type::polynomial && params_.kernel_type != kernel_function_type::rbf) {
            throw invalid_parameter_exception{ "Invalid kernel function " + std::to_string(static_cast<int>(params_.kernel_type)) + " given!" };
        }
        if ((params_.kernel_type == kernel_function_type::polynomial || params_.kernel_type == kernel_function_type::rbf) && !params_.gamma_is_default && params_.gamma <= 0.0) {
            throw invalid_parameter_exception{ "gamma must be greater than 0.0, but is " + std::to_string(params_.gamma) + "!" };
        }
    }
    parameter params_{};
};

/* ------------------------------------------------------------ the MI355X backend (counterpart of hip::csvm, HIP/csvm.hpp:39-99) ------------------------------------------------------------ */
namespace mi355 {

namespace detail {
template <typename T>
inline std::vector<T> flatten(const std::vector<std::vector<T>> &rows, const char *what) {
    if (rows.empty()) throw invalid_parameter_exception{ std::string{ "The " } + what + " must not be empty!" };  // csvm.cpp:73, :189
    const std::size_t d = rows.front().size();
    if (d == 0) throw invalid_parameter_exception{ std::string{ "The " } + what + " must contain at least one feature!" };  // csvm.cpp:74
    std::vector<T> flat(rows.size() * d);
    for (std::size_t i = 0; i < rows.size(); ++i) {
        if (rows[i].size() != d) throw invalid_parameter_exception{ "All data points must have the same number of features!" };  // csvm.cpp:75
        std::copy(rows[i].begin(), rows[i].end(), flat.begin() + static_cast<std::ptrdiff_t>(i * d));
    }
    return flat;
}
template <typename T>
inline lssvm_params to_c(const ::plssvm_amd::detail::parameter<T> &p) {
    return lssvm_params{ static_cast<int32_t>(p.kernel_type), static_cast<int32_t>(p.degree), static_cast<double>(p.gamma), static_cast<double>(p.coef0), static_cast<double>(p.cost) };
}
inline void check(int status) {
    if (status == LSSVM_SUCCESS) return;
    const std::string msg = lssvm_mi355_last_error();
    if (status == LSSVM_ERR_INVALID_ARGUMENT) throw invalid_parameter_exception{ msg };
    throw backend_exception{ msg };  // HIP status -> backend_exception (utility.hip.cpp:19-23)
}
}  // namespace detail

class csvm : public ::plssvm_amd::csvm {
  public:
    explicit csvm(parameter params = {}) : csvm{ target_platform::automatic, params } {}
    explicit csvm(target_platform target, parameter params = {}) : ::plssvm_amd::csvm{ params } { init(target); }
    /* hip::csvm(Args &&...named_args) / (target, Args &&...named_args) (HIP/csvm.hpp:82-99), e.g.
     *   mi355::csvm svm{ plssvm_amd::kernel_type = kernel_function_type::rbf, plssvm_amd::gamma = 0.01, plssvm_amd::num_devices = 8 }; */
    template <typename... Args, std::enable_if_t<(sizeof...(Args) > 0) && named::only_named_v<Args...>, bool> = true>
    explicit csvm(Args &&...named_args) : csvm{ target_platform::automatic, std::forward<Args>(named_args)... } {}
    template <typename... Args, std::enable_if_t<(sizeof...(Args) > 0) && named::only_named_v<Args...>, bool> = true>
    explicit csvm(const target_platform target, Args &&...named_args) : ::plssvm_amd::csvm{} {
        parameter p{};
        ::plssvm_amd::detail::set_named_arguments(p, &use_devices_, named_args...);
        set_params(p);
        init(target);
        set_num_devices(use_devices_);
    }

    /* cg tracking values of the last solve (what the reference logs, csvm.cpp:167-176).  The info block is written by the const
     * solve virtuals, like the reference's performance tracker it is not synchronised: one solve at a time per csvm object. */
    /* what the last solve of THIS object reported (a copy: several threads may call the const virtuals of one object, each solve then
     * replaces the record as a whole under a lock -- the reference's backends keep no such record at all) */
    [[nodiscard]] lssvm_cg_info last_cg_info() const {
        const std::lock_guard<std::mutex> lock(info_mutex_);
        return info_;
    }
    [[nodiscard]] int num_available_devices() const noexcept { return num_devices_; }
    /* devices one solve is sharded over: 0 = automatic (every visible device, at least 4096 points each -- the reference's backends
     * also take every device they find, csvm.hip.cpp:66-75), 1 = device 0 only, k = devices 0 .. k-1 */
    void set_num_devices(int n) {
        if (n < 0 || n > num_devices_) throw backend_exception{ "Requested " + std::to_string(n) + " devices, but only " + std::to_string(num_devices_) + " are available!" };
        use_devices_ = n;
    }
    [[nodiscard]] int get_num_devices() const noexcept { return use_devices_; }

    /* A tuning knob of THIS backend object (names and ranges: lssvm_mi355_set_option in plssvm_amd.h).  The object holds its own lssvm_mi355_options (ABI 4), created by
     * the first call from the process defaults of that moment; nothing process-wide is touched, so two csvm objects with different settings can solve at the same time
     * from two threads -- like the reference's backend objects, which share no state beyond `verbosity` (csvm.hpp:50-83). */
    void set_option(const char *name, long long value) {
        if (!options_) {
            lssvm_mi355_options *o = nullptr;
            detail::check(lssvm_mi355_options_create(&o));
            options_.reset(o);
        }
        detail::check(lssvm_mi355_options_set(options_.get(), name, static_cast<int64_t>(value)));
    }
    [[nodiscard]] long long get_option(const char *name) const {
        int64_t v = 0;
        detail::check(options_ ? lssvm_mi355_options_get(options_.get(), name, &v) : lssvm_mi355_get_option(name, &v));
        return static_cast<long long>(v);
    }

    /* the tracking entries of the last solve in the layout of the reference's performance tracker (performance_tracker.cpp:139-190):
     * one YAML document with the groups `backend` (csvm.hip.cpp:59-60) and `cg` (csvm.cpp:167-174, csvm.hpp:318-320) */
    void write_tracking_yaml(std::ostream &out) const {
        const lssvm_cg_info info = last_cg_info();
        out << "---\n"
            << "backend:\n"
            << "  backend: mi355\n"
            << "  target_platform: gpu_amd\n"
            << "  num_devices: " << info.devices_used << "\n"
            << "\n"
            << "cg:\n"
            << "  iterations: " << info.iterations << "\n"
            << "  max_iterations: " << info.max_iterations << "\n"
            << "  residuum: " << info.residuum << "\n"
            << "  target_residuum: " << info.target_residuum << "\n"
            << "  avg_iteration_time: " << info.avg_iteration_ms << "ms\n"
            << "  epsilon: " << info.epsilon << "\n"
            << "  total_runtime: " << info.total_ms << "ms\n"
            << "\n";
    }

  protected:
    [[nodiscard]] std::pair<std::vector<float>, float> solve_system_of_linear_equations(const ::plssvm_amd::detail::parameter<float> &params, const std::vector<std::vector<float>> &A,
                                                                                        std::vector<float> b, float eps, unsigned long long max_iter) const override {
        return solve_impl<float>(params, A, b, eps, max_iter, &lssvm_mi355_solve_multi_f32);
    }
    [[nodiscard]] std::pair<std::vector<double>, double> solve_system_of_linear_equations(const ::plssvm_amd::detail::parameter<double> &params, const std::vector<std::vector<double>> &A,
                                                                                          std::vector<double> b, double eps, unsigned long long max_iter) const override {
        return solve_impl<double>(params, A, b, eps, max_iter, &lssvm_mi355_solve_multi_f64);
    }
    [[nodiscard]] std::vector<float> predict_values(const ::plssvm_amd::detail::parameter<float> &params, const std::vector<std::vector<float>> &support_vectors, const std::vector<float> &alpha,
                                                    float rho, std::vector<float> &w, const std::vector<std::vector<float>> &predict_points) const override {
        return predict_impl<float>(params, support_vectors, alpha, rho, w, predict_points, &lssvm_mi355_predict_values_f32);
    }
    [[nodiscard]] std::vector<double> predict_values(const ::plssvm_amd::detail::parameter<double> &params, const std::vector<std::vector<double>> &support_vectors,
                                                     const std::vector<double> &alpha, double rho, std::vector<double> &w, const std::vector<std::vector<double>> &predict_points) const override {
        return predict_impl<double>(params, support_vectors, alpha, rho, w, predict_points, &lssvm_mi355_predict_values_f64);
    }

  private:
    void init(target_platform target) {  // csvm.hip.cpp:47-85
        if (target != target_platform::automatic && target != target_platform::gpu_amd) {
            throw backend_exception{ "Invalid target platform '" + std::to_string(static_cast<int>(target)) + "' for the MI355 backend!" };  // csvm.hip.cpp:49-51
        }
        target_ = target_platform::gpu_amd;  // csvm.hip.cpp:63
        num_devices_ = lssvm_mi355_device_count();
        if (num_devices_ <= 0) throw backend_exception{ "MI355 backend selected but no HIP capable devices were found!" };  // csvm.hip.cpp:70-72
    }

    template <typename T, typename F>
    std::pair<std::vector<T>, T> solve_impl(const ::plssvm_amd::detail::parameter<T> &params, const std::vector<std::vector<T>> &A, const std::vector<T> &b, T eps,
                                            unsigned long long max_iter, F fn) const {
        const std::vector<T> flat = detail::flatten(A, "data");
        if (A.size() != b.size()) {
            throw invalid_parameter_exception{ "The number of data points in the matrix A (" + std::to_string(A.size()) + ") and the values in the right hand side vector ("
                                               + std::to_string(b.size()) + ") must be the same!" };  // csvm.cpp:76
        }
        const lssvm_params p = detail::to_c(params);
        std::vector<T> alpha(A.size());
        T rho{};
        // all devices of this process behind ONE call, like gpu_csvm::solve_system_of_linear_equations_impl (gpu_csvm.hpp:477-654)
        lssvm_cg_info info{};
        detail::check(fn(&p, flat.data(), A.size(), A.front().size(), b.data(), eps, static_cast<uint64_t>(max_iter), alpha.data(), &rho, &info, nullptr, use_devices_, options_.get()));
        {
            const std::lock_guard<std::mutex> lock(info_mutex_);
            info_ = info;
        }
        return std::make_pair(std::move(alpha), rho);
    }

    template <typename T, typename F>
    std::vector<T> predict_impl(const ::plssvm_amd::detail::parameter<T> &params, const std::vector<std::vector<T>> &support_vectors, const std::vector<T> &alpha, T rho, std::vector<T> &w,
                                const std::vector<std::vector<T>> &predict_points, F fn) const {
        const std::vector<T> sv = detail::flatten(support_vectors, "support vectors");
        const std::vector<T> pts = detail::flatten(predict_points, "data points to predict");
        const std::size_t d = support_vectors.front().size();
        if (support_vectors.size() != alpha.size()) {
            throw invalid_parameter_exception{ "The number of support vectors (" + std::to_string(support_vectors.size()) + ") and number of weights (" + std::to_string(alpha.size())
                                               + ") must be the same!" };  // csvm.cpp:192
        }
        if (!w.empty() && w.size() != d) {
            throw invalid_parameter_exception{ "Either w must be empty or contain exactly the same number of values (" + std::to_string(w.size()) + ") as features are present ("
                                               + std::to_string(d) + ")!" };  // csvm.cpp:193
        }
        if (predict_points.front().size() != d) {
            throw invalid_parameter_exception{ "The number of features in the support vectors (" + std::to_string(d) + ") must be the same as in the data points to predict ("
                                               + std::to_string(predict_points.front().size()) + ")!" };  // csvm.cpp:197
        }
        const lssvm_params p = detail::to_c(params);
        int w_valid = w.empty() ? 0 : 1;
        std::vector<T> w_buf = w.empty() ? std::vector<T>(d) : w;
        std::vector<T> out(predict_points.size());
        detail::check(fn(&p, sv.data(), support_vectors.size(), d, alpha.data(), rho, w_buf.data(), &w_valid, pts.data(), predict_points.size(), out.data(), nullptr, options_.get()));
        if (params.kernel_type == kernel_function_type::linear && w_valid != 0) w = std::move(w_buf);  // csvm.cpp:204-207: w is filled for the linear kernel only
        return out;
    }

    struct options_deleter {
        void operator()(lssvm_mi355_options *o) const noexcept { (void) lssvm_mi355_options_destroy(o); }
    };
    std::unique_ptr<lssvm_mi355_options, options_deleter> options_{};  // this object's own tuning knobs; empty = the process defaults (move-only, like the object)
    int num_devices_{ 0 };  // visible devices
    int use_devices_{ 1 };  // devices per solve: 1 by default (several devices are opt-in: plssvm_amd::num_devices = k, 0 = every visible device)
    mutable std::mutex info_mutex_;
    mutable lssvm_cg_info info_{};  // record of the last solve (written by the const boundary virtuals: guarded)
};

}  // namespace mi355

/* ------------------------------------------------------------ factory (csvm_factory.hpp:123-171) ------------------------------------------------------------ */
/* The reference decides the default backend from the compiled-in ones (backend_types.cpp:48-71: for gpu_amd hip > opencl > sycl).
 * Here exactly one backend exists; `automatic` and `mi355` select it, `hip` is accepted as its drop-in alias, every other
 * enumerator throws unsupported_backend_exception with the reference's message (csvm_factory.hpp:74-79). */
inline const char *backend_type_to_string(backend_type b) {
    switch (b) {
        case backend_type::automatic: return "automatic";
        case backend_type::openmp: return "openmp";
        case backend_type::cuda: return "cuda";
        case backend_type::hip: return "hip";
        case backend_type::opencl: return "opencl";
        case backend_type::sycl: return "sycl";
        case backend_type::mi355: return "mi355";
    }
    return "unknown";
}

template <typename... Args>
[[nodiscard]] inline std::unique_ptr<csvm> make_csvm(const backend_type backend, Args &&...args) {
    switch (backend) {
        case backend_type::automatic:
        case backend_type::mi355:
        case backend_type::hip:
            return std::make_unique<mi355::csvm>(std::forward<Args>(args)...);
        default:
            throw unsupported_backend_exception{ std::string{ "No " } + backend_type_to_string(backend) + " backend available!" };
    }
}
template <typename... Args>
[[nodiscard]] inline std::unique_ptr<csvm> make_csvm(Args &&...args) {
    return make_csvm(backend_type::automatic, std::forward<Args>(args)...);
}

}  // namespace plssvm_amd

#endif  // PLSSVM_AMD_CSVM_HPP
