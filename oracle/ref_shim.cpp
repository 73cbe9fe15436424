/*
 * ref_shim.cpp -- C-ABI glue around the REFERENCE's own OpenMP kernel translation units.  TEST INFRASTRUCTURE ONLY.
 *
 * oracle/Makefile compiles this file TOGETHER with
 *     /root/reference/src/plssvm/backends/OpenMP/svm_kernel.cpp
 *     /root/reference/src/plssvm/backends/OpenMP/q_kernel.cpp
 * (from where they lie; nothing of the reference is copied into this repository) into oracle/_ref/liblssvm_ref.so.
 * The kernels (device_kernel_*, device_kernel_q_*), kernel_function<> and the BLAS-1 operators used below are the
 * reference's, included from /root/reference/include.  The only restated part is the ~100-line CG driver
 * (src/plssvm/backends/OpenMP/csvm.cpp:71-183) and predict_values (:188-227): their translation unit needs the
 * un-vendored `igor` named-argument library (CMakeLists.txt:192-220) and therefore cannot be compiled here; the
 * restatement below calls the reference's compiled kernels and the reference's operators.hpp line by line.  (Plus sampled_rows: one ROW of the implicit matvec as a
 * loop around the reference's kernel_function<>, for inputs whose full triangle would take the reference hours.)
 *
 * The only third-party header needed is {fmt} (assert message formatting, include/plssvm/detail/assert.hpp:18-19);
 * the copy shipped inside this image's PyTorch is used (a real {fmt}, not a stand-in).
 */
#include "plssvm/backends/OpenMP/q_kernel.hpp"    // plssvm::openmp::device_kernel_q_{linear,polynomial,rbf}
#include "plssvm/backends/OpenMP/svm_kernel.hpp"  // plssvm::openmp::device_kernel_{linear,polynomial,rbf}
#include "plssvm/detail/operators.hpp"            // transposed * vec, sum, vector arithmetic
#include "plssvm/kernel_function_types.hpp"       // plssvm::kernel_function<kernel>

#include "lssvm_oracle.h"  // oracle_cg_info (shared result struct)

#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace {

template <typename T>
std::vector<std::vector<T>> to_rows(const T *X, std::size_t N, std::size_t d) {
    std::vector<std::vector<T>> rows(N, std::vector<T>(d));
    for (std::size_t i = 0; i < N; ++i) {
        std::copy(X + i * d, X + (i + 1) * d, rows[i].begin());
    }
    return rows;
}

// runtime dispatch of kernel_function; restates src/plssvm/kernel_function_types.cpp:69-82 (that TU needs parameter.hpp -> igor)
template <typename T>
T kf(int kernel_type, int degree, T gamma, T coef0, const std::vector<T> &a, const std::vector<T> &b) {
    using plssvm::kernel_function_type;
    switch (kernel_type) {
        case 0:
            return plssvm::kernel_function<kernel_function_type::linear>(a, b);
        case 1:
            return plssvm::kernel_function<kernel_function_type::polynomial>(a, b, degree, gamma, coef0);
        default:
            return plssvm::kernel_function<kernel_function_type::rbf>(a, b, gamma);
    }
}

// openmp::csvm::generate_q, csvm.cpp:232-251
template <typename T>
std::vector<T> gen_q(int kernel_type, int degree, T gamma, T coef0, const std::vector<std::vector<T>> &data) {
    std::vector<T> q(data.size() - 1);
    switch (kernel_type) {
        case 0: plssvm::openmp::device_kernel_q_linear(q, data); break;
        case 1: plssvm::openmp::device_kernel_q_polynomial(q, data, degree, gamma, coef0); break;
        default: plssvm::openmp::device_kernel_q_rbf(q, data, gamma); break;
    }
    return q;
}

// openmp::csvm::run_device_kernel, csvm.cpp:283-306 (cost argument is 1 / params.cost)
template <typename T>
void run_kernel(int kernel_type, int degree, T gamma, T coef0, T inv_cost, const std::vector<T> &q, std::vector<T> &ret,
                const std::vector<T> &d, const std::vector<std::vector<T>> &data, T QA_cost, T add) {
    switch (kernel_type) {
        case 0: plssvm::openmp::device_kernel_linear(q, ret, d, data, QA_cost, inv_cost, add); break;
        case 1: plssvm::openmp::device_kernel_polynomial(q, ret, d, data, QA_cost, inv_cost, add, degree, gamma, coef0); break;
        default: plssvm::openmp::device_kernel_rbf(q, ret, d, data, QA_cost, inv_cost, add, gamma); break;
    }
}

// SAMPLED ROWS of one implicit matvec, for inputs whose full lower triangle takes the reference hours (BASELINE configs[4], 1 000 000 points): row i of
// ret += add * Abar * d as the sum over j of the reference's own per-pair expression (svm_kernel.cpp:45-52: temp = (kernel_function(data[i], data[j]) + QA_cost - q[i] - q[j]) * add,
// the diagonal with "+ cost * add"), with the reference's COMPILED kernel_function<> -- only the loop over one row instead of the triangle is written here.  The sum runs in
// j order in T, so a float64 call is the yardstick (the triangle's own order differs from it in the last digits of a double).
template <typename T>
void sampled_rows(int kernel_type, int degree, T gamma, T coef0, T inv_cost, const std::vector<T> &q, const std::vector<T> &d, const std::vector<std::vector<T>> &data, T QA_cost,
                  T add, const std::uint64_t *rows, std::size_t nrows, T *out) {
    const std::size_t n = data.size() - 1;
    #pragma omp parallel for schedule(dynamic, 1)
    for (std::size_t k = 0; k < nrows; ++k) {
        const std::size_t i = static_cast<std::size_t>(rows[k]);
        T sum{ 0.0 };
        for (std::size_t j = 0; j < n; ++j) {
            const T temp = (kf<T>(kernel_type, degree, gamma, coef0, data[i], data[j]) + QA_cost - q[i] - q[j]) * add;
            sum += (i == j ? temp + inv_cost * add : temp) * d[j];
        }
        out[k] = sum;
    }
}

// csvm.cpp:71-183, statement by statement, with the reference's operators
template <typename T>
int solve(int kernel_type, int degree, T gamma, T coef0, T cost, const T *X, std::size_t N, std::size_t dim, const T *y, T eps,
          std::uint64_t max_iter, T *alpha_out, T *rho_out, oracle_cg_info *info, double *delta_trace, std::size_t trace_cap) {
    using namespace plssvm::operators;
    if (X == nullptr || y == nullptr || N < 2 || dim == 0) return -1;
    if (!(eps > T{ 0.0 })) return -2;
    if (max_iter == 0) return -3;

    const auto t_start = std::chrono::steady_clock::now();
    const std::vector<std::vector<T>> A = to_rows(X, N, dim);
    std::vector<T> b(y, y + N);

    const std::vector<T> q = gen_q(kernel_type, degree, gamma, coef0, A);                        // :83
    const T QA_cost = kf(kernel_type, degree, gamma, coef0, A.back(), A.back()) + T{ 1.0 } / cost;  // :86
    const T inv_cost = 1 / cost;                                                                 // :297

    const T b_back_value = b.back();  // :89-91
    b.pop_back();
    b -= b_back_value;

    std::vector<T> alpha(b.size(), 1.0);  // :95
    const std::size_t dept = b.size();
    std::vector<T> r(b);                                                                           // :101
    run_kernel(kernel_type, degree, gamma, coef0, inv_cost, q, r, alpha, A, QA_cost, T{ -1.0 });  // :104

    T delta = transposed<T>{ r } * r;  // :107
    const T delta0 = delta;
    std::vector<T> Ad(dept);
    std::vector<T> d(r);

    double iter_ms = 0.0;
    unsigned long long iter = 0;
    for (; iter < max_iter; ++iter) {  // :125
        const auto t_it = std::chrono::steady_clock::now();
        const auto lap = [&]() { iter_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_it).count(); };
        std::fill(Ad.begin(), Ad.end(), T{ 0.0 });                                                  // :131
        run_kernel(kernel_type, degree, gamma, coef0, inv_cost, q, Ad, d, A, QA_cost, T{ 1.0 });  // :132
        const T alpha_cd = delta / (transposed<T>{ d } * Ad);                                       // :135
        alpha += alpha_cd * d;                                                                      // :138
        if (iter % 50 == 49) {                                                                      // :140
            r = b;
            run_kernel(kernel_type, degree, gamma, coef0, inv_cost, q, r, alpha, A, QA_cost, T{ -1.0 });
        } else {
            r -= alpha_cd * Ad;  // :148
        }
        const T delta_old = delta;  // :152
        delta = transposed<T>{ r } * r;
        if (delta_trace != nullptr && iter < trace_cap) delta_trace[iter] = static_cast<double>(delta);
        if (delta <= eps * eps * delta0) {  // :155
            lap();
            break;
        }
        const T beta = delta / delta_old;  // :161
        d = beta * d + r;                  // :163
        lap();
    }
    const std::uint64_t its = std::min<unsigned long long>(iter + 1, max_iter);  // :169

    const T bias = b_back_value + QA_cost * sum(alpha) - (transposed<T>{ q } * alpha);  // :179
    alpha.push_back(-sum(alpha));                                                       // :180
    std::copy(alpha.begin(), alpha.end(), alpha_out);
    *rho_out = -bias;  // :182

    if (info != nullptr) {
        info->iterations = its;
        info->delta = static_cast<double>(delta);
        info->delta0 = static_cast<double>(delta0);
        info->target = static_cast<double>(eps * eps * delta0);
        info->avg_iter_ms = iter_ms / static_cast<double>(its);
        info->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    }
    return 0;
}

// csvm.cpp:255-280
template <typename T>
void calc_w(const T *sv, std::size_t nsv, std::size_t dim, const T *alpha, T *w) {
    for (std::size_t f = 0; f < dim; ++f) {
        T temp{ 0.0 };
        for (std::size_t i = 0; i < nsv; ++i) {
            temp = std::fma(alpha[i], sv[i * dim + f], temp);
        }
        w[f] = temp;
    }
}

// csvm.cpp:188-227
template <typename T>
void predict(int kernel_type, int degree, T gamma, T coef0, const T *sv, std::size_t nsv, std::size_t dim, const T *alpha, T rho,
             T *w_inout, int *w_valid, const T *points, std::size_t npoints, T *out) {
    using namespace plssvm::operators;
    const std::vector<std::vector<T>> svs = to_rows(sv, nsv, dim);
    const std::vector<std::vector<T>> pts = to_rows(points, npoints, dim);
    if (kernel_type == 0 && !*w_valid) {
        calc_w(sv, nsv, dim, alpha, w_inout);
        *w_valid = 1;
    }
    const std::vector<T> w(w_inout, w_inout + dim);
    for (std::size_t p = 0; p < npoints; ++p) {
        T o = -rho;
        if (kernel_type == 0) {
            o += transposed<T>{ w } * pts[p];
        } else {
            T temp{ 0.0 };
            for (std::size_t i = 0; i < nsv; ++i) {
                temp += alpha[i] * kf(kernel_type, degree, gamma, coef0, svs[i], pts[p]);
            }
            o += temp;
        }
        out[p] = o;
    }
}

}  // namespace

#define REF_DEFINE(SUF, T)                                                                                                                         \
    extern "C" T ref_kernel_function_##SUF(int kt, int degree, T gamma, T coef0, const T *xi, const T *xj, std::size_t d) {                        \
        return kf<T>(kt, degree, gamma, coef0, std::vector<T>(xi, xi + d), std::vector<T>(xj, xj + d));                                            \
    }                                                                                                                                              \
    extern "C" void ref_q_##SUF(int kt, int degree, T gamma, T coef0, const T *X, std::size_t N, std::size_t d, T *q) {                            \
        const std::vector<T> r = gen_q<T>(kt, degree, gamma, coef0, to_rows(X, N, d));                                                             \
        std::copy(r.begin(), r.end(), q);                                                                                                          \
    }                                                                                                                                              \
    extern "C" void ref_matvec_##SUF(int kt, int degree, T gamma, T coef0, const T *X, std::size_t N, std::size_t d, const T *q, const T *dvec,    \
                                     T *ret, T QA_cost, T cost, T add) {                                                                           \
        const std::vector<T> qv(q, q + N - 1), dv(dvec, dvec + N - 1);                                                                             \
        std::vector<T> rv(ret, ret + N - 1);                                                                                                       \
        run_kernel<T>(kt, degree, gamma, coef0, cost, qv, rv, dv, to_rows(X, N, d), QA_cost, add);                                                 \
        std::copy(rv.begin(), rv.end(), ret);                                                                                                      \
    }                                                                                                                                              \
    extern "C" void ref_matvec_sampled_rows_##SUF(int kt, int degree, T gamma, T coef0, const T *X, std::size_t N, std::size_t d, const T *q, const T *dvec, T QA_cost, T cost,   \
                                                  T add, const std::uint64_t *rows, std::size_t nrows, T *out) {                                   \
        sampled_rows<T>(kt, degree, gamma, coef0, cost, std::vector<T>(q, q + N - 1), std::vector<T>(dvec, dvec + N - 1), to_rows(X, N, d), QA_cost, add, rows, nrows, out);   \
    }                                                                                                                                              \
    extern "C" int ref_solve_##SUF(int kt, int degree, T gamma, T coef0, T cost, const T *X, std::size_t N, std::size_t d, const T *y, T eps,      \
                                   std::uint64_t max_iter, T *alpha, T *rho, oracle_cg_info *info, double *trace, std::size_t cap) {               \
        return solve<T>(kt, degree, gamma, coef0, cost, X, N, d, y, eps, max_iter, alpha, rho, info, trace, cap);                                  \
    }                                                                                                                                              \
    extern "C" void ref_calculate_w_##SUF(const T *sv, std::size_t nsv, std::size_t d, const T *alpha, T *w) { calc_w<T>(sv, nsv, d, alpha, w); } \
    extern "C" void ref_predict_values_##SUF(int kt, int degree, T gamma, T coef0, const T *sv, std::size_t nsv, std::size_t d, const T *alpha,    \
                                             T rho, T *w_inout, int *w_valid, const T *points, std::size_t npoints, T *out) {                      \
        predict<T>(kt, degree, gamma, coef0, sv, nsv, d, alpha, rho, w_inout, w_valid, points, npoints, out);                                      \
    }

REF_DEFINE(f32, float)
REF_DEFINE(f64, double)
