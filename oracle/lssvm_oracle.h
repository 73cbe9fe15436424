/*
 * lssvm_oracle.h -- CPU ORACLE for the LS-SVM CG hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's OpenMP backend for the one path this repository
 * accelerates.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it; the product library (libplssvm_amd.so) never links, loads or calls anything in oracle/.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_vs_ref.py, run in the build
 * container) against oracle/_ref/liblssvm_ref.so, which is compiled from the reference's own
 * src/plssvm/backends/OpenMP/{svm_kernel,q_kernel}.cpp where they lie under /root/reference, and against
 * the golden vectors in tests/golden/ that were generated from that same library.
 *
 * All citations are relative to /root/reference.
 *
 * kernel_type: 0 linear, 1 polynomial, 2 rbf   (include/plssvm/kernel_function_types.hpp:31-38)
 * X is row-major N x d (the reference's std::vector<std::vector<T>>, flattened); n = N - 1 = "dept".
 */
#ifndef LSSVM_ORACLE_H
#define LSSVM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint64_t iterations;   /* min(iter + 1, max_iter), csvm.cpp:169 */
    double delta;          /* final residuum r^T r, csvm.cpp:171 */
    double delta0;         /* initial residuum, csvm.cpp:108 */
    double target;         /* eps * eps * delta0, csvm.cpp:172 */
    double avg_iter_ms;    /* wall clock per CG iteration, csvm.cpp:114-122 */
    double total_ms;       /* wall clock of the whole solve */
} oracle_cg_info;

/* kernel_function<kernel>(xi, xj, args...)   include/plssvm/kernel_function_types.hpp:75-97 */
float  oracle_kernel_function_f32(int kernel_type, int degree, float gamma, float coef0, const float *xi, const float *xj, size_t d);
double oracle_kernel_function_f64(int kernel_type, int degree, double gamma, double coef0, const double *xi, const double *xj, size_t d);

/* device_kernel_q_{linear,polynomial,rbf}    src/plssvm/backends/OpenMP/q_kernel.cpp:18-55 ; q has N-1 entries */
void oracle_q_f32(int kernel_type, int degree, float gamma, float coef0, const float *X, size_t N, size_t d, float *q);
void oracle_q_f64(int kernel_type, int degree, double gamma, double coef0, const double *X, size_t N, size_t d, double *q);

/* device_kernel_{linear,polynomial,rbf}      src/plssvm/backends/OpenMP/svm_kernel.cpp:22-82
 * ret[0..n) += add * Abar * dvec, lower triangle + mirrored atomic update; cost is already 1/C (csvm.cpp:297). */
void oracle_matvec_f32(int kernel_type, int degree, float gamma, float coef0, const float *X, size_t N, size_t d,
                       const float *q, const float *dvec, float *ret, float QA_cost, float cost, float add);
void oracle_matvec_f64(int kernel_type, int degree, double gamma, double coef0, const double *X, size_t N, size_t d,
                       const double *q, const double *dvec, double *ret, double QA_cost, double cost, double add);

/* Same result computed row by row over the FULL square (no symmetry, no atomics), rows [row_begin,row_end) only.
 * Not in the reference: it is the CPU statement of this repository's row-block sharding and is used
 * by the world_size-2 gloo tests and for sub-sampled parity checks at sizes where the triangle is too slow. */
void oracle_matvec_rows_f32(int kernel_type, int degree, float gamma, float coef0, const float *X, size_t N, size_t d,
                            const float *q, const float *dvec, float *ret, float QA_cost, float cost, float add,
                            size_t row_begin, size_t row_end);
void oracle_matvec_rows_f64(int kernel_type, int degree, double gamma, double coef0, const double *X, size_t N, size_t d,
                            const double *q, const double *dvec, double *ret, double QA_cost, double cost, double add,
                            size_t row_begin, size_t row_end);

/* openmp::csvm::solve_system_of_linear_equations_impl   src/plssvm/backends/OpenMP/csvm.cpp:71-183
 * alpha has N entries (alpha[N-1] = -sum), rho = -bias.  delta_trace (may be NULL) receives delta after every
 * iteration (at most trace_cap entries).  Returns 0 on success, <0 for a violated precondition (csvm.cpp:73-78). */
int oracle_solve_f32(int kernel_type, int degree, float gamma, float coef0, float cost, const float *X, size_t N, size_t d,
                     const float *y, float eps, uint64_t max_iter, float *alpha, float *rho, oracle_cg_info *info,
                     double *delta_trace, size_t trace_cap);
int oracle_solve_f64(int kernel_type, int degree, double gamma, double coef0, double cost, const double *X, size_t N, size_t d,
                     const double *y, double eps, uint64_t max_iter, double *alpha, double *rho, oracle_cg_info *info,
                     double *delta_trace, size_t trace_cap);

/* openmp::csvm::calculate_w                  src/plssvm/backends/OpenMP/csvm.cpp:255-280 */
void oracle_calculate_w_f32(const float *sv, size_t nsv, size_t d, const float *alpha, float *w);
void oracle_calculate_w_f64(const double *sv, size_t nsv, size_t d, const double *alpha, double *w);

/* openmp::csvm::predict_values_impl          src/plssvm/backends/OpenMP/csvm.cpp:188-227
 * w_inout: d entries; *w_valid != 0 means w already holds the normal vector (linear kernel only). */
void oracle_predict_values_f32(int kernel_type, int degree, float gamma, float coef0, const float *sv, size_t nsv, size_t d,
                               const float *alpha, float rho, float *w_inout, int *w_valid,
                               const float *points, size_t npoints, float *out);
void oracle_predict_values_f64(int kernel_type, int degree, double gamma, double coef0, const double *sv, size_t nsv, size_t d,
                               const double *alpha, double rho, double *w_inout, int *w_valid,
                               const double *points, size_t npoints, double *out);

int oracle_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
