/*
 * lssvm_oracle.c -- CPU ORACLE for the LS-SVM CG hot path.  TEST INFRASTRUCTURE ONLY (see lssvm_oracle.h).
 *
 * Plain C99 + OpenMP restatement of (citations relative to /root/reference):
 *   src/plssvm/backends/OpenMP/csvm.cpp:71-183, 188-227, 255-280   CG driver, predict_values, calculate_w
 *   src/plssvm/backends/OpenMP/svm_kernel.cpp:22-82                implicit matvec (blocked triangle + atomics)
 *   src/plssvm/backends/OpenMP/q_kernel.cpp:18-55                  q vector
 *   include/plssvm/kernel_function_types.hpp:75-97                 kernel functions
 *   include/plssvm/detail/operators.hpp:117-171                    fma-chained dot / squared distance, sum
 *
 * Parity status: PINNED against oracle/_ref (the reference's own OpenMP kernel TUs compiled in place) and the
 * golden vectors generated from it (tests/golden/make_golden.py).
 *
 * Build: see oracle/Makefile (gcc -std=c99 -O2 -fopenmp -ffp-contract=off; no -ffast-math, matching the reference's
 * default RelWithDebInfo flags, CMakeLists.txt:23).
 */
#define _POSIX_C_SOURCE 200809L
#include "lssvm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double oracle_now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec * 1e3 + (double) ts.tv_nsec * 1e-6;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- float instantiation (std::fma/std::exp/std::pow on float resolve to fmaf/expf/powf) ---- */
#define REAL float
#define FN(name) name##_f32
#define R_FMA fmaf
#define R_EXP expf
#define R_POW powf
#include "lssvm_oracle_impl.inc"
#undef REAL
#undef FN
#undef R_FMA
#undef R_EXP
#undef R_POW

/* ---- double instantiation ---- */
#define REAL double
#define FN(name) name##_f64
#define R_FMA fma
#define R_EXP exp
#define R_POW pow
#include "lssvm_oracle_impl.inc"
#undef REAL
#undef FN
#undef R_FMA
#undef R_EXP
#undef R_POW
