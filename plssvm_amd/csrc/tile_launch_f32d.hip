/*
 * tile_launch_f32d.hip -- instantiates and launches the split tile kernels with 256-row workgroups (lssvm_tile_f32_pair.hip.hpp: eight waves on a
 * block pair, one column stream, in lock step).  Symmetric variant, at most 128 features per pass, both
 * plane kinds (f16x3, bf16x6).  Compiled for gfx950 only.
 */
#include "tile_launch.hip.hpp"

#include "lssvm_tile_f32_pair.hip.hpp"

namespace lssvm {

template <int KT, int NK64, int PL>
static void launch_pair_lag(const TileArgs<float> &a, hipStream_t s) {
    // persistent launch (TileArgs::queue): one workgroup per CU (140 KiB of LDS each), the items drawn from the problem's counters; else one workgroup per item
    const dim3 grid = sym_grid(a, 1), block(PR_THREADS);
#define LSSVM_PAIR_LAG(L)                                                                           \
    case L:                                                                                         \
        ensure_dynamic_lds(tile_matvec_f32_pair<KT, NK64, PL, L>, PR_LDS_BYTES);                    \
        hipLaunchKernelGGL((tile_matvec_f32_pair<KT, NK64, PL, L>), grid, block, PR_LDS_BYTES, s, a); \
        break;
    switch (a.pair_lag) {
#ifdef LSSVM_DEV_SUBSET  // development builds: the lag as a run-time choice (A/B; measured: every lag is SLOWER than lock step)
        LSSVM_PAIR_LAG(1) LSSVM_PAIR_LAG(3) LSSVM_PAIR_LAG(4) LSSVM_PAIR_LAG(5) LSSVM_PAIR_LAG(6) LSSVM_PAIR_LAG(7)
#endif
        LSSVM_PAIR_LAG(0)
        default: throw Error(LSSVM_ERR_INTERNAL, "no 256-row tile kernel for this lag");
    }
#undef LSSVM_PAIR_LAG
}

template <int KT, int PL>
static void launch_pair_kt(const TileArgs<float> &a, hipStream_t s) {
    switch (a.nk64) {
#ifndef LSSVM_DEV_SUBSET
        case 1: launch_pair_lag<KT, 1, PL>(a, s); break;
#endif
        case 2: launch_pair_lag<KT, 2, PL>(a, s); break;
        default: throw Error(LSSVM_ERR_INTERNAL, "no 256-row tile kernel for this number of features");
    }
}

template <int PL>
static void launch_pair(const TileArgs<float> &a, int kernel_type, hipStream_t s) {
    switch (kernel_type) {
        case KT_LINEAR: launch_pair_kt<KT_LINEAR, PL>(a, s); break;
        case KT_POLY:
            if (a.degree == 3) {
                launch_pair_kt<KT_POLY3, PL>(a, s);
            } else if (a.degree == 2) {
                launch_pair_kt<KT_POLY2, PL>(a, s);
            } else {
                throw Error(LSSVM_ERR_INTERNAL, "no 256-row tile kernel for the run-time integer power");  // (Problem<float> does not choose block pairs for it)
            }
            break;
        default:
            if (a.rbf_grid != 0) {  // rbf on grid planes (round 6): f16 planes only
                if constexpr (PL == 2) {
                    launch_pair_kt<KT_RBFG, PL>(a, s);
                } else {
                    throw Error(LSSVM_ERR_INTERNAL, "the grid planes of the rbf kernel are f16 planes");
                }
                break;
            }
            if (a.dc_folded == 0) throw Error(LSSVM_ERR_INTERNAL, "the 256-row rbf kernel needs the folded records");  // (Problem<float> does not choose block pairs otherwise)
            launch_pair_kt<KT_RBFF, PL>(a, s);
            break;
    }
}

/* the rectangular instance (predict_values): polynomial of degree 2 / 3 and rbf with folded records, at most 128 features, both plane kinds */
template <int KT, int PL>
static void launch_rect_kt(const TileArgs<float> &a, hipStream_t s) {
    const dim3 grid = sym_grid(a, 1), block(PR_THREADS);
    switch (a.nk64) {
#ifndef LSSVM_DEV_SUBSET
        case 1:
            ensure_dynamic_lds(tile_matvec_f32_pair_rect<KT, 1, PL>, PR_LDS_BYTES);
            hipLaunchKernelGGL((tile_matvec_f32_pair_rect<KT, 1, PL>), grid, block, PR_LDS_BYTES, s, a);
            break;
#endif
        case 2:
            ensure_dynamic_lds(tile_matvec_f32_pair_rect<KT, 2, PL>, PR_LDS_BYTES);
            hipLaunchKernelGGL((tile_matvec_f32_pair_rect<KT, 2, PL>), grid, block, PR_LDS_BYTES, s, a);
            break;
        default: throw Error(LSSVM_ERR_INTERNAL, "no rectangular 256-row tile kernel for this number of features");
    }
}
template <int PL>
static void launch_rect(const TileArgs<float> &a, int kernel_type, hipStream_t s) {
    switch (kernel_type) {
        case KT_POLY:
            if (a.degree == 3) {
                launch_rect_kt<KT_POLY3, PL>(a, s);
            } else if (a.degree == 2) {
                launch_rect_kt<KT_POLY2, PL>(a, s);
            } else {
                throw Error(LSSVM_ERR_INTERNAL, "no rectangular 256-row tile kernel for the run-time integer power");
            }
            break;
        case KT_RBF:
            if (a.dc_folded == 0) throw Error(LSSVM_ERR_INTERNAL, "the rectangular 256-row rbf kernel needs the folded records");
            launch_rect_kt<KT_RBFF, PL>(a, s);
            break;
        default: throw Error(LSSVM_ERR_INTERNAL, "no rectangular 256-row tile kernel for this kernel function");  // (the linear kernel predicts through w)
    }
}

void launch_pair_tile_kernel(const TileArgs<float> &a, int kernel_type, hipStream_t s) {
    if (a.items == nullptr || a.num_items <= 0) return;
    if (a.Xr16f == nullptr) throw Error(LSSVM_ERR_INTERNAL, "the 256-row tile kernel needs the fragment-major row planes");
    if (a.rect != 0) {
        if (a.planes_f16 != 0) {
            launch_rect<2>(a, kernel_type, s);
        } else {
#ifdef LSSVM_DEV_SUBSET
            throw Error(LSSVM_ERR_INTERNAL, "development build: 256-row kernels for the f16 planes only");
#else
            launch_rect<3>(a, kernel_type, s);
#endif
        }
        LSSVM_HIP_CHECK(hipGetLastError());
        return;
    }
    if (a.planes_f16 != 0) {
        launch_pair<2>(a, kernel_type, s);
    } else {
#ifdef LSSVM_DEV_SUBSET
        throw Error(LSSVM_ERR_INTERNAL, "development build: 256-row kernels for the f16 planes only");
#else
        launch_pair<3>(a, kernel_type, s);
#endif
    }
    LSSVM_HIP_CHECK(hipGetLastError());
}

}  // namespace lssvm

#ifdef LSSVM_ITEM_TRACE
extern "C" int lssvm_debug_set_item_trace(void *device_buffer) {  // measurement builds only: where the work items of the NEXT launches stamp their phases (NULL = off)
    return static_cast<int>(hipMemcpyToSymbol(HIP_SYMBOL(lssvm::lssvm_item_trace), &device_buffer, sizeof(void *)));
}
#endif
