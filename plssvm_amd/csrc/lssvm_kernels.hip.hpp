/*
 * lssvm_kernels.hip.hpp -- the O(n) and set-up kernels of the LS-SVM CG hot path on gfx950 (CDNA4 / MI355X); the O(n^2 d) tile
 * kernels live in lssvm_tile_f32.hip.hpp / lssvm_tile_f64.hip.hpp.
 *
 * What is computed (citations relative to the reference tree SC-SGS/PLSSVM):
 *   the implicit matrix-vector product  ret += add * Abar * d,  Abar_ij = k(x_i,x_j) + delta_ij/C + QA_cost - q_i - q_j
 *   (src/plssvm/backends/OpenMP/svm_kernel.cpp:33-54, include/plssvm/backends/HIP/svm_kernel.hip.hpp:38-270),
 *   the q vector (q_kernel.cpp:18-55 / HIP/q_kernel.hip.hpp:33-85) and the BLAS-1 of the CG loop (csvm.cpp:101-163).
 *
 * How:  Abar * d = K d + d/C + (QA_cost*S - q.d) 1 - S q   with S = sum(d)  (rank-1 terms peeled off).
 *   Only K d is O(n^2 d): the tile kernels write per-work-item row slabs and per-tile column records (symmetric variant), the
 *   kernels here add them in a FIXED order (no atomics anywhere), apply the rank-1 terms in double and run the CG vector
 *   updates with 256 fixed partial sums per dot product, so that every scalar is bit-equal on all ranks of a sharded solve.
 *   Also here: (d_j | c_j) record packing for the LDS-DMA, the k-interleave of the fp32 data, q, centring / scaling, norms.
 */
#pragma once

#include "lssvm_device_common.hip.hpp"

/* The kernels that are not templates are DEFINED by the one translation unit that launches them: lssvm_problem.hip (data set-up, record packing, k_finish2)
 * defines LSSVM_KERNELS_SETUP before it includes this header, lssvm_solver.hip (k_finish_delta) LSSVM_KERNELS_CG.  The templates are instantiated where they are used. */

namespace lssvm {


/* out_i = v_i * s_i (row-scaled f16 planes of the linear kernel: s = 2^-k_i per row; in place where out == v) */
#ifdef LSSVM_KERNELS_SETUP
__global__ void k_scale_vector(const float *v, const float *__restrict__ s, int n, float *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v[i] * s[i];
}
#endif  // LSSVM_KERNELS_SETUP

/* dc[jt][0..127] = d of tile jt, dc[jt][128..255] = c (rbf: -|x_j|^2/2, else unused): one 1-KiB LDS-DMA record per tile */
/* folded != 0 (rbf on the 16x16x32 bf16x6 kernels): dc[jt][0..127] = 2^c_j * d_j, dc[jt][128..255] = 2^c_j -- the tile kernel then starts its
 * accumulators from c_i alone (as the C operand of the first MFMA) and evaluates K_ij d_j = 2^acc * (2^c_j d_j); used only while
 * |c| <= 100, so neither factor leaves the fp32 range */
/* folded == 2 (rbf on grid planes, KT_RBFG): dc[jt][0..127] = E_j * d_j, dc[jt][128..255] = cc_j = sigma^2 ch_j (the exact start value); E_j = efac[j] */
#ifdef LSSVM_KERNELS_SETUP
__global__ void k_pack_dc(const float *__restrict__ dvec, const float *__restrict__ cc, int ncols_padded, float *__restrict__ dc, int folded, float *__restrict__ zero, int nzero,
                          const float *__restrict__ efac) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nzero) zero[j] = 0.0f;  // (the symmetric variant ADDS into K*v: the vector is cleared here instead of by a memset of its own)
    if (j >= ncols_padded) return;
    const int jt = j >> 7, l = j & 127;
    const float c = (cc != nullptr) ? cc[j] : 0.0f;
    if (folded == 2) {
        dc[static_cast<size_t>(jt) * 256 + l] = efac[j] * dvec[j];
        dc[static_cast<size_t>(jt) * 256 + 128 + l] = c;
    } else if (folded) {
        const float e = __builtin_amdgcn_exp2f(c);
        dc[static_cast<size_t>(jt) * 256 + l] = e * dvec[j];
        dc[static_cast<size_t>(jt) * 256 + 128 + l] = e;
    } else {
        dc[static_cast<size_t>(jt) * 256 + l] = dvec[j];
        dc[static_cast<size_t>(jt) * 256 + 128 + l] = c;
    }
}
#endif  // LSSVM_KERNELS_SETUP

/* operand planes [nplanes][rows][ldx16] (row-major, rows a multiple of 16, ldx16 of 64) -> the same data with every block of 16 rows x 32 features stored
 * as ONE MFMA A fragment (64 lanes x 8 halfs, lane 16 g + r holding features 8 g .. 8 g + 7 of row r), ordered
 * [plane][ldx16 / 64 chunks][rows / 16 blocks][2 k32 steps][64 lanes][8]: the 64-feature chunk OUTERMOST, so that the sixteen row blocks a workgroup
 * loads for one chunk are 32 KiB contiguous (with the row block outermost they lay a power of two apart whenever ldx16 is one: every wave on the same
 * L2 channels -- 100 000 x 385 measured 16 % slower that way).  One thread per 16-byte piece. */
#ifdef LSSVM_KERNELS_SETUP
__global__ void k_planes_fragment_major(const uint16_t *__restrict__ src, size_t plane_elems, int rows, int ldx16, int nplanes, uint16_t *__restrict__ dst) {
    const size_t piece = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t per_plane = static_cast<size_t>(rows) * (ldx16 / 8);
    if (piece >= per_plane * nplanes) return;
    const size_t pl = piece / per_plane;
    const size_t in_plane = piece - pl * per_plane;
    const int row = static_cast<int>(in_plane / (ldx16 / 8));
    const int f8 = static_cast<int>(in_plane - static_cast<size_t>(row) * (ldx16 / 8));
    const int rb = row >> 4, r = row & 15, c64 = f8 >> 3, kk = (f8 >> 2) & 1, g = f8 & 3;
    const f32x4 v = *reinterpret_cast<const f32x4 *>(src + pl * plane_elems + static_cast<size_t>(row) * ldx16 + 8 * f8);
    *reinterpret_cast<f32x4 *>(dst + pl * plane_elems + (static_cast<size_t>(c64) * (rows / 16) + rb) * 1024 + kk * 512 + (16 * g + r) * 8) = v;
}
#endif  // LSSVM_KERNELS_SETUP

/* fp64 data [rows][ldx] (row-major, rows a multiple of 16, ldx of 16) -> the same data with every block of 16 rows x 4 features stored as ONE A fragment of
 * v_mfma_f64_16x16x4 (64 lanes, lane 16 q + r = row r, feature q), ordered [ldx / 16 chunks][rows / 16 blocks][4 k-steps][64 lanes]: the 16-feature
 * chunk outermost, so that the eight row blocks a workgroup loads for one chunk are 16 KiB contiguous.  One thread per element.  (The
 * panels-inside-a-sub-tile kernel re-loads its row fragments at every sub-tile and panel -- 22 % of its time, row-major: 32 bytes of each of 16 lines
 * per load instruction.) */
#ifdef LSSVM_KERNELS_SETUP
__global__ void k_rows_fragment_major_f64(const double *__restrict__ src, int rows, int ldx, double *__restrict__ dst) {
    const size_t e = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= static_cast<size_t>(rows) * ldx) return;
    const int row = static_cast<int>(e / ldx), f = static_cast<int>(e - static_cast<size_t>(row) * ldx);
    const int rb = row >> 4, r = row & 15, chunk = f >> 4, s = (f >> 2) & 3, q = f & 3;
    dst[((static_cast<size_t>(chunk) * (rows / 16) + rb) * 4 + s) * 64 + 16 * q + r] = src[e];
}
#endif  // LSSVM_KERNELS_SETUP

/* in place: the features of every aligned group of 8 are reordered to 0,2,4,6,1,3,5,7 (fp32 HBM layout, see above) */
#ifdef LSSVM_KERNELS_SETUP
__global__ void k_interleave_features(float *__restrict__ X, size_t ngroups) {
    const size_t g = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    f32x4 *p = reinterpret_cast<f32x4 *>(X + 8 * g);
    const f32x4 lo = p[0], hi = p[1];
    p[0] = f32x4{ lo.x, lo.z, hi.x, hi.z };
    p[1] = f32x4{ lo.y, lo.w, hi.y, hi.w };
}
#endif  // LSSVM_KERNELS_SETUP


/* fp64 records of the v2 kernel: per 64-column sub-tile st: dc[st][0..63] = d, dc[st][64..127] = c */
#ifdef LSSVM_KERNELS_SETUP
__global__ void k_pack_dc_f64(const double *__restrict__ dvec, const double *__restrict__ cc, int ncols_padded, double *__restrict__ dc, double *__restrict__ zero, int nzero) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nzero) zero[j] = 0.0;
    if (j >= ncols_padded) return;
    const int st = j >> 6, l = j & 63;
    dc[static_cast<size_t>(st) * 128 + l] = dvec[j];
    dc[static_cast<size_t>(st) * 128 + 64 + l] = (cc != nullptr) ? cc[j] : 0.0;
}
#endif  // LSSVM_KERNELS_SETUP

/* =====================================================================================================================
 * O(n) / O(n d) helper kernels
 * ===================================================================================================================== */

/* deterministic block reduction of `NV` doubles per thread; result valid in thread 0 */
template <int NV>
__device__ __forceinline__ void block_reduce(double (&v)[NV], double *lds /* [NV][4] for 256 threads */) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
    }
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) lds[k * 4 + wave] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = (lds[k * 4 + 0] + lds[k * 4 + 1]) + (lds[k * 4 + 2] + lds[k * 4 + 3]);
    }
}

/* The two sums a PREVIOUS kernel left as RED_BLOCKS x 2 partials, reduced by every block of the consuming kernel for itself, in the fixed tree
 * order of k_finish2 (bit-identical to it, whichever block asks): the single-block launch that used to stand between the two kernels -- 5 us of
 * a 50 000-point CG iteration each, and there were three per iteration -- is gone, and no block ever waits for another. */
__device__ __forceinline__ void finish2_in_block(const double *__restrict__ part, double *lds /* [8] */, double *tot /* [2] */, double &s0, double &s1) {
    double acc[2] = { part[threadIdx.x * 2 + 0], part[threadIdx.x * 2 + 1] };
    block_reduce<2>(acc, lds);
    if (threadIdx.x == 0) {
        tot[0] = acc[0];
        tot[1] = acc[1];
    }
    __syncthreads();
    s0 = tot[0];
    s1 = tot[1];
}

/* part[b][0] = sum v, part[b][1] = sum q*v over the block's grid-stride slice */
template <typename T>
__global__ __launch_bounds__(RED_THREADS) void k_sum_and_qdot(const T *__restrict__ v, const T *__restrict__ q, int n, double *__restrict__ part) {
    __shared__ double lds[8];
    double acc[2] = { 0.0, 0.0 };
    for (int i = blockIdx.x * RED_THREADS + threadIdx.x; i < n; i += RED_BLOCKS * RED_THREADS) {
        const double vi = static_cast<double>(v[i]);
        acc[0] += vi;
        acc[1] += vi * static_cast<double>(q[i]);
    }
    block_reduce<2>(acc, lds);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = acc[0];
        part[blockIdx.x * 2 + 1] = acc[1];
    }
}

/* single block: out[slot0] = sum part[.][0], out[slot1] = sum part[.][1]  (fixed tree order) */
#ifdef LSSVM_KERNELS_SETUP
__global__ __launch_bounds__(RED_THREADS) void k_finish2(const double *__restrict__ part, double *__restrict__ sc, int slot0, int slot1) {
    __shared__ double lds[8];
    double acc[2] = { part[threadIdx.x * 2 + 0], part[threadIdx.x * 2 + 1] };
    block_reduce<2>(acc, lds);
    if (threadIdx.x == 0) {
        sc[slot0] = acc[0];
        if (slot1 >= 0) sc[slot1] = acc[1];
    }
}
#endif  // LSSVM_KERNELS_SETUP

/* Kv[row_begin + i] = sum over column chunks of partial[c][i], chunks in ascending order (rows of this device only) */
template <typename T>
__global__ void k_reduce_partials(const T *__restrict__ partial, long part_stride, int num_jc, int row_begin, int nrows, T *__restrict__ Kv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrows) {
        T s = partial[i];
        for (int c = 1; c < num_jc; ++c) s += partial[static_cast<size_t>(c) * part_stride + i];
        Kv[row_begin + i] = s;
    }
}

/* SYM: Kv[column c] += sum over the row blocks ib (of this device) below column record c of colslab[(ib, c)], in a FIXED order.
 * Records are W columns wide, SUB = 128 / W records per 128-column tile (fp32: W = 128, fp64: W = 64); row block ib owns the
 * SUB * ib records below its diagonal tile, packed triangularly.  One block of 1024 threads per record column: 1024 / W groups
 * walk the row blocks with stride 1024 / W (four independent partial sums each, for memory-level parallelism), then the groups'
 * sums are added in group order -- the order depends only on the shape, never on timing. */
template <typename T, int W>
__global__ __launch_bounds__(1024) void k_reduce_colslab(const T *__restrict__ colslab, long pair_origin, int ib_begin, int ib_end, int ib_step, T *__restrict__ Kv,
                                                         const T *__restrict__ col_factor = nullptr) {  // col_factor (rbf on grid planes): the column's folded factor E_j, applied to the summed records
    constexpr int SUB = TILE / W;
    constexpr int G = 1024 / W;
    __shared__ T red[G][W];
    const int c = blockIdx.x;
    const int l = threadIdx.x % W;
    const int g = threadIdx.x / W;
    // ib_step 2 (256-row workgroups): the records of a block pair are those of its odd block; ib_begin is even
    const int first = ib_step == 2 ? (max(c / SUB + 1, ib_begin) | 1) : max(c / SUB + 1, ib_begin);
    auto rec = [&](int ib) { return colslab[(SUB * (static_cast<long>(ib) * (ib - 1) / 2 - pair_origin) + c) * W + l]; };
    T s0 = T(0), s1 = T(0), s2 = T(0), s3 = T(0);
    const int GS = G * ib_step;
    int ib = first + g * ib_step;
    for (; ib + 3 * GS < ib_end; ib += 4 * GS) {
        s0 += rec(ib);
        s1 += rec(ib + GS);
        s2 += rec(ib + 2 * GS);
        s3 += rec(ib + 3 * GS);
    }
    for (; ib < ib_end; ib += GS) s0 += rec(ib);
    red[g][l] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0) {
        T s = red[0][l];
#pragma unroll
        for (int k = 1; k < G; ++k) s += red[k][l];
        if (col_factor != nullptr) s *= col_factor[c * W + l];
        Kv[c * W + l] += s;
    }
}

/* SYM: Kv[row_begin + i] = sum of the row slabs of the column chunks that exist for the row's block (the chunks that begin at or before tile ib) */
template <typename T>
__global__ void k_reduce_partials_sym(const T *__restrict__ partial, long part_stride, int jc_tiles, int jc_head_tiles, int jc_head_count, int ib_begin, int nrows, T *__restrict__ Kv,
                                      int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrows) {
        const int ib = ib_begin + i / TILE;
        const int nchunks = chunks_upto(ib, jc_tiles, jc_head_tiles, jc_head_count);
        T s = partial[i];
        for (int c = 1; c < nchunks; ++c) s += partial[static_cast<size_t>(c) * part_stride + i];
        const int row = ib_begin * TILE + i;
        Kv[row] = accumulate ? Kv[row] + s : s;
    }
}

/* (Abar v)_i = Kv_i + v_i/C + (QA_cost*S - q.v) - S*q_i, evaluated in double */
template <typename T>
__device__ __forceinline__ double abar_row(const T *Kv, const T *v, const T *q, int i, double inv_cost, double QA_cost, double S, double QV) {
    return static_cast<double>(Kv[i]) + static_cast<double>(v[i]) * inv_cost + (QA_cost * S - QV) - S * static_cast<double>(q[i]);
}

/* ret_i += add * (Abar v)_i       (run_device_kernel semantics, csvm.cpp:283-306) */
template <typename T>
__global__ void k_apply_ret(const T *__restrict__ Kv, const T *__restrict__ v, const T *__restrict__ q, const double *__restrict__ sc, int n,
                            double inv_cost, double QA_cost, double add, T *__restrict__ ret) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double val = abar_row(Kv, v, q, i, inv_cost, QA_cost, sc[SC_S], sc[SC_QD]);
        ret[i] = static_cast<T>(static_cast<double>(ret[i]) + add * val);
    }
}

/* Ad_i = (Abar d)_i ; part[b][0] = sum d_i Ad_i     (csvm.cpp:131-135).  S = sum d and q.d come as k_update_d's partials (`part_d`). */
template <typename T>
__global__ __launch_bounds__(RED_THREADS) void k_Ad_and_dAd(const T *__restrict__ Kv, const T *__restrict__ d, const T *__restrict__ q,
                                                            const double *__restrict__ part_d, double *__restrict__ sc, int n, double inv_cost, double QA_cost,
                                                            T *__restrict__ Ad, double *__restrict__ part) {
    __shared__ double lds[8];
    __shared__ double tot[2];
    double S, QD;
    finish2_in_block(part_d, lds, tot, S, QD);
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // (for the record: cg_finish compares the shards' scalars)
        sc[SC_S] = S;
        sc[SC_QD] = QD;
    }
    double acc[2] = { 0.0, 0.0 };
    for (int i = blockIdx.x * RED_THREADS + threadIdx.x; i < n; i += RED_BLOCKS * RED_THREADS) {
        const T adi = static_cast<T>(abar_row(Kv, d, q, i, inv_cost, QA_cost, S, QD));
        Ad[i] = adi;
        acc[0] += static_cast<double>(d[i]) * static_cast<double>(adi);
    }
    block_reduce<2>(acc, lds);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = acc[0];
        part[blockIdx.x * 2 + 1] = 0.0;
    }
}

/* alpha_cd = delta / (d^T Ad) from k_Ad_and_dAd's partials (`part_dad`; csvm.cpp:135) ; x += alpha_cd d ; r -= alpha_cd Ad ; part = sum r^2
 * (csvm.cpp:138, :148, :153).  The scalar is rounded to T first, as the reference's real_type alpha_cd is. */
template <typename T>
__global__ __launch_bounds__(RED_THREADS) void k_update_x_r(T *__restrict__ x, T *__restrict__ r, const T *__restrict__ d, const T *__restrict__ Ad,
                                                            const double *__restrict__ part_dad, double *__restrict__ sc, int n, int update_r, double *__restrict__ part) {
    __shared__ double lds[8];
    __shared__ double tot[2];
    double dAd, unused;
    finish2_in_block(part_dad, lds, tot, dAd, unused);
    const double alpha_cd = sc[SC_DELTA] / dAd;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc[SC_DAD] = dAd;
        sc[SC_ALPHA] = alpha_cd;
    }
    const T alpha = static_cast<T>(alpha_cd);
    double acc[2] = { 0.0, 0.0 };
    for (int i = blockIdx.x * RED_THREADS + threadIdx.x; i < n; i += RED_BLOCKS * RED_THREADS) {
        x[i] += alpha * d[i];
        if (update_r) {
            const T ri = r[i] - alpha * Ad[i];
            r[i] = ri;
            acc[0] += static_cast<double>(ri) * static_cast<double>(ri);
        }
    }
    block_reduce<2>(acc, lds);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = acc[0];
        part[blockIdx.x * 2 + 1] = 0.0;
    }
}

/* (Round 5 also let the block of k_update_x_r that finishes last do k_finish_delta's work -- a ticket, two device-scope fences, a second tree: 2 us SLOWER per
 * iteration than the single-block launch it saves, at every size (profiles/r05_ab_fused_chain.log): a device-scope release writes an XCD's L2 back.  Withdrawn.) */
/* r_i = b_i - (Abar x)_i ; part = sum r^2       (csvm.cpp:101-107 and the refresh :140-145).  Uses SC_SUMX / SC_QX. */
template <typename T>
__global__ __launch_bounds__(RED_THREADS) void k_residual(const T *__restrict__ Kv, const T *__restrict__ x, const T *__restrict__ q, const T *__restrict__ b,
                                                          const double *__restrict__ sc, int n, double inv_cost, double QA_cost, T *__restrict__ r,
                                                          double *__restrict__ part) {
    __shared__ double lds[8];
    const double S = sc[SC_SUMX], QX = sc[SC_QX];
    double acc[2] = { 0.0, 0.0 };
    for (int i = blockIdx.x * RED_THREADS + threadIdx.x; i < n; i += RED_BLOCKS * RED_THREADS) {
        const T ri = static_cast<T>(static_cast<double>(b[i]) - abar_row(Kv, x, q, i, inv_cost, QA_cost, S, QX));
        r[i] = ri;
        acc[0] += static_cast<double>(ri) * static_cast<double>(ri);
    }
    block_reduce<2>(acc, lds);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = acc[0];
        part[blockIdx.x * 2 + 1] = 0.0;
    }
}

/* delta_old = delta ; delta = sum r^2 ; beta = delta / delta_old ; publish delta to the host-mapped word  (csvm.cpp:152-161) */
#ifdef LSSVM_KERNELS_CG
__global__ __launch_bounds__(RED_THREADS) void k_finish_delta(const double *__restrict__ part, double *__restrict__ sc, double *__restrict__ host_delta, int is_initial) {
    __shared__ double lds[8];
    double acc[2] = { part[threadIdx.x * 2 + 0], 0.0 };
    block_reduce<2>(acc, lds);
    if (threadIdx.x == 0) {
        const double delta_old = sc[SC_DELTA];
        sc[SC_DELTA_OLD] = delta_old;
        sc[SC_DELTA] = acc[0];
        if (is_initial) {
            sc[SC_DELTA0] = acc[0];
            sc[SC_BETA] = 0.0;
        } else {
            sc[SC_BETA] = acc[0] / delta_old;
        }
        *host_delta = acc[0];
    }
}
#endif  // LSSVM_KERNELS_CG

__device__ __forceinline__ void pack_dc_entry(const PackDc<float> &pk, int j, float dj) {  // (the statements of k_pack_dc: the same bits)
    const int jt = j >> 7, l = j & 127;
    const float c = (pk.cc != nullptr) ? pk.cc[j] : 0.0f;
    if (pk.folded == 2) {
        pk.dc[static_cast<size_t>(jt) * 256 + l] = pk.efac[j] * dj;
        pk.dc[static_cast<size_t>(jt) * 256 + 128 + l] = c;
    } else if (pk.folded) {
        const float e = __builtin_amdgcn_exp2f(c);
        pk.dc[static_cast<size_t>(jt) * 256 + l] = e * dj;
        pk.dc[static_cast<size_t>(jt) * 256 + 128 + l] = e;
    } else {
        pk.dc[static_cast<size_t>(jt) * 256 + l] = dj;
        pk.dc[static_cast<size_t>(jt) * 256 + 128 + l] = c;
    }
}
__device__ __forceinline__ void pack_dc_entry(const PackDc<double> &pk, int j, double dj) {  // (k_pack_dc_f64)
    const int st = j >> 6, l = j & 63;
    pk.dc[static_cast<size_t>(st) * 128 + l] = dj;
    pk.dc[static_cast<size_t>(st) * 128 + 64 + l] = (pk.cc != nullptr) ? pk.cc[j] : 0.0;
}

/* d = beta d + r ; part = (sum d, sum q d) for the next matvec     (csvm.cpp:163); with pk.dc: the records of that matvec too (d is exactly zero beyond n) */
template <typename T>
__global__ __launch_bounds__(RED_THREADS) void k_update_d(T *__restrict__ d, const T *__restrict__ r, const T *__restrict__ q, const double *__restrict__ sc,
                                                          int n, int copy_only, double *__restrict__ part, const PackDc<T> pk) {
    __shared__ double lds[8];
    const T beta = static_cast<T>(sc[SC_BETA]);
    double acc[2] = { 0.0, 0.0 };
    const int end = pk.dc != nullptr ? max(n, max(pk.ncols, pk.nzero)) : n;
    for (int i = blockIdx.x * RED_THREADS + threadIdx.x; i < end; i += RED_BLOCKS * RED_THREADS) {
        T di = T(0);
        if (i < n) {
            di = copy_only ? r[i] : beta * d[i] + r[i];
            d[i] = di;
            acc[0] += static_cast<double>(di);
            acc[1] += static_cast<double>(di) * static_cast<double>(q[i]);
        }
        if (pk.dc != nullptr) {
            if (i < pk.nzero) pk.zero[i] = T(0);
            if (i < pk.ncols) pack_dc_entry(pk, i, di);
        }
    }
    block_reduce<2>(acc, lds);
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2 + 0] = acc[0];
        part[blockIdx.x * 2 + 1] = acc[1];
    }
}

template <typename T>
__global__ void k_fill(T *__restrict__ v, int n, T value) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = value;
}

/* b_i = y_i - y_last     (csvm.cpp:89-91) */
template <typename T>
__global__ void k_make_b(const T *__restrict__ y, int n, T *__restrict__ b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = y[i] - y[n];
}

/* q_i = k(x_i, x_last): ONE THREAD PER ROW, sequential fma chain over the features in ascending order, exactly the
 * reference's operators.hpp:117-126 / :161-171 chain => q is bit-identical to the OpenMP backend for the linear kernel
 * and differs only through pow/exp otherwise.  O(N d), once per solve (q_kernel.cpp:18-55, HIP/q_kernel.hip.hpp:33-85). */
template <int KT, typename T>
__global__ void k_q(const T *__restrict__ X, int ldx, int dfeat, int n, const T *__restrict__ xlast, int degree, T gamma, T coef0, T *__restrict__ q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const T *xi = X + static_cast<size_t>(i) * ldx;
    T val = T(0);
    for (int f = 0; f < dfeat; ++f) {
        if constexpr (KT == KT_RBF) {
            const T diff = xi[f] - xlast[f];
            val = fma(diff, diff, val);
        } else {
            val = fma(xi[f], xlast[f], val);
        }
    }
    if constexpr (KT == KT_LINEAR) {
        q[i] = val;
    } else if constexpr (KT == KT_POLY) {
        q[i] = ipow(fma(gamma, val, coef0), degree);
    } else {
        q[i] = exp(-gamma * val);
    }
}

/* column sums in double, deterministic two-stage: stage 1 = one block per 256-row slab */
template <typename T>
__global__ void k_colsum_stage1(const T *__restrict__ X, int ldx, int nrows, int rows_per_block, double *__restrict__ part /* [gridDim.x][ldx] */) {
    const int f = threadIdx.x + blockIdx.y * blockDim.x;
    if (f >= ldx) return;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(r0 + rows_per_block, nrows);
    double s = 0.0;
    for (int i = r0; i < r1; ++i) s += static_cast<double>(X[static_cast<size_t>(i) * ldx + f]);
    part[static_cast<size_t>(blockIdx.x) * ldx + f] = s;
}
template <typename T>
__global__ void k_colsum_stage2(const double *__restrict__ part, int nblocks, int ldx, int nrows, T *__restrict__ mean) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= ldx) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += part[static_cast<size_t>(b) * ldx + f];
    mean[f] = static_cast<T>(s / static_cast<double>(nrows));
}
/* X[i][f] = (X[i][f] - mean[f]) * scale for the valid rows / features only (padding stays exactly zero) */
template <typename T>
__global__ void k_center(T *__restrict__ X, int ldx, int dfeat, int nrows, const T *__restrict__ mean, T scale) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (f < dfeat && i < nrows) X[static_cast<size_t>(i) * ldx + f] = (X[static_cast<size_t>(i) * ldx + f] - (mean != nullptr ? mean[f] : T(0))) * scale;
}
/* sq_i = |x_i - mean|^2 in double over the valid features (one wave per row, coalesced); the data is not modified */
template <typename T>
__global__ void k_centred_sqnorm(const T *__restrict__ X, int ldx, int dfeat, int nrows, const T *__restrict__ mean, double *__restrict__ sq) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const T *x = X + static_cast<size_t>(row) * ldx;
    double s = 0.0;
    for (int f = lane; f < dfeat; f += 64) {
        const double dlt = static_cast<double>(x[f]) - static_cast<double>(mean[f]);
        s += dlt * dlt;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) sq[row] = s;
}

/* c_i = -0.5 * |x_i|^2 (one wave per row, coalesced) */
template <typename T>
__global__ void k_half_neg_norms(const T *__restrict__ X, int ldx, int nrows_total, T *__restrict__ c) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= nrows_total) return;
    const T *x = X + static_cast<size_t>(row) * ldx;
    T s = T(0);
    for (int f = lane; f < ldx; f += 64) s = fma(x[f], x[f], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) c[row] = T(-0.5) * s;
}

/* w[f] = sum_i alpha_i X[i][f]   (calculate_w, csvm.cpp:255-280 / HIP/predict_kernel.hip.hpp:34-45).  Stage 1: block (b, c) sums rows [b R, (b + 1) R) of the features
 * [256 c, 256 c + 256) in double, coalesced across the features; stage 2 adds the blocks of a feature in ascending order -- a fixed order, reproducible run to run. */
template <typename T>
__global__ void k_calculate_w_stage1(const T *__restrict__ X, int ldx, int npoints, int rows_per_block, const T *__restrict__ alpha, double *__restrict__ part /* [gridDim.x][ldx] */) {
    const int f = blockIdx.y * blockDim.x + threadIdx.x;
    if (f >= ldx) return;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, npoints);
    double s = 0.0;
    for (int i = r0; i < r1; ++i) s = fma(static_cast<double>(alpha[i]), static_cast<double>(X[static_cast<size_t>(i) * ldx + f]), s);
    part[static_cast<size_t>(blockIdx.x) * ldx + f] = s;
}
template <typename T>
__global__ void k_calculate_w_stage2(const double *__restrict__ part, int nblocks, int ldx, int dfeat, T *__restrict__ w) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= dfeat) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += part[static_cast<size_t>(b) * ldx + f];
    w[f] = static_cast<T>(s);
}
/* out_p = w . x_p - rho  (linear predict, csvm.cpp:213): L lanes per point, 16-byte loads, the lanes' partial chains added by a butterfly */
template <typename T, int L>
__global__ void k_predict_linear_rows(const T *__restrict__ P, int ldx, int npoints, const T *__restrict__ w, T rho, T *__restrict__ out) {
    constexpr int V = 16 / static_cast<int>(sizeof(T));
    using vec = T __attribute__((ext_vector_type(V)));
    const int sub = threadIdx.x % L;
    const int p = blockIdx.x * (blockDim.x / L) + threadIdx.x / L;
    T s = T(0);
    if (p < npoints) {
        const vec *x = reinterpret_cast<const vec *>(P + static_cast<size_t>(p) * ldx);
        const vec *wv = reinterpret_cast<const vec *>(w);
        for (int k = sub; k < ldx / V; k += L) {
            const vec xv = x[k], ww = wv[k];
#pragma unroll
            for (int e = 0; e < V; ++e) s = fma(ww[e], xv[e], s);
        }
    }
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (p < npoints && sub == 0) out[p] = s - rho;
}
/* out_p = Kv_p - rho */
template <typename T>
__global__ void k_sub_rho(const T *__restrict__ Kv, int n, T rho, T *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = Kv[i] - rho;
}

}  // namespace lssvm
