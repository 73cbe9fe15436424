/*
 * lssvm_tile_f64_wide.hip.hpp -- the fp64 tile kernel for rbf / polynomial problems with more than 256 features (the widest row panel
 * tile_matvec_f64_v2 holds in registers).
 *
 * The fp64 counterpart of lssvm_tile_f32_wide.hip.hpp: tile_matvec_f64_v2's pipeline (lssvm_tile_f64.hip.hpp: 64-column sub-tiles, LDS-DMA ring of
 * 64-column x 16-feature chunks gathered into the read-friendly image, hand-over in the middle of a chunk, packed (d_j | c_j) records, fused
 * epilogue, symmetric variant with mirrored column sums) with the feature dimension walked in PANELS of 64 features inside a sub-tile: the
 * accumulators live across the panels, the epilogue runs after the last, and the row panel (64 features: 64 registers -- the budget of two
 * workgroups per CU) is re-loaded from L2 for every (sub-tile, panel), each 16-feature chunk requested as soon as the previous panel is through
 * with it.  The linear kernel does not come here: its Gram matrix is a sum over panels and runs one pass of the v2 kernel per panel
 * (lssvm_problem.hip).  The data is padded to whole panels (padded_features).  A negative polynomial degree stays on the generic kernel.
 * Reference semantics: /root/reference/include/plssvm/backends/HIP/svm_kernel.hip.hpp:129-270 (one code path for any feature count).
 */
#pragma once

#include "lssvm_tile_f64.hip.hpp"

namespace lssvm {

template <int KT, bool SYM>
__global__ __launch_bounds__(TILE_THREADS, 2) void tile_matvec_f64_wide(const TileArgs<double> a) {
    static_assert(KT != KT_LINEAR, "the linear kernel takes feature-panel passes of tile_matvec_f64_v2");
    constexpr int NKC = 4;  // 16-feature chunks per panel
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char *ring = smem_raw;
    char *dcs = smem_raw + V2D_RING * V2D_SLOT_BYTES;
    double *cis = reinterpret_cast<double *>(dcs + V2D_DC_SLOTS * 1024);  // [128] c_i of the row panel (rbf)
    double *dis = cis + TILE;                                              // [128] d_i of the row panel (SYM)
    double *colred = dis + TILE;                                           // [2][4 waves][64] column sums of a sub-tile (SYM)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15;
    const int q = lane >> 4;

    int ibl, jc;
    if constexpr (SYM) {
        const int2 it = a.items[blockIdx.x];
        ibl = __builtin_amdgcn_readfirstlane(it.x);
        jc = __builtin_amdgcn_readfirstlane(it.y);
    } else {
        if (!decode_work_item(a, ibl, jc)) return;
    }
    const int ib = a.ib_begin + ibl;
    const int row0 = ib * TILE;
    const int jt_begin = jc * a.jc_tiles;
    const int jt_end = SYM ? min(jt_begin + a.jc_tiles, ib + 1) : min(jt_begin + a.jc_tiles, a.num_jt);
    const int nsub = 2 * (jt_end - jt_begin);  // 64-column sub-tiles
    if (nsub <= 0) return;
    const int st_begin = 2 * jt_begin;
    const int panels = a.kchunks / NKC;  // (uniform; the data is padded to a multiple of 64 features)
    const int steps_per_sub = panels * NKC;
    const int nsteps = nsub * steps_per_sub;
    const long rec0 = SYM ? (static_cast<long>(ib) * (ib - 1) - 2 * a.pair_origin) : 0;

    // row panel of ONE feature panel: A operand of lane (r, q) for k-step s of chunk c is X[row][64 p + 16 c + 4 s + q]
    // (uniform base in SGPRs + one 32-bit lane offset: no per-lane 64-bit row pointers across the work item)
    double afrag[2][4 * NKC];
    // Symmetric variant (training): from the FRAGMENT-MAJOR copy of the data (TileArgs::Xrf, k_rows_fragment_major_f64) -- a load instruction reads the
    // 512 contiguous bytes of one A fragment instead of 32 bytes of each of 16 lines.  The re-loads are what this kernel loses against the one-pass
    // kernels (ablation: without them 21.6 -> 16.9 ms at 60 000 x 320 rbf = the one-pass rate).
    const unsigned row_lane_off = SYM ? 8u * static_cast<unsigned>(lane) : 8u * static_cast<unsigned>(r * a.ldx + q);
    auto load_row_chunk = [&](int p, auto chunk_c) {
        constexpr int chunk = decltype(chunk_c)::value;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            if constexpr (SYM) {
                const char *base = sgpr_ptr(a.Xrf + ((static_cast<size_t>(p * NKC + chunk) * a.frag_rows16 + static_cast<size_t>(row0 / 16 + wave * 2 + rb)) * 4) * 64);
                const auto *xr = (const __attribute__((address_space(1))) char *) base + lane_off(row_lane_off);
#pragma unroll
                for (int s = 0; s < 4; ++s) afrag[rb][4 * chunk + s] = *reinterpret_cast<const __attribute__((address_space(1))) double *>(xr + 512 * s);
            } else {
                const char *base = sgpr_ptr(a.Xr + static_cast<size_t>(row0 + wave * 32 + rb * 16) * a.ldx + p * (16 * NKC) + 16 * chunk);
                const auto *xr = (const __attribute__((address_space(1))) char *) base + lane_off(row_lane_off);
#pragma unroll
                for (int s = 0; s < 4; ++s) afrag[rb][4 * chunk + s] = *reinterpret_cast<const __attribute__((address_space(1))) double *>(xr + 32 * s);
            }
        }
    };
    static_for<0, NKC>([&](auto c) { load_row_chunk(0, c); });
    if constexpr (KT == KT_RBF) {
        if (tid < TILE) cis[tid] = a.cr[row0 + tid];
    }
    if constexpr (SYM) {
        if (tid < TILE) dis[tid] = a.dvec[row0 + tid];
    }
    // retire these ordinary loads HERE, before any LDS-DMA is in flight
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int s = 0; s < 4 * NKC; ++s) asm volatile("" : "+v"(afrag[rb][s]));

    // LDS image of a chunk and its DMA addressing: tile_matvec_f64_v2's
    unsigned dma_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int col = wave * 16 + (lane & 15);
        const int ks = 4 * i + (lane >> 4);
        dma_off[i] = 8u * static_cast<unsigned>(col * a.ldx + 2 * ks);
    }
    const unsigned ring_lds = static_cast<unsigned>(reinterpret_cast<size_t>(ring));
    const unsigned dma_lds = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(ring_lds + static_cast<unsigned>(wave) * 2048u)));
    const unsigned dc_lds = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(ring_lds + V2D_RING * V2D_SLOT_BYTES + static_cast<unsigned>(wave) * 256u)));
    auto issue_chunk = [&](int step) {  // step = (sub-tile t, panel p, chunk kc)
        const int t = step / steps_per_sub;
        const int in_sub = step - t * steps_per_sub;  // = p * NKC + kc: the 16-feature chunk of the row
        const char *base = sgpr_ptr(a.Xc + static_cast<size_t>(st_begin + t) * 64 * a.ldx + in_sub * 16);
        const unsigned slot = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(dma_lds + static_cast<unsigned>(step % V2D_RING) * V2D_SLOT_BYTES)));
        lds_dma16<0>(dma_off[0], base, slot);
        lds_dma16<1024>(dma_off[1], base, slot);
    };
    auto issue_dc = [&](int t) {
        if (lane < 16) {
            const char *src = sgpr_ptr(a.dc + static_cast<size_t>(st_begin + t) * 128) + __builtin_amdgcn_readfirstlane(wave * 256);
            lds_dma16<0>(16u * (lane_off(threadIdx.x) & 15u), sgpr_ptr(src), static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(dc_lds + static_cast<unsigned>(t % V2D_DC_SLOTS) * 1024u))));
        }
    };

    const int lane_base = (q >> 1) * 256 + r * 16 + (q & 1) * 8;
    auto read_group = [&](const char *slot, int s, double (&b)[4]) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) b[cb] = *((const volatile __attribute__((address_space(3))) double *) (slot + cb * 2048 + s * 512));
    };

    double rowpart[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 4; ++i) rowpart[rb][i] = 0.0;
    f64x4 acc[2][4];
    double dj[4];

    // ---- prologue: chunks 0, 1, 2 (a sub-tile has at least 5 x NKC steps here: only the record of sub-tile 0 falls into it) ----
    issue_dc(0);
    issue_chunk(0);
    issue_chunk(1);
    issue_chunk(2);
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    double bcur[4];
    read_group(ring + lane_base, 0, bcur);

    // hand-over of the next chunk in the middle of a step (tile_matvec_f64_v2's checked form)
    auto handover = [&](int step) {
        if (step + 1 < nsteps) {
            if (step + 2 < nsteps) {
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (step + 3 < nsteps) {
                const int s3 = step + 3;
                const int t3 = s3 / steps_per_sub;
                if (s3 - t3 * steps_per_sub == 0) issue_dc(t3);
                issue_chunk(s3);
            }
        }
    };

    auto flush_cols = [&](int t) {
        if (tid < 64) {
            const double *cr_ = colred + (t & 1) * 256;
            auto *rec = (__attribute__((address_space(1))) double *) (a.colslab + (rec0 + st_begin + t) * 64);
            rec[static_cast<unsigned>(tid)] = (cr_[tid] + cr_[64 + tid]) + (cr_[128 + tid] + cr_[192 + tid]);
        }
    };

    for (int t = 0; t < nsub; ++t) {
        const double *dcr = reinterpret_cast<const double *>(dcs + (t % V2D_DC_SLOTS) * 1024);
        // start values of the chains: rbf c_i + c_j; polynomial coef0 (the data carries sqrt(gamma): the chain leaves gamma <x_i, x_j>)
        if constexpr (KT == KT_RBF) {
            double cj[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) cj[cb] = dcr[64 + cb * 16 + r];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double civ = cis[wave * 32 + rb * 16 + q + 4 * i];
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) acc[rb][cb][i] = civ + cj[cb];
                }
        } else {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[rb][cb][i] = a.coef0;
        }
        // The start values must live in the accumulators' OWN registers (C == D for every MFMA of the chain).  Left alone, the optimiser peels the first
        // panel, folds a splat start value (coef0) into ONE register quad that serves as the C operand of the first MFMA of all eight accumulators,
        // and -- the quad being dead behind the eighth -- loads the next k-step's B fragments into it right there:
        //       v_mfma_f64_16x16x4_f64 v[2:9], v[88:89], v[156:157], v[66:73]
        //       ds_read_b64 v[66:67], v195 offset:512  ...  ds_read_b64 v[72:73], v195 offset:6656
        // On gfx950 a load into the last register pair of C within five wait states of a v_mfma_f64_16x16x4_f64 corrupts the last rows of C (59 % of the
        // lanes at three wait states, none from six on: tests/tools/repro/dgemm_srcc_war.hip; the 16-bit MFMAs are not affected), and ROCm 7.2's
        // hazard recognizer has no rule for it.  That was round 3's wrong instantiation (run-time integer power, symmetric variant: accumulator element
        // [1][3][3] started from a B-fragment value instead of coef0), "fixed" then by keeping the loops rolled; DESIGN.md section 4.1 has the trail.
        // tests/tools/audit_hand_asm.py now checks EVERY kernel of the build for a load into the C operand of an in-flight v_mfma_f64.
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) asm volatile("" : "+v"(acc[rb][cb]));
        for (int p = 0; p < panels; ++p) {
            const bool more_panels = t + 1 < nsub || p + 1 < panels;
            const int p_next = p + 1 < panels ? p + 1 : 0;  // (the row panel depends on the feature panel only, not on the sub-tile)
            const int s0 = t * steps_per_sub + p * NKC;
            static_for<0, NKC>([&](auto kc_c) {
                constexpr int kc = decltype(kc_c)::value;
                const int step = s0 + kc;
                const char *slot = ring + (step % V2D_RING) * V2D_SLOT_BYTES + lane_base;
                const char *slot_next = ring + ((step + 1) % V2D_RING) * V2D_SLOT_BYTES + lane_base;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    double bnext[4];
                    if (s < 3) read_group(slot, s + 1, bnext);
                    if (s == 2) {
                        if constexpr (SYM) {
                            if (kc == 0 && p == 0 && t > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        }
                        handover(step);
                        if constexpr (SYM) {
                            if (kc == 0 && p == 0 && t > 0 && (st_begin + t - 1 < 2 * ib)) flush_cols(t - 1);
                        }
                    }
                    if (s == 3) read_group(slot_next, 0, bnext);  // (behind the last step: a stale but valid slot, never used)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[rb][4 * kc + s], bcur[cb], acc[rb][cb], 0, 0, 0);
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) bcur[cb] = bnext[cb];
                }
                // this chunk's row fragments are dead: the registers take the same chunk of the next (sub-tile, panel)
                if (more_panels) load_row_chunk(p_next, kc_c);
            });
        }
        // ---- epilogue of the sub-tile (tile_matvec_f64_v2's; ONE form per instantiation) ----
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) dj[cb] = dcr[cb * 16 + r];
        double colacc[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double di = 0.0;
                if constexpr (SYM) di = dis[wave * 32 + rb * 16 + q + 4 * i];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    double kv;
                    if constexpr (KT == KT_RBF) {
                        kv = exp2_f64(acc[rb][cb][i]);  // the data is pre-scaled: acc = log2(K)
                    } else {
                        kv = poly_power<v2_degree_class(KT)>(acc[rb][cb][i], a.degree);
                    }
                    rowpart[rb][i] = fma(kv, dj[cb], rowpart[rb][i]);
                    if constexpr (SYM) colacc[cb] = fma(kv, di, colacc[cb]);
                }
            }
        if constexpr (SYM) {
            double *cw = colred + (t & 1) * 256 + wave * 64;
            cw[lane] = column_sums_of_4_blocks(colacc);  // (diagonal sub-tiles: computed, never flushed)
        }
    }
    if constexpr (SYM) {
        if (st_begin + nsub - 1 < 2 * ib) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            flush_cols(nsub - 1);
        }
    }

#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double v = rowpart[rb][i];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            rowpart[rb][i] = v;
        }
    if (r == 0) {
        double *dst = a.partial + static_cast<size_t>(jc) * a.part_stride + ibl * TILE + wave * 32 + q;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[rb * 16 + 4 * i] = rowpart[rb][i];
    }
}

}  // namespace lssvm
