/*
 * lssvm_tile_f32_wide.hip.hpp -- the fp32 split tile kernel for rbf / polynomial problems with MORE features than a row panel in registers can
 * hold (f16x3: more than 384 (rbf) / 512 features; bf16x6: more than 384).
 *
 * The linear kernel of such problems runs one pass of the 128-feature kernels per feature panel (K = sum_p X_p X_p^T, lssvm_problem.hip); rbf and
 * polynomial need the WHOLE dot product before the kernel function, so here the panels are walked INSIDE a tile: a work item's tile is
 * `panels` x (128 features) tile-panels, the accumulators live across them, the epilogue runs after the last.  What changes against s6w_body
 * (lssvm_tile_f32_split.hip.hpp), whose structure this is -- same LDS-DMA ring of 128-column x 64-feature plane-chunks, same swizzle, same mid-step
 * hand-over, same records, same epilogue and symmetric variant:
 *   - the row panel (A fragments of the wave's 32 rows, 128 features, all row planes) is RE-LOADED from global memory at every tile-panel: as
 *     many bytes as the column stream, from L2 (a row block's planes: 128 rows x d x 2 bytes x planes -- 1.5 MB at 2 000 features); a 64-feature
 *     chunk's fragments are requested as soon as the previous tile-panel is through with that chunk, into the same registers;
 *   - the column stream walks (tile, panel, 64-feature chunk, plane) with run-time addressing and the checked hand-over throughout;
 *   - the accumulators are initialised explicitly at panel 0 (no "first MFMA takes C = 0 / c_i" forms: 64 moves per tile against >= 768 MFMAs).
 * Compiler-scheduled MFMA groups, two waves per SIMD (the register budget of the 128-feature kernels).  Symmetric variant and full-square variant
 * (option symmetric = 0; predict_values: rows = points, columns = support vectors); a negative polynomial degree (0^degree on padded columns) stays
 * on the generic native kernel.
 * Reference semantics: /root/reference/include/plssvm/backends/HIP/svm_kernel.hip.hpp:129-270 (one code path for any feature count).
 */
#pragma once

#include "lssvm_tile_f32_split.hip.hpp"

namespace lssvm {

template <int KT, int PL, bool SYM>
__device__ __forceinline__ void s6x_body(const TileArgs<float> &a) {
    static_assert(PL == 3 || PL == 2, "three bf16 planes (bf16x6) or two f16 planes (f16x3)");
    static_assert(KT != KT_LINEAR, "the linear kernel takes feature-panel passes of the 128-feature kernels");
    constexpr bool F16 = PL == 2;
    // KT_RBFG (rbf on GRID planes, round 5; s6w_body / DESIGN.md section 4.1.2): three f16 planes (h | s1 | s2) and TWO sweeps over the feature panels of a tile -- sweep A: the
    // h x h products of ALL panels (they must come first: with the exact start values they leave -sigma^2 |h_i - h_j|^2 / 2 exactly), sweep B: per panel and 64-feature chunk
    // the column planes h, s1, s2 against the row planes (s1, s2), (h, s1), (h).  Sweep A keeps row plane h of the panel in registers, sweep B all three.
    constexpr bool GRID = KT == KT_RBFG;
    static_assert(!GRID || PL == 2, "the grid-plane kernel exists with f16 planes only");
    constexpr int NK64 = 2;          // 64-feature chunks per panel
    constexpr int NKC = PL * NK64;   // plane-chunks (steps) per tile-panel
    constexpr int PLA = F16 ? ((KT == KT_RBF || KT == KT_RBFF || GRID) ? 3 : 2) : 3;  // row planes in registers (s6w_body: the shifted rbf planes)
    constexpr auto row_plane = [](int p, int q) constexpr { return (F16 && PLA == 3) ? (p == 0 ? (q == 0 ? 2 : 1) : 0) : q; };
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char *ring = smem_raw;                                                          // [V2_RING][128 rows][128 B]
    char *dcs = smem_raw + V2_RING * V2_SLOT_BYTES;                                 // [V2_DC_SLOTS][256 floats]
    float *cis = reinterpret_cast<float *>(dcs + V2_DC_SLOTS * 1024);               // [128] c_i of the row panel (rbf)
    float *dis = cis + TILE;                                                        // [128] d_i of the row panel
    float *colred = dis + TILE;                                                     // [2][4 waves][128] column sums of a tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15;
    const int g = lane >> 4;

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
    const int ntiles = jt_end - jt_begin;
    if (ntiles <= 0) return;
    const int panels = a.nk64 / NK64;          // (uniform; the planes are padded to a multiple of 128 features)
    const int steps_per_tile = GRID ? panels * (4 * NK64) : panels * NKC;  // (grid planes: sweep A NK64 steps per panel, sweep B 3 NK64)
    const int nsteps = ntiles * steps_per_tile;
    const long rec0 = SYM ? (static_cast<long>(ib) * (ib - 1) / 2 - a.pair_origin) : 0;

    // ---- the row panel of ONE feature panel: lane (r, g) holds features 128 p + 32 kk + 8 g .. + 7 of row 16 rb + r ----
    bf16x8 afrag[PLA][2 * NK64][2];
    // (uniform base in SGPRs + ONE 32-bit lane offset: per-lane 64-bit row pointers for every plane and row block would live across the whole
    // work item and spill)
    // Symmetric variant (training): the row side comes from a FRAGMENT-MAJOR copy of the planes (TileArgs::Xr16f, k_planes_fragment_major) -- every
    // block of 16 rows x 32 features stored as the 1 KiB a wave loads as one A fragment (64-feature chunk outermost, then the row block, then the
    // chunk's two k32 steps), so a load instruction touches 8 whole 128-byte lines instead of 16 half lines.  The re-loads are what this kernel waits for (ablation: without them 8.6 -> 4.7 ms at 60 000 x 640 rbf,
    // profiles/r04_ablation_wide.log), and the vector memory path they share with the column stream counts line requests, not bytes.
    const unsigned row_lane_off = SYM ? 16u * static_cast<unsigned>(lane) : 2u * static_cast<unsigned>(r * a.ldx16 + 8 * g);
    const size_t frag_rows16 = SYM ? a.plane_stride_r / static_cast<size_t>(a.ldx16) / 16 : 0;  // 16-row blocks of a plane (uniform; once per work item)
    auto load_row_chunk_planes = [&](int p, auto chunk_c, auto pl0_c, auto pl1_c) {  // planes [pl0, pl1) of the 64-feature chunk `chunk` of panel p: k32 steps 2 chunk, 2 chunk + 1
        constexpr int chunk = decltype(chunk_c)::value;
#pragma unroll
        for (int pl = decltype(pl0_c)::value; pl < decltype(pl1_c)::value; ++pl) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                if constexpr (SYM) {
                    const char *base = sgpr_ptr(a.Xr16f + pl * a.plane_stride_r + (static_cast<size_t>(p * NK64 + chunk) * frag_rows16 + static_cast<size_t>(row0 / 16 + wave * 2 + rb)) * 1024);
                    const auto *xr = (const __attribute__((address_space(1))) char *) base + lane_off(row_lane_off);
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) afrag[pl][2 * chunk + kk][rb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const __attribute__((address_space(1))) f32x4 *>(xr + 1024 * kk));
                } else {
                    const char *base = sgpr_ptr(a.Xr16 + pl * a.plane_stride_r + static_cast<size_t>(row0 + wave * 32 + 16 * rb) * a.ldx16 + p * (64 * NK64) + 64 * chunk);
                    const auto *xr = (const __attribute__((address_space(1))) char *) base + lane_off(row_lane_off);
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) afrag[pl][2 * chunk + kk][rb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const __attribute__((address_space(1))) f32x4 *>(xr + 64 * kk));
                }
            }
        }
    };
    auto load_row_chunk = [&](int p, auto chunk_c) { load_row_chunk_planes(p, chunk_c, std::integral_constant<int, 0>{}, std::integral_constant<int, PLA>{}); };
    auto load_row_chunk_h = [&](int p, auto chunk_c) { load_row_chunk_planes(p, chunk_c, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); };  // (grid planes, sweep A)
    if constexpr (GRID) {
#pragma unroll
        for (int pl = 1; pl < PLA; ++pl)
#pragma unroll
            for (int kk = 0; kk < 2 * NK64; ++kk)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) afrag[pl][kk][rb] = bf16x8{};  // (loaded at the end of the first sweep A)
        static_for<0, NK64>([&](auto c) { load_row_chunk_h(0, c); });
    } else {
        static_for<0, NK64>([&](auto c) { load_row_chunk(0, c); });
    }
    if constexpr (KT == KT_RBF || KT == KT_RBFF || GRID) {
        if (tid < TILE) cis[tid] = a.cr[row0 + tid];  // (grid planes: sigma^2 ch_i)
    }
    if constexpr (SYM) {
        if (tid < TILE) dis[tid] = GRID ? a.dvec[row0 + tid] * a.er[row0 + tid] : a.dvec[row0 + tid];  // (grid planes: the row's folded factor E_i rides on d_i)
    }
    // make the compiler retire these ordinary loads HERE, before any LDS-DMA is in flight
#pragma unroll
    for (int pl = 0; pl < PLA; ++pl)
#pragma unroll
        for (int kk = 0; kk < 2 * NK64; ++kk)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) asm volatile("" : "+v"(afrag[pl][kk][rb]));

    // ---- LDS-DMA addressing (the LDS image of a plane-chunk is that of s6w_body) ----
    unsigned dma_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * (4 * wave + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        dma_off[i] = 2u * static_cast<unsigned>(row * a.ldx16 + 8 * c);
    }
    const unsigned ring_lds = static_cast<unsigned>(reinterpret_cast<size_t>(ring));
    const unsigned dma_lds = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(ring_lds + static_cast<unsigned>(wave) * 4096u)));
    const unsigned dc_lds = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(ring_lds + V2_RING * V2_SLOT_BYTES + static_cast<unsigned>(wave) * 256u)));
    auto issue_chunk = [&](int step) {  // step = (tile t, panel p, plane-chunk kc): plane kc % PL of the 64-feature chunk kc / PL of panel p
        const int t = step / steps_per_tile;
        const int in_tile = step - t * steps_per_tile;
        int p, chunk, plane;
        if constexpr (GRID) {
            const int a_steps = panels * NK64;
            if (in_tile < a_steps) {  // sweep A: column plane h
                p = in_tile / NK64;
                chunk = in_tile - p * NK64;
                plane = 0;
            } else {                  // sweep B: per chunk the column planes h, s1, s2
                const int rem = in_tile - a_steps;
                p = rem / (3 * NK64);
                const int kc = rem - p * (3 * NK64);
                chunk = kc / 3;
                plane = kc - 3 * chunk;
            }
        } else {
            p = in_tile / NKC;
            const int kc = in_tile - p * NKC;
            chunk = kc / PL;
            plane = kc % PL;
        }
        const char *base = sgpr_ptr(a.Xc16 + plane * a.plane_stride + static_cast<size_t>(jt_begin + t) * TILE * a.ldx16 + (p * NK64 + chunk) * 64);
        const unsigned slot = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(dma_lds + static_cast<unsigned>(step % V2_RING) * V2_SLOT_BYTES)));
        static_for<0, 4>([&](auto i_c) { lds_dma16<decltype(i_c)::value * 1024>(dma_off[decltype(i_c)::value], base, slot); });
    };
    auto issue_dc = [&](int t) {
        if (lane < 16) {
            const char *src = sgpr_ptr(a.dc + static_cast<size_t>(jt_begin + t) * 256) + __builtin_amdgcn_readfirstlane(wave * 256);
            lds_dma16<0>(16u * (lane_off(threadIdx.x) & 15u), sgpr_ptr(src), static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(dc_lds + static_cast<unsigned>(t % V2_DC_SLOTS) * 1024u))));
        }
    };

    int rd_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) rd_off[kk] = r * 128 + (((4 * kk + g) ^ ((r >> 1) & 7)) << 4);

    float rowpart[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rowpart[i] = 0.0f;
    f32x4 acc[2][8];

    // ---- prologue: chunks 0, 1, 2 (a tile has at least 2 x NKC >= 8 steps here: only the record of tile 0 falls into it) ----
    issue_dc(0);
    issue_chunk(0);
    issue_chunk(1);
    issue_chunk(2);
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    f32x4 civ0[2] = { { 0.f, 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f, 0.f } };
    if constexpr (KT == KT_RBF || KT == KT_RBFF || GRID) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) civ0[rb] = *reinterpret_cast<const f32x4 *>(cis + wave * 32 + 16 * rb + 4 * g);
    }
    f32x4 bbuf[2][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bbuf[0][c] = *reinterpret_cast<const f32x4 *>(ring + c * 2048 + rd_off[0]);

    // hand-over of the next chunk in the middle of a step (s6w_body's checked form): this wave's four pieces of chunk step + 1 are complete once
    // all but its 4 youngest DMA instructions (chunk step + 2) are; the barrier publishes every wave's pieces; the DMA issued here (chunk
    // step + 3) overwrites the slot of chunk step - 1, which every wave finished before this barrier.  (The row-panel re-loads are ordinary
    // loads the compiler waits for itself with vmcnt(0) -- which also lands the chunks in flight, early but in order: the counted waits that
    // follow are then trivially met.)
    auto handover = [&](int step) {
        if (step + 1 < nsteps) {
            if (step + 2 < nsteps) {
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (step + 3 < nsteps && !LSSVM_DBG(a, 16)) {
                const int s3 = step + 3;
                const int t3 = s3 / steps_per_tile;
                if (s3 - t3 * steps_per_tile == 0) issue_dc(t3);
                issue_chunk(s3);
            }
        }
    };

    auto flush_cols = [&](int t) {
        if (tid < TILE) {
            const float *cr_ = colred + (t & 1) * 512;
            const float sum = (cr_[tid] + cr_[128 + tid]) + (cr_[256 + tid] + cr_[384 + tid]);
            auto *rec = (__attribute__((address_space(1))) float *) const_cast<char *>(sgpr_ptr(a.colslab + (rec0 + jt_begin + t) * TILE));
            rec[lane_off(static_cast<unsigned>(tid))] = sum;
        }
    };

    for (int t = 0; t < ntiles; ++t) {
        const bool tile_sym = SYM && (jt_begin + t < ib);  // strictly below the diagonal
        const float *dcr = reinterpret_cast<const float *>(dcs + (t % V2_DC_SLOTS) * 1024);
        // start values of the chain: rbf c_i + c_j, rbf with folded records c_i (c_j is the factor 2^c_j of the record), polynomial 0
        if constexpr (KT == KT_RBF || GRID) {  // (grid planes: sigma^2 (ch_i + ch_j), an exact sum)
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const float cjv = dcr[128 + cb * 16 + r];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[rb][cb][e] = civ0[rb][e] + cjv;
            }
        } else {
#pragma unroll
            for (int cb = 0; cb < 8; ++cb)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) acc[rb][cb] = civ0[rb];  // (zero for the polynomial kernels)
        }
        // SWEEP 0: the bf16x6 / f16x3 kernels (one sweep: per panel and chunk the column planes against the row planes of equal or lower order);
        // grid planes: SWEEP 1 = A (h x h of every panel), SWEEP 2 = B (the remaining five products of every panel)
        auto panel_steps = [&](int p, auto sweep_c) {
            constexpr int SWEEP = decltype(sweep_c)::value;
            constexpr int NKS = SWEEP == 0 ? NKC : (SWEEP == 1 ? NK64 : 3 * NK64);  // steps of this panel in this sweep
            // (the row panel of this tile-panel was requested chunk by chunk while the previous one was being multiplied: see below)
            const bool more_panels = t + 1 < ntiles || p + 1 < panels;
            const int p_next = p + 1 < panels ? p + 1 : 0;  // (the row panel depends on the feature panel only, not on the tile)
            const int s0 = t * steps_per_tile + (SWEEP == 2 ? panels * NK64 : 0) + p * NKS;
            const unsigned phase = static_cast<unsigned>(s0) & (V2_RING - 1);
            static_for<0, NKS>([&](auto kc_c) {
                constexpr int kc = decltype(kc_c)::value;
                constexpr int chunk = SWEEP == 0 ? kc / PL : (SWEEP == 1 ? kc : kc / 3);
                constexpr int plane = SWEEP == 0 ? kc % PL : (SWEEP == 1 ? 0 : kc % 3);
                const int step = s0 + kc;
                const unsigned slot_off = ((phase + kc) & (V2_RING - 1)) * V2_SLOT_BYTES;
                const unsigned slot_next_off = ((phase + kc + 1) & (V2_RING - 1)) * V2_SLOT_BYTES;
                const char *slot = ring + slot_off;
                const char *slot_next = ring + slot_next_off;
                static_for<0, 4>([&](auto mm_c) {
                    constexpr int mm = decltype(mm_c)::value;
                    constexpr int kk = mm >> 1, cbh = mm & 1;
                    f32x4(&bcur)[4] = bbuf[mm & 1];
                    f32x4(&bnext)[4] = bbuf[(mm + 1) & 1];
                    if constexpr (mm < 3) {  // next group of this chunk
#pragma unroll
                        for (int c = 0; c < 4; ++c) bnext[c] = *reinterpret_cast<const f32x4 *>(slot + (4 * ((mm + 1) & 1) + c) * 2048 + rd_off[(mm + 1) >> 1]);
                        if constexpr (mm != 2) LSSVM_SCHED_BARRIER();
                    }
                    if constexpr (mm == 2) {
                        // (the colred writes of the previous tile's epilogue must have completed before the barrier publishes them)
                        if constexpr (SYM && SWEEP != 2) {  // (the first step of a tile lies in sweep 0 / A)
                            if (kc == 0 && p == 0 && t > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        }
                        handover(step);
                        if constexpr (SYM && SWEEP != 2) {
                            if (kc == 0 && p == 0 && t > 0) flush_cols(t - 1);  // (tile t - 1 < t <= the diagonal tile: always off-diagonal)
                        }
                        LSSVM_SCHED_BARRIER();
                    }
                    if constexpr (mm == 3) {  // first group of the next chunk: visible since this step's hand-over
                        if (step + 1 < nsteps) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) bnext[c] = *reinterpret_cast<const f32x4 *>(slot_next + c * 2048 + rd_off[0]);
                        }
                        LSSVM_SCHED_BARRIER();
                    }
                    // the row planes this column plane meets: sweep 0: those of equal or lower order; A: h; B: h -> (s1, s2), s1 -> (h, s1), s2 -> (h)
                    constexpr int NQ = SWEEP == 0 ? PL - plane : (SWEEP == 1 ? 1 : (plane == 2 ? 1 : 2));
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        constexpr auto rp = [](int pl, int qq) constexpr { return SWEEP == 0 ? -1 : (SWEEP == 1 ? 0 : (pl == 0 ? 1 + qq : (pl == 1 ? qq : 0))); };
                        const int rpl = SWEEP == 0 ? row_plane(plane, q) : rp(plane, q);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int cb = 4 * cbh + c;
                            const bf16x8 bv = __builtin_bit_cast(bf16x8, bcur[c]);
#pragma unroll
                            for (int rb = 0; rb < 2; ++rb) acc[rb][cb] = plane_mfma<F16>(afrag[rpl][2 * chunk + kk][rb], bv, acc[rb][cb]);
                        }
                    }
                });
                // the last plane of a 64-feature chunk is through: its row fragments are dead, the registers take the same chunk of the NEXT
                // tile-panel -- requested half a tile-panel or more before its first use instead of in front of it
                if constexpr (SWEEP == 0) {
                    if constexpr (plane == PL - 1) {
                        if (more_panels && !LSSVM_DBG(a, 1)) load_row_chunk(p_next, std::integral_constant<int, chunk>{});  // ablation bit 1: no row-panel re-loads
                    }
                } else if constexpr (SWEEP == 1) {  // sweep A: plane h of the next panel -- or, behind the last panel, ALL planes of panel 0 for sweep B
                    if (p + 1 < panels) {
                        load_row_chunk_h(p + 1, std::integral_constant<int, chunk>{});
                    } else {
                        load_row_chunk(0, std::integral_constant<int, chunk>{});
                    }
                } else {  // sweep B: all planes of the next panel -- or, behind the last panel, plane h of panel 0 for the next tile's sweep A
                    if constexpr (plane == 2) {
                        if (p + 1 < panels) {
                            load_row_chunk(p + 1, std::integral_constant<int, chunk>{});
                        } else if (t + 1 < ntiles) {
                            load_row_chunk_h(0, std::integral_constant<int, chunk>{});
                        }
                    }
                }
            });
        };
        if constexpr (GRID) {
            for (int p = 0; p < panels; ++p) panel_steps(p, std::integral_constant<int, 1>{});
            for (int p = 0; p < panels; ++p) panel_steps(p, std::integral_constant<int, 2>{});
        } else {
            for (int p = 0; p < panels; ++p) panel_steps(p, std::integral_constant<int, 0>{});
        }
        // ---- epilogue of the tile (s6w_body's, with the mirrored column sums where the tile is off the diagonal) ----
        auto epilogue = [&](auto with_cols) {
            constexpr bool COLS = decltype(with_cols)::value;
            using f32x2 = float __attribute__((ext_vector_type(2)));
            f32x4 di[2];
            float colacc[8];
            f32x2 colacc2[8] = {};
            f32x2 kvp = { 0.f, 0.f };
            if constexpr (COLS) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) di[rb] = *reinterpret_cast<const f32x4 *>(dis + wave * 32 + 16 * rb + 4 * g);
            }
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const float djv = dcr[cb * 16 + r];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float kv = apply_kernel_function<v2_base_kt(KT), v2_degree_class(KT)>(GRID ? acc[rb][cb][e] * a.gamma : acc[rb][cb][e], a);  // (grid planes: the chain carries sigma^2)
                        rowpart[4 * rb + e] = fmaf(kv, djv, rowpart[4 * rb + e]);
                        if constexpr (COLS) {
                            kvp[e & 1] = kv;
                            if (e & 1) {
                                const f32x2 dip = { di[rb][e - 1], di[rb][e] };
                                colacc2[cb] = __builtin_elementwise_fma(kvp, dip, colacc2[cb]);
                            }
                        }
                    }
            }
            if constexpr (COLS) {
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) colacc[cb] = colacc2[cb][0] + colacc2[cb][1];
                float *cw = colred + (t & 1) * 512 + wave * 128;
                column_sums_of_8_blocks(colacc);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float v = colacc[4 * h];
                    if constexpr (KT == KT_RBFF) v *= dcr[128 + 64 * h + lane];
                    cw[64 * h + lane] = v;
                }
            }
        };
        if (!LSSVM_DBG(a, 4)) {
            if (tile_sym) {
                epilogue(std::true_type{});
            } else {
                epilogue(std::false_type{});
            }
        }
    }
    if constexpr (SYM) {
        if (jt_begin + ntiles - 1 < ib) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            flush_cols(ntiles - 1);
        }
    }

    // every lane group owns its rows: reduce over the 16 columns of the group and store
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float v = rowpart[i];
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 1);
        if constexpr (GRID) v *= a.er[row0 + wave * 32 + 16 * (i >> 2) + 4 * g + (i & 3)];  // the row's folded factor E_i, once per work item
        rowpart[i] = v;
    }
    if (r == 0) {
        float *dst = a.partial + static_cast<size_t>(jc) * a.part_stride + ibl * TILE + wave * 32 + 4 * g;
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[16 * (i >> 2) + (i & 3)] = rowpart[i];
    }
}

/* PL = 2: f16x3 (two f16 column planes), PL = 3: bf16x6 (three bf16 planes).  Two waves per SIMD. */
/* (the grid-plane instantiation spills 15 ... 41 registers at two waves per SIMD; one wave per SIMD -- no spills -- measured 39 % SLOWER: 40 000 x 1 024 8.4 -> 11.7 ms, 60 000 x 640
 * 11.6 -> 16.2 ms, same box, tests/tools/build_unit_variant.sh widegrid1 tile_launch_f32x -DLSSVM_WIDE_GRID_WAVES=1) */
#ifndef LSSVM_WIDE_GRID_WAVES
#define LSSVM_WIDE_GRID_WAVES 2
#endif
template <int KT, int PL, bool SYM>
__global__ __launch_bounds__(TILE_THREADS, (KT == KT_RBFG ? LSSVM_WIDE_GRID_WAVES : 2)) void tile_matvec_f32_wide(const TileArgs<float> a) {
    s6x_body<KT, PL, SYM>(a);
}

}  // namespace lssvm
