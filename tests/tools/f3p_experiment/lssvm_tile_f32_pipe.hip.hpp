/*
 * lssvm_tile_f32_pipe.hip.hpp -- the SOFTWARE-PIPELINED f16x3 tile kernel ("f3p"): one wave per SIMD, the epilogue of tile t - 1 placed
 * instruction by instruction between the MFMAs of tile t.
 *
 * Why: with two waves per SIMD (tile_matvec_f32_f3h) the vector-ALU epilogue of one wave does not hide beside the MFMAs
 * of the other -- a v_mfma_f32_16x16x32 holds the SIMD's vector issue for 8 of its 16 cycles, so a wave that streams MFMAs leaves exactly
 * the slots one v_exp_f32 or two fmas need, but the hardware's arbitration between two independent waves does not hit them; measured, the
 * f16x3 kernel's time is MFMA time PLUS epilogue time (0.66 of the SIMD cycles busy).  Inside ONE wave the placement is the program
 * order: gen_f3p.py writes the steady-state tile as asm statements in which every MFMA is followed by its share of the previous tile's
 * epilogue.  The price is a second accumulator set (tile t's Gram values wait in registers while tile t + 1 is multiplied) -- which only
 * a one-wave-per-SIMD kernel with the 512-entry register file has room for.
 *
 * Same data movement as s6w_body (LDS-DMA ring of 4 plane-chunks, swizzled 128-byte rows, records, hand-over in the middle of a step),
 * same arithmetic in the same order: results are BIT-IDENTICAL to tile_matvec_f32_f3h / _f3w (asserted by the GPU tests).  Differences:
 *   - the LDS-DMA of chunk step + 3 is ALWAYS issued; behind the last tile of the work item it re-reads the last tile (source pointers
 *     clamped by the code below): no "checked" variant of the tile, at the price of three plane-chunks per work item;
 *   - the column sums of tile t are published while tile t + 1 is multiplied and flushed to their record in tile t + 2.
 * Exists for: rbf with folded records, 65 ... 128 features, symmetric variant (the BASELINE's headline configuration); everything else
 * runs on the two-waves-per-SIMD kernels.
 */
#pragma once

#include "lssvm_device_common.hip.hpp"

namespace lssvm {

#include "lssvm_f3p_tiles.inc"

/* v0 ... v79 / a0 ... a79 for the compiler, everything above for the generated statements (see LSSVM_HAND_VGPR_CAP in
 * lssvm_tile_f32_split.hip.hpp for what the attribute counts; tests/tools/audit_hand_asm.py checks the generated code) */
__global__ __launch_bounds__(TILE_THREADS, 1) __attribute__((amdgpu_num_vgpr(64))) void tile_matvec_f32_f3p_rbff_k2_sym(const TileArgs<float> a) {
    constexpr int NKC = 4;  // plane-chunks (steps) per tile: 2 chunks of 64 features x 2 column planes
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char *ring = smem_raw;                                                          // [V2_RING][128 rows][128 B]
    char *dcs = smem_raw + V2_RING * V2_SLOT_BYTES;                                 // [V2_DC_SLOTS][256 floats]
    float *cis = reinterpret_cast<float *>(dcs + V2_DC_SLOTS * 1024);               // [128] c_i of the row panel
    float *dis = cis + TILE;                                                        // [128] d_i of the row panel
    float *colred = dis + TILE;                                                     // [2][4 waves][128] column sums of a tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15;
    const int g = lane >> 4;

    const int2 it = a.items[blockIdx.x];
    const int ibl = __builtin_amdgcn_readfirstlane(it.x);
    const int jc = __builtin_amdgcn_readfirstlane(it.y);
    const int ib = a.ib_begin + ibl;
    const int row0 = ib * TILE;
    const int jt_begin = jc * a.jc_tiles;
    const int jt_end = min(jt_begin + a.jc_tiles, ib + 1);
    const int ntiles = jt_end - jt_begin;
    if (ntiles <= 0) return;
    const long rec0 = static_cast<long>(ib) * (ib - 1) / 2 - a.pair_origin;

    // ---- the row panel into the private AGPRs (ordinary loads: retired before any LDS-DMA is in flight) ----
    {
        const uint16_t *x[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) x[p][rb] = a.Xr16 + p * a.plane_stride_r + static_cast<size_t>(row0 + wave * 32 + 16 * rb + r) * a.ldx16 + 8 * g;
        f3p_rbff_k2_sym_load_panel(x[0][0], x[0][1], x[1][0], x[1][1], x[2][0], x[2][1]);
    }
    if (tid < TILE) {
        cis[tid] = a.cr[row0 + tid];
        dis[tid] = a.dvec[row0 + tid];
    }

    // ---- LDS-DMA addressing (identical to s6w_body) ----
    unsigned dma_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * (4 * wave + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        dma_off[i] = 2u * static_cast<unsigned>(row * a.ldx16 + 8 * c);
    }
    const size_t tile_bytes = static_cast<size_t>(TILE) * a.ldx16 * 2;
    const size_t plane_bytes = a.plane_stride * 2;
    const char *xc0 = reinterpret_cast<const char *>(a.Xc16) + static_cast<size_t>(jt_begin) * tile_bytes;
    // source of plane-chunk `step` of this work item (clamped to the last tile: see the header comment)
    auto chunk_src = [&](int step) -> const char * {
        const int t = min(step / NKC, ntiles - 1);
        const int kc = step % NKC;
        return sgpr_ptr(xc0 + static_cast<size_t>(t) * tile_bytes + (kc % 2) * plane_bytes + (kc / 2) * 128);
    };
    auto issue_chunk = [&](int step) {  // prologue only
        const char *base = chunk_src(step);
        char *slot = ring + (step % V2_RING) * V2_SLOT_BYTES + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((gbl_ptr_t) (base + lane_off(dma_off[i])), (lds_ptr_t) (slot + i * 1024), 16, 0, 0);
    };
    auto dc_src = [&](int t) -> const char * {  // record of tile min(t, last): 1 KiB, each wave moves a quarter with 16 lanes
        return sgpr_ptr(a.dc + static_cast<size_t>(jt_begin + min(t, ntiles - 1)) * 256) + __builtin_amdgcn_readfirstlane(wave * 256);
    };
    const unsigned dc_off = 16u * (static_cast<unsigned>(lane) & 15u);

    // ---- read addressing (identical to s6w_body) ----
    const unsigned ring_lds = static_cast<unsigned>(reinterpret_cast<size_t>(ring));
    const unsigned dcs_lds = static_cast<unsigned>(reinterpret_cast<size_t>(dcs));
    unsigned rdl[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) rdl[kk] = ring_lds + static_cast<unsigned>(r * 128 + (((4 * kk + g) ^ ((r >> 1) & 7)) << 4));
    const unsigned m0base = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(ring_lds + static_cast<unsigned>(wave) * 4096u)));
    const unsigned dc_m0base = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(dcs_lds + static_cast<unsigned>(wave) * 256u)));
    const unsigned dcr_lane = dcs_lds + 4u * static_cast<unsigned>(r);                                             // + 1024 (t % 4): record of tile t
    const unsigned cw_lane = static_cast<unsigned>(reinterpret_cast<size_t>(colred)) + static_cast<unsigned>(wave) * 512u + 4u * static_cast<unsigned>(r);  // + 2048 (t & 1)

    // ---- prologue: record 0, chunks 0, 1, 2 ----
    {
        if (lane < 16) {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t) (dc_src(0) + lane_off(dc_off)), (lds_ptr_t) (dcs + wave * 256), 16, 0, 0);
        }
        issue_chunk(0);
        issue_chunk(1);
        issue_chunk(2);
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");  // chunk 0, record 0, cis / dis
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    f3p_init_state(static_cast<unsigned>(reinterpret_cast<size_t>(cis)) + static_cast<unsigned>(wave * 32 + 4 * g) * 4u,
                   static_cast<unsigned>(reinterpret_cast<size_t>(dis)) + static_cast<unsigned>(wave * 32 + 4 * g) * 4u, rdl[0]);

    auto flush_cols = [&](int t) {  // the four waves' column sums of tile t, added in a fixed order, to the tile's record of the column slab
        if (tid < TILE) {
            const float *cr_ = colred + (t & 1) * 512;
            const float sum = (cr_[tid] + cr_[128 + tid]) + (cr_[256 + tid] + cr_[384 + tid]);
            auto *rec = (__attribute__((address_space(1))) float *) const_cast<char *>(sgpr_ptr(a.colslab + (rec0 + jt_begin + t) * TILE));
            rec[lane_off(static_cast<unsigned>(tid))] = sum;
        }
    };

    // one tile: MFMAs of tile t into accumulator set t & 1, epilogue of tile t - 1 (off-diagonal: every tile but the last of an item is)
    auto tile = [&](int t, auto variant) {
        constexpr int V = decltype(variant)::value;  // 0: first tile of the item (no epilogue in flight), 1: set 1, 2: set 0
        const int s0 = t * NKC;
        const unsigned dcr_prev = dcr_lane + 1024u * static_cast<unsigned>((t + 3) & 3);  // record of tile t - 1
        const unsigned dcr_cur = dcr_lane + 1024u * static_cast<unsigned>(t & 3);
        const unsigned cw = cw_lane + 2048u * static_cast<unsigned>((t + 1) & 1);         // column sums of tile t - 1
        const char *src0 = chunk_src(s0 + 3), *src1 = chunk_src(s0 + 4), *src2 = chunk_src(s0 + 5), *src3 = chunk_src(s0 + 6);
        const unsigned dc_m0 = dc_m0base + 1024u * static_cast<unsigned>((t + 1) & 3);
        const char *dsrc = dc_src(t + 1);
        if constexpr (V == 0) {
            f3p_rbff_k2_sym_first_a(rdl[0], rdl[1], dcr_prev, dcr_cur, cw);
        } else if constexpr (V == 1) {
            f3p_rbff_k2_sym_set1_a(rdl[0], rdl[1], dcr_prev, dcr_cur, cw);
        } else {
            f3p_rbff_k2_sym_set0_a(rdl[0], rdl[1], dcr_prev, dcr_cur, cw);
        }
        // (behind the first hand-over of the tile: the column sums of tile t - 2, complete since the end of tile t - 1, are visible)
        if (t >= 2) flush_cols(t - 2);
        if constexpr (V == 0) {
            f3p_rbff_k2_sym_first_b(rdl[0], rdl[1], dcr_prev, dcr_cur, cw, dma_off[0], dma_off[1], dma_off[2], dma_off[3], src0, src1, src2, src3, m0base, dc_m0, dc_off, dsrc);
        } else if constexpr (V == 1) {
            f3p_rbff_k2_sym_set1_b(rdl[0], rdl[1], dcr_prev, dcr_cur, cw, dma_off[0], dma_off[1], dma_off[2], dma_off[3], src0, src1, src2, src3, m0base, dc_m0, dc_off, dsrc);
        } else {
            f3p_rbff_k2_sym_set0_b(rdl[0], rdl[1], dcr_prev, dcr_cur, cw, dma_off[0], dma_off[1], dma_off[2], dma_off[3], src0, src1, src2, src3, m0base, dc_m0, dc_off, dsrc);
        }
    };

    tile(0, std::integral_constant<int, 0>{});
    int t = 1;
    for (; t + 1 < ntiles; t += 2) {
        tile(t, std::integral_constant<int, 1>{});
        tile(t + 1, std::integral_constant<int, 2>{});
    }
    if (t < ntiles) {
        tile(t, std::integral_constant<int, 1>{});
        ++t;
    }
    // ---- drain: the epilogue of the last tile (diagonal tile: row sums only), the column sums still in flight ----
    const int last = ntiles - 1;
    const bool last_offdiag = jt_begin + last < ib;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (the LDS-DMA issued beyond the last tile must have landed before the workgroup gives up its LDS)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (last >= 1) flush_cols(last - 1);
    {
        const unsigned dcr_prev = dcr_lane + 1024u * static_cast<unsigned>(last & 3);
        const unsigned cw = cw_lane + 2048u * static_cast<unsigned>(last & 1);
        if (last & 1) {
            if (last_offdiag) f3p_rbff_k2_sym_drain1_cols(dcr_prev, dcr_prev, cw);
            else f3p_rbff_k2_sym_drain1_rows(dcr_prev, dcr_prev, cw);
        } else {
            if (last_offdiag) f3p_rbff_k2_sym_drain0_cols(dcr_prev, dcr_prev, cw);
            else f3p_rbff_k2_sym_drain0_rows(dcr_prev, dcr_prev, cw);
        }
    }
    if (last_offdiag) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        flush_cols(last);
    }

    // every lane group owns its rows: reduce over the 16 columns of the group and store
    float rowpart[8];
    f3p_get_rowsums(rowpart[0], rowpart[1], rowpart[2], rowpart[3], rowpart[4], rowpart[5], rowpart[6], rowpart[7]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float v = rowpart[i];
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 1);
        rowpart[i] = v;
    }
    if (r == 0) {
        float *dst = a.partial + static_cast<size_t>(jc) * a.part_stride + ibl * TILE + wave * 32 + 4 * g;
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[16 * (i >> 2) + (i & 3)] = rowpart[i];
    }
}

}  // namespace lssvm
