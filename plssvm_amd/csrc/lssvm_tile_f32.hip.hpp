/*
 * lssvm_tile_f32.hip.hpp -- the fp32 tile kernels of the implicit kernel-matrix--vector product:
 * tile_matvec_f32_v2 (resident row panel, LDS-DMA ring, symmetric or full square), tile_matvec_f32 (generic), and the
 * direct-form rbf kernel.  Included by tile_launch_f32.hip only.
 */
#pragma once

#include "lssvm_device_common.hip.hpp"

namespace lssvm {

/* =====================================================================================================================
 * fp32 tile kernel: v_mfma_f32_32x32x2_f32
 *   operand maps (cdna_hip_programming.md section 3): lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
 *   the 32x32 result has column j = l&31 on the lane and rows (reg&3) + 8*(reg>>2) + 4*(l>>5) in its 16 registers.
 *   fp32 data is stored in HBM (and hence in LDS) with the features of every aligned group of 8 in the order
 *   k = 0,2,4,6,1,3,5,7 (k_interleave_features, applied once at set-up), so that ONE 16-byte read of lane-half h returns
 *   k = h, 2+h, 4+h, 6+h -- the operands of four consecutive MFMAs -- and the contraction runs through k in ascending
 *   order (bit-identical to the fma chain of the reference's dot product, include/plssvm/detail/operators.hpp:117-126).
 *   LDS image of a k-chunk (this kernel): [128 rows][32 floats + 4 pad].
 * ===================================================================================================================== */

template <int KT>
__global__ __launch_bounds__(TILE_THREADS, 2) void tile_matvec_f32(const TileArgs<float> a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *As = reinterpret_cast<float *>(smem_raw);  // [2][TILE * F32_LS]
    float *Bs = As + 2 * TILE * F32_LS;               // [2][TILE * F32_LS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1;  // wave row (0..1): rows wr*64 .. +63 of the tile
    const int wc = wave & 1;   // wave column
    const int r = lane & 31;
    const int h = lane >> 5;

    // work item -> (row block, column chunk); consecutive blocks share the column chunk (L2 reuse on every XCD)
    int ibl, jc;
    if (!decode_work_item(a, ibl, jc)) return;
    const int row0 = (a.ib_begin + ibl) * TILE;
    const int jt_begin = jc * a.jc_tiles;
    const int jt_end = min(jt_begin + a.jc_tiles, a.num_jt);
    const int ntiles = jt_end - jt_begin;
    if (ntiles <= 0) return;

    // staging: thread -> (row = tid/4 [+64], 8 consecutive floats at (tid%4)*8); 4 threads cover one 128-B line
    const int srow = tid >> 2;
    const int sseg = tid & 3;
    const float *Ag = a.Xr + static_cast<size_t>(row0 + srow) * a.ldx + sseg * 8;
    const size_t rstep = static_cast<size_t>(64) * a.ldx;
    const int lds_w = srow * F32_LS + sseg * 8;  // float offset of this thread's 8 floats in the LDS image

    float rowpart[2][16];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) rowpart[rb][i] = 0.0f;

    // rbf: c_i = -|x_i|^2/2 of the tile's 128 rows lives in LDS (re-read by every tile_init as 4-row float4 broadcasts);
    // keeping the lane's 32 values in registers instead pushes the kernel over the 256-VGPR budget of 2 waves per SIMD
    float *cis = Bs + 2 * TILE * F32_LS;  // [TILE]
    if constexpr (KT == KT_RBF) {
        if (tid < TILE) cis[tid] = a.cr[row0 + tid];
    }

    f32x16 acc[2][2];
    f32x4 sa[2][2], sb[2][2];

    auto stage_load = [&](int jt, int kc) {
        const float *Bg = a.Xc + static_cast<size_t>(jt * TILE + srow) * a.ldx + sseg * 8;
        const int ko = kc * F32_KC;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            sa[p][0] = *reinterpret_cast<const f32x4 *>(Ag + p * rstep + ko);
            sa[p][1] = *reinterpret_cast<const f32x4 *>(Ag + p * rstep + ko + 4);
            sb[p][0] = *reinterpret_cast<const f32x4 *>(Bg + p * rstep + ko);
            sb[p][1] = *reinterpret_cast<const f32x4 *>(Bg + p * rstep + ko + 4);
        }
    };
    auto stage_store = [&](int buf) {
        float *Aw = As + buf * TILE * F32_LS + lds_w;
        float *Bw = Bs + buf * TILE * F32_LS + lds_w;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            // the k-interleave (even k first, odd k second, see header comment) is already part of the HBM layout (k_interleave_features)
            *reinterpret_cast<f32x4 *>(Aw + p * 64 * F32_LS) = sa[p][0];
            *reinterpret_cast<f32x4 *>(Aw + p * 64 * F32_LS + 4) = sa[p][1];
            *reinterpret_cast<f32x4 *>(Bw + p * 64 * F32_LS) = sb[p][0];
            *reinterpret_cast<f32x4 *>(Bw + p * 64 * F32_LS + 4) = sb[p][1];
        }
    };

    // per-lane column data of a tile: d_j and (rbf) c_j = -|x_j|^2/2.  They are fetched ONE TILE AHEAD (col_prefetch at the
    // first k-chunk of the running tile, consumed by tile_init at its end) so their global-load latency is never exposed.
    float dj[2], cj[2], djn[2], cjn[2];
    bool padcol[2] = { false, false };  // polynomial with a negative degree only: (0*gamma+coef0)^degree may be inf on padding
    auto col_prefetch = [&](int jt) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int j = jt * TILE + wc * 64 + cb * 32 + r;
            djn[cb] = a.dvec[j];
            if constexpr (KT == KT_RBF) cjn[cb] = a.cc[j];
        }
    };
    auto tile_init = [&](int jt) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            dj[cb] = djn[cb];
            if constexpr (KT == KT_RBF) cj[cb] = cjn[cb];
            if constexpr (KT == KT_POLY) padcol[cb] = (a.degree < 0) && (jt * TILE + wc * 64 + cb * 32 + r >= a.ncols_valid);
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if constexpr (KT != KT_RBF) acc[rb][cb][i] = 0.0f;
                }
        if constexpr (KT == KT_RBF) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 civ = *reinterpret_cast<const f32x4 *>(cis + wr * 64 + rb * 32 + 8 * g4 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[rb][0][4 * g4 + e] = civ[e] + cj[0];
                        acc[rb][1][4 * g4 + e] = civ[e] + cj[1];
                    }
                }
        }
    };

    const int nsteps = ntiles * a.kchunks;
    stage_load(jt_begin, 0);
    col_prefetch(jt_begin);
    stage_store(0);
    __syncthreads();  // also publishes cis
    tile_init(jt_begin);

    int jt = jt_begin;
    int kc = 0;
    for (int s = 0; s < nsteps; ++s) {
        const int cur = s & 1;
        int njt = jt, nkc = kc + 1;
        if (nkc == a.kchunks) {
            nkc = 0;
            ++njt;
        }
        const bool has_next = (s + 1 < nsteps);
        const bool do_stage = has_next && !LSSVM_DBG(a, 1);
        if (do_stage) stage_load(njt, nkc);
        if (kc == 0 && jt + 1 < jt_end) col_prefetch(jt + 1);

        {
            const float *Ab = As + cur * TILE * F32_LS + (wr * 64 + r) * F32_LS + h * 4;
            const float *Bb = Bs + cur * TILE * F32_LS + (wc * 64 + r) * F32_LS + h * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a0 = *reinterpret_cast<const f32x4 *>(Ab + g * 8);
                const f32x4 a1 = *reinterpret_cast<const f32x4 *>(Ab + 32 * F32_LS + g * 8);
                const f32x4 b0 = *reinterpret_cast<const f32x4 *>(Bb + g * 8);
                const f32x4 b1 = *reinterpret_cast<const f32x4 *>(Bb + 32 * F32_LS + g * 8);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b1[t], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b0[t], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc[1][1], 0, 0, 0);
                }
            }
        }

        if (do_stage) stage_store(cur ^ 1);

        if (kc == a.kchunks - 1 && !LSSVM_DBG(a, 4)) {
            // epilogue of tile jt: K_ij = f(acc), row partial += K_ij * d_j  (vector ALU, fused; nothing is written)
            with_degree_class<KT>(a, [&](auto degc) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            float kv = LSSVM_DBG(a, 2) ? acc[rb][cb][i] : apply_kernel_function<KT, decltype(degc)::value>(acc[rb][cb][i], a);
                            if constexpr (KT == KT_POLY) {
                                if (padcol[cb]) kv = 0.0f;  // d_j is an exact zero there, but inf * 0 would be nan
                            }
                            rowpart[rb][i] = fmaf(kv, dj[cb], rowpart[rb][i]);
                        }
            });
            if (has_next) tile_init(njt);
        }
        if (!LSSVM_DBG(a, 8)) __syncthreads();
        jt = njt;
        kc = nkc;
    }

    // reduce the row partials over the 32 lanes that share the rows (same lane-half), then over the two wave columns
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = rowpart[rb][i];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            rowpart[rb][i] = v;
        }
    float *red = reinterpret_cast<float *>(smem_raw);  // [2][TILE]; the staging buffers are dead (barrier above)
    if (r == 0) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int i = 0; i < 16; ++i) red[wc * TILE + wr * 64 + rb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h] = rowpart[rb][i];
    }
    __syncthreads();
    if (tid < TILE) {
        a.partial[static_cast<size_t>(jc) * a.part_stride + ibl * TILE + tid] = red[tid] + red[TILE + tid];
    }
}

/* =====================================================================================================================
 * fp32 tile kernel, version 2 ("resident row panel"): for num_features <= 512 (1..8, 10, 12, 14, 16 k-chunks).
 *   - the work item's 128-row panel of X stays in REGISTERS for the whole sweep (flash-style): wave w owns rows 32w..32w+31
 *     as MFMA A fragments (16 VGPRs per 32 features), so the panel is read from L2 once per work item instead of once
 *     per column tile, and only the column side is staged on chip;
 *   - column k-chunks (128 rows x 32 features = 16 KiB) travel HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no
 *     staging registers, no ds_write), into a 4-slot ring, two chunks ahead; the hand-over of chunk s+1 (counted
 *     s_waitcnt vmcnt(4) + raw s_barrier, cdna_hip_programming.md section 5 "Pipelining across barriers") is executed in
 *     the MIDDLE of step s, in the shadow of its MFMAs, so a step starts reading its chunk with no wait at its head;
 *   - the LDS image is lane-linear (128-byte rows); bank conflicts are avoided by XOR-swizzling the 16-byte slot with
 *     (row >> 1) & 7 on the SOURCE address of the DMA and on the read address (rule 21 of the guide);
 *   - d_j and c_j of a tile arrive the same way from a packed [tile][256] array (k_pack_dc), so no ordinary global load
 *     (whose use would drain the DMA queue) sits inside the loop.
 * Each wave multiplies its 32 rows with all 128 columns of the tile: 4 accumulators of 32x32, 64 MFMAs + 16 ds_read_b128
 * per chunk.
 * ===================================================================================================================== */
// (V2_RING, V2_SLOT_BYTES, V2_DC_SLOTS, V2_LDS_BYTES: lssvm_device_common.hip.hpp -- shared with the split kernel)


template <int KT, int NKC, bool SYM>
__global__ __launch_bounds__(TILE_THREADS, (NKC <= 4 ? 2 : 1)) void tile_matvec_f32_v2(const TileArgs<float> a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char *ring = smem_raw;                                                          // [V2_RING][128 rows][128 B]
    char *dcs = smem_raw + V2_RING * V2_SLOT_BYTES;                                 // [V2_DC_SLOTS][256 floats]
    float *cis = reinterpret_cast<float *>(dcs + V2_DC_SLOTS * 1024);               // [128] c_i of the row panel (rbf)
    float *dis = cis + TILE;                                                        // [128] d_i of the row panel (SYM)
    float *colred = dis + TILE;                                                     // [2][4 waves][128] column sums of a tile (SYM)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int h = lane >> 5;

    // SYM: the kernel matrix is symmetric, so only the tiles on or below the diagonal are evaluated (as the reference does,
    // svm_kernel.cpp:39); an off-diagonal tile K_IJ contributes K_IJ d_J to the rows of I AND K_IJ^T d_I to the rows of J.
    // Work items come from a host-built list of the non-empty (row block, column chunk) pairs.
    int ibl, jc;
    if constexpr (SYM) {
        const int2 it = a.items[blockIdx.x];
        ibl = __builtin_amdgcn_readfirstlane(it.x);  // uniform, but loaded through the vector memory path: move to SGPRs so
        jc = __builtin_amdgcn_readfirstlane(it.y);   // that everything derived from it is scalar arithmetic
    } else {
        if (!decode_work_item(a, ibl, jc)) return;
    }
    const int ib = a.ib_begin + ibl;
    const int row0 = ib * TILE;
    const int jt_begin = jc * a.jc_tiles;
    const int jt_end = SYM ? min(jt_begin + a.jc_tiles, ib + 1) : min(jt_begin + a.jc_tiles, a.num_jt);
    const int ntiles = jt_end - jt_begin;
    if (ntiles <= 0) return;
    const int nsteps = ntiles * NKC;
    // record index of (ib, jt) in the packed strictly-lower-triangular column slab of this device
    const long rec0 = SYM ? (static_cast<long>(ib) * (ib - 1) / 2 - a.pair_origin) : 0;

    // ---- the row panel: A fragments of this wave's 32 rows, all features (HBM layout is k-interleaved) ----
    f32x4 afrag[4 * NKC];
    {
        const float *xr = a.Xr + static_cast<size_t>(row0 + wave * 32 + r) * a.ldx + 4 * h;
#pragma unroll
        for (int m = 0; m < 4 * NKC; ++m) afrag[m] = *reinterpret_cast<const f32x4 *>(xr + 8 * m);
    }
    if constexpr (KT == KT_RBF) {
        if (tid < TILE) cis[tid] = a.cr[row0 + tid];
    }
    if constexpr (SYM) {
        if (tid < TILE) dis[tid] = a.dvec[row0 + tid];
    }
    // make the compiler retire these ordinary loads HERE, before any LDS-DMA is in flight
#pragma unroll
    for (int m = 0; m < 4 * NKC; ++m) asm volatile("" : "+v"(afrag[m]));

    // ---- LDS-DMA addressing ----
    // instruction q = 4*wave + i moves rows 8q .. 8q+7 of a chunk; lane L -> row 8q + L/8, physical 16-B slot L%8, which
    // holds logical slot (L%8) ^ ((row >> 1) & 7)
    // The source address of a DMA is (uniform 64-bit base in SGPRs) + (32-bit per-lane byte offset): the saddr form of
    // global_load_lds, so a piece costs no 64-bit vector address arithmetic and one VGPR
    unsigned dma_off[4];  // byte offset of this lane's 16 bytes inside a (tile, chunk) = 4 * (row * ldx + 4 * logical_slot)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * (4 * wave + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        dma_off[i] = 4u * static_cast<unsigned>(row * a.ldx + 4 * c);
    }
    auto issue_chunk = [&](int step) {  // step = linear (tile, chunk) index of this work item
        if (LSSVM_DBG(a, 16) && step > 3) return;  // ablation: no DMA after the prologue
        const int t = LSSVM_DBG(a, 1) ? 0 : step / NKC;  // ablation bit 1: always the same (L2-resident) tile
        const int kc = LSSVM_DBG(a, 1) ? 0 : step - t * NKC;
        const char *base = sgpr_ptr(a.Xc + static_cast<size_t>(jt_begin + t) * TILE * a.ldx + kc * 32);
        char *slot = ring + (step % V2_RING) * V2_SLOT_BYTES + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t) (base + lane_off(dma_off[i])), (lds_ptr_t) (slot + i * 1024), 16, 0, 0);
        }
    };
    // one of the four DMA instructions of a chunk (steady state: spread over the MFMA groups that follow the hand-over, an
    // LDS-DMA issue costs the wave ~60-100 cycles, MI355X_MICROARCH.md "LDS-DMA piece issue cost")
    auto issue_chunk_part = [&](int step, int i) {
        const int t = step / NKC;
        const int kc = step - t * NKC;
        const char *base = sgpr_ptr(a.Xc + static_cast<size_t>(jt_begin + t) * TILE * a.ldx + kc * 32);
        char *slot = ring + (step % V2_RING) * V2_SLOT_BYTES + wave * 4096;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t) (base + lane_off(dma_off[i])), (lds_ptr_t) (slot + i * 1024), 16, 0, 0);
    };
    auto issue_dc = [&](int t) {  // (d_j | c_j) of tile jt_begin + t: 1 KiB, each wave moves a quarter with 16 lanes
        if (lane < 16) {
            const char *src = sgpr_ptr(a.dc + static_cast<size_t>(jt_begin + t) * 256) + __builtin_amdgcn_readfirstlane(wave * 256);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t) (src + 16u * (lane_off(threadIdx.x) & 15u)), (lds_ptr_t) (dcs + (t % V2_DC_SLOTS) * 1024 + wave * 256), 16, 0, 0);
        }
    };

    // ---- read addressing: lane (r, h) reads 16-B logical slot 2*mm + h of row cb*32 + r (swizzle depends on r only) ----
    int rd_off[4];
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) rd_off[mm] = r * 128 + (((2 * mm + h) ^ ((r >> 1) & 7)) << 4);

    float rowpart[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) rowpart[i] = 0.0f;
    f32x16 acc[4];
    float dj[4], cj[4];
    bool padcol[4] = { false, false, false, false };

    // ---- prologue: chunks 0, 1, 2 (each preceded by the record of the tile that starts with it) ----
    issue_dc(0);
    issue_chunk(0);
#pragma unroll
    for (int pre = 1; pre <= 2; ++pre) {
        if (pre < nsteps) {
            if (pre % NKC == 0) issue_dc(pre / NKC);
            issue_chunk(pre);
        }
    }
    // chunk 0 (and record 0, and cis) complete: everything but the DMA instructions of the younger chunks is done
    if (nsteps >= 3) {
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    } else if (nsteps == 2) {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    f32x4 bcur[4];  // B fragments of the group about to be multiplied (double buffered against bnext in the loop)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) bcur[cb] = *reinterpret_cast<const f32x4 *>(ring + cb * 4096 + rd_off[0]);

    // ---- hand-over of the NEXT chunk, executed in the MIDDLE of a step (in the shadow of that step's MFMAs) ----
    // Called half-way through step `step`: this wave's DMA of chunk step+1 (issued 2 steps ago) is complete once all but its
    // 4 youngest DMA instructions (chunk step+2) are done; the barrier makes every wave's part visible, so the next step
    // starts reading at once, with no wait and no barrier at its head.  Ring of 4 slots: the DMA issued here (chunk step+3)
    // overwrites the slot of chunk step-1, which every wave finished reading before it arrived at this barrier.
    // CHECKED = false: steady state, step + 3 < nsteps is known, the code is branch free (one basic block per tile, so the
    // compiler can place the scalar address arithmetic and the DMA issue in the shadow of the MFMAs); CHECKED = true: the
    // last tiles of the work item.
    auto handover = [&](int step, int kc_plus3_mod, auto checked) {
        constexpr bool CHECKED = decltype(checked)::value;
        if constexpr (!CHECKED) {
            if (!LSSVM_DBG(a, 16)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (!LSSVM_DBG(a, 8)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // the record of a tile is issued right BEFORE the first chunk of that tile: "chunk landed" implies "record landed"
            if (kc_plus3_mod == 0) issue_dc((step + 3) / NKC);
            // the four DMA instructions of chunk step+3 follow one by one between the MFMAs of this step's second half
        } else {
            if (step + 1 < nsteps) {
                if (step + 2 < nsteps) {
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (step + 3 < nsteps) {
                    if (kc_plus3_mod == 0) issue_dc((step + 3) / NKC);
                    issue_chunk(step + 3);
                }
            }
        }
    };

    // SYM: the four waves' column sums of tile t (written to colred by its epilogue, made visible by the next barrier) are
    // added in a fixed order and stored to the tile's record of the column slab
    auto flush_cols = [&](int t) {
        if (tid < TILE) {
            const float *cr_ = colred + (t & 1) * 512;
            const float sum = (cr_[tid] + cr_[128 + tid]) + (cr_[256 + tid] + cr_[384 + tid]);
            // (an explicit GLOBAL pointer: through the generic one the store is a flat_store, which counts in lgkmcnt as well and completes out of order)
            auto *rec = (__attribute__((address_space(1))) float *) (a.colslab + (rec0 + jt_begin + t) * TILE);  // uniform base + 32-bit lane offset
            rec[static_cast<unsigned>(tid)] = sum;
        }
    };

    auto tile_body = [&](int t, auto checked) {
        const int s0 = t * NKC;
        const bool tile_sym = SYM && (jt_begin + t < ib);  // strictly below the diagonal
        {
            // tile_init: per-lane column data + accumulator start values (the record became visible at the last hand-over)
            const float *dcr = reinterpret_cast<const float *>(dcs + (t % V2_DC_SLOTS) * 1024);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                dj[cb] = dcr[cb * 32 + r];
                if constexpr (KT == KT_RBF) cj[cb] = dcr[128 + cb * 32 + r];
                if constexpr (KT == KT_POLY) padcol[cb] = (a.degree < 0) && ((jt_begin + t) * TILE + cb * 32 + r >= a.ncols_valid);
            }
            // rbf: the accumulators start at c_i + c_j (vector adds; producing the sum with one extra MFMA per accumulator --
            // A = (c_i, 1), B = (1, c_j) -- was measured 0.8 % slower at c5: the adds overlap with the other workgroup's MFMAs)
            if constexpr (KT == KT_RBF) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 civ = *reinterpret_cast<const f32x4 *>(cis + wave * 32 + 8 * g4 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb) acc[cb][4 * g4 + e] = civ[e] + cj[cb];
                }
            }
            // the other kernels start the chain with the constant 0 as the C operand of the first MFMA (no v_mov per register)
        }
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            const int step = s0 + kc;
            const char *slot = ring + (step % V2_RING) * V2_SLOT_BYTES;
            const char *slot_next = ring + ((step + 1) % V2_RING) * V2_SLOT_BYTES;
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                // software prefetch of the NEXT group's B fragments (next chunk for mm == 3: visible since this step's hand-over),
                // issued before the hand-over barrier so that LDS latency and barrier skew hide behind the 16 MFMAs below
                f32x4 bnext[4];
                if (mm < 3) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) bnext[cb] = *reinterpret_cast<const f32x4 *>(slot + cb * 4096 + rd_off[mm + 1]);
                }
                if (mm == 2) {
                    if constexpr (SYM) {
                        // the colred writes of the previous tile's epilogue must have completed before the barrier publishes them
                        if (kc == 0 && t > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    handover(step, (kc + 3) % NKC, checked);
                    if constexpr (SYM) {
                        if (kc == 0 && t > 0) flush_cols(t - 1);  // every tile before the last one of an item is off-diagonal
                    }
                }
                if (mm == 3) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) bnext[cb] = *reinterpret_cast<const f32x4 *>(slot_next + cb * 4096 + rd_off[0]);
                }
                const f32x4 av = afrag[4 * kc + mm];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        if (KT != KT_RBF && kc == 0 && mm == 0 && tt == 0) {
                            const f32x16 zero = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
                            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tt], bcur[cb][tt], zero, 0, 0, 0);
                        } else {
                            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tt], bcur[cb][tt], acc[cb], 0, 0, 0);
                        }
                    }
                    if constexpr (!decltype(checked)::value) {
                        if (mm >= 2 && (tt & 1) == 0) issue_chunk_part(step + 3, (mm - 2) * 2 + (tt >> 1));
                    }
                }
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) bcur[cb] = bnext[cb];
            }
        }
        // epilogue of the tile: K_ij = f(acc), row partial += K_ij * d_j; SYM, off-diagonal tile: column partial += K_ij * d_i
        // (vector ALU, fused; the Gram tile itself is never written)
        if (!LSSVM_DBG(a, 4))
        {  // (the polynomial degree class is a template parameter here: KT_POLY2 / KT_POLY3 / generic KT_POLY)
            auto epilogue = [&](auto with_cols) {
                constexpr bool COLS = decltype(with_cols)::value;
                float di[16];
                float colacc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
                if constexpr (COLS) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4 dv = *reinterpret_cast<const f32x4 *>(dis + wave * 32 + 8 * g4 + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) di[4 * g4 + e] = dv[e];
                    }
                }
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float kv = apply_kernel_function<v2_base_kt(KT), v2_degree_class(KT)>(acc[cb][i], a);
                        if constexpr (KT == KT_POLY) {
                            if (padcol[cb]) kv = 0.0f;
                        }
                        rowpart[i] = fmaf(kv, dj[cb], rowpart[i]);
                        if constexpr (COLS) colacc[cb] = fmaf(kv, di[i], colacc[cb]);
                    }
                if constexpr (COLS) {
                    float *cw = colred + (t & 1) * 512 + wave * 128;
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) colacc[cb] = sum_with_lane_xor32(colacc[cb]);  // the two lane halves hold different rows (no LDS round trip)
                    if (h == 0) {
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb) cw[cb * 32 + r] = colacc[cb];
                    }
                }
            };
            if (tile_sym) {
                epilogue(std::true_type{});
            } else {
                epilogue(std::false_type{});
            }
        }
    };

    // steady state: every tile whose last step still has step + 3 < nsteps; then the (1..3) tail tiles with the checked hand-over
    constexpr int TAIL_TILES = (3 + NKC - 1) / NKC;
    const int nmain = ntiles > TAIL_TILES ? ntiles - TAIL_TILES : 0;
    int t = 0;
    for (; t < nmain; ++t) tile_body(t, std::false_type{});
    for (; t < ntiles; ++t) tile_body(t, std::true_type{});
    if constexpr (SYM) {
        if (jt_begin + ntiles - 1 < ib) {  // the last tile was off-diagonal: publish its column sums
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            flush_cols(ntiles - 1);
        }
    }

    // every wave owns its rows: reduce over the 32 lanes of a lane-half and store
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float v = rowpart[i];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 1);
        rowpart[i] = v;
    }
    if (r == 0) {
        float *dst = a.partial + static_cast<size_t>(jc) * a.part_stride + ibl * TILE + wave * 32 + 4 * h;
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[(i & 3) + 8 * (i >> 2)] = rowpart[i];
    }
}
/* =====================================================================================================================
 * Direct-form RBF on the vector ALU (fp32): accumulates (x_i - x_j)^2 exactly as the reference does
 * (HIP/svm_kernel.hip.hpp:247, operators.hpp:161-171).  1 sub + 1 fma per (i, j, feature): at most half of the fp32 FMA
 * peak.  Kept as the formula-exact alternative to the matrix-core path (option "rbf_form" = 1) and as its on-device
 * cross-check.  Each thread owns an 8 x 8 register tile of a 128 x 128 workgroup tile; operands come from the same
 * k-chunked LDS images (no k interleave needed here; rows padded to 33 floats).
 * ===================================================================================================================== */
constexpr int DIR_KC = 32;
constexpr int DIR_LS = 33;

template <bool ONE_INSTANCE = true>  // (a template so that only the translation unit that launches it carries it)
__global__ __launch_bounds__(TILE_THREADS, 2) void tile_matvec_rbf_direct_f32(const TileArgs<float> a) {
    __shared__ float As[TILE * DIR_LS];
    __shared__ float Bs[TILE * DIR_LS];
    __shared__ float red[16][TILE];

    const int tid = threadIdx.x;
    const int tx = tid & 15;  // column group: columns tx + 16*c
    const int ty = tid >> 4;  // row group:    rows    ty + 16*rr
    int ibl, jc;
    if (!decode_work_item(a, ibl, jc)) return;
    const int row0 = (a.ib_begin + ibl) * TILE;
    const int jt_begin = jc * a.jc_tiles;
    const int jt_end = min(jt_begin + a.jc_tiles, a.num_jt);
    if (jt_end <= jt_begin) return;

    float rowpart[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rowpart[i] = 0.0f;

    const int srow = tid >> 3;  // 0..31
    const int sseg = tid & 7;   // 4 floats each
    for (int jt = jt_begin; jt < jt_end; ++jt) {
        float acc[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = 0.0f;
        for (int kc = 0; kc < a.kchunks; ++kc) {
            __syncthreads();
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = srow + 32 * p;
                const f32x4 va = *reinterpret_cast<const f32x4 *>(a.Xr + static_cast<size_t>(row0 + row) * a.ldx + kc * DIR_KC + sseg * 4);
                const f32x4 vb = *reinterpret_cast<const f32x4 *>(a.Xc + static_cast<size_t>(jt * TILE + row) * a.ldx + kc * DIR_KC + sseg * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    As[row * DIR_LS + sseg * 4 + e] = va[e];
                    Bs[row * DIR_LS + sseg * 4 + e] = vb[e];
                }
            }
            __syncthreads();
#pragma unroll 4
            for (int k = 0; k < DIR_KC; ++k) {
                float av[8], bv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) av[i] = As[(ty + 16 * i) * DIR_LS + k];
#pragma unroll
                for (int j = 0; j < 8; ++j) bv[j] = Bs[(tx + 16 * j) * DIR_LS + k];
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float diff = av[i] - bv[j];
                        acc[i][j] = fmaf(diff, diff, acc[i][j]);
                    }
            }
        }
        // epilogue: exp(-gamma * dist^2) * d_j ; for THIS kernel the gamma field carries -gamma*log2(e)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float dv = a.dvec[jt * TILE + tx + 16 * j];
#pragma unroll
            for (int i = 0; i < 8; ++i) rowpart[i] = fmaf(__builtin_amdgcn_exp2f(acc[i][j] * a.gamma), dv, rowpart[i]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) red[tx][ty + 16 * i] = rowpart[i];
    __syncthreads();
    if (tid < TILE) {
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c) s += red[c][tid];
        a.partial[static_cast<size_t>(jc) * a.part_stride + ibl * TILE + tid] = s;
    }
}

}  // namespace lssvm
