/*
 * lssvm_tile_f32_split.hip.hpp -- fp32 tile kernel on the bf16 matrix cores ("bf16x6", option gram_mode = 1: the DEFAULT for <= 256 features).
 *
 * Every fp32 operand is split EXACTLY into three bf16 planes, x = hi + mid + lo (8 + 8 + 8 mantissa bits), once at set-up
 * (k_split_bf16x3).  The Gram tile is accumulated in fp32 from the six plane products of significance >= 2^-16,
 *     hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid,
 * on v_mfma_f32_32x32x16_bf16 (products of two bf16 are exact in fp32; the dropped products are below 2^-24 |x||y|, the size
 * of one fp32 rounding).  Numerically this is an fp32 contraction with a different summation order: its distance from the
 * float64 Gram matrix equals that of the v_mfma_f32_32x32x2_f32 chain (tests/test_gpu_parity.py asserts the
 * same 32-eps kernel-level bar).  Why: the bf16 MFMA moves 16x the multiply-adds per instruction (6 of them = 3/8 of the time of
 * the f32 MFMAs) AND, unlike the f32 MFMA, co-issues with the vector ALU (tests/tools/microbench_f64.hip), so the epilogue hides.
 * Structure: tile_matvec_f32_v2 with (64-feature chunk, plane) as the step -- same LDS-DMA ring, swizzle, hand-over, records,
 * epilogue and symmetric variant.  For num_features <= 384 (1 ... 6 chunks of 64 features; above 128 one workgroup per CU).
 */
#pragma once

#include "lssvm_device_common.hip.hpp"

#ifdef LSSVM_USE_SCHED_BARRIER
#define LSSVM_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define LSSVM_SCHED_BARRIER() ((void) 0)
#endif

namespace lssvm {

#include "lssvm_s6w_groups.inc"

/* one plane product on the matrix cores: bf16 planes (bf16x6) or f16 planes (f16x3); same operand maps, same cycles */
using f16x8 = _Float16 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ f32x4 plane_mfma(const bf16x8 &av, const bf16x8 &bv, const f32x4 &c) {
    if constexpr (F16) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c, 0, 0, 0);
    }
}

/*
 * The split kernel on v_mfma_f32_16x16x32_{bf16,f16}.  (Round 1's 32x32x16 form of it -- each wave 4 accumulators of 32 x 32 -- was retired in
 * round 3: same matrix-core cycles, but in the power-bound regime this kernel runs in MI355X holds a higher clock on the 16x16x32 shape,
 * MI355X_MICROARCH.md "DVFS give-back" (7); measured 511 against 477 ms at 1 000 000 x 128, profiles/archive/r02_ab_mfma_shape_c5.log.)
 * A wave's 32 rows x 128 columns are 2 x 8 blocks of 16 x 16 (64 accumulator registers), one B fragment (16 columns x 32 features, one
 * ds_read_b128) feeds up to 6 MFMAs (2 row blocks x up to 3 row planes).
 * Operand maps (cdna_hip_programming.md section 3): lane l holds A[row l & 15][k = 8 (l >> 4) + j] and B[k = 8 (l >> 4) + j][col l & 15], j = 0..7;
 * the result has col = l & 15, row = 4 (l >> 4) + reg.
 * A step (= one plane of one 64-feature chunk) is processed in four groups mm = (k32 step kk = mm >> 1, column half cbh = mm & 1) of four column
 * blocks each, so that the B fragments stay double buffered in 2 x 16 registers and the hand-over sits in the middle of a step as before.
 */
template <int KT, int NK64, bool SYM, bool HAND, int PL>
__device__ __forceinline__ void s6w_body(const TileArgs<float> &a) {
    static_assert(PL == 3 || PL == 2, "three bf16 planes (bf16x6) or two f16 planes (f16x3)");
    constexpr bool F16 = PL == 2;
    // KT_RBFG ("grid planes", round 5; f16, hand-scheduled groups only): THREE column planes (h | s1 | s2, all carrying the scale sigma) and FOUR phases per tile, each
    // over all 64-feature chunks:   0: h x h      1: h x (s1, s2)      2: s1 x (h, s1)      3: s2 x h      (column plane x row planes)
    // -- six plane products like bf16x6, the h plane streamed twice -- because ORDER matters here: the accumulators start from sigma^2 (ch_i + ch_j) (ch = -|h|^2 / 2,
    // multiples of g^2 / 2) and take ALL h.h products first (multiples of g^2): every partial sum is exactly representable, so phase 0 leaves
    // -sigma^2 |h_i - h_j|^2 / 2 EXACTLY -- small for near pairs -- and the remaining terms are added at the magnitude of the result, not of the norms.
    constexpr bool GRID = KT == KT_RBFG;
    static_assert(!GRID || PL == 2, "the grid-plane kernel exists with f16 planes only");
    constexpr int NKC = GRID ? 4 * NK64 : PL * NK64;  // steps per tile: for every 64-feature chunk the planes hi, mid (, lo) -- grid planes: four phases x chunks
    constexpr auto col_plane_of = [](int kc) constexpr { return GRID ? (kc / NK64 == 0 ? 0 : kc / NK64 - 1) : kc % PL; };
    constexpr auto chunk_of = [](int kc) constexpr { return GRID ? kc % NK64 : kc / PL; };
    constexpr auto nq_of = [](int kc) constexpr { return GRID ? ((kc / NK64 == 1 || kc / NK64 == 2) ? 2 : 1) : PL - kc % PL; };
    // ROW planes held in registers.  bf16x6: the three planes.  f16x3: the two planes -- or, for rbf (which cannot pre-scale the data: the
    // chain must leave the exponent itself), the SHIFTED planes P0 = 2^-6 hi, P1 = 2^6 mid, P2 = 2^6 hi (k_split_f16x2): the columns stream
    // (P0, P1), the rows hold (P2, P1) against column plane P0 and P0 against column plane P1, so that every product carries the net scale 1
    // while mid stays a normal f16 for entries down to 2^-8 instead of 2^-2.
    constexpr int PLA = F16 ? ((KT == KT_RBF || KT == KT_RBFF || GRID) ? 3 : 2) : 3;
    // row plane of the q-th product of column plane p
    constexpr auto row_plane = [](int p, int q) constexpr { return (F16 && PLA == 3) ? (p == 0 ? (q == 0 ? 2 : 1) : 0) : q; };
    // ... of step kc (grid planes: by phase)
    constexpr auto row_plane_of = [row_plane](int kc, int q) constexpr {
        if (GRID) {
            const int ph = kc / NK64;
            return ph == 0 ? 0 : (ph == 1 ? (q == 0 ? 1 : 2) : (ph == 2 ? (q == 0 ? 0 : 1) : 0));
        }
        return row_plane(kc % PL, q);
    };
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char *ring = smem_raw;                                                          // [V2_RING][128 rows][128 B]
    char *dcs = smem_raw + V2_RING * V2_SLOT_BYTES;                                 // [V2_DC_SLOTS][256 floats]
    float *cis = reinterpret_cast<float *>(dcs + V2_DC_SLOTS * 1024);               // [128] c_i of the row panel (rbf)
    float *dis = cis + TILE;                                                        // [128] d_i of the row panel (SYM)
    float *colred = dis + TILE;                                                     // [2][4 waves][128] column sums of a tile (SYM)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15;  // row of an A block / column of a B block
    const int g = lane >> 4;  // k group (operands), row group (results)

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
    const int nsteps = ntiles * NKC;
    const long rec0 = SYM ? (static_cast<long>(ib) * (ib - 1) / 2 - a.pair_origin) : 0;

    // ---- the row panel: A fragments of this wave's 32 rows (two blocks of 16), all features, all three planes: lane (r, g) holds features
    // 32 kk + 8 g .. + 7 of row 16 rb + r for k32 step kk ----
    bf16x8 afrag[PLA][2 * NK64][2];
#pragma unroll
    for (int p = 0; p < PLA; ++p) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const uint16_t *xr = a.Xr16 + p * a.plane_stride_r + static_cast<size_t>(row0 + wave * 32 + 16 * rb + r) * a.ldx16 + 8 * g;
#pragma unroll
            for (int kk = 0; kk < 2 * NK64; ++kk) afrag[p][kk][rb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(xr + 32 * kk));
        }
    }
    if constexpr (KT == KT_RBF || KT == KT_RBFF || GRID) {
        if (tid < TILE) cis[tid] = a.cr[row0 + tid];  // (grid planes: sigma^2 ch_i)
    }
    if constexpr (SYM) {
        if (tid < TILE) dis[tid] = GRID ? a.dvec[row0 + tid] * a.er[row0 + tid] : a.dvec[row0 + tid];  // (grid planes: the row's folded factor E_i rides on d_i)
    }
    // make the compiler retire these ordinary loads HERE, before any LDS-DMA is in flight
#pragma unroll
    for (int p = 0; p < PLA; ++p)
#pragma unroll
        for (int kk = 0; kk < 2 * NK64; ++kk)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) asm volatile("" : "+v"(afrag[p][kk][rb]));

    // ---- LDS-DMA addressing: identical to tile_matvec_f32_s6 (the LDS image does not depend on the MFMA shape) ----
    unsigned dma_off[4];
#pragma unroll
    for (int i = 0; i < (GRID ? 2 : 4); ++i) {
        const int row = 8 * (4 * wave + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        dma_off[i] = 2u * static_cast<unsigned>(row * a.ldx16 + 8 * c);
    }
    const unsigned ring_lds = static_cast<unsigned>(reinterpret_cast<size_t>(ring));  // the low half of a generic LDS address is the LDS address
    const unsigned dma_lds = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(ring_lds + static_cast<unsigned>(wave) * 4096u)));  // this wave's quarter of ring slot 0
    const unsigned dc_lds = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(ring_lds + V2_RING * V2_SLOT_BYTES + static_cast<unsigned>(wave) * 256u)));  // this wave's quarter of record slot 0
    auto issue_chunk = [&](int step) {
        if (LSSVM_DBG(a, 16) && step > 3) return;  // ablation: no DMA after the prologue
        const int t = LSSVM_DBG(a, 1) ? 0 : step / NKC;  // ablation bit 1: always the same (L2-resident) tile
        const int kc = LSSVM_DBG(a, 1) ? 0 : step - t * NKC;
        const char *base = sgpr_ptr(a.Xc16 + col_plane_of(kc) * a.plane_stride + static_cast<size_t>(jt_begin + t) * TILE * a.ldx16 + chunk_of(kc) * 64);
        const unsigned slot = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(dma_lds + static_cast<unsigned>(step % V2_RING) * V2_SLOT_BYTES)));
        // (grid planes: pieces i and i + 2 lie 16 rows apart with the same swizzle, so two lane offsets + a uniform 32 ldx16 bytes on the base do for four -- the
        // 128-feature symmetric instantiation has no two registers to spare)
        static_for<0, 4>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            if constexpr (GRID) {
                lds_dma16<i * 1024>(dma_off[i & 1], sgpr_ptr(base + (i >> 1) * 32 * a.ldx16), slot);
            } else {
                lds_dma16<i * 1024>(dma_off[i], base, slot);
            }
        });
    };
    // steady state: the chunk (tile t or t + 1, plane-chunk KC known at compile time) costs two scalar adds per DMA instead of the divisions and
    // 64-bit multiplies of the generic form; `xc_tile` = first byte of column tile t in plane 0 (uniform), advanced once per tile
    const size_t tile_bytes = static_cast<size_t>(TILE) * a.ldx16 * 2;
    const size_t plane_bytes = a.plane_stride * 2;
    const char *xc_tile = reinterpret_cast<const char *>(a.Xc16) + static_cast<size_t>(jt_begin) * tile_bytes;
    auto issue_part_static = [&](auto kc3_c, unsigned slot_idx, auto i_c) {  // kc3 = kc + 3 of the issuing step, slot_idx = (step + 3) % V2_RING
        constexpr int KC3 = decltype(kc3_c)::value;
        constexpr int KC = KC3 % NKC;
        constexpr int i = decltype(i_c)::value;
        if (LSSVM_DBG(a, 16)) return;
        if (LSSVM_DBG(a, 64) && i != 0) return;   // bit 64: a quarter of the DMA instructions (timing only)
        if (LSSVM_DBG(a, 128) && wave != 0) return;  // bit 128: only wave 0 issues DMA
        const char *base = xc_tile + (KC3 / NKC) * tile_bytes + col_plane_of(KC) * plane_bytes + chunk_of(KC) * 128;  // (f16x3 at 64 features: NKC = 2, three steps ahead can be TWO tiles ahead)
        if constexpr (GRID) {
            lds_dma16<i * 1024>(dma_off[i & 1], sgpr_ptr(base + (i >> 1) * 32 * a.ldx16), dma_lds + slot_idx * V2_SLOT_BYTES);
        } else {
            lds_dma16<i * 1024>(dma_off[i], sgpr_ptr(base), dma_lds + slot_idx * V2_SLOT_BYTES);
        }
    };
    auto issue_dc = [&](int t) {
        if (lane < 16) {
            const char *src = sgpr_ptr(a.dc + static_cast<size_t>(jt_begin + t) * 256) + __builtin_amdgcn_readfirstlane(wave * 256);
            lds_dma16<0>(16u * (lane_off(threadIdx.x) & 15u), sgpr_ptr(src), static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(dc_lds + static_cast<unsigned>(t % V2_DC_SLOTS) * 1024u))));
        }
    };

    // ---- read addressing: lane (r, g) reads the 16-B logical slot 4 kk + g of row 16 cb + r; the swizzle (row >> 1) & 7 depends on r only, the
    // column block is an immediate offset.  Conflict free for ds_read_b128 (its 16-lane groups see eight distinct XOR values twice, on both
    // halves of the 256-byte bank line) ----
    int rd_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) rd_off[kk] = r * 128 + (((4 * kk + g) ^ ((r >> 1) & 7)) << 4);

    float rowpart[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rowpart[i] = 0.0f;
    f32x4 acc[2][8];
    f32x4 civ0[2] = { { 0.f, 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f, 0.f } };  // KT_RBFF: c_i of the wave's two row blocks, the C operand of the first MFMAs
    bool padcol[8] = { false, false, false, false, false, false, false, false };

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
    if (nsteps >= 3) {
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    } else if (nsteps == 2) {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    if constexpr (KT == KT_RBFF) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) civ0[rb] = *reinterpret_cast<const f32x4 *>(cis + wave * 32 + 16 * rb + 4 * g);
    }
    // B fragments: two buffers used alternately by the groups (group mm multiplies bbuf[mm & 1] while the reads into bbuf[(mm + 1) & 1] are
    // in flight).  The reads are ordinary loads: the compiler's own waits guard them.  (Hand-issued ds_read_b128 + counted s_waitcnt were
    // tried and are WRONG here: the register allocator copies a destination register at a block boundary before the wait, i.e. before the
    // data has landed -- measured as NaNs.)  What the hand-written form was after is obtained with a scheduling barrier instead: it keeps
    // all four reads of the next group at the head of the current group, so that the lgkmcnt(0) the compiler puts in front of the next
    // group's first MFMA waits for requests that are a whole group old, not for ones it sank to that very spot.
    // HAND (num_features <= 128, two waves per SIMD): the B fragments live in v[224:255], which the compiler does not own (the kernel is
    // compiled with amdgpu_num_vgpr(224)); the groups are the hand-scheduled blocks of lssvm_s6w_groups.inc -- reads of the next group's
    // fragments, a COUNTED wait, the MFMAs.  Nothing can copy an in-flight register, because nothing else knows these registers.
    f32x4 bbuf[2][4];
    const unsigned rdl[2] = { ring_lds + static_cast<unsigned>(rd_off[0]), ring_lds + static_cast<unsigned>(rd_off[1]) };
    if constexpr (HAND) {
        s6w_fill_b0<0, 2048, 4096, 6144>(rdl[0]);
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) bbuf[0][c] = *reinterpret_cast<const f32x4 *>(ring + c * 2048 + rd_off[0]);
    }

    auto handover = [&](int step, int kc_plus3_mod, int tile_of_step_plus3, auto checked) {
        constexpr bool CHECKED = decltype(checked)::value;
        if constexpr (!CHECKED) {
            if (!LSSVM_DBG(a, 16) && !LSSVM_DBG(a, 32)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // bit 32: DMA issued but never waited for
            if (!LSSVM_DBG(a, 8)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kc_plus3_mod == 0) issue_dc(tile_of_step_plus3);
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
                    if (kc_plus3_mod == 0) issue_dc(tile_of_step_plus3);
                    issue_chunk(step + 3);
                }
            }
        }
    };

    auto flush_cols = [&](int t) {
        if (tid < TILE) {
            const float *cr_ = colred + (t & 1) * 512;
            const float sum = (cr_[tid] + cr_[128 + tid]) + (cr_[256 + tid] + cr_[384 + tid]);
            // uniform base in SGPRs + 32-bit lane offset (a 64-bit per-lane pointer would be hoisted out of the tile loop and spilled)
            // (and an explicit GLOBAL pointer: through the generic one the store is a flat_store, which counts in lgkmcnt as well and completes out of order)
            auto *rec = (__attribute__((address_space(1))) float *) const_cast<char *>(sgpr_ptr(a.colslab + (rec0 + jt_begin + t) * TILE));
            rec[lane_off(static_cast<unsigned>(tid))] = sum;
        }
    };

    auto tile_body = [&](int t, auto checked) {
        const int s0 = t * NKC;
        const bool tile_sym = SYM && (jt_begin + t < ib);  // strictly below the diagonal
        {
            // (d_j is read from the record in the epilogue, c_j only here: neither lives in registers across the MFMA steps -- this kernel
            // has no register to spare, and a scratch reload in the loop drains the LDS-DMA queue with its vmcnt(0))
            const float *dcr = reinterpret_cast<const float *>(dcs + (t % V2_DC_SLOTS) * 1024);
            if constexpr (KT == KT_POLY) {
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) padcol[cb] = (a.degree < 0) && ((jt_begin + t) * TILE + cb * 16 + r >= a.ncols_valid);
            }
            // (KT_RBFF, folded records: the accumulators start from c_i, which the FIRST MFMA of every accumulator takes as its C operand -- the
            // same four values for all eight column blocks of a row block, loaded once per work item into civ0: no start-value instruction and
            // no LDS read at the head of a tile.  c_j comes in as the factor 2^c_j of the record.)
            if constexpr (KT == KT_RBF || GRID) {  // the accumulators start at c_i + c_j (grid planes: sigma^2 (ch_i + ch_j), an exact sum)
                f32x4 civ[2];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) civ[rb] = *reinterpret_cast<const f32x4 *>(cis + wave * 32 + 16 * rb + 4 * g);
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) {
                    const float cjv = dcr[128 + cb * 16 + r];
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[rb][cb][e] = civ[rb][e] + cjv;
                }
            }
        }
        const unsigned phase = static_cast<unsigned>(s0) & (V2_RING - 1);  // ring slot of the tile's first step (uniform)
        static_for<0, NKC>([&](auto kc_c) {
            constexpr int kc = decltype(kc_c)::value;
            constexpr int chunk = chunk_of(kc), plane = col_plane_of(kc);
            const int step = s0 + kc;
            const unsigned slot_off = ((phase + kc) & (V2_RING - 1)) * V2_SLOT_BYTES;
            const unsigned slot_next_off = ((phase + kc + 1) & (V2_RING - 1)) * V2_SLOT_BYTES;
            const char *slot = ring + slot_off;
            const char *slot_next = ring + slot_next_off;
            static_for<0, 4>([&](auto mm_c) {
                constexpr int mm = decltype(mm_c)::value;
                constexpr int kk = mm >> 1, cbh = mm & 1;
                // is there a group after this one?  (steady state: always; last tiles of the work item: not after the very last group)
                const bool more = !decltype(checked)::value || mm < 3 || step + 1 < nsteps;
                if constexpr (HAND) {
                    constexpr int NQ = nq_of(kc);
                    constexpr int Z = ((KT != KT_RBF && !GRID) && kc == 0 && kk == 0) ? (KT == KT_RBFF ? 2 : 1) : 0;  // first MFMA of every accumulator of this column half: C = 0 (or c_i: KT_RBFF)
                    constexpr int CUR = mm & 1;
                    if constexpr (mm == 2) {
                        if constexpr (SYM) {
                            if (kc == 0 && t > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        }
                        handover(step, (kc + 3) % NKC, t + (kc + 3) / NKC, checked);
                        if constexpr (SYM) {
                            if (kc == 0 && t > 0) flush_cols(t - 1);
                        }
                    }
                    // where the NEXT group's fragments come from: this chunk (mm < 3) or the first group of the next chunk (mm == 3)
                    constexpr int NKK = (mm + 1) >> 1, NH = (mm + 1) & 1;
                    const unsigned paddr = mm < 3 ? rdl[NKK & 1] + slot_off : rdl[0] + slot_next_off;
                    constexpr int PO = mm < 3 ? 4 * NH * 2048 : 0;
                    f32x4 &c0 = acc[0][4 * cbh + 0], &c1 = acc[1][4 * cbh + 0], &c2 = acc[0][4 * cbh + 1], &c3 = acc[1][4 * cbh + 1];
                    f32x4 &c4 = acc[0][4 * cbh + 2], &c5 = acc[1][4 * cbh + 2], &c6 = acc[0][4 * cbh + 3], &c7 = acc[1][4 * cbh + 3];
                    // row planes 0 .. NQ - 1 of this k32 step (the dispatcher ignores the operands beyond 2 NQ)
                    constexpr int P0 = row_plane_of(kc, 0), P1 = NQ >= 2 ? row_plane_of(kc, 1) : P0, P2 = NQ >= 3 ? row_plane_of(kc, 2) : P0;
                    const bf16x8 &a00 = afrag[P0][2 * chunk + kk][0], &a01 = afrag[P0][2 * chunk + kk][1];
                    const bf16x8 &a10 = afrag[P1][2 * chunk + kk][0], &a11 = afrag[P1][2 * chunk + kk][1];
                    const bf16x8 &a20 = afrag[P2][2 * chunk + kk][0], &a21 = afrag[P2][2 * chunk + kk][1];
                    // ONE form of every group, also for the very last one of the work item (whose prefetch then reads a stale but valid slot and is
                    // never used): a branch between a prefetching and a non-prefetching variant made the compiler merge the accumulators of the two
                    // arms with v_mov copies two instructions behind the MFMAs that write them -- inside an asm statement it inserts none of the
                    // wait states an XDL write needs before a VALU read, so the copies read registers still in flight (NaNs in every work item of two
                    // or more tiles: f16x3 rbf at 128 features; most likely also round 2's NaN of the one-wave bf16x6 form).
                    // tests/tools/audit_hand_asm.py checks the generated code for such copies.
                    s6_group<F16, NQ, CUR, 1, Z, PO, PO + 2048, PO + 4096, PO + 6144>(c0, c1, c2, c3, c4, c5, c6, c7, a00, a01, a10, a11, a20, a21, civ0[0], civ0[1], paddr);
                    // the LDS-DMA of chunk step + 3 goes between the groups of the step's second half (two instructions behind each)
                    if constexpr (!decltype(checked)::value && mm >= 2) {
                        issue_part_static(std::integral_constant<int, kc + 3>{}, (phase + kc + 3) & (V2_RING - 1), std::integral_constant<int, (mm - 2) * 2 + 0>{});
                        issue_part_static(std::integral_constant<int, kc + 3>{}, (phase + kc + 3) & (V2_RING - 1), std::integral_constant<int, (mm - 2) * 2 + 1>{});
                    }
                    if constexpr (kc == NKC - 1 && mm == 3) {
                        // the epilogue's vector ALU instructions read what the last MFMAs write: the compiler cannot see inside the groups,
                        // so the wait states an XDL write needs before a VALU read (8 passes: 11) are spent here, once per tile.  The
                        // accumulators are in/out operands of the statement: a "memory" clobber alone does not keep register arithmetic
                        // behind it, and the scheduler did hoist an epilogue v_pk_fma_f32 to two instructions behind the last MFMA
                        // (bf16x6 polynomial at 64 features; tests/tools/audit_hand_asm.py reports such accesses)
                        asm volatile("s_nop 15\n\ts_nop 3"
                                     : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[0][4]), "+v"(acc[0][5]), "+v"(acc[0][6]), "+v"(acc[0][7]),
                                       "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]), "+v"(acc[1][4]), "+v"(acc[1][5]), "+v"(acc[1][6]), "+v"(acc[1][7])
                                     :
                                     : "memory");
                    }
                } else {
                f32x4(&bcur)[4] = bbuf[mm & 1];

                f32x4(&bnext)[4] = bbuf[(mm + 1) & 1];
                if constexpr (mm < 3) {  // next group of this chunk: (kk', cbh') = ((mm + 1) >> 1, (mm + 1) & 1)
#pragma unroll
                    for (int c = 0; c < 4; ++c) bnext[c] = *reinterpret_cast<const f32x4 *>(slot + (4 * ((mm + 1) & 1) + c) * 2048 + rd_off[(mm + 1) >> 1]);
                    if constexpr (mm != 2) LSSVM_SCHED_BARRIER();
                }
                if constexpr (mm == 2) {
                    if constexpr (SYM) {
                        if (kc == 0 && t > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    handover(step, (kc + 3) % NKC, t + (kc + 3) / NKC, checked);
                    if constexpr (SYM) {
                        if (kc == 0 && t > 0) flush_cols(t - 1);
                    }
                    LSSVM_SCHED_BARRIER();
                }
                if constexpr (mm == 3) {  // first group of the next chunk: visible since this step's hand-over
                    if (more) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) bnext[c] = *reinterpret_cast<const f32x4 *>(slot_next + c * 2048 + rd_off[0]);
                    }
                    LSSVM_SCHED_BARRIER();
                }
#pragma unroll
                for (int q = 0; q < (GRID ? 2 : PL); ++q) {
                    if (GRID ? q >= nq_of(kc) : q + plane > PL - 1) continue;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int cb = 4 * cbh + c;
                        const bf16x8 bv = __builtin_bit_cast(bf16x8, bcur[c]);
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb) {
                            const bf16x8 av = afrag[row_plane_of(kc, q)][2 * chunk + kk][rb];
                            if (KT == KT_RBFF && kc == 0 && kk == 0 && q == 0) {
                                acc[rb][cb] = plane_mfma<F16>(av, bv, civ0[rb]);
                            } else if (KT != KT_RBF && !GRID && kc == 0 && kk == 0 && q == 0) {
                                const f32x4 zero = { 0.f, 0.f, 0.f, 0.f };
                                acc[rb][cb] = plane_mfma<F16>(av, bv, zero);
                            } else {
                                acc[rb][cb] = plane_mfma<F16>(av, bv, acc[rb][cb]);
                            }
                        }
                        if constexpr (!decltype(checked)::value && mm >= 2) {
                            if (q == 0 && c == 0) issue_part_static(std::integral_constant<int, kc + 3>{}, (phase + kc + 3) & (V2_RING - 1), std::integral_constant<int, (mm - 2) * 2 + 0>{});
                            if (q == 0 && c == 2) issue_part_static(std::integral_constant<int, kc + 3>{}, (phase + kc + 3) & (V2_RING - 1), std::integral_constant<int, (mm - 2) * 2 + 1>{});
                        }
                    }
                }
                }  // !HAND
            });
        });
        xc_tile += tile_bytes;
        if (!LSSVM_DBG(a, 4)) {
            auto epilogue = [&](auto with_cols) {
                constexpr bool COLS = decltype(with_cols)::value;
                f32x4 di[2];
                using f32x2 = float __attribute__((ext_vector_type(2)));
                float colacc[8];
                f32x2 colacc2[8] = {};  // even / odd rows of the lane's four, added at the end
                f32x2 kvp = { 0.f, 0.f };
                if constexpr (COLS) {
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) di[rb] = *reinterpret_cast<const f32x4 *>(dis + wave * 32 + 16 * rb + 4 * g);
                }
                const float *dcr = reinterpret_cast<const float *>(dcs + (t % V2_DC_SLOTS) * 1024);  // the record stays valid until tile t + 4 is announced
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) {
                    const float djv = dcr[cb * 16 + r];
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float kv = LSSVM_DBG(a, 2) ? acc[rb][cb][e] : apply_kernel_function<v2_base_kt(KT), v2_degree_class(KT)>(GRID ? acc[rb][cb][e] * a.gamma : acc[rb][cb][e], a);  // bit 2: no exp; grid planes: the chain carries sigma^2, gamma = sigma^-2
                            if constexpr (KT == KT_POLY) {
                                if (padcol[cb]) kv = 0.0f;
                            }
                            if (LSSVM_DBG(a, 256)) {  // bit 256: no fmas (one add keeps the value alive)
                                rowpart[4 * rb + e] += kv;
                                continue;
                            }
                            rowpart[4 * rb + e] = fmaf(kv, djv, rowpart[4 * rb + e]);
                            // column sums: rows (e, e + 1) of a block as ONE v_pk_fma_f32 (the row sums above are packed by the compiler itself;
                            // these it leaves scalar -- eight dependent chains of eight -- unless the pairs are spelled out)
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
                }
                if constexpr (COLS) {
                    // the four lane groups hold different rows of the same column: two butterfly steps on the vector ALU (no LDS round trips) for
                    // all eight blocks at once (column_sums_of_8_blocks: 6 swaps + 6 adds); the sums come out one column per lane -- block q in
                    // lane group q of colacc[0], block 4 + q in colacc[4] -- so the record's factor and the store are two instructions of the
                    // whole wave and the epilogue is a single basic block without a branch
                    float *cw = colred + (t & 1) * 512 + wave * 128;
                    column_sums_of_8_blocks(colacc);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float v = colacc[4 * h];
                        if constexpr (KT == KT_RBFF) v *= dcr[128 + 64 * h + lane];  // K_ij = 2^acc * 2^c_j: the column's factor once per column
                        if constexpr (KT == KT_LINEAR && F16) v *= a.out_scale;  // planes pre-scaled by 2^k: undo 2^(2k) (exact)
                        cw[64 * h + lane] = v;
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

    constexpr int TAIL_TILES = (3 + NKC - 1) / NKC;
    const int nmain = ntiles > TAIL_TILES ? ntiles - TAIL_TILES : 0;
    int t = 0;
    for (; t < nmain; ++t) tile_body(t, std::false_type{});
    for (; t < ntiles; ++t) tile_body(t, std::true_type{});
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
        if constexpr (KT == KT_LINEAR && F16) v *= a.out_scale;
        if constexpr (GRID) v *= a.er[row0 + wave * 32 + 16 * (i >> 2) + 4 * g + (i & 3)];  // the row's folded factor E_i, once per work item
        rowpart[i] = v;
    }
    if (r == 0) {
        float *dst = a.partial + static_cast<size_t>(jc) * a.part_stride + ibl * TILE + wave * 32 + 4 * g;
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[16 * (i >> 2) + (i & 3)] = rowpart[i];
    }
}

/* The hand-scheduled kernels keep their B fragments in v[224:255], registers the compiler must never allocate.  The cap is the function attribute
 * "amdgpu-num-vgpr" -- which on gfx950 (unified 512-entry VGPR/AGPR file) counts HALF registers: the backend doubles the requested number
 * (GCNSubtarget::getBaseMaxNumVGPRs) before it reserves everything above it, so amdgpu_num_vgpr(224) reserves nothing below v448 and is silently
 * ineffective -- round 2 shipped it that way and was saved by kernels that happened to need fewer than 224 registers; the f16x3 rbf form needed 229,
 * the compiler put epilogue temporaries into v224 ... v228, and every work item of two or more tiles returned NaNs.  amdgpu_num_vgpr(112) is the
 * cap that holds (measured: highest compiler-allocated register v223, spills instead of trespassing); tests/tools/audit_hand_asm.py checks every
 * hand-scheduled instantiation for compiler code that touches v224 and above, and for scratch traffic. */
#define LSSVM_HAND_VGPR_CAP __attribute__((amdgpu_num_vgpr(112)))

/* the two kernels around s6w_body: compiler-scheduled groups (any supported feature count), and hand-scheduled groups for the
 * two-waves-per-SIMD instantiations, where the compiler is confined to v0..v223 so that v[224:255] can hold the B fragments */
template <int KT, int NK64, bool SYM>
__global__ __launch_bounds__(TILE_THREADS, (NK64 <= 2 ? 2 : 1)) void tile_matvec_f32_s6w(const TileArgs<float> a) {
    s6w_body<KT, NK64, SYM, false, 3>(a);
}
template <int KT, int NK64, bool SYM>
__global__ __launch_bounds__(TILE_THREADS, 2) LSSVM_HAND_VGPR_CAP void tile_matvec_f32_s6h(const TileArgs<float> a) {
    static_assert(NK64 <= 2, "the hand-scheduled groups assume the 256-register budget of two waves per SIMD");
    s6w_body<KT, NK64, SYM, true, 3>(a);
}
/* "f16x3": the same kernels on TWO f16 planes (x = hi + mid, 11 + 11 significant bits, k_split_f16x2) and the three plane products
 * hi*hi + hi*mid + mid*hi on v_mfma_f32_16x16x32_f16 -- half the matrix-core work of bf16x6 and two thirds of its column stream.  What is
 * dropped (mid*mid and the split remainder) is below 2^-23 |x||y| per product while the planes stay in f16's normal range, which the set-up
 * checks row by row on the data itself (Problem<T>: `f16_planes_ok`); data that fails the check runs as bf16x6.  Row panel = 2 planes: 64
 * registers at 128 features, 256 at 512 features. */
constexpr int f16_max_nk64(int kt) { return (kt == KT_RBF || kt == KT_RBFF) ? 6 : 8; }  // rbf holds three row planes (see s6w_body)
template <int KT, int NK64, bool SYM>
__global__ __launch_bounds__(TILE_THREADS, (NK64 <= 2 ? 2 : 1)) void tile_matvec_f32_f3w(const TileArgs<float> a) {
    static_assert(NK64 <= f16_max_nk64(KT), "row panel does not fit the register file");
    s6w_body<KT, NK64, SYM, false, 2>(a);
}
constexpr int F16_HAND_MAX_NK64 = 2;
/* rbf on grid planes (KT_RBFG, round 5): f16 planes (h | s1 | s2), six plane products in four phases; hand-scheduled groups up to 128 features (g6h), the
 * compiler-scheduled ones up to 384 (g6w: one workgroup per CU beyond 128, like the f16x3 rbf kernels -- three row planes in registers) */
template <int NK64, bool SYM>
__global__ __launch_bounds__(TILE_THREADS, (NK64 <= 2 ? 2 : 1)) void tile_matvec_f32_g6w(const TileArgs<float> a) {
    static_assert(NK64 <= 6, "row panel does not fit the register file");
    s6w_body<KT_RBFG, NK64, SYM, false, 2>(a);
}
template <int NK64, bool SYM>
__global__ __launch_bounds__(TILE_THREADS, 2) LSSVM_HAND_VGPR_CAP void tile_matvec_f32_g6h(const TileArgs<float> a) {
    static_assert(NK64 <= F16_HAND_MAX_NK64, "the hand-scheduled groups assume the 256-register budget of two waves per SIMD");
    s6w_body<KT_RBFG, NK64, SYM, true, 2>(a);
}
template <int KT, int NK64, bool SYM>
__global__ __launch_bounds__(TILE_THREADS, 2) LSSVM_HAND_VGPR_CAP void tile_matvec_f32_f3h(const TileArgs<float> a) {
    static_assert(NK64 <= F16_HAND_MAX_NK64, "the hand-scheduled groups assume the 256-register budget of two waves per SIMD");
    s6w_body<KT, NK64, SYM, true, 2>(a);
}

}  // namespace lssvm
