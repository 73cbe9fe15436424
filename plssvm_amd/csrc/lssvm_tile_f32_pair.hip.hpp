/*
 * lssvm_tile_f32_pair.hip.hpp -- the split tile kernels (f16x3 / bf16x6, lssvm_tile_f32_split.hip.hpp) with 256-ROW WORKGROUPS: eight waves, two
 * per SIMD, share ONE column stream (round 4; symmetric variant, num_features <= 128, hand-scheduled MFMA groups).
 *
 * Why.  In the 128-row kernels every wave issues 4 LDS-DMA instructions per plane-chunk (16 per 128 x 128 tile at 128 features) and a CU's two
 * independent workgroups each pull their own copy of the column stream through L2 -> LDS.  Round 3's ablations priced the LDS-DMA at ~10 % of the
 * kernel and showed that nothing beside the MFMAs hides on the 16x16x32 shape, so what is left is fewer instructions and fewer bytes per MFMA:
 *   * a work item is a PAIR of row blocks (2p, 2p + 1) = 256 rows x a chunk of column tiles; wave w owns rows 32 w .. 32 w + 31 exactly as
 *     before (same row panel in registers, same accumulators, same MFMA groups: lssvm_s6w_groups.inc) -- per MFMA the LDS-DMA instructions,
 *     the L2 -> LDS bytes, the barriers and the column-sum records all halve;
 *   * the two waves of a SIMD now belong to ONE workgroup and run in LOCK STEP (one barrier per plane-chunk step for all eight waves), the
 *     second-dispatched half (waves 4-7) at s_setprio 1.  The stagger of MI355X_MICROARCH.md, "Two waves per SIMD", item 9 -- waves 4-7 LAG
 *     plane-chunk steps behind waves 0-3 so that one half's epilogue meets the other half's MFMA step; the ring of EIGHT 16 KiB slots exists for
 *     it -- was built and measured SLOWER at every lag (template parameter LAGT, development builds only): the 16-bit
 *     MFMA stream runs at the board's power cap, where cycles left idle come back as clock and instructions moved under the MFMAs still cost
 *     their energy;
 *   * the diagonal: column tile J against the pair's blocks b = 2p (waves 0-3) and b = 2p + 1 (waves 4-7): J < b row and mirrored column
 *     sums, J == b row sums only (the diagonal tile is evaluated in full), J > b nothing (tile 2p + 1 for the first half: one 128 x 128
 *     sub-tile of idle MFMAs per row pair, 1 / n_tiles of the work).  One record of 128 column sums per (pair, J): record (2p + 1, J) of the
 *     packed triangle, summed over the contributing waves in a fixed order; k_reduce_colslab walks the odd row blocks only.
 * Everything else -- LDS image and swizzle, LDS-DMA with SGPR bases, counted waits, hand-over in mid-step, folded rbf records, shifted
 * rbf planes, epilogue, butterflies -- is s6w_body's, statement by statement.  Shard and band boundaries are even block indices
 * (sym_block_boundary, band_edges), so a pair never straddles two devices or two bands; an odd number of row blocks ends in a pair whose
 * second block is zero padding (its row sums are never read, its column sums are exact zeros).
 * Reference shape this replaces: include/plssvm/backends/HIP/svm_kernel.hip.hpp:208-270 (16 x 16 threads, 6 x 6 register tile, atomicAdd).
 */
#pragma once

#include "lssvm_tile_f32_split.hip.hpp"

namespace lssvm {

constexpr int PR_WAVES = 8;
constexpr int PR_THREADS = 64 * PR_WAVES;
constexpr int PR_ROWS = 2 * TILE;
constexpr int PR_RING = 8;  // 16 KiB slots (power of two)
constexpr size_t PR_LDS_BYTES = static_cast<size_t>(PR_RING) * V2_SLOT_BYTES + V2_DC_SLOTS * 1024 + (2 * PR_ROWS + 2 * PR_WAVES * TILE) * sizeof(float);  // ring + records + cis, dis, colred

#ifdef LSSVM_ITEM_TRACE  // measurement builds only (tests/tools/item_trace.py): wave 0 of every work item stamps its phases with the constant 100 MHz clock
__device__ unsigned long long *lssvm_item_trace = nullptr;  // [num_items][8]: entry, row panel loaded, tile loop entered, tile loop left, end, HW_ID, tiles, (spare)
#define LSSVM_TRACE(slot) do { if (HALF == 0 && lssvm_item_trace != nullptr && threadIdx.x == 0) lssvm_item_trace[static_cast<size_t>(item_pos) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LSSVM_TRACE(slot) do { } while (0)
#endif

/* HALF: 0 = waves 0-3 (row block 2p), 1 = waves 4-7 (row block 2p + 1; LAGT = 0, the shipped form: in lock step with the first half).  Both halves
 * execute the same number of barriers.
 * RECT (round 6): the RECTANGULAR product of predict_values (rows = the points to predict, columns = the support vectors; reference shape
 * include/plssvm/backends/HIP/predict_kernel.hip.hpp:63-117) -- the same eight waves on one column stream, the same MFMA groups and LDS image, but every tile of the
 * item's column chunk is evaluated in full (no diagonal), and there are no mirrored column sums: no d_i, no column butterflies, no records to flush; the epilogue is
 * the kernel function and one fma per element. */
template <int KT, int NK64, int PL, int HALF, int LAGT, bool RECT = false>
__device__ __forceinline__ void pair_body(const TileArgs<float> &a, const int item_pos) {
    static_assert(PL == 3 || PL == 2, "three bf16 planes (bf16x6) or two f16 planes (f16x3)");
    static_assert(NK64 <= 2, "the hand-scheduled groups assume the 256-register budget of two waves per SIMD");
    static_assert(KT != KT_RBF, "rbf runs here with BOTH exponent terms folded (KT_RBFF, see below) or on grid planes (KT_RBFG); the unfolded form stays on the 128-row kernels");
    static_assert(LAGT >= 0 && LAGT <= 7, "0 ... 3: steps of lag; 4 ... 7: priority experiments of the development builds");
    constexpr bool F16 = PL == 2;
    // KT_RBFG (round 6: rbf with a large exponent scale in the 256-row form; s6w_body has the 128-row form and the derivation): three f16 column planes (h | s1 | s2) and FOUR
    // phases per tile, each over all 64-feature chunks -- h x h, h x (s1, s2), s1 x (h, s1), s2 x h -- with the accumulators started from sigma^2 (ch_i + ch_j), an exact sum
    constexpr bool GRID = KT == KT_RBFG;
    static_assert(!GRID || (PL == 2 && !RECT), "the grid-plane kernel exists with f16 planes, symmetric variant");
    constexpr int NKC = GRID ? 4 * NK64 : PL * NK64;
    constexpr auto col_plane_of = [](int kc) constexpr { return GRID ? (kc / NK64 == 0 ? 0 : kc / NK64 - 1) : kc % PL; };
    constexpr auto chunk_of = [](int kc) constexpr { return GRID ? kc % NK64 : kc / PL; };
    constexpr auto nq_of = [](int kc) constexpr { return GRID ? ((kc / NK64 == 1 || kc / NK64 == 2) ? 2 : 1) : PL - kc % PL; };
    // LAGT = 0, the shipped form: lock step, and the second-dispatched half of the workgroup (waves 4-7, the loser of the SIMD's issue arbitration by age) at
    // s_setprio 1 for the whole kernel (MI355X_MICROARCH.md, Two waves per SIMD, item 4): 265.0 -> 262.1 ms at 1 000 000 x 128 rbf, neutral for the linear
    // kernel (profiles/r04_ab_pair_priority.log).  Development builds also carry 1, 3 = steps of lag (no priority); 4 = lock step WITHOUT the priority;
    // 5 = priority 3; 6 = one step of lag + priority 1; 7 = priority 1 for waves 0-3 instead -- all measured slower than 0.
    constexpr int PRIO = (LAGT == 0 || LAGT == 6) ? (HALF == 1 ? 1 : 0) : (LAGT == 5 ? (HALF == 1 ? 3 : 0) : (LAGT == 7 ? (HALF == 0 ? 1 : 0) : 0));
    constexpr int LAGE = LAGT == 6 ? 1 : (LAGT >= 4 ? 0 : LAGT);
    constexpr int LAG = HALF ? LAGE : 0;  // this half's distance behind the global step counter
    constexpr int PLA = F16 ? ((KT == KT_RBF || KT == KT_RBFF || GRID) ? 3 : 2) : 3;
    constexpr auto row_plane = [](int p, int q) constexpr { return (F16 && PLA == 3) ? (p == 0 ? (q == 0 ? 2 : 1) : 0) : q; };
    constexpr auto row_plane_of = [row_plane](int kc, int q) constexpr {  // row plane of the q-th product of step kc (grid planes: by phase)
        if (GRID) {
            const int ph = kc / NK64;
            return ph == 0 ? 0 : (ph == 1 ? (q == 0 ? 1 : 2) : (ph == 2 ? (q == 0 ? 0 : 1) : 0));
        }
        return row_plane(kc % PL, q);
    };
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char *ring = smem_raw;                                               // [PR_RING][128 columns][128 B]
    char *dcs = smem_raw + PR_RING * V2_SLOT_BYTES;                      // [V2_DC_SLOTS][256 floats]
    float *cis = reinterpret_cast<float *>(dcs + V2_DC_SLOTS * 1024);    // [256] c_i of the row pair (rbf)
    float *dis = cis + PR_ROWS;                                          // [256] d_i of the row pair
    float *colred = dis + PR_ROWS;                                       // [2][8 waves][128] column sums of a tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0 .. 7
    const int r = lane & 15;
    const int g = lane >> 4;

    if constexpr (PRIO == 1) asm volatile("s_setprio 1");
    if constexpr (PRIO == 3) asm volatile("s_setprio 3");
    LSSVM_TRACE(0);
    const int2 it = a.items[item_pos];
    const int ibl = __builtin_amdgcn_readfirstlane(it.x);  // local index of the pair's FIRST block (even)
    const int jc = __builtin_amdgcn_readfirstlane(it.y);
    const int ib0 = a.ib_begin + ibl;
    const int my_ib = ib0 + HALF;
    const int row0 = ib0 * TILE;
    const int jt_begin = chunk_begin(jc, a.jc_tiles, a.jc_head_tiles, a.jc_head_count);
    const int jt_end = RECT ? min(jt_begin + chunk_len(jc, a.jc_tiles, a.jc_head_tiles, a.jc_head_count), a.num_jt)
                            : min(min(jt_begin + chunk_len(jc, a.jc_tiles, a.jc_head_tiles, a.jc_head_count), ib0 + 2), a.num_jt);
    const int ntiles = jt_end - jt_begin;
    if (ntiles <= 0) return;
    const int nsteps = ntiles * NKC;
    const long rec0 = static_cast<long>(ib0 + 1) * ib0 / 2 - a.pair_origin;  // records of row block ib0 + 1

    // ---- LDS-DMA addressing: a slot is 16 pieces of 8 columns x 128 B; wave w moves pieces 2 w and 2 w + 1 ----
    unsigned dma_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 8 * (2 * wave + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        dma_off[i] = 2u * static_cast<unsigned>(row * a.ldx16 + 8 * c);
    }
    const unsigned ring_lds = static_cast<unsigned>(reinterpret_cast<size_t>(ring));
    const unsigned dma_lds = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(ring_lds + static_cast<unsigned>(wave) * 2048u)));
    const unsigned dc_lds = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(ring_lds + PR_RING * V2_SLOT_BYTES + static_cast<unsigned>(wave & 3) * 256u)));
    auto issue_chunk = [&](int step) {  // generic form (prologue, last tiles)
        const int t = step / NKC;
        const int kc = step - t * NKC;
        const char *base = sgpr_ptr(a.Xc16 + col_plane_of(kc) * a.plane_stride + static_cast<size_t>(jt_begin + t) * TILE * a.ldx16 + chunk_of(kc) * 64);
        const unsigned slot = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(dma_lds + static_cast<unsigned>(step & (PR_RING - 1)) * V2_SLOT_BYTES)));
        static_for<0, 2>([&](auto i_c) { lds_dma16<decltype(i_c)::value * 1024>(dma_off[decltype(i_c)::value], base, slot); });
    };
    const size_t tile_bytes = static_cast<size_t>(TILE) * a.ldx16 * 2;
    const size_t plane_bytes = a.plane_stride * 2;
    const char *xc_tile = reinterpret_cast<const char *>(a.Xc16) + static_cast<size_t>(jt_begin) * tile_bytes;
    auto issue_part_static = [&](auto kc3_c, unsigned slot_idx, auto i_c) {  // kc3 = kc + 3 + LAG of the issuing (own) step
        constexpr int KC3 = decltype(kc3_c)::value;
        constexpr int KC = KC3 % NKC;
        constexpr int i = decltype(i_c)::value;
        if (LSSVM_DBG(a, 16)) return;  // ablation: no LDS-DMA after the prologue
        const char *base = xc_tile + (KC3 / NKC) * tile_bytes + col_plane_of(KC) * plane_bytes + chunk_of(KC) * 128;
        lds_dma16<i * 1024>(dma_off[i], sgpr_ptr(base), dma_lds + slot_idx * V2_SLOT_BYTES);
    };
    auto issue_dc = [&](int t) {  // the record of column tile t: by the second half (four waves x 256 B)
        if constexpr (HALF == 1) {
            if (lane < 16) {
                const char *src = sgpr_ptr(a.dc + static_cast<size_t>(jt_begin + t) * 256) + __builtin_amdgcn_readfirstlane((wave & 3) * 256);
                lds_dma16<0>(16u * (lane_off(threadIdx.x) & 15u), sgpr_ptr(src), static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(dc_lds + static_cast<unsigned>(t % V2_DC_SLOTS) * 1024u))));
            }
        }
    };

    int rd_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) rd_off[kk] = r * 128 + (((4 * kk + g) ^ ((r >> 1) & 7)) << 4);

    float rowpart[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rowpart[i] = 0.0f;
    f32x4 acc[2][8];
    const f32x4 civ0[2] = { { 0.f, 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f, 0.f } };  // (operands of the group dispatcher that no group of this kernel reads)

    // hand-over in the middle of GLOBAL step gs (runtime form): the chunk of step gs + 1 becomes visible, the DMA of step gs + 3 starts.  The waves
    // have at most the two pieces of step gs + 2 younger than what they wait for.  EVERY wave passes exactly one barrier per global step.
    auto handover_checked = [&](int gs) {
        if (gs + 2 < nsteps) {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (gs + 3 < nsteps) {
            if ((gs + 3) % NKC == 0) issue_dc((gs + 3) / NKC);
            issue_chunk(gs + 3);
        }
    };

    // ---- prologue: the LDS-DMA of chunks 0, 1, 2 goes out FIRST, the row panel (ordinary loads) behind it -- one memory latency per work item instead
    // of two; with one workgroup per CU nothing else covers a work item's start.  One wait for all of it: the row panel is needed at once anyway. ----
    issue_dc(0);
    issue_chunk(0);
#pragma unroll
    for (int pre = 1; pre <= 2; ++pre) {
        if (pre < nsteps) {
            if (pre % NKC == 0) issue_dc(pre / NKC);
            issue_chunk(pre);
        }
    }
    // ---- the row panel (this wave's 32 rows, all features, all row planes) ----
    bf16x8 afrag[PLA][2 * NK64][2];
    // rbf on f16 planes: row plane 0 (2^-6 hi) is row plane 2 (2^6 hi) times 2^-12 -- one exact f16 multiplication (the conversion that made plane 0 rounds the
    // same product the same way) -- so two planes are loaded and the third is derived in registers: a third less of the row panel, whose load is most of a work
    // item's start (192 -> 128 KiB per item at ~11 B / cycle / CU).  Bit-identical; 1 000 000 x 128: 267.8 -> 266.7 ms, 50 000 x 128: 0.759 -> 0.755 ms per
    // iteration, same box, interleaved (profiles/r05_ab_mfma_order_and_derived_row_plane.log, "lib_v_derive")
    constexpr int P_FIRST = (F16 && PLA == 3 && !GRID) ? 1 : 0;  // (grid planes: h, s1, s2 are three planes of their own)
    const size_t frag_chunk = a.plane_stride_r / static_cast<size_t>(a.ldx16) / 16 * 1024;  // elements between the 64-feature chunks of a fragment-major plane
#pragma unroll
    for (int p = P_FIRST; p < PLA; ++p) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            // from the FRAGMENT-MAJOR copy of the planes (TileArgs::Xr16f, k_planes_fragment_major: [plane][64-feature chunk][16-row block][2 k-steps][64 lanes][8]):
            // every load instruction of a wave reads 1 KiB in one piece -- row-major it read 64 bytes of each of 16 rows, and the row panel, which nothing
            // overlaps with (one workgroup per CU), took 4.5 us of a work item (tests/tools/item_trace.py)
#ifndef LSSVM_PAIR_ROW_MAJOR_PANEL
            const uint16_t *xr = a.Xr16f + p * a.plane_stride_r + static_cast<size_t>((row0 >> 4) + wave * 2 + rb) * 1024 + lane * 8;
#pragma unroll
            for (int kk = 0; kk < 2 * NK64; ++kk) afrag[p][kk][rb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(xr + (kk >> 1) * frag_chunk + (kk & 1) * 512));
#else  // (A/B builds: the former loads from the row-major planes)
            const uint16_t *xr = a.Xr16 + p * a.plane_stride_r + static_cast<size_t>(row0 + wave * 32 + 16 * rb + r) * a.ldx16 + 8 * g;
#pragma unroll
            for (int kk = 0; kk < 2 * NK64; ++kk) afrag[p][kk][rb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(xr + 32 * kk));
#endif
        }
    }
    // rbf: K_ij = 2^c_i 2^(x_i . x_j) 2^c_j with BOTH exponent terms folded out of the chain -- the column's as the record's factor e_j (k_pack_dc:
    // (e_j d_j | e_j), as in s6w_body), the row's as e_i = 2^c_i: the mirrored column sums take e_i d_i in place of d_i, the finished row sums are
    // multiplied by e_i once per work item, and the accumulators start from the constant 0 like every other kernel's.  (s6w_body starts them from
    // c_i, eight registers that live across the whole tile loop: with the row panel of three planes, the accumulators and the private B fragments
    // this kernel has none to spare, and a spilled row-panel fragment is re-loaded in front of every MFMA group.)  The host chooses this kernel
    // only while |c| <= PAIR_FOLD_MAX_C keeps e and the partial sums inside the fp32 range.
    if constexpr (HALF == 0) {
        if constexpr (KT == KT_RBFF) cis[tid] = __builtin_amdgcn_exp2f(a.cr[row0 + tid]);
        if constexpr (GRID) cis[tid] = a.cr[row0 + tid];  // sigma^2 ch_i: the start value
    } else if constexpr (!RECT) {
        const float dv = a.dvec[row0 + tid - PR_ROWS];
        if constexpr (KT == KT_RBFF) {
            dis[tid - PR_ROWS] = dv * __builtin_amdgcn_exp2f(a.cr[row0 + tid - PR_ROWS]);
        } else if constexpr (GRID) {
            dis[tid - PR_ROWS] = dv * a.er[row0 + tid - PR_ROWS];  // the row's folded factor E_i rides on d_i
        } else {
            dis[tid - PR_ROWS] = dv;
        }
    }
    // (retire the ordinary loads HERE: none may be outstanding once the counted waits of the hand-overs begin)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    LSSVM_TRACE(1);
    if constexpr (P_FIRST == 1) {
        typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int kk = 0; kk < 2 * NK64; ++kk)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) afrag[0][kk][rb] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(f16x8_t, afrag[PLA - 1][kk][rb]) * static_cast<_Float16>(0x1p-12f));
    }
#pragma unroll
    for (int p = 0; p < PLA; ++p)
#pragma unroll
        for (int kk = 0; kk < 2 * NK64; ++kk)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) asm volatile("" : "+v"(afrag[p][kk][rb]));

    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // the lagging half lets LAG global steps pass (hand-overs only)
#pragma unroll
    for (int gs = 0; gs < LAG; ++gs) handover_checked(gs);

    const unsigned rdl[2] = { ring_lds + static_cast<unsigned>(rd_off[0]), ring_lds + static_cast<unsigned>(rd_off[1]) };
    s6w_fill_b0<0, 2048, 4096, 6144>(rdl[0]);

    auto flush_cols = [&](int t) {  // by waves 4 and 5: the eight waves' sums of a column in a fixed order -> record (ib0 + 1, J)
        if constexpr (HALF == 1) {
            if (tid < PR_ROWS + TILE) {
                const int col = tid - PR_ROWS;
                const float *cr_ = colred + (t & 1) * (PR_WAVES * TILE);
                const float s03 = (cr_[col] + cr_[128 + col]) + (cr_[256 + col] + cr_[384 + col]);
                const float s47 = (cr_[512 + col] + cr_[640 + col]) + (cr_[768 + col] + cr_[896 + col]);
                auto *rec = (__attribute__((address_space(1))) float *) const_cast<char *>(sgpr_ptr(a.colslab + (rec0 + jt_begin + t) * TILE));
                rec[lane_off(static_cast<unsigned>(col))] = s03 + s47;
            }
        }
    };

    auto tile_body = [&](int t, auto checked) {
        const int s0 = t * NKC;
        const int J = jt_begin + t;
        const unsigned phase = static_cast<unsigned>(s0) & (PR_RING - 1);
        if constexpr (GRID) {  // the accumulators start at sigma^2 (ch_i + ch_j), an exact sum (s6w_body)
            const float *dcr0 = reinterpret_cast<const float *>(dcs + (t % V2_DC_SLOTS) * 1024);
            f32x4 civ[2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) civ[rb] = *reinterpret_cast<const f32x4 *>(cis + wave * 32 + 16 * rb + 4 * g);
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const float cjv = dcr0[128 + cb * 16 + r];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[rb][cb][e] = civ[rb][e] + cjv;
            }
        }
        static_for<0, NKC>([&](auto kc_c) {
            constexpr int kc = decltype(kc_c)::value;
            constexpr int chunk = chunk_of(kc);
            const int step = s0 + kc;
            const unsigned slot_off = ((phase + kc) & (PR_RING - 1)) * V2_SLOT_BYTES;
            const unsigned slot_next_off = ((phase + kc + 1) & (PR_RING - 1)) * V2_SLOT_BYTES;
            static_for<0, 4>([&](auto mm_c) {
                constexpr int mm = decltype(mm_c)::value;
                constexpr int kk = mm >> 1, cbh = mm & 1;
                constexpr int NQ = nq_of(kc);
                constexpr int Z = (!GRID && kc == 0 && kk == 0) ? 1 : 0;  // first MFMA of every accumulator of this column half: C = 0 (grid planes: the start values above)
                constexpr int CUR = mm & 1;
                if constexpr (mm == 2) {
                    // BOTH halves retire their LDS traffic before the hand-over barrier that precedes flush_cols(t - 1): the colred stores of tile t - 1 (all eight waves)
                    // must have landed when waves 4 and 5 read them.  (For waves 0-3 the wait is already implied by the lgkmcnt(0) at the head of the two groups in
                    // front of it; spelled out so that a re-schedule of the groups cannot turn it into a race -- ADVICE r04.)
                    if constexpr (!RECT) {
                        if (kc == 0 && t > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    if constexpr (!decltype(checked)::value) {
                        if (!LSSVM_DBG(a, 16)) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                        if (!LSSVM_DBG(a, 8)) __builtin_amdgcn_s_barrier();  // ablation bit 8: no hand-over barrier in the steady state
                        asm volatile("" ::: "memory");
                        if constexpr ((kc + 3 + LAG) % NKC == 0) issue_dc(t + (kc + 3 + LAG) / NKC);
                    } else {
                        handover_checked(step + LAG);
                    }
                    if constexpr (HALF == 1 && !RECT) {
                        if (kc == 0 && t > 0 && J - 1 < ib0 + 1) flush_cols(t - 1);
                    }
                }
                constexpr int NKK = (mm + 1) >> 1, NH = (mm + 1) & 1;
                const unsigned paddr = mm < 3 ? rdl[NKK & 1] + slot_off : rdl[0] + slot_next_off;
                constexpr int PO = mm < 3 ? 4 * NH * 2048 : 0;
                f32x4 &c0 = acc[0][4 * cbh + 0], &c1 = acc[1][4 * cbh + 0], &c2 = acc[0][4 * cbh + 1], &c3 = acc[1][4 * cbh + 1];
                f32x4 &c4 = acc[0][4 * cbh + 2], &c5 = acc[1][4 * cbh + 2], &c6 = acc[0][4 * cbh + 3], &c7 = acc[1][4 * cbh + 3];
                constexpr int P0 = row_plane_of(kc, 0), P1 = NQ >= 2 ? row_plane_of(kc, 1) : P0, P2 = NQ >= 3 ? row_plane_of(kc, 2) : P0;
                const bf16x8 &a00 = afrag[P0][2 * chunk + kk][0], &a01 = afrag[P0][2 * chunk + kk][1];
                const bf16x8 &a10 = afrag[P1][2 * chunk + kk][0], &a11 = afrag[P1][2 * chunk + kk][1];
                const bf16x8 &a20 = afrag[P2][2 * chunk + kk][0], &a21 = afrag[P2][2 * chunk + kk][1];
                s6_group<F16, NQ, CUR, 1, Z, PO, PO + 2048, PO + 4096, PO + 6144>(c0, c1, c2, c3, c4, c5, c6, c7, a00, a01, a10, a11, a20, a21, civ0[0], civ0[1], paddr);
                // the wave's two pieces of the chunk three global steps ahead: one behind each group of the step's second half
                if constexpr (!decltype(checked)::value && mm >= 2) {
                    issue_part_static(std::integral_constant<int, kc + 3 + LAG>{}, (phase + kc + 3 + LAG) & (PR_RING - 1), std::integral_constant<int, mm - 2>{});
                }
                if constexpr (kc == NKC - 1 && mm == 3) {
                    // wait states between the last MFMAs and the epilogue's vector instructions (see s6w_body)
                    asm volatile("s_nop 15\n\ts_nop 3"
                                 : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[0][4]), "+v"(acc[0][5]), "+v"(acc[0][6]), "+v"(acc[0][7]),
                                   "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]), "+v"(acc[1][4]), "+v"(acc[1][5]), "+v"(acc[1][6]), "+v"(acc[1][7])
                                 :
                                 : "memory");
                }
            });
        });
        xc_tile += tile_bytes;
        // ONE epilogue per tile loop, without a branch: the steady-state tiles lie strictly below the pair's first block (rows and mirrored
        // columns for every wave); the last tiles of a work item (MASKED) switch a wave's row sums off above its diagonal and its column sums off
        // on and above it with two factors in {0, 1} -- the wave then adds exact zeros (K is finite), and the first half's zeros for tile
        // (2p + 1, 2p), which the second half does flush, need no code of their own.  (With several epilogue variants behind branches in one
        // loop the register allocator spills row-panel fragments for the whole kernel, and a scratch reload drains the LDS-DMA queue.)
        constexpr bool MASKED = decltype(checked)::value && !RECT;
        if constexpr (RECT) {
            // the rectangular product: K_ij (alpha_j e_j) summed over the tile's columns, nothing mirrored
            const float *dcr = reinterpret_cast<const float *>(dcs + (t % V2_DC_SLOTS) * 1024);
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const float djv = dcr[cb * 16 + r];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float kv = apply_kernel_function<v2_base_kt(KT), v2_degree_class(KT)>(acc[rb][cb][e], a);
                        rowpart[4 * rb + e] = fmaf(kv, djv, rowpart[4 * rb + e]);
                    }
            }
        } else if (!LSSVM_DBG(a, 4)) {  // ablation bit 4: no epilogue
            f32x4 di[2];
            using f32x2 = float __attribute__((ext_vector_type(2)));
            float colacc[8];
            f32x2 colacc2[8] = {};
            f32x2 kvp = { 0.f, 0.f };
            const float rowmask = (MASKED && J > my_ib) ? 0.0f : 1.0f, colmask = (MASKED && J >= my_ib) ? 0.0f : 1.0f;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                di[rb] = *reinterpret_cast<const f32x4 *>(dis + wave * 32 + 16 * rb + 4 * g);
                if constexpr (MASKED) di[rb] *= colmask;
            }
            const float *dcr = reinterpret_cast<const float *>(dcs + (t % V2_DC_SLOTS) * 1024);
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                float djv = dcr[cb * 16 + r];
                if constexpr (MASKED) djv *= rowmask;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float kv = apply_kernel_function<v2_base_kt(KT), v2_degree_class(KT)>(GRID ? acc[rb][cb][e] * a.gamma : acc[rb][cb][e], a);  // (grid planes: the chain carries sigma^2, gamma = sigma^-2)
                        rowpart[4 * rb + e] = fmaf(kv, djv, rowpart[4 * rb + e]);
                        kvp[e & 1] = kv;
                        if (e & 1) {
                            const f32x2 dip = { di[rb][e - 1], di[rb][e] };
                            colacc2[cb] = __builtin_elementwise_fma(kvp, dip, colacc2[cb]);
                        }
                    }
            }
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) colacc[cb] = colacc2[cb][0] + colacc2[cb][1];
            float *cw = colred + (t & 1) * (PR_WAVES * TILE) + wave * TILE;
            column_sums_of_8_blocks(colacc);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v = colacc[4 * h];
                if constexpr (KT == KT_RBFF) v *= dcr[128 + 64 * h + lane];
                if constexpr (KT == KT_LINEAR && F16) v *= a.out_scale;
                cw[64 * h + lane] = v;
            }
        }
    };

    constexpr int TAIL_TILES = (3 + LAG + NKC - 1) / NKC;
    // steady-state tiles: the DMA three steps ahead needs no bounds AND the tile is strictly below the diagonal for both halves (J < ib0), so that
    // the loop carries exactly one epilogue (with both variants in it the register allocator spills the row panel around the epilogues, and a
    // scratch reload in the loop drains the LDS-DMA queue)
    const int nmain = RECT ? max(0, ntiles - TAIL_TILES) : max(0, min(ntiles - TAIL_TILES, ib0 - jt_begin));
    int t = 0;
    LSSVM_TRACE(2);
    for (; t < nmain; ++t) tile_body(t, std::false_type{});
    for (; t < ntiles; ++t) tile_body(t, std::true_type{});
    LSSVM_TRACE(3);
    // the leading half accompanies the lagging half's last LAGT steps (hand-overs only)
    if constexpr (HALF == 0) {
#pragma unroll
        for (int k = 0; k < LAGE; ++k) handover_checked(nsteps + k);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (!RECT) {
        if (jt_begin + ntiles - 1 < ib0 + 1) flush_cols(ntiles - 1);
    }

    // every lane group owns its rows: reduce over the 16 columns of the group and store
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // sum over the 16 columns of the lane group on the vector ALU (DPP: xor 1, xor 2 inside the quads, then the mirrored half row and the
        // mirrored row, which pair quads / halves whose sums are already uniform) -- no LDS round trips at the end of a work item
        float v = rowpart[i];
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1, 0, 3, 2]
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2, 3, 0, 1]
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
        if constexpr (KT == KT_LINEAR && F16) v *= a.out_scale;
        rowpart[i] = v;
    }
    if constexpr (KT == KT_RBFF) {  // the row's folded factor e_i
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const f32x4 ei = *reinterpret_cast<const f32x4 *>(cis + wave * 32 + 16 * rb + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) rowpart[4 * rb + e] *= ei[e];
        }
    }
    if constexpr (GRID) {  // the row's folded factor E_i, once per work item
#pragma unroll
        for (int i = 0; i < 8; ++i) rowpart[i] *= a.er[row0 + wave * 32 + 16 * (i >> 2) + 4 * g + (i & 3)];
    }
    if (r == 0) {
        float *dst = a.partial + static_cast<size_t>(jc) * a.part_stride + ibl * TILE + wave * 32 + 4 * g;
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[16 * (i >> 2) + (i & 3)] = rowpart[i];
    }
#ifdef LSSVM_ITEM_TRACE
    LSSVM_TRACE(4);
    if (HALF == 0 && lssvm_item_trace != nullptr && threadIdx.x == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        lssvm_item_trace[static_cast<size_t>(item_pos) * 8 + 5] = (static_cast<unsigned long long>(xcc) << 32) | hw;
        lssvm_item_trace[static_cast<size_t>(item_pos) * 8 + 6] = static_cast<unsigned long long>(ntiles);
    }
#endif
}

/* PL = 2: f16x3 ("f3d"), PL = 3: bf16x6 ("s6d").  LAGT: steps the second half runs behind (0 = lock step). */
template <int KT, int NK64, int PL, int LAGT>
__global__ __launch_bounds__(PR_THREADS, 2) LSSVM_HAND_VGPR_CAP void tile_matvec_f32_pair(const TileArgs<float> a) {
#ifdef LSSVM_CODE_SHIFT  // placement experiment (cdna_hip_programming.md section 5.4, rule 27): shift the instruction stream by 4 x LSSVM_CODE_SHIFT bytes
#pragma unroll
    for (int i = 0; i < LSSVM_CODE_SHIFT; ++i) asm volatile("s_nop 0");
#endif
    const bool first_half = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6) < 4;  // (a scalar branch: the halves are whole waves)
    for_each_work_item(a, [&](const auto &ai, int pos) {  // (one workgroup per item, or a persistent launch: lssvm_device_common.hip.hpp)
        if (first_half) {
            pair_body<KT, NK64, PL, 0, LAGT>(ai, pos);
        } else {
            pair_body<KT, NK64, PL, 1, LAGT>(ai, pos);
        }
    });
}

/* The rectangular instance (predict_values): one work item = a pair of row blocks of the points x a chunk of the support vectors' column tiles; the item list covers the
 * whole rectangle (Problem-less: lssvm_problem.hip, predict_values_impl builds it), persistent launches as above. */
template <int KT, int NK64, int PL>
__global__ __launch_bounds__(PR_THREADS, 2) LSSVM_HAND_VGPR_CAP void tile_matvec_f32_pair_rect(const TileArgs<float> a) {
    const bool first_half = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6) < 4;
    for_each_work_item(a, [&](const auto &ai, int pos) {
        if (first_half) {
            pair_body<KT, NK64, PL, 0, 0, true>(ai, pos);
        } else {
            pair_body<KT, NK64, PL, 1, 0, true>(ai, pos);
        }
    });
}

}  // namespace lssvm
