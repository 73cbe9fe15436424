/*
 * lssvm_tile_f64.hip.hpp -- the fp64 tile kernels of the implicit kernel-matrix--vector product:
 * tile_matvec_f64_v2 (resident row panel, LDS-DMA ring, symmetric or full square) and tile_matvec_f64 (generic).
 * Included by tile_launch_f64.hip only.
 */
#pragma once

#include "lssvm_device_common.hip.hpp"

namespace lssvm {

/* =====================================================================================================================
 * fp64 tile kernel: v_mfma_f64_16x16x4_f64
 *   lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15] (one f64 each); the 16x16 result has column
 *   j = l&15 on the lane and rows (l>>4) + 4*reg in its 4 registers (NOT the f32 row map).
 *   LDS image of a k-chunk: [128 rows][16 doubles + 2 pad] (144-B rows: conflict-free ds_read_b64).
 * ===================================================================================================================== */

template <int KT>
__global__ __launch_bounds__(TILE_THREADS, 1) void tile_matvec_f64(const TileArgs<double> a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *As = reinterpret_cast<double *>(smem_raw);  // [2][TILE * F64_LS]
    double *Bs = As + 2 * TILE * F64_LS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1;
    const int wc = wave & 1;
    const int r = lane & 15;
    const int qd = lane >> 4;

    int ibl, jc;
    if (!decode_work_item(a, ibl, jc)) return;
    const int row0 = (a.ib_begin + ibl) * TILE;
    const int jt_begin = jc * a.jc_tiles;
    const int jt_end = min(jt_begin + a.jc_tiles, a.num_jt);
    const int ntiles = jt_end - jt_begin;
    if (ntiles <= 0) return;

    // staging: thread -> (row = tid/8 [+32 p], 2 doubles at (tid%8)*2); 8 threads cover one 128-B line
    const int srow = tid >> 3;
    const int sseg = tid & 7;
    const double *Ag = a.Xr + static_cast<size_t>(row0 + srow) * a.ldx + sseg * 2;
    const size_t rstep = static_cast<size_t>(32) * a.ldx;
    const int lds_w = srow * F64_LS + sseg * 2;

    double rowpart[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) rowpart[mt][i] = 0.0;

    f64x4 acc[4][4];
    f64x2 sa[4], sb[4];

    auto stage_load = [&](int jt, int kc) {
        const double *Bg = a.Xc + static_cast<size_t>(jt * TILE + srow) * a.ldx + sseg * 2;
        const int ko = kc * F64_KC;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            sa[p] = *reinterpret_cast<const f64x2 *>(Ag + p * rstep + ko);
            sb[p] = *reinterpret_cast<const f64x2 *>(Bg + p * rstep + ko);
        }
    };
    auto stage_store = [&](int buf) {
        double *Aw = As + buf * TILE * F64_LS + lds_w;
        double *Bw = Bs + buf * TILE * F64_LS + lds_w;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<f64x2 *>(Aw + p * 32 * F64_LS) = sa[p];
            *reinterpret_cast<f64x2 *>(Bw + p * 32 * F64_LS) = sb[p];
        }
    };

    double ci[4][4];  // rbf: c_i of this lane's 16 rows (1 wave per SIMD: the register budget is 512)
    if constexpr (KT == KT_RBF) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) ci[mt][i] = a.cr[row0 + wr * 64 + mt * 16 + qd + 4 * i];
    }
    double dj[4], cj[4], djn[4], cjn[4];
    bool padcol[4] = { false, false, false, false };
    auto col_prefetch = [&](int jt) {  // one tile ahead, see the fp32 kernel
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int j = jt * TILE + wc * 64 + nt * 16 + r;
            djn[nt] = a.dvec[j];
            if constexpr (KT == KT_RBF) cjn[nt] = a.cc[j];
        }
    };
    auto tile_init = [&](int jt) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            dj[nt] = djn[nt];
            cj[nt] = 0.0;
            if constexpr (KT == KT_RBF) cj[nt] = cjn[nt];
            if constexpr (KT == KT_POLY) padcol[nt] = (a.degree < 0) && (jt * TILE + wc * 64 + nt * 16 + r >= a.ncols_valid);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double civ = 0.0;
                if constexpr (KT == KT_RBF) civ = ci[mt][i];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt][i] = civ + cj[nt];
            }
    };

    const int nsteps = ntiles * a.kchunks;
    stage_load(jt_begin, 0);
    col_prefetch(jt_begin);
    tile_init(jt_begin);
    stage_store(0);
    __syncthreads();

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
        if (has_next) stage_load(njt, nkc);
        if (kc == 0 && jt + 1 < jt_end) col_prefetch(jt + 1);

        {
            const double *Ab = As + cur * TILE * F64_LS + (wr * 64 + r) * F64_LS + qd;
            const double *Bb = Bs + cur * TILE * F64_LS + (wc * 64 + r) * F64_LS + qd;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                double av[4], bv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    av[t] = Ab[t * 16 * F64_LS + ks * 4];
                    bv[t] = Bb[t * 16 * F64_LS + ks * 4];
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
            }
        }

        if (has_next) stage_store(cur ^ 1);

        if (kc == a.kchunks - 1) {
            with_degree_class<KT>(a, [&](auto degc) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            double kv = apply_kernel_function<KT, decltype(degc)::value>(acc[mt][nt][i], a);
                            if constexpr (KT == KT_POLY) {
                                if (padcol[nt]) kv = 0.0;
                            }
                            rowpart[mt][i] = fma(kv, dj[nt], rowpart[mt][i]);
                        }
            });
            if (has_next) tile_init(njt);
        }
        __syncthreads();
        jt = njt;
        kc = nkc;
    }

    // rows are shared by the 16 lanes of a quarter-wave (same l>>4)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double v = rowpart[mt][i];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            rowpart[mt][i] = v;
        }
    double *red = reinterpret_cast<double *>(smem_raw);  // [2][TILE]
    if (r == 0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wc * TILE + wr * 64 + mt * 16 + qd + 4 * i] = rowpart[mt][i];
    }
    __syncthreads();
    if (tid < TILE) {
        a.partial[static_cast<size_t>(jc) * a.part_stride + ibl * TILE + tid] = red[tid] + red[TILE + tid];
    }
}

/* =====================================================================================================================
 * fp64 tile kernel, version 2: the fp32 v2 pipeline (row panel resident in registers, LDS-DMA ring two chunks ahead, chunk
 * hand-over in the middle of a step, packed (d_j | c_j) records) on v_mfma_f64_16x16x4_f64, for num_features <= 256.
 *   A column tile of 128 is processed as two 64-column SUB-TILES so that a wave's accumulators (32 rows x 64 columns =
 *   8 tiles of 16x16 = 64 VGPRs) plus its row panel leave room for two workgroups per CU.
 *   Chunk = 64 columns x 16 features = 8 KiB = 4 k-steps of 8 MFMAs per wave.
 *   v_mfma_f64 does NOT overlap with vector ALU instructions (tests/tools/microbench_f64.hip: one integer VALU op per MFMA
 *   costs 9 % of the matrix-core rate, one v_fma_f64 15 %), so the chunk loop consists of MFMAs, LDS reads with immediate
 *   offsets, LDS-DMA with scalar base addresses and scalar instructions only, the accumulators start from the constant 0 as
 *   the C operand of the first MFMA, and the polynomial kernel runs on data pre-scaled by sqrt(gamma).
 * ===================================================================================================================== */
constexpr int V2D_RING = 4;
constexpr int V2D_SLOT_BYTES = 64 * 128;  // 8 KiB
constexpr int V2D_DC_SLOTS = 4;           // (64 d_j | 64 c_j) doubles = 1 KiB per sub-tile
constexpr size_t V2D_LDS_BYTES = static_cast<size_t>(V2D_RING) * V2D_SLOT_BYTES + V2D_DC_SLOTS * 1024 + (2 * TILE + 2 * 4 * 64) * sizeof(double);  // ring + records + cis, dis, colred

template <int KT, int NKC, bool SYM>
__global__ __launch_bounds__(TILE_THREADS, (NKC <= 4 ? 2 : 1)) void tile_matvec_f64_v2(const TileArgs<double> a) {
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

    // SYM: see tile_matvec_f32_v2.  A sub-tile st is strictly below the diagonal block of row block ib iff st < 2 ib; the two
    // sub-tiles of the diagonal tile are evaluated in full and contribute to the rows only.
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
    const int nsub = 2 * (jt_end - jt_begin);  // 64-column sub-tiles
    if (nsub <= 0) return;
    const int st_begin = 2 * jt_begin;
    const int nsteps = nsub * NKC;
    // record index of (ib, st) in this device's packed column slab: row block b owns the 2 b sub-tiles below its diagonal
    const long rec0 = SYM ? (static_cast<long>(ib) * (ib - 1) - 2 * a.pair_origin) : 0;

    // row panel: A operand of lane (r, q) for k-step s is X[row][4 s + q]
    double afrag[2][4 * NKC];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const double *xr = a.Xr + static_cast<size_t>(row0 + wave * 32 + rb * 16 + r) * a.ldx + q;
#pragma unroll
        for (int s = 0; s < 4 * NKC; ++s) afrag[rb][s] = xr[4 * s];
    }
    if constexpr (KT == KT_RBF) {
        if (tid < TILE) cis[tid] = a.cr[row0 + tid];
    }
    if constexpr (SYM) {
        if (tid < TILE) dis[tid] = a.dvec[row0 + tid];
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int s = 0; s < 4 * NKC; ++s) asm volatile("" : "+v"(afrag[rb][s]));

    // LDS image of a chunk (64 columns x 16 features, 8 KiB): [column block cb = 0..3][16-byte k-slot ks = 0..7][column r = 0..15],
    // i.e. byte cb * 2048 + ks * 256 + r * 16 holds features 2 ks, 2 ks + 1 of column cb * 16 + r.  Piece 2 * wave + i of the DMA
    // (1 KiB, lane-linear in LDS) is block cb = wave, k-slots 4 i .. 4 i + 3: lane L fetches the 16 bytes of column L % 16, k-slot
    // 4 i + L / 16 -- a gather on the SOURCE side (16 rows x 64 contiguous bytes per piece).
    unsigned dma_off[2];  // byte offsets (saddr form: uniform base in SGPRs + 32-bit lane offset, see tile_matvec_f32_v2)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int col = wave * 16 + (lane & 15);
        const int ks = 4 * i + (lane >> 4);
        dma_off[i] = 8u * static_cast<unsigned>(col * a.ldx + 2 * ks);
    }
    const unsigned ring_lds = static_cast<unsigned>(reinterpret_cast<size_t>(ring));  // the low half of a generic LDS address is the LDS address
    const unsigned dma_lds = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(ring_lds + static_cast<unsigned>(wave) * 2048u)));  // this wave's quarter of ring slot 0
    const unsigned dc_lds = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(ring_lds + V2D_RING * V2D_SLOT_BYTES + static_cast<unsigned>(wave) * 256u)));
    auto issue_chunk = [&](int step) {
        if (LSSVM_DBG(a, 16) && step > 2) return;  // ablation: no DMA after the prologue
        const int t = step / NKC;
        const int kc = step - t * NKC;
        const char *base = sgpr_ptr(a.Xc + static_cast<size_t>(st_begin + t) * 64 * a.ldx + kc * 16);
        const unsigned slot = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(dma_lds + static_cast<unsigned>(step % V2D_RING) * V2D_SLOT_BYTES)));
        lds_dma16<0>(dma_off[0], base, slot);  // (inline asm: the builtin copied the lane offset into a scratch register in front of every instruction)
        lds_dma16<1024>(dma_off[1], base, slot);
    };
    auto issue_dc = [&](int t) {
        if (lane < 16) {
            const char *src = sgpr_ptr(a.dc + static_cast<size_t>(st_begin + t) * 128) + __builtin_amdgcn_readfirstlane(wave * 256);
            lds_dma16<0>(16u * (lane_off(threadIdx.x) & 15u), sgpr_ptr(src), static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(dc_lds + static_cast<unsigned>(t % V2D_DC_SLOTS) * 1024u))));
        }
    };

    // Read addressing: lane (r, q) needs feature 4 s + q of column cb * 16 + r for k-step s = k-slot 2 s + q / 2, half q % 2:
    // byte (q / 2) * 256 + r * 16 + (q % 2) * 8 [per lane, constant] + cb * 2048 + s * 512 [immediates] + ring slot [one add per
    // chunk].  The 32 lanes of a ds_read_b64 group (q / 2 fixed) read 256 contiguous bytes: conflict free without a swizzle.
    const int lane_base = (q >> 1) * 256 + r * 16 + (q & 1) * 8;
    auto read_group = [&](const char *slot, int s, double (&b)[4]) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) b[cb] = *((const volatile __attribute__((address_space(3))) double *) (slot + cb * 2048 + s * 512));  // volatile: keeps ds_read_b64 (a fused ds_read2st64_b64 is banked modulo 32 dwords: 2-way conflicts here)
    };

    double rowpart[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 4; ++i) rowpart[rb][i] = 0.0;
    f64x4 acc[2][4];
    double dj[4], cj[4];
    bool padcol[4] = { false, false, false, false };

    // ---- prologue: chunks 0, 1, 2 (each preceded by the record of the sub-tile that starts with it) ----
    issue_dc(0);
    issue_chunk(0);
#pragma unroll
    for (int pre = 1; pre <= 2; ++pre) {
        if (pre < nsteps) {
            if (pre % NKC == 0) issue_dc(pre / NKC);
            issue_chunk(pre);
        }
    }
    // chunk 0 (and record 0, cis, dis) complete: everything but the DMA pieces of the younger chunks is done
    if (nsteps >= 3) {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    } else if (nsteps == 2) {
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    double bcur[4];  // B fragments of the k-step about to be multiplied (double buffered against bnext in the loop)
    read_group(ring + lane_base, 0, bcur);

    // ---- hand-over of the NEXT chunk in the MIDDLE of a step (see tile_matvec_f32_v2): called in k-step 2 of chunk `step`.  This
    // wave's two pieces of chunk step + 1 are complete once all but its 2 youngest DMA instructions (chunk step + 2) are; the barrier
    // makes every wave's pieces visible, so k-step 3 can already prefetch the first fragments of chunk step + 1.  Ring of 4 slots:
    // the DMA issued here (chunk step + 3) overwrites the slot of chunk step - 1, which every wave finished before this barrier.
    auto handover = [&](int step, int kc_plus3_mod, auto checked) {
        constexpr bool CHECKED = decltype(checked)::value;
        if constexpr (!CHECKED) {
            if (!LSSVM_DBG(a, 16)) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            if (!LSSVM_DBG(a, 8)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kc_plus3_mod == 0) issue_dc((step + 3) / NKC);
            issue_chunk(step + 3);
        } else {
            if (step + 1 < nsteps) {
                if (step + 2 < nsteps) {
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
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

    auto flush_cols = [&](int t) {  // fixed-order sum of the four waves' column sums of sub-tile t -> its slab record
        if (tid < 64) {
            const double *cr_ = colred + (t & 1) * 256;
            // (an explicit GLOBAL pointer: through the generic one the store is a flat_store, which counts in lgkmcnt as well and completes out of order)
            auto *rec = (__attribute__((address_space(1))) double *) (a.colslab + (rec0 + st_begin + t) * 64);  // uniform base + 32-bit lane offset
            rec[static_cast<unsigned>(tid)] = (cr_[tid] + cr_[64 + tid]) + (cr_[128 + tid] + cr_[192 + tid]);
        }
    };

    auto tile_body = [&](int t, auto checked) {
        const int s0 = t * NKC;
        const bool tile_sym = SYM && (st_begin + t < 2 * ib);
        const double *dcr = reinterpret_cast<const double *>(dcs + (t % V2D_DC_SLOTS) * 1024);
        // rbf: the accumulators start at c_i + c_j; the other kernels start the chain with the constant 0 as the C operand of the
        // first MFMA (no register initialisation: 64 v_mov per sub-tile would cost as much matrix-core time as the cube)
        if constexpr (KT == KT_RBF) {
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
        }
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            const int step = s0 + kc;
            const char *slot = ring + (step % V2D_RING) * V2D_SLOT_BYTES + lane_base;
            const char *slot_next = ring + ((step + 1) % V2D_RING) * V2D_SLOT_BYTES + lane_base;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                // software prefetch of the next k-step's B fragments (next chunk for s == 3: visible since this step's hand-over)
                double bnext[4];
                if (s < 3) read_group(slot, s + 1, bnext);
                if (s == 2) {
                    if constexpr (SYM) {
                        // the colred writes of the previous sub-tile's epilogue must have completed before the barrier publishes them
                        if (kc == 0 && t > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    handover(step, (kc + 3) % NKC, checked);
                    if constexpr (SYM) {
                        // sub-tile t - 1 was off-diagonal unless it is the first of the diagonal pair
                        if (kc == 0 && t > 0 && (st_begin + t - 1 < 2 * ib)) flush_cols(t - 1);
                    }
                }
                if (s == 3) read_group(slot_next, 0, bnext);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        if (KT != KT_RBF && kc == 0 && s == 0) {
                            const f64x4 zero = { 0.0, 0.0, 0.0, 0.0 };
                            acc[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[rb][0], bcur[cb], zero, 0, 0, 0);
                        } else {
                            acc[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[rb][4 * kc + s], bcur[cb], acc[rb][cb], 0, 0, 0);
                        }
                    }
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) bcur[cb] = bnext[cb];
            }
        }
        // ---- epilogue of the sub-tile (vector ALU; every instruction here costs matrix-core time, see the header) ----
        if (!LSSVM_DBG(a, 4)) {
            // d_j is fetched from the sub-tile's record only now: it need not occupy registers during the MFMA loop
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                dj[cb] = dcr[cb * 16 + r];
                if constexpr (KT == KT_POLY) padcol[cb] = (a.degree < 0) && ((st_begin + t) * 64 + cb * 16 + r >= a.ncols_valid);
            }
            if constexpr (v2_base_kt(KT) == KT_POLY) {
                if (a.coef0 != 0.0) {  // uniform; the common coef0 = 0 costs nothing
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[rb][cb][i] += a.coef0;
                }
            }
            auto epilogue = [&](auto with_cols) {
                constexpr bool COLS = decltype(with_cols)::value;
                double colacc[4] = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        double di = 0.0;
                        if constexpr (COLS) di = dis[wave * 32 + rb * 16 + q + 4 * i];
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb) {
                            double kv;
                            if constexpr (v2_base_kt(KT) == KT_POLY) {
                                // the data carries sqrt(gamma) (Problem<double> pre-scales it for this kernel) and coef0 was added above
                                kv = poly_power<v2_degree_class(KT)>(acc[rb][cb][i], a.degree);
                            } else if constexpr (KT == KT_RBF) {
                                kv = exp2_f64(acc[rb][cb][i]);  // the data is pre-scaled: acc = log2(K)
                            } else {
                                kv = acc[rb][cb][i];
                            }
                            if constexpr (KT == KT_POLY) {
                                if (padcol[cb]) kv = 0.0;
                            }
                            rowpart[rb][i] = fma(kv, dj[cb], rowpart[rb][i]);
                            if constexpr (COLS) colacc[cb] = fma(kv, di, colacc[cb]);
                        }
                    }
                if constexpr (COLS) {
                    double *cw = colred + (t & 1) * 256 + wave * 64;
                    // the four quarter-waves hold different rows of the same column: butterfly on the vector ALU (v_permlane*_swap; __shfl_xor
                    // is an LDS round trip per step and, with a store branch per column block, serialised them) for all four blocks at once;
                    // the sums come out one column per lane (block q in lane group q), so the store is one instruction of the whole wave
                    cw[lane] = column_sums_of_4_blocks(colacc);
                }
            };
            // (ONE epilogue per instantiation: with a branch between a column-sum and a row-only variant the compiler gave the eight row sums
            // different registers in the two arms and merged them with 16 v_mov_b64 per sub-tile -- vector instructions that cost matrix-core
            // time here.  The diagonal sub-tiles of a symmetric launch therefore compute column sums too; they are never flushed.)
            (void) tile_sym;
            epilogue(std::integral_constant<bool, SYM>{});
        }
    };

    // steady state: every sub-tile whose last step still has step + 3 < nsteps; then the tail sub-tiles with the checked hand-over
    constexpr int TAIL_TILES = (3 + NKC - 1) / NKC;
    const int nmain = nsub > TAIL_TILES ? nsub - TAIL_TILES : 0;
    int t = 0;
    for (; t < nmain; ++t) tile_body(t, std::false_type{});
    for (; t < nsub; ++t) tile_body(t, std::true_type{});
    if constexpr (SYM) {
        if (st_begin + nsub - 1 < 2 * ib) {  // the last sub-tile was off-diagonal: publish its column sums
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            flush_cols(nsub - 1);
        }
    }

    // rows are shared by the 16 lanes of a quarter-wave
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
