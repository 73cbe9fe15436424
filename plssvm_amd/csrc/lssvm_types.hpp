/*
 * lssvm_types.hpp -- plain types and constants shared by the device kernels (lssvm_kernels.hip.hpp) and the host driver
 * (lssvm_problem.hip.hpp).  No kernels are defined here.
 */
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace lssvm {

constexpr int KT_LINEAR = 0;
constexpr int KT_POLY = 1;
constexpr int KT_RBF = 2;
constexpr int KT_POLY2 = 3;  // v2 tile kernels only: polynomial with degree 2 / 3 (KT_POLY = any other integer degree)
constexpr int KT_POLY3 = 4;
constexpr int KT_RBFF = 5;   // bf16x6 16x16x32 kernels only: rbf with the FOLDED records (w_j = 2^c_j d_j | e_j = 2^c_j): the accumulators start from c_i as
                             // the C operand of their first MFMA (no start-value adds), K_ij = 2^acc * e_j
constexpr int KT_RBFG = 6;   // f16 kernels of s6w_body only (round 5): rbf on GRID planes -- x = h + s1 + s2 with h on a grid of spacing g, so that the accumulator, started from
                             // sigma^2 (ch_i + ch_j) and fed the h.h products first, holds -|h_i - h_j|^2 / 2 EXACTLY before the small terms arrive

constexpr int TILE = 128;          // rows / columns of one workgroup tile of the implicit matrix
constexpr int TILE_THREADS = 256;  // 4 wave64 arranged 2 x 2, each owning a 64 x 64 sub-tile

using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x16 = float __attribute__((ext_vector_type(16)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using f64x2 = double __attribute__((ext_vector_type(2)));
using f64x4 = double __attribute__((ext_vector_type(4)));

/* Arguments of the tile kernel.  "rows" = the points whose output is produced (the I side), "cols" = the points summed
 * over (the J side).  Training matvec: both are the (padded) data matrix.  predict_values: rows = points to predict,
 * cols = support vectors (HIP/predict_kernel.hip.hpp:63-117 is the reference counterpart). */
template <typename T>
struct TileArgs {
    const T *Xr;      // [>= (ib_begin+num_ib)*TILE][ldx] row side, zero padded
    const T *Xrf;     // fp64 panels-inside-a-sub-tile kernel, symmetric variant: the row side FRAGMENT-MAJOR -- [ldx / 16][rows / 16][4 k-steps][64 lanes], lane 16 q + r
                      // = X[16 block + r][16 chunk + 4 step + q] (k_rows_fragment_major_f64): 512 contiguous bytes per A fragment of v_mfma_f64_16x16x4
    int frag_rows16;  // 16-row blocks of Xrf (rows_alloc / 16)
    const T *Xc;      // [>= num_jt*TILE][ldx]            column side, zero padded
    const T *cr;      // rbf: -0.5 * |x_i|^2 per row-side point
    const T *cc;      // rbf: -0.5 * |x_j|^2 per column-side point
    const T *dvec;    // [num_jt*TILE] vector multiplied from the right, EXACT zeros beyond the valid columns
    const T *er;      // KT_RBFG: E_i = 2^(c_i - ch_i) per row (the part of the exponent the grid norms leave out, folded as a factor); NULL otherwise
    int rbf_grid;     // host side only: != 0 selects the grid-plane rbf kernel (KT_RBFG)
    const T *dc;      // v2 kernels: packed [num_jt][256] records (d_j | c_j) for LDS-DMA (k_pack_dc); KT_RBFF: (2^c_j d_j | 2^c_j)
    int dc_folded;    // host side only: 1 if the records carry the folded form (rbf on the 16x16x32 bf16x6 kernels while |c| stays small)
    const uint16_t *Xr16;  // fp32 split kernels: the row side as three bf16 planes [3][rows][ldx16] (hi, mid, lo: x = hi + mid + lo exactly; "bf16x6")
                           // or as two f16 planes [2][rows][ldx16] (hi, mid of 2^k x; "f16x3")
    const uint16_t *Xr16f; // panels-inside-a-tile kernel, symmetric variant: the row side FRAGMENT-MAJOR -- [plane][ldx16 / 64][rows / 16][2][64 lanes][8], lane 16 g + r
                           // = features 8 g .. 8 g + 7 of row r of the block (k_planes_fragment_major); same plane stride as Xr16
    const uint16_t *Xc16;  // fp32 split kernels: the column side, same layout
    int planes_f16;        // host side only: 1 if the planes are the two f16 planes (f16x3 kernels), 0 for the three bf16 planes
    T out_scale;           // f16x3, linear kernel: 2^(-2k), undoes the power-of-two pre-scale of the planes on the finished sums (exact);
                           // polynomial: folded into gamma; rbf: k = 0
    size_t plane_stride;   // elements between the planes of the COLUMN side
    size_t plane_stride_r; // elements between the planes of the ROW side (training: the same matrix; predict_values: the points to predict)
    int ldx16;             // padded features of the planes (multiple of 64): the row stride of the planes
    int wide_panels;       // host side only: != 0 selects the kernel that walks feature panels of 128 inside a tile (rbf / polynomial beyond the register-resident row panel)
    int nk64;              // host side only: 64-feature chunks THIS launch contracts over (= ldx16 / 64, or one feature panel of a wide linear problem)
    const int2 *items; // symmetric variant: list of the non-empty (local row block, column chunk) work items
    int num_items;    // symmetric variant: length of `items` = grid size
    unsigned *queue;       // 256-row workgroups: NULL = one workgroup per item (grid = num_items); else the launch is PERSISTENT (one workgroup per CU) and the workgroups draw
                           // their items from eight counters, one per XCD lane of the item list (positions 8 k + x), [8][32] words apart -- lssvm_tile_f32_pair.hip.hpp
    unsigned *queue_next;  // the counters of this problem's NEXT launch: zeroed by this one
    int queue_grid;        // host side only: workgroups of a persistent launch (the CUs of the device: one 256-row workgroup fits a CU)
    T *colslab;       // symmetric variant: [packed (ib, jt) pairs with jt < ib][TILE] column sums of the off-diagonal tiles
    long pair_origin; // symmetric variant: ib_begin * (ib_begin - 1) / 2, the record index of this device's first pair
    T *partial;       // [num_jc][part_stride] partial row sums, one slab per column chunk, indexed by the LOCAL row
    long part_stride; // elements between slabs (>= num_ib*TILE)
    int ldx;          // padded number of features (multiple of the k-chunk)
    int kchunks;      // ldx / KC
    int ib_begin;     // first row block of this device (row-block sharding)
    int num_ib;       // number of row blocks of this device
    int num_jt;       // number of column tiles in total
    int jc_tiles;     // column tiles per work item
    int num_jc;       // number of column chunks = ceil(num_jt / jc_tiles); with a head (below): jc_head_count + ceil((num_jt - head tiles) / jc_tiles)
    int jc_head_tiles;  // 256-row workgroups only (0 elsewhere): the FIRST jc_head_count column chunks have this many tiles instead of jc_tiles -- short items that
    int jc_head_count;  // every row pair has, dispatched last: they fill the final dispatch round of a small launch (chunk_begin / chunk_len below)
    int row_pair;     // host side only: != 0 selects the 256-row-workgroup kernels (items = block pairs, lssvm_tile_f32_pair.hip.hpp)
    int rect;         // host side only: with row_pair, != 0 selects the RECTANGULAR 256-row kernel of predict_values (every tile in full, no mirrored column sums)
    int pair_lag;     // host side only: steps the second half of such a workgroup runs behind the first
    int mfma_shape;   // host side only: option mfma_shape (2 = 128-row workgroups, 3 = 256-row workgroups where they apply)
    int dbg;          // diagnostic ablations (timing only, results wrong): 1 = no global re-loads, 2 = no kernel function in the epilogue
    int ncols_valid;  // columns >= this are padding (used only where a padded column could produce inf/nan)
    int degree;       // polynomial
    T gamma;          // polynomial: gamma ; rbf fp64: 2 * gamma ; rbf fp32: unused (folded into the pre-scaled data) ; direct rbf: -gamma*log2(e)
    T coef0;          // polynomial
};

/* What k_update_d needs to leave the NEXT implicit matvec's column records behind (k_pack_dc / k_pack_dc_f64 folded into it, round 5: one launch less per CG
 * iteration -- VERDICT r04 item 6): dc == NULL = no packing.  `folded` as k_pack_dc's; `zero` / `nzero` the vector the symmetric variant adds into. */
template <typename T>
struct PackDc {
    T *dc = nullptr;
    const T *cc = nullptr;
    const T *efac = nullptr;
    T *zero = nullptr;
    int ncols = 0, nzero = 0, folded = 0;
};
/* column chunk jc of a launch covers the tiles [chunk_begin, chunk_begin + chunk_len): a head of `head_count` short chunks, then chunks of `tiles` */
__host__ __device__ inline int chunk_begin(int jc, int tiles, int head_tiles, int head_count) {
    return jc < head_count ? jc * head_tiles : head_count * head_tiles + (jc - head_count) * tiles;
}
__host__ __device__ inline int chunk_len(int jc, int tiles, int head_tiles, int head_count) { return jc < head_count ? head_tiles : tiles; }
/* number of chunks that begin at or before tile `last` */
__host__ __device__ inline int chunks_upto(int last, int tiles, int head_tiles, int head_count) {
    const int head = head_count * head_tiles;
    return last < head ? last / head_tiles + 1 : head_count + (last - head) / tiles + 1;
}

constexpr int F32_KC = 32;  // fp32: features per k-chunk (one 128-byte line per row)
constexpr int F32_LS = 36;  // fp32: padded LDS row stride in floats (144 B: 16-B aligned, conflict-free ds_read_b128)
constexpr int F64_KC = 16;  // fp64: features per k-chunk (128 bytes)
constexpr int F64_LS = 18;  // fp64: padded LDS row stride in doubles (144 B: conflict-free ds_read_b64)

constexpr int RED_BLOCKS = 256;   // fixed number of partial sums => reproducible reductions for every n
constexpr int RED_THREADS = 256;

/* device-resident scalars of the CG recursion (all double) */
enum ScalarSlot : int { SC_S = 0, SC_QD = 1, SC_DAD = 2, SC_DELTA = 3, SC_DELTA_OLD = 4, SC_ALPHA = 5, SC_BETA = 6, SC_DELTA0 = 7, SC_SUMX = 8, SC_QX = 9, SC_COUNT = 16 };
/* the partial sums of the reductions, RED_BLOCKS x 2 doubles per set: sum d | q.d (k_update_d), d.Ad (k_Ad_and_dAd), r.r (k_update_x_r, k_residual),
 * sum v | q.v (enqueue_sum_and_qdot) -- separate sets, because a kernel finishes its predecessor's sums while its own blocks write theirs */
enum PartSet : int { PART_D = 0, PART_DAD = 1, PART_RR = 2, PART_SUMS = 3, PART_REGIONS = 4 };

}  // namespace lssvm
