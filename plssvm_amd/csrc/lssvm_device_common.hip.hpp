/*
 * lssvm_device_common.hip.hpp -- device helpers shared by the gfx950 kernels of the LS-SVM CG hot path: kernel functions
 * (include/plssvm/kernel_function_types.hpp:75-97 of the reference), work-item decoding, scalar-base addressing for LDS-DMA.
 * Everything here is a template or __forceinline__, so the header can be included by several translation units.
 */
#pragma once

#include "lssvm_types.hpp"

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <type_traits>

/* timing-only ablations of the tile kernels (option "debug_ablate"); compiled in only with -DLSSVM_ENABLE_ABLATION so that the
 * shipped kernels carry no extra branches */
#ifdef LSSVM_ENABLE_ABLATION
#define LSSVM_DBG(a, bit) (((a).dbg & (bit)) != 0)
#else
#define LSSVM_DBG(a, bit) false
#endif

namespace lssvm {

using lds_ptr_t = __attribute__((address_space(3))) void *;
using gbl_ptr_t = const __attribute__((address_space(1))) void *;

/* integer power by repeated squaring; the reference uses pow(real, int) on the GPU (HIP/svm_kernel.hip.hpp:178) and
 * std::pow(real, real(degree)) on the CPU (kernel_function_types.hpp:86-89): equal up to rounding for integer degrees. */
template <typename T>
__device__ __forceinline__ T ipow(T base, int degree) {
    unsigned e = degree < 0 ? static_cast<unsigned>(-(long) degree) : static_cast<unsigned>(degree);
    T result = T(1);
    T b = base;
    while (e != 0u) {
        if (e & 1u) result *= b;
        b *= b;
        e >>= 1u;
    }
    return degree < 0 ? T(1) / result : result;
}

/* blockIdx.x -> (local row block, column chunk) of the full-square variant: consecutive workgroups walk the row blocks of one column chunk.
 * (An XCD-aware 8 x 8 super-tile map was measured equal at every size in round 2 -- the launch is neither tail- nor L2-bound -- and retired in round 4.) */
template <typename T>
__device__ __forceinline__ bool decode_work_item(const TileArgs<T> &a, int &ibl, int &jc) {
    const int id = blockIdx.x;
    ibl = id % a.num_ib;
    jc = id / a.num_ib;
    return true;
}

/* ------------------------------------------------------------------ work items of a symmetric launch ------------------------------------------------------------------
 * (Used by the 256-row workgroups, lssvm_tile_f32_pair.hip.hpp.  Round 5 also put EVERY kernel of the symmetric variant on it: nothing gained -- the fp64 kernels hold
 * 2.36 GHz on all XCDs, the 128-row fp32 kernels run two workgroups per CU -- and the item loop cost registers: fp64 linear 100 000 x 128 +4.8 %, native fp32 polynomial
 * +2.2 %, 7 000 x 128 +12 %; profiles/r05_ab_generic_queue.log.  Those kernels keep one workgroup per item.)
 * One workgroup per item (TileArgs::queue == NULL: item = list position blockIdx.x, dealt by the hardware), or a PERSISTENT launch: as many workgroups as the device
 * runs at once, each drawing list positions from eight counters until the list is empty.  The item list is laid out for the hardware's round-robin deal of
 * workgroups to the eight XCDs (position 8 k + x = lane x: a lane streams one column chunk at a time through ITS L2); a workgroup draws from the lane of the XCD it
 * runs on and, once that lane is empty, from the lane with the most items left.  Which CU evaluates an item changes no result (every item owns its slab rows and
 * records).  Why: the hardware's own deal is STATIC -- every XCD gets every eighth workgroup whatever its pace, and the XCDs of one chip differ by up to 8 % per tile
 * (per-XCD clocks; tests/tools/item_trace.py, profiles/r05_item_trace.log: at 1 000 000 x 128 the XCDs of one band launch end between 31.9 and 34.3 ms, 3.4 % of
 * CUs x time idle at its end; with the counters they end within 0.1 ms, the fast XCD having taken 9 % more tiles than the slow one).  It also dispatches in
 * order, so a CU waits for its successor while a workgroup further up the list waits for a slot elsewhere (8 ... 40 us in front of the short items at the end of a
 * list). */
__device__ __forceinline__ int work_queue_fetch(unsigned *ctr, int xcc, int num_items) {
    auto lane_items = [&](int x) { return (num_items - x + 7) >> 3; };
    auto taken = [&](int x) { return static_cast<int>(min(__hip_atomic_load(ctr + 32 * x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), static_cast<unsigned>(lane_items(x)))); };
    int x = xcc;  // (the own lane without a look first: one round trip to the counters per item, not two; a counter may overshoot its lane's length)
    for (;;) {
        if (x < 0) {
            int most = 0;
            for (int o = 0; o < 8; ++o) {
                const int left = lane_items(o) - taken(o);
                if (left > most) {
                    most = left;
                    x = o;
                }
            }
            if (x < 0) return -1;
        }
        const unsigned k = atomicAdd(ctr + 32 * x, 1u);
        if (k < static_cast<unsigned>(lane_items(x))) return x + 8 * static_cast<int>(k);
        x = -1;  // (another workgroup took the lane's last item in between)
    }
}

/* body(list position) for this workgroup's item(s).  The counters of the problem's NEXT launch (TileArgs::queue_next) are zeroed by workgroup 0. */
template <typename T, typename Body>
__device__ __forceinline__ void for_each_work_item(const TileArgs<T> &a, Body &&body) {
    __shared__ int next_pos;
    unsigned *const queue = a.queue;
    int xcc = 0;
    if (queue != nullptr) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc = static_cast<int>(id & 7u);
        if (blockIdx.x == 0 && threadIdx.x < 8) a.queue_next[32 * threadIdx.x] = 0u;
    }
    int pos = static_cast<int>(blockIdx.x);
    for (;;) {
        if (queue != nullptr) {
            if (threadIdx.x == 0) next_pos = work_queue_fetch(queue, xcc, a.num_items);
            __syncthreads();
            pos = __builtin_amdgcn_readfirstlane(next_pos);
            if (pos < 0) break;
        }
        // the kernel arguments through a pointer the compiler cannot see across iterations: else every field an item reads (and every 64-bit product of two of
        // them) is hoisted as loop invariant and stays in SGPRs for the whole item -- 20 to 30 of them spill into VGPR lanes, and the two-waves-per-SIMD kernels,
        // which have no VGPR to give, start spilling those.  (The argument struct is the kernel's only parameter: offset 0 of the kernarg segment.)
        // The floating-point scalars stay with the hoisted copy: two of them as SGPR operands of one v_fma_f32 inside the loop is a form this compiler's operand
        // folding produces and its own verifier then rejects ("violates constant bus restriction").
        auto kp = (const __attribute__((address_space(4))) TileArgs<T> *) __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        TileArgs<T> ai = *(const TileArgs<T> *) kp;
        ai.gamma = a.gamma;
        ai.coef0 = a.coef0;
        ai.out_scale = a.out_scale;
        body(ai, pos);
        if (queue == nullptr) break;
        __syncthreads();  // every wave has left the item's LDS (ring, records, column sums, next_pos) before the next prologue writes it
    }
}

/* exp(x) in double for the rbf epilogue: 2^k * p(r), k = rint(x log2 e), r = x - k ln2 (two-part ln2), p = degree-13 Taylor
 * polynomial on |r| <= 0.347 (truncation 4e-18), 19 double-precision VALU operations instead of libm's ~40 with its
 * special-case branches; v_ldexp_f64 handles underflow to 0 for very negative x.  Relative error < 2 ulp. */
__device__ __forceinline__ double fast_exp_f64(double x) {
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;
    p = fma(p, r, 2.08767569878681e-09);
    p = fma(p, r, 2.505210838544172e-08);
    p = fma(p, r, 2.755731922398589e-07);
    p = fma(p, r, 2.7557319223985893e-06);
    p = fma(p, r, 2.48015873015873e-05);
    p = fma(p, r, 1.984126984126984e-04);
    p = fma(p, r, 1.388888888888889e-03);
    p = fma(p, r, 8.333333333333333e-03);
    p = fma(p, r, 4.1666666666666664e-02);
    p = fma(p, r, 1.6666666666666666e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return __builtin_ldexp(p, static_cast<int>(k));
}

/* 2^t in double for the rbf epilogue of the fp64 kernels (the data carries sqrt(2 gamma log2(e)), so the MFMA chain leaves t = log2(K)): t = k + r, k = rint(t),
 * |r| <= 1/2, 2^r by a degree-10 polynomial -- the interpolant of 2^r in the Chebyshev nodes of [-1/2, 1/2] (near-minimax: 1.4 eps of truncation with the
 * coefficients rounded to double; the Taylor polynomial of rounds 1-4 needed degree 12 for 0.8 eps) --, v_ldexp_f64 for the scale (underflow to 0 included).
 * 14 vector instructions per element; every one of them costs matrix-core time beside v_mfma_f64 -- on gfx950 NO vector instruction overlaps with an fp64 MFMA,
 * integer and fp32 ones included (profiles/archive/r01_microbench_mfma_beside_valu.log: 4 v_fma_f64 per MFMA 77 -> 55 TFLOP/s, 4 integer instructions 77 -> 59), which
 * is why a v_exp_f32 seed corrected in fp64 cannot pay: a seed of 24 bits leaves the correction a full-length polynomial, and the
 * instructions it would move to the fp32 pipe cost the same issue time.  What is left is the count: 16 -> 14 here (p(0) = 1 exactly: K_ii = 1).
 * (A 64-entry table + degree-5 polynomial needs 13, but its per-lane LDS gathers serialise in the register-bound epilogue: measured 16 % SLOWER.) */
__device__ __forceinline__ double exp2_f64(double t) {
    const double k = __builtin_rint(t);
    const double r = t - k;
    double p = 7.072585949269224e-09;
    p = fma(p, r, 1.0208690299958306e-07);
    p = fma(p, r, 1.321544258792169e-06);
    p = fma(p, r, 1.5252657260200837e-05);
    p = fma(p, r, 0.0001540353044173605);
    p = fma(p, r, 0.0013333558230164974);
    p = fma(p, r, 0.009618129107606888);
    p = fma(p, r, 0.05550410866444772);
    p = fma(p, r, 0.24022650695910097);
    p = fma(p, r, 0.69314718055995);
    p = fma(p, r, 1.0);
    return __builtin_ldexp(p, static_cast<int>(k));
}

/* DEG: polynomial degree class resolved OUTSIDE the per-element loop (a uniform switch around the whole epilogue):
 * 3 = cube, 2 = square, 0 = generic integer power.  Ignored for the other kernels. */
template <int KT, int DEG, typename T>
__device__ __forceinline__ T apply_kernel_function(T acc, const TileArgs<T> &a) {
    if constexpr (KT == KT_LINEAR) {
        return acc;
    } else if constexpr (KT == KT_POLY) {
        const T v = acc * a.gamma + a.coef0;  // contracted to one fma, = std::fma(gamma, dot, coef0)
        if constexpr (DEG == 3) {
            return v * v * v;
        } else if constexpr (DEG == 2) {
            return v * v;
        } else {
            return ipow(v, a.degree);
        }
    } else {
        if constexpr (std::is_same_v<T, float>) {
            // fp32: the data was pre-scaled by sqrt(2*gamma*log2(e)) at set-up, so acc = -gamma*log2(e)*|xi-xj|^2 already
            return __builtin_amdgcn_exp2f(acc);
        } else {
            return fast_exp_f64(acc * a.gamma);  // acc = -|xi-xj|^2 / 2 ; gamma field = 2*gamma
        }
    }
}

/* v^degree for the degree class DEG (3, 2, or 0 = any integer degree) */
template <int DEG, typename T>
__device__ __forceinline__ T poly_power(T v, int degree) {
    if constexpr (DEG == 3) {
        return v * v * v;
    } else if constexpr (DEG == 2) {
        return v * v;
    } else {
        return ipow(v, degree);
    }
}

/* runs `body(std::integral_constant<int, DEG>)` with the polynomial degree class of `a` (one uniform branch per tile) */
template <int KT, typename T, typename F>
__device__ __forceinline__ void with_degree_class(const TileArgs<T> &a, F &&body) {
    if constexpr (KT == KT_POLY) {
        if (a.degree == 3) {
            body(std::integral_constant<int, 3>{});
        } else if (a.degree == 2) {
            body(std::integral_constant<int, 2>{});
        } else {
            body(std::integral_constant<int, 0>{});
        }
    } else {
        body(std::integral_constant<int, 0>{});
    }
}

/* Pins a uniform pointer into an SGPR pair at this point of the program: "uniform base + 32-bit lane offset" is then selected
 * as the saddr form of global_load_lds / global_store (no 64-bit vector address arithmetic, one VGPR per lane offset), and the
 * compiler cannot re-associate the base into several vector adds. */
__device__ __forceinline__ const char *sgpr_ptr(const void *p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<unsigned>(v))));  // (the builtin returns int:
    const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<unsigned>(v >> 32))));  // no sign extension)
    unsigned long long u = (static_cast<unsigned long long>(hi) << 32) | lo;
    asm volatile("" : "+s"(u));
    return reinterpret_cast<const char *>(u);
}

/* Keeps the 32-bit -> 64-bit extension of a lane offset in the basic block of its use (instruction selection is per block:
 * a zext hoisted out of the loop hides the "SGPR base + 32-bit VGPR offset" addressing mode from it). */
__device__ __forceinline__ unsigned lane_off(unsigned v) {
    asm volatile("" : "+v"(v));
    return v;
}

/* compile-time loop: f(std::integral_constant<int, I>{}) for I = BEGIN .. END-1 (indices usable as template arguments / asm immediates) */
template <int BEGIN, int END, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (BEGIN < END) {
        f(std::integral_constant<int, BEGIN>{});
        static_for<BEGIN + 1, END>(f);
    }
}

/* Hand-issued LDS read + counted wait.  The compiler waits for its own LDS reads with s_waitcnt lgkmcnt(0) when a group of reads and their
 * first use meet in one block, which exposes the full LDS latency in front of every MFMA group although the fragments needed were
 * requested a whole group earlier.  These reads are invisible to its counter logic: the code that uses them states the wait itself
 * (LDS operations return in order, so "all but the N newest" is exact as long as N later reads were issued, whatever else the compiler adds
 * in between only makes the wait longer).  The wait takes the registers as in/out operands so that no consumer can be scheduled above it. */
template <int OFF>
__device__ __forceinline__ f32x4 lds_read_b128(unsigned lds_addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "i"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void lds_wait4(f32x4 &b0, f32x4 &b1, f32x4 &b2, f32x4 &b3) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "i"(N));
}

/* v + (v of lane ^ 32) and v + (v of lane ^ 16) WITHOUT the LDS crossbar: gfx950's v_permlane32_swap / v_permlane16_swap exchange the upper
 * half (odd 16-lane rows) of the first register with the lower half (even rows) of the second; fed with two copies of v the two results are
 * [lo, lo] and [hi, hi], whose sum is the butterfly step.  (__shfl_xor compiles to ds_bpermute_b32 + s_waitcnt lgkmcnt(0): an LDS round
 * trip per step, which serialised the eight column blocks of the symmetric epilogue -- 24 LDS latencies per tile.  The builtin form of the
 * swaps mis-compiles when both operands are the same value (ROCm 7.2: it adds the first result to itself), hence the asm.) */
__device__ __forceinline__ float sum_with_lane_xor32(float v) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float sum_with_lane_xor16(float v) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

/* The same two butterfly steps for EIGHT values at once (the column sums of a tile's eight 16-column blocks: lane (g, r) holds its lane
 * group's rows of column 16 cb + r in c[cb]) without copies: a swap of two DIFFERENT registers followed by one add reduces both by half --
 * v_permlane32_swap a, b leaves a = [a.lo | b.lo], b = [a.hi | b.hi], so a + b = [a reduced over the halves | b reduced over the halves].
 * Afterwards c[0] holds the finished sum of block q in lane group q (column = lane), c[4] that of block 4 + q (column = 64 + lane):
 * 6 swaps + 6 adds instead of 16 + 16 (+ 16 copies), and the values come out one column per lane, so that the factor of the column record
 * and the store are two instructions of all 64 lanes instead of eight of one lane group.  Every sum associates exactly as
 * sum_with_lane_xor16(sum_with_lane_xor32(v)) does in lane group 0: (g0 + g2) + (g1 + g3) -- the bits do not change.
 * (2 wait states between a VALU write and a swap that reads it: cdna_hip_programming.md T21.) */
__device__ __forceinline__ void column_sums_of_8_blocks(float (&c)[8]) {
    asm("s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %2\n\t"
        "v_permlane32_swap_b32 %1, %3\n\t"
        "v_permlane32_swap_b32 %4, %6\n\t"
        "v_permlane32_swap_b32 %5, %7\n\t"
        "v_add_f32 %0, %0, %2\n\t"
        "v_add_f32 %1, %1, %3\n\t"
        "v_add_f32 %4, %4, %6\n\t"
        "v_add_f32 %5, %5, %7\n\t"
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %1\n\t"
        "v_permlane16_swap_b32 %4, %5\n\t"
        "v_add_f32 %0, %0, %1\n\t"
        "v_add_f32 %4, %4, %5"
        : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
}

/* One LDS-DMA instruction, 16 bytes per lane: global (uniform base in SGPRs + 32-bit lane offset) -> LDS (uniform destination M0 = lds_base +
 * LDS_OFF, lane l lands at + 16 l).  As inline asm because the builtin's operands go through the compiler's address selection: with the lane
 * offsets living across the tile loop it copied each offset into a scratch register in front of every instruction (v_mov) and, where the base
 * was a sum of uniform terms, built a 64-bit PER-LANE address with two v_lshl_add_u64 -- 24 vector instructions per tile beside the MFMA
 * stream, each of which costs matrix-core issue time.  Here the base is pinned to SGPRs (scalar adds) and the offset
 * register is used as it is.  The instruction counts in vmcnt like the builtin's; the kernels wait for it with counted s_waitcnt. */
template <int LDS_OFF>
__device__ __forceinline__ void lds_dma16(unsigned lane_offset, const char *uniform_base, unsigned uniform_lds_base) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(lane_offset), "s"(uniform_base), "s"(uniform_lds_base), "i"(LDS_OFF)
                 : "memory", "scc");  // (M0 cannot be declared: it is a reserved register.  A kernel that uses this helper issues ALL its LDS-DMA through it,
                                      //  so the compiler holds no M0 value of its own across these statements.)
}

__device__ __forceinline__ double sum_with_lane_xor32(double v) {
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
    unsigned alo = static_cast<unsigned>(bits), blo = alo, ahi = static_cast<unsigned>(bits >> 32), bhi = ahi;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3" : "+v"(alo), "+v"(blo), "+v"(ahi), "+v"(bhi));
    return __builtin_bit_cast(double, (static_cast<unsigned long long>(ahi) << 32) | alo) + __builtin_bit_cast(double, (static_cast<unsigned long long>(bhi) << 32) | blo);
}
__device__ __forceinline__ double sum_with_lane_xor16(double v) {
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
    unsigned alo = static_cast<unsigned>(bits), blo = alo, ahi = static_cast<unsigned>(bits >> 32), bhi = ahi;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3" : "+v"(alo), "+v"(blo), "+v"(ahi), "+v"(bhi));
    return __builtin_bit_cast(double, (static_cast<unsigned long long>(ahi) << 32) | alo) + __builtin_bit_cast(double, (static_cast<unsigned long long>(bhi) << 32) | blo);
}

/* The paired butterfly for doubles (column sums of the fp64 sub-tile's four 16-column blocks).  A double is a register pair: the swaps work on
 * its two dwords, passed to the asm as separate 32-bit operands (sub-registers of the pairs: no copies).  Inline asm with the wait states INSIDE
 * the statement, not the builtin: the builtin leaves the "VALU write -> v_permlane*_swap read" hazard (2 wait states, cdna_hip_programming.md
 * T21) to the compiler's hazard recognizer, which missed it where the value was written at the end of the previous basic block -- the
 * generic-degree polynomial instantiation of tile_matvec_f64_wide, whose integer-power loop ends right in front of the butterfly, returned
 * wrong column sums in a quarter of the lanes (found by tests/tools/wide_stress.py).
 * pair32(a, b) = [a summed over the wave's halves | b summed over the halves]; pair16(a, b) = [a0 + a1, b0 + b1, a2 + a3, b2 + b3] over the
 * 16-lane rows. */
__device__ __forceinline__ double swap_halves_and_add(double a, double b, std::integral_constant<int, 32>) {
    const unsigned long long ab = __builtin_bit_cast(unsigned long long, a), bb = __builtin_bit_cast(unsigned long long, b);
    unsigned alo = static_cast<unsigned>(ab), ahi = static_cast<unsigned>(ab >> 32), blo = static_cast<unsigned>(bb), bhi = static_cast<unsigned>(bb >> 32);
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3" : "+v"(alo), "+v"(ahi), "+v"(blo), "+v"(bhi));
    return __builtin_bit_cast(double, (static_cast<unsigned long long>(ahi) << 32) | alo) + __builtin_bit_cast(double, (static_cast<unsigned long long>(bhi) << 32) | blo);
}
__device__ __forceinline__ double swap_halves_and_add(double a, double b, std::integral_constant<int, 16>) {
    const unsigned long long ab = __builtin_bit_cast(unsigned long long, a), bb = __builtin_bit_cast(unsigned long long, b);
    unsigned alo = static_cast<unsigned>(ab), ahi = static_cast<unsigned>(ab >> 32), blo = static_cast<unsigned>(bb), bhi = static_cast<unsigned>(bb >> 32);
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3" : "+v"(alo), "+v"(ahi), "+v"(blo), "+v"(bhi));
    return __builtin_bit_cast(double, (static_cast<unsigned long long>(ahi) << 32) | alo) + __builtin_bit_cast(double, (static_cast<unsigned long long>(bhi) << 32) | blo);
}
/* c[cb] = this lane group's rows of column 16 cb + r  ->  returns the finished sum of block q in lane group q (column = lane): 6 swaps + 3 adds
 * instead of 16 + 8 (+ the copies the one-value form needs).  Associates as sum_with_lane_xor16(sum_with_lane_xor32(v)) does in lane group 0. */
__device__ __forceinline__ double column_sums_of_4_blocks(const double (&c)[4]) {
    const double s02 = swap_halves_and_add(c[0], c[2], std::integral_constant<int, 32>{});
    const double s13 = swap_halves_and_add(c[1], c[3], std::integral_constant<int, 32>{});
    return swap_halves_and_add(s02, s13, std::integral_constant<int, 16>{});
}

/* The v2 kernels take the polynomial degree class as part of their kernel-type template parameter, so every instantiation
 * carries ONE epilogue (the three-way runtime switch of with_degree_class made the register allocator budget for the generic
 * integer-power path and spill in the cube path). */
__host__ __device__ constexpr int v2_base_kt(int kt) { return (kt == KT_POLY2 || kt == KT_POLY3) ? KT_POLY : ((kt == KT_RBFF || kt == KT_RBFG) ? KT_RBF : kt); }
__host__ __device__ constexpr int v2_degree_class(int kt) { return kt == KT_POLY3 ? 3 : (kt == KT_POLY2 ? 2 : 0); }

/* LDS geometry of the fp32 v2 kernels (tile_matvec_f32_v2, tile_matvec_f32_s6) */
constexpr int V2_RING = 4;                       // chunk slots in LDS
constexpr int V2_SLOT_BYTES = TILE * 32 * 4;     // 16 KiB
constexpr int V2_DC_SLOTS = 4;                   // ring of per-tile (d_j | c_j) records, 1 KiB each
constexpr size_t V2_LDS_BYTES = static_cast<size_t>(V2_RING) * V2_SLOT_BYTES + V2_DC_SLOTS * 1024 + (2 * TILE + 2 * 4 * TILE) * sizeof(float);  // ring + records + cis, dis, colred

}  // namespace lssvm
