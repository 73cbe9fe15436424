// microbench_bf16.hip -- what a BARE bf16 matrix-core loop sustains on the box, on random and on all-zero operands: the practical ceiling
// the bf16x6 Gram kernel (lssvm_tile_f32_split.hip.hpp) is up against.  The chip lowers its clock under an MFMA-dense load
// (MI355X_MICROARCH.md "DVFS give-back"), so the nominal 2.5 PFLOP/s (4096 FLOP/clk/CU x 256 CUs x 2.4 GHz) is not reachable on
// random data by any kernel; this program measures how much is.  Every arm runs back-to-back launches for >= 2 s before it is timed
// and reports the wall rate next to the in-kernel clock (s_memtime / s_memrealtime around the loop, median over workgroups).
// Build: hipcc -O3 --offload-arch=gfx950 tests/tools/microbench_bf16.hip -o /tmp/microbench_bf16 ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x16 = float __attribute__((ext_vector_type(16)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using u32x4 = unsigned __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

struct Stamp { unsigned long long cyc, rt; };

// SHAPE 0: v_mfma_f32_32x32x16_bf16, a 64x64 wave tile = 2x2 accumulators of 32x32 (64 registers)
// SHAPE 1: v_mfma_f32_16x16x32_bf16, a 64x64 wave tile = 4x4 accumulators of 16x16 (64 registers)
// LDSB 1: the B fragments are re-read from LDS by ds_read_b128 every pass (as the production kernel does), 0: both operands stay in registers
template <int SHAPE, int LDSB>
__global__ __launch_bounds__(256, 2) void k_bare(float *out, const u32x4 *src, Stamp *stamps, int iters) {
    constexpr int NF = SHAPE == 0 ? 2 : 4;
    __shared__ u32x4 lds_b[4][NF][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32x4 araw[NF], braw[NF];
    for (int i = 0; i < NF; ++i) {
        araw[i] = src[(blockIdx.x % 61) * 2048 + (i * 256 + tid)];
        braw[i] = src[(blockIdx.x % 53) * 2048 + 1024 + (i * 256 + tid)];
        lds_b[wave][i][lane] = braw[i];
    }
    __syncthreads();
    f32x16 acc0[2][2];
    f32x4 acc1[4][4];
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 2; ++k) for (int j = 0; j < 16; ++j) acc0[i][k][j] = 0.f;
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) for (int j = 0; j < 4; ++j) acc1[i][k][j] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        bf16x8 a[NF], b[NF];
        asm volatile("" ::: "memory");  // the LDS reads stay inside the loop
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            a[i] = __builtin_bit_cast(bf16x8, araw[i]);
            if (LDSB) {
                b[i] = __builtin_bit_cast(bf16x8, lds_b[wave][i][lane]);
            } else {
                b[i] = __builtin_bit_cast(bf16x8, braw[i]);
            }
        }
#pragma unroll
        for (int u = 0; u < (SHAPE == 0 ? 4 : 2); ++u)
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int k = 0; k < NF; ++k) {
                    if (SHAPE == 0) acc0[i][k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[k], acc0[i][k], 0, 0, 0);
                    else acc1[i][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[k], acc1[i][k], 0, 0, 0);
                }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 2; ++k) for (int j = 0; j < 16; ++j) s += acc0[i][k][j];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) for (int j = 0; j < 4; ++j) s += acc1[i][k][j];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0) stamps[blockIdx.x] = Stamp{ c1 - c0, r1 - r0 };
}

static uint16_t to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return (uint16_t) ((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s %s CUs=%d nominal clock=%d kHz; nominal bf16 peak = 4096 FLOP/clk/CU -> %.1f TFLOP/s\n", prop.name, prop.gcnArchName, cus, prop.clockRate,
           4096.0 * cus * prop.clockRate * 1e3 / 1e12);
    const size_t n16 = 64 * 2048;  // uint4 elements
    std::vector<uint16_t> h(n16 * 8);
    u32x4 *src;
    float *out;
    Stamp *stamps;
    CHECK(hipMalloc(&src, n16 * 16));
    CHECK(hipMalloc(&out, (size_t) cus * 2 * 256 * 4));
    CHECK(hipMalloc(&stamps, (size_t) cus * 2 * sizeof(Stamp)));
    hipEvent_t ea, eb;
    CHECK(hipEventCreate(&ea));
    CHECK(hipEventCreate(&eb));
    const int iters = 60000;  // about 25-50 ms per launch
    for (int data = 0; data < 3; ++data) {  // 0 random normal (the high plane of centred, scaled features), 1 random bits in all 16 (finite), 2 zeros
        std::mt19937 gen(7);
        std::normal_distribution<float> nd(0.f, 1.f);
        for (auto &v : h) {
            if (data == 0) v = to_bf16(nd(gen));
            else if (data == 1) { do { v = (uint16_t) gen(); } while ((v & 0x7F80) == 0x7F80); v = (v & 0x807F) | (uint16_t) ((120 + (v >> 7) % 8) << 7); }  // finite, exponents near 1
            else v = 0;
        }
        CHECK(hipMemcpy(src, h.data(), n16 * 16, hipMemcpyHostToDevice));
        const char *dname = data == 0 ? "normal(0,1) operands" : (data == 1 ? "random mantissa bits" : "all-zero operands");
        for (int wps = 1; wps <= 2; ++wps)
            for (int variant = 0; variant < 4; ++variant) {
                const int shape = variant & 1, ldsb = variant >> 1;
                const int blocks = cus * wps;
                auto launch = [&] {
                    if (variant == 0) hipLaunchKernelGGL((k_bare<0, 0>), dim3(blocks), dim3(256), 0, 0, out, src, stamps, iters);
                    else if (variant == 1) hipLaunchKernelGGL((k_bare<1, 0>), dim3(blocks), dim3(256), 0, 0, out, src, stamps, iters);
                    else if (variant == 2) hipLaunchKernelGGL((k_bare<0, 1>), dim3(blocks), dim3(256), 0, 0, out, src, stamps, iters);
                    else hipLaunchKernelGGL((k_bare<1, 1>), dim3(blocks), dim3(256), 0, 0, out, src, stamps, iters);
                };
                launch();
                CHECK(hipDeviceSynchronize());
                // settle the clock: >= 2 s of back-to-back launches, then time 10 more
                (void) hipEventRecord(ea);
                float spent = 0.f;
                while (spent < 2000.f) {
                    for (int r = 0; r < 10; ++r) launch();
                    (void) hipEventRecord(eb);
                    (void) hipEventSynchronize(eb);
                    (void) hipEventElapsedTime(&spent, ea, eb);
                }
                (void) hipEventRecord(ea);
                for (int r = 0; r < 10; ++r) launch();
                (void) hipEventRecord(eb);
                (void) hipEventSynchronize(eb);
                float ms;
                (void) hipEventElapsedTime(&ms, ea, eb);
                ms /= 10.f;
                std::vector<Stamp> st(blocks);
                CHECK(hipMemcpy(st.data(), stamps, blocks * sizeof(Stamp), hipMemcpyDeviceToHost));
                std::vector<double> ghz(blocks);
                for (int i = 0; i < blocks; ++i) ghz[i] = (double) st[i].cyc / (double) st[i].rt * 0.1;  // s_memrealtime ticks at 100 MHz
                std::sort(ghz.begin(), ghz.end());
                const double flop = 2.0 * 64 * 64 * 64 * (double) iters * blocks * 4;  // a 64x64 wave tile, 64 deep per pass (4 x 16 or 2 x 32)
                const double cyc_per_pass = (double) st[blocks / 2].cyc / iters;
                printf("%-22s %s %s, %d wave/SIMD: %7.1f TFLOP/s = %.3f of nominal; in-kernel clock %.2f GHz (median), %.1f cycles per 64x64x64 pass (ideal %d)\n", dname,
                       shape == 0 ? "32x32x16" : "16x16x32", ldsb ? "B from LDS " : "B in regs  ", wps, flop / ms / 1e9, flop / ms / 1e9 / (4096.0 * cus * prop.clockRate * 1e3 / 1e12),
                       ghz[blocks / 2], cyc_per_pass, 512 * wps);
            }
    }
    return 0;
}
