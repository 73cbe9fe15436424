// mfma_ceiling.hip -- measurement utility behind lssvm_mi355_measure_bf16_mfma_ceiling (include/plssvm_amd.h): what a BARE loop of
// v_mfma_f32_16x16x32_bf16 sustains on this device, on random operands, after the clock has settled.  The chip lowers its clock under
// matrix-core load (MI355X_MICROARCH.md "DVFS give-back"), so the nominal peak (4096 FLOP/clk/CU x CUs x 2.4 GHz) is not what a kernel
// is up against; bench.py reports this number beside roofline.frac, measured on the same device in the same process.  No counterpart in
// the reference (it has no roofline reporting); nothing on the solve path calls it.
#include "lssvm_problem.hip.hpp"

#include <random>

namespace lssvm {

using f32x4_c = float __attribute__((ext_vector_type(4)));
using bf16x8_c = __bf16 __attribute__((ext_vector_type(8)));
using f16x8_c = _Float16 __attribute__((ext_vector_type(8)));
using u32x4_c = unsigned __attribute__((ext_vector_type(4)));

struct CeilingStamp {
    unsigned long long cycles, realtime;
};

/* a 64 x 64 wave tile = 4 x 4 accumulators of 16 x 16, 64 deep per pass (2 k-steps of 32): 32 MFMAs per pass.
 * LDSB = 1: the B fragments are re-read from LDS (ds_read_b128) every pass, as the Gram kernel does; 0: both operands stay in registers.
 * F16: v_mfma_f32_16x16x32_f16 on f16 operands (the f16x3 kernels' instruction) instead of the bf16 form (same cycles; the operand bits differ) */
template <int LDSB, bool F16>
__global__ __launch_bounds__(256, 2) void k_bare_mfma_bf16(float *out, const u32x4_c *src, CeilingStamp *stamps, int passes) {
    __shared__ u32x4_c lds_b[4][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32x4_c araw[4], braw[4];
    for (int i = 0; i < 4; ++i) {
        araw[i] = src[(blockIdx.x % 61) * 2048 + (i * 256 + tid)];
        braw[i] = src[(blockIdx.x % 53) * 2048 + 1024 + (i * 256 + tid)];
        lds_b[wave][i][lane] = braw[i];
    }
    __syncthreads();
    f32x4_c acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 4; ++j) acc[i][k][j] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < passes; ++it) {
        bf16x8_c a[4], b[4];
        asm volatile("" ::: "memory");  // the LDS reads stay inside the loop
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = __builtin_bit_cast(bf16x8_c, araw[i]);
            b[i] = __builtin_bit_cast(bf16x8_c, LDSB ? lds_b[wave][i][lane] : braw[i]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if constexpr (F16) {
                        acc[i][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_c, a[i]), __builtin_bit_cast(f16x8_c, b[k]), acc[i][k], 0, 0, 0);
                    } else {
                        acc[i][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[k], acc[i][k], 0, 0, 0);
                    }
                }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 4; ++j) s += acc[i][k][j];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0) stamps[blockIdx.x] = CeilingStamp{ c1 - c0, r1 - r0 };
}

static uint16_t bf16_bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return static_cast<uint16_t>((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

void measure_bf16_mfma_ceiling(int device, int b_from_lds, double settle_ms, double *tflops_out, double *clock_ghz_out, double *nominal_tflops_out) {
    select_device_checked(device);
    hipDeviceProp_t prop{};
    LSSVM_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    const int cus = prop.multiProcessorCount;
    const int blocks = 2 * cus;  // two workgroups of four waves per CU = two waves per SIMD, the Gram kernel's occupancy
    const int passes = 40000;
    const size_t n16 = 64 * 2048;
    std::vector<uint16_t> host(n16 * 8);
    std::mt19937 gen(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    const bool f16 = (b_from_lds & 2) != 0;  // bit 1 of the flag: f16 operands and the f16 MFMA
    for (uint16_t &v : host) {
        const float x = nd(gen);
        if (f16) {
            const _Float16 h = static_cast<_Float16>(x);
            std::memcpy(&v, &h, 2);
        } else {
            v = bf16_bits(x);
        }
    }
    Stream st;
    st.create();
    DevBuf<u32x4_c> src;
    DevBuf<float> out;
    DevBuf<CeilingStamp> stamps;
    src.alloc_zero(n16, st.s);
    out.alloc_zero(static_cast<size_t>(blocks) * 256, st.s);
    stamps.alloc_zero(static_cast<size_t>(blocks), st.s);
    LSSVM_HIP_CHECK(hipMemcpyAsync(src.p, host.data(), n16 * 16, hipMemcpyHostToDevice, st.s));
    LSSVM_HIP_CHECK(hipStreamSynchronize(st.s));
    Event ea, eb;
    ea.create(true);
    eb.create(true);
    const auto launch = [&] {
        if ((b_from_lds & 1) != 0) {
            if (f16) hipLaunchKernelGGL((k_bare_mfma_bf16<1, true>), dim3(blocks), dim3(256), 0, st.s, out.p, src.p, stamps.p, passes);
            else hipLaunchKernelGGL((k_bare_mfma_bf16<1, false>), dim3(blocks), dim3(256), 0, st.s, out.p, src.p, stamps.p, passes);
        } else {
            if (f16) hipLaunchKernelGGL((k_bare_mfma_bf16<0, true>), dim3(blocks), dim3(256), 0, st.s, out.p, src.p, stamps.p, passes);
            else hipLaunchKernelGGL((k_bare_mfma_bf16<0, false>), dim3(blocks), dim3(256), 0, st.s, out.p, src.p, stamps.p, passes);
        }
    };
    // settle the clock under load, then time ten more launches
    const double t0 = now_ms();
    do {
        for (int r = 0; r < 5; ++r) launch();
        LSSVM_HIP_CHECK(hipGetLastError());
        LSSVM_HIP_CHECK(hipStreamSynchronize(st.s));
    } while (now_ms() - t0 < settle_ms);
    LSSVM_HIP_CHECK(hipEventRecord(ea.e, st.s));
    for (int r = 0; r < 10; ++r) launch();
    LSSVM_HIP_CHECK(hipEventRecord(eb.e, st.s));
    LSSVM_HIP_CHECK(hipEventSynchronize(eb.e));
    float ms = 0.f;
    LSSVM_HIP_CHECK(hipEventElapsedTime(&ms, ea.e, eb.e));
    std::vector<CeilingStamp> hs(static_cast<size_t>(blocks));
    LSSVM_HIP_CHECK(hipMemcpy(hs.data(), stamps.p, hs.size() * sizeof(CeilingStamp), hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (const CeilingStamp &c : hs) {
        if (c.realtime > 0) ghz.push_back(static_cast<double>(c.cycles) / static_cast<double>(c.realtime) * 0.1);  // s_memrealtime ticks at 100 MHz
    }
    std::sort(ghz.begin(), ghz.end());
    const double flop = 2.0 * 64 * 64 * 64 * static_cast<double>(passes) * blocks * 4;
    if (tflops_out != nullptr) *tflops_out = flop / (static_cast<double>(ms) / 10.0) / 1e9;
    if (clock_ghz_out != nullptr) *clock_ghz_out = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
    if (nominal_tflops_out != nullptr) *nominal_tflops_out = 4096.0 * cus * static_cast<double>(prop.clockRate) * 1e3 / 1e12;
}

}  // namespace lssvm
