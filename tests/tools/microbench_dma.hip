// microbench_dma.hip -- what the LDS-DMA instructions of the bf16x6 Gram kernel cost beside its MFMA stream.
// A bare v_mfma_f32_16x16x32_bf16 loop (two waves per SIMD, 64x64 wave tiles) with, per 32 MFMAs, four global_load_lds_dwordx4 issued
// in one of several forms; the source is a small L2-resident buffer, the destination a 64 KiB LDS ring, nothing reads the data.
//   mode 0: no DMA
//   mode 1: four DMA, each with its own M0 (LDS base) and its own per-lane global offset register  (what the kernel does today)
//   mode 2: four DMA with ONE M0 and one global offset register, told apart by the instruction's immediate offset (0, 1024, 2048, 3072)
//   mode 3: as 1 but only ONE DMA per 32 MFMAs (a quarter of the instructions)
// Build: hipcc -O3 --offload-arch=gfx950 tests/tools/microbench_dma.hip -o /tmp/microbench_dma ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

using f32x4 = float __attribute__((ext_vector_type(4)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using u32x4 = unsigned __attribute__((ext_vector_type(4)));
using lds_ptr_t = __attribute__((address_space(3))) void *;
using gbl_ptr_t = const __attribute__((address_space(1))) void *;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_mfma_dma(float *out, const u32x4 *src, const char *stream, int iters) {
    extern __shared__ __attribute__((aligned(16))) char ring[];  // 64 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    u32x4 araw[4], braw[4];
    for (int i = 0; i < 4; ++i) {
        araw[i] = src[(blockIdx.x % 61) * 2048 + (i * 256 + tid)];
        braw[i] = src[(blockIdx.x % 53) * 2048 + 1024 + (i * 256 + tid)];
    }
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 4; ++j) acc[i][k][j] = 0.f;
    unsigned goff[4];
    for (int i = 0; i < 4; ++i) goff[i] = static_cast<unsigned>((blockIdx.x % 32) * 65536 + wave * 4096 + i * 2048 * (MODE == 2 ? 0 : 1) + lane * 16 + (MODE == 2 ? 0 : i * 64));
    for (int it = 0; it < iters; ++it) {
        bf16x8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = __builtin_bit_cast(bf16x8, araw[i]);
            b[i] = __builtin_bit_cast(bf16x8, braw[i]);
        }
        char *slot = ring + (it & 3) * 16384 + wave * 4096;
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((gbl_ptr_t) (stream + goff[i]), (lds_ptr_t) (slot + i * 1024), 16, 0, 0);
        } else if (MODE == 2) {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t) (stream + goff[0]), (lds_ptr_t) slot, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t) (stream + goff[0]), (lds_ptr_t) slot, 16, 1024, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t) (stream + goff[0]), (lds_ptr_t) slot, 16, 2048, 0);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t) (stream + goff[0]), (lds_ptr_t) slot, 16, 3072, 0);
        } else if (MODE == 3) {
            __builtin_amdgcn_global_load_lds((gbl_ptr_t) (stream + goff[0]), (lds_ptr_t) slot, 16, 0, 0);
        }
        if (MODE != 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[i][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[k], acc[i][k], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 4; ++j) s += acc[i][k][j];
    out[blockIdx.x * blockDim.x + tid] = s;
}

// The production shape: a 32 x 128 wave tile (2 x 8 accumulators of 16 x 16), per pass two k-steps = 32 MFMAs whose 16 B fragments are
// read from the LDS ring (ds_read_b128, conflict-free lane-linear image), beside 0 / 4 DMA into the slot three ahead.
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_like(float *out, const u32x4 *src, const char *stream, int iters) {
    extern __shared__ __attribute__((aligned(16))) char ring[];  // 64 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 65536 / 16; i += 256) reinterpret_cast<u32x4 *>(ring)[i] = src[i % 8192];
    __syncthreads();
    u32x4 araw[4];
    for (int i = 0; i < 4; ++i) araw[i] = src[(blockIdx.x % 61) * 2048 + (i * 256 + tid)];
    f32x4 acc[2][8];
    for (int i = 0; i < 2; ++i)
        for (int k = 0; k < 8; ++k)
            for (int j = 0; j < 4; ++j) acc[i][k][j] = 0.f;
    unsigned goff[4];
    for (int i = 0; i < 4; ++i) goff[i] = static_cast<unsigned>((blockIdx.x % 32) * 65536 + wave * 4096 + i * 2048 + lane * 16 + i * 64);
    for (int it = 0; it < iters; ++it) {
        const char *rd = ring + (it & 3) * 16384 + lane * 16;
        char *slot = ring + ((it + 3) & 3) * 16384 + wave * 4096;
        if (MODE & 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((gbl_ptr_t) (stream + goff[i]), (lds_ptr_t) (slot + i * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 b[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (MODE & 2) b[c] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(rd + kk * 8192 + c * 1024));
                else b[c] = __builtin_bit_cast(bf16x8, araw[(c + kk) & 3]);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, araw[2 * kk]), b[c], acc[0][c], 0, 0, 0);
                acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, araw[2 * kk + 1]), b[c], acc[1][c], 0, 0, 0);
            }
        }
        if (MODE & 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        if (MODE & 4) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int k = 0; k < 8; ++k)
            for (int j = 0; j < 4; ++j) s += acc[i][k][j];
    out[blockIdx.x * blockDim.x + tid] = s;
}

static uint16_t to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return (uint16_t) ((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, blocks = 2 * cus, iters = 40000;
    const size_t n16 = 64 * 2048;
    std::vector<uint16_t> h(n16 * 8);
    std::mt19937 gen(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto &v : h) v = to_bf16(nd(gen));
    u32x4 *src;
    char *stream;
    float *out;
    CHECK(hipMalloc(&src, n16 * 16));
    CHECK(hipMalloc(&stream, 4u << 20));
    for (size_t off = 0; off < (4u << 20); off += n16 * 16) CHECK(hipMemcpy(stream + off, h.data(), n16 * 16, hipMemcpyHostToDevice));  // random bf16 like the ring's first fill: zeros would raise the clock
    CHECK(hipMalloc(&out, (size_t) blocks * 256 * 4));
    CHECK(hipMemcpy(src, h.data(), n16 * 16, hipMemcpyHostToDevice));
    hipEvent_t ea, eb;
    CHECK(hipEventCreate(&ea));
    CHECK(hipEventCreate(&eb));
    const char *names[4] = { "no DMA", "4 DMA / 32 MFMA, own M0 + own offset register each", "4 DMA / 32 MFMA, one M0, immediate offsets", "1 DMA / 32 MFMA" };
    double base = 0.0;
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            auto launch = [&] {
                if (mode == 0) hipLaunchKernelGGL((k_mfma_dma<0>), dim3(blocks), dim3(256), 65536, 0, out, src, stream, iters);
                else if (mode == 1) hipLaunchKernelGGL((k_mfma_dma<1>), dim3(blocks), dim3(256), 65536, 0, out, src, stream, iters);
                else if (mode == 2) hipLaunchKernelGGL((k_mfma_dma<2>), dim3(blocks), dim3(256), 65536, 0, out, src, stream, iters);
                else hipLaunchKernelGGL((k_mfma_dma<3>), dim3(blocks), dim3(256), 65536, 0, out, src, stream, iters);
            };
            if (rep == 0) {
                const void *fn = mode == 0 ? (const void *) k_mfma_dma<0> : mode == 1 ? (const void *) k_mfma_dma<1> : mode == 2 ? (const void *) k_mfma_dma<2> : (const void *) k_mfma_dma<3>;
                CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
            }
            launch();
            CHECK(hipDeviceSynchronize());
            float spent = 0.f;
            (void) hipEventRecord(ea);
            while (spent < 1500.f) {
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
            if (mode == 0) base = ms;
            const double flop = 2.0 * 64 * 64 * 64 * (double) iters * blocks * 4;
            printf("rep %d  %-55s %8.3f ms  %7.1f TFLOP/s  %+6.2f %% vs no DMA\n", rep, names[mode], ms, flop / ms / 1e9, (ms / base - 1.0) * 100.0);
        }
    // the production shape
    const char *lnames[8] = { "regs, no DMA", "regs + 4 DMA", "B from LDS, no DMA", "B from LDS + 4 DMA", "regs, barrier", "regs + 4 DMA, barrier", "B from LDS, barrier", "B from LDS + 4 DMA, barrier" };
    const int order[6] = { 0, 1, 2, 3, 6, 7 };
    for (int rep = 0; rep < 2; ++rep)
        for (int oi = 0; oi < 6; ++oi) {
            const int mode = order[oi];
            auto launch = [&] {
                switch (mode) {
                    case 0: hipLaunchKernelGGL((k_like<0>), dim3(blocks), dim3(256), 65536, 0, out, src, stream, iters); break;
                    case 1: hipLaunchKernelGGL((k_like<1>), dim3(blocks), dim3(256), 65536, 0, out, src, stream, iters); break;
                    case 2: hipLaunchKernelGGL((k_like<2>), dim3(blocks), dim3(256), 65536, 0, out, src, stream, iters); break;
                    case 3: hipLaunchKernelGGL((k_like<3>), dim3(blocks), dim3(256), 65536, 0, out, src, stream, iters); break;
                    case 6: hipLaunchKernelGGL((k_like<6>), dim3(blocks), dim3(256), 65536, 0, out, src, stream, iters); break;
                    default: hipLaunchKernelGGL((k_like<7>), dim3(blocks), dim3(256), 65536, 0, out, src, stream, iters); break;
                }
            };
            if (rep == 0) {
                const void *fn = mode == 0 ? (const void *) k_like<0> : mode == 1 ? (const void *) k_like<1> : mode == 2 ? (const void *) k_like<2> : mode == 3 ? (const void *) k_like<3>
                                 : mode == 6 ? (const void *) k_like<6> : (const void *) k_like<7>;
                CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
            }
            launch();
            CHECK(hipDeviceSynchronize());
            float spent = 0.f;
            (void) hipEventRecord(ea);
            while (spent < 1500.f) {
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
            if (mode == 0) base = ms;
            const double flop = 2.0 * 32 * 128 * 64 * (double) iters * blocks * 4;
            printf("rep %d  32x128 wave tile: %-32s %8.3f ms  %7.1f TFLOP/s  %+6.2f %% vs regs, no DMA\n", rep, lnames[mode], ms, flop / ms / 1e9, (ms / base - 1.0) * 100.0);
        }
    return 0;
}
