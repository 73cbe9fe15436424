// microbench.hip -- measures the matrix-core / vector-ALU peaks of the box that the roofline fractions are quoted against.
// Build: hipcc -O3 --offload-arch=gfx950 tests/tools/microbench.hip -o /tmp/microbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x16 = float __attribute__((ext_vector_type(16)));
using f64x4 = double __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int NACC>
__global__ void k_mfma_f32_32x32x2(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void k_mfma_f32_16x16x4(float *out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void k_mfma_f64_16x16x4(double *out, int iters, double a0, double b0) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    double a = a0 + threadIdx.x * 1e-3, b = b0 + threadIdx.x * 2e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MFMA fed from LDS with the production kernel's read pattern (8 ds_read_b128 per 32 MFMAs), one barrier per 64 MFMAs
__global__ __launch_bounds__(256, 2) void k_mfma_lds(float *out, int iters, int use_barrier) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < 2 * 128 * 36; i += 256) lds[i] = (i % 17) * 0.01f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 2; ++k) for (int j = 0; j < 16; ++j) acc[i][k][j] = 0.f;
    const float *Ab = lds + ((wave >> 1) * 64 + r) * 36 + h * 4;
    const float *Bb = lds + 128 * 36 + ((wave & 1) * 64 + r) * 36 + h * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(Ab + g * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4 *>(Ab + 32 * 36 + g * 8);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(Bb + g * 8);
            const f32x4 b1 = *reinterpret_cast<const f32x4 *>(Bb + 32 * 36 + g * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b1[t], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b0[t], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc[1][1], 0, 0, 0);
            }
        }
        if (use_barrier) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 2; ++k) for (int j = 0; j < 16; ++j) s += acc[i][k][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// the production v2 pattern: A fragments resident in registers, B fragments from a swizzled lane-linear LDS image (4 x b128 per 16
// MFMAs), 4 accumulators of 32x32, one barrier per 64 MFMAs; optional LDS-DMA of the next chunk (dma = 1) from a small L2-resident buffer
using lds_ptr_t = __attribute__((address_space(3))) void *;
using gbl_ptr_t = const __attribute__((address_space(1))) void *;
template <int SPREAD, int PRIO>
__global__ __launch_bounds__(256, 2) void k_v2_like(float *out, const float *src, int iters, int use_barrier, int dma) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4 * 16384 / 4; i += 256) reinterpret_cast<float *>(smem)[i] = (i % 19) * 0.01f;
    __syncthreads();
    f32x4 afrag[16];
    for (int m = 0; m < 16; ++m) afrag[m] = f32x4{ 0.01f * m, 0.02f * lane, 0.5f, 0.25f };
    int rd_off[4];
    for (int mm = 0; mm < 4; ++mm) rd_off[mm] = r * 128 + (((2 * mm + h) ^ ((r >> 1) & 7)) << 4);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const float *gsrc = src + (size_t) (blockIdx.x % 64) * 4096 + wave * 1024 + lane * 4;
    for (int it4 = 0; it4 < iters; it4 += 4) {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const int it = it4 + kc;
            const char *slot = smem + kc * 16384;
            char *dst = smem + ((kc + 3) & 3) * 16384 + wave * 4096;
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                if (mm == 2) {
                    if (dma) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    if (use_barrier) __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if (dma && !SPREAD) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((gbl_ptr_t) (gsrc + i * 256), (lds_ptr_t) (dst + i * 1024), 16, 0, 0);
                    }
                }
                f32x4 b[4];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) b[cb] = *reinterpret_cast<const f32x4 *>(slot + cb * 4096 + rd_off[mm]);
                const f32x4 av = afrag[4 * kc + mm];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tt], b[cb][tt], acc[cb], 0, 0, 0);
                    if (PRIO) __builtin_amdgcn_s_setprio(0);
                    if (SPREAD && dma && mm >= 2 && (tt & 1) == 0) {
                        const int i = (mm - 2) * 2 + (tt >> 1);
                        __builtin_amdgcn_global_load_lds((gbl_ptr_t) (gsrc + i * 256), (lds_ptr_t) (dst + i * 1024), 16, 0, 0);
                    }
                }
            }
            (void) it;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_valu_fma_f32(float *out, int iters, float a0) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = a0 + i + threadIdx.x;
    const float m = 1.0000001f, c = 1e-7f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], m, c);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch, int reps = 5) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a);
        launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s %s CUs=%d clock=%d kHz\n", prop.name, prop.gcnArchName, cus, prop.clockRate);
    void *buf;
    CHECK(hipMalloc(&buf, 256u << 20));
    const int iters = 4000;
    for (int wps = 1; wps <= 2; ++wps) {  // waves per SIMD
        const int blocks = cus * wps, threads = 256;  // 4 waves per block -> one per SIMD per block
        {
            double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma_f32_32x32x2<4>, dim3(blocks), dim3(threads), 0, 0, (float *) buf, iters, 1.f, 2.f); });
            double flop = 2.0 * 32 * 32 * 2 * 8.0 * 4 * iters * (double) blocks * 4;
            printf("mfma_f32_32x32x2  4 acc, %d wave/SIMD: %.1f TFLOP/s\n", wps, flop / ms / 1e9);
        }
        {
            double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma_f32_32x32x2<1>, dim3(blocks), dim3(threads), 0, 0, (float *) buf, iters, 1.f, 2.f); });
            double flop = 2.0 * 32 * 32 * 2 * 8.0 * 1 * iters * (double) blocks * 4;
            printf("mfma_f32_32x32x2  1 acc, %d wave/SIMD: %.1f TFLOP/s\n", wps, flop / ms / 1e9);
        }
        {
            double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma_f32_16x16x4<8>, dim3(blocks), dim3(threads), 0, 0, (float *) buf, iters, 1.f, 2.f); });
            double flop = 2.0 * 16 * 16 * 4 * 8.0 * 8 * iters * (double) blocks * 4;
            printf("mfma_f32_16x16x4  8 acc, %d wave/SIMD: %.1f TFLOP/s\n", wps, flop / ms / 1e9);
        }
        {
            double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma_f64_16x16x4<8>, dim3(blocks), dim3(threads), 0, 0, (double *) buf, iters, 1.0, 2.0); });
            double flop = 2.0 * 16 * 16 * 4 * 8.0 * 8 * iters * (double) blocks * 4;
            printf("mfma_f64_16x16x4  8 acc, %d wave/SIMD: %.1f TFLOP/s\n", wps, flop / ms / 1e9);
        }
        {
            double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma_f64_16x16x4<1>, dim3(blocks), dim3(threads), 0, 0, (double *) buf, iters, 1.0, 2.0); });
            double flop = 2.0 * 16 * 16 * 4 * 8.0 * 1 * iters * (double) blocks * 4;
            printf("mfma_f64_16x16x4  1 acc, %d wave/SIMD: %.1f TFLOP/s\n", wps, flop / ms / 1e9);
        }
        for (int bar = 0; bar <= 1; ++bar) {
            const size_t lds = 2 * 128 * 36 * 4;
            double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma_lds, dim3(blocks), dim3(threads), lds, 0, (float *) buf, iters / 4, bar); });
            double flop = 2.0 * 32 * 32 * 2 * 64.0 * (iters / 4) * (double) blocks * 4;
            printf("mfma_f32 fed from LDS (prod. pattern), barrier=%d, %d wave/SIMD: %.1f TFLOP/s\n", bar, wps, flop / ms / 1e9);
        }

        for (int variant = 0; variant < 3; ++variant) {
            for (int mode = 0; mode < 3; ++mode) {  // 0: no barrier, 1: barrier, 2: barrier + LDS-DMA
                const size_t lds = 4 * 16384;
                auto fn = variant == 0 ? k_v2_like<0, 0> : (variant == 1 ? k_v2_like<1, 0> : k_v2_like<1, 1>);
                hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
                double ms = time_ms([&] { hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), lds, 0, (float *) buf, (const float *) buf + (32u << 20), iters / 4, mode >= 1, mode == 2); });
                double flop = 2.0 * 32 * 32 * 2 * 64.0 * (iters / 4) * (double) blocks * 4;
                printf("v2-like variant %d (0 burst DMA, 1 spread DMA, 2 spread+setprio) mode=%d (0 none,1 barrier,2 barrier+DMA), %d wave/SIMD: %.1f TFLOP/s\n", variant, mode, wps, flop / ms / 1e9);
            }
        }
    }
    for (int wps : {2, 4, 8}) {
        const int blocks = cus * wps, threads = 256;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_valu_fma_f32, dim3(blocks), dim3(threads), 0, 0, (float *) buf, iters * 4, 1.f); });
        double flop = 2.0 * 64.0 * (iters * 4) * (double) blocks * threads;
        printf("v_fma_f32 %d waves/SIMD: %.1f TFLOP/s\n", wps, flop / ms / 1e9);
    }
    return 0;
}
