// Micro-benchmarks for the fp64 tile kernel's inner loop on gfx950: what can run beside v_mfma_f64_16x16x4_f64?
//   build: hipcc -O3 --offload-arch=gfx950 tests/tools/microbench_f64.hip -o plssvm_amd/lib/microbench_f64
// Every kernel runs the production shape: 2 row blocks x 4 column blocks = 8 accumulators, 32 MFMAs per "step" (a k-chunk of 16),
// A fragments in registers, B fragments (a) constant registers or (b) ds_read_b64 from an LDS image.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

typedef double f64x4 __attribute__((ext_vector_type(4)));

// LDS: 0 = B in registers, 1 = ds_read_b64 with one precomputed address per (cb) and an immediate per s (compiler may fuse reads),
//      2 = ds_read_b64 with xor + add address arithmetic per read (the production kernel's form)
// IVALU: extra 32-bit integer VALU instructions per MFMA;  DVALU: extra v_fma_f64 per MFMA;  BAR: s_barrier per step
template <int LDS, int IVALU, int DVALU, int BAR>
__global__ __launch_bounds__(256, 2) void k_loop(double *out, int steps, int ring_mask) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int r = lane & 15, q = lane >> 4;
    for (int i = tid; i < 3 * 8192 / 8; i += 256) reinterpret_cast<double *>(smem)[i] = 1e-3 * (i & 31);
    __syncthreads();
    double afrag[2][4];
    for (int rb = 0; rb < 2; ++rb)
        for (int s = 0; s < 4; ++s) afrag[rb][s] = 1.0 + 1e-3 * (lane + s + 4 * rb);
    f64x4 acc[2][4];
    for (int rb = 0; rb < 2; ++rb)
        for (int cb = 0; cb < 4; ++cb)
            for (int i = 0; i < 4; ++i) acc[rb][cb][i] = 0.0;
    int rd_cb[4];
    for (int cb = 0; cb < 4; ++cb) {
        rd_cb[cb] = cb * 2048 + r * 128 + (((q >> 1) ^ ((r >> 1) & 7)) << 4) + ((q & 1) << 3);
        asm volatile("" : "+v"(rd_cb[cb]));
    }
    int iv = lane;
    double dv0 = 1.0 + lane * 1e-6, dv1 = 0.5;
    double breg[4] = { 1.0, 2.0, 3.0, 4.0 };
    for (int st = 0; st < steps; ++st) {
        if (BAR) __builtin_amdgcn_s_barrier();
        const char *slot = smem + (st & ring_mask) * 8192;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            double b[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                if (LDS == 0) b[cb] = breg[cb];
                if (LDS == 1) b[cb] = *reinterpret_cast<const double *>(smem + rd_cb[0] + cb * 2048 + s * 32);
                if (LDS == 2) b[cb] = *reinterpret_cast<const double *>(slot + (rd_cb[cb] ^ (s << 5)));
            }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    acc[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[rb][s], b[cb], acc[rb][cb], 0, 0, 0);
#pragma unroll
                    for (int k = 0; k < IVALU; ++k) asm volatile("v_xor_b32 %0, 0x55, %0" : "+v"(iv));
#pragma unroll
                    for (int k = 0; k < DVALU; ++k) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dv0) : "v"(dv1));
                }
        }
    }
    double sum = dv0 + iv;
    for (int rb = 0; rb < 2; ++rb)
        for (int cb = 0; cb < 4; ++cb)
            for (int i = 0; i < 4; ++i) sum += acc[rb][cb][i];
    out[blockIdx.x * 256 + tid] = sum;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// fp32 counterpart: v_mfma_f32_32x32x2_f32, 4 accumulators, operands in registers; IVALU integer VALU ops, EXPS v_exp_f32 and
// FMAS v_fma_f32 per MFMA
template <int IVALU, int EXPS, int FMAS>
__global__ __launch_bounds__(256, 2) void k_loop32(float *out, int steps) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int cb = 0; cb < 4; ++cb)
        for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
    float av[4], bv[4];
    for (int i = 0; i < 4; ++i) {
        av[i] = 1.f + 1e-3f * (lane + i);
        bv[i] = 0.5f + 1e-3f * i;
    }
    int iv = lane;
    float ev = 1e-3f * lane, fv = 1.0f + lane * 1e-6f;
    for (int st = 0; st < steps; ++st) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tt], bv[cb], acc[cb], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < IVALU; ++k) asm volatile("v_xor_b32 %0, 0x55, %0" : "+v"(iv));
#pragma unroll
                for (int k = 0; k < EXPS; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(ev));
#pragma unroll
                for (int k = 0; k < FMAS; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fv) : "v"(ev));
            }
    }
    float sum = ev + fv + iv;
    for (int cb = 0; cb < 4; ++cb)
        for (int i = 0; i < 16; ++i) sum += acc[cb][i];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// bf16 counterpart: v_mfma_f32_32x32x16_bf16 (the instruction a 3-way bf16 split of fp32 operands would run on)
template <int IVALU, int EXPS, int FMAS>
__global__ __launch_bounds__(256, 2) void k_loop_bf16(float *out, int steps) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int cb = 0; cb < 4; ++cb)
        for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
    bf16x8 av[4], bv[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            av[i][e] = static_cast<__bf16>(1.f + 1e-2f * ((lane + i + e) & 7));
            bv[i][e] = static_cast<__bf16>(0.5f + 1e-2f * ((i + e) & 7));
        }
    int iv = lane;
    float ev = 1e-3f * lane, fv = 1.0f + lane * 1e-6f;
    for (int st = 0; st < steps; ++st) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[tt], bv[cb], acc[cb], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < IVALU; ++k) asm volatile("v_xor_b32 %0, 0x55, %0" : "+v"(iv));
#pragma unroll
                for (int k = 0; k < EXPS; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(ev));
#pragma unroll
                for (int k = 0; k < FMAS; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fv) : "v"(ev));
            }
    }
    float sum = ev + fv + iv;
    for (int cb = 0; cb < 4; ++cb)
        for (int i = 0; i < 16; ++i) sum += acc[cb][i];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

// bf16 MFMAs fed from LDS the way the split tile kernel does: per k-step one ds_read_b128 per column block, feeding NPROD MFMAs each
// (row planes in registers), B fragments double buffered; BAR: one s_barrier per 4 k-steps
template <int NPROD, int BAR>
__global__ __launch_bounds__(256, 2) void k_loop_bf16_lds(float *out, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    for (int i = tid; i < 16384 / 4; i += 256) reinterpret_cast<float *>(smem)[i] = 1e-3f * (i & 63);
    __syncthreads();
    f32x16 acc[4];
    for (int cb = 0; cb < 4; ++cb)
        for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
    bf16x8 av[3][4];
    for (int p = 0; p < 3; ++p)
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 8; ++e) av[p][i][e] = static_cast<__bf16>(1.f + 1e-2f * ((lane + i + e + p) & 7));
    int rd_off[4];
    for (int mm = 0; mm < 4; ++mm) rd_off[mm] = r * 128 + (((2 * mm + h) ^ ((r >> 1) & 7)) << 4);
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 bcur[4];
    for (int cb = 0; cb < 4; ++cb) bcur[cb] = *reinterpret_cast<const f32x4 *>(smem + cb * 4096 + rd_off[0]);
    for (int st = 0; st < steps; ++st) {
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            f32x4 bnext[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) bnext[cb] = *reinterpret_cast<const f32x4 *>(smem + cb * 4096 + rd_off[(mm + 1) & 3]);
            if (BAR && mm == 2) __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int q = 0; q < NPROD; ++q)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[q][mm], __builtin_bit_cast(bf16x8, bcur[cb]), acc[cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) bcur[cb] = bnext[cb];
        }
    }
    float sum = 0.f;
    for (int cb = 0; cb < 4; ++cb)
        for (int i = 0; i < 16; ++i) sum += acc[cb][i];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <typename F>
static double time_ms(F &&launch) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        launch();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = std::min(best, (double) ms);
    }
    return best;
}

template <int LDS, int IVALU, int DVALU, int BAR>
static void run(const char *what, double *buf, int cus) {
    const int steps = 2000;
    for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = cus * wps;
        const size_t lds = 3 * 8192;
        const double ms = time_ms([&] { hipLaunchKernelGGL((k_loop<LDS, IVALU, DVALU, BAR>), dim3(blocks), dim3(256), lds, 0, buf, steps, 1); });
        const double flop = 2.0 * 16 * 16 * 4 * 32.0 * steps * (double) blocks * 4;
        printf("%-64s %d wave/SIMD: %6.1f TFLOP/s\n", what, wps, flop / ms / 1e9);
    }
}

template <int IVALU, int EXPS, int FMAS>
static void run32(const char *what, double *buf, int cus) {
    const int steps = 4000;
    for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = cus * wps;
        const double ms = time_ms([&] { hipLaunchKernelGGL((k_loop32<IVALU, EXPS, FMAS>), dim3(blocks), dim3(256), 0, 0, reinterpret_cast<float *>(buf), steps); });
        const double flop = 2.0 * 32 * 32 * 2 * 16.0 * steps * (double) blocks * 4;
        printf("%-64s %d wave/SIMD: %6.1f TFLOP/s\n", what, wps, flop / ms / 1e9);
    }
}

template <int IVALU, int EXPS, int FMAS>
static void run_bf16(const char *what, double *buf, int cus) {
    const int steps = 8000;
    for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = cus * wps;
        const double ms = time_ms([&] { hipLaunchKernelGGL((k_loop_bf16<IVALU, EXPS, FMAS>), dim3(blocks), dim3(256), 0, 0, reinterpret_cast<float *>(buf), steps); });
        const double flop = 2.0 * 32 * 32 * 16 * 16.0 * steps * (double) blocks * 4;
        printf("%-64s %d wave/SIMD: %7.1f TFLOP/s\n", what, wps, flop / ms / 1e9);
    }
}

template <int NPROD, int BAR>
static void run_bf16_lds(const char *what, double *buf, int cus) {
    const int steps = 4000;
    for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = cus * wps;
        const double ms = time_ms([&] { hipLaunchKernelGGL((k_loop_bf16_lds<NPROD, BAR>), dim3(blocks), dim3(256), 16384, 0, reinterpret_cast<float *>(buf), steps); });
        const double flop = 2.0 * 32 * 32 * 16 * (16.0 * NPROD) * steps * (double) blocks * 4;
        printf("%-64s %d wave/SIMD: %7.1f TFLOP/s\n", what, wps, flop / ms / 1e9);
    }
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs=%d\n", prop.gcnArchName, cus);
    double *buf;
    CHECK(hipMalloc(&buf, 64u << 20));
    run<0, 0, 0, 0>("f64 mfma, B in registers", buf, cus);
    run<1, 0, 0, 0>("f64 mfma, B by ds_read (immediate offsets)", buf, cus);
    run<2, 0, 0, 0>("f64 mfma, B by ds_read_b64 (xor + add per read)", buf, cus);
    run<2, 0, 0, 1>("f64 mfma, B by ds_read_b64 (xor + add), barrier per step", buf, cus);
    run<0, 1, 0, 0>("f64 mfma + 1 int VALU per MFMA", buf, cus);
    run<0, 2, 0, 0>("f64 mfma + 2 int VALU per MFMA", buf, cus);
    run<0, 4, 0, 0>("f64 mfma + 4 int VALU per MFMA", buf, cus);
    run<0, 8, 0, 0>("f64 mfma + 8 int VALU per MFMA", buf, cus);
    run<0, 0, 1, 0>("f64 mfma + 1 v_fma_f64 per MFMA", buf, cus);
    run<0, 0, 2, 0>("f64 mfma + 2 v_fma_f64 per MFMA", buf, cus);
    run<0, 0, 4, 0>("f64 mfma + 4 v_fma_f64 per MFMA", buf, cus);
    run32<0, 0, 0>("f32 mfma 32x32x2, operands in registers", buf, cus);
    run32<1, 0, 0>("f32 mfma + 1 int VALU per MFMA", buf, cus);
    run32<2, 0, 0>("f32 mfma + 2 int VALU per MFMA", buf, cus);
    run32<4, 0, 0>("f32 mfma + 4 int VALU per MFMA", buf, cus);
    run32<8, 0, 0>("f32 mfma + 8 int VALU per MFMA", buf, cus);
    run32<0, 1, 0>("f32 mfma + 1 v_exp_f32 per MFMA", buf, cus);
    run32<0, 1, 2>("f32 mfma + 1 v_exp_f32 + 2 v_fma_f32 per MFMA", buf, cus);
    run32<0, 0, 4>("f32 mfma + 4 v_fma_f32 per MFMA", buf, cus);
    run_bf16<0, 0, 0>("bf16 mfma 32x32x16, operands in registers", buf, cus);
    run_bf16<1, 0, 0>("bf16 mfma + 1 int VALU per MFMA", buf, cus);
    run_bf16<2, 0, 0>("bf16 mfma + 2 int VALU per MFMA", buf, cus);
    run_bf16<0, 1, 0>("bf16 mfma + 1 v_exp_f32 per MFMA", buf, cus);
    run_bf16<0, 1, 2>("bf16 mfma + 1 v_exp_f32 + 2 v_fma_f32 per MFMA", buf, cus);
    run_bf16<0, 0, 4>("bf16 mfma + 4 v_fma_f32 per MFMA", buf, cus);
    run_bf16_lds<3, 0>("bf16 mfma from LDS, 3 products per B fragment", buf, cus);
    run_bf16_lds<2, 0>("bf16 mfma from LDS, 2 products per B fragment", buf, cus);
    run_bf16_lds<1, 0>("bf16 mfma from LDS, 1 product per B fragment", buf, cus);
    run_bf16_lds<3, 1>("bf16 mfma from LDS, 3 products, barrier per 4 k-steps", buf, cus);
    run_bf16_lds<1, 1>("bf16 mfma from LDS, 1 product, barrier per 4 k-steps", buf, cus);
    return 0;
}