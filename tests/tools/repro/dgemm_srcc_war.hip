/*
 * dgemm_srcc_war.hip -- stand-alone reproducer of the hazard behind the wrong instantiation of tile_matvec_f64_wide<KT_POLY, true> (DESIGN.md
 * section 4.1, round 4): on gfx950 an LDS load may overwrite a VGPR that an IN-FLIGHT v_mfma_f64_16x16x4_f64 still has to read as its C operand.
 *
 * The compiler (ROCm 7.2, -O2 / -O3) emitted, for the peeled first feature panel of that kernel:
 *       v_mov_b64 v[66:67] .. v[72:73], coef0          ; one register quad = the start value of ALL eight accumulators
 *       v_mfma_f64_16x16x4_f64 v[58:65], A, B, v[66:73]
 *       ... (eight MFMAs, every one with C = v[66:73], D = its own registers)
 *       v_mfma_f64_16x16x4_f64 v[2:9],   A, B, v[66:73]
 *       ds_read_b64 v[66:67], ...                       ; the next k-step's B fragments go INTO the dead C quad
 *       ds_read_b64 v[72:73], ...
 * v_mfma_f64_16x16x4_f64 occupies the matrix pipe for 64 cycles and reads the rows of C pass by pass; an LDS read returns after ~50-64 cycles.  The
 * last rows of C of the last MFMA are read AFTER the LDS data has landed in v[72:73]: the accumulator starts from a B-fragment value instead of
 * coef0.  The hazard recognizer inserts nothing between the two (no rule "MFMA reads SrcC -> LDS/VMEM load writes it" for gfx90a+), and no
 * hardware interlock holds the load's write-back.
 *
 * This file issues exactly that sequence from inline asm with NOPS wait states between the last MFMA and the LDS reads and counts the accumulator
 * elements that come out wrong (A = B = 0, C = 1.0, LDS holds 1000.0: every element must be 1.0).
 * build: hipcc --offload-arch=gfx950 -O2 dgemm_srcc_war.hip -o dgemm_srcc_war ; run: ./dgemm_srcc_war
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int NOPS>
__global__ __launch_bounds__(256, 2) void k_war(double *out, int iters) {
    __shared__ double lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = 1000.0;
    __syncthreads();
    const unsigned addr = static_cast<unsigned>(reinterpret_cast<size_t>(lds)) + 8u * (threadIdx.x & 63);
    double bad = 0.0;
    for (int it = 0; it < iters; ++it) {
        double r0, r1, r2, r3;
        asm volatile(
            "v_mov_b64 v[66:67], 1.0\n\tv_mov_b64 v[68:69], 1.0\n\tv_mov_b64 v[70:71], 1.0\n\tv_mov_b64 v[72:73], 1.0\n\t"
            "v_mov_b64 v[74:75], 0\n\tv_mov_b64 v[76:77], 0\n\t"
            "s_nop 7\n\t"
            "v_mfma_f64_16x16x4_f64 v[58:65], v[74:75], v[76:77], v[66:73]\n\t"
            "v_mfma_f64_16x16x4_f64 v[50:57], v[74:75], v[76:77], v[66:73]\n\t"
            "v_mfma_f64_16x16x4_f64 v[42:49], v[74:75], v[76:77], v[66:73]\n\t"
            "v_mfma_f64_16x16x4_f64 v[34:41], v[74:75], v[76:77], v[66:73]\n\t"
            "v_mfma_f64_16x16x4_f64 v[26:33], v[74:75], v[76:77], v[66:73]\n\t"
            "v_mfma_f64_16x16x4_f64 v[18:25], v[74:75], v[76:77], v[66:73]\n\t"
            "v_mfma_f64_16x16x4_f64 v[10:17], v[74:75], v[76:77], v[66:73]\n\t"
            "v_mfma_f64_16x16x4_f64 v[2:9], v[74:75], v[76:77], v[66:73]\n\t"
            ".rept %5\n\ts_nop 0\n\t.endr\n\t"
            "ds_read_b64 v[66:67], %4\n\t"
            "ds_read_b64 v[68:69], %4 offset:512\n\t"
            "ds_read_b64 v[70:71], %4 offset:1024\n\t"
            "ds_read_b64 v[72:73], %4 offset:1536\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            "v_mov_b64 %0, v[2:3]\n\tv_mov_b64 %1, v[4:5]\n\tv_mov_b64 %2, v[6:7]\n\tv_mov_b64 %3, v[8:9]"
            : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
            : "v"(addr), "i"(NOPS)
            : "memory", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25",
              "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49",
              "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73",
              "v74", "v75", "v76", "v77");
        bad += (r0 != 1.0) + (r1 != 1.0) + (r2 != 1.0) * 1.0 + (r3 != 1.0) * 1000.0;  // thousands = element 3 (the LAST rows of C) of the LAST MFMA
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = bad;
}

/* the same question for the 16-bit MFMA of the fp32 split kernels: v_mfma_f32_16x16x32_f16 (16 cycles), C = a register quad, followed by a
 * ds_read_b128 into that quad */
template <int NOPS>
__global__ __launch_bounds__(256, 2) void k_war_f16(double *out, int iters) {
    __shared__ float lds[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = 1000.0f;
    __syncthreads();
    const unsigned addr = static_cast<unsigned>(reinterpret_cast<size_t>(lds)) + 16u * (threadIdx.x & 63);
    double bad = 0.0;
    for (int it = 0; it < iters; ++it) {
        float r0, r1, r2, r3;
        asm volatile(
            "v_mov_b32 v66, 1.0\n\tv_mov_b32 v67, 1.0\n\tv_mov_b32 v68, 1.0\n\tv_mov_b32 v69, 1.0\n\t"
            "v_mov_b32 v74, 0\n\tv_mov_b32 v75, 0\n\tv_mov_b32 v76, 0\n\tv_mov_b32 v77, 0\n\t"
            "s_nop 7\n\t"
            "v_mfma_f32_16x16x32_f16 v[58:61], v[74:77], v[74:77], v[66:69]\n\t"
            "v_mfma_f32_16x16x32_f16 v[50:53], v[74:77], v[74:77], v[66:69]\n\t"
            "v_mfma_f32_16x16x32_f16 v[42:45], v[74:77], v[74:77], v[66:69]\n\t"
            "v_mfma_f32_16x16x32_f16 v[34:37], v[74:77], v[74:77], v[66:69]\n\t"
            "v_mfma_f32_16x16x32_f16 v[26:29], v[74:77], v[74:77], v[66:69]\n\t"
            "v_mfma_f32_16x16x32_f16 v[18:21], v[74:77], v[74:77], v[66:69]\n\t"
            "v_mfma_f32_16x16x32_f16 v[10:13], v[74:77], v[74:77], v[66:69]\n\t"
            "v_mfma_f32_16x16x32_f16 v[2:5], v[74:77], v[74:77], v[66:69]\n\t"
            ".rept %5\n\ts_nop 0\n\t.endr\n\t"
            "ds_read_b128 v[66:69], %4\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_nop 15\n\ts_nop 15\n\t"
            "v_mov_b32 %0, v2\n\tv_mov_b32 %1, v3\n\tv_mov_b32 %2, v4\n\tv_mov_b32 %3, v5"
            : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
            : "v"(addr), "i"(NOPS)
            : "memory", "v2", "v3", "v4", "v5", "v10", "v11", "v12", "v13", "v18", "v19", "v20", "v21", "v26", "v27", "v28", "v29", "v34", "v35", "v36", "v37", "v42", "v43", "v44", "v45",
              "v50", "v51", "v52", "v53", "v58", "v59", "v60", "v61", "v66", "v67", "v68", "v69", "v74", "v75", "v76", "v77");
        bad += (r0 != 1.0f) + (r1 != 1.0f) + (r2 != 1.0f) * 1.0 + (r3 != 1.0f) * 1000.0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = bad;
}

template <int NOPS>
static void run(int blocks, int iters) {
    double *d = nullptr;
    const size_t n = static_cast<size_t>(blocks) * 256;
    (void) hipMalloc(reinterpret_cast<void **>(&d), n * sizeof(double));
    hipLaunchKernelGGL(k_war<NOPS>, dim3(blocks), dim3(256), 0, nullptr, d, iters);
    std::vector<double> h(n);
    (void) hipMemcpy(h.data(), d, n * sizeof(double), hipMemcpyDeviceToHost);
    double last = 0.0, others = 0.0;
    size_t lanes = 0;
    for (double v : h) {
        last += static_cast<long>(v) / 1000;
        others += static_cast<long>(v) % 1000;
        lanes += v != 0.0;
    }
    std::printf("v_mfma_f64_16x16x4_f64  -> ds_read_b64  into SrcC, %3d wait states between: wrong accumulator elements per lane and iteration: last rows of C %.4f, other rows %.4f (%zu of %zu lanes affected)\n", NOPS,
                last / (static_cast<double>(n) * iters), others / (static_cast<double>(n) * iters), lanes, n);
    (void) hipFree(d);
}

template <int NOPS>
static void run_f16(int blocks, int iters) {
    double *d = nullptr;
    const size_t n = static_cast<size_t>(blocks) * 256;
    (void) hipMalloc(reinterpret_cast<void **>(&d), n * sizeof(double));
    hipLaunchKernelGGL(k_war_f16<NOPS>, dim3(blocks), dim3(256), 0, nullptr, d, iters);
    std::vector<double> h(n);
    (void) hipMemcpy(h.data(), d, n * sizeof(double), hipMemcpyDeviceToHost);
    double last = 0.0, others = 0.0;
    size_t lanes = 0;
    for (double v : h) {
        last += static_cast<long>(v) / 1000;
        others += static_cast<long>(v) % 1000;
        lanes += v != 0.0;
    }
    std::printf("v_mfma_f32_16x16x32_f16 -> ds_read_b128 into SrcC, %3d wait states between: wrong accumulator elements per lane and iteration: last rows of C %.4f, other rows %.4f (%zu of %zu lanes affected)\n",
                NOPS, last / (static_cast<double>(n) * iters), others / (static_cast<double>(n) * iters), lanes, n);
    (void) hipFree(d);
}

int main() {
    const int blocks = 2048, iters = 200;  // two workgroups per CU: two waves per SIMD compete for the matrix pipe, as in the tile kernel
    run<0>(blocks, iters);
    run<1>(blocks, iters);
    run<2>(blocks, iters);
    run<3>(blocks, iters);
    run<4>(blocks, iters);
    run<8>(blocks, iters);
    run<12>(blocks, iters);
    run<16>(blocks, iters);
    run<24>(blocks, iters);
    run<32>(blocks, iters);
    run<48>(blocks, iters);
    run<64>(blocks, iters);
    run_f16<0>(blocks, iters);
    run_f16<1>(blocks, iters);
    run_f16<2>(blocks, iters);
    run_f16<4>(blocks, iters);
    run_f16<8>(blocks, iters);
    return 0;
}
