// What does the matrix pipe sustain when nothing else is in the way?  Every CU runs 8 waves (2 per SIMD) of back-to-back
// v_mfma_f32_16x16x32_bf16 on registers only -- the ceiling any GEMM K loop on this chip is measured against (the 2.5 PF
// headline is at the 2.4 GHz boost clock; under a full matrix load the chip's power management sets the clock).
// Also prints the shader clock during the run: s_memtime (shader clock) against s_memrealtime (100 MHz).
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_ceiling.hip -o /tmp/mfma_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ __launch_bounds__(512, 2) void mfma_loop(float* out, uint64_t* clk, int iters) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
    const uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.f) out[0] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* d; uint64_t* dc;
    hipMalloc(&d, 4); hipMalloc(&dc, cus * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        const int iters = rep < 2 ? 2000 : 20000;                 // ~1 ms and ~10 ms: does the clock settle lower on a long run?
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_loop<16>, dim3(cus), dim3(512), 0, 0, d, dc, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> h(cus * 2);
        hipMemcpy(h.data(), dc, cus * 16, hipMemcpyDeviceToHost);
        double cyc = 0, real = 0;
        for (int i = 0; i < cus; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
        const double flops = 2.0 * 16 * 16 * 32 * 16.0 * iters * 8 * cus;
        printf("iters %6d: %.3f ms  %.1f TFLOP/s  (%d CUs x 8 waves)  s_memtime/s_memrealtime = %.3f -> shader clock %.0f MHz; MFMA issue = %.2f cycles each\n",
               iters, ms, flops / (ms * 1e-3) / 1e12, cus, cyc / real, cyc / real * 100.0, (cyc / cus) / (16.0 * iters * 2));
    }
    return 0;
}
