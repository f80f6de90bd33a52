// How fast can one CU issue 16-byte-per-lane global stores, by address shape?  (diagnostic for the GEMM epilogue, gemm.hip)
//   pattern 0: one wave-instruction = 16 rows x 64 contiguous bytes   (the register-direct epilogue: row stride = ld bytes)
//   pattern 1: one wave-instruction =  8 rows x 128 bytes
//   pattern 2: one wave-instruction =  4 rows x 256 bytes
//   pattern 3: one wave-instruction =  1 row  x 1024 bytes
// Every workgroup (WAVES waves) writes its own region once: bytes identical across patterns.  Build:
//   hipcc --offload-arch=gfx950 -O3 tools/store_patterns.hip -o /tmp/store_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int PAT>
__global__ __launch_bounds__(512) void store_kernel(float* out, int ld_floats, int rows_per_wg, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    constexpr int SEG = PAT == 0 ? 16 : PAT == 1 ? 32 : PAT == 2 ? 64 : 256;    // floats per row segment
    constexpr int RPI = 256 / SEG;                                                // rows per instruction
    const int lr = lane / (SEG / 4), lc = (lane % (SEG / 4)) * 4;
    float4 v = make_float4(lane, wave, blockIdx.x, 1.f);
    float* base = out + (size_t)blockIdx.x * rows_per_wg * ld_floats;
    // the workgroup's region: rows_per_wg rows x 256 floats wide; waves interleave over (row group, column segment)
    const int col_segs = 256 / SEG;
    const int n_inst = rows_per_wg / RPI * col_segs;                              // instructions to cover the region once
    for (int it = 0; it < iters; ++it)
        for (int q = wave; q < n_inst; q += nw) {
            const int rg = q / col_segs, cs = q % col_segs;
            float* p = base + (size_t)(rg * RPI + lr) * ld_floats + cs * SEG + lc;
            *(float4*)p = v;
        }
}

int main() {
    const int wgs = 256, rows_per_wg = 1024, ld = 768;                           // 1024 x 256 floats = 1 MiB per workgroup and pass
    float* d;
    hipMalloc(&d, (size_t)wgs * rows_per_wg * ld * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 4; waves <= 8; waves += 4)
        for (int pat = 0; pat < 4; ++pat) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                const int iters = 4;
                if (pat == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(wgs), dim3(waves * 64), 0, 0, d, ld, rows_per_wg, iters);
                if (pat == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(wgs), dim3(waves * 64), 0, 0, d, ld, rows_per_wg, iters);
                if (pat == 2) hipLaunchKernelGGL(store_kernel<2>, dim3(wgs), dim3(waves * 64), 0, 0, d, ld, rows_per_wg, iters);
                if (pat == 3) hipLaunchKernelGGL(store_kernel<3>, dim3(wgs), dim3(waves * 64), 0, 0, d, ld, rows_per_wg, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double bytes = (double)wgs * rows_per_wg * 256 * 4 * 4;
            printf("waves/CU %d pattern %d: %.1f us  %.2f TB/s  %.1f B/clk/CU at 2.1 GHz\n", waves, pat, best * 1e3, bytes / (best * 1e-3) / 1e12,
                   bytes / 256 / (best * 1e-3) / 2.1e9);
        }
    // per-CU ceiling: the same stores from FEWER workgroups (the rest of the chip quiet), pattern 0, 8 waves.  In a GEMM only the
    // workgroups that are in their epilogue store; if one CU alone cannot go faster than its share of 7 TB/s, the epilogue of a
    // one-workgroup-per-CU kernel costs tile bytes / that rate whatever the others do.
    for (int n : {8, 32, 64, 128, 256}) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(store_kernel<0>, dim3(n), dim3(512), 0, 0, d, ld, rows_per_wg, 8);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double bytes = (double)n * rows_per_wg * 256 * 4 * 8;
        printf("%3d workgroups (1 per CU), 8 waves, pattern 0: %.1f us  %.2f TB/s  %.1f B/clk per active CU at 2.1 GHz\n", n, best * 1e3,
               bytes / (best * 1e-3) / 1e12, bytes / n / (best * 1e-3) / 2.1e9);
    }
    return 0;
}
