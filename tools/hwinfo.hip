// Where does the dispatcher put the workgroups of a 2-per-CU grid?  Prints, per block: XCC id, HW_ID (CU / SE / SIMD
// fields), the raw HW_REG_LDS_ALLOC word and the start time.  Build: hipcc --offload-arch=gfx950 -O2 tools/hwinfo.hip -o /tmp/hwinfo
// (diagnostic for the phase-staggered persistent GEMM of prcv2025reid_amd/csrc/gemm.hip; not part of the library).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__global__ __launch_bounds__(256, 2) void probe(uint32_t* out, int hold_us) {
    extern __shared__ char smem[];
    const uint32_t hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    const uint32_t ldsa = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11));
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    smem[threadIdx.x] = (char)hwid;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)hold_us * 100) __builtin_amdgcn_s_sleep(32);   // keep every block resident
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = hwid;
        out[blockIdx.x * 4 + 1] = ldsa;
        out[blockIdx.x * 4 + 2] = xcc;
        out[blockIdx.x * 4 + 3] = (uint32_t)t0;
    }
}

int main() {
    const int grid = 512;
    uint32_t* d;
    hipMalloc(&d, grid * 16);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 65536, 0, d, 200);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(grid * 4);
    hipMemcpy(h.data(), d, grid * 16, hipMemcpyDeviceToHost);
    // gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    int second = 0;
    for (int b = 0; b < grid; ++b) {
        const uint32_t hw = h[b * 4], la = h[b * 4 + 1], xc = h[b * 4 + 2];
        if ((la & 0xfff) != 0) ++second;
        if (b < 48 || b % 64 == 0)
            printf("block %3d: xcc %u  hw_id %08x (cu %u sh %u se %u)  lds_alloc %08x (base %u)  t0 %u\n", b, xc & 0xf, hw, (hw >> 8) & 0xf,
                   (hw >> 12) & 1, (hw >> 13) & 7, la, la & 0xfff, h[b * 4 + 3]);
    }
    printf("blocks with a non-zero LDS base: %d of %d\n", second, grid);
    // pairs: which block shares (xcc, hw_id cu/sh/se) with block b?
    int shown = 0;
    for (int b = 0; b < grid && shown < 16; ++b)
        for (int c = b + 1; c < grid; ++c)
            if ((h[b * 4 + 2] & 0xf) == (h[c * 4 + 2] & 0xf) && ((h[b * 4] >> 8) & 0xff) == ((h[c * 4] >> 8) & 0xff)) {
                printf("blocks %d and %d share a CU (bases %u, %u)\n", b, c, h[b * 4 + 1] & 0xfff, h[c * 4 + 1] & 0xfff);
                ++shown;
                break;
            }
    return 0;
}
