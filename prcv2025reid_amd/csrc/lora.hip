// MER-LoRA weight merge for gfx950 (see include/reid_hip.h: reid_merge_lora_table).
//
// MERLinear (mer_lora.py:80-99) is  y = W x + (alpha/r) B_mu A_mu x  with one adapter pair per modality mu.  A batch is packed
// modality by modality, so every row tile of the big GEMMs belongs to ONE modality and the low-rank update can live inside the
// weight:  W_eff[mu] = W + (alpha/r) B_mu A_mu, one 16-bit [N, K] matrix per modality (and its transpose for the dX products).
// 288 GB of HBM make the 4 x copy of the backbone linears (1.36 GB for ViT-B/16 with both orientations) a non-issue, and it
// takes the rank-r side computation (x A^T before every GEMM, dY B before every dX GEMM) off the critical path: the main stream
// runs plain GEMMs, the adapter gradients are formed on a side stream from tensors the main stream produces anyway.
// The merge is HBM-bound (reads W in fp32 once, writes 2 x nmod 16-bit copies) and runs once per optimizer step.
//
// Rounding: W_eff is rounded ONCE from the fp32 sum, so the error of the merged operand is one 16-bit rounding of (W + delta) --
// the same size as the rounding of W alone that the unmerged form carries (plus its own roundings of T and B).  An update much
// smaller than one ulp of W moves a fraction |delta| / ulp of the rounded elements by one ulp: unbiased, like the fp32-master /
// 16-bit-compute weights of ordinary mixed-precision training.
#include "common.h"

namespace {

constexpr int MT = 64;   // tile edge (output features x input features)

// table entry (8 x int64): W pointer (fp32 [N, K], row stride K), arena offsets of Acat [G*Rp, K] and Bcat [N, Rp], destination
// offsets (16-bit elements) of W_eff [nmod][N][K] and W_eff^T [nmod][K][N] (negative: skip), N, K, G
__global__ __launch_bounds__(256) void merge_lora_kernel(const int64_t* __restrict__ table, const float* __restrict__ arena,
                                                         bf16_t* __restrict__ weff, int Rp, int r, int nmod, float s) {
    REID_T16_ENTER();
    const int64_t* e = table + (size_t)blockIdx.y * 8;
    const int N = (int)e[5], K = (int)e[6], G = (int)e[7];
    const int tiles_k = K / MT;
    if ((int)blockIdx.x >= (N / MT) * tiles_k) return;
    const float* __restrict__ W = (const float*)e[0];
    const float* __restrict__ A = arena + e[1];
    const float* __restrict__ B = arena + e[2];
    bf16_t* dst = weff + e[3];
    bf16_t* dstT = e[4] >= 0 ? weff + e[4] : nullptr;
    const int n0 = ((int)blockIdx.x / tiles_k) * MT, k0 = ((int)blockIdx.x % tiles_k) * MT;
    const int g = n0 / (N / G);                          // projection group of the fused q|k|v weight (its own adapter set)
    const int R = nmod * r;                              // adapter rows in use (<= 64)
    // LDS images padded to 16-byte-aligned rows: a thread reads 8 consecutive k of an adapter row (pass 1) or 8 consecutive n of an
    // adapter column (pass 2) as two 16-byte reads, so a rank-r update of 8 outputs costs 3 LDS instructions per j instead of 16
    // (the first version, scalar reads: LDS-bound at 2 TB/s of HBM traffic).
    __shared__ __attribute__((aligned(16))) float sW[MT][MT + 4];
    __shared__ __attribute__((aligned(16))) float sA[MT][MT + 4];    // [adapter row][k]
    __shared__ __attribute__((aligned(16))) float sBt[MT][MT + 4];   // [adapter column][n]
    const int t = threadIdx.x;
    for (int i = t; i < MT * (MT / 4); i += 256) {       // 16-byte loads along k
        const int row = i / (MT / 4), c4 = (i % (MT / 4)) * 4;
        *(f32x4*)&sW[row][c4] = *(const f32x4*)(W + (size_t)(n0 + row) * K + k0 + c4);
        if (row < R) *(f32x4*)&sA[row][c4] = *(const f32x4*)(A + (size_t)(g * Rp + row) * K + k0 + c4);
    }
    for (int i = t; i < MT * R; i += 256) {
        const int row = i / R, c = i % R;
        sBt[c][row] = B[(size_t)(n0 + row) * Rp + c];
    }
    __syncthreads();
    const int c8 = (t & 7) * 8, r0 = t >> 3;             // a thread owns 8 consecutive elements of rows r0 and r0 + 32
    // Both passes evaluate element (n, k) of modality mu by the SAME chain: acc = fma(B[n][mu r + j], A[mu r + j][k], acc) for j = 0..r-1,
    // then fma(s, acc, W[n][k]) -- so W_eff^T is the exact transpose of W_eff.
    for (int mu = 0; mu < nmod; ++mu) {
        bf16_t* d = dst + (size_t)mu * N * K;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = r0 + 32 * h;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < r; ++j) {
                const float b = sBt[mu * r + j][n];
                const f32x4 a0 = *(const f32x4*)&sA[mu * r + j][c8], a1 = *(const f32x4*)&sA[mu * r + j][c8 + 4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { acc[q] = fmaf(b, a0[q], acc[q]); acc[4 + q] = fmaf(b, a1[q], acc[4 + q]); }
            }
            const f32x4 w0 = *(const f32x4*)&sW[n][c8], w1 = *(const f32x4*)&sW[n][c8 + 4];
            float v[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] = fmaf(s, acc[q], w0[q]); v[4 + q] = fmaf(s, acc[4 + q], w1[q]); }
            *(uint4*)(d + (size_t)(n0 + n) * K + k0 + c8) =
                uint4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        }
        if (dstT) {
            bf16_t* dt = dstT + (size_t)mu * N * K;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = r0 + 32 * h;
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int j = 0; j < r; ++j) {
                    const float a = sA[mu * r + j][k];
                    const f32x4 b0 = *(const f32x4*)&sBt[mu * r + j][c8], b1 = *(const f32x4*)&sBt[mu * r + j][c8 + 4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { acc[q] = fmaf(b0[q], a, acc[q]); acc[4 + q] = fmaf(b1[q], a, acc[4 + q]); }
                }
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = fmaf(s, acc[q], sW[c8 + q][k]);
                *(uint4*)(dt + (size_t)(k0 + k) * N + n0 + c8) =
                    uint4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            }
        }
    }
}

}  // namespace

extern "C" int reid_merge_lora_table(const int64_t* table, int32_t n_entries, int32_t max_tiles, const float* arena, void* weff,
                                     int32_t Rp, int32_t r, int32_t nmod, float scaling, void* stream) {
    REID_CHECK_ARG(table && arena && weff, "reid_merge_lora_table: null pointer");
    REID_CHECK_ARG(n_entries > 0 && n_entries <= 65535 && max_tiles > 0, "reid_merge_lora_table: n_entries=%d max_tiles=%d", n_entries, max_tiles);
    REID_CHECK_ARG(r > 0 && nmod > 0 && nmod * r <= MT && nmod * r <= Rp, "reid_merge_lora_table: nmod*r = %d must be <= %d and <= Rp = %d", nmod * r, MT, Rp);
    hipLaunchKernelGGL(merge_lora_kernel, dim3(max_tiles, n_entries), dim3(256), 0, (hipStream_t)stream, table, arena, (bf16_t*)weff, Rp, r, nmod,
                       scaling);
    REID_CHECK_LAUNCH("reid_merge_lora_table");
    return REID_OK;
}
