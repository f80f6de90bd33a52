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
#include <type_traits>

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

// ---------------------------------------------------------------- adapter gradients of one linear: U = dY.B and dB = dY^T.T in ONE pass over dY
// r03's side stream read every dY twice: the skinny GEMM U = mask(dY . B) * (alpha/r) (the rank-r cotangent dA = U^T x needs) and the
// row-reduction dB += dY^T . T.  Leaving the first out takes 1.1 ms off the 32.9 ms step (tools/exp_skip_u.sh: an upper bound), so
// both products are formed from one staged tile here.  Splitting dY by COLUMN panels (as gemm_tn does) would leave U as N/128 partial
// sums per row (fp32 atomics: more bytes than the re-read saves); so a workgroup owns a ROW slab and all N = 768 columns: its eight
// waves own 96 columns each, keep their [96, 32] slice of dB in registers for the whole slab (48 VGPRs) and flush it once, and the
// eight partial U tiles of a 32-row step are summed through LDS.  N = 3072 (fc1: 393 KB of accumulators, 3/4 of a CU's register
// file) stays on the two-launch path.
namespace fused {
constexpr int N = 768, RP = 32, R = 32;                  // columns of dY, adapter columns, rows per step
constexpr int TP = RP * 2 + 32;                           // LDS row pitch of the T tile (bytes): padded for the transpose reads (gemm_tn.hip)
// NW waves per workgroup, each owning N / NW columns.  NW = 8 (96 columns, 113 KB of LDS: one workgroup per CU) is what runs: every
// 32-row step is one exposed HBM round trip (~4.5 us in the train step = 14 GB/s per CU).  NW = 4 (192 columns, 80 KB: TWO workgroups
// per CU, each waiting for its own loads) compiles to 256 VGPRs + 40 spilled (96 accumulator + 48 B^T-fragment + 56 staging registers):
// not instantiated; the staging registers would have to go (LDS-DMA) first.
template <int NW> struct Geo {
    static constexpr int CPW = N / NW, MT = CPW / 16, KS = CPW / 32, CH = CPW / 8, XL = R * CH / 64;
    static constexpr int XP = CPW * 2 + 32;
    static constexpr int WAVE_LDS = R * XP + R * TP;
    static constexpr int LDS_BYTES = NW * WAVE_LDS + NW * R * RP * 4;
};

typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((address_space(3))) s4* lds_s4_ptr;
typedef float f32x2v __attribute__((ext_vector_type(2)));

struct Params {
    const bf16_t* dY; const bf16_t* T; const bf16_t* BT; bf16_t* U; float* dB;
    const int32_t* img_mod;
    int lddy, ldt, ldbt, ldu, lddb, rows_per_img, mask_r, M, slab_rows;
    float scale;
    float* Up;                                            // fp32 [M, 32] partial sums of U over column blocks of a wider dY (or null)
    int u_mode;                                           // bit 0: add Up to this block's sum; bit 1: store the sum to Up instead of U
};

// gemm_tn.hip's transposed fragment: 16x16x32 operand whose 16 MFMA rows are image COLUMNS col0.. and whose k are image rows k0..k0+31
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int pitch, int k0, int col0, int lane) {
    const int l16 = lane & 15, fq = lane >> 4;
    const int q = l16 >> 2, pp = l16 & 3;
    const char* a0 = img + (k0 + 4 * fq + q) * pitch + (col0 + 4 * pp) * 2;
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(a0));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(a0 + 16 * pitch));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void lora_bwd_fused_kernel(const Params p) {
    REID_T16_ENTER();
    using G = Geo<NW>;
    constexpr int CPW = G::CPW, MT = G::MT, KS = G::KS, CH = G::CH, XL = G::XL, XP = G::XP, WAVE_LDS = G::WAVE_LDS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, fq = lane >> 4;
    char* xs = smem + w * WAVE_LDS;                       // this wave's [32 rows][CPW columns] of dY
    char* ts = xs + R * XP;                               // its copy of T[32 rows][32]
    float* ured = (float*)(smem + NW * WAVE_LDS);         // [NW waves][32 rows][32 columns] partial U
    const int col0 = w * CPW;
    const int mbeg = blockIdx.x * p.slab_rows;
    const int mend = min(p.M, mbeg + p.slab_rows);
    if (mbeg >= mend) return;                             // (whole workgroup)
    const int steps = (mend - mbeg + R - 1) / R;
    // B^T fragments of this wave's columns (B operand of U = dY . B: n = adapter column, k = dY column): constant over the slab
    bf16x8 bfr[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) bfr[ks][j] = *(const bf16x8*)(p.BT + (size_t)(j * 16 + l16) * p.ldbt + col0 + ks * 32 + 8 * fq);
    f32x4 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = acc[i][0]; }
#ifndef REID_FUSED_DEPTH
#define REID_FUSED_DEPTH 1
#endif
    // DEPTH steps of operands in flight in registers (1 or 2).  The slot of a step is a compile-time constant (the loop is unrolled by
    // DEPTH): indexing the register arrays with (t & 1) put them into scratch memory (80 bytes per lane: 35.0 ms per step).  Two steps
    // in flight (226 VGPRs) change nothing: 32.61 / 32.73 against 32.88 / 32.71 ms per step on one box (profiles/r03_lora_fused_depth.log)
    // -- the step is not what waits for HBM; one step in flight (194 VGPRs) stays.
    constexpr int DEPTH = REID_FUSED_DEPTH;
    uint4 xr[DEPTH][XL], tr[DEPTH][2];
    auto gload = [&](int t, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        const int mb = mbeg + t * R;
#pragma unroll
        for (int i = 0; i < XL; ++i) {                    // 32 rows x CH chunks of 16 bytes
            const int c = lane + 64 * i, row = c / CH, ch = c % CH;
            const int m = mb + row;
            xr[slot][i] = m < mend ? *(const uint4*)(p.dY + (size_t)m * p.lddy + col0 + ch * 8) : uint4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {                     // 32 rows x 4 chunks
            const int c = lane + 64 * i, row = c >> 2, ch = c & 3;
            const int m = mb + row;
            tr[slot][i] = m < mend ? *(const uint4*)(p.T + (size_t)m * p.ldt + ch * 8) : uint4{0u, 0u, 0u, 0u};
        }
    };
    auto step = [&](int t, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
#pragma unroll
        for (int i = 0; i < XL; ++i) { const int c = lane + 64 * i; *(uint4*)(xs + (c / CH) * XP + (c % CH) * 16) = xr[slot][i]; }
#pragma unroll
        for (int i = 0; i < 2; ++i) { const int c = lane + 64 * i; *(uint4*)(ts + (c >> 2) * TP + (c & 3) * 16) = tr[slot][i]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (t + DEPTH < steps) gload(t + DEPTH, slot_c);
        // dB[CPW columns of dY, 32] += dY^T . T over the 32 rows of this step
        const bf16x8 tf0 = tr_frag(ts, TP, 0, 0, lane), tf1 = tr_frag(ts, TP, 0, 16, lane);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const bf16x8 xf = tr_frag(xs, XP, 0, i * 16, lane);
            acc[i][0] = mfma16(xf, tf0, acc[i][0]);
            acc[i][1] = mfma16(xf, tf1, acc[i][1]);
        }
        // partial U[32 rows, 32] over this wave's columns
        f32x4 ua[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) { ua[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ua[mt][1] = ua[mt][0]; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const bf16x8 af = *(const bf16x8*)(xs + (mt * 16 + l16) * XP + (ks * 32 + 8 * fq) * 2);
                ua[mt][0] = mfma16(af, bfr[ks][0], ua[mt][0]);
                ua[mt][1] = mfma16(af, bfr[ks][1], ua[mt][1]);
            }
        float* ur = ured + w * (R * RP);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) ur[(mt * 16 + 4 * fq + e) * RP + j * 16 + l16] = ua[mt][j][e];
        __syncthreads();
#pragma unroll
        for (int o = tid * 2; o < R * RP; o += NW * 128) {
            const int row = o >> 5, col = o & 31;
            const int m = mbeg + t * R + row;
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) { const f32x2v u2 = *(const f32x2v*)(ured + ww * (R * RP) + o); v0 += u2[0]; v1 += u2[1]; }
            if (m < mend && (p.u_mode & 1)) {              // earlier column blocks of the same rows (launches of one stream: no atomics)
                const f32x2v u2 = *(const f32x2v*)(p.Up + (size_t)m * RP + col);
                v0 += u2[0]; v1 += u2[1];
            }
            if (m < mend && (p.u_mode & 2)) {
                *(f32x2v*)(p.Up + (size_t)m * RP + col) = f32x2v{v0, v1};
            } else if (m < mend) {
                const int modality = p.img_mod[m / p.rows_per_img];
                v0 = (col / p.mask_r == modality) ? v0 * p.scale : 0.f;
                v1 = ((col + 1) / p.mask_r == modality) ? v1 * p.scale : 0.f;
                *(uint32_t*)(p.U + (size_t)m * p.ldu + col) = pack_bf16x2(v0, v1);
            }
        }
        __syncthreads();
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, DEPTH - 1>;
    gload(0, S0{});
    if (DEPTH == 2 && steps > 1) gload(1, S1{});
    for (int t = 0; t < steps; t += DEPTH) {
        step(t, S0{});
        if (DEPTH == 2 && t + 1 < steps) step(t + 1, S1{});
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                atomicAdd(p.dB + (size_t)(col0 + i * 16 + 4 * fq + e) * p.lddb + j * 16 + l16, acc[i][j][e]);
}
}  // namespace fused

// ---------------------------------------------------------------- r04: one IMAGE per workgroup, compact adapters, LDS-DMA staging
// The slab kernel above lasts 54 us alone (64 workgroups x 25 steps, two barriers and an 8-way LDS reduction per step, operands staged
// through 56 VGPRs) and ~125 us inside the training step, where every microsecond of a side-stream kernel is a microsecond a persistent
// GEMM workgroup starts late.  This form is bound by the HBM stream and nothing else, on EVERY CU at once:
//   * a workgroup owns one image (rows_per_img rows: one modality mu), so only a 16-column WINDOW of the adapter columns (the r columns of
//     mu and their neighbours inside an aligned group of 16) takes part: dB is [768, 16] per workgroup = 8 accumulator tiles in six waves
//     (flush: 768 x r fp32 atomics per image instead of 768 x 32 per slab), U is two 16 x 16 tiles per 32-row step;
//   * the dY tile of a step (32 rows x 768 columns, 48 KiB) and the T tile land by LDS-DMA in a source-swizzled image (no staging
//     registers); two buffers, ONE barrier per step; waves 0-1 form U (each a whole K = 768 contraction: no cross-wave reduction), waves
//     2-7 form dB from transposed reads of the same image.
// Contract difference to the slab kernel: T must be modality-masked (zero outside the columns of the row's modality), which is how the
// forward produces it -- columns outside the window are neither read nor accumulated.
namespace image {
constexpr int N = fused::N, RP = fused::RP, R = 32;
constexpr int ROW_BYTES = N * 2;                           // 1536: 96 chunks of 16 bytes
constexpr int IMG_BYTES = R * ROW_BYTES;                   // 48 KiB
constexpr int T_BYTES = R * RP * 2;                        // 2 KiB
constexpr int BUF_BYTES = IMG_BYTES + T_BYTES;
constexpr int LDS_BYTES = 2 * BUF_BYTES;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
using fused::s4; using fused::lds_s4_ptr; using fused::Params;

// transposed operand from the swizzled image: 16 MFMA rows = image columns col0 .. col0 + 15, k = image rows 0 .. 31
__device__ __forceinline__ bf16x8 tr_frag_swz(const char* img, int col0, int lane) {
    const int l16 = lane & 15, fq = lane >> 4;
    const int q = l16 >> 2, pp = l16 & 3;
    const int col = col0 + 4 * pp;
    const int r0 = 4 * fq + q, r1 = r0 + 16;
    const char* a0 = img + r0 * ROW_BYTES + (((col >> 3) ^ (r0 & 15)) << 4) + (col & 7) * 2;
    const char* a1 = img + r1 * ROW_BYTES + (((col >> 3) ^ (r1 & 15)) << 4) + (col & 7) * 2;
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)a0);
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)a1);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ __launch_bounds__(512, 2) void lora_bwd_image_kernel(const Params p) {
    REID_T16_ENTER();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, fq = lane >> 4;
    const int img = blockIdx.x;
    const int rbeg = img * p.rows_per_img;
    const int rend = min(p.M, rbeg + p.rows_per_img);
    if (rbeg >= rend) return;                                 // (whole workgroup)
    const int mu = p.img_mod[img];
    const int w0 = (mu * p.mask_r) & ~15;                     // first adapter column of the 16-column window
    const int c_lo = mu * p.mask_r - w0, c_hi = c_lo + p.mask_r;   // this modality's columns inside the window
    const int steps = (rend - rbeg + R - 1) / R;

    // one step's operands -> LDS: 48 dY instructions (eight waves x six) + two T instructions (waves 0, 1)
    auto issue = [&](int t, int buf) {
        char* base = smem + buf * BUF_BYTES;
        const int r0 = rbeg + t * R;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int qd = w * 6 + j;                         // wave-instruction: linear chunks [64 qd, 64 qd + 64)
            const int g = qd * 64 + lane;
            const int row = g / 96, c = g - row * 96;
            int gr = r0 + row; gr = gr < rend ? gr : rend - 1;
            const bf16_t* src = p.dY + (size_t)gr * p.lddy + ((c ^ (row & 15)) << 3);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + qd * 1024), 16, 0, 0);
        }
        if (w < 2) {
            const int g = w * 64 + lane;                      // T tile: 32 rows x 4 chunks, lane-linear
            const int row = g >> 2, c = g & 3;
            int gr = r0 + row; gr = gr < rend ? gr : rend - 1;
            __builtin_amdgcn_global_load_lds((gptr_t)(p.T + (size_t)gr * p.ldt + c * 8), (lptr_t)(base + IMG_BYTES + w * 1024), 16, 0, 0);
        }
    };
    issue(0, 0);
    // B^T fragments of the window (waves 0, 1: B operand of U = dY . B: n = adapter column w0 + l16, k = dY column)
    bf16x8 bfr[24];
    if (w < 2) {
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) bfr[ks] = *(const bf16x8*)(p.BT + (size_t)(w0 + l16) * p.ldbt + ks * 32 + 8 * fq);
    }
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < steps; ++t) {
        const int buf = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of step t has landed (and its U stores of step t - 1 are out)
        __syncthreads();                                      // ... everybody's; and everybody is done reading the other buffer (step t - 1)
        if (t + 1 < steps) issue(t + 1, buf ^ 1);
        const char* xs = smem + buf * BUF_BYTES;
        const char* ts = xs + IMG_BYTES;
        const int r0 = rbeg + t * R;
        if (w < 2) {
            // U[16 rows of this wave x 16 window columns]: the whole K = 768 contraction in this wave
            f32x4 ua = f32x4{0.f, 0.f, 0.f, 0.f};
            const int row = 16 * w + l16;
#pragma unroll
            for (int ks = 0; ks < 24; ++ks) {
                const bf16x8 af = *(const bf16x8*)(xs + row * ROW_BYTES + (((4 * ks + fq) ^ (row & 15)) << 4));
                ua = mfma16(af, bfr[ks], ua);
            }
            // lane: column n = l16 of the window, rows 4 fq + e
            const bool mine = l16 >= c_lo && l16 < c_hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = r0 + 16 * w + 4 * fq + e;
                if (m >= rend) continue;
                float v = ua[e];
                if (p.u_mode & 1) v += p.Up[(size_t)m * RP + w0 + l16];
                if (p.u_mode & 2) {
                    p.Up[(size_t)m * RP + w0 + l16] = v;
                } else {
                    p.U[(size_t)m * p.ldu + w0 + l16] = f32_to_bf16(mine ? v * p.scale : 0.f);
                    p.U[(size_t)m * p.ldu + (w0 ^ 16) + l16] = f32_to_bf16(0.f);      // the other half of the 32 adapter columns: other modalities
                }
            }
        } else {
            // dB[128 columns of this wave, window] += dY^T . T over the rows of this step (rows beyond the image: T fragment zeroed)
            bf16x8 tf = fused::tr_frag(ts, RP * 2, 0, w0, lane);
            const int nvalid = rend - r0;                     // >= 32 except in the last step
            if (nvalid < R) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {                 // element j of lane-quarter fq: image row 16 (j >> 2) + 4 fq + (j & 3) (tr_frag)
                    if (16 * (j >> 2) + 4 * fq + (j & 3) >= nvalid) tf[j] = 0;
                }
            }
            const int cbase = (w - 2) * 128;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = mfma16(tr_frag_swz(xs, cbase + i * 16, lane), tf, acc[i]);
        }
    }
    if (w >= 2) {
        // acc[i]: rows = dY columns cbase + 16 i + 4 fq + e, column = window column l16; only this modality's columns carry anything
        if (l16 >= c_lo && l16 < c_hi) {
            const int cbase = (w - 2) * 128;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(p.dB + (size_t)(cbase + i * 16 + 4 * fq + e) * p.lddb + w0 + l16, acc[i][e]);
        }
    }
}
}  // namespace image

}  // namespace

extern "C" int reid_lora_bwd_fused(const void* dY, int32_t lddy, const void* T, int32_t ldt, const void* BT, int32_t ldbt, void* U,
                                   int32_t ldu, float* dB, int32_t lddb, const int32_t* img_mod, int32_t rows_per_img, int32_t mask_r,
                                   int32_t M, int32_t N, int32_t Rp, float scale, float* u_partial, int32_t u_mode, void* stream) {
    REID_CHECK_ARG(u_mode >= 0 && u_mode <= 3 && (u_mode == 0 || u_partial), "reid_lora_bwd_fused: u_mode=%d needs u_partial", u_mode);
    REID_CHECK_ARG(dY && T && BT && U && dB && img_mod, "reid_lora_bwd_fused: null pointer");
    REID_CHECK_ARG(N == fused::N && Rp == fused::RP, "reid_lora_bwd_fused: N=%d Rp=%d (this kernel is built for N = 768, Rp = 32)", N, Rp);
    REID_CHECK_ARG(M > 0 && rows_per_img > 0 && mask_r > 0 && mask_r <= Rp, "reid_lora_bwd_fused: M=%d rows_per_img=%d mask_r=%d", M, rows_per_img, mask_r);
    REID_CHECK_ARG(lddy % 8 == 0 && ldt % 8 == 0 && ldbt % 8 == 0 && ldu % 2 == 0 && lddy >= N && ldt >= Rp && ldbt >= N && ldu >= Rp && lddb >= Rp,
                   "reid_lora_bwd_fused: leading dimensions");
    REID_CHECK_ARG(((uintptr_t)dY | (uintptr_t)T | (uintptr_t)BT) % 16 == 0 && (uintptr_t)U % 4 == 0, "reid_lora_bwd_fused: operand alignment");
    fused::Params p{(const bf16_t*)dY, (const bf16_t*)T, (const bf16_t*)BT, (bf16_t*)U, dB, img_mod, lddy, ldt, ldbt, ldu, lddb, rows_per_img,
                    mask_r, M, 0, scale, u_partial, u_mode};
    // REID_LORA_IMPL: 1 = the slab kernel always; default: one image per workgroup where an image spans at least one 32-row step
    if (reid_knob(KNOB_LORA_IMPL) != 1 && rows_per_img >= image::R && mask_r <= 16 && 16 % mask_r == 0) {
        REID_MAX_LDS((image::lora_bwd_image_kernel), image::LDS_BYTES);
        const int n_img = (M + rows_per_img - 1) / rows_per_img;
        hipLaunchKernelGGL(image::lora_bwd_image_kernel, dim3(n_img), dim3(512), image::LDS_BYTES, (hipStream_t)stream, p);
        REID_CHECK_LAUNCH("reid_lora_bwd_fused(image)");
        return REID_OK;
    }
    // few, long slabs: each workgroup flushes its [768, 32] slice of dB with atomics once (64 slabs = the flush traffic of gemm_tn's grid)
    int slabs = reid_knob(KNOB_TN_BLOCKS) > 0 ? reid_knob(KNOB_TN_BLOCKS) / 6 : 64;
    if (slabs < 1) slabs = 1;
    p.slab_rows = ((M + slabs - 1) / slabs + fused::R - 1) / fused::R * fused::R;
    slabs = (M + p.slab_rows - 1) / p.slab_rows;
    REID_MAX_LDS((fused::lora_bwd_fused_kernel<8>), fused::Geo<8>::LDS_BYTES);
    hipLaunchKernelGGL(fused::lora_bwd_fused_kernel<8>, dim3(slabs), dim3(512), fused::Geo<8>::LDS_BYTES, (hipStream_t)stream, p);
    REID_CHECK_LAUNCH("reid_lora_bwd_fused");
    return REID_OK;
}

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
