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
    // nmod * r <= 64: all adapter rows of the tile are staged once.  Larger ranks (r <= 64): ONE modality's rows at a time (per_mod), the
    // same fma chains in the same order -- the results do not depend on which form ran.
    const bool per_mod = R > MT;
    for (int i = t; i < MT * (MT / 4); i += 256) {       // 16-byte loads along k
        const int row = i / (MT / 4), c4 = (i % (MT / 4)) * 4;
        *(f32x4*)&sW[row][c4] = *(const f32x4*)(W + (size_t)(n0 + row) * K + k0 + c4);
        if (!per_mod && row < R) *(f32x4*)&sA[row][c4] = *(const f32x4*)(A + (size_t)(g * Rp + row) * K + k0 + c4);
    }
    if (!per_mod) {
        for (int i = t; i < MT * R; i += 256) {
            const int row = i / R, c = i % R;
            sBt[c][row] = B[(size_t)(n0 + row) * Rp + c];
        }
    }
    __syncthreads();
    const int c8 = (t & 7) * 8, r0 = t >> 3;             // a thread owns 8 consecutive elements of rows r0 and r0 + 32
    // Both passes evaluate element (n, k) of modality mu by the SAME chain: acc = fma(B[n][mu r + j], A[mu r + j][k], acc) for j = 0..r-1,
    // then fma(s, acc, W[n][k]) -- so W_eff^T is the exact transpose of W_eff.
    for (int mu = 0; mu < nmod; ++mu) {
        if (per_mod) {
            __syncthreads();                              // the previous modality's rows are no longer read
            for (int i = t; i < r * (MT / 4); i += 256) {
                const int row = i / (MT / 4), c4 = (i % (MT / 4)) * 4;
                *(f32x4*)&sA[row][c4] = *(const f32x4*)(A + (size_t)(g * Rp + mu * r + row) * K + k0 + c4);
            }
            for (int i = t; i < MT * r; i += 256) {
                const int row = i / r, c = i % r;
                sBt[c][row] = B[(size_t)(n0 + row) * Rp + mu * r + c];
            }
            __syncthreads();
        }
        const int ab = per_mod ? 0 : mu * r;              // first staged adapter row of this modality
        bf16_t* d = dst + (size_t)mu * N * K;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = r0 + 32 * h;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < r; ++j) {
                const float b = sBt[ab + j][n];
                const f32x4 a0 = *(const f32x4*)&sA[ab + j][c8], a1 = *(const f32x4*)&sA[ab + j][c8 + 4];
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
                    const float a = sA[ab + j][k];
                    const f32x4 b0 = *(const f32x4*)&sBt[ab + j][c8], b1 = *(const f32x4*)&sBt[ab + j][c8 + 4];
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
constexpr int NBUF = 3;                                      // ring of step buffers: two steps (96 KiB) in flight while one is read
constexpr int USTAGE_BYTES = R * RP * 2;                    // U tile of a step, row-major, staged for whole-row stores
constexpr int LDS_BYTES = NBUF * BUF_BYTES + USTAGE_BYTES;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
using fused::s4; using fused::lds_s4_ptr; using fused::Params;

// One LDS-DMA wave-instruction (64 lanes x 16 bytes -> 1 KiB of LDS at `lds_base`, wave-uniform) issued from inline assembly: hipcc then
// has no vector-memory operation of the ring in its scoreboard.  With the builtin it places `s_waitcnt vmcnt(0)` in front of every
// ds_read_b64_tr_b16 while a DMA is outstanding -- the transposed reads of every step then wait for the WHOLE ring (first form of this
// kernel: 10 GB/s per workgroup whatever the ring depth).  Consequently no other global access may sit inside the step loops below
// (the compiler's own waits for it would drain the ring as well): outputs are kept in registers and written after the loop.
// M0 = LDS base; nothing else in these kernels uses M0.
typedef __attribute__((address_space(3))) char* lds_cptr;
__device__ __forceinline__ uint32_t lds_addr(const char* ptr) { return (uint32_t)(uintptr_t)(lds_cptr)(char*)ptr; }
__device__ __forceinline__ void dma16(const void* src, uint32_t lds_base) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory");
}

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

// U = mask(Up) * scale in 16 bits once the last column block of a wide cotangent has added its partial sums (fp32 [M, 32]) to Up
__global__ __launch_bounds__(256) void u_finish_kernel(const float* __restrict__ Up, bf16_t* __restrict__ U, int ldu, const int32_t* __restrict__ img_mod,
                                                       int rows_per_img, int mask_r, float scale, int M) {
    REID_T16_ENTER();
    const int i = blockIdx.x * 256 + threadIdx.x;             // one thread per (row, pair of columns)
    const int m = i >> 4, c = (i & 15) * 2;
    if (m >= M) return;
    const int mu = img_mod[m / rows_per_img];
    const float a = Up[(size_t)m * RP + c], b = Up[(size_t)m * RP + c + 1];
    *(uint32_t*)(U + (size_t)m * ldu + c) = pack_bf16x2(c / mask_r == mu ? a * scale : 0.f, (c + 1) / mask_r == mu ? b * scale : 0.f);
}

#ifdef REID_EXPERIMENTS
#define LORA_ABL(bit) ((p.slab_rows >> (bit)) & 1)            // timing experiments (wrong results): REID_LORA_IMPL = 16 + bits, tools/exp_lora_ablate.py
#else
#define LORA_ABL(bit) 0
#endif
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
    const uint32_t smem_a = lds_addr(smem);
    auto issue = [&](int t, int buf) {
        const uint32_t base = smem_a + buf * BUF_BYTES;
        const int r0 = rbeg + t * R;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int qd = w * 6 + j;                         // wave-instruction: linear chunks [64 qd, 64 qd + 64)
            const int g = qd * 64 + lane;
            const int row = g / 96, c = g - row * 96;
            int gr = r0 + row; gr = gr < rend ? gr : rend - 1;
            dma16(p.dY + (size_t)gr * p.lddy + ((c ^ (row & 15)) << 3), base + qd * 1024);
        }
        if (w < 2 && !LORA_ABL(4)) {
            const int g = w * 64 + lane;                      // T tile: 32 rows x 4 chunks, lane-linear
            const int row = g >> 2, c = g & 3;
            int gr = r0 + row; gr = gr < rend ? gr : rend - 1;
            dma16(p.T + (size_t)gr * p.ldt + c * 8, base + IMG_BYTES + w * 1024);
        }
    };
    issue(0, 0);
    if (steps > 1) issue(1, 1);
    // B^T fragments of the window (waves 0, 1: B operand of U = dY . B: n = adapter column w0 + l16, k = dY column)
    bf16x8 bfr[24];
    if (w < 2 && !LORA_ABL(5)) {
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) bfr[ks] = *(const bf16x8*)(p.BT + (size_t)(w0 + l16) * p.ldbt + ks * 32 + 8 * fq);
        // a use of every fragment HERE: the compiler tracks these loads and would otherwise wait for them with `vmcnt(0)` at their first use
        // INSIDE the step loop -- on every step, draining the DMA ring the loop exists to keep full (first form: 10 GB/s per workgroup
        // whatever the ring depth)
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) asm volatile("" ::"v"(bfr[ks]));
    }
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t pend_val = u32x4_t{0u, 0u, 0u, 0u};                // waves 0, 1: the previous step's output, not yet issued
    void* pend_ptr = nullptr;
    int pend_n = 0;                                            // rows (u_mode != 0) / "this lane has a row" (u_mode == 0) still to go out
    auto flush_pending = [&]() {
        if (LORA_ABL(1)) { pend_n = 0; return; }
        if (p.u_mode == 0) {
            if (pend_n) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(pend_ptr), "v"(pend_val) : "memory");
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e >= pend_n) continue;
                float* pp = (float*)pend_ptr + (size_t)e * RP;
                const unsigned vv = pend_val[e];
                if (p.u_mode == 2) asm volatile("global_store_dword %0, %1, off" ::"v"(pp), "v"(vv) : "memory");
                else asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(pp), "v"(vv) : "memory");
            }
        }
        pend_n = 0;
    };
    int buf = 0;
    for (int t = 0; t < steps; ++t) {
        // this wave's share of step t has landed: everything but the DMA instructions of step t + 1 (the newest this wave has issued: six,
        // seven in waves 0 and 1 which also fetch T) -- the two-byte U stores of the previous step are older than those and drain here too
        if (t + 1 < steps) {
            if (w < 2) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // ... everybody's share; and everybody is done reading the buffer of step t - 1
        __builtin_amdgcn_sched_barrier(0);
        const char* xs = smem + buf * BUF_BYTES;
        const char* ts = xs + IMG_BYTES;
        const int r0 = rbeg + t * R;
        if (w < 2) {
            // U[16 rows of this wave x 16 window columns]: the whole K = 768 contraction in this wave
            // (four accumulators, fragment reads issued four at a time: one 24-long chain of read -> wait -> MFMA made this wave the slowest
            //  of the workgroup -- ~1.7 us per step, the first form ran at 10 GB/s per workgroup whatever the depth of the DMA ring)
            flush_pending();                                  // the PREVIOUS step's rows go out first: a whole step ahead of the next counted wait
            f32x4 u4[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) u4[a] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int row = 16 * w + l16;
            const char* xrow = xs + row * ROW_BYTES;
            const int rsw = row & 15;
#pragma unroll
            for (int k0 = 0; k0 < (LORA_ABL(0) ? 0 : 24); k0 += 4) {
                bf16x8 af[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) af[a] = *(const bf16x8*)(xrow + (((4 * (k0 + a) + fq) ^ rsw) << 4));
#pragma unroll
                for (int a = 0; a < 4; ++a) u4[a] = mfma16(af[a], bfr[k0 + a], u4[a]);
            }
            const f32x4 ua = (u4[0] + u4[1]) + (u4[2] + u4[3]);
            // Outputs: fire and forget from inline assembly (see dma16: no compiler-visible global access inside the loop), and issued ONE
            // STEP LATE (`flush_pending` at the start of the next step's arithmetic, above): a store issued here would be the operation right
            // in front of the next counted wait, which would then sit out its whole write acknowledgement (~1.7 us per step: every form of
            // this kernel measured 29.5-29.9 us until the stores moved); issued a step ahead, the acknowledgement passes under the wait for
            // the next step's DMA.
            // lane: column n = l16 of the window, rows 4 fq + e.  u_mode: 0 = U (mask, scale, 16-bit); 2 = first column block of a wide
            // cotangent: partial sum stored to Up; 3 / 1 = later blocks: added to Up (the host entry finishes U after the last block)
            const bool mine = l16 >= c_lo && l16 < c_hi;
            if (p.u_mode == 0) {
                // through LDS ([32 rows][32 adapter columns], 16-bit): the wave then writes its 16 rows as whole 64-byte rows (one 16-byte
                // store per lane) -- two-byte stores straight from the accumulator layout were 112 wave-instructions of eight 32-byte pieces
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bf16_t* ul = (bf16_t*)(smem + NBUF * BUF_BYTES) + (16 * w + 4 * fq + e) * RP;
                    ul[w0 + l16] = f32_to_bf16(mine ? ua[e] * p.scale : 0.f);
                    ul[(w0 ^ 16) + l16] = f32_to_bf16(0.f);                   // the other half of the 32 adapter columns: other modalities
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's 16 rows are in LDS (written and read by this wave only)
                __builtin_amdgcn_wave_barrier();
                const int row = 16 * w + (lane >> 2), ch = lane & 3;
                pend_val = *(const u32x4_t*)(smem + NBUF * BUF_BYTES + row * (RP * 2) + ch * 16);
                pend_ptr = (void*)(p.U + (size_t)(r0 + row) * p.ldu + ch * 8);
                pend_n = (r0 + row) < rend ? 1 : 0;
            } else {
                pend_val = u32x4_t{__float_as_uint(ua[0]), __float_as_uint(ua[1]), __float_as_uint(ua[2]), __float_as_uint(ua[3])};
                pend_ptr = (void*)(p.Up + (size_t)(r0 + 16 * w + 4 * fq) * RP + w0 + l16);
                const int left = rend - (r0 + 16 * w + 4 * fq);            // rows e = 0 .. left - 1 of this lane exist
                pend_n = left < 0 ? 0 : (left > 4 ? 4 : left);
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
            for (int i = 0; i < (LORA_ABL(2) ? 0 : 8); ++i) acc[i] = mfma16(tr_frag_swz(xs, cbase + i * 16, lane), tf, acc[i]);
        }
        // step t + 2 into the buffer step t - 1 was read from (every wave is past this step's barrier, hence done with it); issued AFTER
        // this step's U stores so that the counted wait above has exactly one step's DMA instructions behind everything it waits for
        const int nb = buf == 0 ? NBUF - 1 : buf - 1;        // (t + 2) % NBUF == (t - 1) % NBUF
        if (t + 2 < steps) issue(t + 2, nb);
        buf = buf + 1 == NBUF ? 0 : buf + 1;
    }
    if (w < 2) flush_pending();                               // the last step's U rows
    // dB of this image: through LDS (the ring is idle now) so that every atomic wave-instruction has 64 active lanes = eight dY columns x the
    // modality's r adapter columns -- straight from the accumulator layout half of the lanes of each of 192 instructions were masked out
    __syncthreads();
    float* fl = (float*)smem;                                 // [768 dY columns][16 window columns]
    if (w >= 2) {
        const int cbase = (w - 2) * 128;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) fl[(cbase + i * 16 + 4 * fq + e) * 16 + l16] = acc[i][e];
    }
    __syncthreads();
    {
        const int rr = p.mask_r, total = N * rr;
        for (int idx = tid; idx < (LORA_ABL(3) ? 0 : total); idx += 512) {
            const int col = idx / rr, j = idx - col * rr;
            atomicAdd(p.dB + (size_t)col * p.lddb + mu * rr + j, fl[col * 16 + c_lo + j]);
        }
    }
}

// dA of one MERLinear in the same form: dA[g Rp + c, k] += sum_m U[m, g Rp + c] . X[m, k] -- one image (one modality) per workgroup and
// 768-column block of X, the X tile of a 32-row step by LDS-DMA into the same swizzled image, the U windows (G groups x 16 columns) beside it,
// all eight waves accumulate [96 columns x 16] per group from transposed reads; 768 x r fp32 atomics per group at the end.  Replaces
// reid_gemm_tn(U, X) for the adapter gradients (r03: 48 launches x 70-75 us inside the step).  U must be modality-masked (it is: reid_lora_bwd_fused
// and the mask epilogue of reid_mer_gemm write zeros elsewhere).
struct DaParams {
    const bf16_t* X; const bf16_t* U; float* dA;
    const int32_t* img_mod;
    int ldx, ldu, ldda, rows_per_img, mask_r, M, Rp;
};
template <int G>
__global__ __launch_bounds__(512, 2) void lora_da_image_kernel(const DaParams p) {
    REID_T16_ENTER();
    constexpr int UB = 4096;                                  // U windows of a step: G x [32 rows][16 columns] 16-bit (<= 3 KiB), padded
    constexpr int BUF = IMG_BYTES + UB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, fq = lane >> 4;
    const int img = blockIdx.x, cblk = blockIdx.y;
    const int rbeg = img * p.rows_per_img;
    const int rend = min(p.M, rbeg + p.rows_per_img);
    if (rbeg >= rend) return;
    const int mu = p.img_mod[img];
    const int w0 = (mu * p.mask_r) & ~15;
    const int c_lo = mu * p.mask_r - w0, c_hi = c_lo + p.mask_r;
    const int steps = (rend - rbeg + R - 1) / R;
    const bf16_t* Xb = p.X + (size_t)cblk * N;
    const uint32_t smem_a = lds_addr(smem);
    auto issue = [&](int t, int buf) {
        const uint32_t base = smem_a + buf * BUF;
        const int r0 = rbeg + t * R;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int qd = w * 6 + j;
            const int g = qd * 64 + lane;
            const int row = g / 96, c = g - row * 96;
            int gr = r0 + row; gr = gr < rend ? gr : rend - 1;
            dma16(Xb + (size_t)gr * p.ldx + ((c ^ (row & 15)) << 3), base + qd * 1024);
        }
        if (w < G) {                                          // group w's window: 32 rows x 2 chunks, lane-linear (row pitch 32 bytes)
            const int row = lane >> 1, c = lane & 1;
            int gr = r0 + row; gr = gr < rend ? gr : rend - 1;
            dma16(p.U + (size_t)gr * p.ldu + w * p.Rp + w0 + c * 8, base + IMG_BYTES + w * 1024);
        }
    };
    issue(0, 0);
    if (steps > 1) issue(1, 1);
    f32x4 acc[G][6];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[g][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int buf = 0;
    for (int t = 0; t < steps; ++t) {
        if (t + 1 < steps) {                                  // all but the newest step's DMA instructions (six, seven in the waves that fetch U)
            if (w < G) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* xs = smem + buf * BUF;
        const char* us = xs + IMG_BYTES;
        const int nvalid = rend - (rbeg + t * R);
        bf16x8 uf[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uf[g] = fused::tr_frag(us + g * 1024, 32, 0, 0, lane);
            if (nvalid < R) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (16 * (j >> 2) + 4 * fq + (j & 3) >= nvalid) uf[g][j] = 0;
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const bf16x8 xf = tr_frag_swz(xs, w * 96 + i * 16, lane);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g][i] = mfma16(uf[g], xf, acc[g][i]);      // (U first: a lane then owns ONE X column, l16 = 16 consecutive k of dA)
        }
        const int nb = buf == 0 ? NBUF - 1 : buf - 1;
        if (t + 2 < steps) issue(t + 2, nb);
        buf = buf + 1 == NBUF ? 0 : buf + 1;
    }
    // acc[g][i]: row = window column 4 fq + e of group g (an adapter row of dA), column = X column w 96 + 16 i + l16: an atomic wave-instruction
    // covers 16 consecutive k of up to four adapter rows (64-byte runs).  The first form had the operands the other way round -- every lane a
    // different ROW of dA, 3 KiB apart: the 17x-slower atomic shape of MI355X_MICROARCH.md, and the step 0.85 ms slower than with reid_gemm_tn.
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = 4 * fq + e;
        if (c < c_lo || c >= c_hi) continue;
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < 6; ++i)
                atomicAdd(p.dA + (size_t)(g * p.Rp + w0 + c) * p.ldda + (size_t)cblk * N + w * 96 + i * 16 + l16, acc[g][i][e]);
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
#ifdef REID_EXPERIMENTS
        p.slab_rows = reid_knob(KNOB_LORA_IMPL) >= 16 ? reid_knob(KNOB_LORA_IMPL) - 16 : 0;
#endif
        hipLaunchKernelGGL(image::lora_bwd_image_kernel, dim3(n_img), dim3(512), image::LDS_BYTES, (hipStream_t)stream, p);
        REID_CHECK_LAUNCH("reid_lora_bwd_fused(image)");
        if (u_mode == 1) {                                    // last column block of a wide cotangent: every partial sum is in Up now
            hipLaunchKernelGGL(image::u_finish_kernel, dim3((M * 16 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)u_partial, (bf16_t*)U,
                               ldu, img_mod, rows_per_img, mask_r, scale, M);
            REID_CHECK_LAUNCH("reid_lora_bwd_fused(finish)");
        }
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

extern "C" int reid_lora_da_fused(const void* X, int32_t ldx, const void* U, int32_t ldu, float* dA, int32_t ldda, const int32_t* img_mod,
                                  int32_t rows_per_img, int32_t mask_r, int32_t M, int32_t K, int32_t Rp, int32_t n_groups, void* stream) {
    REID_CHECK_ARG(X && U && dA && img_mod, "reid_lora_da_fused: null pointer");
    REID_CHECK_ARG(M > 0 && K > 0 && K % 768 == 0 && Rp == 32 && (n_groups == 1 || n_groups == 3), "reid_lora_da_fused: M=%d K=%d Rp=%d groups=%d (K a multiple of 768, Rp = 32, 1 or 3 groups)", M, K, Rp, n_groups);
    REID_CHECK_ARG(rows_per_img >= image::R && mask_r > 0 && mask_r <= 16 && 16 % mask_r == 0, "reid_lora_da_fused: rows_per_img=%d (>= 32) mask_r=%d (a divisor of 16)", rows_per_img, mask_r);
    REID_CHECK_ARG(ldx % 8 == 0 && ldu % 8 == 0 && ldx >= K && ldu >= n_groups * Rp && ldda >= K, "reid_lora_da_fused: leading dimensions");
    REID_CHECK_ARG(((uintptr_t)X | (uintptr_t)U) % 16 == 0, "reid_lora_da_fused: operand alignment");
    image::DaParams p{(const bf16_t*)X, (const bf16_t*)U, dA, img_mod, ldx, ldu, ldda, rows_per_img, mask_r, M, Rp};
    const int n_img = (M + rows_per_img - 1) / rows_per_img;
    constexpr int LDS = image::NBUF * (image::IMG_BYTES + 4096);
    if (n_groups == 1) {
        REID_MAX_LDS((image::lora_da_image_kernel<1>), LDS);
        hipLaunchKernelGGL(image::lora_da_image_kernel<1>, dim3(n_img, K / 768), dim3(512), LDS, (hipStream_t)stream, p);
    } else {
        REID_MAX_LDS((image::lora_da_image_kernel<3>), LDS);
        hipLaunchKernelGGL(image::lora_da_image_kernel<3>, dim3(n_img, K / 768), dim3(512), LDS, (hipStream_t)stream, p);
    }
    REID_CHECK_LAUNCH("reid_lora_da_fused");
    return REID_OK;
}

extern "C" int reid_merge_lora_table(const int64_t* table, int32_t n_entries, int32_t max_tiles, const float* arena, void* weff,
                                     int32_t Rp, int32_t r, int32_t nmod, float scaling, void* stream) {
    REID_CHECK_ARG(table && arena && weff, "reid_merge_lora_table: null pointer");
    REID_CHECK_ARG(n_entries > 0 && n_entries <= 65535 && max_tiles > 0, "reid_merge_lora_table: n_entries=%d max_tiles=%d", n_entries, max_tiles);
    REID_CHECK_ARG(r > 0 && nmod > 0 && r <= MT && nmod * r <= Rp, "reid_merge_lora_table: r = %d must be <= %d and nmod*r = %d <= Rp = %d", r, MT, nmod * r, Rp);
    hipLaunchKernelGGL(merge_lora_kernel, dim3(max_tiles, n_entries), dim3(256), 0, (hipStream_t)stream, table, arena, (bf16_t*)weff, Rp, r, nmod,
                       scaling);
    REID_CHECK_LAUNCH("reid_merge_lora_table");
    return REID_OK;
}
