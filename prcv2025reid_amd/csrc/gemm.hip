// MER GEMM for gfx950:  C = epilogue(A.B^T + A2.B2^T + bias)   (see include/reid_hip.h)
//
// Structure (cdna_hip_programming.md section 5): BMxBNx64 tiles, operands staged global->LDS with
// 16-byte LDS-DMA (global_load_lds_dwordx4), two LDS buffers so tile t+1 streams in while tile t
// feeds v_mfma_f32_16x16x32_bf16.  The LDS image is lane-linear (what LDS-DMA writes); the
// bank-conflict swizzle is applied to the per-lane SOURCE address and again to the ds_read_b128
// address (rule 21).  The LoRA pair (A2, B2) simply extends the K loop by K2/32 half-steps, so the
// rank-r update costs one extra MFMA K-step instead of two skinny GEMMs and an add.
//
// Operand roles are swapped with respect to the textbook form (MFMA rows = weight rows n, MFMA
// columns = activation rows m) so that a lane's four accumulator registers are four CONSECUTIVE
// output columns of one output row: the epilogue then loads bias/residual and stores C with
// 8/16-byte accesses instead of 2/4-byte ones.
#include "gemm_core.h"

namespace {

struct GemmParams {
    const bf16_t* A; const bf16_t* B; const bf16_t* A2; const bf16_t* B2;
    const float* bias; const void* R; const bf16_t* aux;
    void* C; void* C2;
    const int32_t* img_mod;
    int M, N, K, K2;
    int lda, ldb, lda2, ldb2, ldr, ldaux, ldc, ldc2;
    int k2_group_n;
    int act, c_dtype, c2_dtype, r_dtype;
    int r_period;
    int mask_r, mask_period, rows_per_img;
    int c_group, c_group_stride, c_row_off;
    float alpha;
    int tiles_m, tiles_n;
};

using namespace gemmcore;

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void mer_gemm_kernel(const GemmParams p) {
    using C = Cfg<BM, BN, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: walk m fastest inside a weight panel so the panel stays L2-resident.
    const int lin = xcd_linear_block(blockIdx.x, gridDim.x);
    const int tn = lin / p.tiles_m, tm = lin % p.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int g = (p.k2_group_n > 0) ? (n0 / p.k2_group_n) : 0;
    const bf16_t* A2 = p.A2 ? p.A2 + (size_t)g * p.K2 : nullptr;

    f32x4 acc[C::TN][C::TM];
#pragma unroll
    for (int j = 0; j < C::TN; ++j)
#pragma unroll
        for (int i = 0; i < C::TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    mainloop<BM, BN, WM, WN>(p.A, p.lda, p.B, p.ldb, A2, p.lda2, p.B2, p.ldb2, p.M, p.N, p.K, p.K2, m0, n0, smem, acc);

    // ------------------------------------------------------------------ epilogue
    const int mrow = lane & 15;          // output row inside a 16x16 sub-tile
    const int ncol4 = (lane >> 4) * 4;   // first of 4 consecutive output columns
#pragma unroll
    for (int i = 0; i < C::TM; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 16 + mrow;
        if (m >= p.M) continue;
        int modality = -1;
        if (p.mask_r > 0) modality = p.img_mod[m / p.rows_per_img];
        const int rrow = p.r_period > 0 ? (m % p.r_period) : m;
        const size_t crow = p.c_group > 0
                                ? (size_t)(m / p.c_group) * p.c_group_stride + (m % p.c_group) + p.c_row_off
                                : (size_t)m;
#pragma unroll
        for (int j = 0; j < C::TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 16 + ncol4;
            if (n >= p.N) continue;
            f32x4 v = acc[j][i];
            if (p.bias) {
                const f32x4 b = *(const f32x4*)(p.bias + n);
                v += b;
            }
            if (p.R) {
                if (p.r_dtype == REID_F32) {
                    const f32x4 rr = *(const f32x4*)((const float*)p.R + (size_t)rrow * p.ldr + n);
                    v += rr;
                } else {
                    const bf16x4 rr = *(const bf16x4*)((const bf16_t*)p.R + (size_t)rrow * p.ldr + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bf16_to_f32((bf16_t)rr[e]);
                }
            }
            if (p.C2) {
                if (p.c2_dtype == REID_F32) {
                    *(f32x4*)((float*)p.C2 + crow * p.ldc2 + n) = v;
                } else {
                    uint2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    *(uint2*)((bf16_t*)p.C2 + crow * p.ldc2 + n) = pk;
                }
            }
            if (p.act != REID_ACT_NONE) {
                if (p.act <= REID_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = v[e];
                        v[e] = p.act == REID_ACT_GELU_ERF ? gelu_erf_f(x)
                               : p.act == REID_ACT_QUICK_GELU ? quick_gelu_f(x) : fmaxf(x, 0.f);
                    }
                } else {
                    const bf16x4 u = *(const bf16x4*)(p.aux + (size_t)m * p.ldaux + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = bf16_to_f32((bf16_t)u[e]);
                        const float d = p.act == REID_ACT_DGELU_ERF ? dgelu_erf_f(x)
                                        : p.act == REID_ACT_DQUICK_GELU ? dquick_gelu_f(x) : (x > 0.f ? 1.f : 0.f);
                        v[e] *= d;
                    }
                }
            }
            if (p.mask_r > 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int cm = ((n + e) % p.mask_period) / p.mask_r;
                    if (cm != modality) v[e] = 0.f;
                }
            }
            v *= p.alpha;
            if (p.c_dtype == REID_F32) {
                *(f32x4*)((float*)p.C + crow * p.ldc + n) = v;
            } else {
                uint2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                *(uint2*)((bf16_t*)p.C + crow * p.ldc + n) = pk;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch(GemmParams& p, hipStream_t s) {
    using C = Cfg<BM, BN, WM, WN>;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)mer_gemm_kernel<BM, BN, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            C::LDS_BYTES);
        attr_set = true;
    }
    const int grid = p.tiles_m * p.tiles_n;
    hipLaunchKernelGGL((mer_gemm_kernel<BM, BN, WM, WN>), dim3(grid), dim3(C::NT), C::LDS_BYTES, s, p);
    REID_CHECK_LAUNCH("reid_mer_gemm");
    return REID_OK;
}

}  // namespace

extern "C" int reid_mer_gemm(const reid_gemm_args* a, void* stream) {
    REID_CHECK_ARG(a != nullptr, "reid_mer_gemm: null args");
    REID_CHECK_ARG(a->A && a->B && a->C, "reid_mer_gemm: A, B, C must be non-null");
    REID_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "reid_mer_gemm: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
    REID_CHECK_ARG(a->K % 64 == 0, "reid_mer_gemm: K=%d must be a multiple of 64", a->K);
    REID_CHECK_ARG(a->N % 4 == 0, "reid_mer_gemm: N=%d must be a multiple of 4", a->N);
    REID_CHECK_ARG(a->lda >= a->K && a->ldb >= a->K && a->lda % 8 == 0 && a->ldb % 8 == 0,
                   "reid_mer_gemm: lda/ldb must be >= K and multiples of 8 (16-byte rows)");
    REID_CHECK_ARG(a->ldc >= a->N && a->ldc % 4 == 0, "reid_mer_gemm: ldc=%d", a->ldc);
    if (a->A2) {
        REID_CHECK_ARG(a->B2 && a->K2 > 0 && a->K2 % 32 == 0, "reid_mer_gemm: K2=%d must be a positive multiple of 32",
                       a->K2);
        REID_CHECK_ARG(a->lda2 % 8 == 0 && a->ldb2 % 8 == 0 && a->ldb2 >= a->K2, "reid_mer_gemm: lda2/ldb2");
        const int groups = a->k2_group_n > 0 ? (a->N + a->k2_group_n - 1) / a->k2_group_n : 1;
        REID_CHECK_ARG(a->lda2 >= groups * a->K2, "reid_mer_gemm: lda2=%d < groups*K2=%d", a->lda2, groups * a->K2);
        REID_CHECK_ARG(a->k2_group_n == 0 || a->k2_group_n % 128 == 0, "reid_mer_gemm: k2_group_n must be a multiple of 128");
    }
    REID_CHECK_ARG(a->act >= 0 && a->act <= REID_ACT_DRELU, "reid_mer_gemm: act=%d", a->act);
    REID_CHECK_ARG(a->act < REID_ACT_DGELU_ERF || (a->aux && a->ldaux >= a->N), "reid_mer_gemm: D* activation needs aux");
    REID_CHECK_ARG(!a->R || a->ldr >= a->N, "reid_mer_gemm: ldr");
    REID_CHECK_ARG(!a->C2 || a->ldc2 >= a->N, "reid_mer_gemm: ldc2");
    REID_CHECK_ARG(a->mask_r == 0 || (a->img_mod && a->rows_per_img > 0 && a->mask_period > 0),
                   "reid_mer_gemm: modality mask needs img_mod, rows_per_img, mask_period");
    REID_CHECK_ARG(a->c_group == 0 || a->c_group_stride >= a->c_group, "reid_mer_gemm: c_group_stride");
    GemmParams p;
    p.A = (const bf16_t*)a->A; p.B = (const bf16_t*)a->B; p.A2 = (const bf16_t*)a->A2; p.B2 = (const bf16_t*)a->B2;
    p.bias = a->bias; p.R = a->R; p.aux = (const bf16_t*)a->aux; p.C = a->C; p.C2 = a->C2; p.img_mod = a->img_mod;
    p.M = a->M; p.N = a->N; p.K = a->K; p.K2 = a->A2 ? a->K2 : 0;
    p.lda = a->lda; p.ldb = a->ldb; p.lda2 = a->lda2; p.ldb2 = a->ldb2; p.ldr = a->ldr; p.ldaux = a->ldaux;
    p.ldc = a->ldc; p.ldc2 = a->ldc2; p.k2_group_n = a->k2_group_n;
    p.act = a->act; p.c_dtype = a->c_dtype; p.c2_dtype = a->c2_dtype; p.r_dtype = a->r_dtype;
    p.r_period = a->r_period; p.mask_r = a->mask_r; p.mask_period = a->mask_period; p.rows_per_img = a->rows_per_img;
    p.c_group = a->c_group; p.c_group_stride = a->c_group_stride; p.c_row_off = a->c_row_off;
    p.alpha = a->alpha == 0.f ? 1.f : a->alpha;
    hipStream_t s = (hipStream_t)stream;
    // skinny outputs (LoRA down-projections, N <= 96) use a tall tile so no MFMA work is spent on padding
    if (a->N <= 32) return launch<256, 32, 4, 1>(p, s);
    if (a->N <= 64) return launch<256, 64, 4, 1>(p, s);
    if (a->N <= 96) return launch<128, 32, 4, 1>(p, s);
    return launch<128, 128, 2, 2>(p, s);
}
