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
#include <stdlib.h>

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

// Epilogue of one wave's [TM*16 x TN*16] sub-tile whose first element is C[m_base][n_base].
// The accumulator layout (lane = one output row, 4 consecutive columns per 16x16 sub-tile) would store 8-byte pieces
// scattered over 16 rows per instruction (measured: that store tail cost as much as the whole K loop at K = 768).
// Instead the wave transposes the sub-tile through its own LDS slice `stg`, 16 output rows at a time, and then works
// on 16-byte ROW pieces: the lanes of one instruction cover whole contiguous row segments, so bias / residual / aux
// loads and the C stores are full-line coalesced.
template <int TM, int TN>
__device__ __forceinline__ void store_tile(const GemmParams& p, f32x4 (&acc)[TN][TM], char* stg, int m_base, int n_base, int lane) {
    constexpr int WTN = TN * 16;
    constexpr int PITCH = WTN * 4 + 16;                 // bytes per staged row (+16: conflict-free 16-byte writes)
    constexpr int LPR = WTN / 4;                        // lanes per staged row
    constexpr int RPI = 64 / LPR;                       // rows per read instruction
    static_assert(64 % LPR == 0 && 16 % RPI == 0, "unsupported wave tile width");
    constexpr int IT = 16 / RPI;                        // read instructions per 16 staged rows
    constexpr bool PREF = (TM * IT <= 16);              // prefetch residual / aux operands of the whole sub-tile
    const int mrow = lane & 15;
    const int ncol4 = (lane >> 4) * 4;
    const int rr = lane / LPR, rc4 = (lane % LPR) * 4;
    const int n = n_base + rc4;                         // this lane's 4 output columns: the same for every row it handles
    const bool nok = n < p.N;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias && nok) bv = *(const f32x4*)(p.bias + n);
    // All global operands of the epilogue are requested BEFORE the LDS transposes start: issued inside the row loop
    // every 16-row step waited a full memory latency (measured ~8 us of fixed cost per 128x128 tile).
    f32x4 rv[PREF ? TM : 1][PREF ? IT : 1];
    bf16x4 av[PREF ? TM : 1][PREF ? IT : 1];
    auto load_r = [&](int m) -> f32x4 {
        const int rrow = p.r_period > 0 ? (m % p.r_period) : m;
        if (p.r_dtype == REID_F32) return *(const f32x4*)((const float*)p.R + (size_t)rrow * p.ldr + n);
        const bf16x4 q = *(const bf16x4*)((const bf16_t*)p.R + (size_t)rrow * p.ldr + n);
        return f32x4{bf16_to_f32((bf16_t)q[0]), bf16_to_f32((bf16_t)q[1]), bf16_to_f32((bf16_t)q[2]), bf16_to_f32((bf16_t)q[3])};
    };
    if (PREF) {
        if (p.R) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int m = m_base + i * 16 + it * RPI + rr;
                    rv[i][it] = (m < p.M && nok) ? load_r(m) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
        }
        if (p.act >= REID_ACT_DGELU_ERF) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int m = m_base + i * 16 + it * RPI + rr;
                    av[i][it] = (m < p.M && nok) ? *(const bf16x4*)(p.aux + (size_t)m * p.ldaux + n) : bf16x4{0, 0, 0, 0};
                }
        }
    }
#pragma clang loop unroll(full)
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) *(f32x4*)(stg + mrow * PITCH + (j * 16 + ncol4) * 4) = acc[j][i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // same-wave LDS ops complete in order; stop compiler reordering
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int r = it * RPI + rr;
            f32x4 v = *(const f32x4*)(stg + r * PITCH + rc4 * 4);
            const int m = m_base + i * 16 + r;
            if (m >= p.M || !nok) continue;
            v += bv;
            if (p.R) v += PREF ? rv[PREF ? i : 0][PREF ? it : 0] : load_r(m);
            const size_t crow = p.c_group > 0
                                    ? (size_t)(m / p.c_group) * p.c_group_stride + (m % p.c_group) + p.c_row_off
                                    : (size_t)m;
            if (p.C2) {
                if (p.c2_dtype == REID_F32) *(f32x4*)((float*)p.C2 + crow * p.ldc2 + n) = v;
                else *(uint2*)((bf16_t*)p.C2 + crow * p.ldc2 + n) = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
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
                    const bf16x4 u = PREF ? av[PREF ? i : 0][PREF ? it : 0] : *(const bf16x4*)(p.aux + (size_t)m * p.ldaux + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = bf16_to_f32((bf16_t)u[e]);
                        v[e] *= p.act == REID_ACT_DGELU_ERF ? dgelu_erf_f(x)
                                : p.act == REID_ACT_DQUICK_GELU ? dquick_gelu_f(x) : (x > 0.f ? 1.f : 0.f);
                    }
                }
            }
            if (p.mask_r > 0) {
                const int modality = p.img_mod[m / p.rows_per_img];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (((n + e) % p.mask_period) / p.mask_r != modality) v[e] = 0.f;
            }
            v *= p.alpha;
            if (p.c_dtype == REID_F32) *(f32x4*)((float*)p.C + crow * p.ldc + n) = v;
            else *(uint2*)((bf16_t*)p.C + crow * p.ldc + n) = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
        asm volatile("" ::: "memory");
    }
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void mer_gemm_kernel(const GemmParams p) {
    using C = Cfg<BM, BN, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware, L2-aware tile order (gemm_core.h)
    const int lin = xcd_linear_block(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(lin, p.tiles_m, p.tiles_n, tm, tn);
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
    constexpr int WTM = BM / WM, WTN = BN / WN;
    static_assert(C::NW * 16 * (WTN * 4 + 16) <= C::LDS_BYTES, "epilogue staging does not fit");
    __syncthreads();                                    // all waves are done reading the operand buffers
    store_tile<C::TM, C::TN>(p, acc, smem + wave * (16 * (WTN * 4 + 16)), m0 + wm * WTM, n0 + wn * WTN, lane);
}

// ------------------------------------------------------------------------------------------------------------------
// Persistent, software-pipelined variant (the one the big projections use).
// One workgroup per CU walks its tiles; the (tile, k-step) pairs of ALL its tiles form one continuous sequence of
// steps that streams through an NSTAGE-deep LDS ring, so the LDS-DMA loads of the next tile's first K-steps are in
// flight while the current tile finishes and runs its epilogue -- the per-tile load latency that dominated the
// one-tile-per-workgroup kernel at K = 768 (~8 us of fixed cost on a ~17 us loop) is paid once per launch.
// Per step: counted `s_waitcnt vmcnt(L)` (one stage may stay in flight), ONE raw s_barrier, issue stage q+2, MFMAs of
// stage q.  (cdna_hip_programming.md "Pipelining across barriers": raw barrier + counted vmcnt, all LDS in one array.)
template <int BM, int BN, int WM, int WN, int NSTAGE>
__global__ __launch_bounds__(WM* WN * 64) void mer_gemm_persist_kernel(const GemmParams p) {
    static_assert(NSTAGE >= 3, "stage q+2 is issued while stage q is being read");
    using C = Cfg<BM, BN, WM, WN>;
    constexpr int STAGE = C::A_BYTES + C::B_BYTES;
    constexpr int LOADS = C::A_INSTR + C::B_INSTR;          // LDS-DMA instructions per wave per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nk = p.K >> 6;
    const int nk2 = p.A2 ? (p.K2 >> 5) : 0;
    const int S = nk + nk2;
    const int total = p.tiles_m * p.tiles_n;
    const int first = blockIdx.x, stride = gridDim.x;
    // workgroup `first` takes one tile of every round it is inside (the last round may be partial)
    const int n_my = total / stride + ((total % stride) > first ? 1 : 0);
    const int Q = n_my * S;
    if (Q == 0) return;

    auto tile_of = [&](int ti, int& m0, int& n0) {
        // round r of the grid covers tiles [r*stride, (r+1)*stride); inside a round the XCD remap gives each XCD a
        // contiguous run, and tile_coords() keeps that run on a few activation blocks / weight panels (L2 reuse)
        const int base = ti * stride;
        const int span = min(stride, total - base);
        const int t = base + xcd_linear_block(first, span);
        int tm, tn;
        tile_coords(t, p.tiles_m, p.tiles_n, tm, tn);
        m0 = tm * BM; n0 = tn * BN;
    };
    // issue pointer (runs two steps ahead of the compute pointer); tile coordinates are recomputed only per tile
    int i_ti = 0, i_ks = 0, i_q = 0, i_m0, i_n0;
    tile_of(0, i_m0, i_n0);
    auto issue_next = [&]() {
        char* la = smem + (i_q % NSTAGE) * STAGE;
        char* lb = la + C::A_BYTES;
        if (i_ks < nk) {
            stage<BM, C::NW, false>(p.A, p.lda, i_m0, p.M - 1, i_ks << 6, la, wave, lane);
            stage<BN, C::NW, false>(p.B, p.ldb, i_n0, p.N - 1, i_ks << 6, lb, wave, lane);
        } else {
            const int g = (p.k2_group_n > 0) ? (i_n0 / p.k2_group_n) : 0;
            const int k2 = (i_ks - nk) << 5;
            stage<BM, C::NW, true>(p.A2 + (size_t)g * p.K2, p.lda2, i_m0, p.M - 1, k2, la, wave, lane);
            stage<BN, C::NW, true>(p.B2, p.ldb2, i_n0, p.N - 1, k2, lb, wave, lane);
        }
        ++i_q;
        if (++i_ks == S) { i_ks = 0; ++i_ti; if (i_ti < n_my) tile_of(i_ti, i_m0, i_n0); }
    };

    f32x4 acc[C::TN][C::TM];
#pragma unroll
    for (int j = 0; j < C::TN; ++j)
#pragma unroll
        for (int i = 0; i < C::TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int EPI_STORES = C::TM * (16 / (64 / (WTN / 4))) * 2;   // upper bound of store instructions of one epilogue
    static_assert(LOADS + EPI_STORES <= 63, "vmcnt immediate range");

    issue_next();
    if (Q > 1) issue_next();
    int after_epi = 0;                                      // steps since an epilogue whose stores may still be in flight
    int ks = 0, ti = 0;
    for (int q = 0; q < Q; ++q) {
        // vmcnt counts loads AND stores in issue order.  Stage q must have landed; younger ops that may stay in flight:
        // stage q+1 (LOADS) and, for two steps after an epilogue, that epilogue's stores.
        if (q + 1 >= Q) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (after_epi > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS + EPI_STORES) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
        if (after_epi > 0) --after_epi;
        __builtin_amdgcn_s_barrier();                       // stage q landed everywhere; everyone is done with stage q-1
        asm volatile("" ::: "memory");
        if (q + 2 < Q) issue_next();
        {
            const char* la = smem + (q % NSTAGE) * STAGE;
            const char* lb = la + C::A_BYTES;
            const int nks = ks < nk ? 2 : 1;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (kk < nks) {
                    bf16x8 af[C::TM], wf[C::TN];
#pragma unroll
                    for (int i = 0; i < C::TM; ++i) {
                        const int row = wm * WTM + i * 16 + frow;
                        af[i] = *(const bf16x8*)(la + row * 128 + (swz(row, kk * 4 + fq) << 4));
                    }
#pragma unroll
                    for (int j = 0; j < C::TN; ++j) {
                        const int row = wn * WTN + j * 16 + frow;
                        wf[j] = *(const bf16x8*)(lb + row * 128 + (swz(row, kk * 4 + fq) << 4));
                    }
#pragma unroll
                    for (int j = 0; j < C::TN; ++j)
#pragma unroll
                        for (int i = 0; i < C::TM; ++i)
                            acc[j][i] = mfma16(wf[j], af[i], acc[j][i]);
                }
            }
        }
        if (++ks == S) {
            // tile finished: stage buffer q % NSTAGE is free until the issue after the next barrier -> staging area
            int m0, n0; tile_of(ti, m0, n0);
            __builtin_amdgcn_s_barrier();                   // every wave has finished its ds_reads of stage q
            asm volatile("" ::: "memory");
            static_assert(C::NW * 16 * (WTN * 4 + 16) <= STAGE, "epilogue staging does not fit in one stage buffer");
            store_tile<C::TM, C::TN>(p, acc, smem + (q % NSTAGE) * STAGE + wave * (16 * (WTN * 4 + 16)), m0 + wm * WTM,
                                     n0 + wn * WTN, lane);
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
#pragma unroll
                for (int i = 0; i < C::TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            ks = 0; ++ti; after_epi = 2;
        }
    }
}

template <int BM, int BN, int WM, int WN, int NSTAGE>
int launch_persist(GemmParams& p, hipStream_t s) {
    using C = Cfg<BM, BN, WM, WN>;
    constexpr int LDS = NSTAGE * (C::A_BYTES + C::B_BYTES);
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    static bool attr_set = false;
    static int n_cu = 256;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)mer_gemm_persist_kernel<BM, BN, WM, WN, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    const int total = p.tiles_m * p.tiles_n;
    const int blocks_per_cu = (160 * 1024) / LDS >= 2 ? 2 : 1;
    int grid = n_cu * blocks_per_cu;
    if (grid > total) grid = total;
    hipLaunchKernelGGL((mer_gemm_persist_kernel<BM, BN, WM, WN, NSTAGE>), dim3(grid), dim3(C::NT), LDS, s, p);
    REID_CHECK_LAUNCH("reid_mer_gemm(persistent)");
    return REID_OK;
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
    // (tall 256-row tiles leave a 50k-row problem with < 256 workgroups: 64-row tiles fill the chip)
    if (a->N <= 32) return a->M >= 65536 ? launch<256, 32, 4, 1>(p, s) : launch<64, 32, 4, 1>(p, s);
    if (a->N <= 64) return a->M >= 65536 ? launch<256, 64, 4, 1>(p, s) : launch<64, 64, 4, 1>(p, s);
    if (a->N <= 96) return launch<128, 32, 4, 1>(p, s);
    static int tile = -1;
    if (tile < 0) { const char* e = getenv("REID_GEMM_TILE"); tile = e ? atoi(e) : 0; }
    if (tile == 4) return launch_persist<256, 128, 4, 2, 3>(p, s);
    if (tile == 5) return launch_persist<128, 256, 2, 4, 3>(p, s);
    if (tile == 6) return launch_persist<128, 128, 2, 2, 4>(p, s);
    if (tile == 1) return launch<256, 128, 4, 2>(p, s);
    if (tile == 2) return launch<256, 256, 2, 4>(p, s);
    if (tile == 3) return launch<128, 128, 2, 2>(p, s);
    // default: 128x256 tile, 8 waves (best of the measured variants on the ViT shapes: tools/bench_gemm.py)
    // (a column tile must not straddle two LoRA groups of the fused q|k|v projection)
    if (tile == 0 && a->N >= 256 && a->M >= 128 && (p.k2_group_n == 0 || p.k2_group_n % 256 == 0)) return launch<128, 256, 2, 4>(p, s);
    return launch<128, 128, 2, 2>(p, s);
}
