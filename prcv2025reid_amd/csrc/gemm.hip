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
    int dbg;      // timing experiments only (REID_GEMM_DBG): 1 = skip the epilogue
};

using namespace gemmcore;

// Epilogue of one wave's [TM*16 x TN*16] sub-tile whose first element is C[m_base][n_base].
// The accumulator layout (lane = one output row, 4 consecutive columns per 16x16 sub-tile) would store 8-byte pieces
// scattered over 16 rows per instruction (measured: that store tail cost as much as the whole K loop at K = 768).
// Instead the wave transposes the sub-tile through its own LDS slice `stg`, 16 output rows at a time, and then works
// on ROW pieces of CW columns per lane (CW = 8 for 16-bit outputs, 4 for fp32 outputs, i.e. always 16-byte stores: the
// store tail is issue-bound, so halving the instruction count halves it -- cdna_hip_programming.md T21): the lanes of
// one instruction cover whole contiguous row segments, so bias / residual / aux loads and the C stores are full-line
// coalesced.  All global operands of the epilogue are requested BEFORE the LDS transposes start.
template <int CW>
struct Piece { float v[CW]; };

template <int CW>
__device__ __forceinline__ void store_piece(void* base, int dtype, size_t elem_off, const float (&v)[CW]) {
    if (dtype == REID_F32) {
#pragma unroll
        for (int q = 0; q < CW / 4; ++q) *(f32x4*)((float*)base + elem_off + 4 * q) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    } else if (CW == 8) {
        *(uint4*)((bf16_t*)base + elem_off) = uint4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    } else {
        *(uint2*)((bf16_t*)base + elem_off) = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
}
template <int CW>
__device__ __forceinline__ void load_piece(const void* base, int dtype, size_t elem_off, float (&v)[CW]) {
    if (dtype == REID_F32) {
#pragma unroll
        for (int q = 0; q < CW / 4; ++q) {
            const f32x4 t = *(const f32x4*)((const float*)base + elem_off + 4 * q);
            v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
        }
    } else if (CW == 8) {
        const bf16x8 t = *(const bf16x8*)((const bf16_t*)base + elem_off);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bf16_to_f32((bf16_t)t[e]);
    } else {
        const bf16x4 t = *(const bf16x4*)((const bf16_t*)base + elem_off);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = bf16_to_f32((bf16_t)t[e]);
    }
}

template <int TM, int TN, int CW>
__device__ __forceinline__ void store_tile_w(const GemmParams& p, f32x4 (&acc)[TN][TM], char* stg, int m_base, int n_base, int lane) {
    constexpr int WTN = TN * 16;
    constexpr int PITCH = WTN * 4 + 16;                 // bytes per staged row (+16: conflict-free 16-byte writes)
    constexpr int LPR = WTN / CW;                       // lanes per staged row
    constexpr int RPI = 64 / LPR;                       // rows per read instruction
    static_assert(64 % LPR == 0 && 16 % RPI == 0, "unsupported wave tile width");
    constexpr int IT = 16 / RPI;                        // read instructions per 16 staged rows
    constexpr bool PREF = (TM * IT * CW <= 64);         // prefetch residual / aux operands of the whole sub-tile (VGPR budget)
    const int mrow = lane & 15;
    const int ncol4 = (lane >> 4) * 4;
    const int rr = lane / LPR, rc = (lane % LPR) * CW;
    const int n = n_base + rc;                          // this lane's CW output columns: the same for every row it handles
    const bool nok = n < p.N;
    const int nc = nok ? n : 0;                         // clamped: operand loads are unconditional (no branch, no wait between them)
    const int mlast = p.M - 1;
    float bv[CW];
#pragma unroll
    for (int e = 0; e < CW; ++e) bv[e] = 0.f;
    if (p.bias) load_piece<CW>(p.bias, REID_F32, nc, bv);
    // LoRA routing mask: which of this lane's columns belong to which modality is a per-lane constant
    int colmod[CW];
    if (p.mask_r > 0) {
#pragma unroll
        for (int e = 0; e < CW; ++e) colmod[e] = ((nc + e) % p.mask_period) / p.mask_r;
    }
    Piece<CW> rv[PREF ? TM : 1][PREF ? IT : 1], av[PREF ? TM : 1][PREF ? IT : 1];
    auto r_off = [&](int m) -> size_t {
        const int mc = m < mlast ? m : mlast;
        return (size_t)(p.r_period > 0 ? mc % p.r_period : mc) * p.ldr + nc;
    };
    if (PREF) {
        if (p.R) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int it = 0; it < IT; ++it) load_piece<CW>(p.R, p.r_dtype, r_off(m_base + i * 16 + it * RPI + rr), rv[i][it].v);
        }
        if (p.act >= REID_ACT_DGELU_ERF) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int m = m_base + i * 16 + it * RPI + rr;
                    load_piece<CW>(p.aux, REID_BF16, (size_t)(m < mlast ? m : mlast) * p.ldaux + nc, av[i][it].v);
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
            const int m = m_base + i * 16 + r;
            float v[CW];
#pragma unroll
            for (int q = 0; q < CW / 4; ++q) {
                const f32x4 t = *(const f32x4*)(stg + r * PITCH + (rc + 4 * q) * 4);
                v[4 * q] = t[0] + bv[4 * q]; v[4 * q + 1] = t[1] + bv[4 * q + 1];
                v[4 * q + 2] = t[2] + bv[4 * q + 2]; v[4 * q + 3] = t[3] + bv[4 * q + 3];
            }
            if (p.R) {
                if (PREF) {
#pragma unroll
                    for (int e = 0; e < CW; ++e) v[e] += rv[PREF ? i : 0][PREF ? it : 0].v[e];
                } else {
                    float t[CW];
                    load_piece<CW>(p.R, p.r_dtype, r_off(m), t);
#pragma unroll
                    for (int e = 0; e < CW; ++e) v[e] += t[e];
                }
            }
            const bool ok = m < p.M && nok;
            size_t crow = (size_t)m;
            if (p.c_group > 0) crow = (size_t)(m / p.c_group) * p.c_group_stride + (m % p.c_group) + p.c_row_off;
            if (p.C2 && ok) store_piece<CW>(p.C2, p.c2_dtype, crow * p.ldc2 + n, v);
            if (p.act != REID_ACT_NONE) {
                if (p.act == REID_ACT_GELU_ERF) {           // (uniform branches: only the selected activation is evaluated)
#pragma unroll
                    for (int e = 0; e < CW; ++e) v[e] = gelu_erf_f(v[e]);
                } else if (p.act == REID_ACT_QUICK_GELU) {
#pragma unroll
                    for (int e = 0; e < CW; ++e) v[e] = quick_gelu_f(v[e]);
                } else if (p.act == REID_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < CW; ++e) v[e] = fmaxf(v[e], 0.f);
                } else {
                    float u[CW];
                    if (PREF) {
#pragma unroll
                        for (int e = 0; e < CW; ++e) u[e] = av[PREF ? i : 0][PREF ? it : 0].v[e];
                    } else {
                        load_piece<CW>(p.aux, REID_BF16, (size_t)(m < mlast ? m : mlast) * p.ldaux + nc, u);
                    }
                    if (p.act == REID_ACT_DGELU_ERF) {
#pragma unroll
                        for (int e = 0; e < CW; ++e) v[e] *= dgelu_erf_f(u[e]);
                    } else if (p.act == REID_ACT_DQUICK_GELU) {
#pragma unroll
                        for (int e = 0; e < CW; ++e) v[e] *= dquick_gelu_f(u[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < CW; ++e) v[e] *= (u[e] > 0.f ? 1.f : 0.f);
                    }
                }
            }
            if (p.mask_r > 0) {
                const int modality = p.img_mod[(m < mlast ? m : mlast) / p.rows_per_img];
#pragma unroll
                for (int e = 0; e < CW; ++e)
                    if (colmod[e] != modality) v[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < CW; ++e) v[e] *= p.alpha;
            if (ok) store_piece<CW>(p.C, p.c_dtype, crow * p.ldc + n, v);
        }
        asm volatile("" ::: "memory");
    }
}

template <int TM, int TN>
__device__ __forceinline__ void store_tile(const GemmParams& p, f32x4 (&acc)[TN][TM], char* stg, int m_base, int n_base, int lane) {
    // 16-byte stores in both cases; N % 8 == 0 is needed for the 8-wide form
    const bool wide = p.c_dtype != REID_F32 && ((p.N | p.ldc) & 7) == 0 && (!p.C2 || (p.ldc2 & 7) == 0) &&
                      (!p.R || (p.ldr & 7) == 0) && (!p.aux || (p.ldaux & 7) == 0);
    if (wide) store_tile_w<TM, TN, 8>(p, acc, stg, m_base, n_base, lane);
    else store_tile_w<TM, TN, 4>(p, acc, stg, m_base, n_base, lane);
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
    if (p.dbg != 1) store_tile<C::TM, C::TN>(p, acc, smem + wave * (16 * (WTN * 4 + 16)), m0 + wm * WTM, n0 + wn * WTN, lane);
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
    // issue pointer (runs two steps ahead of the compute pointer); tile coordinates and the per-lane load offsets are
    // recomputed only per tile
    int i_ti = 0, i_ks = 0, i_q = 0, i_m0, i_n0;
    uint32_t offA[C::A_INSTR], offB[C::B_INSTR];
    auto new_tile = [&]() {
        tile_of(i_ti, i_m0, i_n0);
        stage_offsets<BM, C::NW, false>(p.lda, i_m0, p.M - 1, wave, lane, offA);
        stage_offsets<BN, C::NW, false>(p.ldb, i_n0, p.N - 1, wave, lane, offB);
    };
    new_tile();
    auto issue_next = [&]() {
        char* la = smem + (i_q % NSTAGE) * STAGE;
        char* lb = la + C::A_BYTES;
        if (i_ks < nk) {
            stage_from<BM, C::NW>((const char*)p.A + (size_t)i_ks * 128, offA, la, wave);
            stage_from<BN, C::NW>((const char*)p.B + (size_t)i_ks * 128, offB, lb, wave);
        } else {
            const int g = (p.k2_group_n > 0) ? (i_n0 / p.k2_group_n) : 0;
            const int k2 = (i_ks - nk) << 5;
            stage<BM, C::NW, true>(p.A2 + (size_t)g * p.K2, p.lda2, i_m0, p.M - 1, k2, la, wave, lane);
            stage<BN, C::NW, true>(p.B2, p.ldb2, i_n0, p.N - 1, k2, lb, wave, lane);
        }
        ++i_q;
        if (++i_ks == S) { i_ks = 0; ++i_ti; if (i_ti < n_my) new_tile(); }
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
            if (p.dbg != 1)
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
    REID_CHECK_ARG((int64_t)a->M * a->lda * 2 < (1ll << 32) && (int64_t)a->N * a->ldb * 2 < (1ll << 32),
                   "reid_mer_gemm: operands beyond 4 GiB need 64-bit lane offsets (M*lda=%lld)", (long long)a->M * a->lda);
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
    // experiment knobs (read per call so a benchmark can A/B tiles inside one process)
    const char* e_tile = getenv("REID_GEMM_TILE");
    const char* e_dbg = getenv("REID_GEMM_DBG");
    const int tile = e_tile ? atoi(e_tile) : 0;
    p.dbg = e_dbg ? atoi(e_dbg) : 0;
    if (tile == 4) return launch_persist<256, 128, 4, 2, 3>(p, s);
    if (tile == 5) return launch_persist<128, 256, 2, 4, 3>(p, s);
    if (tile == 6) return launch_persist<128, 128, 2, 2, 4>(p, s);
    if (tile == 1) return launch<256, 128, 4, 2>(p, s);
    if (tile == 2) return launch<256, 256, 2, 4>(p, s);
    if (tile == 3) return launch<128, 128, 2, 2>(p, s);
    if (tile == 8) return launch<128, 256, 2, 4>(p, s);
    // defaults from same-process A/B runs of the seven ViT GEMM variants (tools/bench_gemm_variants.py):
    //   256x256 tile where the epilogue has no per-element global operand (no residual, no saved pre-activation), else
    //   128x256 (its epilogue prefetches residual / aux for the whole sub-tile).  A column tile must not straddle two
    //   LoRA groups of the fused q|k|v projection.
    const bool grp_ok = p.k2_group_n == 0 || p.k2_group_n % 256 == 0;
    if (tile == 0 && grp_ok && a->N >= 512 && a->M >= 512 && !a->R && a->act < REID_ACT_DGELU_ERF) return launch<256, 256, 2, 4>(p, s);
    if (tile == 0 && grp_ok && a->N >= 256 && a->M >= 128) return launch<128, 256, 2, 4>(p, s);
    return launch<128, 128, 2, 2>(p, s);
}
