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
#include <type_traits>

namespace {

struct GemmParams {
    const bf16_t* A; const bf16_t* B; const bf16_t* A2; const bf16_t* B2;
    const float* bias; const void* R; const bf16_t* aux;
    void* C; void* C2;
    const int32_t* img_mod;
    const float* row_scale;
    int M, N, K, K2;
    int lda, ldb, lda2, ldb2, ldr, ldaux, ldc, ldc2;
    int k2_group_n;
    int act, c_dtype, c2_dtype, r_dtype;
    int r_period;
    int mask_r, mask_period, rows_per_img;
    int c_group, c_group_stride, c_row_off;
    float alpha;
    int tiles_m, tiles_n;
    int group_m;  // row tiles per L2 group of the tile order (gemm_core.h tile_coords)
    int stagger;  // ping-pong kernel: spread (in 0.25 us units) of the start times of the first round of workgroups
    int aux_pre;  // multiply-by-derivative epilogue on the 256-row tile: aux tile staged inside the K loop (REID_GELU_IMPL=1: after it, the r03 form)
    int perm_b;   // weight rows staged through perm32() (16-bit C with 16-byte pieces)
    int dbg;      // timing experiments only (REID_GEMM_DBG): 1 = skip the epilogue, 4 = skip the K loop (epilogue only)
    int epi;      // EPI_*: which epilogue the kernel instance was built with (host side choice)
    // Row groups (n_groups > 0): activation rows [grp_row0[g], grp_row_end[g]) multiply weight matrix number grp_b[g] of a stack of
    // [N, K] matrices b_group_stride elements apart (the per-modality merged weights W + s B_mu A_mu: images are packed modality by
    // modality).  Row tiles never straddle groups: group g owns global row-tile indices [grp_tile0[g], grp_tile0[g + 1]).
    int n_groups;
    int grp_row0[REID_GEMM_MAX_GROUPS], grp_row_end[REID_GEMM_MAX_GROUPS], grp_tile0[REID_GEMM_MAX_GROUPS], grp_b[REID_GEMM_MAX_GROUPS];
    long b_group_stride;
    unsigned long long* trace;   // REID_GEMM_TRACE builds: 8 words per workgroup (timestamps at start / loop end / epilogue issued / acknowledged, HW ids)
};

// Epilogue kinds.  The GENERIC epilogue evaluates every option of reid_mer_gemm at run time (~100-600 instructions per 16-byte
// piece, ~1 000 scalar branches in the kernel): it is instruction-issue bound and held the C / residual traffic of the big GEMMs at
// 3-4 TB/s where plain 16-byte stores of the same shape reach 7 TB/s (r02: tools/store_patterns.hip, REID_GEMM_DBG=4).  The four
// kinds below cover the seven ViT GEMM variants of a training step with straight-line code (32-bit offsets, one predicate per
// row, no option tests); everything else (patch embed row remap, routing masks, alpha, quick-GELU, ...) stays on GENERIC.
// GELU2D / MULAUX (r02): the training path saves gelu'(u) in the forward epilogue -- it shares the erfc and exp of the value -- so
// the backward epilogue is one multiply per element instead of ~23 VALU operations (epilogues of this family are VALU-bound at
// two waves per SIMD: ~12 us per 256 x 256 tile, tools/exp_epilogue_scale.py).
// PLAIN16H (r04, bf16 flavor only): the plain 16-bit store in IEEE half (REID_F16 output) -- the residual-branch outputs (out-projection,
// fc2) are consumed by the add + LayerNorm kernel, never by an MFMA, so they can carry half's 11 significant bits instead of bf16's 8.
enum { EPI_GENERIC = 0, EPI_PLAIN16 = 1, EPI_RES32 = 2, EPI_GELU2 = 3, EPI_DGELU = 4, EPI_GELU2D = 5, EPI_MULAUX = 6, EPI_PLAIN16H = 7 };

using namespace gemmcore;

// first row, row limit and weight matrix of global row tile `tm` (tile height BM)
__device__ __forceinline__ void tile_rows(const GemmParams& p, int tm, int BM, int& m0, int& m_end, const bf16_t*& B) {
    B = p.B;
    if (p.n_groups > 0) {
        int g = 0;
#pragma unroll
        for (int i = 1; i < REID_GEMM_MAX_GROUPS; ++i)
            if (i < p.n_groups && tm >= p.grp_tile0[i]) g = i;
        m0 = p.grp_row0[g] + (tm - p.grp_tile0[g]) * BM;
        m_end = p.grp_row_end[g];
        B += (size_t)p.grp_b[g] * p.b_group_stride;
    } else {
        m0 = tm * BM;
        m_end = p.M;
    }
}

// Epilogue of one wave's [TM*16 x TN*16] sub-tile whose first element is C[m_base][n_base], straight from the
// accumulators (no LDS round trip, no waits between pieces).
// A lane owns output row m = lane & 15 of each 16-row group and, per group, NP pieces of CW consecutive columns:
//   MODE 0 (fp32 C): CW = 4, piece pc = MFMA sub-tile pc: columns 16 pc + 4 fq .. +3   (natural accumulator layout)
//   MODE 1 (16-bit C, weight rows staged through perm32()): CW = 8, columns 8 (4 pc + fq) .. +7
//   MODE 2 (16-bit C whose strides are not multiples of 8): as MODE 0 with 8-byte stores
// In modes 0/1 every load/store is 16 bytes per lane and the four lanes fq = 0..3 of one row are adjacent: 64 contiguous
// bytes per output row per instruction, full 64-byte requests (an earlier version without the permutation stored 8-byte
// pieces scattered over 16 rows and cost as much as the K loop at K = 768; the LDS-transposing version that replaced it
// chained LDS write -> wait -> read -> store per 16 rows and ran at ~12 GB/s per CU).
// Residual / saved-pre-activation operands are prefetched RD pieces ahead (a register ring, indices static after
// unrolling); all operand loads are unconditional with clamped addresses (no branch, no wait between them).
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

// column of accumulator element (sub-tile j, register r) of lane-quarter fq inside the wave's sub-tile
template <int MODE>
__device__ __forceinline__ int acc_col(int j, int r, int fq) {
    return MODE == 1 ? (((j >> 1) * 4 + fq) * 8 + (j & 1) * 4 + r) : (j * 16 + fq * 4 + r);
}

// acc := bias (the epilogue then never touches the bias: no registers, no adds)
template <int TM, int TN, int MODE>
__device__ __forceinline__ void init_acc_m(const GemmParams& p, f32x4 (&acc)[TN][TM], int n_base, int lane) {
    const int fq = lane >> 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        f32x4 b = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
            int n = n_base + acc_col<MODE>(j, 0, fq);
            n = n + 3 < p.N ? n : 0;                    // N % 4 == 0: a 4-column run is inside or outside as a whole
            b = *(const f32x4*)(p.bias + n);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = b;
    }
}
template <int TM, int TN>
__device__ __forceinline__ void init_acc(const GemmParams& p, f32x4 (&acc)[TN][TM], int n_base, int lane) {
    if (p.c_dtype != REID_F32 && p.perm_b) init_acc_m<TM, TN, 1>(p, acc, n_base, lane);
    else init_acc_m<TM, TN, 0>(p, acc, n_base, lane);
}

template <int TM, int TN, int MODE>
__device__ __forceinline__ void store_tile_m(const GemmParams& p, f32x4 (&acc)[TN][TM], int m_base, int n_base, int lane, int m_end) {
    constexpr int CW = MODE == 1 ? 8 : 4;
    constexpr int NP = TN * 4 / CW;                     // pieces per lane per 16-row group
    constexpr int NQ = TM * NP;                         // pieces per lane
    constexpr int RD = (32 / CW) < NQ ? (32 / CW) : NQ; // residual / aux pieces in flight per lane (32 VGPRs of fp32)
    static_assert(MODE != 1 || TN % 2 == 0, "8-wide pieces pair two MFMA sub-tiles");
    const int frow = lane & 15, fq = lane >> 4;
    const int mlast = m_end - 1;
    const bool has_r = p.R != nullptr;
    const bool has_aux = p.act >= REID_ACT_DGELU_ERF && p.act <= REID_ACT_MUL_AUX;
    Piece<CW> rv[RD];
    typedef typename std::conditional<CW == 8, bf16x8, bf16x4>::type aux_t;   // saved pre-activations stay packed
    aux_t av[RD];
    auto col_of = [&](int pc) { return n_base + (MODE == 1 ? (pc * 4 + fq) * 8 : pc * 16 + fq * 4); };
    auto fetch = [&](int q, int slot) {
        const int i = q / NP, pc = q % NP;
        const int m = m_base + i * 16 + frow;
        const int mc = m < mlast ? m : mlast;
        const int n = col_of(pc);
        const int nc = n < p.N ? n : 0;
        if (has_r) load_piece<CW>(p.R, p.r_dtype, (size_t)(p.r_period > 0 ? mc % p.r_period : mc) * p.ldr + nc, rv[slot].v);
        if (has_aux) av[slot] = *(const aux_t*)(p.aux + (size_t)mc * p.ldaux + nc);
    };
    if (has_r || has_aux) {
#pragma unroll
        for (int q = 0; q < RD; ++q) fetch(q, q);
    }
#pragma clang loop unroll(full)
    for (int q = 0; q < NQ; ++q) {
        const int i = q / NP, pc = q % NP, slot = q % RD;
        const int m = m_base + i * 16 + frow;
        const int n = col_of(pc);
        const bool ok = m < m_end && n < p.N;
        size_t crow = (size_t)m;
        if (p.c_group > 0) crow = (size_t)(m / p.c_group) * p.c_group_stride + (m % p.c_group) + p.c_row_off;
        float v[CW];
#pragma unroll
        for (int e = 0; e < CW; ++e) v[e] = acc[MODE == 1 ? 2 * pc + (e >> 2) : pc][i][e & 3];
        if (p.row_scale) {                                   // DropPath: the whole branch output of this sample, before the residual
            const float rs = p.row_scale[(m < mlast ? m : mlast) / p.rows_per_img];
#pragma unroll
            for (int e = 0; e < CW; ++e) v[e] *= rs;
        }
        if (has_r) {
#pragma unroll
            for (int e = 0; e < CW; ++e) v[e] += rv[slot].v[e];
        }
        if (p.act == REID_ACT_GELU_ERF_DSAVE) {             // value to C, DERIVATIVE (not the pre-activation) to C2
            float dv[CW];
#pragma unroll
            for (int e = 0; e < CW; ++e) gelu_both_f(v[e], v[e], dv[e]);
            if (p.C2 && ok) store_piece<CW>(p.C2, p.c2_dtype, crow * p.ldc2 + n, dv);
        } else if (p.C2 && ok) store_piece<CW>(p.C2, p.c2_dtype, crow * p.ldc2 + n, v);
        if (p.act != REID_ACT_NONE && p.act != REID_ACT_GELU_ERF_DSAVE) {
            if (p.act == REID_ACT_MUL_AUX) {
#pragma unroll
                for (int e = 0; e < CW; ++e) v[e] *= bf16_to_f32((bf16_t)av[slot][e]);
            } else if (p.act == REID_ACT_GELU_ERF) {        // (uniform branches: only the selected activation is evaluated)
#pragma unroll
                for (int e = 0; e < CW; ++e) v[e] = gelu_erf_f(v[e]);
            } else if (p.act == REID_ACT_QUICK_GELU) {
#pragma unroll
                for (int e = 0; e < CW; ++e) v[e] = quick_gelu_f(v[e]);
            } else if (p.act == REID_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < CW; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (p.act == REID_ACT_DGELU_ERF) {
#pragma unroll
                for (int e = 0; e < CW; ++e) v[e] *= dgelu_erf_f(bf16_to_f32((bf16_t)av[slot][e]));
            } else if (p.act == REID_ACT_DQUICK_GELU) {
#pragma unroll
                for (int e = 0; e < CW; ++e) v[e] *= dquick_gelu_f(bf16_to_f32((bf16_t)av[slot][e]));
            } else {
#pragma unroll
                for (int e = 0; e < CW; ++e) v[e] *= (bf16_to_f32((bf16_t)av[slot][e]) > 0.f ? 1.f : 0.f);
            }
        }
        if (p.mask_r > 0) {
            const int modality = p.img_mod[(m < mlast ? m : mlast) / p.rows_per_img];
            const int nc = n < p.N ? n : 0;
#pragma unroll
            for (int e = 0; e < CW; ++e)
                if (((nc + e) % p.mask_period) / p.mask_r != modality) v[e] = 0.f;
        }
        if (p.alpha != 1.f) {
#pragma unroll
            for (int e = 0; e < CW; ++e) v[e] *= p.alpha;
        }
        if (ok) store_piece<CW>(p.C, p.c_dtype, crow * p.ldc + n, v);
        if ((has_r || has_aux) && q + RD < NQ) fetch(q + RD, slot);
    }
}

// Output stores of the lean epilogues.  -DREID_NT_STORES: non-temporal (the tile is written once and never re-read by this kernel, so it
// need not displace the operand panels the XCD's other workgroups are re-reading from L2).
template <typename T>
__device__ __forceinline__ void st_out(T* ptr, const T& v) {
#ifdef REID_NT_STORES
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    static_assert(sizeof(T) == 16, "16-byte pieces");
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, v), (u32x4_t*)ptr);
#else
    *ptr = v;
#endif
}

// Lean epilogues (host side guarantees: N a multiple of the tile, every stride a multiple of 8 elements, every operand below
// 4 GiB, alpha == 1, no mask / row remap / periodic residual).  Layout as in store_tile_m: the 16-bit kinds use the perm32
// weight-row staging (a lane owns 8 consecutive columns per pair of MFMA sub-tiles), EPI_RES32 the natural one (4 columns).
// aux_lds != nullptr (256 x 256 ping-pong kernel, MULAUX): the tile of the aux operand sits in LDS as [256 rows][512 B] with the
// 16-byte chunks of row r stored at position chunk ^ (r & 15) (stage_aux_tile below); lrow0 / lchunk0 = this wave's first row /
// first chunk in that image.
template <int TM, int TN, int EPI>
__device__ __forceinline__ void store_tile_fast(const GemmParams& p, f32x4 (&acc)[TN][TM], int m_base, int n_base, int lane, int m_end,
                                                const char* aux_lds = nullptr, int lrow0 = 0, int lchunk0 = 0) {
    const int frow = lane & 15, fq = lane >> 4;
    // Operand loads (residual, saved derivative) are ALL issued before the first use, half a tile at a time: one exposed memory
    // round trip per half instead of one per 16-row group (the per-group form cost the multiply-by-aux epilogue 9 of its 15 us per
    // tile -- eight dependent load -> use steps per wave -- although it moves the same bytes as the plain 16-bit epilogue's stores).
    if constexpr (EPI == EPI_RES32) {
        const uint32_t col = (uint32_t)(n_base + 4 * fq);
        char* C = (char*)p.C; const char* R = (const char*)p.R;
        constexpr int HG = (TM + 1) / 2;                                             // 16-row groups per half (the second may be one short)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 r[HG][TN];
            float rs[HG];
#pragma unroll
            for (int ii = 0; ii < HG; ++ii) {
                if (half * HG + ii >= TM) break;
                const int m = m_base + 16 * (half * HG + ii) + frow;
                const int mc = m < m_end ? m : m_end - 1;
                const uint32_t ro = ((uint32_t)mc * (uint32_t)p.ldr + col) * 4u;
#pragma unroll
                for (int j = 0; j < TN; ++j) r[ii][j] = *(const f32x4*)(R + ro + 64u * j);
                rs[ii] = p.row_scale ? p.row_scale[mc / p.rows_per_img] : 1.f;       // DropPath: the branch output of this sample
            }
#pragma unroll
            for (int ii = 0; ii < HG; ++ii) {
                const int i = half * HG + ii;
                if (i >= TM) break;
                const int m = m_base + 16 * i + frow;
                const bool ok = m < m_end;
                const uint32_t co = ((uint32_t)(ok ? m : m_end - 1) * (uint32_t)p.ldc + col) * 4u;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const f32x4 v = acc[j][i] * rs[ii] + r[ii][j];
                    if (ok) st_out((f32x4*)(C + co + 64u * j), v);
                }
            }
        }
    } else {
        constexpr int NP = TN / 2;                                                   // 8-column pieces per 16-row group
        constexpr bool HAS_AUX = EPI == EPI_DGELU || EPI == EPI_MULAUX;
        const uint32_t col = (uint32_t)(n_base + 8 * fq);
        char* C = (char*)p.C;
        bf16x8 av[HAS_AUX ? TM : 1][NP];
        if (HAS_AUX && aux_lds) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int lr = lrow0 + 16 * i + frow;
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) av[i][pc] = *(const bf16x8*)(aux_lds + lr * 512 + (((lchunk0 + fq + 4 * pc) ^ (lr & 15)) << 4));
            }
        } else if constexpr (HAS_AUX) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m_base + 16 * i + frow;
                const int mc = m < m_end ? m : m_end - 1;
                const uint32_t ao = ((uint32_t)mc * (uint32_t)p.ldaux + col) * 2u;
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) av[i][pc] = *(const bf16x8*)((const char*)p.aux + ao + 64u * pc);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m_base + 16 * i + frow;
            const bool ok = m < m_end;
            const int mc = ok ? m : m_end - 1;
            const uint32_t co = ((uint32_t)mc * (uint32_t)p.ldc + col) * 2u;
            const uint32_t c2o = (EPI == EPI_GELU2 || EPI == EPI_GELU2D) ? ((uint32_t)mc * (uint32_t)p.ldc2 + col) * 2u : 0u;
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[2 * pc + (e >> 2)][i][e & 3];
                if constexpr (EPI == EPI_GELU2) {
                    if (ok) st_out((uint4*)((char*)p.C2 + c2o + 64u * pc), uint4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])});
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        f32x2_t gg, dd;
                        gelu_both_x2(f32x2_t{v[e], v[e + 1]}, gg, dd);
                        v[e] = gg.x; v[e + 1] = gg.y;
                    }
                } else if constexpr (EPI == EPI_GELU2D) {
                    float dv[8];
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        f32x2_t gg, dd;
                        gelu_both_x2(f32x2_t{v[e], v[e + 1]}, gg, dd);
                        v[e] = gg.x; v[e + 1] = gg.y; dv[e] = dd.x; dv[e + 1] = dd.y;
                    }
                    if (ok) st_out((uint4*)((char*)p.C2 + c2o + 64u * pc), uint4{pack_bf16x2(dv[0], dv[1]), pack_bf16x2(dv[2], dv[3]), pack_bf16x2(dv[4], dv[5]), pack_bf16x2(dv[6], dv[7])});
                } else if constexpr (EPI == EPI_DGELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= dgelu_erf_f(bf16_to_f32((bf16_t)av[i][pc][e]));
                } else if constexpr (EPI == EPI_MULAUX) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= bf16_to_f32((bf16_t)av[i][pc][e]);
                }
                if constexpr (EPI == EPI_PLAIN16H) {
                    if (ok) st_out((uint4*)(C + co + 64u * pc), uint4{pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3]), pack_f16x2(v[4], v[5]), pack_f16x2(v[6], v[7])});
                } else {
                    if (ok) st_out((uint4*)(C + co + 64u * pc), uint4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])});
                }
            }
        }
    }
}

// host side: number of row tiles of height bm (per row group when groups are on) and, with `fill`, the tile index ranges of the groups
static int row_tiles(GemmParams& p, int bm, bool fill) {
    if (p.n_groups <= 0) return (p.M + bm - 1) / bm;
    int t = 0;
    for (int g = 0; g < p.n_groups; ++g) {
        if (fill) p.grp_tile0[g] = t;
        t += (p.grp_row_end[g] - p.grp_row0[g] + bm - 1) / bm;
    }
    return t;
}

// host side: which lean epilogue (if any) covers this launch
static int pick_epilogue(const GemmParams& p, int BN) {
    if (p.alpha != 1.f || p.mask_r > 0 || p.c_group > 0 || p.r_period > 0 || p.N % BN != 0) return EPI_GENERIC;
    const long mx = (long)p.M * (p.ldc > p.ldr ? p.ldc : p.ldr);
    if (mx * 4 >= (1L << 32) || (long)p.M * p.ldaux * 2 >= (1L << 32) || (long)p.M * p.ldc2 * 2 >= (1L << 32)) return EPI_GENERIC;
    if (p.c_dtype == REID_F32) {
        if (p.R && p.r_dtype == REID_F32 && !p.C2 && !p.aux && p.act == REID_ACT_NONE && (p.ldc & 3) == 0 && (p.ldr & 3) == 0) return EPI_RES32;
        return EPI_GENERIC;
    }
    if (!p.perm_b || p.R || p.row_scale) return EPI_GENERIC;
    if (p.act == REID_ACT_NONE && !p.C2 && !p.aux) return p.c_dtype == REID_F16 ? EPI_PLAIN16H : EPI_PLAIN16;
    if (p.act == REID_ACT_GELU_ERF && p.C2 && p.c2_dtype != REID_F32 && !p.aux) return EPI_GELU2;
    if (p.act == REID_ACT_DGELU_ERF && p.aux && !p.C2) return EPI_DGELU;
    if (p.act == REID_ACT_GELU_ERF_DSAVE && p.C2 && p.c2_dtype != REID_F32 && !p.aux) return EPI_GELU2D;
    if (p.act == REID_ACT_MUL_AUX && p.aux && !p.C2) return EPI_MULAUX;
    return EPI_GENERIC;
}

// 16-byte pieces need 16-byte aligned rows in every epilogue operand; perm_b (host side) must match the MODE chosen here
__host__ __device__ inline bool epilogue_wide16(const GemmParams& p) {
    return p.c_dtype != REID_F32 && ((p.N | p.ldc) & 7) == 0 && (!p.C2 || (p.ldc2 & 7) == 0) && (!p.R || (p.ldr & 7) == 0) &&
           (!p.aux || (p.ldaux & 7) == 0);
}

template <int TM, int TN>
__device__ __forceinline__ void store_tile(const GemmParams& p, f32x4 (&acc)[TN][TM], int m_base, int n_base, int lane, int m_end) {
    if (p.c_dtype == REID_F32) store_tile_m<TM, TN, 0>(p, acc, m_base, n_base, lane, m_end);
    else if (p.perm_b) store_tile_m<TM, TN, 1>(p, acc, m_base, n_base, lane, m_end);
    else store_tile_m<TM, TN, 2>(p, acc, m_base, n_base, lane, m_end);
}

template <int BM, int BN, int WM, int WN, int EPI = EPI_GENERIC>
__global__ __launch_bounds__(WM* WN * 64, 2) void mer_gemm_kernel(const GemmParams p) {
    REID_T16_ENTER();
    using C = Cfg<BM, BN, WM, WN>;
    if constexpr (EPI == EPI_PLAIN16H) REID_F16_SATURATE();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware, L2-aware tile order (gemm_core.h)
    const int lin = xcd_linear_block(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(lin, p.tiles_m, p.tiles_n, tm, tn, p.group_m);
    int m0, m_end;
    const bf16_t* Bw;
    tile_rows(p, tm, BM, m0, m_end, Bw);
    const int n0 = tn * BN;
    const int g = (p.k2_group_n > 0) ? (n0 / p.k2_group_n) : 0;
    const bf16_t* A2 = p.A2 ? p.A2 + (size_t)g * p.K2 : nullptr;

    f32x4 acc[C::TN][C::TM];
    if constexpr (EPI == EPI_GENERIC) init_acc<C::TM, C::TN>(p, acc, n0 + wn * (BN / WN), lane);
    else if constexpr (EPI == EPI_RES32) init_acc_m<C::TM, C::TN, 0>(p, acc, n0 + wn * (BN / WN), lane);
    else init_acc_m<C::TM, C::TN, 1>(p, acc, n0 + wn * (BN / WN), lane);
    if (REID_DBG(p) != 4)
        mainloop<BM, BN, WM, WN>(p.A, p.lda, Bw, p.ldb, A2, p.lda2, p.B2, p.ldb2, m_end, p.N, p.K, p.K2, m0, n0, smem, acc,
                                 p.perm_b != 0);

    // ------------------------------------------------------------------ epilogue (registers -> global, no LDS)
    constexpr int WTM = BM / WM, WTN = BN / WN;
    if (REID_DBG(p) == 1) return;
    if constexpr (EPI == EPI_GENERIC) store_tile<C::TM, C::TN>(p, acc, m0 + wm * WTM, n0 + wn * WTN, lane, m_end);
    else store_tile_fast<C::TM, C::TN, EPI>(p, acc, m0 + wm * WTM, n0 + wn * WTN, lane, m_end);
}

// 256 x 256 tile with the wave-row ping-pong K loop of gemm_core.h (mainloop_pp); epilogue = the same register-direct code
template <int EPI, int BM = 256>
__global__ __launch_bounds__(512, 2) void mer_gemm_pp_kernel(const GemmParams p) {
    REID_T16_ENTER();
    if constexpr (EPI == EPI_PLAIN16H) REID_F16_SATURATE();
    using PC = PPCfg<BM, 256>;
    constexpr int TM = PC::TM, RW = PC::RW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
#ifdef REID_GEMM_TRACE
#define GEMM_TRACE(slot) do { if (p.trace && threadIdx.x == 0) p.trace[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    if (p.trace && threadIdx.x == 0) p.trace[(size_t)blockIdx.x * 8 + 6] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
#else
#define GEMM_TRACE(slot) do { } while (0)
#endif
    GEMM_TRACE(0);
    if (p.stagger > 0 && blockIdx.x < 256) {
        // first round only: CU c of every XCD starts c/32 of the spread late, so the epilogues of the chip's 256 tiles in flight
        // do not all hit HBM in the same few microseconds (see launch_pp)
        const int n = ((blockIdx.x >> 3) & 31) * p.stagger >> 5;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);            // 8 x 64 cycles ~ 0.25 us
    }
    const int lin = xcd_linear_block(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(lin, p.tiles_m, p.tiles_n, tm, tn, p.group_m);
    int m0, m_end;
    const bf16_t* Bw;
    tile_rows(p, tm, BM, m0, m_end, Bw);
    const int n0 = tn * 256;
    const int g = (p.k2_group_n > 0) ? (n0 / p.k2_group_n) : 0;
    const bf16_t* A2 = p.A2 ? p.A2 + (size_t)g * p.K2 : nullptr;
    f32x4 acc[4][TM];
    if constexpr (EPI == EPI_GENERIC) init_acc<TM, 4>(p, acc, n0 + wn * 64, lane);
    else if constexpr (EPI == EPI_RES32) init_acc_m<TM, 4, 0>(p, acc, n0 + wn * 64, lane);
    else init_acc_m<TM, 4, 1>(p, acc, n0 + wn * 64, lane);
#ifdef REID_GEMM_ABLATIONS                                  // K-loop anatomy builds (profiles/r02_gemm_variants10*.log); not in the shipped library
    if (EPI == EPI_PLAIN16 && BM == 256 && REID_DBG(p) >= 16) {
        switch (REID_DBG(p) - 16) {
            case 1: mainloop_pp<BM, 256, 1>(p.A, p.lda, Bw, p.ldb, A2, p.lda2, p.B2, p.ldb2, m_end, p.N, p.K, p.K2, m0, n0, smem, acc, p.perm_b != 0); break;
            case 2: mainloop_pp<BM, 256, 2>(p.A, p.lda, Bw, p.ldb, A2, p.lda2, p.B2, p.ldb2, m_end, p.N, p.K, p.K2, m0, n0, smem, acc, p.perm_b != 0); break;
            case 3: mainloop_pp<BM, 256, 3>(p.A, p.lda, Bw, p.ldb, A2, p.lda2, p.B2, p.ldb2, m_end, p.N, p.K, p.K2, m0, n0, smem, acc, p.perm_b != 0); break;
            case 4: mainloop_pp<BM, 256, 4>(p.A, p.lda, Bw, p.ldb, A2, p.lda2, p.B2, p.ldb2, m_end, p.N, p.K, p.K2, m0, n0, smem, acc, p.perm_b != 0); break;
            case 6: mainloop_pp<BM, 256, 6>(p.A, p.lda, Bw, p.ldb, A2, p.lda2, p.B2, p.ldb2, m_end, p.N, p.K, p.K2, m0, n0, smem, acc, p.perm_b != 0); break;
            case 11: mainloop_pp<BM, 256, 11>(p.A, p.lda, Bw, p.ldb, A2, p.lda2, p.B2, p.ldb2, m_end, p.N, p.K, p.K2, m0, n0, smem, acc, p.perm_b != 0); break;
            default: break;
        }
        if (acc[0][0][0] != 1234.5f) return;
    }
#endif
    // multiply-by-derivative epilogue on the 256-row tile: the aux tile is staged by the K loop's last two steps (gemm_core.h AUXPRE)
    constexpr bool AUX_IN_LOOP = EPI == EPI_MULAUX && BM == 256;
    const bool aux_staged = AUX_IN_LOOP && p.aux_pre != 0 && p.K2 == 0 && (p.K >> 6) >= 2 && REID_DBG(p) != 4;
    if (REID_DBG(p) != 4) {
        mainloop_pp<BM, 256, 0, AUX_IN_LOOP>(p.A, p.lda, Bw, p.ldb, A2, p.lda2, p.B2, p.ldb2, m_end, p.N, p.K, p.K2, m0, n0, smem, acc, p.perm_b != 0,
                                             aux_staged ? (const bf16_t*)p.aux : nullptr, p.ldaux);
    }
    GEMM_TRACE(1);
    if (REID_DBG(p) == 1) return;
    if constexpr (EPI == EPI_MULAUX) {
        if (aux_staged) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            store_tile_fast<TM, 4, EPI>(p, acc, m0 + wm * RW, n0 + wn * 64, lane, m_end, smem, wm * RW, wn * 8);
            GEMM_TRACE(2);
#ifdef REID_GEMM_TRACE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GEMM_TRACE(3);
#endif
            return;
        }
        // The saved-derivative tile (256 rows x 512 B) comes in through LDS, which the K loop has just left: 128 LDS-DMA instructions
        // of two 512-byte row segments each, instead of 128 register loads of sixteen 64-byte row segments -- the register-direct form
        // kept a workgroup 8.5 us in this epilogue (per-workgroup trace, tools/exp_gemm_trace.py), most of it waiting for those reads.
        __syncthreads();                                     // every wave has left the K loop's last fragment reads
#pragma unroll
        for (int q = 0; q < BM / 16; ++q) {
            const int rp = q * 8 + wave;                     // row pair
            const int row = 2 * rp + (lane >> 5);
            const int c = lane & 31;                         // destination chunk position; it holds source chunk c ^ (row & 15)
            const int gm = m0 + row < m_end ? m0 + row : m_end - 1;
            const char* src = (const char*)p.aux + ((size_t)gm * p.ldaux + n0) * 2 + ((c ^ (row & 15)) << 4);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + rp * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        store_tile_fast<TM, 4, EPI>(p, acc, m0 + wm * RW, n0 + wn * 64, lane, m_end, smem, wm * RW, wn * 8);
        GEMM_TRACE(2);
#ifdef REID_GEMM_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GEMM_TRACE(3);
#endif
        return;
    }
    if constexpr (EPI == EPI_GENERIC) store_tile<TM, 4>(p, acc, m0 + wm * RW, n0 + wn * 64, lane, m_end);
    else store_tile_fast<TM, 4, EPI>(p, acc, m0 + wm * RW, n0 + wn * 64, lane, m_end);
    GEMM_TRACE(2);
#ifdef REID_GEMM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GEMM_TRACE(3);
#endif
}
#ifdef REID_GEMM_TRACE
static unsigned long long* g_gemm_trace = nullptr;
extern "C" void reid_debug_gemm_trace(void* buf) { g_gemm_trace = (unsigned long long*)buf; }
#endif

// PERSISTENT ping-pong kernel (gemm_core.h stream_pp): gridDim.x = number of CUs (a multiple of 8), every workgroup walks tiles
// blockIdx.x, + gridDim.x, ... in the same XCD-aware / L2-aware order the one-tile-per-workgroup kernel is dispatched in, and
// prefetches the next tile's first two K-tiles under the last two of the current one.  Lean register-only epilogues only.
template <int EPI, int BM = 256>
__global__ __launch_bounds__(512, 2) void mer_gemm_pps_kernel(const GemmParams p) {
    REID_T16_ENTER();
    if constexpr (EPI == EPI_PLAIN16H) REID_F16_SATURATE();
    using PC = PPCfg<BM, 256>;
    constexpr int TM = PC::TM, RW = PC::RW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int n_tiles = p.tiles_m * p.tiles_n;
    auto tile_of = [&](int v, int& m0, int& m_end, int& n0, const bf16_t*& Bw) {
        const int lin = xcd_linear_block(v, n_tiles);
        int tm, tn;
        tile_coords(lin, p.tiles_m, p.tiles_n, tm, tn, p.group_m);
        tile_rows(p, tm, BM, m0, m_end, Bw);
        n0 = tn * 256;
    };
    auto init = [&](f32x4 (&acc)[4][TM], int n0) {
        if constexpr (EPI == EPI_RES32) init_acc_m<TM, 4, 0>(p, acc, n0 + wn * 64, lane);
        else init_acc_m<TM, 4, 1>(p, acc, n0 + wn * 64, lane);
    };
    auto epi = [&](f32x4 (&acc)[4][TM], int m0, int m_end, int n0) {
        store_tile_fast<TM, 4, EPI>(p, acc, m0 + wm * RW, n0 + wn * 64, lane, m_end);
    };
#ifdef REID_GEMM_TRACE
    // per tile: 0 tile start, 1 accumulators initialised (bias landed), 2 K loop minus its last two K-tiles done, 3 next tile's offsets set,
    // 4 K loop done, 5 epilogue issued (thread 0 = wave row 0); 6 = the same epilogue point seen by wave row 1 (thread 256), 7 = CU id
    auto trace = [&](int v, int slot) {
        if (!p.trace) return;
        if (threadIdx.x == 0) {
            p.trace[(size_t)v * 8 + slot] = __builtin_amdgcn_s_memrealtime();
#ifndef REID_GEMM_TRACE_CLOCK
            if (slot == 0) p.trace[(size_t)v * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
#endif
        }
#ifdef REID_GEMM_TRACE_CLOCK
        // shader clock over the K loop: slots 6 / 7 = s_memtime at trace points 1 / 2 (replaces the row-1 stamp and the CU id)
        if (threadIdx.x == 0 && slot == 1) p.trace[(size_t)v * 8 + 6] = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0 && slot == 2) p.trace[(size_t)v * 8 + 7] = __builtin_amdgcn_s_memtime();
#else
        if (threadIdx.x == 256 && slot == 5) p.trace[(size_t)v * 8 + 6] = __builtin_amdgcn_s_memrealtime();
#endif
    };
#else
    auto trace = [&](int, int) {};
#endif
    stream_pp<BM, 256>(p.A, p.lda, p.ldb, p.N, p.K, n_tiles, smem, p.perm_b != 0, tile_of, init, epi, trace);
}

template <int EPI, int BM>
int launch_pps_e(GemmParams& p, hipStream_t s) {
    using C = PPCfg<BM, 256>;
    constexpr int LDS = C::LDS_BYTES;
    REID_MAX_LDS((mer_gemm_pps_kernel<EPI, BM>), LDS);
    const int tiles = p.tiles_m * p.tiles_n;
    int grid = reid_num_cus() & ~7;
    if (grid > tiles) grid = tiles;                              // (then every workgroup has exactly one tile: v + grid >= tiles)
#ifdef REID_GEMM_TRACE
    p.trace = g_gemm_trace;
#endif
    hipLaunchKernelGGL((mer_gemm_pps_kernel<EPI, BM>), dim3(grid), dim3(C::NT), LDS, s, p);
    REID_CHECK_LAUNCH("reid_mer_gemm");
    return REID_OK;
}


template <int EPI, int BM>
int launch_pp_e(GemmParams& p, hipStream_t s) {
    using C = PPCfg<BM, 256>;
    REID_MAX_LDS((mer_gemm_pp_kernel<EPI, BM>), C::LDS_BYTES);
    hipLaunchKernelGGL((mer_gemm_pp_kernel<EPI, BM>), dim3(p.tiles_m * p.tiles_n), dim3(C::NT), C::LDS_BYTES, s, p);
    REID_CHECK_LAUNCH("reid_mer_gemm");
    return REID_OK;
}
// Tile height of the ping-pong kernel: 224 rows when that needs no more rounds of 256 workgroups than 256 rows do (each round is then 7/8
// as long: fc2 267 -> 246 us, fc1 backward 223 -> 206 us, q|k|v backward 199 -> 190 us at 50 432 rows, N = 768); with an extra round the
// per-tile costs (first operands, epilogue) outweigh the shorter tiles (N = 3072: 318 -> 333 us).  REID_GEMM_TILE=12 / 14 force 256 / 224.
static int pick_pp_bm(const GemmParams& p, int tile_knob) {
    if (tile_knob == 12) return 256;
    if (tile_knob == 14) return 224;
    const long cus = reid_num_cus();
    const long tn = p.N / 256;
    GemmParams q = p;
    const long t256 = (long)row_tiles(q, 256, false) * tn, t224 = (long)row_tiles(q, 224, false) * tn;
    return (t224 + cus - 1) / cus <= (t256 + cus - 1) / cus ? 224 : 256;
}
int launch_pp(GemmParams& p, hipStream_t s, int tile_knob) {
    p.epi = reid_knob(KNOB_GEMM_EPI) == 0 ? EPI_GENERIC : pick_epilogue(p, 256);
    const int bm = p.epi == EPI_GENERIC ? 256 : pick_pp_bm(p, tile_knob);     // (the generic epilogue is only built for the 256-row tile)
    p.tiles_m = row_tiles(p, bm, true);
    p.tiles_n = (p.N + 255) / 256;
    p.stagger = reid_knob(KNOB_GEMM_STAGGER) > 0 ? reid_knob(KNOB_GEMM_STAGGER) : 0;
    p.aux_pre = reid_knob(KNOB_GELU_IMPL) != 1;
#ifdef REID_GEMM_TRACE
    p.trace = g_gemm_trace;
#endif
    // persistent stream form (default where it applies; REID_GEMM_PERSIST=0 keeps one tile per workgroup)
    // REID_GEMM_PERSIST bits: 1 = persistent form on; 2 = GELU epilogues too; 4 = multiply-by-derivative epilogue too (register-direct
    // aux operand); 8 = the short-K narrow-N out-projection shapes on this tile as well.  Default 9.  In-step A/B (r03, bench.py, one box):
    // 1: 33.15-33.25 ms/step, 9: 33.02, 13: 32.98-33.00, 5: 33.40 -- inside the step a persistent GEMM keeps its CUs for its whole duration,
    // so the HBM-bound side-stream kernels cannot take compute units from it (q|k|v backward 263 -> 194 us in-step)
    const int pk = reid_knob(KNOB_GEMM_PERSIST) < 0 ? 9 : reid_knob(KNOB_GEMM_PERSIST);
    const bool persist = pk != 0 && p.K2 == 0 && p.K >= 192 && (reid_num_cus() & ~7) >= 8 &&
                         (p.epi == EPI_PLAIN16 || p.epi == EPI_PLAIN16H || p.epi == EPI_RES32 || ((pk & 2) && (p.epi == EPI_GELU2 || p.epi == EPI_GELU2D)) ||
                          ((pk & 4) && p.epi == EPI_MULAUX));
    // (r03, tools/bench_gemm_shapes.py, profiles/r03_gemm_pps2.log: q|k|v 195 -> 183 us, q|k|v backward 157 -> 151, fc1 backward 207 -> 202,
    //  residual shapes unchanged; the GELU shapes are 2 % SLOWER persistent (327 -> 334 us: their 6 + 3.5 us VALU-bound epilogue dominates the
    //  tile boundary and a static tile sequence cannot rebalance it), so they stay one tile per workgroup unless REID_GEMM_PERSIST=2)
    if (persist) {
#define REID_PPS_CASE(E) case E: return bm == 224 ? launch_pps_e<E, 224>(p, s) : launch_pps_e<E, 256>(p, s);
        switch (p.epi) {
            REID_PPS_CASE(EPI_PLAIN16)
#ifndef REID_FLAVOR_F16
            REID_PPS_CASE(EPI_PLAIN16H)
#endif
            REID_PPS_CASE(EPI_RES32)
            REID_PPS_CASE(EPI_GELU2)
            REID_PPS_CASE(EPI_GELU2D)
            REID_PPS_CASE(EPI_MULAUX)
            default: break;
        }
#undef REID_PPS_CASE
    }
#define REID_PP_CASE(E) case E: return bm == 224 ? launch_pp_e<E, 224>(p, s) : launch_pp_e<E, 256>(p, s);
    switch (p.epi) {
        REID_PP_CASE(EPI_PLAIN16)
#ifndef REID_FLAVOR_F16
        REID_PP_CASE(EPI_PLAIN16H)
#endif
        REID_PP_CASE(EPI_RES32)
        REID_PP_CASE(EPI_GELU2)
        REID_PP_CASE(EPI_DGELU)
        REID_PP_CASE(EPI_GELU2D)
        REID_PP_CASE(EPI_MULAUX)
        default: return launch_pp_e<EPI_GENERIC, 256>(p, s);
    }
#undef REID_PP_CASE
}

template <int BM, int BN, int WM, int WN, int EPI>
int launch_e(GemmParams& p, hipStream_t s) {
    using C = Cfg<BM, BN, WM, WN>;
    REID_MAX_LDS((mer_gemm_kernel<BM, BN, WM, WN, EPI>), C::LDS_BYTES);
    hipLaunchKernelGGL((mer_gemm_kernel<BM, BN, WM, WN, EPI>), dim3(p.tiles_m * p.tiles_n), dim3(C::NT), C::LDS_BYTES, s, p);
    REID_CHECK_LAUNCH("reid_mer_gemm");
    return REID_OK;
}
// the default 128 x 128 tile with the lean epilogue that covers the launch (GENERIC otherwise)
int launch_main(GemmParams& p, hipStream_t s) {
    p.tiles_m = row_tiles(p, 128, true);
    p.tiles_n = (p.N + 127) / 128;
    p.epi = reid_knob(KNOB_GEMM_EPI) == 0 ? EPI_GENERIC : pick_epilogue(p, 128);
    switch (p.epi) {
        case EPI_PLAIN16: return launch_e<128, 128, 2, 2, EPI_PLAIN16>(p, s);
#ifndef REID_FLAVOR_F16
        case EPI_PLAIN16H: return launch_e<128, 128, 2, 2, EPI_PLAIN16H>(p, s);
#endif
        case EPI_RES32: return launch_e<128, 128, 2, 2, EPI_RES32>(p, s);
        case EPI_GELU2: return launch_e<128, 128, 2, 2, EPI_GELU2>(p, s);
        case EPI_DGELU: return launch_e<128, 128, 2, 2, EPI_DGELU>(p, s);
        case EPI_GELU2D: return launch_e<128, 128, 2, 2, EPI_GELU2D>(p, s);
        case EPI_MULAUX: return launch_e<128, 128, 2, 2, EPI_MULAUX>(p, s);
        default: return launch_e<128, 128, 2, 2, EPI_GENERIC>(p, s);
    }
}

template <int BM, int BN, int WM, int WN>
int launch(GemmParams& p, hipStream_t s) {
    using C = Cfg<BM, BN, WM, WN>;
    p.tiles_m = row_tiles(p, BM, true);
    p.tiles_n = (p.N + BN - 1) / BN;
    REID_MAX_LDS((mer_gemm_kernel<BM, BN, WM, WN>), C::LDS_BYTES);
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
    REID_CHECK_ARG(a->act >= 0 && a->act <= REID_ACT_GELU_ERF_DSAVE, "reid_mer_gemm: act=%d", a->act);
    REID_CHECK_ARG(a->act < REID_ACT_DGELU_ERF || a->act > REID_ACT_MUL_AUX || (a->aux && a->ldaux >= a->N), "reid_mer_gemm: D* / MUL_AUX activation needs aux");
    REID_CHECK_ARG(a->act != REID_ACT_GELU_ERF_DSAVE || a->C2, "reid_mer_gemm: GELU_ERF_DSAVE needs C2");
    REID_CHECK_ARG(!a->R || a->ldr >= a->N, "reid_mer_gemm: ldr");
    REID_CHECK_ARG(!a->C2 || a->ldc2 >= a->N, "reid_mer_gemm: ldc2");
    REID_CHECK_ARG(a->mask_r == 0 || (a->img_mod && a->rows_per_img > 0 && a->mask_period > 0),
                   "reid_mer_gemm: modality mask needs img_mod, rows_per_img, mask_period");
    REID_CHECK_ARG(a->c_group == 0 || a->c_group_stride >= a->c_group, "reid_mer_gemm: c_group_stride");
    REID_CHECK_ARG(!a->row_scale || a->rows_per_img > 0, "reid_mer_gemm: row_scale needs rows_per_img");
    REID_CHECK_ARG(a->n_row_groups >= 0 && a->n_row_groups <= REID_GEMM_MAX_GROUPS, "reid_mer_gemm: n_row_groups=%d (max %d)", a->n_row_groups, REID_GEMM_MAX_GROUPS);
    if (a->n_row_groups > 0) {
        int prev = 0;
        for (int g = 0; g < a->n_row_groups; ++g) {
            REID_CHECK_ARG(a->row_group_end[g] > prev, "reid_mer_gemm: row_group_end must be strictly increasing (empty groups are left out by the caller)");
            REID_CHECK_ARG(a->row_group_b[g] >= 0, "reid_mer_gemm: row_group_b[%d] < 0", g);
            REID_CHECK_ARG((int64_t)(a->row_group_b[g] + 1) * a->b_group_stride * 2 < (1ll << 40), "reid_mer_gemm: weight stack too large");
            prev = a->row_group_end[g];
        }
        REID_CHECK_ARG(prev == a->M, "reid_mer_gemm: the row groups must cover exactly M=%d rows (last end %d)", a->M, prev);
        REID_CHECK_ARG(a->b_group_stride >= (int64_t)a->N * a->ldb, "reid_mer_gemm: b_group_stride smaller than one [N, ldb] matrix");
        REID_CHECK_ARG(a->c_group == 0 && a->r_period == 0, "reid_mer_gemm: row groups do not combine with c_group / r_period");
    }
    GemmParams p;
    p.stagger = 0;
    p.trace = nullptr;
    p.A = (const bf16_t*)a->A; p.B = (const bf16_t*)a->B; p.A2 = (const bf16_t*)a->A2; p.B2 = (const bf16_t*)a->B2;
    p.bias = a->bias; p.R = a->R; p.aux = (const bf16_t*)a->aux; p.C = a->C; p.C2 = a->C2; p.img_mod = a->img_mod; p.row_scale = a->row_scale;
    p.M = a->M; p.N = a->N; p.K = a->K; p.K2 = a->A2 ? a->K2 : 0;
    p.n_groups = a->n_row_groups;
    p.b_group_stride = a->b_group_stride;
    for (int g = 0; g < REID_GEMM_MAX_GROUPS; ++g) {
        p.grp_row0[g] = p.grp_row_end[g] = p.grp_tile0[g] = p.grp_b[g] = 0;
        if (g < a->n_row_groups) {
            p.grp_row0[g] = g == 0 ? 0 : a->row_group_end[g - 1];
            p.grp_row_end[g] = a->row_group_end[g];
            p.grp_b[g] = a->row_group_b[g];
        }
    }
    p.lda = a->lda; p.ldb = a->ldb; p.lda2 = a->lda2; p.ldb2 = a->ldb2; p.ldr = a->ldr; p.ldaux = a->ldaux;
    p.ldc = a->ldc; p.ldc2 = a->ldc2; p.k2_group_n = a->k2_group_n;
    p.act = a->act; p.c_dtype = a->c_dtype; p.c2_dtype = a->c2_dtype; p.r_dtype = a->r_dtype;
    p.r_period = a->r_period; p.mask_r = a->mask_r; p.mask_period = a->mask_period; p.rows_per_img = a->rows_per_img;
    p.c_group = a->c_group; p.c_group_stride = a->c_group_stride; p.c_row_off = a->c_row_off;
    p.alpha = a->alpha == 0.f ? 1.f : a->alpha;
    p.dbg = 0;
    if (p.c_dtype == REID_F16 && REID_FLAVOR_ID == 1) p.c_dtype = REID_BF16;          // the f16 flavor's own format (saturating there too)
    REID_CHECK_ARG(!(a->C2 && a->c2_dtype == REID_F16 && REID_FLAVOR_ID == 0) && !(a->R && a->r_dtype == REID_F16 && REID_FLAVOR_ID == 0),
                   "reid_mer_gemm: REID_F16 is supported for C only");
    // L2 group height of the tile order: 16 row tiles when the weight panel set is wide and K short (q|k|v, fc1: the whole
    // [N, K] weight no longer fits one XCD's L2 next to 8 activation tiles and was re-streamed per group; r01 sweep 4..64)
    // r04, 256-row tiles, in-step A/B on one box (tools/exp_r04_ab.sh): 8 for every shape 30.73-30.79 ms per step, 16 for the wide short-K
    // shapes (the r01 choice, made with 128-row tiles) 30.87-30.89, 4: 30.72-30.84, 16 everywhere 31.3, 32: 32.2
    p.group_m = 8;
    if (reid_knob(KNOB_GEMM_GROUPM) > 0) p.group_m = reid_knob(KNOB_GEMM_GROUPM);
    p.perm_b = epilogue_wide16(p) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    if (p.c_dtype == REID_F16) {
        // only the plain lean epilogue stores IEEE half in the bf16 flavor (what the residual-branch GEMMs of the vision blocks use)
        REID_CHECK_ARG(a->N > 96 && a->N % 128 == 0 && reid_knob(KNOB_GEMM_EPI) != 0 && reid_knob(KNOB_GEMM_TILE) <= 0 &&
                       pick_epilogue(p, 128) == EPI_PLAIN16H,
                       "reid_mer_gemm: a REID_F16 output needs the plain 16-bit epilogue (N a multiple of 128, no activation / residual / second output / mask / row remap)");
    }
    // skinny outputs (LoRA down-projections, N <= 96) use a tall tile so no MFMA work is spent on padding
    // (tall 256-row tiles leave a 50k-row problem with < 256 workgroups: 64-row tiles fill the chip; r01: a three-buffer ring
    //  with counted vmcnt -- twice the bytes in flight per workgroup -- changed the step time by < 0.1 %: not kept)
    // (r02: a streaming form of these projections -- activation fragments loaded straight into VGPRs, 24 KiB per wave in flight,
    //  the small weight fragment-linear in LDS -- was built and measured against this tiled form on the three LoRA shapes,
    //  bit-identical outputs: 26 / 93 / 26 us vs 16 / 58 / 16 us stand-alone.  Fragment-shaped global
    //  loads (16 rows x 64 bytes per instruction) cost the vector-memory path more than the LDS round trip saves; stand-alone
    //  the tiled form already streams at 4.9-5.4 TB/s -- the 31 us seen in the train step is interference from the dA/dB
    //  reductions on the side stream, not this kernel.)
    if (a->N <= 32) {
        const int sk = reid_knob(KNOB_SKINNY_TILE);
        if (sk == 1) return launch<256, 32, 4, 1>(p, s);
        if (sk == 2) return launch<128, 32, 4, 1>(p, s);
        return a->M >= 65536 ? launch<256, 32, 4, 1>(p, s) : launch<64, 32, 4, 1>(p, s);
    }
    if (a->N <= 64) return a->M >= 65536 ? launch<256, 64, 4, 1>(p, s) : launch<64, 64, 4, 1>(p, s);
    if (a->N <= 96) return launch<128, 32, 4, 1>(p, s);
    // experiment knobs (cached table, common.h; a benchmark A/Bs tiles inside one process through reid_set_knob)
    const int tile = reid_knob(KNOB_GEMM_TILE) > 0 ? reid_knob(KNOB_GEMM_TILE) : 0;
    p.dbg = reid_knob(KNOB_GEMM_DBG) > 0 ? reid_knob(KNOB_GEMM_DBG) : 0;
    if (tile == 2) return launch<256, 256, 2, 4>(p, s);
    if (tile == 3) return launch_main(p, s);
    if (tile == 8) return launch<128, 256, 2, 4>(p, s);
    {
        // 256 x 256 ping-pong tile (gemm_core.h mainloop_pp) where it wins (r02, same harness, sum of the seven ViT shapes: 1.74 ms vs
        // 1.89 ms for the 128 x 128 tile, both with the lean epilogues; with the GENERIC epilogue it loses, 2.16 ms): every shape
        // except the short-K, narrow-N out-projection (591 tiles of 256 x 256 = 2.3 waves of the chip and only 12 K-tiles each),
        // and only when a lean epilogue covers the launch and there is at least one tile per CU.
        // (r02, measured and dropped, profiles/r02_gemm_experiments2_persistent_stream.patch: a PERSISTENT form of this kernel whose
        // half-tile prefetch runs on across output tiles, so a tile's first operands are in LDS when its loop starts -- 1804 us vs
        // 1667 us per layer.  gfx9 counts loads and stores in one in-order vmcnt: the first counted wait of the next tile has to
        // drain the previous tile's C stores, whereas a workgroup that simply ENDS leaves its stores draining under the next
        // workgroup's K loop.  One tile per workgroup already has the store / K-loop overlap the persistent form was after.)
        // (r02, second attempt, profiles/r02_gemm_experiments3_persistent_queue.patch: persistent workgroups fed from per-XCD atomic
        // tile counters with one tile of lookahead, the two-phase loop streaming across tile boundaries, register-only epilogues.
        // Correct (all GEMM tests pass) and 18 % slower over the seven shapes: per-tile traces show K loops of 30.5 us instead of 20.3 us
        // and 2.8 us between tiles -- hipcc waits vmcnt(0) for the bias loads of the accumulator init and for the queue index, which
        // drains the prefetch at every tile start, and the loop carries ~40 spilled registers across tile boundaries.)
        const bool pp_ok = (p.k2_group_n == 0 || p.k2_group_n % 256 == 0) && p.N % 256 == 0;
        const long tiles256 = (long)row_tiles(p, 256, false) * (p.N / 256);
        const bool small_too = reid_knob(KNOB_GEMM_PERSIST) < 0 || (reid_knob(KNOB_GEMM_PERSIST) & 8);     // out-projection shapes as well (see launch_pp)
        const bool pp_shape = (p.K + p.K2 >= 1536 || p.N >= 1536 || small_too) && tiles256 >= reid_num_cus();
        if (pp_ok && (tile == 12 || tile == 14 || (tile == 0 && pp_shape && reid_knob(KNOB_GEMM_EPI) != 0 && pick_epilogue(p, 256) != EPI_GENERIC)))
            return launch_pp(p, s, tile);
    }
    // Default from same-process A/B runs of the seven ViT GEMM variants (tools/bench_gemm_variants.py, r01): 128x128x64
    // tiles, 4 waves, 64 KiB of LDS -> TWO workgroups per CU.  With the register-direct epilogue its K loop runs as fast
    // as the 256x256 tile's (~1.0-1.1 PF on these shapes) and, unlike one big workgroup per CU, one workgroup's stores
    // overlap the other's MFMAs (sum over the seven shapes: 1.90 ms vs 2.04-2.17 ms for 256x256 / 128x256 tiles).
    // (k2_group_n is a multiple of 128, so a column tile never straddles two LoRA groups of the fused q|k|v projection.)
    // r02 anatomy of this kernel on the seven ViT shapes (profiles/r02_gemm_variants*.log, sum per layer, 1 MI355X):
    //   whole kernel 1.93 ms = K loop alone 1.40-1.43 ms (REID_GEMM_DBG=1) + epilogue alone 0.59-0.60 ms (REID_GEMM_DBG=4): they do
    //   NOT overlap although two workgroups share a CU.  K loop with the MFMAs removed (LDS-DMA + fragment reads only): 1.22 ms =
    //   22 GB of L2->LDS operand traffic at 71 GB/s per CU, the per-CU LDS-DMA rate of MI355X_MICROARCH.md ("Indexed rows:
    //   gather into LDS", 66-73 GB/s): the 128x128 K loop is bound by the CU's vector-memory path, not by the matrix pipe
    //   (MFMAs + fragment reads without any LDS-DMA: 1.10 ms = 1.33 PF); the epilogue streams C / residual / aux at 4.2 TB/s and
    //   needs the same path.  Tried in r02 and dropped (all measured in that harness, code in git history of this round):
    //   * persistent two-workgroups-per-CU grid with the partner of each CU (blocks b and b+256, found with tools/hwinfo.hip via
    //     HW_REG_LDS_ALLOC) started 0.5x..2.5x of a K-loop time late, next tile's first K-step prefetched under the epilogue:
    //     1.99 ms unstaggered, 2.03-2.22 ms staggered (phasing the pair does not create overlap: they contend for that path);
    //   * three-slot LDS ring with counted vmcnt + raw s_barrier, one 8-wave workgroup per CU (256x128, 128x256) or 128x128:
    //     K loops 1.49-1.54 ms / 1.99 ms (eight barrier-coupled waves serialise their read and MFMA phases);
    //   * epilogue through a wave-private LDS transpose (4 rows x 256 B per access instead of 16 rows x 64 B): epilogue alone
    //     0.85 ms vs 0.60 ms -- segment length is not what holds the epilogue at 4.2 TB/s.
    //   What is left is halving the operand traffic per flop (256x256 tiles) TOGETHER with a wave-group ping-pong K loop
    //   and an epilogue that overlaps the next tile; the plain 256x256 / 128x256 tiles here lose more in the un-hidden
    //   epilogue (2.27 / 2.15 ms) than their K loop gains (1.35 ms).
    // Tried and dropped in r01 (same harness): LDS-ring variants with 32-wide K steps and counted vmcnt (128x256 tile with 4
    // waves of 64x128, 128x128 with 2/3/4 stages at 2/3/4 workgroups per CU): 5-25 % slower than this on every shape;
    // a DPP lane exchange that makes the epilogue store 128 contiguous bytes per row: slower (the DPP hazards cost more
    // than the wider segments give).
    return launch_main(p, s);
}
