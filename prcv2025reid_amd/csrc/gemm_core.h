// Shared MFMA tile main loop of the gfx950 GEMM-shaped kernels (mer_gemm, cosine_topk).
// See gemm.hip for the design notes.
#pragma once
#include "common.h"

namespace gemmcore {

// physical 16-byte chunk of logical chunk c in row `row` of a [rows][64] bf16 tile (128-byte rows).
// Two rows share one 256-byte bank row; rows r and r+2 would otherwise collide on every ds_read_b128.
__device__ __forceinline__ int swz(int row, int c) { return c ^ ((row >> 1) & 7); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BM, int BN, int WM, int WN>
struct Cfg {
    static constexpr int NW = WM * WN;
    static constexpr int NT = NW * 64;
    static constexpr int TM = BM / WM / 16;          // 16-row activation sub-tiles per wave
    static constexpr int TN = BN / WN / 16;          // 16-row weight sub-tiles per wave
    static constexpr int A_BYTES = BM * 128;
    static constexpr int B_BYTES = BN * 128;
    static constexpr int BUF_BYTES = A_BYTES + B_BYTES;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;
    static constexpr int A_INSTR = BM / 8 / NW;      // LDS-DMA wave-instructions per wave per tile
    static constexpr int B_INSTR = BN / 8 / NW;
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split over waves");
};

// Row permutation inside aligned groups of 32 operand rows: LDS row 16*jb + 4*fq + r holds operand row 8*fq + 4*jb + r.
// With it a lane's accumulators of two neighbouring 16-row MFMA sub-tiles are EIGHT consecutive output columns and the
// four lanes that share an output row cover 32 consecutive columns, so a 16-bit epilogue stores 16 bytes per lane
// straight from the accumulators (64 contiguous bytes per output row per instruction) without an LDS transpose.
__device__ __forceinline__ int perm32(int x) { return (x & ~31) | (((x >> 2) & 3) << 3) | (((x >> 4) & 1) << 2) | (x & 3); }

// Stage one K-step of one operand: rows [row0, row0+ROWS) x 64 k (HALF: only logical chunks 0..3 are
// meaningful; the other lanes re-load a valid chunk that is never read).
template <int ROWS, int NW, bool HALF>
__device__ __forceinline__ void stage(const bf16_t* __restrict__ src, int ld, int row0, int row_max, int k0,
                                      char* lds, int wave, int lane, bool perm = false) {
    constexpr int INSTR = ROWS / 8 / NW;
#pragma unroll
    for (int i = 0; i < INSTR; ++i) {
        const int rblk = (i * NW + wave) * 8;
        const int r = rblk + (lane >> 3);
        int c = swz(r, lane & 7);
        if (HALF) c &= 3;
        int grow = row0 + (perm ? perm32(r) : r);
        grow = grow < row_max ? grow : row_max;
        const bf16_t* g = src + (size_t)grow * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(lds + rblk * 128), 16, 0, 0);
    }
}


// Per-lane byte offsets of one operand tile's LDS-DMA loads (row clamp + source swizzle), computed ONCE per tile:
// inside the K loop a load is then `uniform base (advances by 128 B per step, SALU) + 32-bit lane offset`, no vector
// address arithmetic (the first version spent ~4 VALU instructions per MFMA, mostly on 64-bit load addresses).
template <int ROWS, int NW, bool HALF>
__device__ __forceinline__ void stage_offsets(int ld, int row0, int row_max, int wave, int lane, uint32_t (&off)[ROWS / 8 / NW],
                                              bool perm = false) {
    constexpr int INSTR = ROWS / 8 / NW;
#pragma unroll
    for (int i = 0; i < INSTR; ++i) {
        const int r = (i * NW + wave) * 8 + (lane >> 3);
        int c = swz(r, lane & 7);
        if (HALF) c &= 3;
        int grow = row0 + (perm ? perm32(r) : r);
        grow = grow < row_max ? grow : row_max;
        off[i] = (uint32_t)grow * (uint32_t)ld * 2u + (uint32_t)c * 16u;
    }
}
template <int ROWS, int NW>
__device__ __forceinline__ void stage_from(const char* __restrict__ base, const uint32_t (&off)[ROWS / 8 / NW], char* lds, int wave) {
    constexpr int INSTR = ROWS / 8 / NW;
#pragma unroll
    for (int i = 0; i < INSTR; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(base + off[i]), (lptr_t)(lds + (i * NW + wave) * 8 * 128), 16, 0, 0);
}

// XCD-aware linear tile id: blocks b and b+8 share an XCD (one L2); every XCD gets a contiguous run
// of tiles (bijective for any grid size, cdna_hip_programming.md section 5 "XCD swizzle").
__device__ __forceinline__ int xcd_linear_block(int bid, int nwg) {
    const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    return (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
}

// L2-aware tile order inside one XCD's contiguous run of tiles: groups of GROUP_M row-tiles are walked across ALL
// column-tiles before moving on, so the ~64 tiles an XCD has in flight reuse the same GROUP_M activation blocks and a
// handful of weight panels (both stay in the 4 MiB L2); with the plain "m fastest" order every weight panel pass
// re-streamed the whole activation matrix from the Infinity Cache (18 x 81 MB per QKV GEMM).
__device__ __forceinline__ void tile_coords(int lin, int tiles_m, int tiles_n, int& tm, int& tn, int GROUP_M = 8) {
    const int per_group = GROUP_M * tiles_n;
    const int gid = lin / per_group;
    const int first_m = gid * GROUP_M;
    const int gsize = min(tiles_m - first_m, GROUP_M);
    const int in_g = lin - gid * per_group;
    tm = first_m + in_g % gsize;
    tn = in_g / gsize;
}

// acc[j][i] (+)= W[n0 + wn-slice + 16j .. , :] . X[m0 + wm-slice + 16i .. , :]^T over K (64-wide steps)
// plus the optional low-rank pair over K2 (32-wide half steps).  MFMA rows = B-operand rows (n),
// MFMA columns = A-operand rows (m): lane holds C[m = lane&15][n = 4*(lane>>4) + reg].
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void mainloop(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
                                         const bf16_t* __restrict__ A2, int lda2, const bf16_t* __restrict__ B2, int ldb2,
                                         int M, int N, int K, int K2, int m0, int n0, char* smem,
                                         f32x4 (&acc)[Cfg<BM, BN, WM, WN>::TN][Cfg<BM, BN, WM, WN>::TM], bool perm_b = false) {
    using C = Cfg<BM, BN, WM, WN>;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nk = K >> 6;
    const int nk2 = A2 ? (K2 >> 5) : 0;
    const int steps = nk + nk2;

    uint32_t offA[C::A_INSTR], offB[C::B_INSTR];
    stage_offsets<BM, C::NW, false>(lda, m0, M - 1, wave, lane, offA);
    stage_offsets<BN, C::NW, false>(ldb, n0, N - 1, wave, lane, offB, perm_b);
    auto issue = [&](int t, int buf) {
        char* la = smem + buf * C::BUF_BYTES;
        char* lb = la + C::A_BYTES;
        if (t < nk) {
            stage_from<BM, C::NW>((const char*)A + (size_t)t * 128, offA, la, wave);
            stage_from<BN, C::NW>((const char*)B + (size_t)t * 128, offB, lb, wave);
        } else {
            const int k2 = (t - nk) << 5;
            stage<BM, C::NW, true>(A2, lda2, m0, M - 1, k2, la, wave, lane);
            stage<BN, C::NW, true>(B2, ldb2, n0, N - 1, k2, lb, wave, lane, perm_b);
        }
    };
    const int frow = lane & 15, fq = lane >> 4;
    auto compute = [&](int buf, int nks) {
        const char* la = smem + buf * C::BUF_BYTES;
        const char* lb = la + C::A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks < nks) {
                bf16x8 af[C::TM], wf[C::TN];
#pragma unroll
                for (int i = 0; i < C::TM; ++i) {
                    const int row = wm * (BM / WM) + i * 16 + frow;
                    af[i] = *(const bf16x8*)(la + row * 128 + (swz(row, ks * 4 + fq) << 4));
                }
#pragma unroll
                for (int j = 0; j < C::TN; ++j) {
                    const int row = wn * (BN / WN) + j * 16 + frow;
                    wf[j] = *(const bf16x8*)(lb + row * 128 + (swz(row, ks * 4 + fq) << 4));
                }
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
#pragma unroll
                    for (int i = 0; i < C::TM; ++i)
                        acc[j][i] = mfma16(wf[j], af[i], acc[j][i]);
            }
        }
    };

    // (r01 experiment: software-pipelining the fragment reads through two register sets -- reads of half-step h+1 issued before
    //  the MFMAs of half-step h, buffer released as soon as its second half sits in registers -- changed nothing end to end
    //  (sum of the seven ViT shapes 1.90 ms either way) and cost the 256x256 tile a register spill; not kept.)
    issue(0, 0);
    __syncthreads();   // vmcnt(0) + barrier: tile 0 landed for every wave
    int cur = 0;
    for (int t = 0; t < steps - 1; ++t) {
        issue(t + 1, cur ^ 1);
        compute(cur, t < nk ? 2 : 1);
        __syncthreads();
        cur ^= 1;
    }
    compute(cur, (steps - 1) < nk ? 2 : 1);
}

}  // namespace gemmcore
