// Shared MFMA tile main loop of the gfx950 GEMM-shaped kernels (mer_gemm, cosine_topk).
// See gemm.hip for the design notes.
#pragma once
#include "common.h"
#include <type_traits>

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


// ---------------------------------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, eight waves as 2 (rows) x 4 (cols), wave-row PING-PONG K loop.
//
// Why: the 128 x 128 K loop moves 15.6 KB of operands per MFLOP through the CU's vector-memory path and is bound by it
// (71 GB/s per CU, gemm.hip); a 256 x 256 tile moves half of that.  One 8-wave workgroup per CU runs every wave in lockstep
// if all of them do "read fragments, then MFMA" between the same two barriers, so the matrix pipe idles during the reads.
// Here each K-tile is cut into two phases of 32 MFMAs (one 64-row half of the wave's 128 x 64 output each), every phase is
//     L: ds_read the fragments the phase needs, issue LDS-DMA prefetch, wait for the reads                            | barrier
//     M: 32 MFMAs (512 matrix-pipe cycles: longer than the partner's 16 fragment reads plus their latency)             | barrier
// and wave row 1 runs ONE BARRIER BEHIND wave row 0 (an extra s_barrier up front): while one row's waves are in M the other
// row's waves (their partners on the same SIMDs) are in L.  Waves w and w+4 share a SIMD.
//
// LDS (128 KiB): two K-tile buffers x {SA0, SA1, SB0, SB1} half-tiles of 128 rows x 128 B.  A wave reads activation rows only
// from SA[its row] and weight rows only from SB[its column pair].  K-tile T lives in buffer T & 1 and is refilled with K-tile
// T+2 as soon as its last reader is done -- every half-tile is in flight for a whole K-tile (~3000 cycles) before its first use:
//     L0(T): read A[rows 0-63 of the wave], B (both 32-row parts)
//     L1(T): read A[rows 64-127]; stage SB0, SB1 of K-tile T+2 (all 8 waves: both rows finished reading SB(T) in their L0,
//            the later of which ended one barrier ago); ONE counted wait, vmcnt(4) = everything but those four instructions,
//            i.e. all of K-tile T+1 (issued during K-tile T-1) has landed for this wave
//     M1(T): stage SA[own row] of K-tile T+2 -- each wave row stages its OWN activation half, whose only readers are its own
//            four waves, all past the barrier that closed L1 -- then the MFMAs
// Wave row 1 passes its wait in global interval 4T+3, row 0 first reads K-tile T+1 in interval 4T+4.  (r02 history: a four-phase
// form, 16 MFMAs per phase, all waves staging every half and two of the four halves issued only 2-3 phases before their use:
// 1667 us per layer; its 12-read phase and the short prefetch distance both showed as matrix-pipe idle time.)
template <int BM, int BN>
struct PPCfg {
    // BM = 224 (wave rows of 112 = 64 + 48 activation rows): same loop, chosen where it packs the chip's 256 CUs better -- M = 50 432,
    // N = 768 is 591 tiles of 256 x 256 = 2.31 rounds (three rounds, 77 % full) but 678 tiles of 224 x 256 = 2.65 rounds of 7/8 the size.
    static_assert((BM == 256 || BM == 224) && BN == 256, "ping-pong loop is written for 256 x 256 and 224 x 256 tiles");
    static constexpr int NW = 8, NT = 512, WM = 2, WN = 4, TN = 4;
    static constexpr int RW = BM / 2;                            // activation rows per wave row
    static constexpr int TM = RW / 16;                           // 16-row sub-tiles per wave: 8 or 7
    static constexpr int TM0 = 4, TM1 = TM - 4;                  // sub-tiles of the two MFMA phases
    static constexpr int A_INSTR = RW / 8;                       // LDS-DMA instructions per activation half-tile (4 waves share them)
    static constexpr int HALF_BYTES = 128 * 128;                 // 16 KiB slots
    static constexpr int BUF_BYTES = 4 * HALF_BYTES;             // SA0 | SA1 | SB0 | SB1
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;
};

__device__ __forceinline__ void pp_wait_vm4() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
__device__ __forceinline__ void pp_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void pp_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ABL (timing experiments only, results are wrong): bit 0 = no LDS-DMA in the loop, 1 = no fragment reads, 2 = no MFMAs, 3 = no barriers
// AUXPRE (256-row tile, aux != null): the epilogue's [256 rows x 512 B] tile of a second operand (the saved GELU derivative of the
// multiply-by-derivative epilogue) rides in on the LAST TWO K-tiles' staging slots -- where K-tiles nk and nk + 1 would be staged -- so it
// lands under the last MFMAs instead of after the loop: rows 128 b + .. of the tile go to buffer b as the image the epilogue reads
// (row r at r * 512, chunk c at position c ^ (r & 15)), the weight halves' slots taking rows 64..127 and each wave row's own activation
// slot rows 32 wm .. + 31, with the instruction counts of the steady state (the counted waits stay as they are).  Needs nk >= 2.
template <int BM, int BN, int ABL = 0, bool AUXPRE = false>
__device__ __forceinline__ void mainloop_pp(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
                                            const bf16_t* __restrict__ A2, int lda2, const bf16_t* __restrict__ B2, int ldb2,
                                            int M, int N, int K, int K2, int m0, int n0, char* smem, f32x4 (&acc)[4][PPCfg<BM, BN>::TM], bool perm_b,
                                            const bf16_t* __restrict__ aux = nullptr, int ldaux = 0) {
    using C = PPCfg<BM, BN>;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = K >> 6;
    const int nk2 = A2 ? (K2 >> 5) : 0;
    const int nt = nk + nk2;
    const int rl = lane >> 3, cl = lane & 7;

    // per-lane source byte offsets (row clamp + source swizzle) of this wave's LDS-DMA instructions:
    //   own activation half SA[wm]: 4 instructions, 8 rows each: half-tile rows (j * 4 + wn) * 8 ...
    //   weight halves SB0, SB1:     2 instructions each:          half-tile rows (j * 8 + wave) * 8 ...
    uint32_t offA[4], offB[2][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (j * 4 + wn) * 8 + rl;
        int ga = m0 + C::RW * wm + r;
        ga = ga < M - 1 ? ga : M - 1;
        offA[j] = (uint32_t)ga * (uint32_t)lda * 2u + (uint32_t)swz(r, cl) * 16u;
    }
    // instruction j of a wave covers half-tile rows (4 j + wn) * 8 ...: the 224-row tile has 14 such blocks, waves 2 and 3 skip j = 3
    const bool a_last = 12 + wn < C::A_INSTR;                                  // wave-uniform
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = (j * 8 + wave) * 8 + rl;
            const int rb = 128 * h + r;
            int gb = n0 + (perm_b ? perm32(rb) : rb);
            gb = gb < N - 1 ? gb : N - 1;
            offB[h][j] = (uint32_t)gb * (uint32_t)ldb * 2u + (uint32_t)swz(r, cl) * 16u;
        }
    // steady state (a full 64-wide K-tile): no conditions, the K offset is the only run-time term
    auto stage_a = [&](int t, int buf) {
        if constexpr (ABL & 1) return;
        char* dst = smem + buf * C::BUF_BYTES + wm * C::HALF_BYTES;
        const char* src = (const char*)A + (size_t)t * 128;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + offA[j]), (lptr_t)(dst + (j * 4 + wn) * 8 * 128), 16, 0, 0);
        if (C::A_INSTR == 16 || a_last)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + offA[3]), (lptr_t)(dst + (12 + wn) * 8 * 128), 16, 0, 0);
    };
    auto stage_b = [&](int t, int buf) {
        if constexpr (ABL & 1) return;
        const char* src = (const char*)B + (size_t)t * 128;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + offB[h][j]),
                                                 (lptr_t)(smem + buf * C::BUF_BYTES + (2 + h) * C::HALF_BYTES + (j * 8 + wave) * 8 * 128), 16, 0, 0);
    };
    // LoRA half-step (32 k, only 16-byte chunks 0..3 of a row are meaningful): same rows, addresses computed in place
    auto stage_a2 = [&](int t, int buf) {
        char* dst = smem + buf * C::BUF_BYTES + wm * C::HALF_BYTES;
        const int k2 = (t - nk) << 5;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j == 3 && C::A_INSTR != 16 && !a_last) break;
            const int r = (j * 4 + wn) * 8 + rl;
            int ga = m0 + C::RW * wm + r;
            ga = ga < M - 1 ? ga : M - 1;
            const bf16_t* g = A2 + (size_t)ga * lda2 + k2 + (swz(r, cl) & 3) * 8;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(dst + (j * 4 + wn) * 8 * 128), 16, 0, 0);
        }
    };
    auto stage_b2 = [&](int t, int buf) {
        const int k2 = (t - nk) << 5;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = (j * 8 + wave) * 8 + rl;
                const int rb = 128 * h + r;
                int gb = n0 + (perm_b ? perm32(rb) : rb);
                gb = gb < N - 1 ? gb : N - 1;
                const bf16_t* g = B2 + (size_t)gb * ldb2 + k2 + (swz(r, cl) & 3) * 8;
                __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(smem + buf * C::BUF_BYTES + (2 + h) * C::HALF_BYTES + (j * 8 + wave) * 8 * 128), 16, 0, 0);
            }
    };
    // one LDS-DMA instruction = 1 KiB = rows 2 rp, 2 rp + 1 of the aux image
    auto stage_aux_piece = [&](int rp, char* dst) {
        const int row = 2 * rp + (lane >> 5), c = lane & 31;
        const int gm = m0 + row < M ? m0 + row : M - 1;
        const char* src = (const char*)aux + ((size_t)gm * ldaux + n0) * 2 + ((c ^ (row & 15)) << 4);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    };
    auto stage_aux_b = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                stage_aux_piece(buf * 64 + (2 + h) * 16 + j * 8 + wave, smem + buf * C::BUF_BYTES + (2 + h) * C::HALF_BYTES + (j * 8 + wave) * 1024);
    };
    auto stage_aux_a = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            stage_aux_piece(buf * 64 + wm * 16 + j * 4 + wn, smem + buf * C::BUF_BYTES + wm * C::HALF_BYTES + (j * 4 + wn) * 1024);
    };
    const bool aux_on = AUXPRE && aux != nullptr && nt >= 2;
    auto stage_a_any = [&](int t, int buf) { if (t < nk) stage_a(t, buf); else stage_a2(t, buf); };
    auto stage_b_any = [&](int t, int buf) { if (t < nk) stage_b(t, buf); else stage_b2(t, buf); };

    // fragment reads: lane-constant part of the address (16-row sub-tiles start at multiples of 16 rows: the swizzle term only
    // depends on the row inside the sub-tile)
    const int frow = lane & 15, fq = lane >> 4;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = frow * 128 + ((swz(frow, ks * 4 + fq)) << 4);
    const int a_base = wm * C::HALF_BYTES;                                   // SA[wm]
    const int b_base = (2 + (wn >> 1)) * C::HALF_BYTES + (wn & 1) * 64 * 128; // SB[wn >> 1], this wave's 64 weight rows
    bf16x8 af[4][2], bfr[2][2][2];
    if constexpr (ABL & 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { af[i][ks] = bf16x8{1, 2, 3, 4, 5, 6, 7, (short)lane}; bfr[i >> 1][i & 1][ks] = af[i][ks]; }
    }
    // (NKS is a compile-time constant: with a run-time k-half count hipcc turns the fragment arrays into scratch memory)
    auto read_a = [&](int buf, int a, auto nks_c) {
        constexpr int NKS = decltype(nks_c)::value;
        if constexpr (ABL & 2) return;
        const char* base = smem + buf * C::BUF_BYTES + a_base + a * 64 * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (a == 1 && i >= C::TM1) break;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) af[i][ks] = *(const bf16x8*)(base + i * 16 * 128 + foff[ks]);
        }
    };
    auto read_b = [&](int buf, auto nks_c) {
        constexpr int NKS = decltype(nks_c)::value;
        if constexpr (ABL & 2) return;
        const char* base = smem + buf * C::BUF_BYTES + b_base;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) bfr[b][j][ks] = *(const bf16x8*)(base + (b * 32 + j * 16) * 128 + foff[ks]);
    };
    auto half = [&](auto a_c, auto nks_c) {                                   // 32 MFMAs: output rows a*64 .. a*64+63 of the wave
        constexpr int NKS = decltype(nks_c)::value, a = decltype(a_c)::value;
        if constexpr (ABL & 4) return;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < (a == 0 ? C::TM0 : C::TM1); ++i)
                        acc[2 * b + j][4 * a + i] = mfma16(bfr[b][j][ks], af[i][ks], acc[2 * b + j][4 * a + i]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    auto ktile = [&](int T, int cur, auto nks_c, auto fast_c) {
        constexpr bool FAST = decltype(fast_c)::value != 0;                  // T + 2 < nk: one branch-free basic block
        read_a(cur, 0, nks_c); read_b(cur, nks_c);
        pp_wait_lgkm0(); bar();
        half(I0{}, nks_c); bar();
        read_a(cur, 1, nks_c);
        if constexpr (FAST) { stage_b(T + 2, cur); pp_wait_vm4(); }
        else if (T + 2 < nt) { stage_b_any(T + 2, cur); pp_wait_vm4(); }
        else if (AUXPRE && aux_on) { stage_aux_b(cur); pp_wait_vm4(); }
        else { pp_wait_vm0(); }
        pp_wait_lgkm0(); bar();
        if constexpr (FAST) stage_a(T + 2, cur); else if (T + 2 < nt) stage_a_any(T + 2, cur); else if (AUXPRE && aux_on) stage_aux_a(cur);
        half(I1{}, nks_c); bar();
    };

    // prologue: K-tiles 0 and 1 in the steady-state issue order
    stage_b_any(0, 0); stage_a_any(0, 0);
    if (nt > 1) {
        stage_b_any(1, 1); stage_a_any(1, 1);
        if (C::A_INSTR == 16 || a_last) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // K-tile 1 = this wave's last 8 (or 7) instructions
        else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    } else { pp_wait_vm0(); }
    bar();
    if (wm == 1) bar();                                                      // wave row 1 runs one barrier behind
    int cur = 0;
    int T = 0;
    for (; T + 2 < nk; ++T) { ktile(T, cur, I2{}, I1{}); cur ^= 1; }
    for (; T < nk; ++T) { ktile(T, cur, I2{}, I0{}); cur ^= 1; }
    for (; T < nt; ++T) { ktile(T, cur, I1{}, I0{}); cur ^= 1; }             // LoRA half-steps (32 k)
    if (wm == 0) bar();                                                      // re-align the two wave rows
}


// ---------------------------------------------------------------------------------------------------------------------
// PERSISTENT form of the ping-pong loop: one workgroup per CU walks output tiles v = blockIdx.x, + gridDim.x, ... and the K-tile
// stream never stops at a tile boundary: while a tile's last two K-tiles are multiplied, K-tiles 0 and 1 of the NEXT tile are
// already being staged (they play the role of K-tiles nk and nk + 1 of the stream), so a tile's first MFMA phase starts right
// after its predecessor's epilogue -- no first-operand round trip (~2 us) and no workgroup dispatch (~1 us) per tile.
//
// What makes this safe on gfx9's single in-order vmcnt (r02's persistent attempts lost to it): no VGPR-destination global load
// and no scratch access sits between LDS-DMA issue and the counted waits inside the K loop; the epilogue's own loads / stores are
// issued AFTER the prefetch of K-tile 1' and are older than K-tile 2' only, whose wait (vmcnt(4) in the next tile's first K-tile)
// comes a whole MFMA phase later, when stores issued ~1 us earlier have long been acknowledged (0.4 us, r02 trace).  The tile
// sequence is static (no queue word to load), the bias enters through the accumulator init as before.
// Requirements: K a multiple of 64 with K >= 192 (nk >= 3), no low-rank K extension (merged weights), gridDim.x a multiple of 8.
//   tile_of(v, m0, m_end, n0, Bw): geometry of tile v;  init_acc(n0);  epilogue(m0, m_end, n0)  (may use no LDS).
template <int BM, int BN, typename TileOf, typename InitAcc, typename Epilogue, typename Trace>
__device__ __forceinline__ void stream_pp(const bf16_t* __restrict__ A, int lda, int ldb, int N, int K, int n_tiles, char* smem,
                                          bool perm_b, TileOf tile_of, InitAcc init_acc, Epilogue epilogue, Trace trace) {
    using C = PPCfg<BM, BN>;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = K >> 6;
    const int rl = lane >> 3, cl = lane & 7;
    const bool a_last = 12 + wn < C::A_INSTR;                                  // wave-uniform (224-row tile: waves 2, 3 skip j = 3)

    // geometry (scalars) of a tile and the per-lane source offsets of the tile whose K-tiles are being STAGED
    int m0 = 0, m_end = 0, n0 = 0;
    const bf16_t* Bt = nullptr;                                                // geometry: weight matrix of (m0, n0)'s row group
    const bf16_t* Bw = nullptr;                                                // staging: weight matrix the offsets below refer to
    uint32_t offA[4], offB[2][2];
    auto set_offsets = [&]() {
        Bw = Bt;
        // lane id recomputed here (v_mbcnt) instead of kept live across the K loop: hipcc spilled the hoisted lane constants to scratch
        // and reloaded them at this point -- a VMEM reload in the middle of the counted LDS-DMA stream
        const int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const int rl = ln >> 3, cl = ln & 7;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (j * 4 + wn) * 8 + rl;
            int ga = m0 + C::RW * wm + r;
            ga = ga < m_end - 1 ? ga : m_end - 1;
            offA[j] = (uint32_t)ga * (uint32_t)lda * 2u + (uint32_t)swz(r, cl) * 16u;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = (j * 8 + wave) * 8 + rl;
                const int rb = 128 * h + r;
                int gb = n0 + (perm_b ? perm32(rb) : rb);
                gb = gb < N - 1 ? gb : N - 1;
                offB[h][j] = (uint32_t)gb * (uint32_t)ldb * 2u + (uint32_t)swz(r, cl) * 16u;
            }
    };
    auto stage_a = [&](int t, int buf) {
        char* dst = smem + buf * C::BUF_BYTES + wm * C::HALF_BYTES;
        const char* src = (const char*)A + (size_t)t * 128;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + offA[j]), (lptr_t)(dst + (j * 4 + wn) * 8 * 128), 16, 0, 0);
        if (C::A_INSTR == 16 || a_last)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + offA[3]), (lptr_t)(dst + (12 + wn) * 8 * 128), 16, 0, 0);
    };
    auto stage_b = [&](int t, int buf) {
        const char* src = (const char*)Bw + (size_t)t * 128;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + offB[h][j]),
                                                 (lptr_t)(smem + buf * C::BUF_BYTES + (2 + h) * C::HALF_BYTES + (j * 8 + wave) * 8 * 128), 16, 0, 0);
    };
    const int frow = lane & 15, fq = lane >> 4;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = frow * 128 + ((swz(frow, ks * 4 + fq)) << 4);
    const int a_base = wm * C::HALF_BYTES;
    const int b_base = (2 + (wn >> 1)) * C::HALF_BYTES + (wn & 1) * 64 * 128;
    f32x4 acc[4][C::TM];
    bf16x8 af[4][2], bfr[2][2][2];
    auto read_a = [&](int buf, int a) {
        const char* base = smem + buf * C::BUF_BYTES + a_base + a * 64 * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (a == 1 && i >= C::TM1) break;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[i][ks] = *(const bf16x8*)(base + i * 16 * 128 + foff[ks]);
        }
    };
    auto read_b = [&](int buf) {
        const char* base = smem + buf * C::BUF_BYTES + b_base;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) bfr[b][j][ks] = *(const bf16x8*)(base + (b * 32 + j * 16) * 128 + foff[ks]);
    };
    auto half = [&](auto a_c) {
        constexpr int a = decltype(a_c)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < (a == 0 ? C::TM0 : C::TM1); ++i)
                        acc[2 * b + j][4 * a + i] = mfma16(bfr[b][j][ks], af[i][ks], acc[2 * b + j][4 * a + i]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    // one K-tile of the stream; `st` >= 0: K-tile index (of the tile described by offA / offB / Bw) staged into the buffer being
    // freed, st < 0: nothing left to stage (the stream ends)
    auto ktile = [&](int cur, int st) {
        read_a(cur, 0); read_b(cur);
        pp_wait_lgkm0(); bar();
        half(I0{}); bar();
        read_a(cur, 1);
        if (st >= 0) { stage_b(st, cur); pp_wait_vm4(); } else { pp_wait_vm0(); }
        pp_wait_lgkm0(); bar();
        if (st >= 0) stage_a(st, cur);
        half(I1{}); bar();
    };
    auto ktile_fast = [&](int cur, int st) {                                 // steady state: one branch-free basic block
        read_a(cur, 0); read_b(cur);
        pp_wait_lgkm0(); bar();
        half(I0{}); bar();
        read_a(cur, 1);
        stage_b(st, cur); pp_wait_vm4();
        pp_wait_lgkm0(); bar();
        stage_a(st, cur);
        half(I1{}); bar();
    };

    int v = blockIdx.x;
    tile_of(v, m0, m_end, n0, Bt);
    set_offsets();
    stage_b(0, 0); stage_a(0, 0);
    stage_b(1, 1); stage_a(1, 1);
    if (C::A_INSTR == 16 || a_last) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    bar();
    int cur = 0;
    for (;;) {
        // Inside a tile wave row 1 runs ONE BARRIER behind row 0 (the ping-pong); around the epilogue the rows are re-aligned (one extra
        // barrier each) so that both rows store at the same time -- without it row 1's last barrier would wait for row 0's whole epilogue
        // and the two epilogues would run one after the other.
        if (wm == 1) bar();
        const int cm0 = m0, cm_end = m_end, cn0 = n0;                        // the tile being COMPUTED (scalars)
        trace(v, 0);
        init_acc(acc, cn0);
        trace(v, 1);
        // geometry of the NEXT tile now, at the start of the K loop: its scalar loads (tile table in the kernel arguments) and integer
        // divisions have a whole K loop to complete; done at the switch point they held every wave for 0.8-1.5 us (r03 trace)
        // (the 256-row tile has no registers to spare for four more live scalars -- the compiler spills to scratch, and a scratch reload
        //  inside the counted LDS-DMA stream is worse than the stall: there the geometry is computed at the switch point)
        constexpr bool EARLY = BM < 256;
        const int vn = v + (int)gridDim.x;
        const bool has_next = vn < n_tiles;                                  // workgroup-uniform
        if (EARLY && has_next) tile_of(vn, m0, m_end, n0, Bt);
        for (int T = 0; T + 2 < nk; ++T) { ktile_fast(cur, T + 2); cur ^= 1; }
        trace(v, 2);
        if (!EARLY && has_next) tile_of(vn, m0, m_end, n0, Bt);
        if (has_next) set_offsets();                                         // from here on the staged K-tiles belong to the next tile
        trace(v, 3);
        ktile(cur, has_next ? 0 : -1); cur ^= 1;
        ktile(cur, has_next ? 1 : -1); cur ^= 1;
        if (wm == 0) bar();
        trace(v, 4);
        epilogue(acc, cm0, cm_end, cn0);
        trace(v, 5);
        if (!has_next) break;
        v = vn;
    }
}


}  // namespace gemmcore
