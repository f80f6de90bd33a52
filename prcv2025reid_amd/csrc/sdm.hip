// Fused SDM loss (sdm_loss_stable, models/sdm_loss.py:13-149) for gfx950, all modality pairs of a step in ONE launch.
//
// The reference builds S = q^ g^T / tau as an [N, M] matrix, takes a row-wise and a column-wise soft-target cross-entropy
// over it and averages (models/sdm_loss.py:86,34-70,121-123), once per non-vis modality (models/model.py:586-622).  Here
//   * the P query sides (one per non-vis modality) are STACKED: rows [p*N, (p+1)*N) belong to pair p, the gallery side (vis) is
//     shared; one grid covers P x tiles_m x tiles_n tiles;
//   * a tile of S lives only in MFMA accumulators: v_mfma_f32_32x32x2_f32 on the fp32 unit vectors (exact fp32: the result is
//     bit-for-bit an fmaf chain, so "within 1e-5 of the fp32 reference" holds without any operand splitting; on gfx950 this
//     instruction runs at the fp32 vector rate = the rate a 3-6-piece bf16 split of the same product would reach);
//     operands are staged by LDS-DMA in 128-byte rows exactly as the bf16 GEMM stages its tiles (gemm_core.h);
//   * forward epilogue: per element valid / exp / positive test, per-row sums over the tile's columns (xor butterflies inside
//     the 32-lane halves) and per-column sums over its rows (register sums), written as per-tile PARTIALS that a tiny second
//     kernel adds in a fixed order -> deterministic loss, no atomics, no [N, M] array anywhere;
//   * backward recomputes the tile, forms dS in registers, parks it in LDS (fp32, row stride 129 floats: conflict-free for
//     both the row-major and the column-major operand read) and multiplies it on the matrix cores into dq^ (+= dS g^) and
//     dg^ (+= dS^T q^), accumulated across tiles with fp32 atomics (128-byte row segments, the fast shape of
//     MI355X_MICROARCH.md "Global float atomics").
// Workspace: O((P N + M) (D + tiles)) floats -- see reid_sdm_ws_floats.
#include "gemm_core.h"

namespace {
using namespace gemmcore;

struct SdmParams {
    const float* qn; const float* gn;                  // unit rows [P*N, D], [M, D]
    const int64_t* qlab; const int64_t* glab;          // [N] (shared by the pairs), [M]
    const uint8_t* qvalid; const uint8_t* gvalid;      // [P*N] / [M] or null
    int P, N, M, D, tiles_m, tiles_n;
    float inv_tau;
    float* rpart; float* cpart;                        // fwd: [tiles_n][P*N][4], [P*tiles_m][M][4]  {sum exp, sum pos score, #pos, -}
    const float* rstat; const float* cstat;            // bwd: [P*N][2], [P][M][2]  {lse, #pos}
    const float* acc; const float* gscale;             // bwd: [P][4] {sum_r, cnt_r, sum_c, cnt_c}, upstream gradient per pair [P]
    float* dqn; float* dgn;                            // bwd: [P*N, D], [M, D] accumulators (zeroed by the caller)
};

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__device__ __forceinline__ f32x16_t mfma_f32(float a, float b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// sum over the 32 lanes of this lane's half-wave (lanes 0-31 / 32-63): the MFMA column index lives on lane & 31
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// TS x TS tile, 4 waves as 2 x 2, each wave (TS/2) x (TS/2) = WT x WT made of (WT/32)^2 MFMA tiles.
template <int TS>
struct Geo {
    static constexpr int WT = TS / 2, NT32 = WT / 32;
    static constexpr int STAGE = 2 * TS * 128;                      // one K-step (32 floats) of both operands
    static constexpr int LDS_STAGE = 2 * STAGE;                     // double buffered
    static constexpr int DS_LD = TS + 1;                            // padded dS row (floats)
    static constexpr int LDS_SIDE = 8 * TS * 4 + 2 * TS * 8;        // row / column scratch {3 floats x 2 waves}, labels
    static constexpr int LDS_FWD = LDS_STAGE > LDS_SIDE ? LDS_STAGE : LDS_SIDE;
    static constexpr int DS_BYTES = TS * DS_LD * 4;                  // the dS tile takes over the staging area after the K loop
    static constexpr int LDS_BWD = (DS_BYTES > LDS_STAGE ? DS_BYTES : LDS_STAGE) + 32 * TS;
};

// S tile: acc[a][b][reg] = q^[m] . g^[n],  m = wm*WT + 32a + (reg&3) + 8(reg>>2) + 4(lane>>5),  n = wn*WT + 32b + (lane&31)
template <int TS>
__device__ __forceinline__ void s_tile(const SdmParams& p, int row_lo, int row_hi, int m0, int n0, char* smem,
                                       f32x16_t (&acc)[Geo<TS>::NT32][Geo<TS>::NT32]) {
    using G = Geo<TS>;
    constexpr int NW = 4, INSTR = TS / 8 / NW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int steps = p.D >> 5;
    uint32_t offA[INSTR], offB[INSTR];
    // rows are D fp32 = 2D 16-bit units wide for the byte arithmetic of gemm_core's staging
    stage_offsets<TS, NW, false>(2 * p.D, row_lo + m0, row_hi, wave, lane, offA);
    stage_offsets<TS, NW, false>(2 * p.D, n0, p.M - 1, wave, lane, offB);
    auto issue = [&](int t, int buf) {
        char* la = smem + buf * G::STAGE;
        char* lb = la + TS * 128;
        stage_from<TS, NW>((const char*)p.qn + (size_t)t * 128, offA, la, wave);
        stage_from<TS, NW>((const char*)p.gn + (size_t)t * 128, offB, lb, wave);
    };
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int a = 0; a < G::NT32; ++a)
#pragma unroll
        for (int b = 0; b < G::NT32; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    auto compute = [&](int buf) {
        const char* la = smem + buf * G::STAGE;
        const char* lb = la + TS * 128;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // lane (i, h) takes the 16-byte chunk 2t+h of its row: element u of it is k = 4(2t+h)+u for BOTH operands
            f32x4 af[G::NT32], bf[G::NT32];
#pragma unroll
            for (int a = 0; a < G::NT32; ++a) {
                const int row = wm * G::WT + 32 * a + i;
                af[a] = *(const f32x4*)(la + row * 128 + (swz(row, 2 * t + h) << 4));
            }
#pragma unroll
            for (int b = 0; b < G::NT32; ++b) {
                const int row = wn * G::WT + 32 * b + i;
                bf[b] = *(const f32x4*)(lb + row * 128 + (swz(row, 2 * t + h) << 4));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int a = 0; a < G::NT32; ++a)
#pragma unroll
                    for (int b = 0; b < G::NT32; ++b) acc[a][b] = mfma_f32(af[a][u], bf[b][u], acc[a][b]);
        }
    };
    issue(0, 0);
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < steps - 1; ++t) {
        issue(t + 1, cur ^ 1);
        compute(cur);
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);
    __syncthreads();                                   // the staging buffers are free for the epilogue's scratch
}

// tile -> (pair, row tile, column tile)
__device__ __forceinline__ void tile_of(const SdmParams& p, int& pair, int& tm, int& tn) {
    const int per_pair = p.tiles_m * p.tiles_n;
    pair = blockIdx.x / per_pair;
    const int r = blockIdx.x - pair * per_pair;
    tm = r / p.tiles_n; tn = r - tm * p.tiles_n;
}

template <int TS>
__global__ __launch_bounds__(256, 2) void sdm_fwd_tile_kernel(const SdmParams p) {
    using G = Geo<TS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int pair, tm, tn;
    tile_of(p, pair, tm, tn);
    const int row_lo = pair * p.N, m0 = tm * TS, n0 = tn * TS;
    f32x16_t acc[G::NT32][G::NT32];
    s_tile<TS>(p, row_lo, row_lo + p.N - 1, m0, n0, smem, acc);

    // scratch in the (now idle) staging area: labels / validity of the tile's rows and columns, then the cross-wave sums
    int64_t* l_ql = (int64_t*)smem; int64_t* l_gl = l_ql + TS;
    float* l_row = (float*)(l_gl + TS);                // [2 (wn)][TS][4]
    float* l_col = l_row + 2 * TS * 4;                 // [2 (wm)][TS][4]
    // validity is folded into the labels: an invalid row gets label -1, an invalid column -2 -> never "positive", and the
    // separate valid flags below gate the exp sums
    uint8_t* l_qv = (uint8_t*)(l_col + 2 * TS * 4); uint8_t* l_gv = l_qv + TS;
    if (tid < TS) {
        const int m = m0 + tid;
        const bool ok = m < p.N && (!p.qvalid || p.qvalid[row_lo + m]);
        l_ql[tid] = p.qlab[m < p.N ? m : p.N - 1];
        l_qv[tid] = ok ? 1 : 0;
    } else if (tid < 2 * TS) {
        const int c = tid - TS, n = n0 + c;
        const bool ok = n < p.M && (!p.gvalid || p.gvalid[n]);
        l_gl[c] = p.glab[n < p.M ? n : p.M - 1];
        l_gv[c] = ok ? 1 : 0;
    }
    __syncthreads();
    const int j = lane & 31, h = lane >> 5;
    // Per accumulator row: this lane's elements (one per column tile b) are summed, the 32 lanes of the half-wave are
    // combined by an xor butterfly at once (nothing but the six column sums stays live across rows: the kernel keeps to
    // 2 waves per SIMD, i.e. two workgroups per CU, which the barrier-synchronised K loop needs to keep the MFMA pipe busy).
    float cs[G::NT32][3];
    bool gv[G::NT32]; int64_t gl[G::NT32];
#pragma unroll
    for (int b = 0; b < G::NT32; ++b) {
        const int nl = wn * G::WT + 32 * b + j;
        gv[b] = l_gv[nl] != 0; gl[b] = l_gl[nl];
        cs[b][0] = cs[b][1] = cs[b][2] = 0.f;
    }
#pragma unroll
    for (int a = 0; a < G::NT32; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = wm * G::WT + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
            const bool qv = l_qv[ml] != 0;
            const int64_t ql = l_ql[ml];
            float rs[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < G::NT32; ++b) {
                const bool ok = gv[b] && qv;
                const float v = fminf(fmaxf(acc[a][b][r] * p.inv_tau, -20.f), 20.f);       // models/sdm_loss.py:94
                const float e = ok ? __expf(v) : 0.f;
                const bool pos = ok && ql == gl[b];
                const float pv = pos ? v : 0.f, pc = pos ? 1.f : 0.f;
                cs[b][0] += e; cs[b][1] += pv; cs[b][2] += pc;
                rs[0] += e; rs[1] += pv; rs[2] += pc;
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float t = half_sum(rs[q]);
                if (j == 0) l_row[(wn * TS + ml) * 4 + q] = t;
            }
        }
#pragma unroll
    for (int b = 0; b < G::NT32; ++b) {
        const int nl = wn * G::WT + 32 * b + j;
#pragma unroll
        for (int q = 0; q < 3; ++q) cs[b][q] += __shfl_xor(cs[b][q], 32, 64);            // both halves: all rows of the a-tiles
        if (h == 0) {
#pragma unroll
            for (int q = 0; q < 3; ++q) l_col[(wm * TS + nl) * 4 + q] = cs[b][q];
        }
    }
    __syncthreads();
    if (tid < TS) {
        const int m = m0 + tid;
        if (m < p.N) {
            float* o = p.rpart + ((size_t)tn * p.P * p.N + row_lo + m) * 4;
#pragma unroll
            for (int q = 0; q < 3; ++q) o[q] = l_row[tid * 4 + q] + l_row[(TS + tid) * 4 + q];
        }
    } else if (tid < 2 * TS) {
        const int c = tid - TS, n = n0 + c;
        if (n < p.M) {
            float* o = p.cpart + ((size_t)(pair * p.tiles_m + tm) * p.M + n) * 4;
#pragma unroll
            for (int q = 0; q < 3; ++q) o[q] = l_col[c * 4 + q] + l_col[(TS + c) * 4 + q];
        }
    }
}

// fixed-order sums of the per-tile partials -> {lse, #pos} and the row's / column's loss term
__global__ __launch_bounds__(256) void sdm_stat_kernel(const float* __restrict__ rpart, const float* __restrict__ cpart, int P, int N, int M,
                                                       int tiles_m, int tiles_n, float* __restrict__ rstat, float* __restrict__ cstat,
                                                       float* __restrict__ row_loss, float* __restrict__ col_loss) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int nr = P * N, nc = P * M;
    if (t < nr) {
        float se = 0.f, ps = 0.f, np = 0.f;
        for (int k = 0; k < tiles_n; ++k) {
            const float* s = rpart + ((size_t)k * nr + t) * 4;
            se += s[0]; ps += s[1]; np += s[2];
        }
        const float lse = np > 0.f ? logf(se) : 0.f;
        rstat[2 * t] = lse; rstat[2 * t + 1] = np;
        row_loss[t] = np > 0.f ? lse - ps / np : 0.f;
    } else if (t < nr + nc) {
        const int c = t - nr, pair = c / M, n = c - pair * M;
        float se = 0.f, ps = 0.f, np = 0.f;
        for (int k = 0; k < tiles_m; ++k) {
            const float* s = cpart + ((size_t)(pair * tiles_m + k) * M + n) * 4;
            se += s[0]; ps += s[1]; np += s[2];
        }
        const float lse = np > 0.f ? logf(se) : 0.f;
        cstat[2 * c] = lse; cstat[2 * c + 1] = np;
        col_loss[c] = np > 0.f ? lse - ps / np : 0.f;
    }
}

// one workgroup per pair: acc[pair] = {sum_q2g, cnt_q2g, sum_g2q, cnt_g2q}; result[pair] = {0.5 (mean + mean), contributes}
__global__ __launch_bounds__(1024) void sdm_reduce_kernel(const float* __restrict__ rstat, const float* __restrict__ cstat,
                                                         const float* __restrict__ row_loss, const float* __restrict__ col_loss, int N, int M,
                                                         float* __restrict__ acc, float* __restrict__ result) {
    __shared__ float red[16][4];
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, nt = blockDim.x;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = tid; r < N; r += nt)
        if (rstat[2 * (pair * N + r) + 1] > 0.f) { v[0] += row_loss[pair * N + r]; v[1] += 1.f; }
    for (int c = tid; c < M; c += nt)
        if (cstat[2 * (pair * M + c) + 1] > 0.f) { v[2] += col_loss[pair * M + c]; v[3] += 1.f; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = wave_sum(v[k]);
        if (lane == 0) red[tid >> 6][k] = v[k];
    }
    __syncthreads();
    if (tid == 0) {
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            t[k] = 0.f;
            for (int w = 0; w < nt / 64; ++w) t[k] += red[w][k];
            acc[pair * 4 + k] = t[k];
        }
        const float a = t[1] > 0.f ? t[0] / t[1] : 0.f;
        const float b = t[3] > 0.f ? t[2] / t[3] : 0.f;
        const bool any = t[1] > 0.f;                       // models/sdm_loss.py:105-106: no row with a positive -> 0
        result[2 * pair] = any ? 0.5f * (a + b) : 0.f;
        result[2 * pair + 1] = any ? 1.f : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------- backward
// out[TS rows, 32-wide d chunk] += A[TS, TS] . B[TS rows of `src`, chunk]  on v_mfma_f32_32x32x2_f32, A from the LDS dS tile:
//   TRANS = false: A[r][k] = dS[r][k]   (dq^ += dS g^ :  r = query row, k = gallery column, src = g^ rows n0+k)
//   TRANS = true : A[r][k] = dS[k][r]   (dg^ += dS^T q^: r = gallery column, k = query row, src = q^ rows)
template <int TS, bool TRANS>
__device__ __forceinline__ void ds_times(const float* __restrict__ ds, const float* __restrict__ src, int src_row0, int src_row_max, int D,
                                         float* __restrict__ out, int out_row0, int out_rows_valid, int wave, int lane) {
    using G = Geo<TS>;
    constexpr int RT = TS / 32;                          // 32-row output tiles
    const int i = lane & 31, h = lane >> 5;
    for (int c = wave; c < (D >> 5); c += 4) {
        f32x16_t acc[RT];
#pragma unroll
        for (int a = 0; a < RT; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        const float* bcol = src + 32 * c + i;
#pragma unroll 4
        for (int s = 0; s < TS / 2; ++s) {
            const int k = 2 * s + h;
            int sr = src_row0 + k;
            sr = sr < src_row_max ? sr : src_row_max;
            const float bv = bcol[(size_t)sr * D];
#pragma unroll
            for (int a = 0; a < RT; ++a) {
                const float av = TRANS ? ds[k * G::DS_LD + 32 * a + i] : ds[(32 * a + i) * G::DS_LD + k];
                acc[a] = mfma_f32(av, bv, acc[a]);
            }
        }
#pragma unroll
        for (int a = 0; a < RT; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < out_rows_valid) atomicAdd(out + (size_t)(out_row0 + row) * D + 32 * c + i, acc[a][r]);
            }
    }
}

template <int TS>
__global__ __launch_bounds__(256, TS == 128 ? 1 : 2) void sdm_bwd_tile_kernel(const SdmParams p) {      // (128: 2 waves/SIMD would spill)
    using G = Geo<TS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int pair, tm, tn;
    tile_of(p, pair, tm, tn);
    const float cnt_r = p.acc[pair * 4 + 1], cnt_c = p.acc[pair * 4 + 3];
    if (!(cnt_r > 0.f)) return;                          // the pair contributes nothing (uniform over the workgroup)
    const int row_lo = pair * p.N, m0 = tm * TS, n0 = tn * TS;
    f32x16_t acc[G::NT32][G::NT32];
    s_tile<TS>(p, row_lo, row_lo + p.N - 1, m0, n0, smem, acc);

    float* ds = (float*)smem;                           // [TS][TS + 1], over the (idle) staging buffers
    int64_t* l_ql = (int64_t*)(smem + G::LDS_BWD - 32 * TS); int64_t* l_gl = l_ql + TS;
    float* l_rs = (float*)(l_gl + TS);                  // [TS][2] {lse, #pos} of the rows; #pos < 0 marks an invalid row
    float* l_cs = l_rs + 2 * TS;                        // [TS][2] of the columns
    if (tid < TS) {
        const int m = m0 + tid;
        const bool ok = m < p.N && (!p.qvalid || p.qvalid[row_lo + m]);
        const int mc = row_lo + (m < p.N ? m : p.N - 1);
        l_ql[tid] = p.qlab[mc - row_lo];
        l_rs[2 * tid] = p.rstat[2 * mc];
        l_rs[2 * tid + 1] = ok ? p.rstat[2 * mc + 1] : -1.f;
    } else if (tid < 2 * TS) {
        const int c = tid - TS, n = n0 + c;
        const bool ok = n < p.M && (!p.gvalid || p.gvalid[n]);
        const int nc = pair * p.M + (n < p.M ? n : p.M - 1);
        l_gl[c] = p.glab[nc - pair * p.M];
        l_cs[2 * c] = p.cstat[2 * nc];
        l_cs[2 * c + 1] = ok ? p.cstat[2 * nc + 1] : -1.f;
    }
    __syncthreads();
    const int j = lane & 31, h = lane >> 5;
    const float ir = 1.f / cnt_r, ic = cnt_c > 0.f ? 1.f / cnt_c : 0.f;
    const float gsc = 0.5f * p.gscale[pair] * p.inv_tau;
#pragma unroll
    for (int b = 0; b < G::NT32; ++b) {
        const int nl = wn * G::WT + 32 * b + j;
        const float lse_c = l_cs[2 * nl], np_c = l_cs[2 * nl + 1];
        const int64_t gl = l_gl[nl];
#pragma unroll
        for (int a = 0; a < G::NT32; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm * G::WT + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float lse_r = l_rs[2 * ml], np_r = l_rs[2 * ml + 1];
                const float raw = acc[a][b][r] * p.inv_tau;
                const float v = fminf(fmaxf(raw, -20.f), 20.f);
                float t = 0.f;
                if (np_r >= 0.f && np_c >= 0.f) {                       // both valid
                    const float y = l_ql[ml] == gl ? 1.f : 0.f;
                    if (np_r > 0.f) t += (__expf(v - lse_r) - y / np_r) * ir;
                    if (np_c > 0.f && ic > 0.f) t += (__expf(v - lse_c) - y / np_c) * ic;
                    t = (raw > -20.f && raw < 20.f) ? t * gsc : 0.f;    // the clamp's gradient
                }
                ds[ml * G::DS_LD + nl] = t;
            }
    }
    __syncthreads();
    const int rows_valid = min(TS, p.N - m0), cols_valid = min(TS, p.M - n0);
    ds_times<TS, false>(ds, p.gn, n0, p.M - 1, p.D, p.dqn, row_lo + m0, rows_valid, wave, lane);
    ds_times<TS, true>(ds, p.qn, row_lo + m0, row_lo + p.N - 1, p.D, p.dgn, n0, cols_valid, wave, lane);
}

// y = x / max(||x||, eps) -> dx += (dy - y (y . dy)) / max(||x||, eps); `rep` source rows share one destination row block
// (the gallery side is shared by the pairs; each query row has its own).  One wave per row.
__global__ __launch_bounds__(256) void sdm_unnorm_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, float* __restrict__ dx,
                                                         int lddx, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx; const float* g = dy + (size_t)row * D;
    float ss = 0.f, dot = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 a = *(const f32x4*)(xr + c), b = *(const f32x4*)(g + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { ss += a[e] * a[e]; dot += a[e] * b[e]; }
    }
    const float n = fmaxf(sqrtf(wave_sum(ss)), eps), rn = 1.f / n;
    dot = wave_sum(dot) * rn * rn;
    float* d = dx + (size_t)row * lddx;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 a = *(const f32x4*)(xr + c), b = *(const f32x4*)(g + c);
        f32x4 o = *(const f32x4*)(d + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += (b[e] - a[e] * dot) * rn;
        *(f32x4*)(d + c) = o;
    }
}

struct WsLayout { int64_t qn, gn, rstat, cstat, acc, rloss, closs, rpart, cpart, dqn, dgn, total; int tiles_m, tiles_n, ts; };
WsLayout ws_layout(int P, int N, int M, int D) {
    WsLayout w;
    w.ts = (N <= 512 && M <= 512) ? 64 : 128;          // small problems: more, smaller tiles (the grid is tiny either way)
    w.tiles_m = (N + w.ts - 1) / w.ts; w.tiles_n = (M + w.ts - 1) / w.ts;
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t at = o; o += (n + 3) & ~(int64_t)3; return at; };
    w.qn = take((int64_t)P * N * D); w.gn = take((int64_t)M * D);
    w.rstat = take(2LL * P * N); w.cstat = take(2LL * P * M); w.acc = take(4LL * P);
    w.rloss = take((int64_t)P * N); w.closs = take((int64_t)P * M);
    // forward partials and backward accumulators never live at the same time: they share one region
    const int64_t part = 4LL * w.tiles_n * P * N + 4LL * P * w.tiles_m * M;
    const int64_t grad = (int64_t)P * N * D + (int64_t)M * D;
    const int64_t at = take(part > grad ? part : grad);
    w.rpart = at; w.cpart = at + 4LL * w.tiles_n * P * N;
    w.dqn = at; w.dgn = at + (int64_t)P * N * D;
    w.total = o;
    return w;
}

}  // namespace

extern "C" int reid_l2norm_rows(const float* x, int32_t ldx, float* y, void* y_bf16, int32_t ldy, int32_t rows, int32_t cols, float eps,
                                float scale, void* stream);

// shape / size rules shared by the forward and the backward entry points (32-bit element offsets inside the kernels)
static int sdm_check_shape(const char* who, int32_t P, int32_t N, int32_t Mg, int32_t D) {
    REID_CHECK_ARG(P > 0 && N > 0 && Mg > 0 && D % 32 == 0 && D >= 32 && D <= 1024, "%s: shape P=%d N=%d Mg=%d D=%d (D %% 32, 32..1024)", who, P, N, Mg, D);
    REID_CHECK_ARG((int64_t)P * N * D * 4 < (1ll << 32) && (int64_t)Mg * D * 4 < (1ll << 32), "%s: operands beyond 4 GiB", who);
    return REID_OK;
}

// workspace (floats): q^ [P N D] | g^ [M D] | rstat [2 P N] | cstat [2 P M] | acc [4 P] | row / column loss terms [P N + P M] |
//                     max( per-tile partials 4 (tiles_n P N + P tiles_m M),  gradient accumulators (P N + M) D )
extern "C" int64_t reid_sdm_ws_floats(int32_t P, int32_t N, int32_t Mg, int32_t D) { return ws_layout(P, N, Mg, D).total; }

extern "C" int reid_sdm_fwd(const float* q, int32_t ldq, const float* g, int32_t ldg, const int64_t* q_label, const int64_t* g_label,
                            const uint8_t* q_valid, const uint8_t* g_valid, int32_t P, int32_t N, int32_t Mg, int32_t D, float tau,
                            float* ws, float* result, void* stream) {
    REID_CHECK_ARG(q && g && q_label && g_label && ws && result, "reid_sdm_fwd: null pointer");
    if (int rc = sdm_check_shape("reid_sdm_fwd", P, N, Mg, D)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const WsLayout w = ws_layout(P, N, Mg, D);
    int rc;
    if ((rc = reid_l2norm_rows(q, ldq, ws + w.qn, nullptr, D, P * N, D, 1e-8f, 1.0f, stream))) return rc;      // models/sdm_loss.py:31-32
    if ((rc = reid_l2norm_rows(g, ldg, ws + w.gn, nullptr, D, Mg, D, 1e-8f, 1.0f, stream))) return rc;
    SdmParams p{};
    p.qn = ws + w.qn; p.gn = ws + w.gn; p.qlab = q_label; p.glab = g_label; p.qvalid = q_valid; p.gvalid = g_valid;
    p.P = P; p.N = N; p.M = Mg; p.D = D; p.tiles_m = w.tiles_m; p.tiles_n = w.tiles_n;
    p.inv_tau = 1.0f / fminf(fmaxf(tau, 0.15f), 0.5f);                                                        // models/sdm_loss.py:28
    p.rpart = ws + w.rpart; p.cpart = ws + w.cpart;
    const int grid = P * w.tiles_m * w.tiles_n;
    if (w.ts == 64) {
        hipLaunchKernelGGL(sdm_fwd_tile_kernel<64>, dim3(grid), dim3(256), Geo<64>::LDS_FWD, s, p);
    } else {
        REID_MAX_LDS((sdm_fwd_tile_kernel<128>), Geo<128>::LDS_FWD);
        hipLaunchKernelGGL(sdm_fwd_tile_kernel<128>, dim3(grid), dim3(256), Geo<128>::LDS_FWD, s, p);
    }
    REID_CHECK_LAUNCH("reid_sdm_fwd(tiles)");
    const int nst = P * N + P * Mg;
    hipLaunchKernelGGL(sdm_stat_kernel, dim3((nst + 255) / 256), dim3(256), 0, s, ws + w.rpart, ws + w.cpart, P, N, Mg, w.tiles_m, w.tiles_n,
                       ws + w.rstat, ws + w.cstat, ws + w.rloss, ws + w.closs);
    REID_CHECK_LAUNCH("reid_sdm_fwd(stat)");
    hipLaunchKernelGGL(sdm_reduce_kernel, dim3(P), dim3(N + Mg > 2048 ? 1024 : 256), 0, s, ws + w.rstat, ws + w.cstat, ws + w.rloss, ws + w.closs,
                       N, Mg, ws + w.acc, result);
    REID_CHECK_LAUNCH("reid_sdm_fwd(reduce)");
    return REID_OK;
}

extern "C" int reid_sdm_bwd(const float* q, int32_t ldq, const float* g, int32_t ldg, const int64_t* q_label, const int64_t* g_label,
                            const uint8_t* q_valid, const uint8_t* g_valid, int32_t P, int32_t N, int32_t Mg, int32_t D, float tau,
                            float* ws, const float* gscale, float* dq, int32_t lddq, float* dg, int32_t lddg, void* stream) {
    REID_CHECK_ARG(q && g && q_label && g_label && ws && gscale && dq && dg, "reid_sdm_bwd: null pointer");
    if (int rc = sdm_check_shape("reid_sdm_bwd", P, N, Mg, D)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const WsLayout w = ws_layout(P, N, Mg, D);
    SdmParams p{};
    p.qn = ws + w.qn; p.gn = ws + w.gn; p.qlab = q_label; p.glab = g_label; p.qvalid = q_valid; p.gvalid = g_valid;
    p.P = P; p.N = N; p.M = Mg; p.D = D; p.tiles_m = w.tiles_m; p.tiles_n = w.tiles_n;
    p.inv_tau = 1.0f / fminf(fmaxf(tau, 0.15f), 0.5f);
    p.rstat = ws + w.rstat; p.cstat = ws + w.cstat; p.acc = ws + w.acc; p.gscale = gscale;
    p.dqn = ws + w.dqn; p.dgn = ws + w.dgn;
    REID_CHECK_HIP(hipMemsetAsync(ws + w.dqn, 0, ((size_t)P * N * D + (size_t)Mg * D) * sizeof(float), s), "reid_sdm_bwd: clearing the gradient accumulators");
    const int grid = P * w.tiles_m * w.tiles_n;
    if (w.ts == 64) {
        REID_MAX_LDS((sdm_bwd_tile_kernel<64>), Geo<64>::LDS_BWD);
        hipLaunchKernelGGL(sdm_bwd_tile_kernel<64>, dim3(grid), dim3(256), Geo<64>::LDS_BWD, s, p);
    } else {
        REID_MAX_LDS((sdm_bwd_tile_kernel<128>), Geo<128>::LDS_BWD);
        hipLaunchKernelGGL(sdm_bwd_tile_kernel<128>, dim3(grid), dim3(256), Geo<128>::LDS_BWD, s, p);
    }
    REID_CHECK_LAUNCH("reid_sdm_bwd(tiles)");
    hipLaunchKernelGGL(sdm_unnorm_kernel, dim3((P * N + 3) / 4), dim3(256), 0, s, q, ldq, ws + w.dqn, dq, lddq, P * N, D, 1e-8f);
    REID_CHECK_LAUNCH("reid_sdm_bwd(dq)");
    hipLaunchKernelGGL(sdm_unnorm_kernel, dim3((Mg + 3) / 4), dim3(256), 0, s, g, ldg, ws + w.dgn, dg, lddg, Mg, D, 1e-8f);
    REID_CHECK_LAUNCH("reid_sdm_bwd(dg)");
    return REID_OK;
}
