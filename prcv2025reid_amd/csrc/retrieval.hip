// Cosine retrieval with exact top-k on gfx950 (reference: train.py:499,463; eval_mm_protocol.py:50-53,
// 401-423): sim = Q.G^T on L2-normalised rows, ranking by (score desc, index asc).
//
// The 10k x 200k similarity matrix (8 GB in fp32) is never written.  Three phases:
//   A  threshold: the bf16 MFMA tile GEMM scores every query against a small gallery sample and keeps
//      the scores; one wavefront per query then finds the k-th best sample score s_k (LDS bitonic
//      select) and sets thr[q] = s_k - 2*eps.  eps = 2^-8 bounds |bf16 score - fp32 score| for unit
//      rows, so every member of the true top-k (ties included) scores >= thr in bf16.
//   B  filter: the same MFMA GEMM over the whole gallery; the epilogue appends (score, index) of every
//      entry >= thr[q] to the query's candidate list (global atomic counter; ~0.1 % of entries pass).
//   C  select: per query, candidates are re-scored in fp32 from the fp32 rows (bit-compatible with a
//      k-ordered fp32 dot product is not promised; the ORDER is: the fp32 score decides, index breaks
//      ties) and the best k are extracted in LDS.
// A query whose list overflowed is flagged (out_idx[q][0] = -2) and handled by the caller's exact
// fallback (brute-force kernel below), so the result never silently degrades.
#include "gemm_core.h"
#include <stdlib.h>

namespace {

using namespace gemmcore;

constexpr float EPS_BF16 = REID_T16_EPS * 1.01f;   // |q~.g~ - q.g| <= 2u (+1 %) for unit q, g rounded to the 16-bit format (Cauchy-Schwarz)

struct TopkParams {
    const bf16_t* Q; const bf16_t* G;
    int Nq, Ng, D;
    int g_begin, g_end;            // gallery slice scored by this launch
    const int32_t* exq; const int32_t* exg;
    const float* thr;              // [Nq] or null (phase A: keep everything)
    float* dense; int ld_dense;    // phase A: dense scores [Nq, g_end-g_begin]
    int32_t* cand_idx; float* cand_score; int32_t* cand_cnt; int cap;   // phase B
    int tiles_m, tiles_n;
    int dbg;                       // timing experiments (REID_TOPK_DBG): 1 = skip the compare epilogue
};

// Compare epilogue shared by the filter kernels.  A lane owns, per 16-row group i, ONE query row and 16 of its scores
// (4 sub-tiles x 4 columns).  Survivors (0.1 % of the scores, but ~300 per tile) are first collected as a bit mask per
// (lane, i); the per-(row, lane) counts then go through a 1 KiB wave-private LDS slice so that lane L of the wave owns ROW L
// of the wave's 64-row group: ONE atomicAdd wave-instruction per 64 rows reserves the slots of every row at once (64 lanes,
// 64 different counters), the bases come back through the same LDS slice, and only then are the survivors written.
// History (10k x 200k, 128x256 tiles): one dependent global atomic per survivor inside the compare loop 1.7 ms of the 3.4 ms
// filter pass; one atomic instruction per (i) 1.0 ms; this form: see DESIGN.md.  Global atomics cost ~50 ns of CU
// throughput per wave-instruction whatever the number of active lanes (MI355X_MICROARCH.md), so the lever is the
// instruction count, not the survivor count.
template <int TM, int TN>
__device__ __forceinline__ void filter_epilogue(const TopkParams& p, f32x4 (&acc)[TN][TM], const float (&th)[TM], int m_base, int n_base,
                                                int lane, int* wl /* >= 256 ints of wave-private LDS */) {
    static_assert(TM % 4 == 0 && TN <= 8, "64-row groups; survivor mask of TN*4 <= 32 bits");
    const int frow = lane & 15, fq = lane >> 4;
    unsigned msk[TM];
    int slot0[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m_base + i * 16 + frow;
        const bool mok = m < p.Nq;
        const int eq = p.exq ? p.exq[mok ? m : 0] : -1;
        unsigned b = 0;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n_base + j * 16 + fq * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[j][i][e];
                if (eq >= 0 && n + e < p.g_end && p.exg[n + e] == eq) { v = -1e9f; acc[j][i][e] = v; }
                if (mok && v >= th[i] && n + e < p.g_end) b |= 1u << (j * 4 + e);
            }
        }
        msk[i] = b;
    }
#pragma unroll
    for (int h = 0; h < TM / 4; ++h) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) wl[(ii * 16 + frow) * 4 + fq] = __builtin_popcount(msk[4 * h + ii]);
        __builtin_amdgcn_wave_barrier();
        typedef __attribute__((ext_vector_type(4))) int i32x4;
        const i32x4 c = *(const i32x4*)(wl + lane * 4);       // lane L: the four per-quarter counts of row L of this 64-row group
        const int tot = c[0] + c[1] + c[2] + c[3];
        const int base = tot ? atomicAdd(p.cand_cnt + (m_base + h * 64 + lane), tot) : 0;   // (tot > 0 implies the row is < Nq)
        __builtin_amdgcn_wave_barrier();
        *(i32x4*)(wl + lane * 4) = i32x4{base, base + c[0], base + c[0] + c[1], base + c[0] + c[1] + c[2]};
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) slot0[4 * h + ii] = wl[(ii * 16 + frow) * 4 + fq];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if (msk[i] == 0) continue;
        const int m = m_base + i * 16 + frow;
        int slot = slot0[i];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if ((msk[i] >> (j * 4 + e)) & 1u) {
                    if (slot < p.cap) {
                        p.cand_idx[(size_t)m * p.cap + slot] = n_base + j * 16 + fq * 4 + e;
                        p.cand_score[(size_t)m * p.cap + slot] = acc[j][i][e];
                    }
                    ++slot;
                }
    }
}

template <int BM, int BN, int WM, int WN, bool DENSE>
__global__ __launch_bounds__(WM* WN * 64) void score_kernel(const TopkParams p) {
    using C = Cfg<BM, BN, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // query tiles fastest: concurrently running blocks share a gallery panel in L2
    const int lin = xcd_linear_block(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(lin, p.tiles_m, p.tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = p.g_begin + tn * BN;
    f32x4 acc[C::TN][C::TM];
#pragma unroll
    for (int j = 0; j < C::TN; ++j)
#pragma unroll
        for (int i = 0; i < C::TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int mrow = lane & 15, ncol4 = (lane >> 4) * 4;
    float th[C::TM];
    if (!DENSE) {                                         // requested before the K loop: its L2 latency hides behind the first K-steps
#pragma unroll
        for (int i = 0; i < C::TM; ++i) {
            const int m = m0 + wm * (BM / WM) + i * 16 + mrow;
            th[i] = REID_DBG(p) == 2 ? INFINITY : p.thr[m < p.Nq ? m : 0];
        }
    }
    // no low-rank pair: K2 = 0 (non-null dummies keep the staging code free of constant-null pointers)
    mainloop<BM, BN, WM, WN>(p.Q, p.D, p.G, p.D, p.Q, p.D, p.G, p.D, p.Nq, p.g_end, p.D, 0, m0, n0, smem, acc);
    if (REID_DBG(p) == 1 && acc[0][0][0] != 12345.678f) return;
    if (!DENSE) {
        __syncthreads();                                    // every wave is done reading the operand buffers: reuse them
        filter_epilogue<C::TM, C::TN>(p, acc, th, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane, (int*)(smem + wave * 1024));
        return;
    }
#pragma unroll
    for (int i = 0; i < C::TM; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 16 + mrow;
        const bool mok = m < p.Nq;
        const int eq = p.exq ? p.exq[mok ? m : 0] : -1;
#pragma unroll
        for (int j = 0; j < C::TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 16 + ncol4;
            f32x4 v = acc[j][i];
            if (eq >= 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < p.g_end && p.exg[n + e] == eq) v[e] = -1e9f;
            }
            if (mok) {
                float* d = p.dense + (size_t)m * p.ld_dense + (n - p.g_begin);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < p.g_end) d[e] = v[e];
            }
        }
    }
}

// phase A tail: k-th largest of each row of dense [Nq, n] (n <= 4096) -> thr[q] = kth - 2 eps
__global__ __launch_bounds__(256) void kth_kernel(const float* __restrict__ dense, int ld, int n, int k, float* __restrict__ thr, int Nq) {
    extern __shared__ float sm[];       // [4][n]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    if (q >= Nq) return;
    volatile float* v = sm + (size_t)w * n;
    for (int i = lane; i < n; i += 64) v[i] = dense[(size_t)q * ld + i];
    // k rounds of wave-wide arg-max extraction (k is small: 10..100)
    float kth = -INFINITY;
    const int rounds = k < n ? k : n;
    for (int r = 0; r < rounds; ++r) {
        float best = -INFINITY; int bi = -1;
        for (int i = lane; i < n; i += 64) if (v[i] > best) { best = v[i]; bi = i; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi >= 0 && (bi < 0 || oi < bi))) { best = ob; bi = oi; }
        }
        kth = best;
        if (bi >= 0 && (bi & 63) == lane) v[bi] = -INFINITY;
    }
    if (lane == 0) thr[q] = (rounds < k ? -INFINITY : kth - 2.f * EPS_BF16);
}

// Fast form of the same threshold for small k: every lane streams its share of the row with 16-byte loads and keeps its T
// largest values in registers; the wave then pops the k largest of those 64 T values.  The popped values are k distinct sample
// elements, so their smallest is a LOWER bound of the sample's k-th largest (equal unless more than T of the top k fell on one
// lane): still a valid filter threshold, at most a few % more candidates, and no LDS / no k passes over the row
// (the kernel above took 1.84 ms of the 7.1 ms retrieval at 10k x 8192; this one is HBM-bound).
template <int T>
__global__ __launch_bounds__(256) void kth_fast_kernel(const float* __restrict__ dense, int ld, int n, int k, float* __restrict__ thr, int Nq) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    if (q >= Nq) return;
    float top[T];
#pragma unroll
    for (int t = 0; t < T; ++t) top[t] = -INFINITY;
    auto insert = [&](float v) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float hi = fmaxf(top[t], v);
            v = fminf(top[t], v);
            top[t] = hi;
        }
    };
    const float* row = dense + (size_t)q * ld;
    int done = 0;
    if (((ld | n) & 3) == 0) {
        for (int i = lane * 4; i < n; i += 256) {
            const f32x4 v = *(const f32x4*)(row + i);
            insert(v[0]); insert(v[1]); insert(v[2]); insert(v[3]);
        }
        done = n;
    }
    for (int i = done + lane; i < n; i += 64) insert(row[i]);
    float kth = -INFINITY;
    for (int r = 0; r < k; ++r) {
        const float m = wave_max(top[0]);
        kth = m;
        const unsigned long long b = __ballot(top[0] == m);
        const int win = __builtin_ctzll(b);
        if (lane == win) {
#pragma unroll
            for (int t = 0; t + 1 < T; ++t) top[t] = top[t + 1];
            top[T - 1] = -INFINITY;
        }
    }
    if (lane == 0) thr[q] = (n < k ? -INFINITY : kth - 2.f * EPS_BF16);
}

// The fp32 score of a (query, gallery row) pair is DEFINED by this evaluation order (explicit fma chain: the compiler has no
// contraction freedom), lane l taking elements 4l + 256 j, then the xor butterfly of wave_sum.  Every path that produces a final
// score (select_kernel, the brute-force fallback, the streaming form) uses it, so they agree bit for bit.
__device__ __forceinline__ float dot4_acc(float s, const f32x4 a, const f32x4 b) {
    float t = a[0] * b[0];
    t = __builtin_fmaf(a[1], b[1], t);
    t = __builtin_fmaf(a[2], b[2], t);
    t = __builtin_fmaf(a[3], b[3], t);
    return s + t;
}

// phase C: exact fp32 re-score of the candidates + top-k by (score desc, index asc)
__global__ __launch_bounds__(256) void select_kernel(const float* __restrict__ Qf, const float* __restrict__ Gf, int D,
                                                     const int32_t* __restrict__ exq, const int32_t* __restrict__ exg,
                                                     const int32_t* __restrict__ cand_idx, const float* __restrict__ cand_score,
                                                     const int32_t* __restrict__ cand_cnt, int cap, int k,
                                                     int32_t* __restrict__ out_idx, float* __restrict__ out_score, int Nq) {
    extern __shared__ char sm2[];
    __shared__ float thr2;
    volatile float* sc = (volatile float*)sm2;                  // [cap]
    volatile int32_t* ix = (volatile int32_t*)(sc + cap);       // [cap]
    float* qrow = (float*)((float*)sm2 + 2 * cap);         // [D]
    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int cnt = cand_cnt[q];
    if (cnt > cap) {                          // overflow: caller must take the exact fallback
        if (tid == 0) { out_idx[(size_t)q * k] = -2; out_score[(size_t)q * k] = 0.f; }
        return;
    }
    for (int i = tid; i < D; i += 256) qrow[i] = Qf[(size_t)q * D + i];
    // Second-level filter on the 16-bit-operand scores the filter pass saved: with a = k-th largest of them, a candidate below
    // a - 2 eps cannot be in the exact top k (its true score is < a - eps <= the true score of each of the k candidates at or
    // above a).  Only the survivors (about k + a few) pay the 2 KB fp32 gallery-row gather of the exact re-score; before,
    // all ~250 candidates per query did (5 GB of gathers at 10k x 200k).
    const float* cs = cand_score + (size_t)q * cap;
    for (int c = tid; c < cnt; c += 256) sc[c] = cs[c];
    __syncthreads();
    if (w == 0) {
        float kth = -INFINITY;
        const int rounds = k < cnt ? k : cnt;
        for (int r = 0; r < rounds; ++r) {
            float best = -INFINITY; int bpos = -1;
            for (int c = lane; c < cnt; c += 64) {
                const float v = sc[c];
                if (v > best || bpos < 0) { best = v; bpos = c; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(best, o, 64); const int op = __shfl_xor(bpos, o, 64);
                if (op >= 0 && (bpos < 0 || ob > best)) { best = ob; bpos = op; }
            }
            kth = best;
            if (bpos >= 0 && (bpos & 63) == lane) sc[bpos] = -INFINITY;
        }
        if (lane == 0) thr2 = cnt < k ? -INFINITY : kth - 2.f * EPS_BF16;
    }
    __syncthreads();
    const float t2 = thr2;
    const int eq = exq ? exq[q] : -1;
    for (int c = w; c < cnt; c += 4) {
        const int gi = cand_idx[(size_t)q * cap + c];
        if (cs[c] < t2) {                                      // wave-uniform
            if (lane == 0) { sc[c] = -INFINITY; ix[c] = -1; }
            continue;
        }
        const float* g = Gf + (size_t)gi * D;
        float s = 0.f;
        for (int i = lane * 4; i < D; i += 256) {
            const f32x4 a = *(const f32x4*)(qrow + i), b = *(const f32x4*)(g + i);
            s = dot4_acc(s, a, b);
        }
        s = wave_sum(s);
        if (eq >= 0 && exg[gi] == eq) s = -1e9f;
        if (lane == 0) { sc[c] = s; ix[c] = gi; }
    }
    __syncthreads();
    if (w != 0) return;
    for (int r = 0; r < k; ++r) {
        float best = -INFINITY; int bi = 0x7fffffff, bpos = -1;
        for (int c = lane; c < cnt; c += 64) {
            const float s = sc[c]; const int gi = ix[c];
            if (gi >= 0 && (s > best || (s == best && gi < bi))) { best = s; bi = gi; bpos = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64); const int op = __shfl_xor(bpos, o, 64);
            if (op >= 0 && (bpos < 0 || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; bpos = op; }
        }
        if (lane == 0) {
            out_idx[(size_t)q * k + r] = bpos >= 0 ? bi : -1;
            out_score[(size_t)q * k + r] = bpos >= 0 ? best : -INFINITY;
        }
        if (bpos >= 0 && (bpos & 63) == lane) ix[bpos] = -1;
    }
}

// exact brute force for flagged queries, fp32 throughout: (1) scores of every gallery row, 64 workgroups per query;
// (2) one workgroup per query extracts the k best by (score desc, index asc)
// (slots != nullptr: the flagged queries are the list slots[1 .. slots[0]] (compacted on the device, slots[0] <= n_slots = capacity of the
//  list); entry e uses scratch row e and is taken by workgroup row e % gridDim.y -- ANY number of flagged queries is resolved by the one
//  launch, nothing is read back.  slots == nullptr: blockIdx.y is the query, flagged or not, and the scratch row is the query's.)
__global__ __launch_bounds__(256) void brute_score_kernel(const float* __restrict__ Qf, const float* __restrict__ Gf, int Ng, int D,
                                                          const int32_t* __restrict__ exq, const int32_t* __restrict__ exg, int k,
                                                          const int32_t* __restrict__ out_idx, float* __restrict__ scratch,
                                                          const int32_t* __restrict__ slots, int n_slots) {
    __shared__ float qrow[1024];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n_ent = slots ? min(slots[0], n_slots) : (int)gridDim.y;
    for (int e = blockIdx.y; e < n_ent; e += gridDim.y) {
        const int q = slots ? slots[1 + e] : e;
        if (out_idx[(size_t)q * k] != -2) continue;                  // (workgroup-uniform)
        __syncthreads();                                             // the previous entry's readers of qrow are done
        for (int i = tid; i < D; i += 256) qrow[i] = Qf[(size_t)q * D + i];
        __syncthreads();
        float* sc = scratch + (size_t)e * Ng;
        const int eq = exq ? exq[q] : -1;
        for (int gi = blockIdx.x * 4 + w; gi < Ng; gi += gridDim.x * 4) {
            const float* g = Gf + (size_t)gi * D;
            float s = 0.f;
            for (int i = lane * 4; i < D; i += 256) {
                const f32x4 a = *(const f32x4*)(qrow + i), b = *(const f32x4*)(g + i);
                s = dot4_acc(s, a, b);
            }
            s = wave_sum(s);
            if (eq >= 0 && exg[gi] == eq) s = -1e9f;
            if (lane == 0) sc[gi] = s;
        }
    }
}

__global__ __launch_bounds__(256) void brute_select_kernel(int Ng, int k, int32_t* __restrict__ out_idx, float* __restrict__ out_score,
                                                           float* __restrict__ scratch, const int32_t* __restrict__ slots, int n_slots) {
    __shared__ float rbest[4]; __shared__ int ridx[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n_ent = slots ? min(slots[0], n_slots) : (int)gridDim.x;
    for (int e = blockIdx.x; e < n_ent; e += gridDim.x) {
        const int q = slots ? slots[1 + e] : e;
        if (out_idx[(size_t)q * k] != -2) continue;                  // (workgroup-uniform)
        float* sc = scratch + (size_t)e * Ng;
        for (int r = 0; r < k; ++r) {
            float best = -INFINITY; int bi = 0x7fffffff;
            for (int gi = tid; gi < Ng; gi += 256) {
                const float s = sc[gi];
                if (s > best || (s == best && gi < bi)) { best = s; bi = gi; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (lane == 0) { rbest[w] = best; ridx[w] = bi; }
            __syncthreads();
            if (tid == 0) {
                for (int i = 1; i < 4; ++i)
                    if (rbest[i] > rbest[0] || (rbest[i] == rbest[0] && ridx[i] < ridx[0])) { rbest[0] = rbest[i]; ridx[0] = ridx[i]; }
                const bool ok = ridx[0] != 0x7fffffff;
                out_idx[(size_t)q * k + r] = ok ? ridx[0] : -1;      // (r = 0 overwrites the -2 marker: tested once per entry, above)
                out_score[(size_t)q * k + r] = ok ? rbest[0] : -INFINITY;
                if (ok) sc[ridx[0]] = -INFINITY;
            }
            __syncthreads();
        }
    }
}

constexpr int SAMPLE = 8192;
inline int cap_for(int Ng, int k) {
    // expected survivors ~ k*Ng/SAMPLE (the sample's k-th best is about the (k*Ng/SAMPLE)-th best overall) times ~1.7 for the
    // 2*eps safety margin; 6x head-room
    long c = 6L * k * ((Ng + SAMPLE - 1) / SAMPLE) + 64;
    if (c < 256) c = 256;
    if (c > 8192) c = 8192;
    return (int)c;
}

}  // namespace

extern "C" int64_t reid_topk_ws_bytes(int32_t Nq, int32_t Ng, int32_t k) {
    const int64_t cap = cap_for(Ng, k);
    const int64_t ns = Ng < SAMPLE ? Ng : SAMPLE;
    // thr[Nq] | cnt[Nq] | cand_idx[Nq*cap] | cand_score[Nq*cap] | dense[Nq*ns]
    return (int64_t)Nq * 8 + (int64_t)Nq * cap * 8 + (int64_t)Nq * ns * 4 + 256;
}

namespace {
// fast form of phase C for k <= STREAM_K_MAX (defined with the sorting helpers of the streaming section below)
int launch_select_fast(const float* Qf, const float* Gf, int D, const int32_t* exq, const int32_t* exg, const int32_t* cand_idx,
                       const float* cand_score, const int32_t* cand_cnt, int cap, int k, int32_t* out_idx, float* out_score, int Nq,
                       hipStream_t s, int cnt_stride = 1);
constexpr int SELECT_FAST_K_MAX = 32;

template <int BM, int BN, int WM, int WN>
int launch_filter(TopkParams p, hipStream_t s) {
    using C = Cfg<BM, BN, WM, WN>;
    REID_MAX_LDS((score_kernel<BM, BN, WM, WN, false>), C::LDS_BYTES);
    p.tiles_m = (p.Nq + BM - 1) / BM;
    p.tiles_n = (p.Ng + BN - 1) / BN;
    hipLaunchKernelGGL((score_kernel<BM, BN, WM, WN, false>), dim3(p.tiles_m * p.tiles_n), dim3(C::NT), C::LDS_BYTES, s, p);
    REID_CHECK_LAUNCH("reid_cosine_topk(filter)");
    return REID_OK;
}
}  // namespace

// ------------------------------------------------------------------------------------------ query-resident scan (<= 128 queries)
// The reference ranks its queries one at a time or in small groups (tools/eval_mm_protocol.py:401-455); between the one-pass fp32 form
// (<= 4 queries) and the tiled filter pass (hundreds of queries) the batched pipeline above re-stages the query panel for every gallery
// tile and spends four launches on what is ONE pass over the 16-bit gallery.  Here (phases A and B of reid_cosine_topk in one launch):
//   * one workgroup per compute unit; the gallery's 64-row chunks are dealt round-robin (chunk i * grid + b to workgroup b), so the chip
//     reads one contiguous window at a time; each chunk is two 32-row tiles that pass through a ring of four LDS slots, three in flight
//     (LDS-DMA issued from inline assembly; counted vmcnt waits with a run-time count: the bare stream runs at 6.0 TB/s);
//   * the queries are MFMA operands held in REGISTERS: wave (qg, rh) keeps queries 32 qg .. 32 qg + 31 (128 VGPRs at D = 512) and scores
//     them against rows 16 rh .. + 15 of EVERY tile with v_mfma_f32_16x16x32 -- a lane then owns two queries and four rows of each;
//   * no sample pass: the bar a score has to clear comes from the scan itself.  Workgroup b belongs to group b % k; the running maximum of
//     every query is folded lane -> workgroup (LDS atomic max) -> gmax[query][group] (one agent-scope atomic max of an order-preserving key
//     per query and workgroup, at steps 0, 1, 3, 7, ...).  The k group maxima of a query are scores of k DISTINCT gallery rows, so their
//     minimum B is a lower bound of the k-th best 16-bit score, and a row of the final top-k has a 16-bit score >= B - 2 eps (the margin
//     of the filter pass: EPS_BF16 bounds |q~.g~ - q.g|).  The k values are re-read through the same DMA stream (sc1: past the L2s) two
//     steps after they were requested; any mixture of old and new values is valid, they only grow.  These lines are the SAME for every
//     workgroup of the chip: read or updated every step by every wave they, not the gallery, set the pace;
//   * steps a wave scored before all k groups had reported are scored AGAIN at the end (candidates only; their rows come from L2), and a
//     workgroup that is through before the bar exists polls a bounded number of times: no workgroup depends on another one to finish;
//   * survivors wait in a 4-entry queue per lane in LDS; at the end the lanes reserve places per workgroup and query in LDS and ONE lane
//     per query adds the workgroup's total to the query's counter (one counter per 128-byte line).  cand_idx / cand_score / cand_cnt are
//     those of the filter pass (cand_cnt strided): phase C is unchanged.
namespace {
namespace scan {
constexpr int KG = 16;            // bound groups held per query (k <= KG)
constexpr int QCAP = 4;           // queued survivors per lane between flushes
constexpr int NQ_MAX = 128;
constexpr int CNT_STRIDE = 32;     // one candidate counter per 128-byte line (every workgroup of the chip adds to them at the same time)

struct ScanParams {
    const bf16_t* Q; const bf16_t* G;
    int Nq, Ng;
    const int32_t* exq; const int32_t* exg;
    int k, cap;
    uint32_t* gmax;               // [4 query groups][KG][32] keys of the group maxima; zero (= nothing yet) when the kernel starts
    int32_t* cand_idx; float* cand_score; int32_t* cand_cnt;
    unsigned long long* trace;    // -DREID_SCAN_TRACE builds: 8 stamps per workgroup (tools/exp_scan_trace.py)
    int dbg;                      // -DREID_SCAN_TRACE builds: ablations (wrong results): 1 = no scoring, 2 = no publish / bar requests, 4 = no enqueue
};
#ifdef REID_SCAN_TRACE
#define SCAN_TRACE(slot, v) do { if (p.trace && threadIdx.x == 0) p.trace[(size_t)blockIdx.x * 8 + (slot)] = (v); } while (0)
#else
#define SCAN_TRACE(slot, v) do { } while (0)
#endif
#define SCAN_STAMP(slot) SCAN_TRACE(slot, __builtin_amdgcn_s_memrealtime())
#ifdef REID_SCAN_TRACE
#define SCAN_DBG(bit) ((p.dbg & (bit)) != 0)
#else
#define SCAN_DBG(bit) false
#endif

// order-preserving unsigned key of a float; 0 is below every float
__device__ __forceinline__ uint32_t key_of(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float key_value(uint32_t key) {
    return key == 0 ? -INFINITY : __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
}
typedef __attribute__((address_space(3))) char* lds_cptr;
typedef __attribute__((address_space(3))) uint32_t* lds_u32ptr;
__device__ __forceinline__ uint32_t lds_addr(const void* ptr) { return (uint32_t)(uintptr_t)(lds_cptr)(char*)ptr; }
// LDS-DMA from inline assembly (see lora.hip dma16: hipcc then keeps no scoreboard entry for the ring and places no waits of its own)
__device__ __forceinline__ void dma16(const void* src, uint32_t lds_base) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory");
}
__device__ __forceinline__ void dma16_coherent(const void* src, uint32_t lds_base) {      // sc1: past the non-coherent caches (atomics' home)
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1" ::"v"(src), "s"(m0v) : "memory");
}
__device__ __forceinline__ void dma4(const void* src, uint32_t lds_base) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(m0v) : "memory");
}

// `s_waitcnt vmcnt(n)` with a run-time, wave-uniform n (the counter completes in issue order: n = the operations allowed to stay in flight)
__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
#define REID_VM_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        REID_VM_CASE(1) REID_VM_CASE(2) REID_VM_CASE(3) REID_VM_CASE(4) REID_VM_CASE(5) REID_VM_CASE(6) REID_VM_CASE(7) REID_VM_CASE(8)
        REID_VM_CASE(9) REID_VM_CASE(10) REID_VM_CASE(11) REID_VM_CASE(12) REID_VM_CASE(13) REID_VM_CASE(14) REID_VM_CASE(15) REID_VM_CASE(16)
        REID_VM_CASE(17) REID_VM_CASE(18) REID_VM_CASE(19) REID_VM_CASE(20)
#undef REID_VM_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <int KS /* D / 32 */, bool EXCL>
__global__ __launch_bounds__(512) void scan_filter_kernel(const ScanParams p) {
    constexpr int D = KS * 32, ROWB = D * 2, TILE_B = 32 * ROWB, CPR = ROWB / 16, RPI = 1024 / ROWB;
    constexpr int WAVE_INSTR = (32 / RPI) / 8;            // DMA instructions per wave and tile (32 rows)
    constexpr int NO_BAR = 0x7fffffff;
    static_assert(RPI == 1 || RPI == 2, "D = 512 or 256");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* tiles = smem;                                   // ring of 4 tiles of 32 rows; chunk c of row r at position c ^ (r & 15)
    float* qsc = (float*)(smem + 4 * TILE_B);             // [512][QCAP] queued scores
    int32_t* qix = (int32_t*)(qsc + 512 * QCAP);          // [512][QCAP] queued gallery rows (bit 31: the lane's second query)
    uint32_t* bars = (uint32_t*)(qix + 512 * QCAP);       // [4][KG][32] this workgroup's copy of gmax
    int32_t* exs = (int32_t*)(bars + 4 * KG * 32);        // [4][64] image ids of the rows of each ring slot
    int* fv = (int*)(exs + 256);                          // [16] per wave: steps scored without a bar | bar complete
    uint32_t* wgmax = (uint32_t*)(fv + 16);               // [128] keys of the workgroup's running maxima, one per query
    int* wgcnt = (int*)(wgmax + NQ_MAX);                  // [128] survivors of the workgroup per query (final flush)
    const uint32_t smem_lds = lds_addr(smem);             // (one cast of the array itself; LDS addresses below are offsets from it)
    auto lds_of = [&](const void* ptr) { return smem_lds + (uint32_t)((const char*)ptr - smem); };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wave (qg, rh): queries 32 qg .. + 31 against rows 16 rh .. + 15 of EVERY tile (all eight waves work on a tile at once).  With
    // v_mfma_f32_16x16x32 (A = 16 gallery rows, B = 16 queries) a lane owns queries qa and 16 + qa of the group and rows 4 rb .. + 3.
    const int qg = wave & 3, rh = wave >> 2;
    const int qa = lane & 15, rb = lane >> 4;
    const bool active = qg < ((p.Nq + 31) >> 5);
    int qv[2];
    bool qk[2];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) { qv[jb] = qg * 32 + jb * 16 + qa; qk[jb] = active && qv[jb] < p.Nq; }
    // step i of workgroup b covers the 64-row chunk i * gridDim.x + b (two tiles): at any moment the chip reads one contiguous window
    const int n_chunks = (p.Ng + 63) >> 6;
    if ((int)blockIdx.x >= n_chunks) return;
    const int nstep = (n_chunks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int ntile = 2 * nstep;
    const int row_end = p.Ng;
    auto tile_row0 = [&](int t) { return ((t >> 1) * (int)gridDim.x + (int)blockIdx.x) * 64 + (t & 1) * 32; };
    SCAN_STAMP(0);

    auto stage = [&](int t) -> int {                      // the 32 rows of tile t -> ring slot t & 3; returns the operations issued
        const uint32_t base = lds_of(tiles + (t & 3) * TILE_B);
#pragma unroll
        for (int u = 0; u < WAVE_INSTR; ++u) {
            const int j = wave * WAVE_INSTR + u;
            const int r = j * RPI + lane / CPR, pos = lane % CPR;
            int gr = tile_row0(t) + r;
            gr = gr < row_end ? gr : row_end - 1;
            dma16(p.G + (size_t)gr * D + ((pos ^ (r & 15)) << 3), base + j * 1024);
        }
        if (EXCL && wave == 7) {
            int gr = tile_row0(t) + (lane & 31);
            gr = gr < row_end ? gr : row_end - 1;
            dma4(p.exg + gr, lds_of(exs + (t & 3) * 64));
            return WAVE_INSTR + 1;
        }
        return WAVE_INSTR;
    };
    // three tiles in flight before anything else (the query loads below overlap them)
    int a2 = 0, a1 = 0;
    stage(0);
    if (ntile > 1) a2 = stage(1);
    if (ntile > 2) a1 = stage(2);
    for (int i = tid; i < 4 * KG * 32; i += 512) bars[i] = 0;
    if (tid < NQ_MAX) { wgmax[tid] = 0; wgcnt[tid] = 0; }

    bf16x8 bq[2][KS];                                     // the wave's 32 queries as B operands
    int eq[2] = {-1, -1};
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
        const bf16_t* qp = p.Q + (size_t)(qk[jb] ? qv[jb] : 0) * D + rb * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bq[jb][ks] = *(const bf16x8*)(qp + ks * 32);
            if (!qk[jb]) bq[jb][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        if (EXCL && qk[jb]) eq[jb] = p.exq[qv[jb]];
    }
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(bq[jb][ks]));   // the loads above are waited for HERE, not inside the loop
        asm volatile("" : "+v"(eq[jb]));
    }
    uint32_t* my_gmax = p.gmax + ((size_t)qg * KG + (blockIdx.x % p.k)) * 32 + (lane & 31);   // lanes 0..31 of the rh = 0 wave: one query each
    const float margin = 2.f * EPS_BF16;

    auto request_bars = [&]() {                           // this query group's KG x 32 keys (2 KiB), past the non-coherent caches
#pragma unroll
        for (int u = 0; u < KG * 32 * 4 / 1024; ++u)
            dma16_coherent((const char*)(p.gmax + (size_t)qg * KG * 32) + u * 1024 + lane * 16, lds_of(bars + qg * KG * 32) + u * 1024);
    };
    constexpr int BAR_INSTR = KG * 32 * 4 / 1024;

    // chunk 4 ks + rb of row 16 rh + qa sits at position (4 ks + rb) ^ qa: four per-lane offsets (ks & 3) + an immediate (ks >> 2)
    int aoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) aoff[j] = (rh * 16 + qa) * ROWB + (((4 * j + rb) ^ qa) << 4);
    float mx[2] = {-INFINITY, -INFINITY}, mx_loc[2] = {-INFINITY, -INFINITY}, bar[2] = {INFINITY, INFINITY};
    uint32_t pub_key = 0;
    bool valid = false;                                   // (wave-uniform) every group of every query of this wave has reported
    int first_valid = NO_BAR;
    int qn = 0;

    auto flush = [&]() {                                  // queued survivors -> the queries' candidate lists
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            int n = 0;
            for (int j = 0; j < qn && j < QCAP; ++j) n += ((qix[tid * QCAP + j] >> 31) & 1) == jb ? 1 : 0;
            if (n > 0) {
                int slot = atomicAdd(p.cand_cnt + qv[jb] * CNT_STRIDE, n);
                for (int j = 0; j < qn && j < QCAP; ++j) {
                    const int e = qix[tid * QCAP + j];
                    if (((e >> 31) & 1) != jb) continue;
                    if (slot < p.cap) {
                        p.cand_idx[(size_t)qv[jb] * p.cap + slot] = e & 0x7fffffff;
                        p.cand_score[(size_t)qv[jb] * p.cap + slot] = qsc[tid * QCAP + j];
                    }
                    ++slot;
                }
            }
        }
        qn = 0;
    };
    auto read_bars = [&]() {
        bool ok = true;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            float b = INFINITY;
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                const float v = key_value(bars[(qg * KG + g) * 32 + jb * 16 + qa]);
                b = g < p.k ? fminf(b, v) : b;
            }
            ok = ok && (!qk[jb] || b > -INFINITY);
            bar[jb] = qk[jb] ? b - margin : INFINITY;
        }
        valid = __ballot(ok) == ~0ull;
    };
    // the wave's 16 rows of tile t against its 32 queries; track: fold into the running maxima; enqueue: compare with the bars
    auto score_tile = [&](int t, bool track, bool enqueue) {
        const char* tile = tiles + (t & 3) * TILE_B;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        bf16x8 a[2][4];                                   // fragments are read four k-steps ahead of their use
#pragma unroll
        for (int u = 0; u < 4; ++u) a[0][u] = *(const bf16x8*)(tile + aoff[u]);
#pragma unroll
        for (int g = 0; g < KS / 4; ++g) {
            if (g + 1 < KS / 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) a[(g + 1) & 1][u] = *(const bf16x8*)(tile + aoff[u] + (g + 1) * 256);
            }
            __builtin_amdgcn_sched_barrier(0);            // (hipcc otherwise sinks every read to just before its MFMA: one LDS latency per MFMA)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0] = mfma16(a[g & 1][u], bq[0][4 * g + u], acc[0]);
                acc[1] = mfma16(a[g & 1][u], bq[1][4 * g + u], acc[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const int trow0 = tile_row0(t) + rh * 16 + rb * 4;  // element e of a lane: row trow0 + e
        if (trow0 + 4 > row_end) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (trow0 + e >= row_end) { acc[0][e] = -INFINITY; acc[1][e] = -INFINITY; }
        }
        if (EXCL) {
            typedef __attribute__((ext_vector_type(4))) int i32x4;
            const i32x4 ids = *(const i32x4*)(exs + (t & 3) * 64 + rh * 16 + rb * 4);
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (eq[jb] >= 0 && ids[e] == eq[jb]) acc[jb][e] = -INFINITY;
        }
        float m[2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            m[jb] = fmaxf(fmaxf(acc[jb][0], acc[jb][1]), fmaxf(acc[jb][2], acc[jb][3]));
            if (track) mx[jb] = fmaxf(mx[jb], m[jb]);
        }
        if (enqueue && __ballot(m[0] >= bar[0] || m[1] >= bar[1]) != 0) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool hit = acc[jb][e] >= bar[jb];
                    if (__ballot(hit && qn >= QCAP) != 0) flush();
                    if (hit) {
                        qsc[tid * QCAP + qn] = acc[jb][e];
                        qix[tid * QCAP + qn] = (int32_t)((uint32_t)(trow0 + e) | ((uint32_t)jb << 31));
                        ++qn;
                    }
                }
        }
    };
    // running maxima: lane -> workgroup (LDS atomic max, only when a lane's maximum moved) -> chip (one lane per query and workgroup)
    auto fold_local = [&]() {
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const bool up = qk[jb] && mx[jb] > mx_loc[jb];
            if (__ballot(up) == 0) continue;
            if (up) __hip_atomic_fetch_max((lds_u32ptr)wgmax + qv[jb], key_of(mx[jb]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            mx_loc[jb] = mx[jb];
        }
    };
    auto publish = [&]() -> int {                         // (rh = 0 waves) returns the operations issued (wave-uniform)
        const uint32_t key = wgmax[qg * 32 + (lane & 31)];
        const bool up = lane < 32 && qg * 32 + lane < p.Nq && key > pub_key;
        if (__ballot(up) == 0) return 0;
        if (up) asm volatile("global_atomic_umax %0, %1, off sc1" ::"v"(my_gmax), "v"(key) : "memory");   // sc1: agent scope (the XCDs' L2s are not coherent)
        pub_key = key > pub_key ? key : pub_key;
        return 1;
    };
    // The group maxima are published after steps 0, 1 and every fourth step, and re-read after steps 0, 1, 3, 7 and then every eighth
    // step, the reads staggered over the workgroups (every step while the bar is incomplete).  The 2 KiB a query group reads are the SAME
    // lines for every workgroup of the chip, served past the L2s, and so are the atomics' targets: with every wave publishing its own
    // maxima and re-reading every step this traffic, not the gallery, set the pace (120 us at 128 queries against 44 us without it;
    // 34 us = the bare stream).  A request is issued at the end of step i and consumed at the start of step i + 2: by then it has landed
    // with the tile that was staged just before it, and nothing waits for the atomics behind it.
    auto refresh_after = [&](int i) { return i == 0 || i == 1 || i == 3 || i == 7 || (i > 7 && ((i + (int)blockIdx.x) & 7) == 7); };
    auto publish_after = [&](int i) { return i == 0 || i == 1 || (i & 3) == 3 || i + 1 == nstep; };

    __syncthreads();                                      // (bars, wgmax zeroed)
    SCAN_STAMP(1);
    // Operations in flight per wave, in issue order: A(t-3) = tile t .. R(t-3), P(t-3), A(t-2), .., A(t-1), R(t-1), P(t-1) with A = the stage
    // at the top of an iteration, R = the bar request and P = the atomic of its tail.  Tile t has landed when at most R(t-3) + P(t-3) +
    // A(t-2) + ... + P(t-1) operations are left, the request R(t-3) when at most P(t-3) + A(t-2) + ... are.
    int r3 = 0, p3 = 0, r2 = 0, p2 = 0, r1 = 0, p1 = 0;
    for (int t = 0; t < ntile; ++t) {
        const int i = t >> 1;
        // the bars requested after tile t - 3 (the last tile of step i - 2, by the rh = 0 waves) are read at tile t = 2 i by both halves
        const bool read_due = active && (t & 1) == 0 && i > 1 && (first_valid == NO_BAR || refresh_after(i - 2));
        wait_vm((read_due ? 0 : r3) + p3 + a2 + r2 + p2 + a1 + r1 + p1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t == 2) SCAN_STAMP(7);
        const int a0 = t + 3 < ntile ? stage(t + 3) : 0;
        int r0 = 0, p0 = 0;
        if (active) {
            if (read_due) {
                read_bars();
                if (valid && first_valid == NO_BAR) first_valid = i;
            }
            if (!SCAN_DBG(1)) {
                score_tile(t, true, valid && !SCAN_DBG(4));
                fold_local();
            }
            if ((t & 1) == 1 && rh == 0 && !SCAN_DBG(2)) {       // end of step i
                if (first_valid == NO_BAR || refresh_after(i)) { request_bars(); r0 = BAR_INSTR; }
                if (first_valid == NO_BAR || publish_after(i)) p0 = publish();
            }
        }
        r3 = r2; p3 = p2; a2 = a1; r2 = r1; p2 = p1; a1 = a0; r1 = r0; p1 = p0;
    }
    SCAN_STAMP(2);
    int polls = 0;
    // Tail.  A workgroup that is through before every group has reported (short scans: all workgroups finish together) polls a bounded
    // number of times -- it never depends on another workgroup to terminate.
    for (int poll = 0;; ++poll) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (active) read_bars();
        if (lane == 0) fv[8 + wave] = (!active || valid) ? 1 : 0;
        __syncthreads();
        bool all = true;
#pragma unroll
        for (int w = 0; w < 8; ++w) all = all && fv[8 + w] != 0;
        polls = poll;
        if (all || poll >= 64) break;
        if (active && rh == 0) request_bars();
        __builtin_amdgcn_s_sleep(16);
    }
    // the steps scored without a bar, again (candidates only; their rows are in L2)
    if (lane == 0) fv[wave] = (active && valid) ? (first_valid < nstep ? first_valid : nstep) : 0;
    if (active && !valid) {
        // some group of some query has still not reported (fewer scoring workgroups than k, a group whose rows are all excluded, or a chip
        // shared with another kernel): no bar, this wave's rows were never compared, so its queries take the exact fallback (count past
        // the capacity = the overflow mark phase C understands)
        if (rb == 0) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
                if (qk[jb]) atomicAdd(p.cand_cnt + qv[jb] * CNT_STRIDE, p.cap + 1);
        }
    }
    __syncthreads();
    int nrev = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) nrev = fv[w] > nrev ? fv[w] : nrev;
    SCAN_STAMP(3);
    SCAN_TRACE(6, (unsigned long long)polls | ((unsigned long long)nrev << 16) | ((unsigned long long)(first_valid & 0xffff) << 32));
    (void)polls;
    for (int t0 = 0; t0 < 2 * nrev; t0 += 4) {            // four tiles per round trip
        if (t0 > 0) __syncthreads();
        for (int u = 0; u < 4 && t0 + u < 2 * nrev; ++u) stage(t0 + u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int u = 0; u < 4 && t0 + u < 2 * nrev; ++u) {
            const int t = t0 + u;
            if (active && valid && (t >> 1) < first_valid) score_tile(t, false, true);
        }
    }
    SCAN_STAMP(4);
    // Final flush: the lanes of the workgroup first reserve their places in LDS, then ONE lane per query adds the workgroup's total to the
    // query's counter (each lane adding for itself: ~50 k lane-atomics of the whole chip on the same few lines at the same moment, 12 us
    // median / 26 us worst per workgroup), and the entries are written behind the returned base.
    int off[2] = {0, 0};
    if (active) {
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            int n = 0;
            for (int j = 0; j < qn && j < QCAP; ++j) n += ((qix[tid * QCAP + j] >> 31) & 1) == jb ? 1 : 0;
            if (n > 0) off[jb] = __hip_atomic_fetch_add((lds_u32ptr)wgcnt + qv[jb], (uint32_t)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    int* wgbase = (int*)bars;                             // (the bars are no longer needed)
    if (tid < NQ_MAX) wgbase[tid] = wgcnt[tid] > 0 ? atomicAdd(p.cand_cnt + tid * CNT_STRIDE, wgcnt[tid]) : 0;
    __syncthreads();
    if (active) {
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            int slot = wgbase[qv[jb] & (NQ_MAX - 1)] + off[jb];
            for (int j = 0; j < qn && j < QCAP; ++j) {
                const int e = qix[tid * QCAP + j];
                if (((e >> 31) & 1) != jb) continue;
                if (slot < p.cap) {
                    p.cand_idx[(size_t)qv[jb] * p.cap + slot] = e & 0x7fffffff;
                    p.cand_score[(size_t)qv[jb] * p.cap + slot] = qsc[tid * QCAP + j];
                }
                ++slot;
            }
        }
    }
    SCAN_STAMP(5);
}
#ifdef REID_SCAN_TRACE
unsigned long long* g_scan_trace = nullptr;
#endif
}  // namespace scan
}  // namespace
#ifdef REID_SCAN_TRACE
extern "C" void reid_debug_scan_trace(void* buf) { scan::g_scan_trace = (unsigned long long*)buf; }
#endif

/* 1 when reid_cosine_topk takes the query-resident scan for this shape (the caller may then prefer it to the one-pass fp32 form for 2-4
 * queries: 83-88 us against 91-105 us per call at 200k x 512, r04). */
extern "C" int32_t reid_topk_scan_ok(int32_t Nq, int32_t Ng, int32_t D, int32_t k) {
    if (!(Nq >= 1 && Nq <= scan::NQ_MAX && k >= 1 && k <= scan::KG && k <= SELECT_FAST_K_MAX && k <= Ng && (D == 512 || D == 256))) return 0;
    if (reid_knob(KNOB_TOPK_SCAN) == 0) return 0;
    int cus = reid_num_cus() & ~7;
    if (cus < 8) cus = 8;
    const int n_chunks = (Ng + 63) / 64;
    const int grid = n_chunks < cus ? n_chunks : cus;
    const int64_t ns = Ng < SAMPLE ? Ng : SAMPLE;
    const int64_t scan_ws = (int64_t)(scan::NQ_MAX * scan::CNT_STRIDE + 4 * scan::KG * 32) * (int64_t)sizeof(int32_t);   // inside the sample area of ws
    return grid >= 4 * k && (int64_t)Nq * ns * (int64_t)sizeof(float) >= scan_ws;
}

extern "C" int reid_cosine_topk(const void* Q_bf16, const void* G_bf16, const float* Qf, const float* Gf, int32_t Nq, int32_t Ng,
                                int32_t D, int32_t k, const int32_t* exclude_q, const int32_t* exclude_g, void* ws, int32_t* out_idx,
                                float* out_score, void* stream) {
    REID_CHECK_ARG(Q_bf16 && G_bf16 && Qf && Gf && ws && out_idx && out_score, "reid_cosine_topk: null pointer");
    REID_CHECK_ARG(Nq > 0 && Ng > 0 && k > 0 && k <= Ng && k <= 1024, "reid_cosine_topk: Nq=%d Ng=%d k=%d", Nq, Ng, k);
    REID_CHECK_ARG(D % 64 == 0 && D <= 1024, "reid_cosine_topk: D=%d must be a multiple of 64, <= 1024", D);
    REID_CHECK_ARG((exclude_q == nullptr) == (exclude_g == nullptr), "reid_cosine_topk: exclude_q and exclude_g go together");
    hipStream_t s = (hipStream_t)stream;
    const int cap = cap_for(Ng, k);
    const int ns = Ng < SAMPLE ? Ng : SAMPLE;
    float* thr = (float*)ws;
    int32_t* cnt = (int32_t*)(thr + Nq);
    int32_t* cidx = cnt + Nq;
    float* cscore = (float*)(cidx + (size_t)Nq * cap);
    float* dense = cscore + (size_t)Nq * cap;
    constexpr int BM = 128, BN = 128;
    using C = Cfg<BM, BN, 2, 2>;
    REID_MAX_LDS((score_kernel<BM, BN, 2, 2, true>), C::LDS_BYTES);
    REID_MAX_LDS((score_kernel<BM, BN, 2, 2, false>), C::LDS_BYTES);
    // <= 128 queries: phases A and B as ONE pass of the query-resident scan kernel (scan::scan_filter_kernel); REID_TOPK_SCAN=0: the tiled form
    if (reid_topk_scan_ok(Nq, Ng, D, k)) {
        int cus = reid_num_cus() & ~7;
        if (cus < 8) cus = 8;
        const int n_chunks = (Ng + 63) / 64;
        const int grid = n_chunks < cus ? n_chunks : cus;       // one workgroup per compute unit, 64-row chunks dealt round-robin
        {
            // cnt2[128 x CNT_STRIDE] | gmax[4][KG][32] live in the (unused) sample area: one fill
            int32_t* cnt2 = (int32_t*)dense;
            uint32_t* gmax = (uint32_t*)(cnt2 + scan::NQ_MAX * scan::CNT_STRIDE);
            REID_CHECK_HIP(hipMemsetAsync(cnt2, 0, (scan::NQ_MAX * scan::CNT_STRIDE + 4 * scan::KG * 32) * sizeof(int32_t), s), "hipMemsetAsync");
            scan::ScanParams sp{};
            sp.Q = (const bf16_t*)Q_bf16; sp.G = (const bf16_t*)G_bf16; sp.Nq = Nq; sp.Ng = Ng; sp.exq = exclude_q; sp.exg = exclude_g;
            sp.k = k; sp.cap = cap; sp.gmax = gmax; sp.cand_idx = cidx; sp.cand_score = cscore; sp.cand_cnt = cnt2;
#ifdef REID_SCAN_TRACE
            sp.trace = scan::g_scan_trace;
            sp.dbg = getenv("REID_SCAN_DBG") ? atoi(getenv("REID_SCAN_DBG")) : 0;
#endif
#define REID_SCAN_LAUNCH(KS, EX) do {                                                                                           \
            constexpr size_t lds = 4 * 32 * (KS * 64) + 512 * scan::QCAP * 8 + 4 * scan::KG * 32 * 4 + 1024 + 64 + 1024;                \
            REID_MAX_LDS((scan::scan_filter_kernel<KS, EX>), lds);                                                              \
            hipLaunchKernelGGL((scan::scan_filter_kernel<KS, EX>), dim3(grid), dim3(512), lds, s, sp); } while (0)
            if (D == 512) { if (exclude_q) REID_SCAN_LAUNCH(16, true); else REID_SCAN_LAUNCH(16, false); }
            else { if (exclude_q) REID_SCAN_LAUNCH(8, true); else REID_SCAN_LAUNCH(8, false); }
#undef REID_SCAN_LAUNCH
            REID_CHECK_LAUNCH("reid_cosine_topk(scan)");
            return launch_select_fast(Qf, Gf, D, exclude_q, exclude_g, cidx, cscore, cnt2, cap, k, out_idx, out_score, Nq, s, scan::CNT_STRIDE);
        }
    }
    TopkParams p{};
    p.Q = (const bf16_t*)Q_bf16; p.G = (const bf16_t*)G_bf16; p.Nq = Nq; p.Ng = Ng; p.D = D;
    p.exq = exclude_q; p.exg = exclude_g; p.cap = cap;
    p.dbg = reid_knob(KNOB_TOPK_DBG) > 0 ? reid_knob(KNOB_TOPK_DBG) : 0;
    p.tiles_m = (Nq + BM - 1) / BM;
    // phase A: sample = first ns gallery rows, dense scores, k-th best -> thr
    p.g_begin = 0; p.g_end = ns; p.thr = nullptr; p.dense = dense; p.ld_dense = ns;
    p.tiles_n = (ns + BN - 1) / BN;
    hipLaunchKernelGGL((score_kernel<BM, BN, 2, 2, true>), dim3(p.tiles_m * p.tiles_n), dim3(C::NT), C::LDS_BYTES, s, p);
    REID_CHECK_LAUNCH("reid_cosine_topk(sample)");
    REID_MAX_LDS((kth_kernel), 4 * SAMPLE * 4);   // 128 KiB: 4 waves x 8192 floats
    if (k <= 32) hipLaunchKernelGGL(kth_fast_kernel<2>, dim3((Nq + 3) / 4), dim3(256), 0, s, dense, ns, ns, k, thr, Nq);
    else if (k <= 128) hipLaunchKernelGGL(kth_fast_kernel<4>, dim3((Nq + 3) / 4), dim3(256), 0, s, dense, ns, ns, k, thr, Nq);
    else hipLaunchKernelGGL(kth_kernel, dim3((Nq + 3) / 4), dim3(256), 4 * ns * sizeof(float), s, dense, ns, ns, k, thr, Nq);
    REID_CHECK_LAUNCH("reid_cosine_topk(kth)");
    // phase B: filter the whole gallery
    REID_CHECK_HIP(hipMemsetAsync(cnt, 0, (size_t)Nq * sizeof(int32_t), s), "hipMemsetAsync");
    p.g_begin = 0; p.g_end = Ng; p.thr = thr; p.dense = nullptr;
    p.cand_idx = cidx; p.cand_score = cscore; p.cand_cnt = cnt;
    {
        // Filter-pass anatomy at 10k x 200k x 512 (REID_TOPK_DBG=1/2, r01): K loops 1.1-1.7 ms depending on the tile, compare of
        // every score against its row threshold 0.7 ms, candidate appends 0.6 ms; 128x128 / 128x256 / 256x128 tiles all end
        // at 4.0-4.1 ms per top-10 call.  Tried and dropped: a persistent variant with a 3-stage ring across tiles (4.02 ms: the
        // per-tile latency it removes is not the bottleneck), the 256x256 tile (fastest K loop, but its epilogue spills).
        // r02: the MER GEMM's wave-row ping-pong K loop (gemm_core.h mainloop_pp) under this epilogue, 256x256: 4.68 ms against 4.10 ms
        // for the default on the same box (profiles/r02_retrieval_tiles_pingpong.log, patch next to it): the loop leaves the compare
        // epilogue no registers (176-228 B/lane of scratch in it), and the epilogue, not the K loop, is what this pass waits on.
        const int tile = reid_knob(KNOB_TOPK_TILE) >= 0 ? reid_knob(KNOB_TOPK_TILE) : 2;
        int rc;
        if (tile == 1) rc = launch_filter<256, 128, 4, 2>(p, s);
        else if (tile == 3) rc = launch_filter<256, 256, 2, 4>(p, s);
        else if (tile == 4) rc = launch_filter<128, 256, 2, 4>(p, s);
        else if (tile == 5) rc = launch_filter<64, 256, 1, 4>(p, s);
        else if (tile == 6) rc = launch_filter<64, 128, 1, 4>(p, s);
        else if (tile == 2 && Nq <= 64 && Ng >= 1024) rc = launch_filter<64, 256, 1, 4>(p, s);   // few queries: half the query panel staged per gallery row (r03: 123 -> 105 us at 5-32 queries x 200k)
        else if (tile == 0 || Nq < 256 || Ng < 1024) rc = launch_filter<128, 128, 2, 2>(p, s);
        else rc = launch_filter<128, 256, 2, 4>(p, s);
        if (rc) return rc;
    }
    // phase C
    if (k <= SELECT_FAST_K_MAX && reid_knob(KNOB_TOPK_TILE) != 9)
        return launch_select_fast(Qf, Gf, D, exclude_q, exclude_g, cidx, cscore, cnt, cap, k, out_idx, out_score, Nq, s);
    const size_t lds = (size_t)cap * 8 + (size_t)D * 4;
    REID_MAX_LDS((select_kernel), 8192 * 8 + 1024 * 4);
    hipLaunchKernelGGL(select_kernel, dim3(Nq), dim3(256), lds, s, Qf, Gf, D, exclude_q, exclude_g, cidx, cscore, cnt, cap, k, out_idx, out_score, Nq);
    REID_CHECK_LAUNCH("reid_cosine_topk(select)");
    return REID_OK;
}

// ------------------------------------------------------------------------------------------ streaming form (a few queries)
// The reference ranks ONE query at a time (tools/eval_mm_protocol.py:401-455: sim = q @ G.T, argsort).  For a handful of
// queries the batched pipeline above is all launch latency (sample, threshold, filter, select: ~190 us at Nq = 1) although
// the problem is one pass over the gallery.  Here: ONE kernel streams the fp32 gallery once (Ng*D*4 bytes, HBM-bound), scores
// up to SQ queries per row with the very arithmetic of select_kernel's re-score (so both paths produce the same fp32
// scores, bit for bit), and keeps a k-entry list per wave and query in LDS (replace-the-worst; after the first few rows
// almost no row qualifies).  A second, small kernel merges the per-workgroup lists.  Order: score descending, gallery index
// ascending on ties, as everywhere.
namespace {
constexpr int SQ = 4;               // queries per pass over the gallery
constexpr int STREAM_K_MAX = 32;    // a sort window of 64 lanes holds the best k plus at least as many new entries
constexpr int STREAM_LIST_BUDGET = 16384;   // list entries per query over all workgroups (bounds the merge pass and the workspace)

__device__ __forceinline__ bool ranks_before(float sa, int ia, float sb, int ib) { return sa > sb || (sa == sb && ia < ib); }

inline int stream_groups(int k) {
    int g = STREAM_LIST_BUDGET / k;
    return g > 1024 ? 1024 : g;
}

// A wave's candidates of one query live in REGISTERS, one entry per lane: appending is two v_cndmask (no LDS, no shuffles).
// When all 64 lanes are taken, the entries are ranked against each other (64 readlane broadcasts), moved to the lane of their
// rank with one ds_permute -- i.e. sorted -- and everything behind rank k is dropped; the k-th entry becomes the bar a row has
// to clear from then on.  Rows that clear the bar get rarer as the scan proceeds (~k ln(rows/k) in total).
struct LaneList { float s; int i; };
constexpr int INVALID_IDX0 = 0x7fffffc0;     // 64 distinct "after everything" keys for unused lanes

// sort the wave's entries best-first across the lanes; entries of lanes >= cnt or with a negative index are void and end up
// last.  Returns the number of real entries.
__device__ __forceinline__ int lanelist_sort(LaneList& e, int cnt, int lane) {
    const bool real = lane < cnt && e.i >= 0;
    const float ms = real ? e.s : -INFINITY;
    const int mi = real ? e.i : INVALID_IDX0 + lane;
    int rank = 0;
#pragma unroll
    for (int m = 0; m < 64; ++m) {
        const float os = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ms), m));
        const int oi = __builtin_amdgcn_readlane(mi, m);
        rank += ranks_before(os, oi, ms, mi) ? 1 : 0;
    }
    e.s = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(rank * 4, __builtin_bit_cast(int, ms)));
    e.i = __builtin_amdgcn_ds_permute(rank * 4, mi);
    return __builtin_popcountll(__ballot(real));
}

// the k best of n entries in LDS (void entries: index < 0), sorted into lanes [0, k) of the calling wave; k <= STREAM_K_MAX <= 32:
// a window of 64 lanes = the best k so far + up to 64 - k new entries per sort
__device__ __forceinline__ int wave_select_lds(const float* sc, const int32_t* ix, int n, int k, int lane, LaneList& e) {
    int have = n < 64 ? n : 64;
    e = lane < have ? LaneList{sc[lane], ix[lane]} : LaneList{-INFINITY, -1};
    int next = have;
    int real = lanelist_sort(e, have, lane);
    while (next < n) {
        const int keep = real < k ? real : k;
        const int take = (n - next) < (64 - keep) ? (n - next) : (64 - keep);
        if (lane >= keep && lane < keep + take) e = LaneList{sc[next + lane - keep], ix[next + lane - keep]};
        next += take;
        real = lanelist_sort(e, keep + take, lane);
    }
    return real < k ? real : k;
}

__host__ __device__ inline int merge_survivor_cap(int n, int k) { const int c = 16 * k * k + 64; return c < n ? c : n; }
__device__ __forceinline__ void merge_lists_in_lds(const float* sc, const int32_t* ix, float* ssc, int32_t* six, int* lcnt, int groups, int k,
                                                   int32_t* __restrict__ out_idx, float* __restrict__ out_score, int tid, int lane);

// FUSE (one query per pass): the workgroup whose lists arrive LAST merges them and writes the final top-k -- no second launch (r03: the
// merge kernel was 10.8 us of a 77 us call).  Hand-off per MI355X_MICROARCH "Valid forms", first row of the sc1 table: every list entry is
// stored and loaded with agent-scope relaxed atomics (global_store / global_load ... sc1: write-through, L1-bypassing), every storing wave
// drains its stores (vmcnt(0)) before the workgroup barrier, ONE lane then adds to the arrival counter, and the workgroup that sees the
// last ticket loads the lists behind a barrier that follows the returned add.  The counter is left at zero for the next call.
template <int DJ, int STREAM_ROWS, int NQP, bool FUSE = false>      // D = 256 * DJ; NQP = queries of the pass rounded up to 1, 2 or 4
__global__ __launch_bounds__(256) void stream_topk_kernel(const float* __restrict__ Qf, const float* __restrict__ Gf, int Ng,
                                                          const int32_t* __restrict__ exq, const int32_t* __restrict__ exg, int nq,
                                                          int k, float* __restrict__ part_score, int32_t* __restrict__ part_idx,
                                                          int32_t* __restrict__ counter = nullptr, int32_t* __restrict__ out_idx = nullptr,
                                                          float* __restrict__ out_score = nullptr) {
    constexpr int D = 256 * DJ;
    __shared__ float lsc[SQ][4][STREAM_K_MAX];
    __shared__ int32_t lix[SQ][4][STREAM_K_MAX];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    f32x4 qv[SQ][DJ];
    int eq[SQ];
    LaneList list[SQ];
    int cnt[SQ]; float bar_s[SQ]; int bar_i[SQ];           // wave-uniform
#pragma unroll
    for (int q = 0; q < SQ; ++q) {
        const int qq = q < nq ? q : nq - 1;
#pragma unroll
        for (int j = 0; j < DJ; ++j) qv[q][j] = *(const f32x4*)(Qf + (size_t)qq * D + lane * 4 + 256 * j);
        eq[q] = exq ? exq[qq] : -1;
        list[q] = LaneList{-INFINITY, -1};
        cnt[q] = 0; bar_s[q] = -INFINITY; bar_i[q] = 0x7fffffff;
    }
    const int nwaves = gridDim.x * 4, gw = blockIdx.x * 4 + w;
    for (long base = (long)gw * STREAM_ROWS; base < Ng; base += (long)nwaves * STREAM_ROWS) {
        f32x4 gv[STREAM_ROWS][DJ];
#pragma unroll
        for (int r = 0; r < STREAM_ROWS; ++r) {
            const long row = base + r < Ng ? base + r : Ng - 1;
#pragma unroll
            for (int j = 0; j < DJ; ++j) gv[r][j] = __builtin_nontemporal_load((const f32x4*)(Gf + (size_t)row * D + lane * 4 + 256 * j));
        }
        // All ROWS x nq partial dot products of this iteration are reduced over the wave TOGETHER: at each butterfly level a lane
        // keeps one half of its values and sends the other half to its partner (8 + 4 + 2 + 1 exchanges for 16 values, then two plain
        // levels), instead of six exchanges per value.  Every value is still summed over exactly the pairs of wave_sum's xor
        // butterfly (32, 16, ..., 1), so the totals are bit-identical to select_kernel's; value v ends up in lanes [64/NV * v, ...).
        // Used for up to 8 values (1-2 queries per pass); four queries reduce value by value.
        {
            constexpr int NV = STREAM_ROWS * NQP;
            float v[NV];
#pragma unroll
            for (int r = 0; r < STREAM_ROWS; ++r)
#pragma unroll
                for (int q = 0; q < NQP; ++q) {
                    float sdot = 0.f;
#pragma unroll
                    for (int j = 0; j < DJ; ++j) sdot = dot4_acc(sdot, qv[q][j], gv[r][j]);   // the arithmetic of select_kernel, term for term
                    v[r * NQP + q] = sdot;
                }
            constexpr bool TOGETHER = NV <= 8;      // (16 values together cost 172 VGPRs -> two waves per SIMD: 206 us instead of 125)
            if constexpr (TOGETHER) {
                int dist = 32;
#pragma unroll
                for (int n = NV; n > 1; n >>= 1, dist >>= 1) {
                    const bool hi = (lane & dist) != 0;
#pragma unroll
                    for (int i = 0; i < n / 2; ++i) {
                        const float send = hi ? v[i] : v[i + n / 2];
                        const float keep = hi ? v[i + n / 2] : v[i];
                        v[i] = keep + __shfl_xor(send, dist, 64);
                    }
                }
#pragma unroll
                for (int dd = 32 / NV; dd >= 1; dd >>= 1) v[0] += __shfl_xor(v[0], dd, 64);
            } else {
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
            }
#pragma unroll
            for (int r = 0; r < STREAM_ROWS; ++r) {
                const int row = (int)(base + r);
                if (row >= Ng) break;                               // wave-uniform
                const int eg = exg ? exg[row] : -2;
#pragma unroll
                for (int q = 0; q < NQP; ++q) {
                    if (q >= nq) break;
                    float s = TOGETHER ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[0]), (64 / NV) * (r * NQP + q)))
                                       : __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v[r * NQP + q])));
                    if (eq[q] >= 0 && eg == eq[q]) s = -1e9f;
                    if (ranks_before(s, row, bar_s[q], bar_i[q])) {     // scalar branch
                        if (lane == cnt[q]) { list[q].s = s; list[q].i = row; }
                        if (++cnt[q] == 64) {
                            lanelist_sort(list[q], 64, lane);
                            cnt[q] = k;
                            bar_s[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, list[q].s), k - 1));
                            bar_i[q] = __builtin_amdgcn_readlane(list[q].i, k - 1);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < SQ; ++q) {
        if (q >= nq) break;
        lanelist_sort(list[q], cnt[q], lane);
        if (lane < STREAM_K_MAX) {
            const bool ok = lane < cnt[q] && lane < k;
            lsc[q][w][lane] = ok ? list[q].s : -INFINITY;
            lix[q][w][lane] = ok ? list[q].i : -1;
        }
    }
    __syncthreads();
    // the workgroup's list of each query = best k of its four wave lists; wave q takes query q
    if (w < nq) {
        LaneList e;
        const int real = wave_select_lds(&lsc[w][0][0], &lix[w][0][0], 4 * STREAM_K_MAX, k, lane, e);
        if (lane < k) {
            const size_t o = ((size_t)w * gridDim.x + blockIdx.x) * k + lane;
            if constexpr (FUSE) {
                __hip_atomic_store(part_score + o, lane < real ? e.s : -INFINITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(part_idx + o, lane < real ? e.i : -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                part_score[o] = lane < real ? e.s : -INFINITY;
                part_idx[o] = lane < real ? e.i : -1;
            }
        }
    }
    if constexpr (FUSE) {
        extern __shared__ __attribute__((aligned(16))) char smm[];
        __shared__ int last_flag, lcnt;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's list entries have left for memory
        __syncthreads();
        if (tid == 0) {
            const int ticket = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag = ticket == (int)gridDim.x - 1;
            lcnt = 0;
        }
        __syncthreads();                                              // (every wave loads behind the barrier that follows the returned add)
        if (!last_flag) return;
        const int groups = gridDim.x, n = groups * k, n4 = (n + 3) & ~3;
        float* sc = (float*)smm;
        int32_t* ix = (int32_t*)(sc + n4);
        float* ssc = (float*)(ix + n4);
        int32_t* six = (int32_t*)(ssc + merge_survivor_cap(n, k));
        // all lists -> LDS by LDS-DMA with the sc1 policy (16 bytes per lane, L1 bypassed: the table row's `buffer_load_dwordx4` form); every
        // load is in flight before the one wait (a loop of 4-byte atomic loads was issued one round trip at a time: 86 vs 77 us per call)
        {
            const int nchunks = n4 >> 2;                              // 16-byte chunks per array (the workspace extends beyond both arrays)
            const int w4 = tid >> 6;
            for (int c0 = w4 * 64; c0 < nchunks; c0 += 256) {
                if (c0 + lane < nchunks) {
                    __builtin_amdgcn_global_load_lds((gptr_t)(part_score + 4 * (size_t)(c0 + lane)), (lptr_t)((char*)sc + (size_t)c0 * 16), 16, 0, 16);
                    __builtin_amdgcn_global_load_lds((gptr_t)(part_idx + 4 * (size_t)(c0 + lane)), (lptr_t)((char*)ix + (size_t)c0 * 16), 16, 0, 16);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (tid == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next call (stream-ordered)
        __syncthreads();
        merge_lists_in_lds(sc, ix, ssc, six, &lcnt, groups, k, out_idx, out_score, tid, lane);
    }
}


// (all entries in LDS: sc / ix [groups * k], lists sorted best-first; ssc / six: survivor region; *lcnt == 0 on entry; 256 threads)
__device__ __forceinline__ void merge_lists_in_lds(const float* sc, const int32_t* ix, float* ssc, int32_t* six, int* lcnt, int groups, int k,
                                                   int32_t* __restrict__ out_idx, float* __restrict__ out_score, int tid, int lane) {
    // every wave derives the bar for itself (no cross-wave exchange)
    LaneList hb{-INFINITY, -1};
    for (int g = lane; g < groups; g += 64) {
        const float hs = sc[g * k]; const int hi = ix[g * k];
        if (hi >= 0 && (hb.i < 0 || ranks_before(hs, hi, hb.s, hb.i))) hb = LaneList{hs, hi};
    }
    const int nh = lanelist_sort(hb, 64, lane);
    float bar_s = -INFINITY; int bar_i = 0x7fffffff;                // fewer than k non-empty lanes: no bar
    if (nh >= k) {
        bar_s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hb.s), k - 1));
        bar_i = __builtin_amdgcn_readlane(hb.i, k - 1);
    }
    for (int g = tid; g < groups; g += 256) {
        const float* ls = sc + g * k; const int32_t* li = ix + g * k;
        for (int c = 0; c < k; ++c) {                               // lists are sorted: stop at the first entry behind the bar
            const float s = ls[c]; const int gi = li[c];
            if (gi < 0 || ranks_before(bar_s, bar_i, s, gi)) break;
            const int pos = atomicAdd(lcnt, 1);
            ssc[pos] = s; six[pos] = gi;
        }
    }
    __syncthreads();
    if (tid >= 64) return;
    LaneList e;
    const int real = wave_select_lds(ssc, six, *lcnt, k, lane, e);
    if (lane < k) {
        out_idx[lane] = lane < real ? e.i : -1;
        out_score[lane] = lane < real ? e.s : -INFINITY;
    }
}

// One workgroup per query merges the workgroups' lists, each sorted best-first.  ALL entries come into LDS with one round of
// coalesced 16-byte loads (r01 walked the lists in global memory: head, then entry after entry, each a dependent L2 round trip --
// 16.6 us for 80 KB, a fifth of a single-query call; this form 10.9 us, the wave-wide rank sorts of ~1.2 us each being what is left;
// 1024 threads for the staging: 12.8 us).  In LDS: the k-th best of 64 list heads (the best head each lane sees) is a
// bar no result can rank behind, and only lists whose head clears it can hold entries that do: one pass over the lists leaves a
// few dozen survivors (at most 16 k^2: fewer than k lanes have a head above the bar, each lane stands for <= 16 lists of k entries)
// in a second LDS region, from which one wave takes the k best.  Used when both regions fit the LDS (k <= 12 at the default list
// budget); otherwise the global-memory form below.
__global__ __launch_bounds__(256) void stream_merge_lds_kernel(const float* __restrict__ part_score, const int32_t* __restrict__ part_idx,
                                                           int groups, int k, int32_t* __restrict__ out_idx, float* __restrict__ out_score) {
    extern __shared__ __attribute__((aligned(16))) char smm[];
    const int n = groups * k;
    const int n4 = (n + 3) & ~3;
    float* sc = (float*)smm;                       // [n4] all scores
    int32_t* ix = (int32_t*)(sc + n4);             // [n4] all indices
    const int cap = merge_survivor_cap(n, k);
    float* ssc = (float*)(ix + n4);                // [cap] survivors
    int32_t* six = (int32_t*)(ssc + cap);
    __shared__ int lcnt;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const float* ps = part_score + (size_t)q * n;
    const int32_t* pi = part_idx + (size_t)q * n;
    if (tid == 0) lcnt = 0;
    if ((n & 3) == 0 && ((((size_t)q * n) & 3) == 0)) {
        for (int i = tid * 4; i < n; i += 1024) {
            *(f32x4*)(sc + i) = *(const f32x4*)(ps + i);
            *(int4*)(ix + i) = *(const int4*)(pi + i);
        }
    } else {
        for (int i = tid; i < n; i += 256) { sc[i] = ps[i]; ix[i] = pi[i]; }
    }
    __syncthreads();
    merge_lists_in_lds(sc, ix, ssc, six, &lcnt, groups, k, out_idx + (size_t)q * k, out_score + (size_t)q * k, tid, lane);
}

// One workgroup per query merges the workgroups' lists, each sorted best-first.  The k-th best of 64 list heads (the best head
// each lane sees) is a bar no result can rank behind, and only lists whose head clears it can hold entries that do: one pass
// over the lists leaves a few dozen survivors in LDS (n in the worst case: the buffer holds them all), from which one wave
// takes the k best.
__global__ __launch_bounds__(256) void stream_merge_kernel(const float* __restrict__ part_score, const int32_t* __restrict__ part_idx,
                                                           int groups, int k, int32_t* __restrict__ out_idx, float* __restrict__ out_score) {
    extern __shared__ char smm[];
    const int n = groups * k;
    float* sc = (float*)smm;
    int32_t* ix = (int32_t*)(sc + n);
    __shared__ int lcnt;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const float* ps = part_score + (size_t)q * n;
    const int32_t* pi = part_idx + (size_t)q * n;
    if (tid == 0) lcnt = 0;
    // every wave derives the bar for itself (no cross-wave exchange)
    LaneList hb{-INFINITY, -1};
    for (int g = lane; g < groups; g += 64) {
        const float hs = ps[(size_t)g * k]; const int hi = pi[(size_t)g * k];
        if (hi >= 0 && (hb.i < 0 || ranks_before(hs, hi, hb.s, hb.i))) hb = LaneList{hs, hi};
    }
    const int nh = lanelist_sort(hb, 64, lane);
    float bar_s = -INFINITY; int bar_i = 0x7fffffff;                // fewer than k non-empty lanes: no bar
    if (nh >= k) {
        bar_s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hb.s), k - 1));
        bar_i = __builtin_amdgcn_readlane(hb.i, k - 1);
    }
    __syncthreads();
    for (int g = tid; g < groups; g += 256) {
        const float* ls = ps + (size_t)g * k; const int32_t* li = pi + (size_t)g * k;
        for (int c = 0; c < k; ++c) {                               // lists are sorted: stop at the first entry behind the bar
            const float s = ls[c]; const int gi = li[c];
            if (gi < 0 || ranks_before(bar_s, bar_i, s, gi)) break;
            const int pos = atomicAdd(&lcnt, 1);
            sc[pos] = s; ix[pos] = gi;
        }
    }
    __syncthreads();
    if (tid >= 64) return;
    LaneList e;
    const int real = wave_select_lds(sc, ix, lcnt, k, lane, e);
    if (lane < k) {
        out_idx[(size_t)q * k + lane] = lane < real ? e.i : -1;
        out_score[(size_t)q * k + lane] = lane < real ? e.s : -INFINITY;
    }
}

// ------------------------------------------------------------------------------------------ phase C, fast form (k <= 32)
// select_kernel walked ALL candidates of a query (~400 at 200k rows) four at a time, each step a dependent global load of the
// candidate's index and a wave-uniform branch: ~100 steps of ~1 us = 105 us per call at 128 queries, the largest kernel of a
// mid-size retrieval and ~0.5 ms of the 10k-query one (rocprofv3, r03).  Here every phase is one parallel sweep:
//   1. candidate scores -> LDS; every thread keeps the two largest of its share, one wave takes the k-th largest of those 512
//      values: a lower bound a of the k-th largest candidate score (exact unless three of the top k share a thread);
//   2. survivors (score >= a - 2 eps: about k + a few) are compacted with one LDS atomic each, their gallery indices fetched
//      in ONE round of loads;
//   3. the waves re-score the survivors from the fp32 rows, four rows in flight per wave (dot4_acc order: same bits as everywhere);
//   4. one wave sorts them by (score desc, index asc).
// More than SELECT_SV survivors (thousands of exact ties): the query is flagged like a candidate-list overflow (exact fallback).
constexpr int SELECT_SV = 512;

__global__ __launch_bounds__(256) void select_fast_kernel(const float* __restrict__ Qf, const float* __restrict__ Gf, int D,
                                                          const int32_t* __restrict__ exq, const int32_t* __restrict__ exg,
                                                          const int32_t* __restrict__ cand_idx, const float* __restrict__ cand_score,
                                                          const int32_t* __restrict__ cand_cnt, int cap, int k,
                                                          int32_t* __restrict__ out_idx, float* __restrict__ out_score, int Nq, int cnt_stride) {
    extern __shared__ __attribute__((aligned(16))) char smf[];
    float* sc = (float*)smf;                               // [cap] 16-bit-operand scores of the candidates
    float* qrow = sc + ((cap + 3) & ~3);                   // [D], 16-byte aligned
    float* t2s = qrow + D;                                 // [512] per-thread top two
    int32_t* t2i = (int32_t*)(t2s + 512);                  // [512] (positions: distinct keys for the sort)
    float* svs = (float*)(t2i + 512);                      // [SELECT_SV] survivors: exact scores
    int32_t* svi = (int32_t*)(svs + SELECT_SV);            // [SELECT_SV] gallery indices
    int32_t* svc = svi + SELECT_SV;                        // [SELECT_SV] candidate positions
    __shared__ int nsv;
    __shared__ float thr2;
    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int cnt = cand_cnt[(size_t)q * cnt_stride];
    if (cnt > cap) {                                       // overflow: caller takes the exact fallback
        if (tid == 0) { out_idx[(size_t)q * k] = -2; out_score[(size_t)q * k] = 0.f; }
        return;
    }
    if (tid == 0) nsv = 0;
    for (int i = tid * 4; i < D; i += 1024) *(f32x4*)(qrow + i) = *(const f32x4*)(Qf + (size_t)q * D + i);
    const float* cs = cand_score + (size_t)q * cap;
    float a0 = -INFINITY, a1 = -INFINITY;
    for (int c = tid; c < cnt; c += 256) {
        const float v = cs[c];
        sc[c] = v;
        const float hi = fmaxf(a0, v);
        a1 = fmaxf(a1, fminf(a0, v));
        a0 = hi;
    }
    t2s[tid] = a0; t2i[tid] = a0 > -INFINITY ? tid : -1;
    t2s[256 + tid] = a1; t2i[256 + tid] = a1 > -INFINITY ? 256 + tid : -1;
    __syncthreads();
    if (w == 0) {
        LaneList e;
        const int real = wave_select_lds(t2s, t2i, 512, k, lane, e);
        const float kth = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e.s), real > 0 ? real - 1 : 0));
        if (lane == 0) thr2 = (real < k || cnt < k) ? -INFINITY : kth - 2.f * EPS_BF16;
    }
    __syncthreads();
    const float t2 = thr2;
    for (int c = tid; c < cnt; c += 256) {
        if (sc[c] >= t2) {
            const int pos = atomicAdd(&nsv, 1);
            if (pos < SELECT_SV) svc[pos] = c;
        }
    }
    __syncthreads();
    const int n = nsv;
    if (n > SELECT_SV) {
        if (tid == 0) { out_idx[(size_t)q * k] = -2; out_score[(size_t)q * k] = 0.f; }
        return;
    }
    for (int i = tid; i < n; i += 256) svi[i] = cand_idx[(size_t)q * cap + svc[i]];
    __syncthreads();
    const int eq = exq ? exq[q] : -1;
    const int nd = D >> 8;                                 // 16-byte pieces per lane (D = 256 nd; D % 64 == 0: a ragged tail below)
    for (int i0 = w * 4; i0 < n; i0 += 16) {               // four rows in flight per wave
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        int gi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) gi[u] = svi[i0 + u < n ? i0 + u : n - 1];
        for (int j = 0; j <= nd; ++j) {
            const int i = lane * 4 + j * 256;
            if (i >= D) break;
            const f32x4 a = *(const f32x4*)(qrow + i);
            f32x4 b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) b[u] = *(const f32x4*)(Gf + (size_t)gi[u] * D + i);
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] = dot4_acc(s[u], a, b[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v = wave_sum(s[u]);
            if (eq >= 0 && exg[gi[u]] == eq) v = -1e9f;
            if (lane == 0 && i0 + u < n) svs[i0 + u] = v;
        }
    }
    __syncthreads();
    if (w != 0) return;
    LaneList e;
    const int real = wave_select_lds(svs, svi, n, k, lane, e);
    if (lane < k) {
        out_idx[(size_t)q * k + lane] = lane < real ? e.i : -1;
        out_score[(size_t)q * k + lane] = lane < real ? e.s : -INFINITY;
    }
}

int launch_select_fast(const float* Qf, const float* Gf, int D, const int32_t* exq, const int32_t* exg, const int32_t* cand_idx,
                       const float* cand_score, const int32_t* cand_cnt, int cap, int k, int32_t* out_idx, float* out_score, int Nq,
                       hipStream_t s, int cnt_stride) {
    const size_t lds = (size_t)((cap + 3) & ~3) * 4 + (size_t)D * 4 + 512 * 8 + (size_t)SELECT_SV * 12;
    REID_MAX_LDS((select_fast_kernel), 8192 * 4 + 1024 * 4 + 512 * 8 + SELECT_SV * 12);
    hipLaunchKernelGGL(select_fast_kernel, dim3(Nq), dim3(256), lds, s, Qf, Gf, D, exq, exg, cand_idx, cand_score, cand_cnt, cap, k, out_idx,
                       out_score, Nq, cnt_stride);
    REID_CHECK_LAUNCH("reid_cosine_topk(select)");
    return REID_OK;
}

}  // namespace

extern "C" int32_t reid_topk_stream_ok(int32_t Nq, int32_t Ng, int32_t D, int32_t k) {
    return Nq >= 1 && Nq <= SQ && k >= 1 && k <= STREAM_K_MAX && k <= Ng && D % 256 == 0 && D >= 256 && D <= 1024 && Ng >= 1;
}
// lists of SQ queries + 256 bytes for the arrival counter of the one-query form (zero before the first call; every call leaves it zero)
extern "C" int64_t reid_topk_stream_ws_bytes(int32_t k) { return (int64_t)SQ * stream_groups(k) * k * 8 + 256; }

/* Top-k of a few queries in ONE pass over the fp32 gallery (the reference's one-query-at-a-time form).  Same results as
 * reid_cosine_topk; allowed when reid_topk_stream_ok().  ws: reid_topk_stream_ws_bytes(k). */
extern "C" int reid_cosine_topk_stream(const float* Qf, const float* Gf, int32_t Nq, int32_t Ng, int32_t D, int32_t k,
                                       const int32_t* exclude_q, const int32_t* exclude_g, void* ws, int32_t* out_idx,
                                       float* out_score, void* stream) {
    REID_CHECK_ARG(Qf && Gf && ws && out_idx && out_score, "reid_cosine_topk_stream: null pointer");
    REID_CHECK_ARG(reid_topk_stream_ok(Nq, Ng, D, k), "reid_cosine_topk_stream: Nq=%d Ng=%d D=%d k=%d outside the streaming form", Nq, Ng, D, k);
    REID_CHECK_ARG((exclude_q == nullptr) == (exclude_g == nullptr), "reid_cosine_topk_stream: exclude_q and exclude_g go together");
    REID_CHECK_ARG((((uintptr_t)Qf | (uintptr_t)Gf) & 15) == 0, "reid_cosine_topk_stream: operands must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int groups = stream_groups(k);
    // one query: 512 workgroups scan as fast as 1024 and leave the merge half as many lists (77.6 vs 81.8 us per call); with 2-4 queries
    // per pass the larger grid wins (91.9 vs 102.6 us, 111 vs 135 us).  REID_STREAM_GROUPS overrides.
    if (Nq == 1 && groups > 512) groups = 512;
    if (reid_knob(KNOB_STREAM_GROUPS) > 0 && reid_knob(KNOB_STREAM_GROUPS) < stream_groups(k)) groups = reid_knob(KNOB_STREAM_GROUPS);
    const int need = (Ng + 15) / 16;        // no more workgroups than there is work
    if (groups > need) groups = need;
    float* ps = (float*)ws;
    int32_t* pi = (int32_t*)(ps + (size_t)SQ * groups * k);
    REID_MAX_LDS((stream_merge_kernel), STREAM_LIST_BUDGET * 8);
    REID_MAX_LDS((stream_merge_lds_kernel), 160 * 1024 - 64);
    for (int q0 = 0; q0 < Nq; q0 += SQ) {
        const int nq = Nq - q0 < SQ ? Nq - q0 : SQ;
        const float* Q = Qf + (size_t)q0 * D;
        const int32_t* eq = exclude_q ? exclude_q + q0 : nullptr;
        // gallery rows per wave and iteration (the loads of all rows are issued before the first dot): 4 rows = 8 KiB in flight per wave at D = 512
        const int rows_env = reid_knob(KNOB_STREAM_ROWS);
        const int n_ent = groups * k;
        const size_t lds_fast = (size_t)((n_ent + 3) & ~3) * 8 + (size_t)merge_survivor_cap(n_ent, k) * 8;
        // REID_STREAM_FUSE=1 (one query, lists that fit the LDS next to a second resident workgroup): the last-arriving workgroup merges inside
        // the scan launch.  Built for the r03 verdict's ">= 70 % of HBM end to end"; measured (r04, profiles/r04_stream_fused_merge.log):
        // 78.1-78.5 us per call against 77.0-78 us with the separate merge launch -- the launch boundary it removes (~1.5 us) is what the
        // hand-off costs (drain + barrier + ticket + 40 KiB of sc1 loads behind the slowest scan workgroup), so it is OFF by default.
        const bool fuse = nq == 1 && lds_fast <= 72 * 1024 && reid_knob(KNOB_STREAM_FUSE) == 1;
        int32_t* counter = (int32_t*)((char*)ws + (size_t)SQ * stream_groups(k) * k * 8);
#define REID_STREAM_LAUNCH1(DJ, R, P) hipLaunchKernelGGL((stream_topk_kernel<DJ, R, P>), dim3(groups), dim3(256), 0, s, Q, Gf, Ng, eq, exclude_g, nq, k, ps, pi)
#define REID_STREAM_LAUNCHF(DJ, R) do { REID_MAX_LDS((stream_topk_kernel<DJ, R, 1, true>), 72 * 1024); \
        hipLaunchKernelGGL((stream_topk_kernel<DJ, R, 1, true>), dim3(groups), dim3(256), lds_fast, s, Q, Gf, Ng, eq, exclude_g, nq, k, ps, pi, counter, \
                           out_idx + (size_t)q0 * k, out_score + (size_t)q0 * k); } while (0)
#define REID_STREAM_LAUNCH(DJ, R) do { if (fuse) REID_STREAM_LAUNCHF(DJ, R); else if (nq == 1) REID_STREAM_LAUNCH1(DJ, R, 1); else if (nq == 2) REID_STREAM_LAUNCH1(DJ, R, 2); else REID_STREAM_LAUNCH1(DJ, R, 4); } while (0)
        switch (D / 256) {
            case 1: REID_STREAM_LAUNCH(1, 4); break;
            case 2: if (rows_env == 2) REID_STREAM_LAUNCH(2, 2); else if (rows_env == 8) REID_STREAM_LAUNCH(2, 8); else REID_STREAM_LAUNCH(2, 4); break;
            case 3: REID_STREAM_LAUNCH(3, 2); break;
            default: REID_STREAM_LAUNCH(4, 2); break;
        }
#undef REID_STREAM_LAUNCH
#undef REID_STREAM_LAUNCHF
#undef REID_STREAM_LAUNCH1
        REID_CHECK_LAUNCH("reid_cosine_topk_stream(scan)");
        if (fuse) continue;
        if (lds_fast <= 160 * 1024 - 64)
            hipLaunchKernelGGL(stream_merge_lds_kernel, dim3(nq), dim3(256), lds_fast, s, ps, pi, groups, k, out_idx + (size_t)q0 * k,
                               out_score + (size_t)q0 * k);
        else
            hipLaunchKernelGGL(stream_merge_kernel, dim3(nq), dim3(256), (size_t)groups * k * 8, s, ps, pi, groups, k, out_idx + (size_t)q0 * k,
                               out_score + (size_t)q0 * k);
        REID_CHECK_LAUNCH("reid_cosine_topk_stream(merge)");
    }
    return REID_OK;
}

/* Exact fp32 pass for queries flagged -2 by reid_cosine_topk (candidate overflow).  scratch: Nq*Ng floats. */
extern "C" int reid_cosine_topk_exact(const float* Qf, const float* Gf, int32_t Nq, int32_t Ng, int32_t D, int32_t k,
                                      const int32_t* exclude_q, const int32_t* exclude_g, float* scratch, int32_t* out_idx,
                                      float* out_score, void* stream) {
    REID_CHECK_ARG(Qf && Gf && scratch && out_idx && out_score && Nq > 0 && Ng > 0 && k > 0 && k <= Ng && D % 4 == 0 && D <= 1024,
                   "reid_cosine_topk_exact: bad args");
    hipLaunchKernelGGL(brute_score_kernel, dim3(64, Nq), dim3(256), 0, (hipStream_t)stream, Qf, Gf, Ng, D, exclude_q, exclude_g, k, out_idx, scratch,
                       (const int32_t*)nullptr, 0);
    REID_CHECK_LAUNCH("reid_cosine_topk_exact(score)");
    hipLaunchKernelGGL(brute_select_kernel, dim3(Nq), dim3(256), 0, (hipStream_t)stream, Ng, k, out_idx, out_score, scratch, (const int32_t*)nullptr, 0);
    REID_CHECK_LAUNCH("reid_cosine_topk_exact(select)");
    return REID_OK;
}

namespace {
__global__ __launch_bounds__(256) void flag_compact_kernel(const int32_t* __restrict__ out_idx, int Nq, int k, int32_t* __restrict__ slots, int n_slots) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Nq || out_idx[(size_t)q * k] != -2) return;
    const int pos = atomicAdd(slots, 1);
    if (pos < n_slots) slots[1 + pos] = q;
}
}  // namespace

extern "C" int reid_cosine_topk_exact_slots(const float* Qf, const float* Gf, int32_t Nq, int32_t Ng, int32_t D, int32_t k,
                                            const int32_t* exclude_q, const int32_t* exclude_g, int32_t n_slots, int32_t* slots,
                                            float* scratch, int32_t* out_idx, float* out_score, void* stream) {
    REID_CHECK_ARG(Qf && Gf && scratch && slots && out_idx && out_score && Nq > 0 && Ng > 0 && k > 0 && k <= Ng && D % 4 == 0 && D <= 1024 &&
                   n_slots > 0, "reid_cosine_topk_exact_slots: bad args");
    hipStream_t s = (hipStream_t)stream;
    REID_CHECK_HIP(hipMemsetAsync(slots, 0, sizeof(int32_t), s), "hipMemsetAsync");
    hipLaunchKernelGGL(flag_compact_kernel, dim3((Nq + 255) / 256), dim3(256), 0, s, out_idx, Nq, k, slots, n_slots);
    REID_CHECK_LAUNCH("reid_cosine_topk_exact_slots(compact)");
    // workgroup rows walk the list with a stride: 256 rows in flight whatever the list's length (the usual length is 0: 256 x 64 workgroups
    // that read one word and leave)
    const int rows = n_slots < 256 ? n_slots : 256;
    hipLaunchKernelGGL(brute_score_kernel, dim3(64, rows), dim3(256), 0, s, Qf, Gf, Ng, D, exclude_q, exclude_g, k, out_idx, scratch,
                       (const int32_t*)slots, n_slots);
    REID_CHECK_LAUNCH("reid_cosine_topk_exact_slots(score)");
    hipLaunchKernelGGL(brute_select_kernel, dim3(rows), dim3(256), 0, s, Ng, k, out_idx, out_score, scratch, (const int32_t*)slots, n_slots);
    REID_CHECK_LAUNCH("reid_cosine_topk_exact_slots(select)");
    return REID_OK;
}
