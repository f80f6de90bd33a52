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
                       hipStream_t s);
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
                                                          int32_t* __restrict__ out_idx, float* __restrict__ out_score, int Nq) {
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
    const int cnt = cand_cnt[q];
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
                       hipStream_t s) {
    const size_t lds = (size_t)((cap + 3) & ~3) * 4 + (size_t)D * 4 + 512 * 8 + (size_t)SELECT_SV * 12;
    REID_MAX_LDS((select_fast_kernel), 8192 * 4 + 1024 * 4 + 512 * 8 + SELECT_SV * 12);
    hipLaunchKernelGGL(select_fast_kernel, dim3(Nq), dim3(256), lds, s, Qf, Gf, D, exq, exg, cand_idx, cand_score, cand_cnt, cap, k, out_idx,
                       out_score, Nq);
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
