// On-device AP / CMC from a block of fp32 similarity rows (reference: tools/eval_mm_protocol.py:401-469 rank_and_metrics,
// train.py:450-479 _reid_map).
//
// The reference sorts every query's whole gallery row (argsort over Ng) and walks it in Python.  The metric needs far
// less than the permutation: only the RANK of each positive, i.e. how many unmasked non-positive gallery entries precede
// it.  One workgroup per query:
//   1. the query's positives come from a pid -> gallery-rows CSR built once per gallery; same-image rows are dropped
//      (the `& mask` of eval_mm_protocol.py:428), their scores are gathered and bitonic-sorted in LDS by
//      (score descending, gallery index ascending) -- the tie rule of a stable descending argsort;
//   2. ONE streaming pass over the score row (16-byte loads): an entry below the weakest positive is skipped outright,
//      anything else is placed among the sorted positives (score-bucket table, then a search inside the bucket) and counted
//      in an LDS histogram;
//   3. prefix sum of the histogram -> rank of the r-th positive = 1 + r + #non-positives before it;
//      AP = mean_r (r + 1) / rank_r (double), first-positive rank for CMC@k.
// HBM-bound: 4 bytes of score + 8 bytes of L2-resident pid / image id per gallery entry per query.
#include "common.h"

namespace {

constexpr int NBUCKET = 2048;      // score buckets between the weakest and the strongest positive of a query
constexpr int MAXP_LIMIT = 8192;    // positives per query held in LDS (12 bytes each; more -> npos = -1, not evaluated)

struct MetricParams {
    const float* S; long long lds_;            // scores [nq, lds_]
    const int32_t* g_pid; const int32_t* g_img;  // [Ng]; g_img may be null
    const int32_t* q_pid; const int32_t* q_slot; // [nq]; CSR row of the query's pid, -1 = pid absent from the gallery
    const int32_t* q_excl;                     // [nq, 4] image ids to ignore (-1 = none); may be null
    const int32_t* csr_off; const int32_t* csr_idx;
    double* ap; int32_t* rank1; int32_t* npos;
    int nq, Ng, cap;                           // cap: power of two >= the longest CSR row, LDS capacity of this launch
};

__device__ __forceinline__ bool before(float sa, int ia, float sb, int ib) {   // (sa, ia) ranks before (sb, ib)
    return sa > sb || (sa == sb && ia < ib);
}

__global__ __launch_bounds__(256) void rank_metrics_kernel(const MetricParams p) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    float* ps = (float*)dyn;
    int* pi = (int*)(dyn + 4 * (size_t)p.cap);
    int* hist = (int*)(dyn + 8 * (size_t)p.cap);           // cap + 1 entries
    const int MAXP = p.cap;
    __shared__ int cnt;
    __shared__ double red[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    const float* row = p.S + (long long)q * p.lds_;
    const int slot = p.q_slot[q];
    int ex[4] = {-1, -1, -1, -1};
    if (p.q_excl && p.g_img) {
#pragma unroll
        for (int k = 0; k < 4; ++k) ex[k] = p.q_excl[q * 4 + k];
    }
    const bool has_ex = (ex[0] & ex[1] & ex[2] & ex[3]) != -1;      // any id other than -1
    auto excluded = [&](int j) {
        if (!has_ex) return false;
        const int g = p.g_img[j];
        return g >= 0 && (g == ex[0] || g == ex[1] || g == ex[2] || g == ex[3]);
    };
    if (tid == 0) cnt = 0;
    __syncthreads();
    if (slot >= 0) {
        const int b = p.csr_off[slot], e = p.csr_off[slot + 1];
        for (int t = b + tid; t < e; t += 256) {
            const int j = p.csr_idx[t];
            if (excluded(j)) continue;
            const int o = atomicAdd(&cnt, 1);
            if (o < MAXP) { ps[o] = row[j]; pi[o] = j; }
        }
    }
    __syncthreads();
    const int np = cnt;
    if (np == 0 || np > MAXP) {
        if (tid == 0) { p.ap[q] = 0.0; p.rank1[q] = 0; p.npos[q] = np == 0 ? 0 : -1; }
        return;
    }
    int n2 = 1;
    while (n2 < np) n2 <<= 1;
    for (int t = np + tid; t < n2; t += 256) { ps[t] = -INFINITY; pi[t] = 0x7fffffff; }
    for (int t = tid; t <= np; t += 256) hist[t] = 0;
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < n2; t += 256) {
                const int u = t ^ j;
                if (u > t) {
                    const bool up = (t & k) == 0;              // ascending position = earlier rank
                    const float sa = ps[t], sb = ps[u];
                    const int ia = pi[t], ib = pi[u];
                    const bool swap = up ? before(sb, ib, sa, ia) : before(sa, ia, sb, ib);
                    if (swap) { ps[t] = sb; ps[u] = sa; pi[t] = ib; pi[u] = ia; }
                }
            }
            __syncthreads();
        }
    const float s_last = ps[np - 1];
    const int i_last = pi[np - 1];
    const int pid = p.q_pid[q];
    // Score buckets over [weakest, strongest positive]: bucket(s) is monotone in s, so positives in earlier buckets rank before an
    // entry and positives in later buckets after it -- only the positives of the entry's own bucket (usually none) need the
    // (score, index) comparison.  Replaces a log2(np)-step binary search of dependent LDS reads per gallery entry by one lookup
    // (with hundreds of positives per query that search, not HBM, set the pace: 29.9 -> 23.9 ms per 10k x 200k evaluation).
    __shared__ int bstart[NBUCKET + 1];
    __shared__ int bpart[256];
    const float s_first = ps[0];
    const float bscale = s_first > s_last ? (float)NBUCKET / (s_first - s_last) : 0.f;
    auto bucket = [&](float s) {
        const float t = fminf(fmaxf((s_first - s) * bscale, 0.f), (float)(NBUCKET - 1));
        return (int)t;
    };
    for (int b = tid; b <= NBUCKET; b += 256) bstart[b] = 0;
    __syncthreads();
    for (int r = tid; r < np; r += 256) atomicAdd(&bstart[bucket(ps[r])], 1);
    __syncthreads();
    {   // exclusive prefix sum in place: NBUCKET / 256 consecutive buckets per thread
        constexpr int PER = NBUCKET / 256;
        int loc[PER], sum = 0;
#pragma unroll
        for (int u = 0; u < PER; ++u) { loc[u] = bstart[tid * PER + u]; sum += loc[u]; }
        bpart[tid] = sum;
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int t = 0; t < 256; ++t) { const int x = bpart[t]; bpart[t] = run; run += x; }
            bstart[NBUCKET] = run;
        }
        __syncthreads();
        int run = bpart[tid];
#pragma unroll
        for (int u = 0; u < PER; ++u) { bstart[tid * PER + u] = run; run += loc[u]; }
    }
    __syncthreads();
    auto visit = [&](int j, float s) {
        if (!before(s, j, s_last, i_last)) return;             // behind every positive: affects no rank we need
        if (p.g_pid[j] == pid || excluded(j)) return;          // positives are counted by their own position; masked rows rank last
        const int b = bucket(s);
        int lo = bstart[b], hi = bstart[b + 1];                 // first position whose positive is NOT before (s, j)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (before(ps[mid], pi[mid], s, j)) lo = mid + 1; else hi = mid;
        }
        atomicAdd(&hist[lo], 1);
    };
    // the row start is 16-byte aligned when lds_ % 4 == 0 (checked on the host); scalar tail otherwise handled below
    const int n4 = p.Ng >> 2;
    for (int t = tid; t < n4; t += 256) {
        const f32x4 v = *(const f32x4*)(row + 4 * t);
#pragma unroll
        for (int k = 0; k < 4; ++k) visit(4 * t + k, v[k]);
    }
    for (int j = (n4 << 2) + tid; j < p.Ng; j += 256) visit(j, row[j]);
    __syncthreads();
    // inclusive prefix over hist[0 .. np-1] (one thread per ceil(np/256) entries, then a serial pass over 256 partials)
    __shared__ int part[256];
    const int per = (np + 255) / 256;
    int loc = 0;
    for (int t = tid * per; t < min(np, (tid + 1) * per); ++t) loc += hist[t];
    part[tid] = loc;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < 256; ++t) { const int x = part[t]; part[t] = run; run += x; }
    }
    __syncthreads();
    double acc = 0.0;
    {
        int run = part[tid];
        for (int t = tid * per; t < min(np, (tid + 1) * per); ++t) {
            run += hist[t];
            const int rank = 1 + t + run;
            acc += (double)(t + 1) / (double)rank;
            if (t == 0) p.rank1[q] = rank;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) { p.ap[q] = ((red[0] + red[1]) + (red[2] + red[3])) / (double)np; p.npos[q] = np; }
}

}  // namespace

extern "C" int reid_rank_metrics(const float* scores, int64_t ld, const int32_t* g_pid, const int32_t* g_img,
                                 const int32_t* q_pid, const int32_t* q_slot, const int32_t* q_excl, const int32_t* csr_off,
                                 const int32_t* csr_idx, int32_t nq, int32_t Ng, int32_t max_pos, double* ap, int32_t* rank1,
                                 int32_t* npos, void* stream) {
    REID_CHECK_ARG(scores && g_pid && q_pid && q_slot && csr_off && csr_idx && ap && rank1 && npos, "reid_rank_metrics: null pointer");
    REID_CHECK_ARG(nq > 0 && Ng > 0 && ld >= Ng && ld % 4 == 0, "reid_rank_metrics: nq=%d Ng=%d ld=%lld (ld %% 4 == 0)", nq, Ng, (long long)ld);
    REID_CHECK_ARG(((uintptr_t)scores & 15) == 0, "reid_rank_metrics: scores must be 16-byte aligned");
    REID_CHECK_ARG(max_pos >= 1, "reid_rank_metrics: max_pos");
    int cap = 64;
    while (cap < max_pos && cap < MAXP_LIMIT) cap <<= 1;
    MetricParams p{scores, (long long)ld, g_pid, g_img, q_pid, q_slot, q_excl, csr_off, csr_idx, ap, rank1, npos, nq, Ng, cap};
    const int lds = 12 * cap + 16;
    REID_MAX_LDS((rank_metrics_kernel), 12 * MAXP_LIMIT + 16);
    hipLaunchKernelGGL(rank_metrics_kernel, dim3(nq), dim3(256), lds, (hipStream_t)stream, p);
    REID_CHECK_LAUNCH("reid_rank_metrics");
    return REID_OK;
}
