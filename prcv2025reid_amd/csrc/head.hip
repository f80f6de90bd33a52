// Head of the hot path on gfx950: small exact-fp32 GEMM, BN-neck, label-smoothed CE, SDM loss.
// Shapes here are tiny (rows = P*K per GPU, or the global batch under data parallelism; D = 512),
// so these kernels are HBM/latency bound: one wavefront per row, 16-byte coalesced accesses,
// wave-shuffle reductions, fp32 throughout (the losses are compared with an fp32 oracle at 1e-3).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------ sgemm
// C[M,N] = act(alpha * sum_k A(m,k) B(k,n) + bias[n]) + beta * C, arbitrary element strides for A and B.
// The head's GEMMs are tiny (a few hundred rows, K = 320..2048) and sit on the step's critical path one after another,
// so the kernel is built for LATENCY, not throughput: 32x32 output tiles (hundreds of workgroups even for 320x512),
// the K range split over the workgroup's 4 or 8 waves (each wave streams its own quarter through a wave-private LDS slice:
// no workgroup barrier inside the loop), the next K-step's operands prefetched into registers while the current one is
// multiplied, and ONE deterministic in-LDS reduction of the partial tiles at the end (fixed order: bit-reproducible).
// (The first version -- 64x64 tiles, K = 16 per barrier pair, no prefetch -- took 42 us per call, 4.9 ms per train step.)
constexpr int SG_PITCH = 36;                            // floats per staged k-row: 32 + 4 (conflict-free transposed writes)
template <int SG_WAVES>
__global__ __launch_bounds__(SG_WAVES * 64) void sgemm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                              float* __restrict__ C, int M, int N, int K, long sam, long sak,
                                                              long sbk, long sbn, int ldc, float alpha, float beta,
                                                              const float* __restrict__ bias, int act) {
    REID_T16_ENTER();
    __shared__ __attribute__((aligned(16))) float lds[2 * SG_WAVES * 16 * SG_PITCH];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* As = lds + wave * 16 * SG_PITCH;
    float* Bs = lds + (SG_WAVES + wave) * 16 * SG_PITCH;
    const int tx = lane & 7, ty = lane >> 3;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int kchunk = ((K + 16 * SG_WAVES - 1) / (16 * SG_WAVES)) * 16;   // per-wave K range, a multiple of 16
    const int kb = wave * kchunk;
    const int ke = min(K, kb + kchunk);
    // element e = lane + 64 i of a 32 x 16 operand tile: the fast lane index follows the contiguous stride
    const bool a_kfast = sak == 1, b_kfast = sbk == 1;
    float ra[2][8], rb[2][8];
    auto coords = [&](bool kfast, int i, int& x, int& kk) {
        const int e = lane + 64 * i;
        if (kfast) { kk = e & 15; x = e >> 4; } else { x = e & 31; kk = e >> 5; }
    };
    auto fetch = [&](int k0, int slot) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int x, kk;
            coords(a_kfast, i, x, kk);
            const int m = m0 + x, ka = k0 + kk;
            ra[slot][i] = (m < M && ka < ke) ? A[(long)m * sam + (long)ka * sak] : 0.f;
            coords(b_kfast, i, x, kk);
            const int n = n0 + x, kq = k0 + kk;
            rb[slot][i] = (n < N && kq < ke) ? B[(long)kq * sbk + (long)n * sbn] : 0.f;
        }
    };
    float acc[4][4] = {};
    auto step = [&](int k0, int slot) {                  // slot is a compile-time constant at both call sites
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int x, kk;
            coords(a_kfast, i, x, kk);
            As[kk * SG_PITCH + x] = ra[slot][i];
            coords(b_kfast, i, x, kk);
            Bs[kk * SG_PITCH + x] = rb[slot][i];
        }
        __builtin_amdgcn_wave_barrier();                // wave-private slices: LDS ops of one wave execute in order
        if (k0 + 32 < ke) fetch(k0 + 32, slot);         // two K-steps of operands stay in flight
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const f32x4 a = *(const f32x4*)(As + kk * SG_PITCH + ty * 4);
            const f32x4 b = *(const f32x4*)(Bs + kk * SG_PITCH + tx * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __builtin_amdgcn_wave_barrier();
    };
    if (kb < ke) fetch(kb, 0);
    if (kb + 16 < ke) fetch(kb + 16, 1);
    for (int k0 = kb; k0 < ke; k0 += 32) {
        step(k0, 0);
        if (k0 + 16 < ke) step(k0 + 16, 1);
    }
    __syncthreads();                                     // every wave is done with its staging slice: reuse LDS for the partials
    static_assert(2 * 16 * SG_PITCH >= 1024, "partial tiles alias the staging slices");
    float* red = lds + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) *(f32x4*)(red + (ty * 4 + i) * 32 + tx * 4) = f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
    __syncthreads();
    if (tid >= 256) return;
    const int row = tid >> 3, c4 = (tid & 7) * 4;
    f32x4 sum = *(const f32x4*)(lds + row * 32 + c4);
#pragma unroll
    for (int w = 1; w < SG_WAVES; ++w) {                 // fixed order: bit-reproducible
        const f32x4 t = *(const f32x4*)(lds + w * 1024 + row * 32 + c4);
        sum[0] += t[0]; sum[1] += t[1]; sum[2] += t[2]; sum[3] += t[3];
    }
    const int m = m0 + row;
    if (m >= M) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + c4 + j;
        if (n >= N) continue;
        float v = alpha * sum[j];
        if (bias) v += bias[n];
        if (act == REID_ACT_GELU_ERF) v = gelu_erf_f(v);
        else if (act == REID_ACT_QUICK_GELU) v = quick_gelu_f(v);
        else if (act == REID_ACT_RELU) v = fmaxf(v, 0.f);
        float* c = C + (size_t)m * ldc + n;
        *c = beta == 0.f ? v : v + beta * *c;
    }
}

// ------------------------------------------------------------------------------------------ BN-neck
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int ldx, int rows, int D, int rows_per_block,
                                                       float* __restrict__ sum, float* __restrict__ sqsum) {
    REID_T16_ENTER();
    // block = 64 columns x 4 row lanes; grid = (D/64, row splits)
    __shared__ float s1[4][64], s2[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(rows, r0 + rows_per_block);
    float a = 0.f, b = 0.f;
    if (c < D)
        for (int r = r0 + rl; r < r1; r += 4) { const float v = x[(size_t)r * ldx + c]; a += v; b += v * v; }
    s1[rl][threadIdx.x & 63] = a; s2[rl][threadIdx.x & 63] = b;
    __syncthreads();
    if (rl == 0 && c < D) {
        const int t = threadIdx.x;
        atomicAdd(sum + c, s1[0][t] + s1[1][t] + s1[2][t] + s1[3][t]);
        atomicAdd(sqsum + c, s2[0][t] + s2[1][t] + s2[2][t] + s2[3][t]);
    }
}

__global__ void bn_finalize_kernel(const float* __restrict__ sum, const float* __restrict__ sqsum, float count, int training,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ mean,
                                   float* __restrict__ invstd, int D, float eps, float momentum) {
    REID_T16_ENTER();
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= D) return;
    float mu, var;
    if (training) {
        mu = sum[c] / count;
        var = fmaxf(sqsum[c] / count - mu * mu, 0.f);
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (count / fmaxf(count - 1.f, 1.f));
    } else {
        mu = running_mean[c]; var = running_var[c];
    }
    mean[c] = mu;
    invstd[c] = rsqrtf(var + eps);
}

constexpr int MAXV = 4;

__global__ __launch_bounds__(256) void bnneck_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, float* __restrict__ y,
                                                         bf16_t* __restrict__ yb, int ldy, float* __restrict__ rnorm, int rows,
                                                         int D, float scale) {
    REID_T16_ENTER();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = D >> 2;
    f32x4 z[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nv) {
            const f32x4 xv = *(const f32x4*)(x + (size_t)row * ldx + c * 4);
            const f32x4 g = *(const f32x4*)(gamma + c * 4), b = *(const f32x4*)(beta + c * 4);
            const f32x4 mu = *(const f32x4*)(mean + c * 4), is = *(const f32x4*)(invstd + c * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { z[i][e] = (xv[e] - mu[e]) * is[e] * g[e] + b[e]; s += z[i][e] * z[i][e]; }
        }
    }
    const float rn = 1.0f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    if (lane == 0) rnorm[row] = rn;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const f32x4 o = z[i] * (rn * scale);
            *(f32x4*)(y + (size_t)row * ldy + c * 4) = o;
            if (yb) *(uint2*)(yb + (size_t)row * ldy + c * 4) = uint2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        }
    }
}

__global__ __launch_bounds__(256) void bnneck_bwd1_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ rnorm, float* __restrict__ dz,
                                                          float* __restrict__ sum_dz, float* __restrict__ sum_dz_xhat, int rows,
                                                          int D, float scale) {
    REID_T16_ENTER();
    __shared__ float r1[4][MAXV * 256], r2[4][MAXV * 256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + w;
    const int nv = D >> 2;
    f32x4 dzv[MAXV], xh[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) { dzv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; xh[i] = dzv[i]; }
    if (row < rows) {
        const float rn = rnorm[row];
        f32x4 u[MAXV], g[MAXV];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + i * 64;
            u[i] = f32x4{0.f, 0.f, 0.f, 0.f}; g[i] = u[i];
            if (c < nv) {
                const f32x4 xv = *(const f32x4*)(x + (size_t)row * ldx + c * 4);
                const f32x4 gm = *(const f32x4*)(gamma + c * 4), b = *(const f32x4*)(beta + c * 4);
                const f32x4 mu = *(const f32x4*)(mean + c * 4), is = *(const f32x4*)(invstd + c * 4);
                g[i] = *(const f32x4*)(dy + (size_t)row * lddy + c * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[i][e] = (xv[e] - mu[e]) * is[e];
                    u[i][e] = (xh[i][e] * gm[e] + b[e]) * rn;       // unit vector
                    dot += u[i][e] * g[i][e];
                }
            }
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) dzv[i][e] = scale * rn * (g[i][e] - u[i][e] * dot);
                *(f32x4*)(dz + (size_t)row * D + c * 4) = dzv[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            r1[w][(lane + i * 64) * 4 + e] = dzv[i][e];
            r2[w][(lane + i * 64) * 4 + e] = dzv[i][e] * xh[i][e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        atomicAdd(sum_dz + c, r1[0][c] + r1[1][c] + r1[2][c] + r1[3][c]);
        atomicAdd(sum_dz_xhat + c, r2[0][c] + r2[1][c] + r2[2][c] + r2[3][c]);
    }
}

__global__ void bnneck_bwd2_kernel(const float* __restrict__ dz, const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                   const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ sum_dz,
                                   const float* __restrict__ sum_dz_xhat, float count, int training, float* __restrict__ dx, int lddx,
                                   int rows, int D) {
    REID_T16_ENTER();
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * D) return;
    const int r = i / D, c = i % D;
    const float is = invstd[c], g = gamma[c];
    float v = dz[i];
    if (training) {
        const float xh = (x[(size_t)r * ldx + c] - mean[c]) * is;
        v = v - sum_dz[c] / count - xh * sum_dz_xhat[c] / count;
    }
    dx[(size_t)r * lddx + c] = g * is * v;
}

// ------------------------------------------------------------------------------------------ CE with label smoothing
__device__ __forceinline__ void row_softmax_stats(const float* __restrict__ z, int C, int lane, float& mx, float& se, float& sz) {
    mx = -INFINITY; sz = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = z[c]; mx = fmaxf(mx, v); sz += v; }
    mx = wave_max(mx); sz = wave_sum(sz);
    se = 0.f;
    for (int c = lane; c < C; c += 64) se += __expf(z[c] - mx);
    se = wave_sum(se);
}

__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ labels,
                                                     const uint8_t* __restrict__ valid, int rows, int C, float eps,
                                                     float* __restrict__ row_loss, float* __restrict__ loss_sum) {
    REID_T16_ENTER();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long y = labels[row];
    const bool ok = (!valid || valid[row]) && y >= 0 && y < C;
    float loss = 0.f;
    if (ok) {
        const float* z = logits + (size_t)row * ld;
        float mx, se, sz;
        row_softmax_stats(z, C, lane, mx, se, sz);
        const float lse = mx + logf(se);
        loss = (1.f - eps) * (lse - z[y]) + eps * (lse - sz / C);
    }
    if (lane == 0) {
        if (row_loss) row_loss[row] = loss;
        if (ok) { atomicAdd(loss_sum, loss); atomicAdd(loss_sum + 1, 1.0f); }
    }
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ labels,
                                                     const uint8_t* __restrict__ valid, int rows, int C, float eps,
                                                     const float* __restrict__ grad_scale, float* __restrict__ dl, int lddl) {
    REID_T16_ENTER();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long y = labels[row];
    const bool ok = (!valid || valid[row]) && y >= 0 && y < C;
    float* d = dl + (size_t)row * lddl;
    if (!ok) {
        for (int c = lane; c < C; c += 64) d[c] = 0.f;
        return;
    }
    const float* z = logits + (size_t)row * ld;
    float mx, se, sz;
    row_softmax_stats(z, C, lane, mx, se, sz);
    const float gs = grad_scale[0], inv = 1.f / se;
    for (int c = lane; c < C; c += 64) {
        const float p = __expf(z[c] - mx) * inv;
        d[c] = gs * (p - eps / C - (c == y ? 1.f - eps : 0.f));
    }
}

// y = x / max(||x||, eps): dx (+)= (dy - y (y.dy)) / max(||x||, eps)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                                                         float* __restrict__ dx, int lddx, int rows, int D, float eps,
                                                         int accumulate) {
    REID_T16_ENTER();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = D >> 2;
    f32x4 xv[MAXV], gv[MAXV];
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        xv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; gv[i] = xv[i];
        if (c < nv) {
            xv[i] = *(const f32x4*)(x + (size_t)row * ldx + c * 4);
            gv[i] = *(const f32x4*)(dy + (size_t)row * lddy + c * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { ss += xv[i][e] * xv[i][e]; dot += xv[i][e] * gv[i][e]; }
        }
    }
    const float n = fmaxf(sqrtf(wave_sum(ss)), eps);
    const float rn = 1.f / n;
    dot = wave_sum(dot) * rn * rn;      // (y.dy)/n with y = x/n  ->  x.dy / n^2
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            float* d = dx + (size_t)row * lddx + c * 4;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (gv[i][e] - xv[i][e] * dot) * rn;
            if (accumulate) o += *(const f32x4*)d;
            *(f32x4*)d = o;
        }
    }
}

// ------------------------------------------------------------------------------------------ small fp32 helpers of the head
// (SDM module models/model.py:57-77 and FeatureFusion :113-183 work on [B,512] / [B,5,512] tensors: latency bound)
__global__ void eltwise_kernel(int op, const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, long n, float alpha) {
    REID_T16_ENTER();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float a = x[i];
        float r;
        switch (op) {
            case 0: r = a + alpha * y[i]; break;                       // add
            case 1: r = fmaxf(a, 0.f); break;                          // relu
            case 2: r = a > 0.f ? y[i] : 0.f; break;                   // relu backward: x = pre-activation, y = dy
            case 3: r = gelu_erf_f(a); break;                          // gelu
            case 4: r = y[i] * dgelu_erf_f(a); break;                  // gelu backward
            case 5: r = a * y[i]; break;                               // mul
            case 7: r = a >= alpha ? 1.0f / (1.0f - alpha) : 0.f; break;   // dropout multiplier from a uniform draw: keep iff u >= p
            default: r = isfinite(a) ? a : (a != a ? 0.f : (a > 0.f ? 1e4f : -1e4f)); break;   // 6: nan_to_num(0, 1e4, -1e4)
        }
        out[i] = r;
    }
}

// Attention over S <= 8 tokens, head_dim 64, fp32 (nn.MultiheadAttention of FeatureFusion, model.py:152-155).
// One wave per (sequence, head); lane = one of the 64 head dimensions; scores by wave reductions.
__global__ __launch_bounds__(256) void small_attn_fwd_kernel(const float* __restrict__ qkv, int ld, const uint8_t* __restrict__ key_mask,
                                                             const float* __restrict__ drop, float* __restrict__ out, int ldo,
                                                             float* __restrict__ probs, int n_seq, int S, int heads) {
    REID_T16_ENTER();
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_seq * heads) return;
    const int seq = item / heads, head = item % heads, d = heads * 64;
    float q[8], k[8], v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float* r = qkv + (size_t)(seq * S + (t < S ? t : 0)) * ld + head * 64 + lane;
        q[t] = r[0]; k[t] = r[d]; v[t] = r[2 * d];
    }
    float* pr = probs + (size_t)item * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i >= S) break;
        float sc[8], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float sj = wave_sum(q[i] * k[j]) * 0.125f;
            const bool ok = j < S && (!key_mask || key_mask[seq * S + j] != 0);
            sc[j] = ok ? sj : -INFINITY;
            mx = fmaxf(mx, sc[j]);
        }
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = __expf(sc[j] - mx); den += sc[j]; }
        float o = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[j] /= den;
            if (lane == j) pr[i * 8 + j] = sc[j];                                   // saved: the softmax output, before dropout
            o += sc[j] * (drop ? drop[(size_t)item * 64 + i * 8 + j] : 1.f) * v[j];
        }
        out[(size_t)(seq * S + i) * ldo + head * 64 + lane] = o;
    }
}

__global__ __launch_bounds__(256) void small_attn_bwd_kernel(const float* __restrict__ qkv, int ld, const float* __restrict__ probs,
                                                             const float* __restrict__ drop, const float* __restrict__ dout, int ldo,
                                                             float* __restrict__ dqkv, int lddq, int n_seq, int S, int heads) {
    REID_T16_ENTER();
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_seq * heads) return;
    const int seq = item / heads, head = item % heads, d = heads * 64;
    float q[8], k[8], v[8], go[8], dq[8], dk[8], dv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const size_t row = (size_t)(seq * S + (t < S ? t : 0));
        const float* r = qkv + row * ld + head * 64 + lane;
        q[t] = r[0]; k[t] = r[d]; v[t] = r[2 * d];
        go[t] = t < S ? dout[row * ldo + head * 64 + lane] : 0.f;
        dq[t] = 0.f; dk[t] = 0.f; dv[t] = 0.f;
    }
    const float* pr = probs + (size_t)item * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i >= S) break;
        float p[8], dp[8], dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            p[j] = j < S ? pr[i * 8 + j] : 0.f;
            const float mj = drop ? drop[(size_t)item * 64 + i * 8 + j] : 1.f;     // out = sum_j (p_j m_j) v_j
            dp[j] = wave_sum(go[i] * v[j]) * mj;
            dot += p[j] * dp[j];
            dv[j] += p[j] * mj * go[i];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float ds = p[j] * (dp[j] - dot) * 0.125f;
            dq[i] += ds * k[j];
            dk[j] += ds * q[i];
        }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (t < S) {
            float* r = dqkv + (size_t)(seq * S + t) * lddq + head * 64 + lane;
            r[0] = dq[t]; r[d] = dk[t]; r[2 * d] = dv[t];
        }
}

// out[b,:] = sum_m mask[b,m] x[b,m,:] / max(sum_m mask[b,m], 1)   (model.py:168-178);  bwd: dx[b,m,:] = mask[b,m] dout[b,:] / cnt
// One workgroup per (b, 64 columns): thread = (column, one of 4 interleaved m-lanes), partials combined through LDS in a fixed
// order.  M is 5 (fusion slots) or B*M (the global mean of the valid rows for all-masked samples, model.py:141-149: M = 320).
__global__ __launch_bounds__(256) void masked_mean_kernel(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ out,
                                                          int B, int M, int D, int bwd) {
    REID_T16_ENTER();
    __shared__ float pc[4][64], ps[4][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, ml = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool cok = c < D;
    const float* mk = mask + (size_t)b * M;
    float cnt = 0.f, s = 0.f;
    for (int m = ml; m < M; m += 4) {
        const float w = mk[m];
        cnt += w;
        if (!bwd && cok) s += w * x[((size_t)b * M + m) * D + c];
    }
    pc[ml][cl] = cnt; ps[ml][cl] = s;
    __syncthreads();
    cnt = fmaxf((pc[0][cl] + pc[1][cl]) + (pc[2][cl] + pc[3][cl]), 1.f);
    if (!cok) return;
    if (!bwd) {
        if (ml == 0) out[(size_t)b * D + c] = ((ps[0][cl] + ps[1][cl]) + (ps[2][cl] + ps[3][cl])) / cnt;
    } else {
        const float g = x[(size_t)b * D + c] / cnt;      // x = dout [B, D]
        for (int m = ml; m < M; m += 4) out[((size_t)b * M + m) * D + c] = mk[m] * g;
    }
}

int launch_sgemm(const float* A, const float* B, float* C, int M, int N, int K, long sam, long sak, long sbk, long sbn, int ldc,
                 float alpha, float beta, const float* bias, int act, hipStream_t s) {
    // K split over 8 waves while the grid is small (pure latency), over 4 once there are more tiles than CUs
    const dim3 grid((N + 31) / 32, (M + 31) / 32);
    if (grid.x * grid.y <= 256)
        hipLaunchKernelGGL(sgemm_kernel<8>, grid, dim3(512), 0, s, A, B, C, M, N, K, sam, sak, sbk, sbn, ldc, alpha, beta, bias, act);
    else
        hipLaunchKernelGGL(sgemm_kernel<4>, grid, dim3(256), 0, s, A, B, C, M, N, K, sam, sak, sbk, sbn, ldc, alpha, beta, bias, act);
    REID_CHECK_LAUNCH("reid_sgemm");
    return REID_OK;
}

}  // namespace

extern "C" int reid_sgemm(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, int64_t sam, int64_t sak,
                          int64_t sbk, int64_t sbn, int32_t ldc, float alpha, float beta, const float* bias, int32_t act,
                          void* stream) {
    REID_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && ldc >= N, "reid_sgemm: bad args");
    REID_CHECK_ARG(act >= 0 && act <= REID_ACT_RELU, "reid_sgemm: act=%d", act);
    return launch_sgemm(A, B, C, M, N, K, sam, sak, sbk, sbn, ldc, alpha, beta, bias, act, (hipStream_t)stream);
}

extern "C" int reid_bnneck_stats(const float* x, int32_t ldx, int32_t rows, int32_t D, float* sum, float* sqsum, void* stream) {
    REID_CHECK_ARG(x && sum && sqsum && rows > 0 && D > 0, "reid_bnneck_stats: bad args");
    hipStream_t s = (hipStream_t)stream;
    REID_CHECK_HIP(hipMemsetAsync(sum, 0, D * sizeof(float), s), "hipMemsetAsync");
    REID_CHECK_HIP(hipMemsetAsync(sqsum, 0, D * sizeof(float), s), "hipMemsetAsync");
    int splits = (rows + 63) / 64; if (splits > 64) splits = 64;
    const int rpb = (rows + splits - 1) / splits;
    hipLaunchKernelGGL(bn_stats_kernel, dim3((D + 63) / 64, splits), dim3(256), 0, s, x, ldx, rows, D, rpb, sum, sqsum);
    REID_CHECK_LAUNCH("reid_bnneck_stats");
    return REID_OK;
}

extern "C" int reid_bnneck_fwd(const float* x, int32_t ldx, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, const float* sum, const float* sqsum, float count, int32_t training, float* y,
                               void* y_bf16, int32_t ldy, float* mean, float* invstd, float* rnorm, int32_t rows, int32_t D,
                               float eps, float momentum, float scale, void* stream) {
    REID_CHECK_ARG(x && gamma && beta && y && mean && invstd && rnorm, "reid_bnneck_fwd: null pointer");
    REID_CHECK_ARG(rows > 0 && D % 4 == 0 && D <= 64 * 4 * MAXV && ldx % 4 == 0 && ldy % 4 == 0, "reid_bnneck_fwd: shape");
    REID_CHECK_ARG(training ? (sum && sqsum && count > 0) : (running_mean && running_var), "reid_bnneck_fwd: statistics missing");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((D + 255) / 256), dim3(256), 0, s, sum, sqsum, count, training, running_mean,
                       running_var, mean, invstd, D, eps, momentum);
    REID_CHECK_LAUNCH("reid_bnneck_fwd(finalize)");
    hipLaunchKernelGGL(bnneck_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, gamma, beta, mean, invstd, y, (bf16_t*)y_bf16,
                       ldy, rnorm, rows, D, scale);
    REID_CHECK_LAUNCH("reid_bnneck_fwd");
    return REID_OK;
}

extern "C" int reid_bnneck_bwd_p1(const float* dy, int32_t lddy, const float* x, int32_t ldx, const float* gamma, const float* beta,
                                  const float* mean, const float* invstd, const float* rnorm, float* dz, float* sum_dz,
                                  float* sum_dz_xhat, int32_t rows, int32_t D, float scale, void* stream) {
    REID_CHECK_ARG(dy && x && gamma && beta && mean && invstd && rnorm && dz && sum_dz && sum_dz_xhat, "reid_bnneck_bwd_p1: null pointer");
    REID_CHECK_ARG(rows > 0 && D % 4 == 0 && D <= 64 * 4 * MAXV, "reid_bnneck_bwd_p1: shape");
    hipStream_t s = (hipStream_t)stream;
    REID_CHECK_HIP(hipMemsetAsync(sum_dz, 0, D * sizeof(float), s), "hipMemsetAsync");
    REID_CHECK_HIP(hipMemsetAsync(sum_dz_xhat, 0, D * sizeof(float), s), "hipMemsetAsync");
    hipLaunchKernelGGL(bnneck_bwd1_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, dy, lddy, x, ldx, gamma, beta, mean, invstd, rnorm, dz,
                       sum_dz, sum_dz_xhat, rows, D, scale);
    REID_CHECK_LAUNCH("reid_bnneck_bwd_p1");
    return REID_OK;
}

extern "C" int reid_bnneck_bwd_p2(const float* dz, const float* x, int32_t ldx, const float* gamma, const float* mean,
                                  const float* invstd, const float* sum_dz, const float* sum_dz_xhat, float count, int32_t training,
                                  float* dx, int32_t lddx, int32_t rows, int32_t D, void* stream) {
    REID_CHECK_ARG(dz && x && gamma && mean && invstd && sum_dz && sum_dz_xhat && dx && rows > 0 && count > 0, "reid_bnneck_bwd_p2: bad args");
    const long n = (long)rows * D;
    hipLaunchKernelGGL(bnneck_bwd2_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dz, x, ldx, gamma, mean, invstd,
                       sum_dz, sum_dz_xhat, count, training, dx, lddx, rows, D);
    REID_CHECK_LAUNCH("reid_bnneck_bwd_p2");
    return REID_OK;
}

extern "C" int reid_ce_ls_fwd(const float* logits, int32_t ld, const int64_t* labels, const uint8_t* valid, int32_t rows, int32_t C,
                              float smoothing, float* row_loss, float* loss_sum, void* stream) {
    REID_CHECK_ARG(logits && labels && loss_sum && rows > 0 && C > 0 && ld >= C, "reid_ce_ls_fwd: bad args");
    hipLaunchKernelGGL(ce_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, valid, rows, C, smoothing,
                       row_loss, loss_sum);
    REID_CHECK_LAUNCH("reid_ce_ls_fwd");
    return REID_OK;
}

extern "C" int reid_ce_ls_bwd(const float* logits, int32_t ld, const int64_t* labels, const uint8_t* valid, int32_t rows, int32_t C,
                              float smoothing, const float* grad_scale, float* dlogits, int32_t lddl, void* stream) {
    REID_CHECK_ARG(logits && labels && grad_scale && dlogits && rows > 0 && C > 0 && ld >= C && lddl >= C, "reid_ce_ls_bwd: bad args");
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, valid, rows, C, smoothing,
                       grad_scale, dlogits, lddl);
    REID_CHECK_LAUNCH("reid_ce_ls_bwd");
    return REID_OK;
}

extern "C" int reid_eltwise_f32(int32_t op, const float* x, const float* y, float* out, int64_t n, float alpha, void* stream) {
    REID_CHECK_ARG(x && out && n > 0 && op >= 0 && op <= 7, "reid_eltwise_f32: bad args");
    REID_CHECK_ARG(y || op == 1 || op == 3 || op == 6 || op == 7, "reid_eltwise_f32: op %d needs a second operand", op);
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(eltwise_kernel, dim3((int)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, op, x, y, out, (long)n, alpha);
    REID_CHECK_LAUNCH("reid_eltwise_f32");
    return REID_OK;
}

extern "C" int reid_small_attn_fwd(const float* qkv, int32_t ld, const uint8_t* key_mask, const float* drop, float* out, int32_t ldo, float* probs,
                                   int32_t n_seq, int32_t S, int32_t heads, void* stream) {
    REID_CHECK_ARG(qkv && out && probs && n_seq > 0 && S >= 1 && S <= 8 && heads > 0 && ld >= 3 * heads * 64 && ldo >= heads * 64,
                   "reid_small_attn_fwd: bad args (S=%d must be 1..8)", S);
    hipLaunchKernelGGL(small_attn_fwd_kernel, dim3((n_seq * heads + 3) / 4), dim3(256), 0, (hipStream_t)stream, qkv, ld, key_mask, drop, out, ldo,
                       probs, n_seq, S, heads);
    REID_CHECK_LAUNCH("reid_small_attn_fwd");
    return REID_OK;
}

extern "C" int reid_small_attn_bwd(const float* qkv, int32_t ld, const float* probs, const float* drop, const float* dout, int32_t ldo, float* dqkv,
                                   int32_t lddqkv, int32_t n_seq, int32_t S, int32_t heads, void* stream) {
    REID_CHECK_ARG(qkv && probs && dout && dqkv && n_seq > 0 && S >= 1 && S <= 8 && heads > 0, "reid_small_attn_bwd: bad args");
    hipLaunchKernelGGL(small_attn_bwd_kernel, dim3((n_seq * heads + 3) / 4), dim3(256), 0, (hipStream_t)stream, qkv, ld, probs, drop, dout, ldo,
                       dqkv, lddqkv, n_seq, S, heads);
    REID_CHECK_LAUNCH("reid_small_attn_bwd");
    return REID_OK;
}

extern "C" int reid_masked_mean(const float* x, const float* mask, float* out, int32_t B, int32_t M, int32_t D, int32_t backward, void* stream) {
    REID_CHECK_ARG(x && mask && out && B > 0 && M > 0 && D > 0, "reid_masked_mean: bad args");
    REID_CHECK_ARG(B <= 65535, "reid_masked_mean: B");
    hipLaunchKernelGGL(masked_mean_kernel, dim3((D + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x, mask, out, B, M, D, backward);
    REID_CHECK_LAUNCH("reid_masked_mean");
    return REID_OK;
}
