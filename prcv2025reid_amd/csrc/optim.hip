// Optimizer step of the training driver on gfx950 (reference: train.py:85-96 _sanitize_grads, :975-1047 clip + step,
// torch.optim.AdamW as configured at train.py:1460).
//
// The reference walks ~1.2 k parameter tensors on the host (one `.item()` per tensor for the gradient norm, one
// boolean-mask kernel per tensor for the sanitiser).  Here the trainable set is a handful of flat fp32 buffers (the
// LoRA arena is ONE tensor), described by a device-resident table, and a step is three launches with no host
// synchronisation:
//   1. reid_opt_sumsq   non-finite gradient entries -> 0 (in place) and per-workgroup partial sums of g^2
//   2. reid_opt_clip    fixed-order (double) sum of the partials -> ||g||, adaptive / fixed max-norm, clip coefficient;
//                       the adaptive rule's 10-entry history and its 70th percentile live in the device state
//   3. reid_opt_adamw   AdamW with the clip coefficient folded into the gradient read; optionally zeroes the gradient
// All three are HBM-bound streaming kernels: 16-byte accesses, grid-stride over each table entry.
#include "common.h"

namespace {

struct OptEntry {            // 48 bytes, mirrored by prcv2025reid_amd/trainer.py
    float* p; float* g; float* m; float* v;
    long long n;
    float lr, wd;
};

constexpr int OPT_BLOCKS = 128;     // workgroups per table entry (grid = OPT_BLOCKS x n_entries)
constexpr int ST_SUMSQ = 0, ST_NORM = 1, ST_COEF = 2, ST_MAXNORM = 3, ST_BAD = 4, ST_HCOUNT = 5, ST_HIST = 6;   // + 10 floats
constexpr int ST_STEP = 16, ST_BC1 = 17, ST_BC2S = 18, ST_BATCH = 19, ST_FLOATS = 20;   // device-side counters (graph replay)

__device__ __forceinline__ bool finite_f(float x) { return (__builtin_bit_cast(uint32_t, x) & 0x7f800000u) != 0x7f800000u; }

__global__ __launch_bounds__(256) void opt_sumsq_kernel(const OptEntry* __restrict__ table, float* __restrict__ partial,
                                                        float* __restrict__ bad_partial) {
    const OptEntry e = table[blockIdx.y];
    float s = 0.f, bad = 0.f;
    if (e.g) {
        const long long n4 = e.n >> 2;
        f32x4* g4 = (f32x4*)e.g;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            f32x4 v = g4[i];
            bool any_bad = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!finite_f(v[k])) { v[k] = 0.f; any_bad = true; bad += 1.f; }
                s = fmaf(v[k], v[k], s);
            }
            if (any_bad) g4[i] = v;
        }
        if (blockIdx.x == 0) {
            for (long long i = (n4 << 2) + threadIdx.x; i < e.n; i += 256) {
                float v = e.g[i];
                if (!finite_f(v)) { v = 0.f; e.g[i] = 0.f; bad += 1.f; }
                s = fmaf(v, v, s);
            }
        }
    }
    __shared__ float red[2][4];
    s = wave_sum(s); bad = wave_sum(bad);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int o = blockIdx.y * gridDim.x + blockIdx.x;
        partial[o] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        bad_partial[o] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// One wave: the reduction order is fixed, so the norm (and everything derived from it) is bit-reproducible.
__global__ void opt_clip_kernel(const float* __restrict__ partial, const float* __restrict__ bad_partial, int n_partial,
                                float* __restrict__ state, int adaptive, float fixed_max_norm, int record) {
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    // one wave, fixed order: lane l adds partial[l], partial[l + 64], ... then a fixed shuffle tree (bit-reproducible, and
    // 64 independent load streams instead of one dependent chain: 63 -> ~5 us on the step's critical path)
    double s = 0.0, bad = 0.0;
    for (int i = threadIdx.x; i < n_partial; i += 64) { s += (double)partial[i]; bad += (double)bad_partial[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); bad += __shfl_xor(bad, o, 64); }
    if (threadIdx.x != 0) return;
    const double norm = sqrt(s);
    float max_norm = fixed_max_norm;
    if (record < 0) {                                        // decided on the device: batch_idx % (-record) == 0 (HIP-graph replay)
        const int bi = (int)state[ST_BATCH];
        state[ST_BATCH] = (float)(bi + 1);
        record = (bi % (-record)) == 0;
    }
    if (adaptive) {
        float cnt = state[ST_HCOUNT];
        if (record) {                                      // grad_norms.append(total_norm): ring of the last 10
            const int c = (int)cnt;
            state[ST_HIST + (c % 10)] = (float)norm;
            cnt += 1.f;
            state[ST_HCOUNT] = cnt;
        }
        if (cnt > 10.f) {                                  // np.percentile(grad_norms[-10:], 70): sorted a, a[6] + 0.3 (a[7] - a[6])
            double a[10];
            for (int i = 0; i < 10; ++i) a[i] = (double)state[ST_HIST + i];
            for (int i = 1; i < 10; ++i) {
                const double x = a[i];
                int j = i - 1;
                while (j >= 0 && a[j] > x) { a[j + 1] = a[j]; --j; }
                a[j + 1] = x;
            }
            const double p70 = a[6] + (a[7] - a[6]) * (0.7 * 9.0 - 6.0);
            const double m = fmin(3.0, fmax(0.5, p70 * 1.15));
            max_norm = (float)m;
        } else {
            max_norm = 1.0f;
        }
    }
    // torch.nn.utils.clip_grad_norm_: coef = clamp(max_norm / (total_norm + 1e-6), max = 1)
    const float nf = (float)norm;
    float coef = max_norm / (nf + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
    state[ST_SUMSQ] = (float)s; state[ST_NORM] = nf; state[ST_COEF] = coef; state[ST_MAXNORM] = max_norm; state[ST_BAD] = (float)bad;
}

struct AdamHp { float beta1, beta2, eps, step_bc1, bc2_sqrt; int zero_grad; const float* dev_bc; };

// step counter and bias corrections kept on the device (a captured HIP graph replays the same kernel arguments every step)
__global__ void opt_tick_kernel(float* __restrict__ state, float beta1, float beta2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double t = (double)state[ST_STEP] + 1.0;
    state[ST_STEP] = (float)t;
    state[ST_BC1] = (float)(1.0 - pow((double)beta1, t));
    state[ST_BC2S] = (float)sqrt(1.0 - pow((double)beta2, t));
}

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, float lr, float wd, const AdamHp& h) {
    p *= 1.f - lr * wd;                                     // decoupled weight decay
    m += (g - m) * (1.f - h.beta1);                         // exp_avg.lerp_(grad, 1 - beta1)
    v = v * h.beta2 + (1.f - h.beta2) * g * g;              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
    p -= (lr / h.step_bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void opt_adamw_kernel(const OptEntry* __restrict__ table, const float* __restrict__ coef_ptr,
                                                        AdamHp h) {
    const OptEntry e = table[blockIdx.y];
    if (!e.g) return;
    if (h.dev_bc) { h.step_bc1 = h.dev_bc[0]; h.bc2_sqrt = h.dev_bc[1]; }
    const float coef = coef_ptr ? *coef_ptr : 1.f;
    const long long n4 = e.n >> 2;
    f32x4 *p4 = (f32x4*)e.p, *g4 = (f32x4*)e.g, *m4 = (f32x4*)e.m, *v4 = (f32x4*)e.v;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 p = p4[i], g = g4[i], m = m4[i], v = v4[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = p[k], mk = m[k], vk = v[k];
            adamw_one(pk, g[k] * coef, mk, vk, e.lr, e.wd, h);
            p[k] = pk; m[k] = mk; v[k] = vk;
        }
        p4[i] = p; m4[i] = m; v4[i] = v;
        if (h.zero_grad) g4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (blockIdx.x == 0) {
        for (long long i = (n4 << 2) + threadIdx.x; i < e.n; i += 256) {
            float p = e.p[i], m = e.m[i], v = e.v[i];
            adamw_one(p, e.g[i] * coef, m, v, e.lr, e.wd, h);
            e.p[i] = p; e.m[i] = m; e.v[i] = v;
            if (h.zero_grad) e.g[i] = 0.f;
        }
    }
}

}  // namespace

extern "C" int32_t reid_opt_entry_bytes(void) { return (int32_t)sizeof(OptEntry); }
extern "C" int32_t reid_opt_ws_floats(int32_t n_entries) { return 2 * OPT_BLOCKS * n_entries; }
extern "C" int32_t reid_opt_state_floats(void) { return ST_FLOATS; }

extern "C" int reid_opt_sumsq(const void* table, int32_t n_entries, float* ws, void* stream) {
    REID_CHECK_ARG(table && ws && n_entries > 0, "reid_opt_sumsq: bad args");
    hipLaunchKernelGGL(opt_sumsq_kernel, dim3(OPT_BLOCKS, n_entries), dim3(256), 0, (hipStream_t)stream, (const OptEntry*)table, ws,
                       ws + OPT_BLOCKS * n_entries);
    REID_CHECK_LAUNCH("reid_opt_sumsq");
    return REID_OK;
}

extern "C" int reid_opt_clip(const float* ws, int32_t n_entries, float* state, int32_t adaptive, float fixed_max_norm,
                             int32_t record, void* stream) {
    REID_CHECK_ARG(ws && state && n_entries > 0, "reid_opt_clip: bad args");
    REID_CHECK_ARG(fixed_max_norm > 0.f, "reid_opt_clip: max_norm must be positive");
    hipLaunchKernelGGL(opt_clip_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, ws + OPT_BLOCKS * n_entries,
                       OPT_BLOCKS * n_entries, state, adaptive, fixed_max_norm, record);
    REID_CHECK_LAUNCH("reid_opt_clip");
    return REID_OK;
}

extern "C" int reid_opt_adamw(const void* table, int32_t n_entries, const float* coef, float beta1, float beta2, float eps,
                              int32_t step, int32_t zero_grad, float* state, void* stream) {
    REID_CHECK_ARG(table && n_entries > 0 && (step >= 1 || (step == 0 && state)), "reid_opt_adamw: bad args (step counts from 1; 0 = device counter in state)");
    REID_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f, "reid_opt_adamw: betas/eps");
    AdamHp h;
    h.beta1 = beta1; h.beta2 = beta2; h.eps = eps;
    h.zero_grad = zero_grad;
    h.dev_bc = nullptr;
    if (step >= 1) {
        h.step_bc1 = (float)(1.0 - pow((double)beta1, (double)step));
        h.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    } else {
        h.step_bc1 = h.bc2_sqrt = 1.f;
        hipLaunchKernelGGL(opt_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, beta1, beta2);
        REID_CHECK_LAUNCH("reid_opt_adamw(tick)");
        h.dev_bc = state + ST_BC1;
    }
    hipLaunchKernelGGL(opt_adamw_kernel, dim3(OPT_BLOCKS, n_entries), dim3(256), 0, (hipStream_t)stream, (const OptEntry*)table, coef, h);
    REID_CHECK_LAUNCH("reid_opt_adamw");
    return REID_OK;
}
