// Row-wise HBM-bound kernels for gfx950: LayerNorm fwd/bwd, patch extraction, CLS rows, casts,
// row gather, L2 normalisation.  One 64-lane wavefront owns one row; every global access is a
// 16-byte (float4 / 8 x bf16) coalesced load or store; reductions are wave shuffles only.
#include "common.h"
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>

// ---------------------------------------------------------------- error plumbing (shared by all files)
static thread_local char g_err[512] = "";
void reid_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* reid_last_error(void) { return g_err; }
extern "C" int reid_version(void) { return 200; }

// ---------------------------------------------------------------- experiment knobs (common.h)
static const char* const g_knob_names[] = {
    "GEMM_TILE", "GEMM_DBG", "GEMM_GROUPM", "GEMM_EPI", "GEMM_STAGGER",
    "ATTN_DBG", "TN_BLOCKS", "TOPK_DBG", "TOPK_TILE", "STREAM_ROWS", "STREAM_GROUPS", "SDM_IMPL", "SKINNY_TILE", "GEMM_PERSIST",
    "ATTN_BWD", "LORA_IMPL", "GELU_IMPL", "HEAD_IMPL", "STREAM_FUSE", "TOPK_SCAN", "LN_IMPL"};
static_assert(sizeof(g_knob_names) / sizeof(g_knob_names[0]) == KNOB_COUNT, "one name per reid_knob_id");
static int g_knobs[KNOB_COUNT];
static bool knob_is_debug(int i) { return i == KNOB_GEMM_DBG || i == KNOB_ATTN_DBG || i == KNOB_TOPK_DBG; }
static int* knob_table() {
    static const bool init = [] {
        for (int i = 0; i < KNOB_COUNT; ++i) {
            char name[64];
            snprintf(name, sizeof(name), "REID_%s", g_knob_names[i]);
            const char* e = getenv(name);
            g_knobs[i] = e ? atoi(e) : -1;          // -1 = "not set": every reader has its own default
            if (knob_is_debug(i)) g_knobs[i] = -1;  // wrong-result modes: experiment builds only, and only through reid_set_knob()
        }
        return true;
    }();
    (void)init;
    return g_knobs;
}
int reid_knob(int id) { return knob_table()[id]; }
extern "C" int reid_set_knob(const char* name, int value) {
    int* t = knob_table();
    for (int i = 0; i < KNOB_COUNT; ++i)
        if (strcmp(name, g_knob_names[i]) == 0) {
#ifndef REID_EXPERIMENTS
            if (knob_is_debug(i) && value > 0) {
                reid_set_error("reid_set_knob: %s is a wrong-result timing mode, available only in -DREID_EXPERIMENTS builds", name);
                return REID_ERR_ARG;
            }
#endif
            t[i] = value; return REID_OK;
        }
    reid_set_error("reid_set_knob: unknown knob %s", name);
    return REID_ERR_ARG;
}
int reid_num_cus() {
    static const int n = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }();
    return n;
}
extern "C" int reid_flavor(void) { return REID_FLAVOR_ID; }
extern "C" int reid_check_device(int dev) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        reid_set_error("reid_check_device: hipGetDeviceProperties(%d) failed", dev);
        return REID_ERR_DEVICE;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        reid_set_error("reid_check_device: device %d is %s, this library is built for gfx950 only", dev, prop.gcnArchName);
        return REID_ERR_DEVICE;
    }
    return REID_OK;
}

namespace {

constexpr int MAXV = 4;   // float4 vectors per lane: cols <= 64*4*4 = 1024

// ---------------------------------------------------------------- LayerNorm forward
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int ldx, const int32_t* __restrict__ row_index,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     bf16_t* __restrict__ yb, float* __restrict__ yf, int ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                     int cols, float eps) {
    REID_T16_ENTER();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const size_t src = row_index ? (size_t)row_index[row] : (size_t)row;
    const float* xr = x + src * ldx;
    const int nv = cols >> 2;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        v[i] = c < nv ? *(const f32x4*)(xr + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mu = wave_sum(s) / cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / cols + eps);
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const f32x4 g = *(const f32x4*)(gamma + c * 4);
            const f32x4 b = *(const f32x4*)(beta + c * 4);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mu) * rs * g[e] + b[e];
            if (yf) *(f32x4*)(yf + (size_t)row * ldy + c * 4) = o;
            if (yb) *(uint2*)(yb + (size_t)row * ldy + c * 4) = uint2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        }
    }
}

// ---------------------------------------------------------------- residual add + LayerNorm forward
// x_out = x + scale[row / rows_per_img] * y (y = the 16-bit output of the out-projection / fc2 GEMM), h = LayerNorm(x_out).
// Why: with the residual in the GEMM epilogue every tile read and wrote 4 bytes per element of fp32 while its matrix pipe idled
// (all 256 workgroups reach their epilogues together: 117 MB per round, ~23 us at HBM speed -- the out-projection ran at 21 % of the
// MFMA peak, r02 verdict); here the GEMM stores 2 bytes per element and this kernel, which streams x anyway, does the add.  Same
// bytes in total (GEMM 1.5 + this 9 KB per row against 6 + 4.5), but all of them at streaming speed and none under an idle MFMA.
// Y_HALF: y is IEEE half whatever the flavor (REID_F16: r04, the branch output is not an MFMA operand, so it carries 11 significant
// bits instead of bf16's 8 in the bf16 flavor as well).
template <bool Y_HALF>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const float* __restrict__ x, int ldx, const bf16_t* __restrict__ y, int ldy,
                                                         const float* __restrict__ row_scale, int rows_per_img,
                                                         float* __restrict__ xo, int ldxo, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, bf16_t* __restrict__ hb, int ldh,
                                                         float* __restrict__ mean, float* __restrict__ rstd, int rows, int cols, float eps) {
    REID_T16_ENTER();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ldx;
    const bf16_t* yr = y + (size_t)row * ldy;
    const float sc = row_scale ? row_scale[row / rows_per_img] : 1.f;
    const int nv = cols >> 2;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const f32x4 a = *(const f32x4*)(xr + c * 4);
            const bf16x4 b = *(const bf16x4*)(yr + c * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] = fmaf(sc, Y_HALF ? f16_to_f32((unsigned short)b[e]) : bf16_to_f32((bf16_t)b[e]), a[e]);
            *(f32x4*)(xo + (size_t)row * ldxo + c * 4) = v[i];
        } else {
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mu = wave_sum(s) / cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / cols + eps);
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const f32x4 g = *(const f32x4*)(gamma + c * 4);
            const f32x4 b = *(const f32x4*)(beta + c * 4);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mu) * rs * g[e] + b[e];
            *(uint2*)(hb + (size_t)row * ldh + c * 4) = uint2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        }
    }
}

// ---------------------------------------------------------------- LayerNorm backward
// Rows are walked with a grid stride so that, when the affine gradients are wanted, every wave keeps its dgamma / dbeta
// partial sums in registers over all its rows and adds them to memory ONCE (one atomic per column per wave instead of one per
// element: 50k rows x 768 columns would otherwise serialise on 768 addresses).  Without dgamma/dbeta the launch has one
// wave per row and the loop body runs once.
// DX_HALF: the residual-stream gradient (dres read, dx written) in IEEE half instead of fp32: 12 instead of 16 bytes per element of
// this HBM-bound kernel.  The caller scales the loss so that the stream sits in half's range (engine.py: loss_scaling).
// VN: float4 vectors per lane (3 covers the towers' 768 columns: 78 -> 66 VGPRs = 7 waves per SIMD.  Measured r04, 50432 x 768, half dx:
// 88.8 us at 7 waves, 90-92 us at 8 (forced: 2 spills) and at 6 -- occupancy is not what holds this kernel at 5.2 TB/s)
template <bool DY_BF16, bool AFFINE, bool DX_HALF, int VN = MAXV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                     const int32_t* __restrict__ row_index, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const void* __restrict__ dres_, void* __restrict__ dx_,
                                                     bf16_t* __restrict__ dxb, int lddx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int rows, int cols,
                                                     const float* __restrict__ bscale, int rows_per_img) {
    REID_T16_ENTER();
    if (DX_HALF) REID_F16_SATURATE();
    const int lane = threadIdx.x & 63;
    const int nv = cols >> 2;
    constexpr bool affine = AFFINE;                       // (a separate instantiation: the accumulators cost the default path 5 % of its bandwidth)
    f32x4 ag[VN], ab[VN];
#pragma unroll
    for (int i = 0; i < VN; ++i) { ag[i] = f32x4{0.f, 0.f, 0.f, 0.f}; ab[i] = ag[i]; }
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        const size_t xrow = row_index ? (size_t)row_index[row] : (size_t)row;
        const float* xr = x + xrow * ldx;
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[VN], g[VN];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VN; ++i) {
            const int c = lane + i * 64;
            xh[i] = f32x4{0.f, 0.f, 0.f, 0.f}; g[i] = xh[i];
            if (c < nv) {
                const f32x4 xv = *(const f32x4*)(xr + c * 4);
                const f32x4 gm = *(const f32x4*)(gamma + c * 4);
                f32x4 d;
                if (DY_BF16) {
                    const bf16x4 t = *(const bf16x4*)((const bf16_t*)dy + (size_t)row * lddy + c * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = bf16_to_f32((bf16_t)t[e]);
                } else {
                    d = *(const f32x4*)((const float*)dy + (size_t)row * lddy + c * 4);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[i][e] = (xv[e] - mu) * rs;
                    if (affine) { ag[i][e] = fmaf(d[e], xh[i][e], ag[i][e]); ab[i][e] += d[e]; }
                    g[i][e] = d[e] * gm[e];
                    s1 += g[i][e];
                    s2 += g[i][e] * xh[i][e];
                }
            }
        }
        const float m1 = wave_sum(s1) / cols, m2 = wave_sum(s2) / cols;
        const float bs = bscale ? bscale[xrow / rows_per_img] : 1.f;
#pragma unroll
        for (int i = 0; i < VN; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (g[i][e] - m1 - xh[i][e] * m2);
                if (DX_HALF) {
                    if (dres_) {
                        const uint2 r = *(const uint2*)((const unsigned short*)dres_ + xrow * lddx + c * 4);
                        o[0] += f16_to_f32((unsigned short)(r.x & 0xffffu)); o[1] += f16_to_f32((unsigned short)(r.x >> 16));
                        o[2] += f16_to_f32((unsigned short)(r.y & 0xffffu)); o[3] += f16_to_f32((unsigned short)(r.y >> 16));
                    }
                    *(uint2*)((unsigned short*)dx_ + xrow * lddx + c * 4) = uint2{pack_f16x2(o[0], o[1]), pack_f16x2(o[2], o[3])};
                } else {
                    if (dres_) {
                        const f32x4 r = *(const f32x4*)((const float*)dres_ + xrow * lddx + c * 4);
                        o += r;
                    }
                    *(f32x4*)((float*)dx_ + xrow * lddx + c * 4) = o;
                }
                if (dxb) *(uint2*)(dxb + xrow * lddx + c * 4) = uint2{pack_bf16x2(o[0] * bs, o[1] * bs), pack_bf16x2(o[2] * bs, o[3] * bs)};
            }
        }
    }
    if (affine) {
#pragma unroll
        for (int i = 0; i < VN; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (dgamma) atomicAdd(dgamma + c * 4 + e, ag[i][e]);
                    if (dbeta) atomicAdd(dbeta + c * 4 + e, ab[i][e]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------- patch extraction (im2col, k = s = patch)
// one thread = 8 consecutive pixels of one patch row -> one 16-byte bf16 store
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int n_img, int H,
                                                     int W, int P, int cin) {
    REID_T16_ENTER();
    const int gw = W / P, gh = H / P;
    const int kc = cin * P * P;                    // columns of the patch matrix
    const int chunks_per_row = kc / 8;
    const long total = (long)n_img * gh * gw * chunks_per_row;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = i % chunks_per_row;
        const long prow = i / chunks_per_row;                 // patch-matrix row
        const int pw = prow % gw, ph = (prow / gw) % gh;
        const long b = prow / (gw * gh);
        const int k = ch * 8;                                  // column = c*P*P + py*P + px
        const int c = k / (P * P), py = (k / P) % P, px = k % P;
        const size_t base = ((size_t)b * 3) * H * W + (size_t)(ph * P + py) * W + pw * P + px;
        float v[8];
        if (cin == 3) {
            const float* s = img + base + (size_t)c * H * W;
            const f32x4 a = *(const f32x4*)s, bb = *(const f32x4*)(s + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = bb[e]; }
        } else {   // 3-channel input to a 1-channel embed: channel mean first (patch_embeds.py:63-65)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
            for (int cc = 0; cc < 3; ++cc) {
                const float* s = img + base + (size_t)cc * H * W;
                const f32x4 a = *(const f32x4*)s, bb = *(const f32x4*)(s + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += bb[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] / 3.0f;
        }
        uint4 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        *(uint4*)(out + (size_t)prow * kc + k) = o;
    }
}

__global__ void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos0, float* __restrict__ x, int ldx,
                                int n_img, int tokens, int cols) {
    REID_T16_ENTER();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nv = cols >> 2;
    if (i >= n_img * nv) return;
    const int b = i / nv, c = i % nv;
    const f32x4 a = *(const f32x4*)(cls + c * 4), p = *(const f32x4*)(pos0 + c * 4);
    *(f32x4*)(x + (size_t)b * tokens * ldx + c * 4) = a + p;
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, long n) {
    REID_T16_ENTER();
    const long n8 = n >> 3;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const f32x4 a = *(const f32x4*)(s + i * 8), b = *(const f32x4*)(s + i * 8 + 4);
        *(uint4*)(d + i * 8) = uint4{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) d[n8 * 8 + threadIdx.x] = f32_to_bf16(s[n8 * 8 + threadIdx.x]);
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ s, float* __restrict__ d, long n) {
    REID_T16_ENTER();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) d[i] = bf16_to_f32(s[i]);
}

__global__ void gather_rows_kernel(const float* __restrict__ src, int lds, const int32_t* __restrict__ index, float* __restrict__ dst,
                                   int ldd, int rows, int cols) {
    REID_T16_ENTER();
    const int nv = cols >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * nv) return;
    const int r = i / nv, c = i % nv;
    *(f32x4*)(dst + (size_t)r * ldd + c * 4) = *(const f32x4*)(src + (size_t)index[r] * lds + c * 4);
}

__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, bf16_t* __restrict__ yb,
                                                     int ldy, int rows, int D, float eps, float scale) {
    REID_T16_ENTER();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = D >> 2;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        v[i] = c < nv ? *(const f32x4*)(x + (size_t)row * ldx + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    }
    const float inv = scale / fmaxf(sqrtf(wave_sum(s)), eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const f32x4 o = v[i] * inv;
            if (y) *(f32x4*)(y + (size_t)row * ldy + c * 4) = o;
            if (yb) *(uint2*)(yb + (size_t)row * ldy + c * 4) = uint2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        }
    }
}

// The hot form of the towers' backward (16-bit cotangent in, residual-stream gradient in IEEE half, no affine gradients) with EIGHT
// consecutive columns per lane: every 16-bit stream then moves 16 bytes per lane and instruction instead of 8 (8-byte accesses run at
// 0.54-0.70x the 16-byte rate, MI355X_MICROARCH.md) -- four of this kernel's five streams are 16-bit.  cols % 8 == 0, cols <= 1024.
__global__ __launch_bounds__(256) void ln_bwd8_kernel(const bf16_t* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                      const int32_t* __restrict__ row_index, const float* __restrict__ gamma,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const unsigned short* __restrict__ dres, unsigned short* __restrict__ dx,
                                                      bf16_t* __restrict__ dxb, int lddx, int rows, int cols,
                                                      const float* __restrict__ bscale, int rows_per_img) {
    REID_T16_ENTER();
    REID_F16_SATURATE();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int n8 = cols >> 3;
    const size_t xrow = row_index ? (size_t)row_index[row] : (size_t)row;
    const float* xr = x + xrow * ldx;
    const float mu = mean[row], rs = rstd[row];
    float xh[2][8], g[2][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = lane + i * 64;
#pragma unroll
        for (int e = 0; e < 8; ++e) { xh[i][e] = 0.f; g[i][e] = 0.f; }
        if (c < n8) {
            const f32x4 x0 = *(const f32x4*)(xr + c * 8), x1 = *(const f32x4*)(xr + c * 8 + 4);
            const f32x4 g0 = *(const f32x4*)(gamma + c * 8), g1 = *(const f32x4*)(gamma + c * 8 + 4);
            const uint4 d = *(const uint4*)(dy + (size_t)row * lddy + c * 8);
            const uint32_t dw[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xv = e < 4 ? x0[e] : x1[e - 4], gm = e < 4 ? g0[e] : g1[e - 4];
                const float dv = bf16_to_f32((bf16_t)((dw[e >> 1] >> (16 * (e & 1))) & 0xffffu));
                xh[i][e] = (xv - mu) * rs;
                g[i][e] = dv * gm;
                s1 += g[i][e];
                s2 = fmaf(g[i][e], xh[i][e], s2);
            }
        }
    }
    const float m1 = wave_sum(s1) / cols, m2 = wave_sum(s2) / cols;
    const float bs = bscale ? bscale[xrow / rows_per_img] : 1.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = lane + i * 64;
        if (c < n8) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rs * (g[i][e] - m1 - xh[i][e] * m2);
            if (dres) {
                const uint4 r = *(const uint4*)(dres + xrow * lddx + c * 8);
                const uint32_t rw[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += f16_to_f32((unsigned short)((rw[e >> 1] >> (16 * (e & 1))) & 0xffffu));
            }
            *(uint4*)(dx + xrow * lddx + c * 8) = uint4{pack_f16x2(o[0], o[1]), pack_f16x2(o[2], o[3]), pack_f16x2(o[4], o[5]), pack_f16x2(o[6], o[7])};
            if (dxb)
                *(uint4*)(dxb + xrow * lddx + c * 8) = uint4{pack_bf16x2(o[0] * bs, o[1] * bs), pack_bf16x2(o[2] * bs, o[3] * bs),
                                                            pack_bf16x2(o[4] * bs, o[5] * bs), pack_bf16x2(o[6] * bs, o[7] * bs)};
        }
    }
}
}  // namespace


extern "C" int reid_layernorm_fwd(const float* x, int32_t ldx, const int32_t* row_index, const float* gamma, const float* beta,
                                  void* y_bf16, float* y_f32, int32_t ldy, float* mean, float* rstd, int32_t rows,
                                  int32_t cols, float eps, void* stream) {
    REID_CHECK_ARG(x && gamma && beta && (y_bf16 || y_f32), "reid_layernorm_fwd: null pointer");
    REID_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= 64 * 4 * MAXV, "reid_layernorm_fwd: cols=%d unsupported", cols);
    REID_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && ldx >= cols && ldy >= cols, "reid_layernorm_fwd: ld");
    hipLaunchKernelGGL(ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, row_index, gamma, beta,
                       (bf16_t*)y_bf16, y_f32, ldy, mean, rstd, rows, cols, eps);
    REID_CHECK_LAUNCH("reid_layernorm_fwd");
    return REID_OK;
}

extern "C" int reid_add_layernorm_fwd(const float* x, int32_t ldx, const void* y_bf16, int32_t y_dtype, int32_t ldy, const float* row_scale,
                                      int32_t rows_per_img, float* x_out, int32_t ldxo, const float* gamma, const float* beta,
                                      void* h_bf16, int32_t ldh, float* mean, float* rstd, int32_t rows, int32_t cols, float eps,
                                      void* stream) {
    REID_CHECK_ARG(x && y_bf16 && x_out && gamma && beta && h_bf16, "reid_add_layernorm_fwd: null pointer");
    REID_CHECK_ARG(!row_scale || rows_per_img > 0, "reid_add_layernorm_fwd: row_scale needs rows_per_img");
    REID_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= 64 * 4 * MAXV, "reid_add_layernorm_fwd: cols=%d unsupported", cols);
    REID_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && ldxo % 4 == 0 && ldh % 4 == 0 && ldx >= cols && ldy >= cols && ldxo >= cols && ldh >= cols,
                   "reid_add_layernorm_fwd: ld");
    REID_CHECK_ARG(y_dtype == REID_BF16 || y_dtype == REID_F16, "reid_add_layernorm_fwd: y_dtype=%d (REID_BF16 = flavor format, or REID_F16)", y_dtype);
    if (y_dtype == REID_F16 && REID_FLAVOR_ID == 0)
        hipLaunchKernelGGL(add_ln_fwd_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, (const bf16_t*)y_bf16, ldy, row_scale,
                           rows_per_img, x_out, ldxo, gamma, beta, (bf16_t*)h_bf16, ldh, mean, rstd, rows, cols, eps);
    else
        hipLaunchKernelGGL(add_ln_fwd_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, (const bf16_t*)y_bf16, ldy, row_scale,
                           rows_per_img, x_out, ldxo, gamma, beta, (bf16_t*)h_bf16, ldh, mean, rstd, rows, cols, eps);
    REID_CHECK_LAUNCH("reid_add_layernorm_fwd");
    return REID_OK;
}

extern "C" int reid_layernorm_bwd(const void* dy, int32_t dy_dtype, int32_t lddy, const float* x, int32_t ldx,
                                  const int32_t* row_index, const float* gamma, const float* mean, const float* rstd,
                                  const void* dres, void* dx, int32_t dx_dtype, void* dx_bf16, int32_t lddx, float* dgamma, float* dbeta,
                                  int32_t rows, int32_t cols, const float* bf16_row_scale, int32_t rows_per_img, void* stream) {
    REID_CHECK_ARG(dy && x && gamma && mean && rstd && dx, "reid_layernorm_bwd: null pointer");
    REID_CHECK_ARG(dx_dtype == REID_F32 || dx_dtype == REID_F16, "reid_layernorm_bwd: dx_dtype must be REID_F32 or REID_F16");
    REID_CHECK_ARG(!bf16_row_scale || rows_per_img > 0, "reid_layernorm_bwd: bf16_row_scale needs rows_per_img");
    REID_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= 64 * 4 * MAXV, "reid_layernorm_bwd: cols=%d unsupported", cols);
    REID_CHECK_ARG(lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0, "reid_layernorm_bwd: ld");
    int blocks = (rows + 3) / 4;
    if ((dgamma || dbeta) && blocks > 1024) blocks = 1024;    // grid-stride rows: per-wave partial sums, one atomic per column per wave
    dim3 g(blocks), b(256);
    hipStream_t s = (hipStream_t)stream;
    const bool aff = dgamma || dbeta;
    const bool hx = dx_dtype == REID_F16;
    if (hx && dy_dtype == REID_BF16 && !aff && cols % 8 == 0 && cols <= 1024 && lddy % 8 == 0 && lddx % 8 == 0 && reid_knob(KNOB_LN_IMPL) != 1) {     // REID_LN_IMPL=1: the four-column form
        hipLaunchKernelGGL(ln_bwd8_kernel, dim3((rows + 3) / 4), b, 0, s, (const bf16_t*)dy, lddy, x, ldx, row_index, gamma, mean, rstd,
                           (const unsigned short*)dres, (unsigned short*)dx, (bf16_t*)dx_bf16, lddx, rows, cols, bf16_row_scale, rows_per_img);
        REID_CHECK_LAUNCH("reid_layernorm_bwd");
        return REID_OK;
    }
#define REID_LN_BWD3(B16, AFF, HX, VN)                                                                                         \
    hipLaunchKernelGGL((ln_bwd_kernel<B16, AFF, HX, VN>), g, b, 0, s, dy, lddy, x, ldx, row_index, gamma, mean, rstd, dres, dx, \
                       (bf16_t*)dx_bf16, lddx, dgamma, dbeta, rows, cols, bf16_row_scale, rows_per_img)
#define REID_LN_BWD(B16, AFF)                                                                                                  \
    do { if (hx) { if (cols <= 768) REID_LN_BWD3(B16, AFF, true, 3); else REID_LN_BWD3(B16, AFF, true, MAXV); }                 \
         else { if (cols <= 768) REID_LN_BWD3(B16, AFF, false, 3); else REID_LN_BWD3(B16, AFF, false, MAXV); } } while (0)
    if (dy_dtype == REID_BF16) { if (aff) REID_LN_BWD(true, true); else REID_LN_BWD(true, false); }
    else { if (aff) REID_LN_BWD(false, true); else REID_LN_BWD(false, false); }
#undef REID_LN_BWD3
#undef REID_LN_BWD
    REID_CHECK_LAUNCH("reid_layernorm_bwd");
    return REID_OK;
}

extern "C" int reid_patch_im2col(const float* images, void* patches, int32_t n_img, int32_t H, int32_t W, int32_t patch,
                                 int32_t cin, void* stream) {
    REID_CHECK_ARG(images && patches && n_img > 0, "reid_patch_im2col: null/empty");
    REID_CHECK_ARG((cin == 1 || cin == 3) && patch % 8 == 0 && H % patch == 0 && W % patch == 0 && W % 4 == 0,
                   "reid_patch_im2col: unsupported geometry H=%d W=%d patch=%d cin=%d", H, W, patch, cin);
    const long total = (long)n_img * (H / patch) * (W / patch) * (cin * patch * patch / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(im2col_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, images, (bf16_t*)patches, n_img, H, W, patch, cin);
    REID_CHECK_LAUNCH("reid_patch_im2col");
    return REID_OK;
}

extern "C" int reid_cls_rows(const float* cls, const float* pos0, float* x, int32_t ldx, int32_t n_img, int32_t tokens,
                             int32_t cols, void* stream) {
    REID_CHECK_ARG(cls && pos0 && x && n_img > 0 && cols % 4 == 0 && ldx % 4 == 0, "reid_cls_rows: bad args");
    const int n = n_img * (cols / 4);
    hipLaunchKernelGGL(cls_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, cls, pos0, x, ldx, n_img, tokens, cols);
    REID_CHECK_LAUNCH("reid_cls_rows");
    return REID_OK;
}

namespace {
// out[index[r], :] += src[r, :]  (adjoint of the row gather; one wave per source row, fp32 atomics)
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src, int lds_, const int32_t* __restrict__ index,
                                                               float* __restrict__ out, int ldo, int rows, int cols, int out_rows) {
    REID_T16_ENTER();
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int t = index[r];
    if (t < 0 || t >= out_rows) return;
    for (int c = lane; c < cols; c += 64) atomicAdd(out + (size_t)t * ldo + c, src[(size_t)r * lds_ + c]);
}
}  // namespace

extern "C" int reid_scatter_add_rows_f32(const float* src, int32_t lds_, const int32_t* index, float* out, int32_t ldo,
                                         int32_t rows, int32_t cols, int32_t out_rows, void* stream) {
    REID_CHECK_ARG(src && index && out && rows > 0 && cols > 0 && out_rows > 0, "reid_scatter_add_rows_f32: bad args");
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, src, lds_, index, out, ldo,
                       rows, cols, out_rows);
    REID_CHECK_LAUNCH("reid_scatter_add_rows_f32");
    return REID_OK;
}

extern "C" int reid_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
    REID_CHECK_ARG(src && dst && n > 0, "reid_cast_f32_bf16: bad args");
    REID_CHECK_ARG(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0), "reid_cast_f32_bf16: pointers must be 16-byte aligned");
    const long blocks = ((n >> 3) + 255) / 256;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((int)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks))), dim3(256), 0,
                       (hipStream_t)stream, src, (bf16_t*)dst, (long)n);
    REID_CHECK_LAUNCH("reid_cast_f32_bf16");
    return REID_OK;
}

extern "C" int reid_cast_bf16_f32(const void* src, float* dst, int64_t n, void* stream) {
    REID_CHECK_ARG(src && dst && n > 0, "reid_cast_bf16_f32: bad args");
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((int)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, dst, (long)n);
    REID_CHECK_LAUNCH("reid_cast_bf16_f32");
    return REID_OK;
}

extern "C" int reid_gather_rows_f32(const float* src, int32_t lds, const int32_t* index, float* dst, int32_t ldd, int32_t rows,
                                    int32_t cols, void* stream) {
    REID_CHECK_ARG(src && index && dst && rows > 0 && cols % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0, "reid_gather_rows_f32: bad args");
    const long n = (long)rows * (cols / 4);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, lds, index, dst, ldd, rows, cols);
    REID_CHECK_LAUNCH("reid_gather_rows_f32");
    return REID_OK;
}

namespace {
// x[b*T + t, :] = tok[ids[b, t], :] + pos[t, :]   (HF CLIPTextEmbeddings, clip_backbone.py:307): one 16-byte piece per thread
__global__ __launch_bounds__(256) void embed_tokens_kernel(const float* __restrict__ tok, const float* __restrict__ pos, const int64_t* __restrict__ ids,
                                                          float* __restrict__ out, int rows, int T, int D, int vocab) {
    REID_T16_ENTER();
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int per = D >> 2;
    if (t >= (long)rows * per) return;
    const int r = (int)(t / per), c = (int)(t - (long)r * per) * 4;
    long id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const f32x4 a = *(const f32x4*)(tok + (size_t)id * D + c), b = *(const f32x4*)(pos + (size_t)(r % T) * D + c);
    *(f32x4*)(out + (size_t)r * D + c) = a + b;
}
}  // namespace
extern "C" int reid_embed_tokens(const float* tok, const float* pos, const int64_t* ids, float* out, int32_t B, int32_t T, int32_t D,
                                 int32_t vocab, void* stream) {
    REID_CHECK_ARG(tok && pos && ids && out && B > 0 && T > 0 && D > 0 && D % 4 == 0 && vocab > 0, "reid_embed_tokens: bad args");
    const long n = (long)B * T * (D / 4);
    hipLaunchKernelGGL(embed_tokens_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tok, pos, ids, out, B * T, T, D, vocab);
    REID_CHECK_LAUNCH("reid_embed_tokens");
    return REID_OK;
}

extern "C" int reid_l2norm_rows(const float* x, int32_t ldx, float* y, void* y_bf16, int32_t ldy, int32_t rows, int32_t D,
                                float eps, float scale, void* stream) {
    REID_CHECK_ARG(x && (y || y_bf16) && rows > 0 && D % 4 == 0 && D <= 64 * 4 * MAXV, "reid_l2norm_rows: bad args");
    hipLaunchKernelGGL(l2norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, y, (bf16_t*)y_bf16, ldy, rows, D, eps, scale);
    REID_CHECK_LAUNCH("reid_l2norm_rows");
    return REID_OK;
}

// ---------------------------------------------------------------- LoRA arena repack (one launch per optimizer step)
// table[e] = {src_off, rows, cols, dst_off, dstT_off}: dst[dst_off + i] = bf16(src[src_off + i]) (same layout) and,
// when dstT_off >= 0, dstT[dstT_off + c*rows + r] = bf16(src[src_off + r*cols + c]) (transposed copy).
namespace {
__global__ void pack_table_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, const int64_t* __restrict__ table, int n) {
    REID_T16_ENTER();
    const int e = blockIdx.y;
    if (e >= n) return;
    const int64_t* t = table + (size_t)e * 5;
    const int64_t so = t[0], rows = t[1], cols = t[2], d0 = t[3], dT = t[4];
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const bf16_t v = f32_to_bf16(src[so + i]);
        if (d0 >= 0) dst[d0 + i] = v;
        if (dT >= 0) { const int64_t r = i / cols, c = i % cols; dst[dT + c * rows + r] = v; }
    }
}
}  // namespace

extern "C" int reid_pack_bf16_table(const float* src, void* dst, const int64_t* table, int32_t n_entries, void* stream) {
    REID_CHECK_ARG(src && dst && table && n_entries > 0, "reid_pack_bf16_table: bad args");
    hipLaunchKernelGGL(pack_table_kernel, dim3(32, n_entries), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, table, n_entries);
    REID_CHECK_LAUNCH("reid_pack_bf16_table");
    return REID_OK;
}
