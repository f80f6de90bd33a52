// Reduce-over-rows GEMM for gfx950:  C[P,Q] = beta*C + alpha * X[M,P]^T . Y[M,Q]   (fp32 out)
//
// Weight-gradient products (LoRA dA, dB; dW of unfrozen linears).  Both operands are "k-strided"
// for the MFMA (the reduction index m is the slow index of both X and Y), so tiles are staged
// row-major into LDS (16-byte global loads, 16-byte LDS stores, rows padded by 32 B so the
// transposed reads are conflict free) and MFMA fragments are fetched with gfx950's hardware
// transpose read ds_read_b64_tr_b16 (cdna_hip_programming.md T10): lane i of a 16-lane group gets
// column i of a 4-row x 16-column block, i.e. 4 consecutive k for its own matrix row.
// The row range is cut into slabs over blockIdx.z-like slices; partial tiles are combined with
// fp32 atomics (sum order is not fixed: results can differ in the last bits between runs).
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((address_space(3))) s4* lds_s4_ptr;

struct TnParams {
    const bf16_t* X; const bf16_t* Y; float* C;
    int M, P, Q, ldx, ldy, ldc;
    float alpha;
    int tiles_p, tiles_q, slab_rows;
};

__device__ __forceinline__ bf16x8 tr_frag(const char* img, int pitch, int k0, int col0, int lane) {
    // 16x16x32 operand from a row-major [k][col] bf16 image.  Each tr read covers a 4-row x 16-col
    // block: lane 4q+p' of a 16-lane group supplies the address of block row q, columns 4p'..4p'+3,
    // and lane i receives column i of the 4 rows.  The MFMA sums over k, so any assignment of image
    // rows to (lane group fq, element j) is valid as long as BOTH operands use it: element j<4 is
    // row 4fq+j, element j>=4 is row 16+4fq+(j-4).  A 32-lane half then touches 8 consecutive rows
    // per read, which the 32-byte row padding spreads over all 64 banks (conflict free).
    const int l16 = lane & 15, fq = lane >> 4;
    const int q = l16 >> 2, pp = l16 & 3;
    const char* a0 = img + (k0 + 4 * fq + q) * pitch + (col0 + 4 * pp) * 2;
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(a0));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(a0 + 16 * pitch));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int BP, int BQ, int WP, int WQ>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const TnParams p) {
    constexpr int KW = 64;                       // rows of M per step
    constexpr int XP = BP * 2 + 32, YP = BQ * 2 + 32;   // LDS row pitches (bytes)
    constexpr int XB = KW * XP, YB = KW * YP;
    constexpr int TP = BP / WP / 16, TQ = BQ / WQ / 16;
    constexpr int XCH = KW * (BP / 8), YCH = KW * (BQ / 8);   // 16-byte chunks per tile
    constexpr int XL = (XCH + 255) / 256, YL = (YCH + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave / WQ, wq = wave % WQ;

    int b = blockIdx.x;
    const int tq = b % p.tiles_q; b /= p.tiles_q;
    const int tp = b % p.tiles_p; b /= p.tiles_p;
    const int slab = b;
    const int p0 = tp * BP, q0 = tq * BQ;
    const int mbeg = slab * p.slab_rows;
    const int mend = min(p.M, mbeg + p.slab_rows);
    if (mbeg >= mend) return;
    const int steps = (mend - mbeg + KW - 1) / KW;

    uint4 xr[XL], yr[YL];
    auto gload = [&](int t) {
        const int mb = mbeg + t * KW;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int c = tid + i * 256;
            const int row = c / (BP / 8), ch = c % (BP / 8);
            const int m = mb + row, col = p0 + ch * 8;
            xr[i] = (c < XCH && m < mend && col < p.P) ? *(const uint4*)(p.X + (size_t)m * p.ldx + col)
                                                       : uint4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < YL; ++i) {
            const int c = tid + i * 256;
            const int row = c / (BQ / 8), ch = c % (BQ / 8);
            const int m = mb + row, col = q0 + ch * 8;
            yr[i] = (c < YCH && m < mend && col < p.Q) ? *(const uint4*)(p.Y + (size_t)m * p.ldy + col)
                                                       : uint4{0u, 0u, 0u, 0u};
        }
    };
    auto lstore = [&](int buf) {
        char* xs = smem + buf * (XB + YB);
        char* ys = xs + XB;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int c = tid + i * 256;
            if (c < XCH) *(uint4*)(xs + (c / (BP / 8)) * XP + (c % (BP / 8)) * 16) = xr[i];
        }
#pragma unroll
        for (int i = 0; i < YL; ++i) {
            const int c = tid + i * 256;
            if (c < YCH) *(uint4*)(ys + (c / (BQ / 8)) * YP + (c % (BQ / 8)) * 16) = yr[i];
        }
    };

    f32x4 acc[TP][TQ];
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    gload(0);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < steps; ++t) {
        if (t + 1 < steps) gload(t + 1);
        const char* xs = smem + cur * (XB + YB);
        const char* ys = xs + XB;
#pragma unroll
        for (int ks = 0; ks < KW / 32; ++ks) {
            bf16x8 xf[TP], yf[TQ];
#pragma unroll
            for (int i = 0; i < TP; ++i) xf[i] = tr_frag(xs, XP, ks * 32, wp * (BP / WP) + i * 16, lane);
#pragma unroll
            for (int j = 0; j < TQ; ++j) yf[j] = tr_frag(ys, YP, ks * 32, wq * (BQ / WQ) + j * 16, lane);
#pragma unroll
            for (int i = 0; i < TP; ++i)
#pragma unroll
                for (int j = 0; j < TQ; ++j)
                    acc[i][j] = mfma16(xf[i], yf[j], acc[i][j]);
        }
        if (t + 1 < steps) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // D[row = p (4*fq + reg), col = q (lane & 15)]
    const int fq = lane >> 4, l16 = lane & 15;
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            const int qq = q0 + wq * (BQ / WQ) + j * 16 + l16;
            if (qq >= p.Q) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pp = p0 + wp * (BP / WP) + i * 16 + fq * 4 + e;
                if (pp < p.P) atomicAdd(p.C + (size_t)pp * p.ldc + qq, acc[i][j][e] * p.alpha);
            }
        }
}

__global__ void scale_fill_kernel(float* C, int P, int Q, int ldc, float beta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * Q) return;
    float* c = C + (size_t)(i / Q) * ldc + (i % Q);
    *c = beta == 0.f ? 0.f : *c * beta;
}

template <int BP, int BQ, int WP, int WQ>
int launch_tn(TnParams& p, hipStream_t s) {
    constexpr int LDS = 2 * 64 * ((BP * 2 + 32) + (BQ * 2 + 32));
    p.tiles_p = (p.P + BP - 1) / BP;
    p.tiles_q = (p.Q + BQ - 1) / BQ;
    const int tiles = p.tiles_p * p.tiles_q;
    // workgroups per launch: few, long-lived slabs (each pays its load latency once) -- 384 instead of 1536 took the small
    // products from 30 to 22 us stand-alone and 0.6 ms off the train step (less interference with the dX GEMMs; r01 sweep 128..6144)
    int target = 384;
    if (reid_knob(KNOB_TN_BLOCKS) > 0) target = reid_knob(KNOB_TN_BLOCKS);
    int slabs = target / tiles;
    if (slabs < 1) slabs = 1;
    const int max_slabs = (p.M + 255) / 256;
    if (slabs > max_slabs) slabs = max_slabs;
    p.slab_rows = (((p.M + slabs - 1) / slabs) + 63) / 64 * 64;
    slabs = (p.M + p.slab_rows - 1) / p.slab_rows;
    REID_MAX_LDS((gemm_tn_kernel<BP, BQ, WP, WQ>), LDS);
    hipLaunchKernelGGL((gemm_tn_kernel<BP, BQ, WP, WQ>), dim3(tiles * slabs), dim3(256), LDS, s, p);
    REID_CHECK_LAUNCH("reid_gemm_tn");
    return REID_OK;
}

}  // namespace

extern "C" int reid_gemm_tn(const void* X, const void* Y, float* C, int32_t M, int32_t P, int32_t Q, int32_t ldx,
                            int32_t ldy, int32_t ldc, float alpha, float beta, void* stream) {
    REID_CHECK_ARG(X && Y && C, "reid_gemm_tn: null pointer");
    REID_CHECK_ARG(M > 0 && P > 0 && Q > 0, "reid_gemm_tn: empty problem");
    REID_CHECK_ARG(P % 8 == 0 && Q % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "reid_gemm_tn: P, Q, ldx, ldy must be multiples of 8");
    REID_CHECK_ARG(ldx >= P && ldy >= Q && ldc >= Q, "reid_gemm_tn: leading dimensions");
    hipStream_t s = (hipStream_t)stream;
    if (beta != 1.f) {
        const int n = P * Q;
        hipLaunchKernelGGL(scale_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, s, C, P, Q, ldc, beta);
        REID_CHECK_LAUNCH("reid_gemm_tn(fill)");
    }
    TnParams p{(const bf16_t*)X, (const bf16_t*)Y, C, M, P, Q, ldx, ldy, ldc, alpha, 0, 0, 0};
    if (Q <= 32) return launch_tn<128, 32, 4, 1>(p, s);
    if (P <= 32) return launch_tn<32, 128, 1, 4>(p, s);
    if (Q <= 64) return launch_tn<128, 64, 4, 1>(p, s);
    if (P <= 64) return launch_tn<64, 128, 1, 4>(p, s);
    return launch_tn<128, 128, 2, 2>(p, s);
}
