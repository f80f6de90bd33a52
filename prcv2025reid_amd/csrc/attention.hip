// Multi-head attention for short sequences (S <= 224, head_dim 64) on gfx950.
//
// One workgroup = one (sequence, head); the whole K and V of the head live in LDS (<= 56 KB), one
// wavefront per 32-row tile of the other operand.  All products run on v_mfma_f32_32x32x16_bf16 in
// the "keys/queries on the lane" orientation of cdna_hip_programming.md (section 3, "An accumulator
// tile as the next MFMA's operand"): the score tile is computed TRANSPOSED (S^T = K.Q^T) so that a
// lane owns one query column -- the softmax row reduction is in-register plus one cross-half
// shuffle, and the exponentiated tile P^T is already the B operand of O^T = V^T.P^T, no LDS round
// trip.  V is staged row-major like K and consumed through the hardware transpose read
// ds_read_b64_tr_b16.  The backward pass is two kernels of the same shape (dK/dV with keys on the
// lane, dQ with queries on the lane), each needing only products that sum over the accumulator's
// row index; P is recomputed from the saved log-sum-exp.
//
// r02 experiments, measured and not kept (profiles/r02_attn_*.log, r02_attn_persistent_experiment.patch): starting the second
// workgroup of every CU late, waiting for K only before the first sweep (V lands under it), and PERSISTENT workgroups walking the
// (sequence, head) items with a grid stride.  The last is 3-5 % faster alone (fwd 105 vs 110 us) but the item loop costs registers (dQ
// kernel 120 -> 136 VGPRs = one workgroup per CU instead of two: 113 -> 152 us) and, inside the training step, workgroups that start late
// behind the side-stream reductions still own a fixed share of the items.  Per-workgroup timestamps of the forward kernel: K/V/Q
// landing 3.9 us, first sweep 3.3, second sweep 3.6, stores 1.2; per item and CU the matrix pipe needs 2.2 us, the exponentials ~3 us,
// HBM 4 us -- they add up rather than overlap with two 7-wave workgroups per CU.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((address_space(3))) s4* lds_s4_ptr;

constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int swz(int row, int c) { return c ^ ((row >> 1) & 7); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Stage rows [0, 32 NT) x 64 of one head (column offset already applied to `base`, row stride ld) into a swizzled
// [32 NT][64] LDS image with 16-byte LDS-DMA: every wave issues its 4 wave-instructions (8 rows each) back to back, so
// all of the head's bytes are in flight at once (the first version looped load -> LDS store per 16-byte chunk and paid
// eight exposed HBM round trips per workgroup: ~16 of the ~22 us a workgroup lived).  The LDS image is lane-linear, the
// swizzle is applied to the SOURCE chunk.  Rows >= S repeat row S-1 (finite values): every consumer masks them (padded
// keys get probability 0, padded queries get P = 0 through lse = +inf), so no zero fill is needed.
template <int NT>
__device__ __forceinline__ void stage_head(const bf16_t* __restrict__ base, int ld, int S, char* lds, int wave, int lane,
                                           int row_limit = 1 << 30) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rblk = (i * NT + wave) * 8;
        if (rblk >= row_limit) continue;                    // (wave-uniform) rows nobody will read: query tiles left out by q_tiles
        const int row = rblk + (lane >> 3);
        const int c = swz(row, lane & 7);
        const int grow = row < S ? row : S - 1;
        __builtin_amdgcn_global_load_lds((gptr_t)(base + (size_t)grow * ld + c * 8), (lptr_t)(lds + rblk * 128), 16, 0, 0);
    }
}

// A operand (32 rows x 16 k) of mfma_32x32x16 from a swizzled row-major image: rows = image rows.
__device__ __forceinline__ bf16x8 row_frag(const char* img, int row0, int kchunk, int lane) {
    const int row = row0 + (lane & 31);
    return *(const bf16x8*)(img + row * 128 + (swz(row, kchunk + (lane >> 5)) << 4));
}

// A operand whose 32 MFMA rows are image COLUMNS col0..col0+31 and whose 16 k are image rows, in the
// accumulator-as-operand k order: element j of lane half h = image row krow0 + 8(j>>2) + 4h + (j&3).
__device__ __forceinline__ bf16x8 col_frag(const char* img, int krow0, int col0, int lane) {
    const int h = lane >> 5, half16 = (lane >> 4) & 1, i = lane & 15;
    const int q = i >> 2, pp = i & 3;
    const int col = col0 + 16 * half16 + 4 * pp;          // 4 consecutive columns = 8 bytes inside chunk col>>3
    const int r_lo = krow0 + 4 * h + q, r_hi = r_lo + 8;
    const char* a_lo = img + r_lo * 128 + (swz(r_lo, col >> 3) << 4) + (col & 7) * 2;
    const char* a_hi = img + r_hi * 128 + (swz(r_hi, col >> 3) << 4) + (col & 7) * 2;
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)a_lo);
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)a_hi);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// B operand (16 k x 32 cols) straight from global memory: column = matrix row (row0 + lane&31), k = 8 contiguous elements.
__device__ __forceinline__ bf16x8 gfrag(const bf16_t* __restrict__ base, int ld, int row, int kchunk, int lane) {
    return *(const bf16x8*)(base + (size_t)row * ld + (kchunk + (lane >> 5)) * 8);
}

// Lane-constant byte offsets of the fragment reads.  Tiles start at multiples of 32 rows (and k sub-steps at multiples
// of 16 rows), which leaves the swizzle term ((row >> 1) & 7) unchanged, so every read below is `base + tile * 4096 (+ s * 2048)
// + lane offset`: the address arithmetic is done once per kernel instead of once per MFMA operand (the first version
// spent 51 VALU instructions per MFMA, mostly on these addresses and on the softmax).
struct FragOff {
    int row[4];      // row_frag: [ks]
    int col_lo[2];   // col_frag low 4 rows:  [dt]
    int col_hi[2];   // col_frag high 4 rows: [dt]
};
__device__ __forceinline__ FragOff make_frag_off(int lane) {
    FragOff f;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f.row[ks] = r * 128 + (swz(r, 2 * ks + h) << 4);
    const int half16 = (lane >> 4) & 1, i = lane & 15, q = i >> 2, pp = i & 3;
    const int r_lo = 4 * h + q, r_hi = r_lo + 8;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const int col = dt * 32 + 16 * half16 + 4 * pp;
        f.col_lo[dt] = r_lo * 128 + (swz(r_lo, col >> 3) << 4) + (col & 7) * 2;
        f.col_hi[dt] = r_hi * 128 + (swz(r_hi, col >> 3) << 4) + (col & 7) * 2;
    }
    return f;
}
__device__ __forceinline__ bf16x8 row_frag_o(const char* img, int tile_row0, int ks, const FragOff& f) {
    return *(const bf16x8*)(img + tile_row0 * 128 + f.row[ks]);
}
__device__ __forceinline__ bf16x8 col_frag_o(const char* img, int krow0, int dt, const FragOff& f) {
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(img + krow0 * 128 + f.col_lo[dt]));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(img + krow0 * 128 + f.col_hi[dt]));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32, x <= 0 here

__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (short)f32_to_bf16(a[8 * s + j]);
    return r;
}

__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

struct AttnParams {
    const bf16_t* qkv; int ld; const uint8_t* key_mask;
    bf16_t* out; int ldo; float* lse;
    const bf16_t* dout; bf16_t* dqkv; int lddqkv; float* delta;
    int n_seq, S, heads, causal;
    int q_tiles;  // > 0: only the first q_tiles 32-row QUERY tiles of every sequence are computed (all keys still take part)
    int dbg;     // timing experiments (REID_ATTN_DBG): 1 = no output stores, 2 = also no softmax / P.V, 3 = staging only
};

// A 32x32 MFMA result tile holds, for the row on lanes l and l+32, the two 4-element halves of every 8-element column
// group.  One v_permlane32_swap per packed dword regroups a PAIR of groups so that lane l owns group 2j whole and lane
// l+32 group 2j+1: 16-byte stores instead of 8-byte ones.  All 64 lanes must be active.
__device__ __forceinline__ void store_tile_row16(bf16_t* row, bool ok, const f32x16& t, float scale, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int a = 8 * j, b = 8 * j + 4;
        const auto r0 = __builtin_amdgcn_permlane32_swap(pack_bf16x2(t[a] * scale, t[a + 1] * scale),
                                                         pack_bf16x2(t[b] * scale, t[b + 1] * scale), false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(pack_bf16x2(t[a + 2] * scale, t[a + 3] * scale),
                                                         pack_bf16x2(t[b + 2] * scale, t[b + 3] * scale), false, false);
        if (ok) *(uint4*)(row + 8 * (2 * j + (lane >> 5))) = uint4{r0[0], r1[0], r0[1], r1[1]};
    }
}


// ------------------------------------------------------------------------------------------ forward
// TWO_PASS (long sequences): the score tiles are NOT kept in registers.  Pass 1 computes them for the row maximum only,
// pass 2 recomputes each tile, exponentiates it and feeds it straight into the P.V MFMAs.  The 28 extra MFMAs per wave
// are cheap next to what they buy: ~100 instead of ~200 VGPRs, so TWO workgroups fit a CU (LDS 56 KiB each) and one's
// K/V staging and output stores overlap the other's MFMAs and softmax (a single resident workgroup ran load -> compute ->
// store strictly in sequence).  Short sequences (text tower, NT <= 3) keep the single pass.
template <int NT, bool TWO_PASS>   // number of 32-row tiles: S <= 32*NT
__global__ __launch_bounds__(NT * 64, TWO_PASS ? 4 : 1) void attn_fwd_kernel(const AttnParams p) {
    REID_T16_ENTER();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + NT * 32 * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int seq = blockIdx.x / p.heads, head = blockIdx.x % p.heads;
    const int d = p.heads * 64;
    const bf16_t* qb = p.qkv + (size_t)seq * p.S * p.ld + head * 64;
    stage_head<NT>(qb + d, p.ld, p.S, Ks, wave, lane);
    const int q0 = wave * 32;
    const int qi = q0 + (lane & 31);
    const int qrow = qi < p.S ? qi : p.S - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = gfrag(qb, p.ld, qrow, 2 * ks, lane);
    stage_head<NT>(qb + 2 * d, p.ld, p.S, Vs, wave, lane);
    __syncthreads();
    if (q0 >= p.S || (p.q_tiles > 0 && wave >= p.q_tiles)) return;   // wave-uniform; no barrier follows

    if (REID_DBG(p) == 3) return;
    const FragOff fo = make_frag_off(lane);
    const uint8_t* km = p.key_mask ? p.key_mask + (size_t)seq * p.S : nullptr;
    const int h4 = 4 * (lane >> 5);
    const bool general = km != nullptr || p.causal;    // general masks (text tower, fusion): per-element predicate
    // masked score tile kt: S^T = K.Q^T, rows = keys (registers), columns = queries (lanes)
    auto score_tile = [&](int kt) -> f32x16 {
        f32x16 t;
#pragma unroll
        for (int e = 0; e < 16; ++e) t[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) t = mfma32(row_frag_o(Ks, kt * 32, ks, fo), qf[ks], t);
        if (general) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + h4;
                bool ok = key < p.S;
                if (ok && km) ok = km[key] != 0;
                if (p.causal) ok = ok && key <= qi;
                t[e] = ok ? t[e] : -INFINITY;
            }
        } else if (kt == NT - 1) {                      // vision: only the padded keys of the last tile are masked
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + h4;
                t[e] = key < p.S ? t[e] : -INFINITY;
            }
        }
        return t;
    };
    const float c = 0.125f * LOG2E;     // head_dim^-0.5 (mer_lora.py:128-129), exp via exp2
    f32x16 ot[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) ot[dt][e] = 0.f;
    float mx = -INFINITY, l = 0.f;
    if (TWO_PASS) {
#pragma unroll 1
        for (int kt = 0; kt < NT; ++kt) {               // (not unrolled: one tile's temporaries at a time keeps the kernel at <= 128 VGPRs)
            const f32x16 t = score_tile(kt);
#pragma unroll
            for (int e = 0; e < 16; e += 2) mx = fmaxf(mx, fmaxf(t[e], t[e + 1]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (mx == -INFINITY) mx = 0.f;
        const float nmc = -mx * c;
#pragma unroll 1
        for (int kt = 0; kt < (REID_DBG(p) == 2 ? 0 : NT); ++kt) {
            f32x16 t = score_tile(kt);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pe = fast_exp2(fmaf(t[e], c, nmc));
                t[e] = pe;
                l += pe;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = pack8(t, s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) ot[dt] = mfma32(col_frag_o(Vs, kt * 32 + 16 * s2, dt, fo), pf, ot[dt]);
            }
        }
    } else {
        f32x16 st[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) st[kt] = score_tile(kt);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int e = 0; e < 16; e += 2) mx = fmaxf(mx, fmaxf(st[kt][e], st[kt][e + 1]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (mx == -INFINITY) mx = 0.f;
        const float nmc = -mx * c;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pe = fast_exp2(fmaf(st[kt][e], c, nmc));
                st[kt][e] = pe;
                l += pe;
            }
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = pack8(st[kt], s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) ot[dt] = mfma32(col_frag_o(Vs, kt * 32 + 16 * s2, dt, fo), pf, ot[dt]);
            }
    }
    l += __shfl_xor(l, 32, 64);
    if (REID_DBG(p) >= 1 && l != 12345.678f) return;
    {
        const float inv = 1.0f / l;
        bf16_t* orow = p.out + ((size_t)seq * p.S + qrow) * p.ldo + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) store_tile_row16(orow + dt * 32, qi < p.S, ot[dt], inv, lane);
        if (qi < p.S && p.lse && lane < 32) p.lse[((size_t)seq * p.heads + head) * p.S + qi] = mx * 0.125f + logf(l);
    }
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
// Wave w owns key tile w (keys on the lane).  Per query tile: S = Q.K^T and dP = dO.V^T land as
// [query rows (regs) x key columns (lanes)] and feed dV^T += dO^T.P and dK^T += Q^T.dS as B operands.
// (160 VGPRs = 3 waves per SIMD, i.e. one 7-wave workgroup per CU.  r01 experiment: dV and dK in two sweeps over the query
//  tiles -- 120 VGPRs, two workgroups per CU, S recomputed (20 instead of 16 MFMAs per tile pair) -- was 0.2 ms per step
//  SLOWER: occupancy is not what holds this kernel back.)
template <int NT>
__global__ __launch_bounds__(NT * 64) void attn_bwd_dkv_kernel(const AttnParams p) {
    REID_T16_ENTER();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;
    char* Gs = smem + NT * 32 * 128;                 // dO
    float* rowc = (float*)(smem + 2 * NT * 32 * 128);   // [2][NT*32]: lse, delta
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int seq = blockIdx.x / p.heads, head = blockIdx.x % p.heads;
    const int d = p.heads * 64;
    const bf16_t* qb = p.qkv + (size_t)seq * p.S * p.ld + head * 64;
    const bf16_t* gb = p.dout + (size_t)seq * p.S * p.ldo + head * 64;
    const int q_rows = p.q_tiles > 0 ? p.q_tiles * 32 : NT * 32;
    stage_head<NT>(qb, p.ld, p.S, Qs, wave, lane, q_rows);
    stage_head<NT>(gb, p.ldo, p.S, Gs, wave, lane, q_rows);
    for (int t = tid; t < NT * 32; t += NT * 64) {
        const size_t o = ((size_t)seq * p.heads + head) * p.S + t;
        rowc[t] = t < p.S ? -p.lse[o] * LOG2E : -INFINITY;          // exp2(s*c + rowc) = P; padded queries give 0
        rowc[NT * 32 + t] = t < p.S ? -p.delta[o] * 0.125f : 0.f;
    }
    const int k0 = wave * 32;
    const int ki = k0 + (lane & 31);
    const int krow = ki < p.S ? ki : p.S - 1;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = gfrag(qb + d, p.ld, krow, 2 * ks, lane);
        vf[ks] = gfrag(qb + 2 * d, p.ld, krow, 2 * ks, lane);
    }
    __syncthreads();
    if (k0 >= p.S) return;
    const uint8_t* km = p.key_mask ? p.key_mask + (size_t)seq * p.S : nullptr;
    const bool key_ok = ki < p.S && (!km || km[ki] != 0);

    const FragOff fo = make_frag_off(lane);
    f32x16 dkt[2], dvt[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dkt[dt][e] = 0.f; dvt[dt][e] = 0.f; }
    const float c = 0.125f * LOG2E;
    const int h4 = 4 * (lane >> 5);
    for (int qt = 0; qt < NT; ++qt) {
        if (qt * 32 >= p.S || (p.q_tiles > 0 && qt >= p.q_tiles)) break;
        f32x16 s, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s = mfma32(row_frag_o(Qs, qt * 32, ks, fo), kf[ks], s);
            dp = mfma32(row_frag_o(Gs, qt * 32, ks, fo), vf[ks], dp);
        }
        // P = exp2(s*c - lse*log2e); dS = P * (dP - delta) / 8.  A lane is one key column: an invalid key only spoils its own
        // (unused / zeroed) output column, so only the causal mask needs a per-element predicate here.
        f32x16 pm, ds;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 nl = *(const f32x4*)(rowc + qt * 32 + 8 * g + h4);
            const f32x4 nd = *(const f32x4*)(rowc + NT * 32 + qt * 32 + 8 * g + h4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * g + j;
                float pe = fast_exp2(fmaf(s[e], c, nl[j]));
                if (p.causal && ki > qt * 32 + 8 * g + h4 + j) pe = 0.f;
                pm[e] = pe;
                ds[e] = pe * fmaf(dp[e], 0.125f, nd[j]);
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 pf = pack8(pm, s2), df = pack8(ds, s2);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dvt[dt] = mfma32(col_frag_o(Gs, qt * 32 + 16 * s2, dt, fo), pf, dvt[dt]);
                dkt[dt] = mfma32(col_frag_o(Qs, qt * 32 + 16 * s2, dt, fo), df, dkt[dt]);
            }
        }
    }
    if (!key_ok) {                                   // masked-out key: zero gradient
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { dkt[dt][e] = 0.f; dvt[dt][e] = 0.f; }
    }
    {
        const bool ok = ki < p.S;
        bf16_t* drow = p.dqkv + ((size_t)seq * p.S + (ok ? ki : 0)) * p.lddqkv + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            store_tile_row16(drow + d + dt * 32, ok, dkt[dt], 1.0f, lane);
            store_tile_row16(drow + 2 * d + dt * 32, ok, dvt[dt], 1.0f, lane);
        }
    }
}

// ------------------------------------------------------------------------------------------ backward: dQ
// Wave w owns query tile w (queries on the lane).  S^T = K.Q^T and dP^T = V.dO^T land as
// [key rows (regs) x query columns (lanes)]; dQ^T += K^T.dS^T.
template <int NT>
__global__ __launch_bounds__(NT * 64) void attn_bwd_dq_kernel(const AttnParams p) {
    REID_T16_ENTER();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + NT * 32 * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int seq = blockIdx.x / p.heads, head = blockIdx.x % p.heads;
    const int d = p.heads * 64;
    const bf16_t* qb = p.qkv + (size_t)seq * p.S * p.ld + head * 64;
    const bf16_t* gb = p.dout + (size_t)seq * p.S * p.ldo + head * 64;
    stage_head<NT>(qb + d, p.ld, p.S, Ks, wave, lane);
    stage_head<NT>(qb + 2 * d, p.ld, p.S, Vs, wave, lane);
    const int q0 = wave * 32;
    const int qi = q0 + (lane & 31);
    const int qrow = qi < p.S ? qi : p.S - 1;
    bf16x8 qf[4], gf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = gfrag(qb, p.ld, qrow, 2 * ks, lane);
        gf[ks] = gfrag(gb, p.ldo, qrow, 2 * ks, lane);
    }
    const size_t so = ((size_t)seq * p.heads + head) * p.S + qrow;
    const float nl = -p.lse[so] * LOG2E;
    // delta[q] = sum_d dO[q, d] O[q, d], formed HERE from the dO fragments this wave holds anyway and the matching O fragments (a lane has
    // 32 of the row's 64 products, lane ^ 32 the rest) and published for the dK/dV kernel, which is launched after this one: the separate
    // delta kernel (one more pass over O and dO, 30-50 us per layer) is gone.
    float dsum = 0.f;
    {
        const bf16_t* ob = p.out + (size_t)seq * p.S * p.ldo + head * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 of = gfrag(ob, p.ldo, qrow, 2 * ks, lane);
#pragma unroll
            for (int j = 0; j < 8; ++j) dsum = fmaf(bf16_to_f32((bf16_t)gf[ks][j]), bf16_to_f32((bf16_t)of[j]), dsum);
        }
    }
    dsum += __shfl_xor(dsum, 32, 64);
    if (qi < p.S && lane < 32) p.delta[so] = dsum;
    const float nd = -dsum * 0.125f;
    __syncthreads();
    if (q0 >= p.S) return;
    if (p.q_tiles > 0 && wave >= p.q_tiles) {               // query tile left out: its dQ rows are exactly zero
        if (qi < p.S) {
            bf16_t* drow = p.dqkv + ((size_t)seq * p.S + qi) * p.lddqkv + head * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) *(uint2*)(drow + dt * 32 + 8 * g + 4 * (lane >> 5)) = uint2{0u, 0u};
        }
        return;
    }
    const uint8_t* km = p.key_mask ? p.key_mask + (size_t)seq * p.S : nullptr;
    const FragOff fo = make_frag_off(lane);
    f32x16 dqt[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) dqt[dt][e] = 0.f;
    const float c = 0.125f * LOG2E;
    const int h4 = 4 * (lane >> 5);
    const bool slow = km != nullptr || p.causal;
    for (int kt = 0; kt < NT; ++kt) {
        if (kt * 32 >= p.S) break;
        f32x16 s, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s = mfma32(row_frag_o(Ks, kt * 32, ks, fo), qf[ks], s);
            dp = mfma32(row_frag_o(Vs, kt * 32, ks, fo), gf[ks], dp);
        }
        const bool edge = slow || (kt + 1) * 32 > p.S;    // only the last key tile holds padded keys
        f32x16 ds;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float pe = fast_exp2(fmaf(s[e], c, nl));
            if (edge) {
                const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + h4;
                bool ok = key < p.S;
                if (ok && km) ok = km[key] != 0;
                if (p.causal) ok = ok && key <= qi;
                pe = ok ? pe : 0.f;
            }
            ds[e] = pe * fmaf(dp[e], 0.125f, nd);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 df = pack8(ds, s2);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) dqt[dt] = mfma32(col_frag_o(Ks, kt * 32 + 16 * s2, dt, fo), df, dqt[dt]);
        }
    }
    {
        const bool ok = qi < p.S;
        bf16_t* drow = p.dqkv + ((size_t)seq * p.S + (ok ? qi : 0)) * p.lddqkv + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) store_tile_row16(drow + dt * 32, ok, dqt[dt], 1.0f, lane);
    }
}

// ------------------------------------------------------------------------------------------ backward: one pass (r04)
// dQ, dK and dV of one (sequence, head) from ONE staging of Q, K, V and dO (the two kernels above each stage half of them and
// each read dO / Q / K / V again: 8 LDS-DMA head images and ~930 MB of HBM traffic per ViT layer where 4 images and ~620 MB do).
// One workgroup holds the four swizzled images (4 x NT x 4 KiB = 112 KiB at S = 197) and runs, without any barrier between them,
//   phase 1 -- wave w owns KEY tile w (keys on the lane):  S = Q K^T, dP = dO V^T per query tile -> dV^T += dO^T P, dK^T += Q^T dS
//   phase 2 -- wave w owns QUERY tile w (queries on the lane):  S^T = K Q^T, dP^T = V dO^T per key tile -> dQ^T += K^T dS^T
// i.e. the products of the two kernels above, fed from LDS.  delta = rowsum(dO . O) is formed first by the query-owning waves (O is
// read once, from global memory) and shared through LDS with the saved log-sum-exp.
// In-wave pipelining: the S / dP MFMAs of tile t + 1 are issued BEFORE the exponentials of tile t (two accumulator sets), so that
// a wave's matrix work does not wait behind its own exp -> pack -> MFMA chain (r03: with one workgroup per CU that chain is exposed).
// No key mask / causal form: the text tower's backward (rare: freeze_text_backbone=False) keeps the two-kernel path.
#ifdef REID_ATTN_TRACE
// experiment builds (tools/exp_attn_bwd_trace.py): s_memrealtime stamps of wave 0 / lane 0 of every workgroup at the phase boundaries
__device__ unsigned long long* g_attn_bwd_trace = nullptr;
#define ATTN_BWD_STAMP(slot) do { if (g_attn_bwd_trace && threadIdx.x == 0) g_attn_bwd_trace[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ATTN_BWD_STAMP(slot) do { } while (0)
#endif
template <int NT>
__global__ __launch_bounds__(NT * 64) void attn_bwd_fused_kernel(const AttnParams p) {
    REID_T16_ENTER();
    ATTN_BWD_STAMP(0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int IMG = NT * 32 * 128;
    char* Qs = smem;
    char* Gs = smem + IMG;                           // dO
    char* Ks = smem + 2 * IMG;
    char* Vs = smem + 3 * IMG;
    float* rowc = (float*)(smem + 4 * IMG);          // [2][NT*32]: -lse * log2(e), -delta / 8
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int seq = blockIdx.x / p.heads, head = blockIdx.x % p.heads;
    const int d = p.heads * 64;
    const bf16_t* qb = p.qkv + (size_t)seq * p.S * p.ld + head * 64;
    const bf16_t* gb = p.dout + (size_t)seq * p.S * p.ldo + head * 64;
    const bf16_t* ob = p.out + (size_t)seq * p.S * p.ldo + head * 64;
    const int nq = p.q_tiles > 0 ? (p.q_tiles < NT ? p.q_tiles : NT) : NT;     // query tiles that take part (class-row pruning)
    const int q_rows = nq * 32;
    // all four head images in flight at once (16 LDS-DMA instructions per wave)
    stage_head<NT>(qb, p.ld, p.S, Qs, wave, lane, q_rows);
    stage_head<NT>(gb, p.ldo, p.S, Gs, wave, lane, q_rows);
    stage_head<NT>(qb + d, p.ld, p.S, Ks, wave, lane);
    stage_head<NT>(qb + 2 * d, p.ld, p.S, Vs, wave, lane);
    const int t0 = wave * 32;                        // first row of this wave's tile (key tile in phase 1, query tile in phase 2)
    const int ti = t0 + (lane & 31);
    const int trow = ti < p.S ? ti : p.S - 1;
    const bool own_q = wave < nq && t0 < p.S;        // wave-uniform: this wave's QUERY tile takes part
    // delta of this wave's query rows: O fragments straight from global memory (their latency hides behind the image staging)
    bf16x8 of[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) of[ks] = gfrag(ob, p.ldo, trow, 2 * ks, lane);
    const size_t so = ((size_t)seq * p.heads + head) * p.S + trow;
    const float lse_v = p.lse[so];
    const FragOff fo = make_frag_off(lane);
    __syncthreads();                                 // vmcnt(0) + barrier: the images are in LDS
    ATTN_BWD_STAMP(1);
    {
        float dsum = 0.f;
        if (own_q) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 gfk = row_frag_o(Gs, t0, ks, fo);
#pragma unroll
                for (int j = 0; j < 8; ++j) dsum = fmaf(bf16_to_f32((bf16_t)gfk[j]), bf16_to_f32((bf16_t)of[ks][j]), dsum);
            }
        }
        dsum += __shfl_xor(dsum, 32, 64);
        if (lane < 32) {
            const bool live = own_q && ti < p.S;     // padded / pruned query rows: P = exp2(-inf) = 0
            rowc[ti] = live ? -lse_v * LOG2E : -INFINITY;
            rowc[NT * 32 + ti] = live ? -dsum * 0.125f : 0.f;
            if (live && p.delta) p.delta[so] = dsum;
        }
    }
    __syncthreads();
    ATTN_BWD_STAMP(2);
    if (t0 >= p.S) return;                           // (wave-uniform; no barrier follows)
    const float c = 0.125f * LOG2E;
    const int h4 = 4 * (lane >> 5);
    const bool tile_ok = ti < p.S;

    // ------------------------------------------------------------------ phase 1: this wave's key tile
    {
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kf[ks] = row_frag_o(Ks, t0, ks, fo); vf[ks] = row_frag_o(Vs, t0, ks, fo); }
        f32x16 dkt[2], dvt[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { dkt[dt][e] = 0.f; dvt[dt][e] = 0.f; }
        const int nqt = nq < (p.S + 31) / 32 ? nq : (p.S + 31) / 32;      // query tiles with at least one real row
        auto scores = [&](int qt, f32x16& s, f32x16& dp) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = mfma32(row_frag_o(Qs, qt * 32, ks, fo), kf[ks], s);
                dp = mfma32(row_frag_o(Gs, qt * 32, ks, fo), vf[ks], dp);
            }
        };
        f32x16 s, dp, s_n, dp_n;
        if (nqt > 0) scores(0, s, dp);
        for (int qt = 0; qt < nqt; ++qt) {
            if (qt + 1 < nqt) scores(qt + 1, s_n, dp_n);          // next tile's matrix work first: it runs under this tile's exponentials
            // one 16-row half of the tile at a time, packed to 16 bits at once (P and dS never exist as two fp32 tiles: registers)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 pf, df;
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int g = 2 * s2 + g2;
                    const f32x4 nl = *(const f32x4*)(rowc + qt * 32 + 8 * g + h4);
                    const f32x4 nd = *(const f32x4*)(rowc + NT * 32 + qt * 32 + 8 * g + h4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * g + j;
                        const float pe = fast_exp2(fmaf(s[e], c, nl[j]));
                        pf[4 * g2 + j] = (short)f32_to_bf16(pe);
                        df[4 * g2 + j] = (short)f32_to_bf16(pe * fmaf(dp[e], 0.125f, nd[j]));
                    }
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dvt[dt] = mfma32(col_frag_o(Gs, qt * 32 + 16 * s2, dt, fo), pf, dvt[dt]);
                    dkt[dt] = mfma32(col_frag_o(Qs, qt * 32 + 16 * s2, dt, fo), df, dkt[dt]);
                }
            }
            s = s_n; dp = dp_n;
        }
        ATTN_BWD_STAMP(3);
        bf16_t* drow = p.dqkv + ((size_t)seq * p.S + trow) * p.lddqkv + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            store_tile_row16(drow + d + dt * 32, tile_ok, dkt[dt], 1.0f, lane);
            store_tile_row16(drow + 2 * d + dt * 32, tile_ok, dvt[dt], 1.0f, lane);
        }
    }

    // ------------------------------------------------------------------ phase 2: this wave's query tile
    ATTN_BWD_STAMP(4);
    bf16_t* qrow_out = p.dqkv + ((size_t)seq * p.S + trow) * p.lddqkv + head * 64;
    if (!own_q) {                                    // query tile left out by q_tiles: its dQ rows are exactly zero
        if (tile_ok) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) *(uint2*)(qrow_out + dt * 32 + 8 * g + 4 * (lane >> 5)) = uint2{0u, 0u};
        }
        return;
    }
    {
        bf16x8 qf[4], gf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qf[ks] = row_frag_o(Qs, t0, ks, fo); gf[ks] = row_frag_o(Gs, t0, ks, fo); }
        const float nl = rowc[ti], nd = rowc[NT * 32 + ti];
        f32x16 dqt[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) dqt[dt][e] = 0.f;
        const int nkt = (p.S + 31) / 32;
        auto scores_t = [&](int kt, f32x16& s, f32x16& dp) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = mfma32(row_frag_o(Ks, kt * 32, ks, fo), qf[ks], s);
                dp = mfma32(row_frag_o(Vs, kt * 32, ks, fo), gf[ks], dp);
            }
        };
        f32x16 s, dp, s_n, dp_n;
        scores_t(0, s, dp);
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + 1 < nkt) scores_t(kt + 1, s_n, dp_n);
            const bool edge = (kt + 1) * 32 > p.S;    // only the last key tile holds padded keys
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 df;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = 8 * s2 + j;
                    float pe = fast_exp2(fmaf(s[e], c, nl));
                    if (edge) {
                        const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + h4;
                        pe = key < p.S ? pe : 0.f;
                    }
                    df[j] = (short)f32_to_bf16(pe * fmaf(dp[e], 0.125f, nd));
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) dqt[dt] = mfma32(col_frag_o(Ks, kt * 32 + 16 * s2, dt, fo), df, dqt[dt]);
            }
            s = s_n; dp = dp_n;
        }
        ATTN_BWD_STAMP(5);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) store_tile_row16(qrow_out + dt * 32, tile_ok, dqt[dt], 1.0f, lane);
        ATTN_BWD_STAMP(6);
    }
}

// ------------------------------------------------------------------------------------------ backward: one pass, PERSISTENT (r04)
// The kernel above is bound by what happens BETWEEN items: 112 KiB of LDS images = one workgroup per CU, so the ~4 us it takes the four
// images to land is exposed once per (sequence, head) -- about as long as the item's arithmetic.  Here one workgroup per CU walks the items
// (blockIdx.x, + gridDim.x, ...) and the images of an item land under arithmetic that does not need them:
//   top of item n:  [wait: Q / dO / O images of item n landed]  stage K / V (n)  -- they land under delta + phase 1, which take K and V of the
//                   wave's own key tile from REGISTERS (fetched one item ahead) and read only the Q / dO images
//   after phase 1:  the wave's own Q / dO fragments go to registers; barrier; stage Q / dO / O (n + 1) into the freed regions -- they land
//                   under phase 2, which reads only the K / V images; [counted wait: K / V (n) landed]
// Waits are COUNTED: the wait in the middle sits right after exactly 12 LDS-DMA instructions of this wave (three images x 4), so
// `vmcnt(12)` means "everything issued before those twelve has landed" whatever else is in flight; outputs are stored after a wait,
// never in front of one.  Barriers are raw s_barrier (a __syncthreads would drain the prefetch).
// hipcc places `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 BUILTIN while any LDS-DMA is outstanding (it does not for plain
// ds_read_b128), which would serialise exactly the overlap this kernel is built for: the transposed reads here are inline assembly
// (eight reads and one lgkmcnt wait per statement).
typedef __attribute__((address_space(3))) char* lds_cptr;
__device__ __forceinline__ uint32_t lds_addr(const char* ptr) { return (uint32_t)(uintptr_t)(lds_cptr)(char*)ptr; }

// four col_frag operands (eight transposed 8-byte reads) issued back to back, one wait; a[2 i], a[2 i + 1] = LDS byte addresses of operand i
__device__ __forceinline__ void tr_read4(const uint32_t (&a)[8], bf16x8 (&f)[4]) {
    unsigned long long r0, r1, r2, r3, r4, r5, r6, r7;
    asm volatile("ds_read_b64_tr_b16 %0, %8\n\t"
                 "ds_read_b64_tr_b16 %1, %9\n\t"
                 "ds_read_b64_tr_b16 %2, %10\n\t"
                 "ds_read_b64_tr_b16 %3, %11\n\t"
                 "ds_read_b64_tr_b16 %4, %12\n\t"
                 "ds_read_b64_tr_b16 %5, %13\n\t"
                 "ds_read_b64_tr_b16 %6, %14\n\t"
                 "ds_read_b64_tr_b16 %7, %15\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7])
                 : "memory");
    const unsigned long long r[8] = {r0, r1, r2, r3, r4, r5, r6, r7};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const s4 lo = __builtin_bit_cast(s4, r[2 * i]), hi = __builtin_bit_cast(s4, r[2 * i + 1]);
        f[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
}
__device__ __forceinline__ void tr_read2(const uint32_t (&a)[4], bf16x8 (&f)[2]) {
    unsigned long long r0, r1, r2, r3;
    asm volatile("ds_read_b64_tr_b16 %0, %4\n\t"
                 "ds_read_b64_tr_b16 %1, %5\n\t"
                 "ds_read_b64_tr_b16 %2, %6\n\t"
                 "ds_read_b64_tr_b16 %3, %7\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3])
                 : "memory");
    const unsigned long long r[4] = {r0, r1, r2, r3};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const s4 lo = __builtin_bit_cast(s4, r[2 * i]), hi = __builtin_bit_cast(s4, r[2 * i + 1]);
        f[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
}

// stage_head with the LDS-DMA issued from inline assembly: hipcc then has NO vector-memory operation of this kernel's prefetch in its
// scoreboard and leaves the LDS reads of the phases alone (with the builtin it puts `s_waitcnt vmcnt(0)` in front of every transposed
// read while a DMA is outstanding).  M0 = LDS base of the wave-instruction; nothing else in the persistent kernel uses M0 (no builtin DMA).
template <int NT>
__device__ __forceinline__ void stage_head_asm(const bf16_t* __restrict__ base, int ld, int S, uint32_t lds_base, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rblk = (i * NT + wave) * 8;
        const int row = rblk + (lane >> 3);
        const int c = swz(row, lane & 7);
        const int grow = row < S ? row : S - 1;
        const bf16_t* src = base + (size_t)grow * ld + c * 8;
        const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)rblk * 128u);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory");
    }
}

template <int NT>
__global__ __launch_bounds__(NT * 64) void attn_bwd_pers_kernel(const AttnParams p, int n_items) {
    REID_T16_ENTER();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int IMG = NT * 32 * 128;
    char* Qs = smem;
    char* Gs = smem + IMG;                           // dO
    char* Os = smem + 2 * IMG;                       // O (delta only)
    char* Ks = smem + 3 * IMG;
    char* Vs = smem + 4 * IMG;
    float* rowc = (float*)(smem + 5 * IMG);          // [2][NT*32]: -lse * log2(e), -delta / 8
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = p.heads * 64;
    const int t0 = wave * 32;
    const int ti = t0 + (lane & 31);
    const int trow = ti < p.S ? ti : p.S - 1;
    const bool tile_ok = ti < p.S;
    const float c = 0.125f * LOG2E;
    const int h4 = 4 * (lane >> 5);
    const FragOff fo = make_frag_off(lane);
    const int nt = (p.S + 31) / 32;                  // == NT (launch_bwd picks NT = ceil(S / 32)): every wave owns a real tile
    const uint32_t aQ = lds_addr(Qs), aG = lds_addr(Gs), aO = lds_addr(Os), aK = lds_addr(Ks), aV = lds_addr(Vs);
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto item_ptrs = [&](int item, const bf16_t*& qb, const bf16_t*& gb, const bf16_t*& ob, size_t& so0) {
        const int seq = item / p.heads, head = item % p.heads;
        qb = p.qkv + (size_t)seq * p.S * p.ld + head * 64;
        gb = p.dout + (size_t)seq * p.S * p.ldo + head * 64;
        ob = p.out + (size_t)seq * p.S * p.ldo + head * 64;
        so0 = ((size_t)seq * p.heads + head) * p.S;
    };

    int item = blockIdx.x;
    const bf16_t *qb, *gb, *ob;
    size_t so0;
    item_ptrs(item, qb, gb, ob, so0);
    // prologue: Q / dO / O images and the register operands of the first item
    stage_head_asm<NT>(qb, p.ld, p.S, aQ, wave, lane);
    stage_head_asm<NT>(gb, p.ldo, p.S, aG, wave, lane);
    stage_head_asm<NT>(ob, p.ldo, p.S, aO, wave, lane);
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = gfrag(qb + d, p.ld, trow, 2 * ks, lane);
        vf[ks] = gfrag(qb + 2 * d, p.ld, trow, 2 * ks, lane);
    }
    float lse_v = p.lse[so0 + trow];
#ifdef REID_ATTN_TRACE
#define PSTAMP(slot) do { if (g_attn_bwd_trace && threadIdx.x == 0) g_attn_bwd_trace[(size_t)item * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PSTAMP(slot) do { } while (0)
#endif
    for (;;) {
        PSTAMP(0);
        // everything in flight has landed: this item's Q / dO / O images (staged under the previous item's phase 2) and register operands
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (the compiler tracks the register operands itself and does not see the wait above: a use of each of them HERE makes it place its
        //  own wait here, where nothing else is in flight, rather than behind the K / V staging below)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" :: "v"(kf[ks]), "v"(vf[ks]));
        asm volatile("" :: "v"(lse_v));
        bar();                                       // ... for every wave; and nobody reads the previous item's K / V images any more
        PSTAMP(1);
        stage_head_asm<NT>(qb + d, p.ld, p.S, aK, wave, lane);          // this item's K / V images: they land under delta + phase 1
        stage_head_asm<NT>(qb + 2 * d, p.ld, p.S, aV, wave, lane);
        // ---- delta of this wave's query rows -> LDS, with the saved log-sum-exp
        {
            float dsum = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 gfk = row_frag_o(Gs, t0, ks, fo), ofk = row_frag_o(Os, t0, ks, fo);
#pragma unroll
                for (int j = 0; j < 8; ++j) dsum = fmaf(bf16_to_f32((bf16_t)gfk[j]), bf16_to_f32((bf16_t)ofk[j]), dsum);
            }
            dsum += __shfl_xor(dsum, 32, 64);
            if (lane < 32) {
                rowc[ti] = tile_ok ? -lse_v * LOG2E : -INFINITY;
                rowc[NT * 32 + ti] = tile_ok ? -dsum * 0.125f : 0.f;
            }
        }
        bar();                                       // rowc complete
        PSTAMP(2);
        // ---- phase 1: this wave's key tile (K, V of the tile in registers; Q / dO images)
        f32x16 dkt[2], dvt[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { dkt[dt][e] = 0.f; dvt[dt][e] = 0.f; }
        {
            auto scores = [&](int qt, f32x16& s, f32x16& dp) {
#pragma unroll
                for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    s = mfma32(row_frag_o(Qs, qt * 32, ks, fo), kf[ks], s);
                    dp = mfma32(row_frag_o(Gs, qt * 32, ks, fo), vf[ks], dp);
                }
            };
            f32x16 s, dp, s_n, dp_n;
            scores(0, s, dp);
            for (int qt = 0; qt < nt; ++qt) {
                if (qt + 1 < nt) scores(qt + 1, s_n, dp_n);          // next tile's matrix work first: it runs under this tile's exponentials
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 pf, df;
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const int g = 2 * s2 + g2;
                        const f32x4 nl = *(const f32x4*)(rowc + qt * 32 + 8 * g + h4);
                        const f32x4 nd = *(const f32x4*)(rowc + NT * 32 + qt * 32 + 8 * g + h4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int e = 4 * g + j;
                            const float pe = fast_exp2(fmaf(s[e], c, nl[j]));
                            pf[4 * g2 + j] = (short)f32_to_bf16(pe);
                            df[4 * g2 + j] = (short)f32_to_bf16(pe * fmaf(dp[e], 0.125f, nd[j]));
                        }
                    }
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        dvt[dt] = mfma32(col_frag_o(Gs, qt * 32 + 16 * s2, dt, fo), pf, dvt[dt]);
                        dkt[dt] = mfma32(col_frag_o(Qs, qt * 32 + 16 * s2, dt, fo), df, dkt[dt]);
                    }
                }
                s = s_n; dp = dp_n;
            }
        }
        PSTAMP(3);
        // this wave's own query-tile fragments and row constants for phase 2: the last reads of the Q / dO images and of rowc
        bf16x8 qf[4], gf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qf[ks] = row_frag_o(Qs, t0, ks, fo); gf[ks] = row_frag_o(Gs, t0, ks, fo); }
        const float nl = rowc[ti], nd = rowc[NT * 32 + ti];
        bf16_t* drow = p.dqkv + ((size_t)(item / p.heads) * p.S + trow) * p.lddqkv + (item % p.heads) * 64;
        bar();                                       // nobody reads the Q / dO / O images (or rowc) of this item any more
        const int nxt = item + (int)gridDim.x;
        const bool has_next = nxt < n_items;         // (workgroup-uniform)
        const bf16_t *qb_n = qb, *gb_n = gb, *ob_n = ob;
        size_t so_n = so0;
        if (has_next) item_ptrs(nxt, qb_n, gb_n, ob_n, so_n);
        if (has_next) {
            // the next item's Q / dO / O images into the freed regions: twelve DMA instructions per wave, then the counted wait
            stage_head_asm<NT>(qb_n, p.ld, p.S, aQ, wave, lane);
            stage_head_asm<NT>(gb_n, p.ldo, p.S, aG, wave, lane);
            stage_head_asm<NT>(ob_n, p.ldo, p.S, aO, wave, lane);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");       // everything older has landed: in particular this item's K / V images
            __builtin_amdgcn_sched_barrier(0);
        } else {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        bar();                                       // ... for every wave
        PSTAMP(4);
        // (outputs of phase 1 are stored only now: a store in front of the wait would have to be acknowledged before it passes)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            store_tile_row16(drow + d + dt * 32, tile_ok, dkt[dt], 1.0f, lane);
            store_tile_row16(drow + 2 * d + dt * 32, tile_ok, dvt[dt], 1.0f, lane);
        }
        PSTAMP(5);
        // ---- phase 2: this wave's query tile (Q, dO of the tile in registers; K / V images)
        f32x16 dqt[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) dqt[dt][e] = 0.f;
        {
            auto scores_t = [&](int kt, f32x16& s, f32x16& dp) {
#pragma unroll
                for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    s = mfma32(row_frag_o(Ks, kt * 32, ks, fo), qf[ks], s);
                    dp = mfma32(row_frag_o(Vs, kt * 32, ks, fo), gf[ks], dp);
                }
            };
            f32x16 s, dp;
            scores_t(0, s, dp);
            for (int kt = 0; kt < nt; ++kt) {
                if (kt > 0) scores_t(kt, s, dp);     // (not pipelined like phase 1: with the next item's operands in registers a second accumulator pair spills, and a scratch reload is a VMEM operation in the middle of the counted waits)
                const bool edge = (kt + 1) * 32 > p.S;
                bf16x8 df[2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int e = 8 * s2 + j;
                        float pe = fast_exp2(fmaf(s[e], c, nl));
                        if (edge) {
                            const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + h4;
                            pe = key < p.S ? pe : 0.f;
                        }
                        df[s2][j] = (short)f32_to_bf16(pe * fmaf(dp[e], 0.125f, nd));
                    }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) dqt[dt] = mfma32(col_frag_o(Ks, kt * 32 + 16 * s2, dt, fo), df[s2], dqt[dt]);
            }
        }
        PSTAMP(6);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) store_tile_row16(drow + dt * 32, tile_ok, dqt[dt], 1.0f, lane);
        PSTAMP(7);
        if (!has_next) break;
        // register operands of the next item: K and V rows of this wave's key tile and the saved log-sum-exp.  Fetched HERE, after the
        // phases (the compiler tracks these loads and would make every transposed LDS read of a phase wait for them); they are waited for
        // at the top of the loop together with the dQ stores above (~1 us, the only exposed memory latency of an item)
        bf16x8 kf_n[4], vf_n[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf_n[ks] = gfrag(qb_n + d, p.ld, trow, 2 * ks, lane);
            vf_n[ks] = gfrag(qb_n + 2 * d, p.ld, trow, 2 * ks, lane);
        }
        const float lse_n = p.lse[so_n + trow];                        // (nothing of this wave is in flight towards LDS: the last DMA was waited for above)
        item = nxt; qb = qb_n; gb = gb_n; ob = ob_n; so0 = so_n;
        lse_v = lse_n;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { kf[ks] = kf_n[ks]; vf[ks] = vf_n[ks]; }
    }
}

template <int NT>
int launch_fwd(const AttnParams& p, hipStream_t s) {
    constexpr int LDS = 2 * NT * 32 * 128;
    constexpr bool TP = NT >= 4;
    REID_MAX_LDS((attn_fwd_kernel<NT, TP>), LDS);
    hipLaunchKernelGGL((attn_fwd_kernel<NT, TP>), dim3(p.n_seq * p.heads), dim3(NT * 64), LDS, s, p);
    REID_CHECK_LAUNCH("reid_attn_fwd");
    return REID_OK;
}

template <int NT>
int launch_bwd(const AttnParams& p, hipStream_t s) {
    constexpr int LDS1 = 2 * NT * 32 * 128 + 2 * NT * 32 * 4;
    constexpr int LDS2 = 2 * NT * 32 * 128;
    REID_MAX_LDS((attn_bwd_dkv_kernel<NT>), LDS1);
    REID_MAX_LDS((attn_bwd_dq_kernel<NT>), LDS2);
    // REID_ATTN_BWD: 1 = the two-kernel form, 2 = one pass, one item per workgroup (default), 3 = one pass, persistent.  Masked / causal
    // attention: two kernels.  Measured (r04, 256 images x 12 heads, profiles/r04_attn_bwd.log): 292 / 257 / 257 us alone; inside the training
    // step 31.93 / 31.42 / 31.54 ms per step on one box -- the persistent form hides the image staging (top-of-item wait 1.0 us instead of
    // 5.5 us) but pays it back in barriers and DMA issue (3.1 us between the phases), and like every persistent kernel it keeps its CUs
    // from the side stream; the one-item form is the default.
    const int impl = reid_knob(KNOB_ATTN_BWD);
    if (!p.key_mask && !p.causal && impl != 1) {
        constexpr int LDSF = 4 * NT * 32 * 128 + 2 * NT * 32 * 4;
        constexpr int LDSP = 5 * NT * 32 * 128 + 2 * NT * 32 * 4;          // + the O image
        const int n_items = p.n_seq * p.heads;
        const int cus = reid_num_cus();
        const bool pers_ok = p.q_tiles <= 0 && (p.S + 31) / 32 == NT;
        if (pers_ok && impl == 3) {
            REID_MAX_LDS((attn_bwd_pers_kernel<NT>), LDSP);
            hipLaunchKernelGGL(attn_bwd_pers_kernel<NT>, dim3(n_items < cus ? n_items : cus), dim3(NT * 64), LDSP, s, p, n_items);
            REID_CHECK_LAUNCH("reid_attn_bwd(persistent)");
            return REID_OK;
        }
        REID_MAX_LDS((attn_bwd_fused_kernel<NT>), LDSF);
        hipLaunchKernelGGL(attn_bwd_fused_kernel<NT>, dim3(n_items), dim3(NT * 64), LDSF, s, p);
        REID_CHECK_LAUNCH("reid_attn_bwd(fused)");
        return REID_OK;
    }
    // dQ first: it also produces delta (rows of the query tiles that take part), which the dK/dV kernel reads
    hipLaunchKernelGGL(attn_bwd_dq_kernel<NT>, dim3(p.n_seq * p.heads), dim3(NT * 64), LDS2, s, p);
    REID_CHECK_LAUNCH("reid_attn_bwd(dq)");
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<NT>, dim3(p.n_seq * p.heads), dim3(NT * 64), LDS1, s, p);
    REID_CHECK_LAUNCH("reid_attn_bwd(dkv)");
    return REID_OK;
}

int check_common(const char* name, const void* qkv, int ld, int n_seq, int S, int heads) {
    REID_CHECK_ARG(qkv != nullptr, "%s: null qkv", name);
    REID_CHECK_ARG(n_seq > 0 && heads > 0 && S > 0 && S <= 224, "%s: S=%d must be in 1..224 (n_seq=%d heads=%d)", name, S, n_seq, heads);
    REID_CHECK_ARG(ld >= 3 * heads * 64 && ld % 8 == 0, "%s: ld=%d must be >= 3*heads*64 and a multiple of 8", name, ld);
    return REID_OK;
}

}  // namespace

#ifdef REID_ATTN_TRACE
extern "C" void reid_debug_attn_bwd_trace(void* buf) {
    unsigned long long* ptr = (unsigned long long*)buf;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_bwd_trace), &ptr, sizeof(ptr));
}
#endif

#define DISPATCH_NT(nt, fn, ...)                \
    switch (nt) {                               \
        case 1: return fn<1>(__VA_ARGS__);      \
        case 2: return fn<2>(__VA_ARGS__);      \
        case 3: return fn<3>(__VA_ARGS__);      \
        case 4: return fn<4>(__VA_ARGS__);      \
        case 5: return fn<5>(__VA_ARGS__);      \
        case 6: return fn<6>(__VA_ARGS__);      \
        default: return fn<7>(__VA_ARGS__);     \
    }

extern "C" int reid_attn_fwd(const void* qkv, int32_t ld, const uint8_t* key_mask, void* out, int32_t ldo, float* lse,
                             int32_t n_seq, int32_t S, int32_t heads, int32_t causal, int32_t q_tiles, void* stream) {
    int rc = check_common("reid_attn_fwd", qkv, ld, n_seq, S, heads);
    if (rc) return rc;
    REID_CHECK_ARG(out && ldo >= heads * 64 && ldo % 4 == 0, "reid_attn_fwd: out/ldo");
    AttnParams p{(const bf16_t*)qkv, ld, key_mask, (bf16_t*)out, ldo, lse, nullptr, nullptr, 0, nullptr, n_seq, S, heads, causal, q_tiles, 0};
    if (reid_knob(KNOB_ATTN_DBG) > 0) p.dbg = reid_knob(KNOB_ATTN_DBG);
    DISPATCH_NT((S + 31) / 32, launch_fwd, p, (hipStream_t)stream)
}

extern "C" int reid_attn_bwd(const void* qkv, int32_t ld, const uint8_t* key_mask, const void* out, const void* dout,
                             int32_t ldo, const float* lse, void* dqkv, int32_t lddqkv, float* delta_ws, int32_t n_seq,
                             int32_t S, int32_t heads, int32_t causal, int32_t q_tiles, void* stream) {
    int rc = check_common("reid_attn_bwd", qkv, ld, n_seq, S, heads);
    if (rc) return rc;
    REID_CHECK_ARG(out && dout && lse && dqkv && delta_ws, "reid_attn_bwd: null pointer");
    REID_CHECK_ARG(ldo >= heads * 64 && ldo % 8 == 0 && lddqkv >= 3 * heads * 64 && lddqkv % 4 == 0, "reid_attn_bwd: ldo/lddqkv");
    AttnParams p{(const bf16_t*)qkv, ld, key_mask, (bf16_t*)out, ldo, (float*)lse, (const bf16_t*)dout, (bf16_t*)dqkv, lddqkv,
                 delta_ws, n_seq, S, heads, causal, q_tiles, 0};
    DISPATCH_NT((S + 31) / 32, launch_bwd, p, (hipStream_t)stream)
}
