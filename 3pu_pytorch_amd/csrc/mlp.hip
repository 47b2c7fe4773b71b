// mlp.hip -- per-point linear layers of a Level on the fp32 matrix cores (inference, gfx950).
//
// The reference runs every 1x1 convolution as its own cuDNN call followed by separate bias / ReLU
// kernels (network/layers.py:115-204, network/upsampler.py:293-369).  Two shapes dominate what is
// left of a Level after the DenseEdgeConv blocks:
//   * the "prep" convolutions 84 / 144 / 204 -> 24 + ReLU (upsampler.py:298,303,308): skinny (24
//     outputs), bound by streaming the input rows; a vendor GEMM tile wastes most of its width;
//   * the regressor tail (upsampler.py:363-372): relu(a_i + c_j) -> 128 -> 128 -> 64 -> 3 + residual
//     for each of the r replicas of a point, where a_i = W_x x_i + b is the per-point half of
//     up_layer1 and c_j = W_c code_j the per-replica half: three GEMMs and five elementwise passes
//     over (B, N*r, 128) tensors in the unfused form.
//
// Both kernels use v_mfma_f32_16x16x4_f32 with the weights as A operand (16 outputs x 4 k) and the
// activations as B operand (4 k x 16 points): lane l supplies point l & 15, k slot l >> 4, and holds
// of the 16x16 result the outputs 4 (l >> 4) + r, r = 0..3, of ITS point.  The k slots of the four
// MFMAs of a 16-channel slab are permuted (MFMA j, slot q <-> channel 16 s + 4 q + j) so that
//   - a lane fetches its four B (and A) values of a slab as ONE float4, and
//   - the accumulator tile t of a layer IS the B operand of slab t of the next layer (bias / ReLU
//     applied in place): the whole tail runs register to register.
// fp32 MFMA accumulates in k order with one rounding per product, so results differ from a vendor
// GEMM only by the summation order (tests compare at 1e-5).
#include "tpu3_dev.h"

#include <cstdlib>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4f ld4(const float *p) { return *(const v4f *)p; }

// ---------------------------------------------------------------------------------------------
// y[m, cout] = act(x[m, cin] W^T + b), cout <= 32
// ---------------------------------------------------------------------------------------------
struct LinArgs {
    long m;
    int cin, cout, xs, ys, relu;
    const float *x, *w, *b;
    float *y;
};

constexpr int LS_CIN_MAX = 320;           // weights staged in LDS: 32 x (320 + 4) floats = 41 KB

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// four input channels of a row: fp32 rows, or (XH) the fp16 rows of a feature buffer stored as TPU3_STORE_F16 -- the
// value that enters the matrix instruction is the same fp16 number either way
template <bool XH>
__device__ __forceinline__ v4f ldx4(const float *row_f32, long off)
{
    if constexpr (XH) {
        const h4 h = *(const h4 *)((const _Float16 *)row_f32 + off);
        return (v4f){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    } else {
        return ld4(row_f32 + off);
    }
}

__device__ __forceinline__ h8 cat_h8(v4f lo, v4f hi)
{
    h8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[j] = (_Float16)lo[j];
        r[4 + j] = (_Float16)hi[j];
    }
    return r;
}

// fp16-operand flavour (TPU3_MFMA_F16): two 16-channel slabs per v_mfma_f32_16x16x32_f16.  The eight k slots
// of lane (pt, q) are the channels {16 s + 4 q + j} u {16 (s+1) + 4 q + j}, j < 4 -- the SAME two float4 pieces
// the fp32 flavour fetches, converted in registers; weights likewise.  fp32 accumulate, fp32 rows in memory.
template <int TOUT, bool XH = false>
__global__ __launch_bounds__(256) void linear_small_f16_kernel(LinArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [16 * TOUT][cin + 4], zero padded (fp32)
    const int S = a.cin + 4;
    for (int i = threadIdx.x; i < 16 * TOUT * (a.cin >> 2); i += blockDim.x) {
        const int o = i / (a.cin >> 2), c4 = i - o * (a.cin >> 2);
        const v4f v = o < a.cout ? ld4(a.w + (size_t)o * a.cin + 4 * c4) : (v4f){0.f, 0.f, 0.f, 0.f};
        *(v4f *)(wl + o * S + 4 * c4) = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 15, q = lane >> 4;
    const long ntiles = (a.m + 15) >> 4;
    const int nslab = (a.cin + 15) >> 4;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        const long row = tile * 16 + pt;
        const long xrow = (row < a.m ? row : a.m - 1) * a.xs;          // (elements of the stored type)
        v4f acc[TOUT];
#pragma unroll
        for (int t = 0; t < TOUT; ++t)
            acc[t] = zero;
        for (int s0 = 0; s0 < nslab; s0 += 8) {
            v4f bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ch = 16 * (s0 + u) + 4 * q;
                bv[u] = ldx4<XH>(a.x, xrow + (ch < a.cin ? ch : 0));        // cin % 4 == 0
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const int ch0 = 16 * (s0 + u) + 4 * q, ch1 = ch0 + 16;
                const bool ok0 = ch0 < a.cin, ok1 = ch1 < a.cin;
                const h8 b = cat_h8(ok0 ? bv[u] : zero, ok1 ? bv[u + 1] : zero);
#pragma unroll
                for (int t = 0; t < TOUT; ++t) {
                    const float *wr = wl + (16 * t + pt) * S;
                    const h8 av = cat_h8(ok0 ? *(const v4f *)(wr + ch0) : zero, ok1 ? *(const v4f *)(wr + ch1) : zero);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b, acc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < TOUT; ++t) {
            const int o = 16 * t + 4 * q;                       // cout % 4 == 0
            if (o < a.cout && row < a.m) {
                v4f v = acc[t];
                if (a.b)
                    v += ld4(a.b + o);
                if (a.relu) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                *(v4f *)(a.y + row * a.ys + o) = v;
            }
        }
    }
}

// fp16-operand flavour for WIDE layers (32 < cout <= 128: the per-point half of up_layer1, 264 -> 128, in
// mlp_precision "f16" -- a library fp16 GEMM until round 2).  Same tiling and k-slot permutation as
// linear_small_f16_kernel; the weights are converted ONCE per workgroup and staged in LDS as fp16
// ([16 TOUT][cin + 8] halves: 70 KB for 128 x 264, two workgroups per compute unit), so the A operand of a
// v_mfma_f32_16x16x32_f16 is two 8-byte LDS reads and no conversions.  fp32 accumulate, fp32 rows in memory.
template <int TOUT, bool XH = false>
__global__ __launch_bounds__(256) void linear_wide_f16_kernel(LinArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float wl[];
    _Float16 *wh = (_Float16 *)wl;                                   // [16 * TOUT][S]
    const int S = a.cin + 8;
    for (int i = threadIdx.x; i < 16 * TOUT * (a.cin >> 2); i += blockDim.x) {
        const int o = i / (a.cin >> 2), c4 = i - o * (a.cin >> 2);
        const v4f v = o < a.cout ? ld4(a.w + (size_t)o * a.cin + 4 * c4) : (v4f){0.f, 0.f, 0.f, 0.f};
        h4 h;
        h[0] = (_Float16)v[0]; h[1] = (_Float16)v[1]; h[2] = (_Float16)v[2]; h[3] = (_Float16)v[3];
        *(h4 *)(wh + o * S + 4 * c4) = h;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 15, q = lane >> 4;
    const long ntiles = (a.m + 15) >> 4;
    const int nslab = (a.cin + 15) >> 4;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    const h4 hzero = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        const long row = tile * 16 + pt;
        const long xrow = (row < a.m ? row : a.m - 1) * a.xs;          // (elements of the stored type)
        v4f acc[TOUT];
#pragma unroll
        for (int t = 0; t < TOUT; ++t)
            acc[t] = zero;
        for (int s0 = 0; s0 < nslab; s0 += 8) {
            v4f bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ch = 16 * (s0 + u) + 4 * q;
                bv[u] = ldx4<XH>(a.x, xrow + (ch < a.cin ? ch : 0));        // cin % 4 == 0
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                const int ch0 = 16 * (s0 + u) + 4 * q, ch1 = ch0 + 16;
                const bool ok0 = ch0 < a.cin, ok1 = ch1 < a.cin;
                const h8 b = cat_h8(ok0 ? bv[u] : zero, ok1 ? bv[u + 1] : zero);
#pragma unroll
                for (int t = 0; t < TOUT; ++t) {
                    const _Float16 *wr = wh + (16 * t + pt) * S;
                    const h4 a0 = ok0 ? *(const h4 *)(wr + ch0) : hzero, a1 = ok1 ? *(const h4 *)(wr + ch1) : hzero;
                    h8 av;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        av[j] = a0[j];
                        av[4 + j] = a1[j];
                    }
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b, acc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < TOUT; ++t) {
            const int o = 16 * t + 4 * q;                       // cout % 4 == 0
            if (o < a.cout && row < a.m) {
                v4f v = acc[t];
                if (a.b)
                    v += ld4(a.b + o);
                if (a.relu) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                *(v4f *)(a.y + row * a.ys + o) = v;
            }
        }
    }
}

template <int TOUT>
__global__ __launch_bounds__(256) void linear_small_kernel(LinArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [16 * TOUT][cin + 4], zero padded
    const int S = a.cin + 4;
    for (int i = threadIdx.x; i < 16 * TOUT * (a.cin >> 2); i += blockDim.x) {
        const int o = i / (a.cin >> 2), c4 = i - o * (a.cin >> 2);
        const v4f v = o < a.cout ? ld4(a.w + (size_t)o * a.cin + 4 * c4) : (v4f){0.f, 0.f, 0.f, 0.f};
        *(v4f *)(wl + o * S + 4 * c4) = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 15, q = lane >> 4;
    const long ntiles = (a.m + 15) >> 4;
    const int nslab = (a.cin + 15) >> 4;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        const long row = tile * 16 + pt;
        const float *xr = a.x + (row < a.m ? row : a.m - 1) * a.xs;
        v4f acc[TOUT];
#pragma unroll
        for (int t = 0; t < TOUT; ++t)
            acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
        // eight slabs per trip: all their row loads are in flight together (the kernel is bound by
        // streaming x); slabs past cin contribute zero operands
        for (int s0 = 0; s0 < nslab; s0 += 8) {
            v4f bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ch = 16 * (s0 + u) + 4 * q;
                bv[u] = ld4(xr + (ch < a.cin ? ch : 0));        // cin % 4 == 0
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ch = 16 * (s0 + u) + 4 * q;
                const bool ok = ch < a.cin;
                const v4f b = ok ? bv[u] : (v4f){0.f, 0.f, 0.f, 0.f};
                v4f av[TOUT];
#pragma unroll
                for (int t = 0; t < TOUT; ++t)
                    av[t] = *(const v4f *)(wl + (16 * t + pt) * S + (ok ? ch : 0));
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < TOUT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][j], b[j], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < TOUT; ++t) {
            const int o = 16 * t + 4 * q;                       // cout % 4 == 0
            if (o < a.cout && row < a.m) {
                v4f v = acc[t];
                if (a.b)
                    v += ld4(a.b + o);
                if (a.relu) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                *(v4f *)(a.y + row * a.ys + o) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// regressor tail: out[(i*r + j), 0..2] = W4 relu(W3 relu(W2 relu(a_i + c_j) + b2) + b3) + b4 + res_i
// a (m,128) incl. the bias of up_layer1; c (r,128); W2 (128,128); W3 (64,128); W4 (3,64)
// ---------------------------------------------------------------------------------------------
constexpr int RT_C1 = 128, RT_C2 = 128, RT_C3 = 64, RT_C4 = 3;
constexpr int RT_S = 132;                 // LDS row stride of the 128-wide weight rows (floats)
constexpr int RT_S4 = 68;                 // ... of the 64-wide rows of W4 (padded to 16 rows)
constexpr int RT_RMAX = 4;                // replicas per point

struct TailArgs {
    long m;
    int r;
    const float *a, *c, *w2, *b2, *w3, *b3, *w4, *b4, *res;
    float *out;
};

constexpr size_t rt_lds_floats()
{
    return (size_t)RT_C2 * RT_S + RT_C3 * RT_S + 16 * RT_S4 + RT_C2 + RT_C3 + 16 + RT_RMAX * RT_C1;
}

__global__ __launch_bounds__(512) void regress_tail_kernel(TailArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *w2 = lds;                              // [128][132]
    float *w3 = w2 + RT_C2 * RT_S;                // [64][132]
    float *w4 = w3 + RT_C3 * RT_S;                // [16][68], rows >= 3 zero
    float *b2 = w4 + 16 * RT_S4;                  // [128]
    float *b3 = b2 + RT_C2;                       // [64]
    float *b4 = b3 + RT_C3;                       // [16]
    float *cc = b4 + 16;                          // [r][128]
    const int tid = threadIdx.x;
    for (int i = tid; i < RT_C2 * RT_C1; i += blockDim.x)
        w2[(i >> 7) * RT_S + (i & 127)] = a.w2[i];
    for (int i = tid; i < RT_C3 * RT_C2; i += blockDim.x)
        w3[(i >> 7) * RT_S + (i & 127)] = a.w3[i];
    for (int i = tid; i < 16 * RT_C3; i += blockDim.x)
        w4[(i >> 6) * RT_S4 + (i & 63)] = (i >> 6) < RT_C4 ? a.w4[i] : 0.f;
    for (int i = tid; i < RT_C2; i += blockDim.x)
        b2[i] = a.b2[i];
    for (int i = tid; i < RT_C3; i += blockDim.x)
        b3[i] = a.b3[i];
    for (int i = tid; i < 16; i += blockDim.x)
        b4[i] = i < RT_C4 ? a.b4[i] : 0.f;
    for (int i = tid; i < a.r * RT_C1; i += blockDim.x)
        cc[i] = a.c[i];
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int pt = lane & 15, q = lane >> 4;
    const long ntiles = (a.m + 15) >> 4;
    for (long tile = (long)blockIdx.x * nw + wave; tile < ntiles; tile += (long)gridDim.x * nw) {
        const long row = tile * 16 + pt;
        const long rowc = row < a.m ? row : a.m - 1;
        // a_i in B-operand form: slab s holds channels 16 s + 4 q .. + 3
        v4f av[RT_C1 / 16];
#pragma unroll
        for (int s = 0; s < RT_C1 / 16; ++s)
            av[s] = ld4(a.a + rowc * RT_C1 + 16 * s + 4 * q);
        float rx = 0.f, ry = 0.f, rz = 0.f;
        if (q == 0) {
            rx = a.res[rowc * 3 + 0]; ry = a.res[rowc * 3 + 1]; rz = a.res[rowc * 3 + 2];
        }
        for (int j = 0; j < a.r; ++j) {
            // h0 = relu(a + c_j)
            v4f h0[RT_C1 / 16];
#pragma unroll
            for (int s = 0; s < RT_C1 / 16; ++s) {
                const v4f c = ld4(cc + j * RT_C1 + 16 * s + 4 * q);
                v4f v = av[s] + c;
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                h0[s] = v;
            }
            // layer 2: 128 -> 128, ReLU
            v4f h1[RT_C2 / 16];
#pragma unroll
            for (int t = 0; t < RT_C2 / 16; ++t)
                h1[t] = (v4f){0.f, 0.f, 0.f, 0.f};
            // (consecutive MFMAs go to different accumulator tiles: a tile's own chain is dependent)
#pragma unroll
            for (int s = 0; s < RT_C1 / 16; ++s) {
                v4f w[RT_C2 / 16];
#pragma unroll
                for (int t = 0; t < RT_C2 / 16; ++t)
                    w[t] = ld4(w2 + (16 * t + pt) * RT_S + 16 * s + 4 * q);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int t = 0; t < RT_C2 / 16; ++t)
                        h1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][k], h0[s][k], h1[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);      // keep the next slabs' LDS reads from piling up in VGPRs
            }
#pragma unroll
            for (int t = 0; t < RT_C2 / 16; ++t) {
                v4f v = h1[t] + ld4(b2 + 16 * t + 4 * q);
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                h1[t] = v;
            }
            // layer 3: 128 -> 64, ReLU
            v4f h2[RT_C3 / 16];
#pragma unroll
            for (int t = 0; t < RT_C3 / 16; ++t)
                h2[t] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < RT_C2 / 16; ++s) {
                v4f w[RT_C3 / 16];
#pragma unroll
                for (int t = 0; t < RT_C3 / 16; ++t)
                    w[t] = ld4(w3 + (16 * t + pt) * RT_S + 16 * s + 4 * q);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int t = 0; t < RT_C3 / 16; ++t)
                        h2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][k], h1[s][k], h2[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < RT_C3 / 16; ++t) {
                v4f v = h2[t] + ld4(b3 + 16 * t + 4 * q);
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                h2[t] = v;
            }
            // layer 4: 64 -> 3 (one output tile, rows >= 3 are zero weights), bias, residual
            v4f o = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < RT_C3 / 16; ++s) {
                const v4f w = ld4(w4 + pt * RT_S4 + 16 * s + 4 * q);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    o = __builtin_amdgcn_mfma_f32_16x16x4f32(w[k], h2[s][k], o, 0, 0, 0);
            }
            if (q == 0 && row < a.m) {          // lanes 0-15 hold outputs 0..3 of their point
                float *dst = a.out + (row * a.r + j) * 3;
                dst[0] = (o.x + b4[0]) + rx;
                dst[1] = (o.y + b4[1]) + ry;
                dst[2] = (o.z + b4[2]) + rz;
            }
        }
    }
}

// fp16-operand flavour of the tail (TPU3_MFMA_F16): the same register-to-register dataflow with two
// accumulator tiles of a layer forming the B operand of ONE v_mfma_f32_16x16x32_f16 of the next (k slots of
// lane (pt, q): outputs 16 t + 4 q + j of tiles t = 2 S and 2 S + 1).  Weights in LDS as fp16 (50 KB instead
// of 100): 32 + 16 + 2 MFMAs per 16 rows instead of 256 + 128 + 16.
constexpr int RH_S = 136;                 // fp16 row stride of the 128-wide weight rows (16-byte aligned, skewed)
constexpr int RH_S4 = 72;

constexpr size_t rh_lds_bytes()
{
    return ((size_t)RT_C2 * RH_S + RT_C3 * RH_S + 16 * RH_S4) * 2 + (RT_C2 + RT_C3 + 16 + RT_RMAX * RT_C1) * 4;
}

// the eight weights of output row `m` a lane needs for slab pair S: channels 32 S + {4 q + j, 16 + 4 q + j}
__device__ __forceinline__ h8 rh_w8(const _Float16 *w, int stride, int m, int S, int q)
{
    const h4 lo = *(const h4 *)(w + m * stride + 32 * S + 4 * q), hi = *(const h4 *)(w + m * stride + 32 * S + 16 + 4 * q);
    h8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[j] = lo[j];
        r[4 + j] = hi[j];
    }
    return r;
}

__global__ __launch_bounds__(512) void regress_tail_f16_kernel(TailArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    _Float16 *w2 = (_Float16 *)ldsb;              // [128][136]
    _Float16 *w3 = w2 + RT_C2 * RH_S;             // [64][136]
    _Float16 *w4 = w3 + RT_C3 * RH_S;             // [16][72], rows >= 3 zero
    float *b2 = (float *)(w4 + 16 * RH_S4);       // [128]
    float *b3 = b2 + RT_C2;                       // [64]
    float *b4 = b3 + RT_C3;                       // [16]
    float *cc = b4 + 16;                          // [r][128]
    const int tid = threadIdx.x;
    for (int i = tid; i < RT_C2 * RT_C1; i += blockDim.x)
        w2[(i >> 7) * RH_S + (i & 127)] = (_Float16)a.w2[i];
    for (int i = tid; i < RT_C3 * RT_C2; i += blockDim.x)
        w3[(i >> 7) * RH_S + (i & 127)] = (_Float16)a.w3[i];
    for (int i = tid; i < 16 * RT_C3; i += blockDim.x)
        w4[(i >> 6) * RH_S4 + (i & 63)] = (_Float16)((i >> 6) < RT_C4 ? a.w4[i] : 0.f);
    for (int i = tid; i < RT_C2; i += blockDim.x)
        b2[i] = a.b2[i];
    for (int i = tid; i < RT_C3; i += blockDim.x)
        b3[i] = a.b3[i];
    for (int i = tid; i < 16; i += blockDim.x)
        b4[i] = i < RT_C4 ? a.b4[i] : 0.f;
    for (int i = tid; i < a.r * RT_C1; i += blockDim.x)
        cc[i] = a.c[i];
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int pt = lane & 15, q = lane >> 4;
    const long ntiles = (a.m + 15) >> 4;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    for (long tile = (long)blockIdx.x * nw + wave; tile < ntiles; tile += (long)gridDim.x * nw) {
        const long row = tile * 16 + pt;
        const long rowc = row < a.m ? row : a.m - 1;
        v4f av[RT_C1 / 16];
#pragma unroll
        for (int s = 0; s < RT_C1 / 16; ++s)
            av[s] = ld4(a.a + rowc * RT_C1 + 16 * s + 4 * q);
        float rx = 0.f, ry = 0.f, rz = 0.f;
        if (q == 0) {
            rx = a.res[rowc * 3 + 0]; ry = a.res[rowc * 3 + 1]; rz = a.res[rowc * 3 + 2];
        }
        for (int j = 0; j < a.r; ++j) {
            // h0 = relu(a + c_j), as the B operands of the four slab pairs
            h8 h0[RT_C1 / 32];
#pragma unroll
            for (int S = 0; S < RT_C1 / 32; ++S) {
                v4f v[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    v[u] = av[2 * S + u] + ld4(cc + j * RT_C1 + 16 * (2 * S + u) + 4 * q);
                    v[u].x = fmaxf(v[u].x, 0.f); v[u].y = fmaxf(v[u].y, 0.f);
                    v[u].z = fmaxf(v[u].z, 0.f); v[u].w = fmaxf(v[u].w, 0.f);
                }
                h0[S] = cat_h8(v[0], v[1]);
            }
            // layer 2: 128 -> 128, ReLU
            v4f t1[RT_C2 / 16];
#pragma unroll
            for (int t = 0; t < RT_C2 / 16; ++t)
                t1[t] = ld4(b2 + 16 * t + 4 * q);
#pragma unroll
            for (int S = 0; S < RT_C1 / 32; ++S) {
#pragma unroll
                for (int t = 0; t < RT_C2 / 16; ++t)
                    t1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh_w8(w2, RH_S, 16 * t + pt, S, q), h0[S], t1[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);      // keep the next slab pairs' LDS reads from piling up in VGPRs
            }
            h8 h1[RT_C2 / 32];
#pragma unroll
            for (int S = 0; S < RT_C2 / 32; ++S) {
                v4f v[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    v[u] = t1[2 * S + u];
                    v[u].x = fmaxf(v[u].x, 0.f); v[u].y = fmaxf(v[u].y, 0.f);
                    v[u].z = fmaxf(v[u].z, 0.f); v[u].w = fmaxf(v[u].w, 0.f);
                }
                h1[S] = cat_h8(v[0], v[1]);
            }
            // layer 3: 128 -> 64, ReLU
            v4f t2[RT_C3 / 16];
#pragma unroll
            for (int t = 0; t < RT_C3 / 16; ++t)
                t2[t] = ld4(b3 + 16 * t + 4 * q);
#pragma unroll
            for (int S = 0; S < RT_C2 / 32; ++S) {
#pragma unroll
                for (int t = 0; t < RT_C3 / 16; ++t)
                    t2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh_w8(w3, RH_S, 16 * t + pt, S, q), h1[S], t2[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // layer 4: 64 -> 3 (one output tile, rows >= 3 are zero weights), bias, residual
            v4f o = zero;
#pragma unroll
            for (int S = 0; S < RT_C3 / 32; ++S) {
                v4f v[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    v[u] = t2[2 * S + u];
                    v[u].x = fmaxf(v[u].x, 0.f); v[u].y = fmaxf(v[u].y, 0.f);
                    v[u].z = fmaxf(v[u].z, 0.f); v[u].w = fmaxf(v[u].w, 0.f);
                }
                o = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh_w8(w4, RH_S4, pt, S, q), cat_h8(v[0], v[1]), o, 0, 0, 0);
            }
            if (q == 0 && row < a.m) {          // lanes 0-15 hold outputs 0..3 of their point
                float *dst = a.out + (row * a.r + j) * 3;
                dst[0] = (o.x + b4[0]) + rx;
                dst[1] = (o.y + b4[1]) + ry;
                dst[2] = (o.z + b4[2]) + rz;
            }
        }
    }
}


// (r6) SPLIT-bf16 flavour of the tail (the default; tpu3_split_bf16 / TPU3_SPLIT_BF16=0 select the fp32 kernel): fp32
// arithmetic on the bf16 matrix pipe.  Every fp32 operand is split EXACTLY into three bf16 terms, x = x1 + x2 + x3
// (8 + 8 + 8 mantissa bits: x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2), each subtraction exact), and a
// product w x becomes the six partial products whose magnitude can reach the fp32 result's last bits (w1 x1, w1 x2,
// w2 x1, w1 x3, w2 x2, w3 x1; the three left out are below 2^-32 relative), accumulated in fp32 by
// v_mfma_f32_16x16x32_bf16 -- 16x the fp32 MFMA's rate per instruction, 6 instructions for 8 of the 16x16x4 ones, and
// the splits are VALU work the matrix pipe does not wait for.  Same register-to-register dataflow as the fp16 flavour
// above (two accumulator tiles = the B operand of one slab pair); the weights are split once per workgroup into three
// bf16 arrays in LDS (147 KB for layers 2 and 3, operand-major: rb_slot); the 64 -> 3 layer stays on v_mfma_f32_16x16x4_f32 (16 instructions,
// and its weights would not fit).  The result is NOT bit-identical to the fp32 kernel's -- products are exact in both,
// the fp32 accumulation order differs -- but as accurate (probe: 1.2x / 1.0x the fp32 kernel's error against fp64).
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
constexpr int RB_S = 128;                 // bf16 elements per weight row; in LDS the rows are stored OPERAND-major (rb_slot): a lane's eight k
                                          // values of an (output tile, slab pair) are 16 contiguous bytes at 16 * lane -- one conflict-free
                                          // ds_read_b128 (two ds_read_b64 per operand reach a fifth of their rate at two waves per SIMD)
constexpr int RB_S4 = 68;                 // fp32 row stride of W4, FOUR rows (row 3 = zeros, read by every point >= 3)

constexpr size_t rb_lds_bytes()
{
    return ((size_t)3 * RT_C2 * RB_S + 3 * RT_C3 * RB_S) * 2 + (4 * RB_S4 + RT_C2 + RT_C3 + 16 + RT_RMAX * RT_C1) * 4;
}

// LDS position of weight (output row m, input channel ch) of a 128-input layer: [output tile][slab pair S][k quad q][row pt][8],
// the eight being the lane's k slots of the slab pair: channels 32 S + 4 q + (0..3) and 32 S + 16 + 4 q + (0..3) -- the two
// accumulator quads the B operand is made of
__device__ __forceinline__ int rb_slot(int m, int ch)
{
    const int S = ch >> 5, w = ch & 31, half = w >> 4, q = (w & 15) >> 2, j = w & 3;
    return ((((m >> 4) * (RB_S / 32) + S) * 4 + q) * 16 + (m & 15)) * 8 + half * 4 + j;
}

__device__ __forceinline__ void rb_split3(const v4f lo, const v4f hi, bf8 &t1, bf8 &t2, bf8 &t3)
{
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = j < 4 ? lo[j] : hi[j - 4];
        const __bf16 a = (__bf16)x;
        const float r1 = x - (float)a;
        const __bf16 b = (__bf16)r1;
        const float r2 = r1 - (float)b;
        t1[j] = a; t2[j] = b; t3[j] = (__bf16)r2;
    }
}

// the six partial products of one (output tile, slab pair), smallest first
#define RB_MFMA6(acc, A1, A2, A3, X1, X2, X3)                                        \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A3, X1, acc, 0, 0, 0);             \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, X3, acc, 0, 0, 0);             \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, X2, acc, 0, 0, 0);             \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, X1, acc, 0, 0, 0);             \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, X2, acc, 0, 0, 0);             \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, X1, acc, 0, 0, 0);

__global__ __launch_bounds__(512) void regress_tail_sb_kernel(TailArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    __bf16 *w2 = (__bf16 *)ldsb;                  // [3][128 x 128 in rb_slot order]
    __bf16 *w3 = w2 + 3 * RT_C2 * RB_S;           // [3][64 x 128 in rb_slot order]
    float *w4 = (float *)(w3 + 3 * RT_C3 * RB_S); // [4][68] fp32, row 3 zero
    float *b2 = w4 + 4 * RB_S4;                   // [128]
    float *b3 = b2 + RT_C2;                       // [64]
    float *b4 = b3 + RT_C3;                       // [16]
    float *cc = b4 + 16;                          // [r][128]
    const int tid = threadIdx.x;
    auto stage3 = [&](__bf16 *dst, int rows, const float *src, int i) __attribute__((always_inline)) {
        const float x = src[i];
        const __bf16 p1 = (__bf16)x;
        const float r1 = x - (float)p1;
        const __bf16 p2 = (__bf16)r1;
        const int o = rb_slot(i >> 7, i & 127);
        dst[o] = p1;
        dst[rows * RB_S + o] = p2;
        dst[2 * rows * RB_S + o] = (__bf16)(r1 - (float)p2);
    };
    for (int i = tid; i < RT_C2 * RT_C1; i += blockDim.x)
        stage3(w2, RT_C2, a.w2, i);
    for (int i = tid; i < RT_C3 * RT_C2; i += blockDim.x)
        stage3(w3, RT_C3, a.w3, i);
    for (int i = tid; i < 4 * RT_C3; i += blockDim.x)
        w4[(i >> 6) * RB_S4 + (i & 63)] = (i >> 6) < RT_C4 ? a.w4[i] : 0.f;
    for (int i = tid; i < RT_C2; i += blockDim.x)
        b2[i] = a.b2[i];
    for (int i = tid; i < RT_C3; i += blockDim.x)
        b3[i] = a.b3[i];
    for (int i = tid; i < 16; i += blockDim.x)
        b4[i] = i < RT_C4 ? a.b4[i] : 0.f;
    for (int i = tid; i < a.r * RT_C1; i += blockDim.x)
        cc[i] = a.c[i];
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int pt = lane & 15, q = lane >> 4;
    const long ntiles = (a.m + 15) >> 4;
    // one LANE base per array (16 bytes per lane), laundered: folded back into "w2 + a large constant" every operand
    // would get its own address register (offsets beyond the 16-bit immediate), ~60 of them spilled around the loop
    auto lane_base = [&](const __bf16 *w) __attribute__((always_inline)) {
        uint32_t o = (uint32_t)(uintptr_t)(w + 8 * lane);
        asm volatile("" : "+v"(o));
        return (const __attribute__((address_space(3))) __bf16 *)(uintptr_t)o;
    };
    const auto w2a = lane_base(w2), w2b = lane_base(w2 + RT_C2 * RB_S), w2c = lane_base(w2 + 2 * RT_C2 * RB_S);
    const auto w3a = lane_base(w3), w3b = lane_base(w3 + RT_C3 * RB_S), w3c = lane_base(w3 + 2 * RT_C3 * RB_S);
    // operand of output tile t, slab pair S: the lane's eight k slots, one ds_read_b128
    auto w8 = [](const __attribute__((address_space(3))) __bf16 *base, int t, int S) __attribute__((always_inline)) {
        return *(const __attribute__((address_space(3))) bf8 *)(base + (t * (RB_S / 32) + S) * 512);
    };
    // The tile's rows (32 registers) and residuals are fetched a TILE ahead: into the registers the second layer has
    // just finished with, while the third and fourth run (for more than two replicas the row is read again per pair).
    const long tstep = (long)gridDim.x * nw;
    v4f av[RT_C1 / 16];
    float rx = 0.f, ry = 0.f, rz = 0.f;
    auto fetch = [&](long tile, bool res_too) __attribute__((always_inline)) {
        const long r = min(tile * 16 + pt, a.m - 1);
#pragma unroll
        for (int s = 0; s < RT_C1 / 16; ++s)
            av[s] = ld4(a.a + r * RT_C1 + 16 * s + 4 * q);
        if (res_too && q == 0) {
            rx = a.res[r * 3 + 0]; ry = a.res[r * 3 + 1]; rz = a.res[r * 3 + 2];
        }
    };
    {
        const long first = (long)blockIdx.x * nw + wave;
        if (first < ntiles)
            fetch(first, true);
    }
    for (long tile = (long)blockIdx.x * nw + wave; tile < ntiles; tile += tstep) {
        const long row = tile * 16 + pt;
        const float cx = rx, cy = ry, cz = rz;
        // TWO replicas of a point at a time (the step ratio is 2: all of them): relu(a + c_j) differs, the weights do
        // not, so one A operand read from LDS feeds both (0.25 instead of 0.5 ds_read_b128-equivalents per MFMA: LDS
        // was this kernel's bound), and the six products of a (tile, slab pair) run PRODUCT-major over a pair of
        // output tiles x two replicas -- four independent accumulators between two MFMAs on the same one.
        for (int j0 = 0; j0 < a.r; j0 += 2) {
            const int jr[2] = {j0, min(j0 + 1, a.r - 1)};       // (odd r: the last one twice, its store once)
            if (j0 > 0)
                fetch(tile, false);
            // layer 2: 128 -> 128 on relu(a + c_j)
            v4f t1[2][RT_C2 / 16];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int t = 0; t < RT_C2 / 16; ++t)
                    t1[jj][t] = (v4f){0.f, 0.f, 0.f, 0.f};
#define RB_P(ACC, AU, XV)                                                                                      \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                             \
        _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                      \
            ACC[jj][tg + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Aq[cur][AU][t], XV[jj], ACC[jj][tg + t], 0, 0, 0);
            // the weight operands of a group of two output tiles are read one group AHEAD of their products (two register
            // sets): read just before use, every group began with an LDS round trip that two waves per SIMD do not hide
            // -- an ablation without the reads ran at 0.35 instead of 0.69 ms per chunk
#define RB_LA(BUF, WA, WB, WC, TG, SS)                                                                         \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                           \
        Aq[BUF][0][t] = w8(WA, (TG) + t, SS); Aq[BUF][1][t] = w8(WB, (TG) + t, SS); Aq[BUF][2][t] = w8(WC, (TG) + t, SS); \
    }
            bf8 Aq[2][3][2];
            RB_LA(0, w2a, w2b, w2c, 0, 0)
#pragma unroll
            for (int S = 0; S < RT_C1 / 32; ++S) {
                bf8 x1[2], x2[2], x3[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    v4f v[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        v[u] = av[2 * S + u] + ld4(cc + jr[jj] * RT_C1 + 16 * (2 * S + u) + 4 * q);
                        v[u].x = fmaxf(v[u].x, 0.f); v[u].y = fmaxf(v[u].y, 0.f);
                        v[u].z = fmaxf(v[u].z, 0.f); v[u].w = fmaxf(v[u].w, 0.f);
                    }
                    rb_split3(v[0], v[1], x1[jj], x2[jj], x3[jj]);
                }
#pragma unroll
                for (int g = 0; g < RT_C2 / 32; ++g) {
                    const int gi = S * (RT_C2 / 32) + g, cur = gi & 1, tg = 2 * g;
                    if (gi + 1 < (RT_C1 / 32) * (RT_C2 / 32)) {
                        RB_LA(cur ^ 1, w2a, w2b, w2c, 2 * ((g + 1) % (RT_C2 / 32)), S + (g + 1) / (RT_C2 / 32))
                    } else {
                        RB_LA(cur ^ 1, w3a, w3b, w3c, 0, 0)         // the third layer's first group
                    }
                    __builtin_amdgcn_sched_barrier(0);      // (the reads FIRST: left to itself the scheduler issues them last)
                    RB_P(t1, 2, x1) RB_P(t1, 0, x3) RB_P(t1, 1, x2) RB_P(t1, 1, x1) RB_P(t1, 0, x2) RB_P(t1, 0, x1)
                    __builtin_amdgcn_sched_barrier(0);      // (two groups' operand registers at a time)
                }
            }
            if (j0 + 2 >= a.r)
                fetch(tile + tstep < ntiles ? tile + tstep : tile, true);
            // layer 3: 128 -> 64 on relu(t1 + b2)
            v4f t2[2][RT_C3 / 16];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int t = 0; t < RT_C3 / 16; ++t)
                    t2[jj][t] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int S = 0; S < RT_C2 / 32; ++S) {
                bf8 x1[2], x2[2], x3[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    v4f v[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        v[u] = t1[jj][2 * S + u] + ld4(b2 + 16 * (2 * S + u) + 4 * q);
                        v[u].x = fmaxf(v[u].x, 0.f); v[u].y = fmaxf(v[u].y, 0.f);
                        v[u].z = fmaxf(v[u].z, 0.f); v[u].w = fmaxf(v[u].w, 0.f);
                    }
                    rb_split3(v[0], v[1], x1[jj], x2[jj], x3[jj]);
                }
#pragma unroll
                for (int g = 0; g < RT_C3 / 32; ++g) {
                    const int gi = S * (RT_C3 / 32) + g, cur = gi & 1, tg = 2 * g;      // (sixteen groups before: even)
                    if (gi + 1 < (RT_C2 / 32) * (RT_C3 / 32)) {
                        RB_LA(cur ^ 1, w3a, w3b, w3c, 2 * ((g + 1) % (RT_C3 / 32)), S + (g + 1) / (RT_C3 / 32))
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    RB_P(t2, 2, x1) RB_P(t2, 0, x3) RB_P(t2, 1, x2) RB_P(t2, 1, x1) RB_P(t2, 0, x2) RB_P(t2, 0, x1)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#undef RB_P
#undef RB_LA
            // layer 4: 64 -> 3 on relu(t2 + b3), fp32 operands (slab s of the B operand IS accumulator tile s)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                v4f o = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < RT_C3 / 16; ++s) {
                    v4f v = t2[jj][s] + ld4(b3 + 16 * s + 4 * q);
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    const v4f w = ld4(w4 + min(pt, 3) * RB_S4 + 16 * s + 4 * q);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        o = __builtin_amdgcn_mfma_f32_16x16x4f32(w[k], v[k], o, 0, 0, 0);
                }
                if (q == 0 && row < a.m && (jj == 0 || j0 + 1 < a.r)) {      // lanes 0-15 hold outputs 0..3 of their point
                    float *dst = a.out + (row * a.r + jr[jj]) * 3;
                    dst[0] = (o.x + b4[0]) + cx;
                    dst[1] = (o.y + b4[1]) + cy;
                    dst[2] = (o.z + b4[2]) + cz;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// y[m, cout] = x[m, cin] W^T + b with cout = 16 NT <= 128 (the per-point half of up_layer1: 264 -> 128)
// ---------------------------------------------------------------------------------------------
// MFMA-bound (67.6 KFLOP for 1.5 KB per row).  The whole weight matrix sits in LDS (128 x 276 floats = 141 KB,
// one workgroup per CU = one wave per SIMD), a wave owns 16 rows at a time: its B operands (the rows, one float4
// per 16-channel slab and lane, permuted k slots as in linear_small_kernel) are fetched a whole tile AHEAD into
// registers -- a lone wave per SIMD has 512 of them and nobody else to hide its memory latency -- and NT
// independent accumulator chains take the A operands from LDS as float4s, four MFMAs each.
struct WideArgs {
    long m;
    int cin, cout, xs, ws, ys;
    const float *x, *w, *b;
    float *y;
};

constexpr int LW_WAVES = 8;         // two per SIMD: one's loads, LDS reads and address arithmetic hide under the other's MFMAs

template <int NT, int NSL>
__global__ __launch_bounds__(64 * LW_WAVES) void linear_wide_kernel(WideArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [16 NT][LD], zero padded
    constexpr int LD = 16 * NSL + 4;
    // (the rows of a column slice are not 16-byte aligned: scalar loads, eight in flight per lane)
    constexpr int WTOT = 16 * NT * 16 * NSL, WSTEP = 64 * LW_WAVES;
    for (int e0 = threadIdx.x; e0 < WTOT; e0 += 8 * WSTEP) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = min(e0 + u * WSTEP, WTOT - 1), o = e / (16 * NSL), ch = e - o * (16 * NSL);
            v[u] = a.w[(size_t)o * a.ws + (ch < a.cin ? ch : 0)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * WSTEP, o = e / (16 * NSL), ch = e - o * (16 * NSL);
            if (e < WTOT)
                wl[o * LD + ch] = ch < a.cin ? v[u] : 0.f;
        }
    }
    float *bl = wl + 16 * NT * LD;                                   // bias, added when a tile is finished
    for (int e = threadIdx.x; e < 16 * NT; e += 64 * LW_WAVES)
        bl[e] = a.b ? a.b[e] : 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = lane & 15, q = lane >> 4;
    const long ntiles = (a.m + 15) >> 4;
    const long stride = (long)gridDim.x * LW_WAVES;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    const bool last_ok = 16 * (NSL - 1) + 4 * q < a.cin;            // cin % 4 == 0: a float4 is all in or all out
    auto row_of = [&](long tile) __attribute__((always_inline)) {
        const long row = tile * 16 + pt;
        return a.x + (row < a.m ? row : a.m - 1) * (long)a.xs + 4 * q;
    };
    auto slab = [&](const float *xr, int s) __attribute__((always_inline)) {
        if (s + 1 < NSL)
            return ld4(xr + 16 * s);                                // (cin > 16 (NSL - 1): checked by the entry)
        const v4f v = ld4(xr + (last_ok ? 16 * s : 0));
        return last_ok ? v : zero;
    };
    long tile = (long)blockIdx.x * LW_WAVES + wave;
    // the wave's B operands: slab s of the NEXT tile is fetched into its register as soon as the MFMAs of slab s
    // of this tile are issued -- a whole tile (17 k cycles) ahead of its use, one register set
    v4f xb[NSL];
    if (tile < ntiles) {
        const float *xr = row_of(tile);
#pragma unroll
        for (int s = 0; s < NSL; ++s)
            xb[s] = slab(xr, s);
    }
    for (; tile < ntiles; tile += stride) {
        const float *xrn = row_of(tile + stride < ntiles ? tile + stride : tile);
        v4f acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
            acc[t] = zero;
        const float *wr = wl + pt * LD + 4 * q;
        // A operands one slab ahead of the MFMAs that use them (the fences keep the compiler from hoisting all 17
        // slabs of LDS reads -- 544 registers -- to the top)
        v4f av[2][NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
            av[0][t] = *(const v4f *)(wr + 16 * t * LD);
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
            if (s + 1 < NSL) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    av[(s + 1) & 1][t] = *(const v4f *)(wr + 16 * t * LD + 16 * (s + 1));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s & 1][t][j], xb[s][j], acc[t], 0, 0, 0);
            xb[s] = slab(xrn, s);
            __builtin_amdgcn_sched_barrier(0);
        }
        const long row = tile * 16 + pt;
        if (row < a.m) {
            float *yr = a.y + row * (long)a.ys + 4 * q;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                *(v4f *)(yr + 16 * t) = acc[t] + *(const v4f *)(bl + 16 * t + 4 * q);
        }
    }
}

// (r6) SPLIT-bf16 flavour of the 264 -> 128 layer (TPU3_SPLIT_BF16=1, with regress_tail_sb_kernel): every fp32 operand
// as three bf16 terms, six partial products per (output tile, 32-channel slab) on v_mfma_f32_16x16x32_bf16.  The split
// weights (3 x 128 x 288 bf16 = 221 KB) do not fit LDS, so this is a GEMM main loop: a small pre-pass splits W once per
// call into slab-major order [slab][term][tile][k octet][output][8 k], a workgroup walks 128-row blocks, and per block the nine
// weight slabs (24 KB each) stream through a two-slot LDS ring one slab ahead (global -> registers -> LDS, one barrier
// per slab) while every wave keeps TWO 16-row tiles: an A operand read from LDS feeds two MFMAs (0.25 ds_read_b128 per
// MFMA -- the tail kernel's 0.5 made LDS its bound).  The rows come straight from memory, one slab ahead, 32 contiguous
// bytes per lane, and are split in registers just before their products.  Channels 264 .. 287 of the last slab are
// zeros on both sides.  Not bit-identical to linear_wide_kernel (fp32 accumulation order), as accurate.
constexpr int WSB_WAVES = 4, WSB_RT = 2, WSB_ROWS = WSB_WAVES * WSB_RT * 16;
constexpr int WSB_TG = 2;                           // output tiles whose products are interleaved
constexpr int WSB_RESIDENT = 2;                     // workgroups per compute unit (207 registers: two waves per SIMD)
constexpr int WSB_SLAB = 3 * 128 * 32;              // bf16 elements of one weight slab

struct WideSbArgs {
    long m;
    int cin, xs, ys;
    const float *x, *b;
    const __bf16 *ws;                               // [nsb][3][8 tiles][4 k octets][16 outputs][8 k]
    float *y;
};

__global__ __launch_bounds__(256) void wide_split_kernel(int cin, int ws, int nsb, const float *__restrict__ w,
                                                         __bf16 *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nsb * 128 * 32)
        return;
    const int S = i >> 12, o = (i >> 5) & 127, k = i & 31, ch = 32 * S + k;
    const float x = ch < cin ? w[(size_t)o * ws + ch] : 0.f;
    const __bf16 p1 = (__bf16)x;
    const float r1 = x - (float)p1;
    const __bf16 p2 = (__bf16)r1;
    // inside a (term, 16-output tile): [k octet q][output pt][8 k] -- lane 16 q + pt of the consumer reads 16 bytes at
    // 16 * lane, the one ds_read_b128 pattern without bank conflicts (rows of 64 bytes at [pt][q] conflict two ways)
    __bf16 *dst = out + (size_t)S * WSB_SLAB + (((o >> 4) * 4 + (k >> 3)) * 16 + (o & 15)) * 8 + (k & 7);
    dst[0] = p1;
    dst[128 * 32] = p2;
    dst[2 * 128 * 32] = (__bf16)(r1 - (float)p2);
}

template <int NSB>
__global__ __launch_bounds__(64 * WSB_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void linear_wide_sb_kernel(WideSbArgs a)
{
    __shared__ __attribute__((aligned(16))) __bf16 wl[2][WSB_SLAB];         // 48 KB: the weight slab ring
    __shared__ __attribute__((aligned(16))) float bl[128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pt = lane & 15, q = lane >> 4;
    if (tid < 128)
        bl[tid] = a.b ? a.b[tid] : 0.f;
    constexpr int WCH = WSB_SLAB * 2 / 16 / (64 * WSB_WAVES);               // 16-byte chunks of a slab per thread (6)
    const long nblocks = (a.m + WSB_ROWS - 1) / WSB_ROWS;
    constexpr int nsb = NSB;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    for (long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const float *xr[WSB_RT];
        long row[WSB_RT];
#pragma unroll
        for (int rt = 0; rt < WSB_RT; ++rt) {
            row[rt] = blk * WSB_ROWS + (wave * WSB_RT + rt) * 16 + pt;
            xr[rt] = a.x + (row[rt] < a.m ? row[rt] : a.m - 1) * (long)a.xs + 8 * q;
        }
        // (the loads are NOT masked here: a select on the loaded value would make the wave wait for it at once; slab
        // lanes beyond cin read the row's first channels instead and are zeroed when the slab is consumed)
        auto load_x = [&](int S, v4f (&v)[WSB_RT][2]) __attribute__((always_inline)) {
            const bool ok = 32 * S + 8 * q + 8 <= a.cin;                    // cin % 8 == 0 (checked by the entry)
#pragma unroll
            for (int rt = 0; rt < WSB_RT; ++rt) {
                const float *p = xr[rt] + (ok ? 32 * S : 0);
                v[rt][0] = *(const v4f *)p;
                v[rt][1] = *((const v4f *)p + 1);
            }
        };
        // (the staging registers are plain scalars handed around by value: as an array captured by the lambdas they
        // were "promoted" into LDS, a round trip per slab)
        struct WReg { uint4 v0, v1, v2, v3, v4, v5; };
        static_assert(WCH == 6, "six 16-byte chunks of a slab per thread");
        auto load_w = [&](int S) __attribute__((always_inline)) {
            const uint4 *src = (const uint4 *)(a.ws + (size_t)S * WSB_SLAB) + tid;
            constexpr int ST = 64 * WSB_WAVES;
            return WReg{src[0], src[ST], src[2 * ST], src[3 * ST], src[4 * ST], src[5 * ST]};
        };
        auto store_w = [&](int slot, const WReg &r) __attribute__((always_inline)) {
            uint4 *dst = (uint4 *)wl[slot] + tid;
            constexpr int ST = 64 * WSB_WAVES;
            dst[0] = r.v0; dst[ST] = r.v1; dst[2 * ST] = r.v2; dst[3 * ST] = r.v3; dst[4 * ST] = r.v4; dst[5 * ST] = r.v5;
        };
        // the rows TWO slabs ahead (a slab's products last ~1 us: less than a loaded memory system's latency), in three
        // register sets that take turns (the loop is unrolled by three: a copy from set to set would wait for the load)
        v4f xb[3][WSB_RT][2];
        WReg wreg = load_w(0);
        load_x(0, xb[0]);
        load_x(nsb > 1 ? 1 : 0, xb[1]);
        store_w(0, wreg);
        __syncthreads();
        v4f acc[8][WSB_RT];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int rt = 0; rt < WSB_RT; ++rt)
                acc[t][rt] = zero;
        auto step = [&](int S, v4f (&cur)[WSB_RT][2], v4f (&fut)[WSB_RT][2]) __attribute__((always_inline)) {
            // the weights FIRST: the wait for them at the end of the step then leaves the rows' loads in flight
            // (vmcnt counts in issue order);
            // (both UNCONDITIONAL -- the last steps fetch the last slab again: behind a branch the wait-count pass
            // must assume the loads were not issued and waits for everything)
            wreg = load_w(min(S + 1, nsb - 1));
            __builtin_amdgcn_sched_barrier(0);
            load_x(min(S + 2, nsb - 1), fut);
            __builtin_amdgcn_sched_barrier(0);
            const bool ok = 32 * S + 8 * q + 8 <= a.cin;
            bf8 x1[WSB_RT], x2[WSB_RT], x3[WSB_RT];
#pragma unroll
            for (int rt = 0; rt < WSB_RT; ++rt)
                rb_split3(ok ? cur[rt][0] : zero, ok ? cur[rt][1] : zero, x1[rt], x2[rt], x3[rt]);
            // Products in PRODUCT-major order over a group of WSB_TG output tiles x two row tiles: four independent
            // accumulators between two MFMAs on the same one (a dependent pair with anything issued in between costs
            // a ~40-cycle bubble: MI355X guide, cycle table), the group's six A operands read in one batch
            const __bf16 *wa = wl[S & 1] + 8 * lane;
#pragma unroll
            for (int tg = 0; tg < 8; tg += WSB_TG) {
                bf8 A[3][WSB_TG];
#pragma unroll
                for (int u = 0; u < 3; ++u)
#pragma unroll
                    for (int t = 0; t < WSB_TG; ++t)
                        A[u][t] = *(const bf8 *)(wa + (8 * u + (tg + t)) * 512);
#define WSB_P(AU, XV)                                                                                          \
    _Pragma("unroll") for (int t = 0; t < WSB_TG; ++t)                                                        \
        _Pragma("unroll") for (int rt = 0; rt < WSB_RT; ++rt)                                                 \
            acc[tg + t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[AU][t], XV[rt], acc[tg + t][rt], 0, 0, 0);
                WSB_P(2, x1) WSB_P(0, x3) WSB_P(1, x2) WSB_P(1, x1) WSB_P(0, x2) WSB_P(0, x1)
#undef WSB_P
            }
            // (the fence keeps the compiler from merging this with the loads above: it would wait for the next slab's
            // global loads at the TOP of the step, before the products that are there to hide them)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
            store_w((S + 1) & 1, wreg);
            __syncthreads();
        };
#pragma unroll 1
        for (int S = 0; S < nsb; S += 3) {
            step(S, xb[0], xb[2]);
            if (S + 1 < nsb)
                step(S + 1, xb[1], xb[0]);
            if (S + 2 < nsb)
                step(S + 2, xb[2], xb[1]);
        }
#pragma unroll
        for (int rt = 0; rt < WSB_RT; ++rt) {
            if (row[rt] < a.m) {
                float *yr = a.y + row[rt] * (long)a.ys + 4 * q;
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    *(v4f *)(yr + 16 * t) = acc[t][rt] + *(const v4f *)(bl + 16 * t + 4 * q);
            }
        }
    }
}
#undef RB_MFMA6

// ---------------------------------------------------------------------------------------------
// y[m, cout] = act(x[m, cin] W^T + b) for a handful of input channels (the 3 -> 24 lift of a Level)
// ---------------------------------------------------------------------------------------------
// HBM-bound: 4 cin B read, 4 cout B written per row (twice when the row is also stored into the level's feature
// buffer).  A lane owns four outputs of a row; the weights sit in registers as wave-uniform values.
struct LiftArgs {
    long m;
    int cin, cout, xs, ys, y2s, relu;
    const float *x, *w, *b;
    float *y, *y2;
};

constexpr int LIFT_CIN_MAX = 8;

__global__ __launch_bounds__(256) void linear_lift_kernel(LiftArgs a)
{
    const int q = a.cout >> 2;                  // float4 groups per row
    const long total = a.m * q;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long row = t / q;
        const int o = (int)(t - row * q) * 4;
        float xv[LIFT_CIN_MAX];
#pragma unroll
        for (int c = 0; c < LIFT_CIN_MAX; ++c)
            xv[c] = c < a.cin ? a.x[row * a.xs + c] : 0.f;
        float acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[u] = a.b ? a.b[o + u] : 0.f;
#pragma unroll
            for (int c = 0; c < LIFT_CIN_MAX; ++c)
                if (c < a.cin)
                    acc[u] = __builtin_fmaf(a.w[(o + u) * a.cin + c], xv[c], acc[u]);
            if (a.relu)
                acc[u] = fmaxf(acc[u], 0.f);
        }
        const float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *(float4 *)(a.y + row * a.ys + o) = v;
        if (a.y2)
            *(float4 *)(a.y2 + row * a.y2s + o) = v;
    }
}

} // namespace

// split-bf16 arithmetic of the regressor's matrix layers (regress_tail_sb_kernel, linear_wide_sb_kernel): default from
// TPU3_SPLIT_BF16 (unset = SPLIT_BF16_DEFAULT), switched at run time by tpu3_split_bf16
#ifndef SPLIT_BF16_DEFAULT
#define SPLIT_BF16_DEFAULT 1
#endif
static int g_split_bf16 = getenv("TPU3_SPLIT_BF16") ? atoi(getenv("TPU3_SPLIT_BF16")) != 0 : SPLIT_BF16_DEFAULT;

extern "C" int tpu3_split_bf16(int on)
{
    const int old = g_split_bf16;
    if (on >= 0)
        g_split_bf16 = on != 0;
    return old;
}

extern "C" int tpu3_linear_wide_f32(tpu3_stream_t stream, long m, int cin, int cout, const float *x, int x_stride,
                                    const float *w, int w_stride, const float *bias, float *y, int y_stride)
{
    if (m < 0 || cin <= 0 || cout <= 0) return TPU3_EINVAL;
    if (x_stride < cin || y_stride < cout || w_stride < cin) return TPU3_EINVAL;
    // instantiated: 128 outputs, 257 .. 272 input channels (the reference's 264)
    if (cout != 128 || cin % 4 || cin <= 256 || cin > 272 || x_stride % 4 || y_stride % 4) return TPU3_ELIMIT;
    if (m == 0) return TPU3_OK;
    if (!x || !w || !y) return TPU3_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)bias) & 15) != 0) return TPU3_ELIMIT;
    WideArgs a{m, cin, cout, x_stride, w_stride, y_stride, x, w, bias, y};
    constexpr int NT = 8, NSL = 17;
    const int lds = (16 * NT * (16 * NSL + 4) + 16 * NT) * (int)sizeof(float);
    const hipError_t e = hipFuncSetAttribute((const void *)linear_wide_kernel<NT, NSL>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    long blocks = ((m + 15) / 16 + LW_WAVES - 1) / LW_WAVES;
    if (blocks > 256) blocks = 256;                 // persistent: one workgroup per CU holds the weights
    hipLaunchKernelGGL((linear_wide_kernel<NT, NSL>), dim3((unsigned)blocks), dim3(64 * LW_WAVES), lds,
                       (hipStream_t)stream, a);
    return tpu3_launch_status();
}

// (r6) split-bf16 flavour of tpu3_linear_wide_f32 in two steps, so that the split of W is paid once per set of weights:
// tpu3_linear_wide_split_bf16 writes the slab-major three-term image (tpu3_linear_wide_split_bytes(cin) bytes, 16-byte
// aligned), tpu3_linear_wide_sb_f32 consumes it.
extern "C" size_t tpu3_linear_wide_split_bytes(int cin)
{
    return cin > 0 ? (size_t)((cin + 31) / 32) * WSB_SLAB * sizeof(__bf16) : 0;
}

extern "C" int tpu3_linear_wide_split_bf16(tpu3_stream_t stream, int cin, int cout, const float *w, int w_stride, void *ws)
{
    if (cin <= 0 || cout != 128 || w_stride < cin) return cout > 0 && cout != 128 ? TPU3_ELIMIT : TPU3_EINVAL;
    if (!w || !ws || ((uintptr_t)ws & 15)) return TPU3_EINVAL;
    const int nsb = (cin + 31) / 32;
    hipLaunchKernelGGL(wide_split_kernel, dim3((nsb * 128 * 32 + 255) / 256), dim3(256), 0, (hipStream_t)stream, cin,
                       w_stride, nsb, w, (__bf16 *)ws);
    return tpu3_launch_status();
}

extern "C" int tpu3_linear_wide_sb_f32(tpu3_stream_t stream, long m, int cin, int cout, const float *x, int x_stride,
                                       const void *ws, const float *bias, float *y, int y_stride)
{
    if (m < 0 || cin <= 0 || cout <= 0) return TPU3_EINVAL;
    if (x_stride < cin || y_stride < cout) return TPU3_EINVAL;
    if (cout != 128 || cin % 8 || cin <= 256 || cin > 288 || x_stride % 8 || y_stride % 4) return TPU3_ELIMIT;     // nine slabs
    if (m == 0) return TPU3_OK;
    if (!x || !ws || !y) return TPU3_EINVAL;
    if (((uintptr_t)x & 31) || (((uintptr_t)y | (uintptr_t)bias | (uintptr_t)ws) & 15)) return TPU3_ELIMIT;
    static const int cus = []() { int d = 0, v = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d); return v; }();
    WideSbArgs sa{m, cin, x_stride, y_stride, x, bias, (const __bf16 *)ws, y};
    long blocks = (m + WSB_ROWS - 1) / WSB_ROWS;
    const long cap = (long)cus * WSB_RESIDENT;          // what is resident at once: a third wave of workgroups would run alone at the end
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(linear_wide_sb_kernel<9>, dim3((unsigned)blocks), dim3(64 * WSB_WAVES), 0, (hipStream_t)stream, sa);
    return tpu3_launch_status();
}

extern "C" int tpu3_linear_lift_f32(tpu3_stream_t stream, long m, int cin, int cout, const float *x, int x_stride,
                                    const float *w, const float *bias, int relu, float *y, int y_stride,
                                    float *y2, int y2_stride)
{
    if (m < 0 || cin <= 0 || cout <= 0) return TPU3_EINVAL;
    if (cin > LIFT_CIN_MAX || cout > 64 || cout % 4 || y_stride % 4 || (y2 && y2_stride % 4)) return TPU3_ELIMIT;
    if (x_stride < cin || y_stride < cout || (y2 && y2_stride < cout)) return TPU3_EINVAL;
    if (m == 0) return TPU3_OK;
    if (!x || !w || !y) return TPU3_EINVAL;
    if ((((uintptr_t)y | (uintptr_t)y2) & 15) != 0) return TPU3_ELIMIT;
    LiftArgs a{m, cin, cout, x_stride, y_stride, y2_stride, relu, x, w, bias, y, y2};
    long blocks = (m * (cout / 4) + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(linear_lift_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return tpu3_launch_status();
}


namespace {
int linear_small_impl(tpu3_stream_t stream, long m, int cin, int cout, const float *x, int x_stride, const float *w,
                      const float *bias, int relu, float *y, int y_stride, int mfma, bool xh);
}

extern "C" int tpu3_linear_small_f32(tpu3_stream_t stream, long m, int cin, int cout, const float *x,
                                     int x_stride, const float *w, const float *bias, int relu, float *y,
                                     int y_stride, int mfma)
{
    return linear_small_impl(stream, m, cin, cout, x, x_stride, w, bias, relu, y, y_stride, mfma, false);
}

extern "C" int tpu3_linear_small_st_f32(tpu3_stream_t stream, long m, int cin, int cout, const void *x,
                                        int x_stride, const float *w, const float *bias, int relu, float *y,
                                        int y_stride, int mfma, int x_store)
{
    if (x_store != TPU3_STORE_F32 && x_store != TPU3_STORE_F16) return TPU3_EINVAL;
    // fp16 rows only feed fp16-operand matrix instructions (rounding them again would be exact; widening them for the
    // fp32 flavour would claim a precision the rows do not have)
    if (x_store == TPU3_STORE_F16 && mfma != TPU3_MFMA_F16) return TPU3_EINVAL;
    return linear_small_impl(stream, m, cin, cout, (const float *)x, x_stride, w, bias, relu, y, y_stride, mfma,
                             x_store == TPU3_STORE_F16);
}

namespace {
int linear_small_impl(tpu3_stream_t stream, long m, int cin, int cout, const float *x, int x_stride, const float *w,
                      const float *bias, int relu, float *y, int y_stride, int mfma, bool xh)
{
    if (mfma != TPU3_MFMA_F32 && mfma != TPU3_MFMA_F16) return TPU3_EINVAL;
    if (m < 0 || cin <= 0 || cout <= 0) return TPU3_EINVAL;
    // fp32 operands: cout <= 32 (wider layers: tpu3_linear_wide_f32); fp16 operands: cout <= 128
    if (cout > (mfma == TPU3_MFMA_F16 ? 128 : 32) || cin > LS_CIN_MAX || cin % 4 || cout % 4 || x_stride % 4 ||
        y_stride % 4)
        return TPU3_ELIMIT;
    if (x_stride < cin || y_stride < cout) return TPU3_EINVAL;
    if (m == 0) return TPU3_OK;
    if (!x || !w || !y) return TPU3_EINVAL;
    if ((((uintptr_t)w | (uintptr_t)y | (uintptr_t)bias) & 15) != 0 || ((uintptr_t)x & (xh ? 7 : 15)) != 0)
        return TPU3_ELIMIT;
    LinArgs a{m, cin, cout, x_stride, y_stride, relu, x, w, bias, y};
    const long tiles = (m + 15) / 16;
    long blocks = (tiles + 3) / 4;
    if (cout > 32) {            // fp16 operands, wide: weights as fp16 in LDS, two persistent workgroups per CU
        const size_t ldsw = (size_t)128 * (cin + 8) * sizeof(_Float16);
        const hipError_t e = hipFuncSetAttribute((const void *)linear_wide_f16_kernel<8>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
        if (e != hipSuccess) return (int)e;
        if (blocks > 512) blocks = 512;
        if (xh) {
            const hipError_t e2 = hipFuncSetAttribute((const void *)linear_wide_f16_kernel<8, true>,
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
            if (e2 != hipSuccess) return (int)e2;
            hipLaunchKernelGGL((linear_wide_f16_kernel<8, true>), dim3((unsigned)blocks), dim3(256), ldsw,
                               (hipStream_t)stream, a);
        } else {
            hipLaunchKernelGGL(linear_wide_f16_kernel<8>, dim3((unsigned)blocks), dim3(256), ldsw, (hipStream_t)stream, a);
        }
        return tpu3_launch_status();
    }
    if (blocks > 256 * 8) blocks = 256 * 8;       // persistent: the weights are staged once per workgroup
    const int tout = cout <= 16 ? 1 : 2;
    const size_t lds = (size_t)16 * tout * (cin + 4) * sizeof(float);
    if (mfma == TPU3_MFMA_F16 && xh) {
        if (tout == 1)
            hipLaunchKernelGGL((linear_small_f16_kernel<1, true>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL((linear_small_f16_kernel<2, true>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, a);
    } else if (mfma == TPU3_MFMA_F16) {
        if (tout == 1)
            hipLaunchKernelGGL(linear_small_f16_kernel<1>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL(linear_small_f16_kernel<2>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, a);
    } else if (tout == 1)
        hipLaunchKernelGGL(linear_small_kernel<1>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(linear_small_kernel<2>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, a);
    return tpu3_launch_status();
}
} // namespace

extern "C" int tpu3_regress_tail_f32(tpu3_stream_t stream, long m, int r, const float *a, const float *c,
                                     const float *w2, const float *b2, const float *w3, const float *b3,
                                     const float *w4, const float *b4, const float *residual, float *out, int mfma)
{
    if (m < 0 || r <= 0 || r > RT_RMAX) return TPU3_EINVAL;
    if (mfma != TPU3_MFMA_F32 && mfma != TPU3_MFMA_F16) return TPU3_EINVAL;
    if (m == 0) return TPU3_OK;
    if (!a || !c || !w2 || !b2 || !w3 || !b3 || !w4 || !b4 || !residual || !out) return TPU3_EINVAL;
    if (((uintptr_t)a & 15) != 0) return TPU3_ELIMIT;
    TailArgs t{m, r, a, c, w2, b2, w3, b3, w4, b4, residual, out};
    const long tiles = (m + 15) / 16;
    long blocks = (tiles + 7) / 8;
    if (mfma == TPU3_MFMA_F16) {
        const size_t lds = rh_lds_bytes();
        hipError_t e = hipFuncSetAttribute((const void *)regress_tail_f16_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        if (blocks > 512) blocks = 512;         // two persistent workgroups per CU (55 KB of weights each)
        hipLaunchKernelGGL(regress_tail_f16_kernel, dim3((unsigned)blocks), dim3(512), lds, (hipStream_t)stream, t);
        return tpu3_launch_status();
    }
    // (r6) split-bf16 arithmetic (tpu3_split_bf16 / TPU3_SPLIT_BF16): fp32 operands as three bf16 terms on the bf16
    // matrix pipe (regress_tail_sb_kernel)
    if (g_split_bf16) {
        const size_t ldsb = rb_lds_bytes();
        hipError_t eb = hipFuncSetAttribute((const void *)regress_tail_sb_kernel,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        if (eb != hipSuccess) return (int)eb;
        if (blocks > 256) blocks = 256;         // one persistent workgroup per CU (157 KB of split weights)
        hipLaunchKernelGGL(regress_tail_sb_kernel, dim3((unsigned)blocks), dim3(512), ldsb, (hipStream_t)stream, t);
        return tpu3_launch_status();
    }
    const size_t lds = rt_lds_floats() * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void *)regress_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
    if (blocks > 256) blocks = 256;             // one persistent workgroup per CU (106 KB of weights each)
    hipLaunchKernelGGL(regress_tail_kernel, dim3((unsigned)blocks), dim3(512), lds, (hipStream_t)stream, t);
    return tpu3_launch_status();
}

// ---------------------------------------------------------------------------------------------
// training: weight gradient of a per-point linear layer with few outputs over very many rows
//   dW[o][c] = sum_rows dy[row][o] * x[row][c]        (cout <= 16, cin <= 64)
// The dense layers of DenseEdgeConv see B*N*k = 319 488 rows per step (config C3) with 12 outputs:
// a vendor GEMM runs this 48 x 12 reduction at 0.6 TFLOP/s (576 us); it is a streaming read of x
// and dy.  Stage 1: every workgroup streams a contiguous range of rows through LDS and accumulates
// its cout x cin block on the matrix cores (A = dy^T: 16 outputs x 4 rows, B = x: 4 rows x 16
// inputs); stage 2 adds the workgroups' blocks in a fixed order (deterministic, no atomics).
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int WG_ROWS = 64;              // rows per LDS tile
constexpr int WG_CMAX = 64;              // cin <= 64 (4 column tiles)

struct WgradArgs {
    long m;
    int cin, cout, xs, dys;
    const float *x, *dy;
    float *partial;                      // (blocks, 16, WG_CMAX)
    long rows_per_block;
};

template <int T>     // column tiles: cin <= 16 T
__global__ __launch_bounds__(256) void linear_wgrad_kernel(WgradArgs a)
{
    __shared__ float xs[WG_ROWS][16 * T + 1];
    __shared__ float dys[WG_ROWS][17];
    __shared__ v4f red[4][T][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const long r_lo = (long)blockIdx.x * a.rows_per_block;
    const long r_hi = r_lo + a.rows_per_block < a.m ? r_lo + a.rows_per_block : a.m;
    v4f acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
        acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (long r0 = r_lo; r0 < r_hi; r0 += WG_ROWS) {
        __syncthreads();
        for (int e = tid; e < WG_ROWS * 16 * T; e += 256) {
            const int r = e / (16 * T), c = e - r * (16 * T);
            xs[r][c] = (r0 + r < r_hi && c < a.cin) ? a.x[(r0 + r) * a.xs + c] : 0.f;
        }
        for (int e = tid; e < WG_ROWS * 16; e += 256) {
            const int r = e >> 4, o = e & 15;
            dys[r][o] = (r0 + r < r_hi && o < a.cout) ? a.dy[(r0 + r) * a.dys + o] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < WG_ROWS / 16; ++s) {            // this wave's 16 rows, 4 at a time
            const int r = wave * 16 + 4 * s + g;
            const float av = dys[r][i];
#pragma unroll
            for (int t = 0; t < T; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xs[r][16 * t + i], acc[t], 0, 0, 0);
        }
    }
    // waves -> one block: acc[t][q] = dW[4 g + q][16 t + i]
#pragma unroll
    for (int t = 0; t < T; ++t)
        red[wave][t][lane] = acc[t];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const v4f v = (red[0][t][lane] + red[1][t][lane]) + (red[2][t][lane] + red[3][t][lane]);
            float *p = a.partial + ((size_t)blockIdx.x * 16) * WG_CMAX;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                p[(4 * g + q) * WG_CMAX + 16 * t + i] = v[q];
        }
    }
}

// one wave per element of dW: lanes stride over the workgroups' blocks, then a fixed butterfly
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(int blocks, int cin, int cout,
                                                                  const float *__restrict__ partial,
                                                                  float *__restrict__ dw)
{
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= cin * cout)
        return;
    const int o = e / cin, c = e - o * cin;
    float s = 0.f;
    for (int b = lane; b < blocks; b += 64)
        s += partial[((size_t)b * 16 + o) * WG_CMAX + c];
    s = tpu3_wave_sum_f32(s);
    if (lane == 0)
        dw[e] = s;
}

// ---- any per-point layer of a Level (training): dW and db together --------------------------------------------
// The prep convolutions, the lift and the regressor layers (3 .. 265 inputs, 3 .. 128 outputs over B*N or B*N*r
// = 9 984 / 19 968 rows at config C3): autograd's weight gradient is a vendor GEMM of 65 .. 150 us (one or three
// 32 x 32 tiles over K = 10^4) and its bias gradient a separate reduction.  Here the (cout x cin+1) block is cut
// into 16-output x 64-input pieces (the bias rides along as the input column that is 1 everywhere), every piece
// of every row range is one workgroup of the streaming kernel above, and the second stage adds the ranges'
// pieces in a fixed order.
struct WgradWideArgs {
    long m;
    int cin, cout, xs, dys;
    const float *x, *dy;
    float *partial;                      // (blocks, og * 16, cg * 64)
    long rows_per_block;
    int cg, og;                          // column groups of 64 (cin + 1 columns), output groups of 16
};

__global__ __launch_bounds__(256) void linear_wgrad_wide_kernel(WgradWideArgs a)
{
    constexpr int T = 4;
    __shared__ float xs[WG_ROWS][16 * T + 1];
    __shared__ float dys[WG_ROWS][17];
    __shared__ v4f red[4][T][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int o0 = (blockIdx.y / a.cg) * 16, c0 = (blockIdx.y % a.cg) * 64;
    const long r_lo = (long)blockIdx.x * a.rows_per_block;
    const long r_hi = r_lo + a.rows_per_block < a.m ? r_lo + a.rows_per_block : a.m;
    v4f acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
        acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (long r0 = r_lo; r0 < r_hi; r0 += WG_ROWS) {
        __syncthreads();
        for (int e = tid; e < WG_ROWS * 64; e += 256) {
            const int r = e >> 6, c = c0 + (e & 63);
            xs[r][e & 63] = r0 + r < r_hi ? (c < a.cin ? a.x[(r0 + r) * a.xs + c] : (c == a.cin ? 1.f : 0.f)) : 0.f;
        }
        for (int e = tid; e < WG_ROWS * 16; e += 256) {
            const int r = e >> 4, o = o0 + (e & 15);
            dys[r][e & 15] = (r0 + r < r_hi && o < a.cout) ? a.dy[(r0 + r) * a.dys + o] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < WG_ROWS / 16; ++s) {
            const int r = wave * 16 + 4 * s + g;
            const float av = dys[r][i];
#pragma unroll
            for (int t = 0; t < T; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xs[r][16 * t + i], acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t)
        red[wave][t][lane] = acc[t];
    __syncthreads();
    if (wave == 0) {
        const int cpad = a.cg * 64;
        float *p = a.partial + ((size_t)blockIdx.x * (a.og * 16) + o0) * cpad + c0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const v4f v = (red[0][t][lane] + red[1][t][lane]) + (red[2][t][lane] + red[3][t][lane]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                p[(size_t)(4 * g + q) * cpad + 16 * t + i] = v[q];
        }
    }
}

// a thread per element of [dW | db]: the row ranges' pieces in order (coalesced over the elements)
__global__ __launch_bounds__(256) void linear_wgrad_wide_reduce_kernel(int blocks, int cin, int cout, int cg, int og,
                                                                       const float *__restrict__ partial,
                                                                       float *__restrict__ dw, float *__restrict__ db)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= cout * (cin + 1))
        return;
    const int o = e / (cin + 1), c = e - o * (cin + 1);
    const size_t cpad = (size_t)cg * 64, step = (size_t)og * 16 * cpad;
    const float *p = partial + (size_t)o * cpad + c;
    float s = 0.f;
    for (int b = 0; b < blocks; ++b)
        s += p[b * step];
    if (c < cin)
        dw[(size_t)o * cin + c] = s;
    else if (db)
        db[o] = s;
}

// The same gradient with the whole (cout x cin+1) block in ONE workgroup per row range (cout <= 128, cin <= 319: every
// layer of a Level): a 64-row tile of x (all columns, the ones column, zero padding) and of dy sits in LDS, each
// wave owns every fourth 16-column tile for all OG output groups -- x and dy are read from memory exactly once, an A
// operand serves CTW matrix instructions, and a tile of the 265 -> 128 layer is 640 of them per wave against ~100
// loads per thread (the piecewise kernel above re-reads x per output group and dy per column group and spends most
// of its time staging: 150 us for that layer, the vendor GEMM's time).
struct WgradAllArgs {
    long m;
    int cin, cout, xs, dys;
    const float *x, *dy;
    float *partial;                      // (blocks, OG * 16, ct * 16)
    long rows_per_block;
    int ct;                              // 16-column tiles of cin + 1 columns
};

// FLAT (CTW == 1 only): x and dy are dense row-major with 4 | cin, 4 | cout -- a tile's 64 rows are one contiguous
// block of each, fetched as float4 (6 loads per lane for the 48 -> 36 edge tensors of a DenseEdgeConv block instead of
// 32 dword loads of which a quarter of the lanes idle: 62 -> 40 us for its 107 MB)
template <int OG, int CTW, bool FLAT = false>
__global__ __launch_bounds__(256) void linear_wgrad_all_kernel(WgradAllArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float wg_lds[];
    const int xc = a.ct * 16, xw = xc + 1, yw = OG * 16 + 1;
    float *xs = wg_lds;                  // [64][xw]
    float *dys = wg_lds + 64 * xw;       // [64][yw]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const long r_lo = (long)blockIdx.x * a.rows_per_block;
    const long r_hi = r_lo + a.rows_per_block < a.m ? r_lo + a.rows_per_block : a.m;
    v4f acc[OG][CTW];
#pragma unroll
    for (int og = 0; og < OG; ++og)
#pragma unroll
        for (int t = 0; t < CTW; ++t)
            acc[og][t] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (long r0 = r_lo; r0 < r_hi; r0 += 64) {
        __syncthreads();
        // this wave's 16 rows, four at a time: all their loads in flight before the first store (a load, a wait and
        // a store per element was 56 us per tile of the 265 -> 128 layer)
        constexpr int YL = (OG * 16 + 63) / 64;
        constexpr int RB = CTW == 1 ? 16 : CTW == 2 ? 8 : 4;        // rows per batch: ~30 loads per lane in flight
        if constexpr (FLAT) {
            const long left = (r_hi - r0 < 64 ? r_hi - r0 : 64);
            const int nx = (int)left * a.cin / 4, ny = (int)left * a.cout / 4;
            const float4 *X4 = (const float4 *)(a.x + r0 * a.cin), *Y4 = (const float4 *)(a.dy + r0 * a.cout);
            float4 xq[4], yq[OG];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                xq[u] = tid + 256 * u < nx ? X4[tid + 256 * u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < OG; ++u)
                yq[u] = tid + 256 * u < ny ? Y4[tid + 256 * u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = 4 * (tid + 256 * u);
                if (f < 64 * a.cin) {
                    const int r = f / a.cin, c = f - r * a.cin;
                    float *d = xs + r * xw + c;
                    d[0] = xq[u].x; d[1] = xq[u].y; d[2] = xq[u].z; d[3] = xq[u].w;
                }
            }
#pragma unroll
            for (int u = 0; u < OG; ++u) {
                const int f = 4 * (tid + 256 * u);
                if (f < 64 * a.cout) {
                    const int r = f / a.cout, c = f - r * a.cout;
                    float *d = dys + r * yw + c;
                    d[0] = yq[u].x; d[1] = yq[u].y; d[2] = yq[u].z; d[3] = yq[u].w;
                }
            }
            if (tid < 64)
                xs[tid * xw + a.cin] = tid < left ? 1.f : 0.f;
            // (columns beyond cin and outputs beyond cout keep whatever LDS held: they only reach entries of the
            // partial block that nobody reads)
        } else
        for (int rb = 0; rb < 16; rb += RB) {
            float xv[RB][CTW], yv[RB][YL];
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const long row = r0 + wave + 4 * (rb + j);
                const bool live = row < r_hi;
                const float *X = a.x + (live ? row : r_lo) * a.xs;
                const float *Y = a.dy + (live ? row : r_lo) * a.dys;
#pragma unroll
                for (int u = 0; u < CTW; ++u) {
                    const int c = lane + 64 * u;
                    xv[j][u] = (live && c < a.cin) ? X[c] : ((live && c == a.cin) ? 1.f : 0.f);
                }
#pragma unroll
                for (int u = 0; u < YL; ++u) {
                    const int o = lane + 64 * u;
                    yv[j][u] = (live && o < a.cout) ? Y[o] : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int r = wave + 4 * (rb + j);
#pragma unroll
                for (int u = 0; u < CTW; ++u)
                    if (lane + 64 * u < xc)
                        xs[r * xw + lane + 64 * u] = xv[j][u];
#pragma unroll
                for (int u = 0; u < YL; ++u)
                    if (lane + 64 * u < OG * 16)
                        dys[r * yw + lane + 64 * u] = yv[j][u];
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int s = 0; s < 16; ++s) {
            const float *xr = xs + (4 * s + g) * xw + i;
            const float *yr = dys + (4 * s + g) * yw + i;
#pragma unroll
            for (int og = 0; og < OG; ++og) {
                const float av = yr[og * 16];
#pragma unroll
                for (int t = 0; t < CTW; ++t)
                    if (wave + 4 * t < a.ct)
                        acc[og][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xr[(wave + 4 * t) * 16], acc[og][t], 0, 0, 0);
            }
        }
    }
    float *p = a.partial + (size_t)blockIdx.x * (OG * 16) * xc;
#pragma unroll
    for (int og = 0; og < OG; ++og)
#pragma unroll
        for (int t = 0; t < CTW; ++t)
            if (wave + 4 * t < a.ct) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    p[(size_t)(og * 16 + 4 * g + q) * xc + (wave + 4 * t) * 16 + i] = acc[og][t][q];
            }
}

// 64 elements of [dW | db] per workgroup, its four waves each a quarter of the row ranges' blocks (eight loads in
// flight per lane), then the four sums in a fixed order
__global__ __launch_bounds__(256) void linear_wgrad_all_reduce_kernel(int blocks, int cin, int cout, int opad, int cpad,
                                                                      const float *__restrict__ partial,
                                                                      float *__restrict__ dw, float *__restrict__ db)
{
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const bool live = e < cout * (cin + 1);
    const int o = live ? e / (cin + 1) : 0, c = live ? e - o * (cin + 1) : 0;
    const size_t step = (size_t)opad * cpad;
    const float *p = partial + (size_t)o * cpad + c;
    const int per = (blocks + 3) / 4, b0 = q * per, b1 = min(blocks, b0 + per);
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
        acc[u] = 0.f;
    for (int b = b0; b < b1; b += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            acc[u] += (live && b + u < b1) ? p[(size_t)(b + u) * step] : 0.f;
    }
    part[q][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (q == 0 && live) {
        const float s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        if (c < cin)
            dw[(size_t)o * cin + c] = s;
        else if (db)
            db[o] = s;
    }
}

struct WgradAllPlan {
    bool ok;
    int og, ct, ctw;
    long blocks, rpb;
    size_t lds, bytes;
};

WgradAllPlan wgrad_all_plan(long m, int cin, int cout)
{
    WgradAllPlan p;
    p.ok = cout <= 128 && cin + 1 <= 320;
    const int og = (cout + 15) / 16;
    p.og = og <= 1 ? 1 : og <= 2 ? 2 : og <= 4 ? 4 : 8;
    p.ct = (cin + 1 + 15) / 16;
    p.ctw = (p.ct + 3) / 4;
    // row ranges: ~160 workgroups for a Level's layers (10^4 rows: one or two tiles each); for the 3e5 edge rows of a
    // DenseEdgeConv block (a streaming read of 107 MB) up to 1024 (four per compute unit: 512 measured 39 vs 32 us), five tiles each
    const long tiles = (m + 63) / 64;
    long blocks = tiles < 160 ? tiles : 160;
    if (tiles / 4 > blocks) blocks = tiles / 4 < 1024 ? tiles / 4 : 1024;
    if (blocks < 1) blocks = 1;
    p.rpb = ((tiles + blocks - 1) / blocks) * 64;
    p.blocks = m > 0 ? (m + p.rpb - 1) / p.rpb : 1;
    p.lds = (size_t)64 * (p.ct * 16 + 1 + p.og * 16 + 1) * sizeof(float);
    p.bytes = (size_t)p.blocks * p.og * 16 * p.ct * 16 * sizeof(float);
    return p;
}

template <int OG, int CTW>
int wgrad_all_launch(hipStream_t s, const WgradAllPlan &p, const WgradAllArgs &a)
{
    auto kern = linear_wgrad_all_kernel<OG, CTW>;
    if constexpr (CTW == 1) {
        const bool flat = a.xs == a.cin && a.dys == a.cout && a.cin % 4 == 0 && a.cout % 4 == 0 && a.cin <= 60 &&
                          (((uintptr_t)a.x | (uintptr_t)a.dy) & 15) == 0;
        if (flat)
            kern = linear_wgrad_all_kernel<OG, 1, true>;
    }
    if (p.lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)p.blocks), dim3(256), p.lds, s, a);
    return TPU3_OK;
}

template <int OG>
int wgrad_all_dispatch(hipStream_t s, const WgradAllPlan &p, const WgradAllArgs &a)
{
    switch (p.ctw) {
    case 1: return wgrad_all_launch<OG, 1>(s, p, a);
    case 2: return wgrad_all_launch<OG, 2>(s, p, a);
    case 3: return wgrad_all_launch<OG, 3>(s, p, a);
    case 4: return wgrad_all_launch<OG, 4>(s, p, a);
    default: return wgrad_all_launch<OG, 5>(s, p, a);
    }
}

struct WgradWidePlan {
    int cg, og;
    long blocks, rpb;
    size_t bytes;
};

WgradWidePlan wgrad_wide_plan(long m, int cin, int cout)
{
    WgradWidePlan p;
    p.cg = (cin + 1 + 63) / 64;
    p.og = (cout + 15) / 16;
    long blocks = 2048 / (p.cg * p.og);                     // ~2048 workgroups in flight
    const long most = (m + 4 * WG_ROWS - 1) / (4 * WG_ROWS);
    if (blocks > most) blocks = most;
    if (blocks < 1) blocks = 1;
    p.rpb = (m + blocks - 1) / blocks;
    p.rpb = (p.rpb + WG_ROWS - 1) / WG_ROWS * WG_ROWS;
    if (p.rpb < WG_ROWS) p.rpb = WG_ROWS;
    p.blocks = m > 0 ? (m + p.rpb - 1) / p.rpb : 1;
    p.bytes = (size_t)p.blocks * p.og * 16 * p.cg * 64 * sizeof(float);
    return p;
}

} // namespace

namespace {

// Input gradient of a per-point layer with FEW outputs (the lift and the prep convolutions, 24 outputs: autograd's
// GEMM has K = 24 and runs a 256-deep tile, 56 us): dx[i][c] = sum_o dy[i][o] * w[o][c].  The weight sits in LDS, a
// wave takes four rows at a time, lanes across the columns; a row of dy is read by the first lanes and handed round
// with v_readlane.  Bound by the write of dx (8 MB for 9984 x 204).
constexpr int DG_OMAX = 32, DG_CMAX = 320, DG_ROWS = 4;

struct DgradArgs {
    long m;
    int cin, cout, dys, dxs;
    const float *dy, *w;
    float *dx;
};

__global__ __launch_bounds__(256) void linear_dgrad_small_kernel(DgradArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float dg_w[];       // [cout][cin]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < a.cout * a.cin; e += 256)
        dg_w[e] = a.w[e];
    __syncthreads();
    const long stride = (long)gridDim.x * 4 * DG_ROWS;
    for (long r0 = ((long)blockIdx.x * 4 + wave) * DG_ROWS; r0 < a.m; r0 += stride) {
        float gv[DG_ROWS];
#pragma unroll
        for (int j = 0; j < DG_ROWS; ++j)
            gv[j] = (r0 + j < a.m && lane < a.cout) ? a.dy[(r0 + j) * a.dys + lane] : 0.f;
        float acc[DG_ROWS][DG_CMAX / 64];
#pragma unroll
        for (int j = 0; j < DG_ROWS; ++j)
#pragma unroll
            for (int u = 0; u < DG_CMAX / 64; ++u)
                acc[j][u] = 0.f;
        for (int o = 0; o < a.cout; ++o) {
            float g[DG_ROWS];
#pragma unroll
            for (int j = 0; j < DG_ROWS; ++j)
                g[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gv[j]), o));
#pragma unroll
            for (int u = 0; u < DG_CMAX / 64; ++u) {
                const int c = lane + 64 * u;
                const float wv = c < a.cin ? dg_w[o * a.cin + c] : 0.f;
#pragma unroll
                for (int j = 0; j < DG_ROWS; ++j)
                    acc[j][u] = __builtin_fmaf(g[j], wv, acc[j][u]);
            }
        }
#pragma unroll
        for (int j = 0; j < DG_ROWS; ++j)
            if (r0 + j < a.m) {
#pragma unroll
                for (int u = 0; u < DG_CMAX / 64; ++u)
                    if (lane + 64 * u < a.cin)
                        a.dx[(r0 + j) * a.dxs + lane + 64 * u] = acc[j][u];
            }
    }
}

} // namespace

extern "C" int tpu3_linear_dgrad_f32(tpu3_stream_t stream, long m, int cin, int cout, const float *dy, int dy_stride,
                                     const float *w, float *dx, int dx_stride)
{
    if (m < 0 || cin <= 0 || cout <= 0 || dy_stride < cout || dx_stride < cin) return TPU3_EINVAL;
    if (cout > DG_OMAX || cin > DG_CMAX) return TPU3_ELIMIT;
    if (m == 0) return TPU3_OK;
    if (!dy || !w || !dx) return TPU3_EINVAL;
    long blocks = (m + 4 * DG_ROWS - 1) / (4 * DG_ROWS);
    if (blocks > 1024) blocks = 1024;
    DgradArgs a{m, cin, cout, dy_stride, dx_stride, dy, w, dx};
    hipLaunchKernelGGL(linear_dgrad_small_kernel, dim3((unsigned)blocks), dim3(256),
                       (size_t)cin * cout * sizeof(float), (hipStream_t)stream, a);
    return tpu3_launch_status();
}

extern "C" size_t tpu3_linear_wgrad_bias_workspace_bytes(long m, int cin, int cout)
{
    if (m < 0 || cin <= 0 || cout <= 0) return 0;
    const WgradAllPlan q = wgrad_all_plan(m, cin, cout);
    return q.ok ? q.bytes : wgrad_wide_plan(m, cin, cout).bytes;
}

extern "C" int tpu3_linear_wgrad_bias_f32(tpu3_stream_t stream, long m, int cin, int cout, const float *x,
                                          int x_stride, const float *dy, int dy_stride, float *dw, float *db,
                                          void *workspace, size_t workspace_bytes)
{
    if (m < 0 || cin <= 0 || cout <= 0 || x_stride < cin || dy_stride < cout) return TPU3_EINVAL;
    if (cin > 1023 || cout > 1024) return TPU3_ELIMIT;
    if (!dw) return TPU3_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (m == 0) {
        hipError_t e = hipMemsetAsync(dw, 0, (size_t)cin * cout * sizeof(float), s);
        if (e == hipSuccess && db)
            e = hipMemsetAsync(db, 0, (size_t)cout * sizeof(float), s);
        return (int)e;
    }
    if (!x || !dy) return TPU3_EINVAL;
    const WgradAllPlan q = wgrad_all_plan(m, cin, cout);
    if (q.ok) {
        if (!workspace || workspace_bytes < q.bytes) return TPU3_EINVAL;
        WgradAllArgs b{m, cin, cout, x_stride, dy_stride, x, dy, (float *)workspace, q.rpb, q.ct};
        int r;
        switch (q.og) {
        case 1: r = wgrad_all_dispatch<1>(s, q, b); break;
        case 2: r = wgrad_all_dispatch<2>(s, q, b); break;
        case 4: r = wgrad_all_dispatch<4>(s, q, b); break;
        default: r = wgrad_all_dispatch<8>(s, q, b); break;
        }
        if (r) return r;
        hipLaunchKernelGGL(linear_wgrad_all_reduce_kernel, dim3((cout * (cin + 1) + 63) / 64), dim3(256), 0, s,
                           (int)q.blocks, cin, cout, q.og * 16, q.ct * 16, (const float *)workspace, dw, db);
        return tpu3_launch_status();
    }
    const WgradWidePlan p = wgrad_wide_plan(m, cin, cout);
    if (!workspace || workspace_bytes < p.bytes) return TPU3_EINVAL;
    WgradWideArgs a{m, cin, cout, x_stride, dy_stride, x, dy, (float *)workspace, p.rpb, p.cg, p.og};
    hipLaunchKernelGGL(linear_wgrad_wide_kernel, dim3((unsigned)p.blocks, (unsigned)(p.cg * p.og)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(linear_wgrad_wide_reduce_kernel, dim3((cout * (cin + 1) + 255) / 256), dim3(256), 0, s,
                       (int)p.blocks, cin, cout, p.cg, p.og, (const float *)workspace, dw, db);
    return tpu3_launch_status();
}

// ---- the weight and bias gradients of a DenseEdgeConv block from what its backward kernel leaves behind ---------
// (csrc/dec_train.hip: one block G^T Z per workgroup of the backward launch -- G (edges,36) = [g2 | g1 | g0], Z (edges,48)
// = [h1 | h0 | d_j] never reach memory since round 4 --, S (points,36) = G summed over a point's edges.)  One streaming
// pass S^T [x | 1] over the points and ONE kernel that adds the blocks and writes the three layers' gradients in their
// own layout:
//   W_2 (12,48) = [G2^T Z[:, 0:24]  | S2^T x],  W_1 (12,36) = [G1^T Z[:, 12:24] | S1^T x],
//   W_0 (12,48) = [S0^T x           | G0^T Z[:, 24:48]],  biases = column sums of S.
namespace {

__global__ __launch_bounds__(1024) void dec_wgrad_assemble_kernel(int blocks_e, int blocks_p, const float *__restrict__ pe,
                                                                 const float *__restrict__ pp, float *__restrict__ gw0,
                                                                 float *__restrict__ gw1, float *__restrict__ gw2,
                                                                 float *__restrict__ gb)
{
    // partial layouts: edges (blocks_e, 64, 64) (36 x 49 used), points (blocks_p, 64, 32) (36 x 25 used)
    // (16 waves share the row ranges' blocks of an output: with 4, the up-to-1024 partial blocks were 256 dependent
    // strided loads per thread on 26 compute units -- 22 us per call, 0.36 ms of a training step)
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    constexpr int N2 = 12 * 48, N1 = 12 * 36, N0 = 12 * 48, NB = 36;
    const bool live = e < N2 + N1 + N0 + NB;
    bool edge = false;
    int row = 0, col = 0;
    float *dst = gb;
    if (e < N2) {
        const int o = e / 48, c = e - o * 48;
        edge = c < 24; row = o; col = edge ? c : c - 24; dst = gw2 + e;
    } else if (e < N2 + N1) {
        const int f = e - N2, o = f / 36, c = f - o * 36;
        edge = c < 12; row = 12 + o; col = edge ? 12 + c : c - 12; dst = gw1 + f;
    } else if (e < N2 + N1 + N0) {
        const int f = e - N2 - N1, o = f / 48, c = f - o * 48;
        edge = c >= 24; row = 24 + o; col = c; if (!edge) col = c;
        dst = gw0 + f;
    } else if (live) {
        row = e - N2 - N1 - N0; col = 24; dst = gb + row;
    }
    const int blocks = edge ? blocks_e : blocks_p;
    const size_t step = edge ? 64 * 64 : 64 * 32;
    const float *p = (edge ? pe : pp) + (size_t)row * (edge ? 64 : 32) + col;
    const int per = (blocks + 15) / 16, b0 = q * per, b1 = min(blocks, b0 + per);
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
        acc[u] = 0.f;
    for (int b = b0; b < b1; b += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            acc[u] += (live && b + u < b1) ? p[(size_t)(b + u) * step] : 0.f;
    }
    part[q][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (q == 0 && live) {
        float t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            t[u] = (part[4 * u][lane] + part[4 * u + 1][lane]) + (part[4 * u + 2][lane] + part[4 * u + 3][lane]);
        *dst = (t[0] + t[1]) + (t[2] + t[3]);
    }
}

} // namespace

int tpu3_dec_train_bwd_blocks(long points);           // csrc/dec_train.hip

extern "C" size_t tpu3_dec_train_wgrad_workspace_bytes(long points)
{
    if (points <= 0) return 0;
    return (size_t)tpu3_dec_train_bwd_blocks(points) * 4096 * sizeof(float) + wgrad_all_plan(points, 24, 36).bytes;
}

extern "C" int tpu3_dec_train_wgrad_f32(tpu3_stream_t stream, long points, const float *x, const float *S, float *gw0,
                                        float *gw1, float *gw2, float *gb, void *workspace, size_t workspace_bytes)
{
    if (points <= 0) return TPU3_EINVAL;
    if (!x || !S || !gw0 || !gw1 || !gw2 || !gb) return TPU3_EINVAL;
    const int blocks_e = tpu3_dec_train_bwd_blocks(points);
    const size_t ebytes = (size_t)blocks_e * 4096 * sizeof(float);
    const WgradAllPlan pp = wgrad_all_plan(points, 24, 36);
    if (!workspace || workspace_bytes < ebytes + pp.bytes) return TPU3_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float *we = (float *)workspace, *wp = (float *)((char *)workspace + ebytes);
    WgradAllArgs ap{points, 24, 36, 24, 36, x, S, wp, pp.rpb, pp.ct};
    const int r = wgrad_all_launch<4, 1>(s, pp, ap);    // (25 columns: 2 tiles)
    if (r) return r;
    hipLaunchKernelGGL(dec_wgrad_assemble_kernel, dim3((12 * 48 * 2 + 12 * 36 + 36 + 63) / 64), dim3(1024), 0, s,
                       blocks_e, (int)pp.blocks, (const float *)we, (const float *)wp, gw0, gw1, gw2, gb);
    return tpu3_launch_status();
}

extern "C" size_t tpu3_linear_wgrad_workspace_bytes(long m)
{
    long blocks = (m + 4 * WG_ROWS - 1) / (4 * WG_ROWS);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    return (size_t)blocks * 16 * WG_CMAX * sizeof(float);
}

extern "C" int tpu3_linear_wgrad_f32(tpu3_stream_t stream, long m, int cin, int cout, const float *x,
                                     int x_stride, const float *dy, int dy_stride, float *dw, void *workspace,
                                     size_t workspace_bytes)
{
    if (m < 0 || cin <= 0 || cout <= 0 || x_stride < cin || dy_stride < cout) return TPU3_EINVAL;
    if (cin > WG_CMAX || cout > 16) return TPU3_ELIMIT;
    if (!dw) return TPU3_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (m == 0) {
        const hipError_t e = hipMemsetAsync(dw, 0, (size_t)cin * cout * sizeof(float), s);
        return (int)e;
    }
    if (!x || !dy) return TPU3_EINVAL;
    const size_t need = tpu3_linear_wgrad_workspace_bytes(m);
    if (!workspace || workspace_bytes < need) return TPU3_EINVAL;
    long blocks = (m + 4 * WG_ROWS - 1) / (4 * WG_ROWS);
    if (blocks > 1024) blocks = 1024;
    long rpb = (m + blocks - 1) / blocks;
    rpb = (rpb + WG_ROWS - 1) / WG_ROWS * WG_ROWS;
    blocks = (m + rpb - 1) / rpb;
    WgradArgs a{m, cin, cout, x_stride, dy_stride, x, dy, (float *)workspace, rpb};
    switch ((cin + 15) / 16) {
    case 1: hipLaunchKernelGGL(linear_wgrad_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(linear_wgrad_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL(linear_wgrad_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(linear_wgrad_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, a); break;
    }
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((cin * cout + 3) / 4), dim3(256), 0, s, (int)blocks, cin,
                       cout, (const float *)workspace, dw);
    return tpu3_launch_status();
}
