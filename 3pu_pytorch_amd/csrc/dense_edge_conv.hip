// dense_edge_conv.hip -- fused DenseEdgeConv block (inference) for gfx950 on fp32 MFMA.
//
// Replaces the body of the reference's DenseEdgeConv.forward (network/layers.py:44-64) for the
// configuration every Level uses (24 input channels, growth rate 12, 3 dense layers):
//
//   e_ij = [x_i, x_j - x_i]                       (48)      j in kNN(i), k neighbours
//   h0   = relu(W0 e_ij + b0)                     (12)
//   h1   = relu(W1 [h0, x_i] + b1)                (12)
//   h2   =      W2 [h1, h0, x_i] + b2             (12)
//   y_i  = max_j [h2, h1, h0, x_i]                (60)
//
// The reference materialises (B,48..60,N,k) tensors between six ATen kernels (1x1 convolutions with
// 12 output channels, concatenations, max): ~4.6 GB per tensor at the level-4 batch, memory bound.
// Here a workgroup owns one patch: its (N,24) features sit in LDS, every edge's 60-channel vector
// lives only in MFMA accumulators, and HBM sees x once, the neighbour indices once and y once.
//
// MFMA mapping (v_mfma_f32_16x16x4_f32, exact fp32 fma chains): rows = output channels (12 of 16),
// columns = 16 edges of one point, K = input channels 4 at a time.
//   * A operand = weights: lane (m = l&15, g = l>>4) keeps W[m][chan(step,g)] in VGPRs for the
//     whole kernel (38 registers);
//   * B operand = inputs: lane (edge = l&15, g) supplies channel chan(step,g) of its edge;
//   * D = 4 consecutive channels (4g..4g+3) of the lane's edge -- exactly what the next layer's B
//     operand needs when its step r uses channel 4g+r, so h0/h1 feed the next MFMA straight from
//     the accumulator registers, no shuffles, no LDS;
//   * everything that depends on ONE point only is hoisted out of the edge loop:
//       W0 e = (W0a - W0b) x_i + W0b x_j,  W1 [h0,x_i] = W1a h0 + W1b x_i,  W2 [..] = W2a h1 + W2b h0 + W2c x_i
//     the x_i terms (+ bias) are computed for 16 points at a time with the same MFMA and become the
//     accumulators' initial values, and z_j = W0b x_j is a per-POINT table (12 floats per point in
//     LDS, computed once per patch): the first layer of an edge is a gather of z_j, an add and a
//     ReLU -- 12 MFMAs per 16-edge tile instead of 33, and 12 instead of 24 gathered floats per edge;
//   * max over the k edges = elementwise max over the tiles, then one 16-lane DPP row reduction.
// Summation order differs from a BLAS GEMM (documented tolerance 1e-5 on the network outputs).
#include "tpu3_dev.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DEC_C = 24;        // input channels
constexpr int DEC_G = 12;        // growth rate
constexpr int DEC_S = 26;        // LDS row stride of the patch features (floats)
constexpr int DEC_NW = 8;        // waves per workgroup
constexpr int DEC_ZS = 12;       // floats per point in the z table
constexpr int DEC_TS = 52;       // floats per point in the per-wave T buffer (3 x 16 + pad)

struct DecArgs {
    int n;                       // points per patch
    int k;                       // neighbours per point (multiple of 16)
    const float *x;              // (P, n, 24)
    const void *idx;             // (P, n, idx_stride) neighbour indices, first `idx_off` skipped
    int idx64, idx_stride, idx_off;
    const float *w0, *b0;        // (12,48), (12)
    const float *w1, *b1;        // (12,36), (12)
    const float *w2, *b2;        // (12,48), (12)
    float *out;                  // (P, n, out_stride): y written at channels [0,60)
    int out_stride;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float row_max_f32(float v)
{
    // max over the 16 lanes of a DPP row; every lane of the row gets the result
    v = fmaxf(v, __int_as_float(tpu3_dpp<0xB1>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(tpu3_dpp<0x4E>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(tpu3_dpp<0x141>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(tpu3_dpp<0x140>(__float_as_int(v))));
    return v;
}

// One DPP step of the 16-lane row max for twelve values at once: the modifier is fused into
// v_max_f32 (hipcc emits v_mov_dpp + v_max + wait states per step and value), and the twelve
// independent chains fill each other's DPP wait states.
#define DEC_DPP_MAX12(CTRL)                                                                              \
    asm volatile("s_nop 1\n\t"                                                                           \
                 "v_max_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_max_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_max_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_max_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_max_f32_dpp %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_max_f32_dpp %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_max_f32_dpp %6, %6, %6 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_max_f32_dpp %7, %7, %7 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_max_f32_dpp %8, %8, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_max_f32_dpp %9, %9, %9 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                       \
                 "v_max_f32_dpp %10, %10, %10 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                    \
                 "v_max_f32_dpp %11, %11, %11 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                    \
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8),  \
                   "+v"(v9), "+v"(v10), "+v"(v11))

__device__ __forceinline__ void row_max12(f32x4 &a, f32x4 &b, f32x4 &c)
{
    float v0 = a[0], v1 = a[1], v2 = a[2], v3 = a[3], v4 = b[0], v5 = b[1], v6 = b[2], v7 = b[3];
    float v8 = c[0], v9 = c[1], v10 = c[2], v11 = c[3];
    DEC_DPP_MAX12("quad_perm:[1,0,3,2]");
    DEC_DPP_MAX12("quad_perm:[2,3,0,1]");
    DEC_DPP_MAX12("row_half_mirror");
    DEC_DPP_MAX12("row_mirror");
    a = (f32x4){v0, v1, v2, v3};
    b = (f32x4){v4, v5, v6, v7};
    c = (f32x4){v8, v9, v10, v11};
}

// ZTAB: the per-point table z = W0b x fits LDS next to the features (patches up to ~700 points);
// otherwise the first layer's W0b x_j is evaluated per edge (6 more MFMAs per tile).
template <int TILES, bool ZTAB>     // TILES = k / 16
__global__ __launch_bounds__(DEC_NW * 64) void dec_fused_kernel(DecArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *xs = lds;                                   // n * DEC_S
    float *tb = lds + ((a.n * DEC_S + 3) & ~3);        // DEC_NW * 16 * DEC_TS
    float *zt = tb + DEC_NW * 16 * DEC_TS;             // n * DEC_ZS (+ 4 floats of slack)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e = lane & 15, g = lane >> 4;
    const int n = a.n;
    const float *X = a.x + (size_t)blockIdx.x * n * DEC_C;
    float *O = a.out + (size_t)blockIdx.x * n * a.out_stride;

    // ---- the patch's features -> LDS; x_i also goes straight to the output (channels 36..59) ------
    for (int t = tid; t < n * DEC_C; t += DEC_NW * 64) {
        const int i = t / DEC_C, c = t - i * DEC_C;
        const float v = X[t];
        xs[i * DEC_S + c] = v;
        O[(size_t)i * a.out_stride + 3 * DEC_G + c] = v;
    }

    // ---- weights: lane (m = e, g) holds the A operands of every step -----------------------------
    const int m = e;
    const bool live = m < DEC_G;
    float wt0[6], wt1[6], wt2[6];    // centre terms: (W0a-W0b), W1b, W2c   channel 6g+s
    float w0b[6];                    // W0b                                 channel 6g+s
    float w1a[4], w2a[4], w2b[4];    // W1a, W2a (h1), W2b (h0)             channel 4g+r  (g = 3: padding)
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int c = 6 * g + s;
        const float wa = live ? a.w0[m * 48 + c] : 0.f, wb = live ? a.w0[m * 48 + 24 + c] : 0.f;
        wt0[s] = wa - wb;
        w0b[s] = wb;
        wt1[s] = live ? a.w1[m * 36 + 12 + c] : 0.f;
        wt2[s] = live ? a.w2[m * 48 + 24 + c] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = 4 * g + r;
        const bool ok = live && c < DEC_G;
        w1a[r] = ok ? a.w1[m * 36 + c] : 0.f;
        w2a[r] = ok ? a.w2[m * 48 + c] : 0.f;
        w2b[r] = ok ? a.w2[m * 48 + 12 + c] : 0.f;
    }
    // bias of the 4 channels this lane's accumulator rows hold (rows 4g..4g+3)
    f32x4 bias0, bias1, bias2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = 4 * g + r;
        bias0[r] = c < DEC_G ? a.b0[c] : 0.f;
        bias1[r] = c < DEC_G ? a.b1[c] : 0.f;
        bias2[r] = c < DEC_G ? a.b2[c] : 0.f;
    }
    __syncthreads();

    // ---- z_p = W0b x_p for every point of the patch (columns = points) ------------------------------
    for (int pb = wave * 16; ZTAB && pb < n; pb += DEC_NW * 16) {
        const float *xp = xs + min(pb + e, n - 1) * DEC_S + 6 * g;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 6; ++s)
            z = mfma4(w0b[s], xp[s], z);
        if (g < 3 && pb + e < n)
            *(f32x4 *)(zt + (pb + e) * DEC_ZS + 4 * g) = z;
    }
    __syncthreads();
    const int zoff = g < 3 ? 4 * g : 0;     // lanes of the padding channel group read finite values (x 0 weights)

    float *T = tb + wave * 16 * DEC_TS;
    for (int pb = wave * 16; pb < n; pb += DEC_NW * 16) {
        // ---- centre terms of 16 points (columns = points) -> T[p][0..15 | 16..31 | 32..47] --------
        {
            const int p = min(pb + e, n - 1);
            const float *xp = xs + p * DEC_S + 6 * g;
            f32x4 t0 = bias0, t1 = bias1, t2 = bias2;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const float xv = xp[s];
                t0 = mfma4(wt0[s], xv, t0);
                t1 = mfma4(wt1[s], xv, t1);
                t2 = mfma4(wt2[s], xv, t2);
            }
            *(f32x4 *)(T + e * DEC_TS + 4 * g) = t0;
            *(f32x4 *)(T + e * DEC_TS + 16 + 4 * g) = t1;
            *(f32x4 *)(T + e * DEC_TS + 32 + 4 * g) = t2;
        }
        const int pend = min(16, n - pb);
        // neighbour indices of the point about to be processed (one register per tile); the next
        // point's are requested while the current one computes, so the global-load latency is hidden
        auto load_idx = [&](int i, int (&j)[TILES]) {
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                const size_t io = ((size_t)blockIdx.x * n + i) * a.idx_stride + a.idx_off + 16 * t + e;
                j[t] = a.idx64 ? (int)((const long long *)a.idx)[io] : ((const int *)a.idx)[io];
            }
        };
        int jn[TILES];
        load_idx(pb, jn);
        for (int p = 0; p < pend; ++p) {
            const int i = pb + p;
            const float *zj[TILES];
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                const int j = min(max(jn[t], 0), n - 1);
                zj[t] = ZTAB ? zt + j * DEC_ZS + zoff : xs + j * DEC_S + 6 * g;
            }
            if (p + 1 < pend)
                load_idx(i + 1, jn);
            const f32x4 c0 = *(const f32x4 *)(T + p * DEC_TS + 4 * g);
            const f32x4 c1 = *(const f32x4 *)(T + p * DEC_TS + 16 + 4 * g);
            const f32x4 c2 = *(const f32x4 *)(T + p * DEC_TS + 32 + 4 * g);
            // the TILES edge tiles of the point are independent accumulator chains: issuing their
            // MFMAs alternately hides the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32
            f32x4 h0[TILES], h1[TILES], h2[TILES];
            if (ZTAB) {
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    const f32x4 z = *(const f32x4 *)zj[t];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        h0[t][r] = fmaxf(c0[r] + z[r], 0.f);
                }
            } else {
#pragma unroll
                for (int t = 0; t < TILES; ++t)
                    h0[t] = c0;
#pragma unroll
                for (int s = 0; s < 6; ++s)
#pragma unroll
                    for (int t = 0; t < TILES; ++t)
                        h0[t] = mfma4(w0b[s], zj[t][s], h0[t]);
#pragma unroll
                for (int t = 0; t < TILES; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        h0[t][r] = fmaxf(h0[t][r], 0.f);
            }
#pragma unroll
            for (int t = 0; t < TILES; ++t)
                h1[t] = c1;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < TILES; ++t)
                    h1[t] = mfma4(w1a[r], h0[t][r], h1[t]);
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    h1[t][r] = fmaxf(h1[t][r], 0.f);
                h2[t] = c2;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < TILES; ++t)
                    h2[t] = mfma4(w2a[r], h1[t][r], h2[t]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < TILES; ++t)
                    h2[t] = mfma4(w2b[r], h0[t][r], h2[t]);
            f32x4 m0 = h0[0], m1 = h1[0], m2 = h2[0];
#pragma unroll
            for (int t = 1; t < TILES; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m0[r] = fmaxf(m0[r], h0[t][r]);
                    m1[r] = fmaxf(m1[r], h1[t][r]);
                    m2[r] = fmaxf(m2[r], h2[t][r]);
                }
            // ---- max over the 16 edges of the row; one lane per channel group writes ----------------
            row_max12(m0, m1, m2);
            if (e == 0 && g < 3) {
                float *o = O + (size_t)i * a.out_stride + 4 * g;
                *(f32x4 *)(o) = m2;                  // [0,12)  max h2
                *(f32x4 *)(o + DEC_G) = m1;          // [12,24) max h1
                *(f32x4 *)(o + 2 * DEC_G) = m0;      // [24,36) max h0
            }
        }
    }
}

} // namespace

// Internal entry (declared in include/tpu3.h as tpu3_dense_edge_conv_f32).
extern "C" int tpu3_dense_edge_conv_f32(tpu3_stream_t stream, int patches, int n, int k, const float *x,
                                        const void *idx, int idx_elem_size, int idx_stride, int idx_off,
                                        const float *w0, const float *b0, const float *w1, const float *b1,
                                        const float *w2, const float *b2, float *out, int out_stride)
{
    if (patches < 0 || n <= 0 || k <= 0 || (k % 16) != 0 || k > 64) return TPU3_EINVAL;
    if (idx_elem_size != 4 && idx_elem_size != 8) return TPU3_EINVAL;
    if (idx_off < 0 || idx_stride < idx_off + k || out_stride < 60 || (out_stride % 4) != 0) return TPU3_EINVAL;
    if (patches == 0) return TPU3_OK;
    if (!x || !idx || !w0 || !b0 || !w1 || !b1 || !w2 || !b2 || !out) return TPU3_EINVAL;
    if (((uintptr_t)out % 16) != 0) return TPU3_EINVAL;
    const size_t base = (size_t)((n * DEC_S + 3) & ~3) + (size_t)DEC_NW * 16 * DEC_TS;
    const size_t with_z = (base + (size_t)n * DEC_ZS + 4) * sizeof(float);
    const bool ztab = with_z <= 160 * 1024;
    const size_t lds = ztab ? with_z : base * sizeof(float);
    if (lds > 160 * 1024) return TPU3_ELIMIT;
    DecArgs a{n, k, x, idx, idx_elem_size == 8, idx_stride, idx_off, w0, b0, w1, b1, w2, b2, out, out_stride};
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipSuccess;
#define DEC_LAUNCH1(T, Z)                                                                                \
    e = hipFuncSetAttribute((const void *)dec_fused_kernel<T, Z>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)lds);                                                                   \
    if (e != hipSuccess) return (int)e;                                                                  \
    hipLaunchKernelGGL((dec_fused_kernel<T, Z>), dim3(patches), dim3(DEC_NW * 64), lds, s, a)
#define DEC_LAUNCH(T)                                                                                    \
    if (ztab) { DEC_LAUNCH1(T, true); } else { DEC_LAUNCH1(T, false); }
    switch (k / 16) {
    case 1: DEC_LAUNCH(1); break;
    case 2: DEC_LAUNCH(2); break;
    case 3: DEC_LAUNCH(3); break;
    default: DEC_LAUNCH(4); break;
    }
#undef DEC_LAUNCH1
#undef DEC_LAUNCH
    return tpu3_launch_status();
}
