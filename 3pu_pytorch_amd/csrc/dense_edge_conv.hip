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
// columns = 16 POINTS (one neighbour slot of each at a time), K = input channels 4 at a time.
//   * A operand = weights: lane (m = l&15, g = l>>4) keeps W[m][chan(step,g)] in VGPRs for the
//     whole kernel (38 registers);
//   * B operand = inputs: lane (point = l&15, g) supplies channel chan(step,g) of its point's neighbour;
//   * D = 4 consecutive channels (4g..4g+3) of the lane's edge (point, slot) -- exactly what the next layer's B
//     operand needs when its step r uses channel 4g+r, so h0/h1 feed the next MFMA straight from
//     the accumulator registers, no shuffles, no LDS;
//   * everything that depends on ONE point only is hoisted out of the edge loop:
//       W0 e = (W0a - W0b) x_i + W0b x_j,  W1 [h0,x_i] = W1a h0 + W1b x_i,  W2 [..] = W2a h1 + W2b h0 + W2c x_i
//     the x_i terms (+ bias) are computed for 16 points at a time with the same MFMA and become the
//     accumulators' initial values, and z_j = W0b x_j is a per-POINT table (12 floats per point in
//     LDS, computed once per patch): the first layer of an edge is a gather of z_j, an add and a
//     ReLU -- 12 MFMAs per 16-edge tile instead of 33, and 12 instead of 24 gathered floats per edge;
//   * max over the k edges = a running per-lane v_max over the slots (v_max is a half-rate VALU
//     instruction on gfx950 and fp32 MFMA shares the VALU datapath: with 16 edges of one point along the
//     columns, the 16-lane DPP row reduction per point cost 18 % of the kernel).
// Summation order differs from a BLAS GEMM (documented tolerance 1e-5 on the network outputs).
#include "tpu3_dev.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DEC_C = 24;        // input channels
constexpr int DEC_G = 12;        // growth rate
constexpr int DEC_S = 26;        // LDS row stride of the patch features (floats)
constexpr int DEC_NW = 8;        // waves per workgroup
constexpr int DEC_ZS = 12;       // floats per point in the z table
constexpr int DEC_TS = 64;       // words per point-slot row of the per-wave neighbour tile (k <= 64)

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

// max(v, 0) as ONE v_max_f32: fmaxf() on an MFMA result costs a second (canonicalising) v_max, and
// v_max is a half-rate instruction on gfx950
__device__ __forceinline__ float dec_relu(float v)
{
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
    return r;
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

    // ---- 16 points per wave step: columns = POINTS, one neighbour slot at a time ---------------------
    // Everything a lane holds belongs to its own point pb + e: the centre terms stay in the registers
    // the MFMA left them in, the max over the k neighbours is a running per-lane v_max (no cross-lane
    // reduction), and lane (e, g) gathers rows 4g..4g+3 of ITS point's neighbour.
    constexpr int K = 16 * TILES;
    constexpr int U = 2;                        // neighbour slots in flight (independent MFMA chains)
    int *IT = (int *)(tb + wave * 16 * DEC_TS);  // per-wave tile [slot][16 points]: byte offset of the neighbour's row
    const char *gbase = ZTAB ? (const char *)(zt + zoff) : (const char *)(xs + 6 * g);
    for (int pb = wave * 16; pb < n; pb += DEC_NW * 16) {
        // neighbour indices of the 16 points, clamped and scaled once, read coalesced (k contiguous per point)
        for (int t = lane; t < 16 * K; t += 64) {
            const int pt = t / K, sl = t - pt * K;
            const int i = min(pb + pt, n - 1);
            const size_t io = ((size_t)blockIdx.x * n + i) * a.idx_stride + a.idx_off + sl;
            int j = a.idx64 ? (int)((const long long *)a.idx)[io] : ((const int *)a.idx)[io];
            j = min(max(j, 0), n - 1);
            IT[sl * 16 + pt] = j * (int)((ZTAB ? DEC_ZS : DEC_S) * sizeof(float));
        }
        // centre terms of the lane's point (+ bias): the initial values of the three accumulators
        f32x4 c0 = bias0, c1 = bias1, c2 = bias2;
        {
            const float *xp = xs + min(pb + e, n - 1) * DEC_S + 6 * g;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const float xv = xp[s];
                c0 = mfma4(wt0[s], xv, c0);
                c1 = mfma4(wt1[s], xv, c1);
                c2 = mfma4(wt2[s], xv, c2);
            }
        }
        __builtin_amdgcn_wave_barrier();
        const float ninf = -__builtin_inff();
        f32x4 m0 = {ninf, ninf, ninf, ninf}, m1 = m0, m2 = m0;
#pragma unroll 1
        for (int kk = 0; kk < K; kk += U) {
            f32x4 h0[U], h1[U], h2[U];
            int off[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                off[u] = IT[(kk + u) * 16 + e];
            if (ZTAB) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const f32x4 z = *(const f32x4 *)(gbase + off[u]);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        h0[u][r] = fmaxf(c0[r] + z[r], 0.f);
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    h0[u] = c0;
#pragma unroll
                for (int s = 0; s < 6; ++s)
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        h0[u] = mfma4(w0b[s], ((const float *)(gbase + off[u]))[s], h0[u]);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        h0[u][r] = dec_relu(h0[u][r]);
            }
            // the U slots are independent accumulator chains: issuing their MFMAs alternately hides the
            // 40-cycle dependent latency of v_mfma_f32_16x16x4_f32
#pragma unroll
            for (int u = 0; u < U; ++u)
                h1[u] = c1;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int u = 0; u < U; ++u)
                    h1[u] = mfma4(w1a[r], h0[u][r], h1[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    h1[u][r] = dec_relu(h1[u][r]);
                h2[u] = c2;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int u = 0; u < U; ++u)
                    h2[u] = mfma4(w2a[r], h1[u][r], h2[u]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int u = 0; u < U; ++u)
                    h2[u] = mfma4(w2b[r], h0[u][r], h2[u]);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m0[r] = fmaxf(m0[r], h0[u][r]);
                    m1[r] = fmaxf(m1[r], h1[u][r]);
                    m2[r] = fmaxf(m2[r], h2[u][r]);
                }
        }
        if (g < 3 && pb + e < n) {
            float *o = O + (size_t)(pb + e) * a.out_stride + 4 * g;
            *(f32x4 *)(o) = m2;                  // [0,12)  max h2
            *(f32x4 *)(o + DEC_G) = m1;          // [12,24) max h1
            *(f32x4 *)(o + 2 * DEC_G) = m0;      // [24,36) max h0
        }
        __builtin_amdgcn_wave_barrier();
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
