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
// Here a workgroup owns one patch (or, for patches beyond ~2700 points, a 512-point slice of it): what
// an edge needs of its neighbour is a 12-float row of a per-point table (below) kept in LDS (in global
// memory / L2 for the large patches), every edge's 60-channel vector lives only in MFMA accumulators,
// and HBM sees x once, the neighbour indices once and y once -- y as one contiguous 240-byte run per point.
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

#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int DEC_C = 24;        // input channels
constexpr int DEC_G = 12;        // growth rate
constexpr int DEC_NW = 8;        // waves per workgroup
constexpr int DEC_ZS = 12;       // floats per point in the z table
constexpr int DEC_TS = 64;       // words per point-slot row of the per-wave neighbour tile (k <= 64)
constexpr int DEC_SLICE = 512;   // points per workgroup when a patch is split (global z table)

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
    float *zg;                   // (P, n, 12) z table in global memory (patches too large for LDS), else null
    // (r3) the next prep convolutions folded into this block's write-out (lane-per-point kernel only), see
    // tpu3_dense_edge_conv_fold_f32: fold_n in {0, 24, 48, 72} outputs over the block's 60-channel row
    int fold_n;
    const float *fold_w, *fold_b;    // (fold_n, 60) row-major; (fold_n) or null = the sums continue from `acc`
    float *acc;                      // (P, n, acc_stride) partial sums of the later prep convolutions
    int acc_stride, seed_off, store_off;
    float *xnext;                    // (P, n, 24): relu of the first 24 outputs = the next block's input rows
    int out_half;                    // `out` is a fp16 buffer (TPU3_STORE_F16; fp16-operand kernel only), stride in halves
    int nosplit = 0;                 // tuning hook TPU3_DEC_SPLIT=0: left-over steps whole, as before round 4
    int patches = 0;                 // lane-per-point kernel: a workgroup walks patches blockIdx.x, + gridDim.x, ...
    // (r5) the operand tables of the lane-per-point kernel as tpu3_dense_edge_conv_pack_f32 left them (null: the
    // workgroup builds them from the weights): DEC4_PACK_FLOATS floats + fold_n * 60
    const float *pack = nullptr;
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

// The same single instructions for values that come straight out of a BUILTIN MFMA: inline asm is invisible to
// hipcc's hazard recognizer, so a v_max in an asm statement may be issued inside the wait states an MFMA result
// needs before a VALU instruction reads it (stale accumulator: the lane-per-point kernel produced wrong maxima in
// one of its instantiations).  v_med3_f32 with an infinite third operand IS max (and clamps at zero for ReLU), one
// instruction like v_max, and the compiler sees it.
__device__ __forceinline__ float dec_relu_c(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()); }
__device__ __forceinline__ float dec_max_c(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }

// (r5) 2 * max(v, 0) as ONE FULL-RATE instruction, v_add_f32 v, v, |v|: exact (v + v for v > 0, v - v = +0 otherwise).
// v_max / v_med3 are half-rate on gfx950, and on an MFMA result the compiler puts a canonicalising v_max in front
// of them -- the ReLU of a hidden layer cost two half-rate instructions per value.  The factor 2 is taken out of
// the weights that multiply the value (0.5 * W is exact, and fma(0.5 W, 2 h, acc) == fma(W, h, acc) bit for bit).
__device__ __forceinline__ float dec_relu2(float v) { return v + __builtin_fabsf(v); }

// running maximum as ONE v_max_f32 (fmaxf() canonicalises both operands first: three instructions)
__device__ __forceinline__ float dec_max(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float dec_max3(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ f16x4 to_h4(f32x4 v)
{
    f16x4 r;
    r[0] = (_Float16)v[0]; r[1] = (_Float16)v[1]; r[2] = (_Float16)v[2]; r[3] = (_Float16)v[3];
    return r;
}

// ---- operand sets of the two arithmetic flavours -----------------------------------------------------------
// fp32: v_mfma_f32_16x16x4_f32, exact fp32 fma chains.  Lane (m = e, g) keeps the A operands of every step:
//   centre terms (W0a-W0b), W1b, W2c and W0b over the 24 input channels: channel 6g+s at step s (6 steps);
//   W1a, W2a, W2b over the 12 hidden channels: channel 4g+r at step r (4 steps, g = 3 is padding).
struct DecW32 {
    float wt0[6], wt1[6], wt2[6], w0b[6], w1a[4], w2a[4], w2b[4];
    __device__ __forceinline__ void load(const DecArgs &a, int m, int g)
    {
        const bool live = m < DEC_G;
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
    }
    // x6 = channels 6g .. 6g+5 of the lane's point
    __device__ __forceinline__ void centre(const float *x6, f32x4 &c0, f32x4 &c1, f32x4 &c2) const
    {
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            c0 = mfma4(wt0[s], x6[s], c0);
            c1 = mfma4(wt1[s], x6[s], c1);
            c2 = mfma4(wt2[s], x6[s], c2);
        }
    }
    __device__ __forceinline__ f32x4 ztab(const float *x6) const
    {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 6; ++s)
            z = mfma4(w0b[s], x6[s], z);
        return z;
    }
    template <int U>
    __device__ __forceinline__ void layers(const f32x4 (&h0)[U], f32x4 (&h1)[U], f32x4 (&h2)[U], f32x4 c1, f32x4 c2) const
    {
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
                h1[u][r] = dec_relu_c(h1[u][r]);
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
    }
};

// fp16 inputs on the matrix cores, fp32 accumulate (config C5, "fp16 feature MLPs on MFMA"): the weights and
// the activations entering an MFMA are rounded to fp16, everything between them (bias, ReLU, max, the z table)
// stays fp32.  v_mfma_f32_16x16x32_f16 takes eight k values per lane: lane (point e, g) supplies channels
// 8g .. 8g+7 of the 24 inputs (g = 3: zeros) -- ONE instruction per 24-channel product -- and for the last
// layer [h1 (4g..4g+3) | h0 (4g..4g+3)], i.e. W2a h1 + W2b h0 in ONE instruction; W1a h0 is one
// v_mfma_f32_16x16x16_f16.  2 MFMAs per 16-edge tile instead of 12.
struct DecW16 {
    f16x8 wt0, wt1, wt2, w0b, w2ab;
    f16x4 w1a;
    __device__ __forceinline__ void load(const DecArgs &a, int m, int g)
    {
        const bool live = m < DEC_G;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 8 * g + j;
            const bool ok = live && c < DEC_C;
            const float wa = ok ? a.w0[m * 48 + c] : 0.f, wb = ok ? a.w0[m * 48 + 24 + c] : 0.f;
            wt0[j] = (_Float16)(wa - wb);
            w0b[j] = (_Float16)wb;
            wt1[j] = (_Float16)(ok ? a.w1[m * 36 + 12 + c] : 0.f);
            wt2[j] = (_Float16)(ok ? a.w2[m * 48 + 24 + c] : 0.f);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * g + r;
            const bool ok = live && c < DEC_G;
            w1a[r] = (_Float16)(ok ? a.w1[m * 36 + c] : 0.f);
            w2ab[r] = (_Float16)(ok ? a.w2[m * 48 + c] : 0.f);
            w2ab[4 + r] = (_Float16)(ok ? a.w2[m * 48 + 12 + c] : 0.f);
        }
    }
    // x8 = channels 8g .. 8g+7 of the lane's point (zeros for g = 3)
    __device__ __forceinline__ void centre(f16x8 x8, f32x4 &c0, f32x4 &c1, f32x4 &c2) const
    {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wt0, x8, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wt1, x8, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wt2, x8, c2, 0, 0, 0);
    }
    __device__ __forceinline__ f32x4 ztab(f16x8 x8) const
    {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(w0b, x8, z, 0, 0, 0);
    }
    template <int U>
    __device__ __forceinline__ void layers(const f32x4 (&h0)[U], f32x4 (&h1)[U], f32x4 (&h2)[U], f32x4 c1, f32x4 c2) const
    {
        f16x4 h0h[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            h0h[u] = to_h4(h0[u]);
            h1[u] = __builtin_amdgcn_mfma_f32_16x16x16f16(w1a, h0h[u], c1, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                h1[u][r] = dec_relu_c(h1[u][r]);
            const f16x4 h1h = to_h4(h1[u]);
            f16x8 b;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                b[r] = h1h[r];
                b[4 + r] = h0h[u][r];
            }
            h2[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2ab, b, c2, 0, 0, 0);
        }
    }
};

// The lane's fragment of its point's input row, straight from global memory (read once per point):
// fp32 flavour channels 6g .. 6g+5 (all four lane groups), fp16 flavour channels 8g .. 8g+7 (g < 3).
// Kept in registers across the neighbour loop: it is also the x_i pass-through part of the output row.
template <bool F16>
struct DecX {
    static constexpr int NV = F16 ? 8 : 6;
    float v[NV];
    __device__ __forceinline__ void load(const float *xrow, int g)
    {
        if constexpr (F16) {
            const int gg = g < 3 ? g : 0;
            const f32x4 a = *(const f32x4 *)(xrow + 8 * gg), b = *(const f32x4 *)(xrow + 8 * gg + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = a[j];
                v[4 + j] = b[j];
            }
        } else {
            const f32x2 *p = (const f32x2 *)(xrow + 6 * g);           // 24 g bytes: 8-byte aligned
            const f32x2 a = p[0], b = p[1], c = p[2];
            v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1]; v[4] = c[0]; v[5] = c[1];
        }
    }
    __device__ __forceinline__ f16x8 half8(int g) const        // F16 only: zeros for the padding lane group
    {
        f16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            r[j] = g < 3 ? (_Float16)v[j < NV ? j : 0] : (_Float16)0.f;
        return r;
    }
    // into the staged output row (floats [36, 60) of the point's 60)
    __device__ __forceinline__ void stage(float *row, int g) const
    {
        if constexpr (F16) {
            if (g < 3) {
                *(f32x4 *)(row + 36 + 8 * g) = (f32x4){v[0], v[1], v[2], v[3]};
                *(f32x4 *)(row + 36 + 8 * g + 4) = (f32x4){v[4], v[5], v[6], v[7]};
            }
        } else {
            f32x2 *p = (f32x2 *)(row + 36 + 6 * g);
            p[0] = (f32x2){v[0], v[1]}; p[1] = (f32x2){v[2], v[3]}; p[2] = (f32x2){v[4], v[5]};
        }
    }
};

// z_p = W0b x_p for a block of 16 points (columns = points) -> table rows (LDS or global)
template <bool F16, typename W>
__device__ __forceinline__ void dec_ztab_block(const W &w, const float *X, int n, int pb, int e, int g, float *zt)
{
    DecX<F16> xf;
    xf.load(X + (size_t)min(pb + e, n - 1) * DEC_C, g);
    f32x4 z;
    if constexpr (F16)
        z = w.ztab(xf.half8(g));
    else
        z = w.ztab(xf.v);
    if (g < 3 && pb + e < n)
        *(f32x4 *)(zt + (size_t)(pb + e) * DEC_ZS + 4 * g) = z;
}

// z table of patches too large for LDS: one wave per 16 points, whole GPU
template <bool F16>
__global__ __launch_bounds__(256) void dec_ztab_kernel(DecArgs a)
{
    const int lane = threadIdx.x & 63, e = lane & 15, g = lane >> 4;
    std::conditional_t<F16, DecW16, DecW32> w;
    w.load(a, e, g);
    const int pb = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (pb >= a.n)
        return;
    dec_ztab_block<F16>(w, a.x + (size_t)blockIdx.y * a.n * DEC_C, a.n, pb, e, g,
                        a.zg + (size_t)blockIdx.y * a.n * DEC_ZS);
}

// TILES = k / 16.  ZG: the z table lives in global memory (a.zg) and blockIdx.y selects a DEC_SLICE-point
// slice of the patch; otherwise the table is built in LDS and the workgroup owns the whole patch.
template <int TILES, bool F16, bool ZG>
__global__ __launch_bounds__(DEC_NW * 64) void dec_fused_kernel(DecArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *tb = lds;                                   // DEC_NW * 16 * DEC_TS
    float *zl = lds + DEC_NW * 16 * DEC_TS + 48;       // n * DEC_ZS (+ 4 floats of slack), !ZG only
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e = lane & 15, g = lane >> 4;
    const int n = a.n;
    const float *X = a.x + (size_t)blockIdx.x * n * DEC_C;
    float *O = a.out + (size_t)blockIdx.x * n * a.out_stride;

    std::conditional_t<F16, DecW16, DecW32> w;
    w.load(a, e, g);
    // bias of the 4 channels this lane's accumulator rows hold (rows 4g..4g+3): re-read from LDS per block
    // of 16 points rather than held in 12 registers, which would cost a wave of occupancy
    float *bl = lds + DEC_NW * 16 * DEC_TS;            // [3][16]: b0 | b1 | b2, rows 12..15 zero
    if (tid < 48)
        bl[tid] = (tid & 15) < DEC_G ? (tid < 16 ? a.b0 : tid < 32 ? a.b1 : a.b2)[tid & 15] : 0.f;

    const float *ztab;
    int p_lo = 0, p_hi = n;
    if constexpr (ZG) {
        ztab = a.zg + (size_t)blockIdx.x * n * DEC_ZS;
        p_lo = blockIdx.y * DEC_SLICE;
        p_hi = min(n, p_lo + DEC_SLICE);
        __syncthreads();
    } else {
        // ---- z_p = W0b x_p for every point of the patch, once, into LDS ------------------------------
        for (int pb = wave * 16; pb < n; pb += DEC_NW * 16)
            dec_ztab_block<F16>(w, X, n, pb, e, g, zl);
        __syncthreads();
        ztab = zl;
    }
    const int zoff = g < 3 ? 4 * g : 0;     // lanes of the padding channel group read finite values (x 0 weights)

    // ---- 16 points per wave step: columns = POINTS, one neighbour slot at a time ---------------------
    // Everything a lane holds belongs to its own point pb + e: the centre terms stay in the registers
    // the MFMA left them in, the max over the k neighbours is a running per-lane v_max (no cross-lane
    // reduction), and lane (e, g) gathers rows 4g..4g+3 of ITS point's neighbour.
    constexpr int K = 16 * TILES;
    constexpr int U = 2;                        // neighbour slots in flight (independent MFMA chains)
    int *IT = (int *)(tb + wave * 16 * DEC_TS);  // per-wave tile [slot][16 points]: byte offset of the neighbour's z row
    float *ST = (float *)IT;                     // ... re-used as the output staging area [16 points][60]
    const char *gbase = (const char *)(ztab + zoff);
    DecX<F16> xf, xnext;
    if (p_lo + wave * 16 < p_hi)
        xnext.load(X + (size_t)min(p_lo + wave * 16 + e, n - 1) * DEC_C, g);
    for (int pb = p_lo + wave * 16; pb < p_hi; pb += DEC_NW * 16) {
        // neighbour indices of the 16 points, clamped and scaled once, read coalesced (k contiguous per point)
        for (int t = lane; t < 16 * K; t += 64) {
            const int pt = t / K, sl = t - pt * K;
            const int i = min(pb + pt, n - 1);
            const size_t io = ((size_t)blockIdx.x * n + i) * a.idx_stride + a.idx_off + sl;
            int j = a.idx64 ? (int)((const long long *)a.idx)[io] : ((const int *)a.idx)[io];
            j = min(max(j, 0), n - 1);
            IT[sl * 16 + pt] = j * (int)(DEC_ZS * sizeof(float));
        }
        // centre terms of the lane's point (+ bias): the initial values of the three accumulators
        xf = xnext;
        f32x4 c0 = *(const f32x4 *)(bl + 4 * g), c1 = *(const f32x4 *)(bl + 16 + 4 * g),
              c2 = *(const f32x4 *)(bl + 32 + 4 * g);
        if constexpr (F16)
            w.centre(xf.half8(g), c0, c1, c2);
        else
            w.centre(xf.v, c0, c1, c2);
        // the next block's row fragment arrives while this block's neighbours are processed
        if (pb + DEC_NW * 16 < p_hi)
            xnext.load(X + (size_t)min(pb + DEC_NW * 16 + e, n - 1) * DEC_C, g);
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
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f32x4 z = *(const f32x4 *)(gbase + off[u]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    h0[u][r] = fmaxf(c0[r] + z[r], 0.f);
            }
            w.template layers<U>(h0, h1, h2, c1, c2);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m0[r] = fmaxf(m0[r], h0[u][r]);
                    m1[r] = fmaxf(m1[r], h1[u][r]);
                    m2[r] = fmaxf(m2[r], h2[u][r]);
                }
        }
        // ---- write-out: the 60-float row [max h2 | max h1 | max h0 | x_i] of every point is assembled in the
        // wave's tile and leaves as the 15 aligned 16-byte pieces of ONE contiguous 240-byte run per point
        // (three scattered 16-byte stores per lane + a separate x_i copy cost 4x the payload in fabric writes)
        __builtin_amdgcn_wave_barrier();
        {
            float *st = ST + e * 60;
            if (g < 3) {
                *(f32x4 *)(st + 4 * g) = m2;                  // [0,12)  max h2
                *(f32x4 *)(st + DEC_G + 4 * g) = m1;          // [12,24) max h1
                *(f32x4 *)(st + 2 * DEC_G + 4 * g) = m0;      // [24,36) max h0
            }
            xf.stage(st, g);                                  // [36,60) x_i
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll 2      // (fully unrolled the four pieces in flight cost a wave of occupancy: 129 VGPRs)
        for (int t0 = 0; t0 < 16 * 15; t0 += 64) {
            const int t = t0 + lane;
            const int pt = t / 15, q4 = t - pt * 15;
            if (t < 16 * 15 && pb + pt < n) {
                const f32x4 v = *(const f32x4 *)(ST + pt * 60 + 4 * q4);
                if (a.out_half) {       // feature buffer stored as fp16: 15 eight-byte pieces of one 120-byte run
                    typedef _Float16 dh4 __attribute__((ext_vector_type(4)));
                    const dh4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    *(dh4 *)((_Float16 *)a.out + ((size_t)blockIdx.x * n + pb + pt) * a.out_stride + 4 * q4) = h;
                } else {
                    *(f32x4 *)(O + (size_t)(pb + pt) * a.out_stride + 4 * q4) = v;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------
// fp32 flavour, second form (round 3): a LANE per point, v_mfma_f32_4x4x1_16b_f32.
//
// The 16x16x4 form above pays for padding twice: 12 output channels occupy 16 MFMA rows and the 12 hidden
// channels 16 k slots (4 lane groups x 4 steps, one group idle) -- 56 % of the executed FLOPs of an edge are
// useful.  v_mfma_f32_4x4x1 computes 16 independent 4x4 outer products per instruction at the same FLOP rate
// (512 FLOP in 8 cycles): block b = lanes 4b..4b+3, A_b[i] from lane 4b+i, B_b[j] from lane 4b+j, lane 4b+j
// receives column j of its block's 4x4 result.  With
//     B = channel k of the LANE'S OWN (point, slot) pair,     A = W[4 rg + (lane & 3)][k]  (the same in every block),
// one instruction adds the contribution of input channel k to output channels 4rg..4rg+3 of 64 pairs: 3 row
// groups x K instructions per layer, NO padding anywhere (layer 1: 36, layer 2: 72 instructions of 8 cycles per
// 64 edges = 13.5 cycles per edge against 24 of the 16x16x4 form), and everything an edge needs and produces is
// lane-local: its input is the lane's own registers, the running maximum over the k slots is per lane, no
// shuffles, no LDS hand-offs, no DPP.  108 distinct A operands would not fit the register file as one VGPR each, and
// need not: the instruction's A-BROADCAST control (cbsz = 4, abid = b) takes the four A values of block b and uses
// them for all 16 blocks, so ONE VGPR carries 16 different (row group, k) operands -- lane 4b + i holds
// W[4 rg_b + i][k_b] -- and the 108 operands of both edge layers live in 7 registers, selected per instruction by
// an immediate.  (A first version read them from LDS, 27 ds_read_b128 per slot: 0.71 ms per launch, slower than the
// 16x16x4 form, every MFMA group waiting on its read.)  The per-POINT products (centre terms, z table) are 6 % of
// the work and read their operands from a 6 KB LDS table.
// The sums run over k in ascending order: one fp32 fma chain per output, i.e. the plain dot-product order.
// W2c x_i + b2 does not depend on the slot and the last layer has no ReLU, so it is added to the maximum at
// the end (max_j (c + v_j) = c + max_j v_j; the rounding of the sum differs by one ulp from seeding the chain).
// ---------------------------------------------------------------------------------------------------------
constexpr int DEC4_MAXW = 8;                 // waves per workgroup (at most)
// -DDEC4_TRACE (tools/dec_trace.py builds its own copy of the library with it): wave w of workgroup g writes the
// shader clock at mark m to trace[(4 g + w) * 16 + m]; mark 15 = HW_ID.  Not compiled into the product library.
#ifdef DEC4_TRACE
__device__ long long *g_dec4_trace;
#define DEC4_MARK(m)                                                                                            \
    do {                                                                                                        \
        if (g_dec4_trace && (threadIdx.x & 63) == 0)                                                            \
            g_dec4_trace[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (m)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define DEC4_MARK(m) do { } while (0)
#endif
// LDS tables of A operands, float4 entries [rg][kq][lane & 3] = W[4 rg + i][4 kq .. 4 kq + 3]
constexpr int DEC4_T_W1A = 0;                // 3 x 3 x 4   W1[:, 0:12]            (h0 -> h1)
constexpr int DEC4_T_W2 = 36;                // 3 x 6 x 4   W2[:, 0:24]            ([h1 | h0] -> h2)
constexpr int DEC4_T_C0 = 108;               // 3 x 6 x 4   W0[:, 0:24] - W0[:, 24:48]   (x_i -> centre of layer 0)
constexpr int DEC4_T_C1 = 180;               // 3 x 6 x 4   W1[:, 12:36]           (x_i -> centre of layer 1)
constexpr int DEC4_T_C2 = 252;               // 3 x 6 x 4   W2[:, 24:48]           (x_i -> centre of layer 2)
constexpr int DEC4_T_Z = 324;                // 3 x 6 x 4   W0[:, 24:48]           (x_j -> z table)
constexpr int DEC4_T_END = 396;              // float4 entries; then the biases [3][12] (+ 12 pad), then the z table
constexpr int DEC4_BIAS = DEC4_T_END * 4;    // float offset
constexpr int DEC4_ZTAB = DEC4_BIAS + 48;    // float offset
// packed operands (tpu3_dense_edge_conv_pack_f32): [0, DEC4_ZTAB) the tables and biases as they sit in LDS,
// [DEC4_PACK_WP, + 7 * 64) the packed (halved) edge-layer operands register by register, then the fold table
constexpr int DEC4_PACK_WP = DEC4_ZTAB;
constexpr int DEC4_PACK_FOLD = DEC4_PACK_WP + 7 * 64;
constexpr int DEC4_PACK_FLOATS = DEC4_PACK_FOLD;

__device__ __forceinline__ f32x4 mfma411(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

template <int I, int N, typename F>
__device__ __forceinline__ void dec4_static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        dec4_static_for<I + 1, N>(f);
    }
}

// operand Q of the packed edge-layer weights: register Q / 16, block Q % 16 broadcast to all blocks
template <int Q>
__device__ __forceinline__ f32x4 mfma411_bc(const float (&wp)[7], float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_4x4x1f32(wp[Q / 16], b, c, 4, Q % 16, 0);
}

// acc[rg] += W[4 rg .. 4 rg + 3][4 kq .. 4 kq + 3] in[0..3]  for the three row groups (tab = table + kq block)
template <int KQ>
__device__ __forceinline__ void dec4_step(const f32x4 *tab, int li, int kq, float i0, float i1, float i2, float i3,
                                          f32x4 (&acc)[3])
{
    const f32x4 a0 = tab[(0 * KQ + kq) * 4 + li], a1 = tab[(1 * KQ + kq) * 4 + li], a2 = tab[(2 * KQ + kq) * 4 + li];
    acc[0] = mfma411(a0[0], i0, acc[0]); acc[1] = mfma411(a1[0], i0, acc[1]); acc[2] = mfma411(a2[0], i0, acc[2]);
    acc[0] = mfma411(a0[1], i1, acc[0]); acc[1] = mfma411(a1[1], i1, acc[1]); acc[2] = mfma411(a2[1], i1, acc[2]);
    acc[0] = mfma411(a0[2], i2, acc[0]); acc[1] = mfma411(a1[2], i2, acc[1]); acc[2] = mfma411(a2[2], i2, acc[2]);
    acc[0] = mfma411(a0[3], i3, acc[0]); acc[1] = mfma411(a1[3], i3, acc[1]); acc[2] = mfma411(a2[3], i3, acc[2]);
}

// (r5) acc[rg] += W[4 rg .. 4 rg + 3][0 .. 4 KQ) x for NRG row groups, the A operands of k-quad kq + 1 requested from the
// LDS table BEFORE the MFMAs of k-quad kq are issued (two register sets).  Written as "read the three operands, issue
// their twelve MFMAs" per k-quad the compiler kept one register set and every group of MFMAs waited a full LDS round
// trip for its operands: the folded prep convolutions ran at a fifth of the matrix rate (90 reads x ~100 cycles
// against 360 x 8 per chunk of 24 outputs).
// k-quad visited at position t.  The fold (KQ = 15: [max h2 | max h1 | max h0 | x_i]) keeps the order its sums have had
// since round 3 -- group by group across the three maxima, then x -- because the 16x end-to-end fixtures were recorded
// against those bits: another (equally valid) order rounds a prep output differently, that flips a near-tie of the next
// block's kNN graph in some patch, and from the next level's FPS seeds on every point differs.
template <int KQ>
__device__ constexpr int dec4_kq_at(int t)
{
    return KQ == 15 ? (t < 9 ? 3 * (t % 3) + t / 3 : t) : t;
}

template <int KQ, int NRG>
__device__ __forceinline__ void dec4_mm(const f32x4 *tab, int li, const f32x4 (&x)[KQ], f32x4 (&acc)[NRG])
{
    f32x4 a[2][NRG];
#pragma unroll
    for (int rg = 0; rg < NRG; ++rg)
        a[0][rg] = tab[(rg * KQ + dec4_kq_at<KQ>(0)) * 4 + li];
    dec4_static_for<0, KQ>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        constexpr int kq = dec4_kq_at<KQ>(t);
        if constexpr (t + 1 < KQ) {
#pragma unroll
            for (int rg = 0; rg < NRG; ++rg)
                a[(t + 1) & 1][rg] = tab[(rg * KQ + dec4_kq_at<KQ>(t + 1)) * 4 + li];
        }
        // (the k-quad's first B operand passes through the asm: its products -- and, through the accumulators, the
        // rest -- cannot be issued above the requests; without that the scheduler hoists them and folds the two
        // register sets into one)
        float b0 = x[kq][0];
        asm volatile("" : "+v"(b0) : : "memory");
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int rg = 0; rg < NRG; ++rg)
                acc[rg] = mfma411(a[t & 1][rg][r], r == 0 ? b0 : x[kq][r], acc[rg]);
    });
}

// 12 outputs of a 24-channel input row held in six float4 registers
__device__ __forceinline__ void dec4_mm24(const f32x4 *tab, int li, const f32x4 (&x)[6], f32x4 (&acc)[3])
{
    dec4_mm<6, 3>(tab, li, x, acc);
}

// running maximum in LDS (ds_max_f32; no return value)
__device__ __forceinline__ void dec4_lds_max(float *p, float v)
{
    __builtin_amdgcn_ds_fmaxf((__attribute__((address_space(3))) float *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP,
                             false);
}

// The workgroup's operand tables, built in LDS from the raw weights: the 1620 weight and bias floats are first
// copied into LDS with coalesced loads (ONE global round trip; `raw` may alias the z table, which is written later),
// then re-arranged LDS -> LDS.  (Gathering the table entries straight from the weight matrices cost 20 us of
// dependent global loads per workgroup; a per-launch blob built by a one-block kernel needs a stream-ordered
// allocation per call, which serialised the eight network streams of the bench: 341 vs 311 ms per step.)
constexpr int DEC4_RAW_W0 = 0, DEC4_RAW_W1 = 576, DEC4_RAW_W2 = 1008, DEC4_RAW_B = 1584, DEC4_RAW_FLOATS = 1620;

__device__ __forceinline__ void dec4_setup(const DecArgs &a, float *lds, float *raw, float (&wp)[7])
{
    const int tid = threadIdx.x, lane = tid & 63;
    // (r5) eight loads per thread in flight: the address is selected, then ONE unconditional load (as a loop of
    // `raw[t] = t < .. ? w0[t] : ..` every element was its own branch + load + wait: seven to nine serial global round
    // trips, most of the 50 k cycles a workgroup spent before its first MFMA)
    DEC4_MARK(0);
    const float *pw0 = a.w0, *pw1 = a.w1, *pw2 = a.w2, *pb0 = a.b0, *pb1 = a.b1, *pb2 = a.b2;
    // (the six pointers in scalar registers first: selecting among kernel-argument FIELDS per lane made the compiler
    // fetch the pointer itself with a vector load inside a branch)
    asm volatile("" : "+s"(pw0), "+s"(pw1), "+s"(pw2), "+s"(pb0), "+s"(pb1), "+s"(pb2));
    for (int t0 = tid; t0 < DEC4_RAW_FLOATS; t0 += 8 * blockDim.x) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = min(t0 + i * (int)blockDim.x, DEC4_RAW_FLOATS - 1);
            const int tb = t - DEC4_RAW_B;
            const float *bsrc = tb < 12 ? pb0 : tb < 24 ? pb1 : pb2;
            const float *src = t < DEC4_RAW_W1 ? pw0 + t : t < DEC4_RAW_W2 ? pw1 + (t - DEC4_RAW_W1)
                             : t < DEC4_RAW_B ? pw2 + (t - DEC4_RAW_W2) : bsrc + (tb < 12 ? tb : tb < 24 ? tb - 12 : tb - 24);
            v[i] = *src;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = t0 + i * (int)blockDim.x;
            if (t < DEC4_RAW_FLOATS)
                raw[t] = v[i];
        }
    }
    __syncthreads();
    DEC4_MARK(1);
    const float *w0 = raw + DEC4_RAW_W0, *w1 = raw + DEC4_RAW_W1, *w2 = raw + DEC4_RAW_W2;
    // table entry e = (rg, kq, i): row 4 rg + i, columns 4 kq .. 4 kq + 3 of the table's matrix slice
    for (int e = tid; e < DEC4_T_END; e += blockDim.x) {
        const int base = e < DEC4_T_W2 ? DEC4_T_W1A : DEC4_T_W2 + (e - DEC4_T_W2) / 72 * 72;
        const int KQ = e < DEC4_T_W2 ? 3 : 6;
        const int r = e - base, i = r & 3, kq = (r >> 2) % KQ, rg = (r >> 2) / KQ;
        const int row = 4 * rg + i, c = 4 * kq;
        // source matrix, row length, first column, and (layer-0 centre term only) the column block to subtract
        const float *src;
        int ld, col, sub = -1;
        if (base == DEC4_T_W1A) { src = w1; ld = 36; col = 0; }
        else if (base == DEC4_T_W2) { src = w2; ld = 48; col = 0; }
        else if (base == DEC4_T_C0) { src = w0; ld = 48; col = 0; sub = 24; }
        else if (base == DEC4_T_C1) { src = w1; ld = 36; col = 12; }
        else if (base == DEC4_T_C2) { src = w2; ld = 48; col = 24; }
        else { src = w0; ld = 48; col = 24; }
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float w = src[row * ld + col + c + q];
            v[q] = sub >= 0 ? w - src[row * ld + sub + c + q] : w;
        }
        ((f32x4 *)lds)[e] = v;
    }
    if (tid < 48)
        lds[DEC4_BIAS + tid] = tid < 36 ? raw[DEC4_RAW_B + tid] : 0.f;
    // packed A operands of the two edge layers, HALVED (their inputs h0, h1 are kept as 2 * relu): operand q < 36:
    // (rg, k) = (q / 12, q % 12) of W1[:, 0:12];
    // 36 <= q < 108: (rg, k) = ((q - 36) / 24, (q - 36) % 24) of W2[:, 0:24]; lane 4b + i of register v holds
    // operand q = 16 v + b for row 4 rg + i
#pragma unroll
    for (int v = 0; v < 7; ++v) {
        const int q = 16 * v + (lane >> 2), i = lane & 3;
        float w = 0.f;
        if (q < 36)
            w = w1[(4 * (q / 12) + i) * 36 + q % 12];
        else if (q < 108)
            w = w2[(4 * ((q - 36) / 24) + i) * 48 + (q - 36) % 24];
        wp[v] = 0.5f * w;       // the edge layers see 2 * relu (dec_relu2)
    }
    __syncthreads();            // `raw` may be overwritten from here on
    DEC4_MARK(2);
}

// (FOLD) A operands of the folded prep convolutions: float4 entries
// [chunk of 24 outputs][row group 6][kq 15][i 4] = fold_w[24 chunk + 4 rg + i][4 kq .. 4 kq + 3]
__device__ __forceinline__ void dec4_fold_table(const DecArgs &a, f32x4 *ftab)
{
    const int tid = threadIdx.x;
    // (r5: the loads of a thread in flight together, as in dec4_setup)
    const int fe = a.fold_n * 15;
    for (int e0 = tid; e0 < fe; e0 += 5 * blockDim.x) {
        f32x4 v[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int e = min(e0 + q * (int)blockDim.x, fe - 1);
            const int i = e & 3, kq = (e >> 2) % 15, rgc = (e >> 2) / 15;      // rgc = 6 chunk + rg
            v[q] = *(const f32x4 *)(a.fold_w + (size_t)(4 * rgc + i) * 60 + 4 * kq);
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int e = e0 + q * (int)blockDim.x;
            if (e < fe)
                ftab[e] = v[q];
        }
    }
    // (visible after the barrier behind phase A)
}

// The tables of one set of weights written out once: what dec4_setup / dec4_fold_table leave in LDS and in the wp
// registers, in the layout DEC4_PACK_* (one workgroup of 256 threads; the launches that pass the blob copy it).
__global__ __launch_bounds__(256) void dec4_pack_kernel(DecArgs a, float *blob)
{
    __shared__ __attribute__((aligned(16))) float lds[DEC4_ZTAB + DEC4_RAW_FLOATS + 4];
    float wp[7];
    dec4_setup(a, lds, lds + DEC4_ZTAB, wp);
    const int tid = threadIdx.x;
    for (int e = tid; e < DEC4_ZTAB; e += blockDim.x)
        blob[e] = lds[e];
    if (tid < 64)
#pragma unroll
        for (int v = 0; v < 7; ++v)
            blob[DEC4_PACK_WP + 64 * v + tid] = wp[v];
    if (a.fold_n)
        dec4_fold_table(a, (f32x4 *)(blob + DEC4_PACK_FOLD));
}

// U = neighbour slots per loop iteration.  U = 2 folds two slots into one v_max3 per channel (18 instead of 36
// running-maximum instructions per slot) at the price of 36 more live registers (3 instead of 4 waves per SIMD).
template <bool IDX64, int U, bool FOLD>
__global__ __launch_bounds__(DEC4_MAXW * 64) __attribute__((amdgpu_waves_per_eu(U == 1 ? 4 : 3, U == 1 ? 4 : 3)))
void dec_fused4_kernel(DecArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
    const int li = lane & 3;
    const int n = a.n, k = a.k;
    f32x4 *tab = (f32x4 *)lds;
    float *bias = lds + DEC4_BIAS;
    float *zl = lds + DEC4_ZTAB;                 // z_p  = W0b x_p            (n x 12)
    // (FOLD) A operands of the folded prep convolutions behind the table: float4 entries
    // [chunk of 24 outputs][row group 6][kq 15][i 4] = fold_w[24 chunk + 4 rg + i][4 kq .. 4 kq + 3]
    f32x4 *ftab = (f32x4 *)(zl + (size_t)n * DEC_ZS + 4);

    float wp[7];
    if (a.pack) {
        // (r5) the tables as a per-weights blob: a straight copy, every load of a thread in flight, one barrier.
        // Built in place (dec4_setup) the set-up is ~1500 integer / LDS instructions per wave, each of which waits its
        // turn behind the MFMAs of the two other workgroups of the compute unit: 35 - 45 k of a wave's 180 k cycles.
        DEC4_MARK(0);
        const f32x4 *src = (const f32x4 *)a.pack;
        constexpr int TE = DEC4_ZTAB / 4;                     // 408 float4: tables + biases
        {
            f32x4 v[2];
            for (int e0 = tid; e0 < TE; e0 += 2 * blockDim.x) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    v[q] = src[min(e0 + q * (int)blockDim.x, TE - 1)];
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (e0 + q * (int)blockDim.x < TE)
                        tab[e0 + q * (int)blockDim.x] = v[q];
            }
        }
#pragma unroll
        for (int v = 0; v < 7; ++v)
            wp[v] = a.pack[DEC4_PACK_WP + 64 * v + lane];
        __syncthreads();
        DEC4_MARK(1);
        if constexpr (FOLD) {
            const f32x4 *fsrc = (const f32x4 *)(a.pack + DEC4_PACK_FOLD);
            const int fe = a.fold_n * 15;
            for (int e0 = tid; e0 < fe; e0 += 5 * blockDim.x) {
                f32x4 v[5];
#pragma unroll
                for (int q = 0; q < 5; ++q)
                    v[q] = fsrc[min(e0 + q * (int)blockDim.x, fe - 1)];
#pragma unroll
                for (int q = 0; q < 5; ++q)
                    if (e0 + q * (int)blockDim.x < fe)
                        ftab[e0 + q * (int)blockDim.x] = v[q];
            }
        }
        DEC4_MARK(2);
        // (visible after the barrier behind phase A)
    } else {
        dec4_setup(a, lds, zl, wp);
        if constexpr (FOLD)
            dec4_fold_table(a, ftab);
    }
    const int nstep = (n + 63) >> 6;
    // (r4) steps left over by the round-robin deal (nstep mod 4 = 1 or 2) are split by neighbour slots over 4 or 2
    // waves, see the end of phase B; their running maxima meet in `comb` [left-over step][36 channels][64 lanes]
    const int rem = nstep % nwave;
    const int split = (nwave == 4 && !a.nosplit && (k % (4 * U)) == 0) ? (rem == 1 ? 4 : rem == 2 ? 2 : 1) : 1;
    float *comb = (float *)ftab + (size_t)a.fold_n * 60;

    // per-patch state (the workgroup walks patches blockIdx.x, blockIdx.x + gridDim.x, ...)
    const float *X = nullptr;
    float *O = nullptr;
    size_t prow = 0;
    int vw = 0;
    f32x4 x[6];
    const char *zb = (const char *)zl;
    const int zmax = (n - 1) * (int)(DEC_ZS * sizeof(float));
    const float ninf = -__builtin_inff();

    // centre terms of a step: c0 = W0c x_i + b0, c1 = W1c x_i + b1 (x = the step's rows)
    auto centre = [&](const f32x4 *tabs, int st, int pc, f32x4 (&c0)[3], f32x4 (&c1)[3]) __attribute__((always_inline)) {
        {
            const f32x4 *xr = (const f32x4 *)(X + (size_t)pc * DEC_C);
#pragma unroll
            for (int q = 0; q < 6; ++q)
                x[q] = xr[q];
        }
#pragma unroll
        for (int rg = 0; rg < 3; ++rg) {
            c0[rg] = *(const f32x4 *)(bias + 4 * rg);
            c1[rg] = *(const f32x4 *)(bias + 12 + 4 * rg);
        }
        dec4_mm24(tabs + DEC4_T_C0, li, x, c0);
        dec4_mm24(tabs + DEC4_T_C1, li, x, c1);
    };

    // neighbour slots [s0, s1) of the lane's point folded into the running maxima; index and z row of the following
    // slots are requested before an iteration's MFMAs.  h0 and h1 are kept as TWICE their ReLU (dec_relu2; the packed
    // weights are halved), so m0 and m1 come out doubled: `finish` halves them.
    auto slots = [&](int pc, int s0, int s1, const f32x4 (&c0)[3], const f32x4 (&c1)[3], f32x4 (&m0)[3], f32x4 (&m1)[3],
                     f32x4 (&m2)[3]) __attribute__((always_inline)) {
        const size_t ibase = (prow + pc) * a.idx_stride + a.idx_off;
        auto ldidx = [&](int sl) __attribute__((always_inline)) {  // the raw neighbour index of slot sl
            return IDX64 ? (int)((const long long *)a.idx)[ibase + sl] : ((const int *)a.idx)[ibase + sl];
        };
        auto zoff = [&](int j) __attribute__((always_inline)) {    // byte offset of the neighbour's z row
            return min(max(j * (int)(DEC_ZS * sizeof(float)), 0), zmax);
        };
#pragma unroll
        for (int rg = 0; rg < 3; ++rg)
            m0[rg] = m1[rg] = m2[rg] = (f32x4){ninf, ninf, ninf, ninf};
        // (r5) the pipeline of an iteration: [h0 from the z rows requested at the end of the previous one] [layer 1] [layer 2]
        // [index loaded an iteration ago -> z rows of the next iteration requested, the index after that loaded] [maxima].
        // (Written as "use z, request the next z" at the top of the loop, the compiler merged the request into the use --
        // a phi of two loads is a load of the phi -- and every iteration began with an LDS round trip and ended with a
        // global one, both waited for on the spot: a lone wave took 25 % longer than its instructions.  The empty asm
        // with a memory clobber behind the requests keeps them where they are; requested before the MFMAs, as the source
        // once read, the 24 registers of the rows cost a wave of occupancy.)
        int jr[U];
        f32x4 zn[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x4 *zr = (const f32x4 *)(zb + zoff(ldidx(min(s0 + u, s1 - 1))));
            zn[u][0] = zr[0]; zn[u][1] = zr[1]; zn[u][2] = zr[2];
            jr[u] = ldidx(min(s0 + U + u, s1 - 1));
        }
#pragma unroll 1
        for (int sl = s0; sl < s1; sl += U) {
            f32x4 h0[U][3], h1[U][3], h2[U][3];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int rg = 0; rg < 3; ++rg) {
                    const f32x4 pre = c0[rg] + zn[u][rg];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        h0[u][rg][r] = dec_relu2(pre[r]);
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int rg = 0; rg < 3; ++rg)
                    h1[u][rg] = c1[rg];
            // layer 1: input channel kk of h0; the row groups (and slots) are independent accumulator chains
            dec4_static_for<0, 12>([&](auto kc) __attribute__((always_inline)) {
                constexpr int kk = decltype(kc)::value;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float b = h0[u][kk / 4][kk % 4];
                    h1[u][0] = mfma411_bc<0 * 12 + kk>(wp, b, h1[u][0]);
                    h1[u][1] = mfma411_bc<1 * 12 + kk>(wp, b, h1[u][1]);
                    h1[u][2] = mfma411_bc<2 * 12 + kk>(wp, b, h1[u][2]);
                }
            });
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int rg = 0; rg < 3; ++rg) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        h1[u][rg][r] = dec_relu2(h1[u][rg][r]);
                    h2[u][rg] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            // layer 2: [h1 | h0]
            dec4_static_for<0, 24>([&](auto kc) __attribute__((always_inline)) {
                constexpr int kk = decltype(kc)::value;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float b = kk < 12 ? h1[u][(kk % 12) / 4][kk % 4] : h0[u][(kk % 12) / 4][kk % 4];
                    h2[u][0] = mfma411_bc<36 + 0 * 24 + kk>(wp, b, h2[u][0]);
                    h2[u][1] = mfma411_bc<36 + 1 * 24 + kk>(wp, b, h2[u][1]);
                    h2[u][2] = mfma411_bc<36 + 2 * 24 + kk>(wp, b, h2[u][2]);
                }
            });
            // requests for the following iterations (clamped at the end: a repeated slot is harmless)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f32x4 *zr = (const f32x4 *)(zb + zoff(jr[u]));
                zn[u][0] = zr[0]; zn[u][1] = zr[1]; zn[u][2] = zr[2];
                jr[u] = ldidx(min(sl + 2 * U + u, s1 - 1));
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int rg = 0; rg < 3; ++rg)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (U == 2) {
                        // (h2 comes straight out of the MFMAs: its first read is a compiler-visible instruction)
                        m0[rg][r] = dec_max3(m0[rg][r], h0[0][rg][r], h0[1][rg][r]);
                        m1[rg][r] = dec_max3(m1[rg][r], h1[0][rg][r], h1[1][rg][r]);
                        m2[rg][r] = dec_max_c(m2[rg][r], dec_max_c(h2[0][rg][r], h2[1][rg][r]));
                    } else {
                        m0[rg][r] = dec_max(m0[rg][r], h0[0][rg][r]);
                        m1[rg][r] = dec_max(m1[rg][r], h1[0][rg][r]);
                        m2[rg][r] = dec_max_c(m2[rg][r], h2[0][rg][r]);
                    }
                }
        }
    };

    // what follows a point's maxima: c2_p = W2c x_p + b2 (the point's row again, six 16-byte loads) added to the last
    // layer's, then (rows) the write-out [max h2 + c2 | max h1 | max h0] = floats [0, 36) of the lane's own row, nine
    // back-to-back 16-byte stores (staging 16 rows at a time through LDS so that 15 consecutive lanes write one row was
    // measured: no faster, and the tile costs the LDS of another workgroup per compute unit), then (FOLD) the chunks
    // ch0, ch0 + chs, ... of the next prep convolutions
    auto finish = [&](const f32x4 *tabs, int tofs, int p, int pc, f32x4 (&m0)[3], f32x4 (&m1)[3], f32x4 (&m2)[3], bool rows,
                      int ch0, int chs) __attribute__((always_inline)) {
        {
            const f32x4 *xr2 = (const f32x4 *)(X + (size_t)pc * DEC_C);
#pragma unroll
            for (int q = 0; q < 6; ++q)
                x[q] = xr2[q];
            f32x4 c2[3];
#pragma unroll
            for (int rg = 0; rg < 3; ++rg)
                c2[rg] = *(const f32x4 *)(bias + 24 + 4 * rg);
            dec4_mm24(tabs + DEC4_T_C2, li, x, c2);
#pragma unroll
            for (int rg = 0; rg < 3; ++rg) {
                m2[rg] = m2[rg] + c2[rg];
                m1[rg] = m1[rg] * 0.5f;         // maxima of 2 * relu (exact)
                m0[rg] = m0[rg] * 0.5f;
            }
        }
        if (rows && p < n) {
            f32x4 *orow = (f32x4 *)(O + (size_t)p * a.out_stride);
#pragma unroll
            for (int rg = 0; rg < 3; ++rg)
                orow[rg] = m2[rg];
#pragma unroll
            for (int rg = 0; rg < 3; ++rg)
                orow[3 + rg] = m1[rg];
#pragma unroll
            for (int rg = 0; rg < 3; ++rg)
                orow[6 + rg] = m0[rg];
        }
        if constexpr (FOLD) {
            // ---- the next prep convolutions, folded: they are linear in the concatenated feature row, so this block's
            // 60 channels contribute W[:, their columns] . row to each of them NOW, while the row is in registers --
            // the level's feature buffer is then never re-read by a prep convolution (84 / 144 / 204 channels per
            // point and layer before).  Outputs in chunks of 24 (six accumulators); the first chunk completes the
            // NEXT block's input: ReLU, contiguous rows; the others are partial sums for the blocks after it.
            float *arow = a.acc + (prow + pc) * a.acc_stride;
#pragma unroll 1
            for (int ch = ch0; ch < a.fold_n / 24; ch += chs) {
                asm volatile("" : "+s"(tofs));
                const f32x4 *ft = ftab + tofs + ch * (6 * 15 * 4);
                f32x4 acc[6];
#pragma unroll
                for (int rg = 0; rg < 6; ++rg)
                    acc[rg] = a.fold_b ? *(const f32x4 *)(a.fold_b + 24 * ch + 4 * rg)
                                       : *(const f32x4 *)(arow + a.seed_off + 24 * ch + 4 * rg);
                // the row [max h2 | max h1 | max h0 | x_i] as 15 k-quads, in the order of the weight columns
                const f32x4 row[15] = {m2[0], m2[1], m2[2], m1[0], m1[1], m1[2], m0[0], m0[1], m0[2],
                                       x[0], x[1], x[2], x[3], x[4], x[5]};
                dec4_mm<15, 6>(ft, li, row, acc);
                if (p < n) {
                    if (ch == 0) {
                        f32x4 *xn = (f32x4 *)(a.xnext + (prow + p) * DEC_C);
#pragma unroll
                        for (int rg = 0; rg < 6; ++rg) {
                            f32x4 v = acc[rg];
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                v[r] = dec_relu_c(v[r]);
                            xn[rg] = v;
                        }
                    } else {
#pragma unroll
                        for (int rg = 0; rg < 6; ++rg)
                            *(f32x4 *)(arow + a.store_off + 24 * (ch - 1) + 4 * rg) = acc[rg];
                    }
                }
            }
        }
    };

    for (int patch_v = blockIdx.x; patch_v < a.patches; patch_v += gridDim.x) {
        const int patch = __builtin_amdgcn_readfirstlane(patch_v);
        X = a.x + (size_t)patch * n * DEC_C;
        O = a.out + (size_t)patch * n * a.out_stride;
        prow = (size_t)patch * n;
        if (split > 1)
            for (int e = tid; e < rem * 36 * 16; e += blockDim.x)
                ((f32x4 *)comb)[e] = (f32x4){ninf, ninf, ninf, ninf};
        // Which wave takes an extra step (no split: TPU3_DEC_SPLIT=0, or three left-over steps) rotates with the
        // patch: wave w of every workgroup sits on SIMD w -- with the long wave always on SIMD 0 that SIMD alone would
        // bound the compute unit.
        vw = __builtin_amdgcn_readfirstlane((wave + patch) % nwave);
        DEC4_MARK(3);
        // ---- phase A, per point (lane = point): the z table into LDS, the x_i part of the output row straight from
        // the registers.  (Keeping the first step's rows for phase B made all 24 registers live across the slot loop
        // once the code around it grew: phase B reads its rows again.)  (r4: the slot-independent part of the last layer, c2_p = W2c x_p + b2, used
        // to be a second n x 12 table here; it is re-derived by the point's own lane at the write-out instead: 72
        // MFMAs per step, 15 KB less per workgroup.)
        {
            int st = vw;
            while (st + nwave < nstep)
                st += nwave;
            for (; st >= 0; st -= nwave) {
                int tofs = 0;                   // (opaque zero: keeps the A-operand reads inside the loop, see phase B)
                asm volatile("" : "+s"(tofs));
                const f32x4 *tabs = tab + tofs;
                const int p = st * 64 + lane;
                const f32x4 *xr = (const f32x4 *)(X + (size_t)min(p, n - 1) * DEC_C);
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    x[q] = xr[q];
                f32x4 z[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                dec4_mm24(tabs + DEC4_T_Z, li, x, z);
                if (p < n) {
                    f32x4 *zr = (f32x4 *)(zl + p * DEC_ZS);
                    zr[0] = z[0]; zr[1] = z[1]; zr[2] = z[2];
                    f32x4 *orow = (f32x4 *)(O + (size_t)p * a.out_stride);       // [36, 60): one 96-byte run
#pragma unroll
                    for (int q = 0; q < 6; ++q)
                        orow[9 + q] = x[q];
                }
            }
        }
        DEC4_MARK(4);
        __syncthreads();
        DEC4_MARK(5);

        // whole steps, dealt round-robin
        const int nwhole = split > 1 ? nstep - rem : nstep;
        for (int st = vw; st < nwhole; st += nwave) {
            // (opaque zero: keeps the table reads of this step inside the loop -- hoisted, the loop-invariant A operands of
            // the centre terms would occupy 144 registers)
            int tofs = 0;
            asm volatile("" : "+s"(tofs));
            const f32x4 *tabs = tab + tofs;
            const int p = st * 64 + lane;
            const int pc = min(p, n - 1);
            f32x4 c0[3], c1[3], m0[3], m1[3], m2[3];
            centre(tabs, st, pc, c0, c1);
            DEC4_MARK(6);
            slots(pc, 0, k, c0, c1, m0, m1, m2);
            DEC4_MARK(7);
            finish(tabs, tofs, p, pc, m0, m1, m2, true, 0, 1);
            DEC4_MARK(8);
        }
        if (split > 1) {
            // ---- (r4) the steps left over (a 312-point patch: the fifth of five on four waves), split by SLOTS: wave
            // (j, q) folds slots [q k / split, (q + 1) k / split) of left-over step j, the waves' maxima meet in LDS
            // (ds_max_f32 on a table that phase A set to -inf; max is order-independent, so the result is the unsplit
            // step's bit for bit), and what follows the maxima is dealt out again: the rows to part split - 1, fold chunk ch
            // to part ch mod split.  (One wave taking the whole step held the workgroup's place on the compute unit
            // for two steps while three SIMDs waited -- the kernel took launches / 3 x the LONG wave's lifetime, measured
            // with s_memtime marks: 0.575 -> 0.535 ms per 3840-patch launch, medians of 40; TPU3_DEC_SPLIT=0 restores it.)
            int tofs = 0;
            asm volatile("" : "+s"(tofs));
            const f32x4 *tabs = tab + tofs;
            const int j = vw / split, q = vw - j * split;
            const int st = nwhole + j;
            const int p = st * 64 + lane;
            const int pc = min(p, n - 1);
            float *cb = comb + j * (36 * 64) + lane;
            const int ks = k / split;
            {
                f32x4 c0[3], c1[3], m0[3], m1[3], m2[3];
                centre(tabs, st, pc, c0, c1);
                slots(pc, q * ks, (q + 1) * ks, c0, c1, m0, m1, m2);
#pragma unroll
                for (int rg = 0; rg < 3; ++rg)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dec4_lds_max(cb + (0 + 4 * rg + r) * 64, m2[rg][r]);
                        dec4_lds_max(cb + (12 + 4 * rg + r) * 64, m1[rg][r]);
                        dec4_lds_max(cb + (24 + 4 * rg + r) * 64, m0[rg][r]);
                    }
            }
            DEC4_MARK(9);
            __syncthreads();
            DEC4_MARK(10);
            const bool rows = q == split - 1;
            if (rows || (FOLD && q < a.fold_n / 24)) {
                f32x4 m0[3], m1[3], m2[3];
#pragma unroll
                for (int rg = 0; rg < 3; ++rg)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        m2[rg][r] = cb[(0 + 4 * rg + r) * 64];
                        m1[rg][r] = cb[(12 + 4 * rg + r) * 64];
                        m0[rg][r] = cb[(24 + 4 * rg + r) * 64];
                    }
                finish(tabs, tofs, p, pc, m0, m1, m2, rows, q, split);
            }
        }
        DEC4_MARK(11);
#ifdef DEC4_TRACE
        if (g_dec4_trace && lane == 0) {
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            g_dec4_trace[((size_t)blockIdx.x * 4 + wave) * 16 + 15] = hw;
            g_dec4_trace[((size_t)blockIdx.x * 4 + wave) * 16 + 14] = __builtin_amdgcn_s_memrealtime();
        }
#endif
        // (the z table and `comb` are rewritten for the next patch)
        if (patch + (int)gridDim.x < a.patches)
            __syncthreads();
    }
}

constexpr size_t dec4_lds_bytes(int n, int fold_n = 0)
{
    const size_t tables = (size_t)n * DEC_ZS + 4;              // the z table; the raw weights alias it during setup
    const int nstep = (n + 63) / 64, rem = nstep % 4;           // left-over steps split over the waves: their maxima
    const size_t comb = nstep > 4 && (rem == 1 || rem == 2) ? (size_t)rem * 36 * 64 : 0;
    return ((size_t)DEC4_ZTAB + (tables > (size_t)DEC4_RAW_FLOATS ? tables : (size_t)DEC4_RAW_FLOATS) +
            (size_t)fold_n * 60 + comb) * sizeof(float);
}

// left-over steps split over the waves (default) or dealt whole as before round 4: TPU3_DEC_SPLIT=0 / tpu3_debug_dec_split
int g_dec_nosplit = getenv("TPU3_DEC_SPLIT") ? atoi(getenv("TPU3_DEC_SPLIT")) == 0 : 0;

int dec4_launch(hipStream_t s, int patches, const DecArgs &a)
{
    // FOUR waves per workgroup, the whole 64-point steps dealt round-robin and the left-over ones split by neighbour
    // slots (see the kernel): with a wave per step a 312-point patch is 5 waves on 4 SIMDs -- two of them share a SIMD,
    // run at half speed and hold the workgroup's LDS and register slots while three SIMDs wait (0.70 vs 0.535 ms per
    // 3840-patch launch).  TPU3_DEC_NW: tuning hook.
    static const int nw_env = getenv("TPU3_DEC_NW") ? atoi(getenv("TPU3_DEC_NW")) : 4;
    const int nw = min(min(DEC4_MAXW, max(1, nw_env)), (a.n + 63) / 64);
    // TPU3_DEC_LDS_MIN (tuning hook): a floor under the LDS request, e.g. 56000 = two workgroups per compute unit instead
    // of three (leaves a third of the registers and wave slots to kernels of the other streams)
    static const size_t lds_min = getenv("TPU3_DEC_LDS_MIN") ? (size_t)atol(getenv("TPU3_DEC_LDS_MIN")) : 0;
    const size_t lds = max(dec4_lds_bytes(a.n, a.fold_n), min(lds_min, (size_t)160 * 1024));
    // two neighbour slots per loop iteration (U = 1: 4 waves per SIMD but 36 instead of 18 running-maximum instructions
    // per slot, 298.5 vs 294.9 ms per bench step in round 3, and it spills since the left-over steps are split)
    void (*kern)(DecArgs);
    if (a.fold_n)
        kern = a.idx64 ? dec_fused4_kernel<true, 2, true> : dec_fused4_kernel<false, 2, true>;
    else
        kern = a.idx64 ? dec_fused4_kernel<true, 2, false> : dec_fused4_kernel<false, 2, false>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess)
        return (int)e;
    const int nosplit = g_dec_nosplit;
    // TPU3_DEC_PERSIST = workgroups per compute unit that walk the patches (0: one workgroup per patch)
    static const int persist = getenv("TPU3_DEC_PERSIST") ? atoi(getenv("TPU3_DEC_PERSIST")) : 0;
    static const int ncu = []() { int d = 0, v = 256; hipGetDevice(&d); hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d); return v; }();
    DecArgs a2 = a;
    a2.nosplit = nosplit;
    a2.patches = patches;
    const int grid = persist > 0 ? min(patches, ncu * persist) : patches;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nw * 64), lds, s, a2);
    return tpu3_launch_status();
}

constexpr size_t DEC_LDS_TILE_BYTES = ((size_t)DEC_NW * 16 * DEC_TS + 48) * sizeof(float);    // tiles + bias table

template <bool F16>
int dec_launch(hipStream_t s, int patches, DecArgs &a)
{
    const size_t with_z = DEC_LDS_TILE_BYTES + ((size_t)a.n * DEC_ZS + 4) * sizeof(float);
    const bool zg = with_z > 160 * 1024;
    hipError_t e = hipSuccess;
    if (zg) {
        e = hipMallocAsync((void **)&a.zg, (size_t)patches * a.n * DEC_ZS * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(dec_ztab_kernel<F16>, dim3((a.n + 63) / 64, patches), dim3(256), 0, s, a);
    }
    const size_t lds = zg ? DEC_LDS_TILE_BYTES : with_z;
    const dim3 grid(patches, zg ? (a.n + DEC_SLICE - 1) / DEC_SLICE : 1);
#define DEC_LAUNCH1(T, Z)                                                                                   \
    e = hipFuncSetAttribute((const void *)dec_fused_kernel<T, F16, Z>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)lds);                                                                      \
    if (e == hipSuccess) hipLaunchKernelGGL((dec_fused_kernel<T, F16, Z>), grid, dim3(DEC_NW * 64), lds, s, a)
#define DEC_LAUNCH(T)                                                                                       \
    if (zg) { DEC_LAUNCH1(T, true); } else { DEC_LAUNCH1(T, false); }
    switch (a.k / 16) {
    case 1: DEC_LAUNCH(1); break;
    case 2: DEC_LAUNCH(2); break;
    case 3: DEC_LAUNCH(3); break;
    default: DEC_LAUNCH(4); break;
    }
#undef DEC_LAUNCH1
#undef DEC_LAUNCH
    int r = e != hipSuccess ? (int)e : tpu3_launch_status();
    if (zg) {
        const hipError_t fe = hipFreeAsync(a.zg, s);
        if (!r) r = (int)fe;
    }
    return r;
}

} // namespace

#ifdef DEC4_TRACE
extern "C" int tpu3_debug_dec_trace(long long *buf)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dec4_trace), &buf, sizeof(buf));
}
#endif

extern "C" int tpu3_debug_dec_split(int on)
{
    const int old = g_dec_nosplit ? 0 : 1;
    g_dec_nosplit = on ? 0 : 1;
    return old;
}

extern "C" int tpu3_dense_edge_conv_st_f32(tpu3_stream_t stream, int patches, int n, int k, const float *x,
                                           const void *idx, int idx_elem_size, int idx_stride, int idx_off,
                                           const float *w0, const float *b0, const float *w1, const float *b1,
                                           const float *w2, const float *b2, void *out, int out_stride, int mfma,
                                           int out_store)
{
    if (out_store == TPU3_STORE_F32)
        return tpu3_dense_edge_conv_f32(stream, patches, n, k, x, idx, idx_elem_size, idx_stride, idx_off, w0, b0, w1,
                                        b1, w2, b2, (float *)out, out_stride, mfma);
    // fp16 rows come out of the fp16-operand kernel only (its results carry fp16-operand error already)
    if (out_store != TPU3_STORE_F16 || mfma != TPU3_MFMA_F16) return TPU3_EINVAL;
    if (patches < 0 || n <= 0 || k <= 0 || (k % 16) != 0 || k > 64) return TPU3_EINVAL;
    if (idx_elem_size != 4 && idx_elem_size != 8) return TPU3_EINVAL;
    if (idx_off < 0 || idx_stride < idx_off + k || out_stride < 60 || (out_stride % 4) != 0) return TPU3_EINVAL;
    if (patches == 0) return TPU3_OK;
    if (!x || !idx || !w0 || !b0 || !w1 || !b1 || !w2 || !b2 || !out) return TPU3_EINVAL;
    if (((uintptr_t)out % 8) != 0 || ((uintptr_t)x % 16) != 0) return TPU3_EINVAL;
    if (patches > 2147483647 / (n > 0 ? n : 1)) return TPU3_ELIMIT;
    DecArgs a{n, k, x, idx, idx_elem_size == 8, idx_stride, idx_off, w0, b0, w1, b1, w2, b2, (float *)out, out_stride,
              nullptr, 0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, 1};
    return dec_launch<true>((hipStream_t)stream, patches, a);
}

extern "C" int tpu3_dense_edge_conv_f32(tpu3_stream_t stream, int patches, int n, int k, const float *x,
                                        const void *idx, int idx_elem_size, int idx_stride, int idx_off,
                                        const float *w0, const float *b0, const float *w1, const float *b1,
                                        const float *w2, const float *b2, float *out, int out_stride, int mfma)
{
    if (patches < 0 || n <= 0 || k <= 0 || (k % 16) != 0 || k > 64) return TPU3_EINVAL;
    if (idx_elem_size != 4 && idx_elem_size != 8) return TPU3_EINVAL;
    if (idx_off < 0 || idx_stride < idx_off + k || out_stride < 60 || (out_stride % 4) != 0) return TPU3_EINVAL;
    if (mfma != TPU3_MFMA_F32 && mfma != TPU3_MFMA_F16) return TPU3_EINVAL;
    if (patches == 0) return TPU3_OK;
    if (!x || !idx || !w0 || !b0 || !w1 || !b1 || !w2 || !b2 || !out) return TPU3_EINVAL;
    if (((uintptr_t)out % 16) != 0 || ((uintptr_t)x % 16) != 0) return TPU3_EINVAL;
    if (patches > 65535 * 0 + 2147483647 / (n > 0 ? n : 1)) return TPU3_ELIMIT;       // patches * n must fit an int
    DecArgs a{n, k, x, idx, idx_elem_size == 8, idx_stride, idx_off, w0, b0, w1, b1, w2, b2, out, out_stride, nullptr,
              0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, 0};
    hipStream_t s = (hipStream_t)stream;
    if (mfma == TPU3_MFMA_F16)
        return dec_launch<true>(s, patches, a);
    // fp32: the lane-per-point 4x4x1 form whenever the patch's z table fits LDS (n <= ~3270 points); larger patches
    // take the 16x16x4 form with the table in global memory.  TPU3_DEC_FORM=16 (tuning hook) forces the latter.
    static const int form = getenv("TPU3_DEC_FORM") ? atoi(getenv("TPU3_DEC_FORM")) : 4;
    if (form == 4 && dec4_lds_bytes(n) <= 160 * 1024)
        return dec4_launch(s, patches, a);
    return dec_launch<false>(s, patches, a);
}

extern "C" size_t tpu3_dense_edge_conv_pack_floats(int fold_n)
{
    return (fold_n == 0 || fold_n == 24 || fold_n == 48 || fold_n == 72) ? (size_t)DEC4_PACK_FLOATS + (size_t)fold_n * 60 : 0;
}

extern "C" int tpu3_dense_edge_conv_pack_f32(tpu3_stream_t stream, const float *w0, const float *b0, const float *w1,
                                             const float *b1, const float *w2, const float *b2, int fold_n,
                                             const float *fold_w, float *pack)
{
    if (fold_n != 0 && fold_n != 24 && fold_n != 48 && fold_n != 72) return TPU3_EINVAL;
    if (!w0 || !b0 || !w1 || !b1 || !w2 || !b2 || !pack || (fold_n && !fold_w)) return TPU3_EINVAL;
    if ((((uintptr_t)pack | (uintptr_t)fold_w) & 15) != 0) return TPU3_EINVAL;
    DecArgs a{};
    a.w0 = w0; a.b0 = b0; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2;
    a.fold_n = fold_n;
    a.fold_w = fold_w;
    hipLaunchKernelGGL(dec4_pack_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, pack);
    return tpu3_launch_status();
}

extern "C" int tpu3_dense_edge_conv_pk_f32(tpu3_stream_t stream, int patches, int n, int k, const float *x,
                                           const void *idx, int idx_elem_size, int idx_stride, int idx_off,
                                           const float *pack, float *out, int out_stride)
{
    if (patches < 0 || n <= 0 || k <= 0 || (k % 16) != 0 || k > 64) return TPU3_EINVAL;
    if (idx_elem_size != 4 && idx_elem_size != 8) return TPU3_EINVAL;
    if (idx_off < 0 || idx_stride < idx_off + k || out_stride < 60 || (out_stride % 4) != 0) return TPU3_EINVAL;
    if (patches == 0) return TPU3_OK;
    if (!x || !idx || !pack || !out) return TPU3_EINVAL;
    if ((((uintptr_t)out | (uintptr_t)x | (uintptr_t)pack) & 15) != 0) return TPU3_EINVAL;
    if (patches > 2147483647 / n) return TPU3_ELIMIT;
    if (dec4_lds_bytes(n) > 160 * 1024) return TPU3_ELIMIT;                // (callers then pass the weights)
    DecArgs a{n, k, x, idx, idx_elem_size == 8, idx_stride, idx_off, nullptr, nullptr, nullptr, nullptr, nullptr,
              nullptr, out, out_stride, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, 0};
    a.pack = pack;
    return dec4_launch((hipStream_t)stream, patches, a);
}

extern "C" int tpu3_dense_edge_conv_fold_pk_f32(tpu3_stream_t stream, int patches, int n, int k, const float *x,
                                                const void *idx, int idx_elem_size, int idx_stride, int idx_off,
                                                const float *pack, float *out, int out_stride, int fold_n,
                                                const float *fold_b, float *acc, int acc_stride, int seed_off,
                                                int store_off, float *xnext)
{
    if (patches < 0 || n <= 0 || k <= 0 || (k % 16) != 0 || k > 64) return TPU3_EINVAL;
    if (idx_elem_size != 4 && idx_elem_size != 8) return TPU3_EINVAL;
    if (idx_off < 0 || idx_stride < idx_off + k || out_stride < 60 || (out_stride % 4) != 0) return TPU3_EINVAL;
    if (fold_n != 24 && fold_n != 48 && fold_n != 72) return TPU3_EINVAL;
    if (seed_off < 0 || store_off < 0 || (acc_stride % 4) || (seed_off % 4) || (store_off % 4)) return TPU3_EINVAL;
    if (!fold_b && acc_stride < seed_off + fold_n) return TPU3_EINVAL;
    if (fold_n > 24 && acc_stride < store_off + fold_n - 24) return TPU3_EINVAL;
    if (patches == 0) return TPU3_OK;
    if (!x || !idx || !pack || !out || !xnext) return TPU3_EINVAL;
    if ((fold_n > 24 || !fold_b) && !acc) return TPU3_EINVAL;
    if ((((uintptr_t)out | (uintptr_t)x | (uintptr_t)pack | (uintptr_t)fold_b | (uintptr_t)acc | (uintptr_t)xnext) & 15) != 0)
        return TPU3_EINVAL;
    if (patches > 2147483647 / n) return TPU3_ELIMIT;
    if (dec4_lds_bytes(n, fold_n) > 160 * 1024) return TPU3_ELIMIT;
    DecArgs a{n, k, x, idx, idx_elem_size == 8, idx_stride, idx_off, nullptr, nullptr, nullptr, nullptr, nullptr,
              nullptr, out, out_stride, nullptr, fold_n, nullptr, fold_b, acc, acc_stride, seed_off, store_off, xnext, 0};
    a.pack = pack;
    return dec4_launch((hipStream_t)stream, patches, a);
}

extern "C" int tpu3_dense_edge_conv_fold_f32(tpu3_stream_t stream, int patches, int n, int k, const float *x,
                                             const void *idx, int idx_elem_size, int idx_stride, int idx_off,
                                             const float *w0, const float *b0, const float *w1, const float *b1,
                                             const float *w2, const float *b2, float *out, int out_stride, int fold_n,
                                             const float *fold_w, const float *fold_b, float *acc, int acc_stride,
                                             int seed_off, int store_off, float *xnext)
{
    if (patches < 0 || n <= 0 || k <= 0 || (k % 16) != 0 || k > 64) return TPU3_EINVAL;
    if (idx_elem_size != 4 && idx_elem_size != 8) return TPU3_EINVAL;
    if (idx_off < 0 || idx_stride < idx_off + k || out_stride < 60 || (out_stride % 4) != 0) return TPU3_EINVAL;
    if (fold_n != 24 && fold_n != 48 && fold_n != 72) return TPU3_EINVAL;
    if (seed_off < 0 || store_off < 0 || (acc_stride % 4) || (seed_off % 4) || (store_off % 4)) return TPU3_EINVAL;
    if (!fold_b && acc_stride < seed_off + fold_n) return TPU3_EINVAL;
    if (fold_n > 24 && acc_stride < store_off + fold_n - 24) return TPU3_EINVAL;
    if (patches == 0) return TPU3_OK;
    if (!x || !idx || !w0 || !b0 || !w1 || !b1 || !w2 || !b2 || !out || !fold_w || !xnext) return TPU3_EINVAL;
    if ((fold_n > 24 || !fold_b) && !acc) return TPU3_EINVAL;
    if ((((uintptr_t)out | (uintptr_t)x | (uintptr_t)fold_w | (uintptr_t)fold_b | (uintptr_t)acc | (uintptr_t)xnext) & 15) != 0)
        return TPU3_EINVAL;
    if (patches > 2147483647 / n) return TPU3_ELIMIT;
    if (dec4_lds_bytes(n, fold_n) > 160 * 1024) return TPU3_ELIMIT;        // (callers then run the layers unfolded)
    DecArgs a{n, k, x, idx, idx_elem_size == 8, idx_stride, idx_off, w0, b0, w1, b1, w2, b2, out, out_stride, nullptr,
              fold_n, fold_w, fold_b, acc, acc_stride, seed_off, store_off, xnext, 0};
    return dec4_launch((hipStream_t)stream, patches, a);
}
