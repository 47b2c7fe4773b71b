// dec_train.hip -- DenseEdgeConv block for TRAINING on gfx950: forward with arg-max record, fused backward.
//
// Reference: network/layers.py:44-64 (C = 24 input channels, growth 12, three layers, k = 32 neighbours) under
// autograd (model.py:53-66).  The autograd formulation launches ~45 kernels per block and direction over
// (B, N, k, 36..60) edge tensors: library GEMMs with 12 outputs, cat, ReLU, max and their backward passes.  Here
// one launch per direction; the only edge tensors that reach memory are what the weight gradients need.
//
// Mathematics (the hoisting of the inference kernel): with d_j = x_j - x_i and W_0 = [W_0a | W_0b],
// W_1 = [W_1h | W_1x] (inputs [h_0, x_i]), W_2 = [W_2h | W_2x] (inputs [h_1, h_0, x_i])
//     a_0 = (b_0 + W_0a x_i) + W_0b d_j              h_0 = relu(a_0)
//     a_1 = (b_1 + W_1x x_i) + W_1h h_0              h_1 = relu(a_1)
//     h_2 = (b_2 + W_2x x_i) + W_2h [h_1, h_0]
//     y   = [max_k h_2 | max_k h_1 | max_k h_0 | x_i]
// A half wave (32 lanes) owns a point, a lane one of its 32 edges; the 36 per-point terms in brackets are computed
// once per point by the half wave (a lane per output) and handed over through LDS.  All weights sit in LDS and are
// read as wave-uniform float4 broadcasts.
//
// forward:  y (P,N,60) and arg (P,N,36) u8 = the edge slot that attains each maximum (lowest slot on ties).
// backward: the forward chain is recomputed per edge, the incoming gradient of channel c goes to the edge
//           arg[c], and with g_2, g_1 = relu'(a_1) (.. + W_2h1^T g_2), g_0 = relu'(a_0) (.. + W_2h0^T g_2 + W_1h^T g_1)
//   G (36 per edge) = [g_2 | g_1 | g_0],  Z (48 per edge) = [h_1 | h_0 | d_j]  -> weight gradients of the edge parts:
//                                      G^T Z is ACCUMULATED IN THIS KERNEL (r4, dt_wgrad_round; one 36 x 48 block per
//                                      workgroup for tpu3_dec_train_wgrad_f32 to add up) -- G and Z never reach memory
//   S (points, 36) = sum over the point's edges of G                            -> weight gradients of the x_i parts
//                                                                                  (S^T X) and all bias gradients
//   gx (points, 24) += gy_x + [W_2x; W_1x; W_0a - W_0b]^T S   (own point)   and   gx[j] += W_0b^T g_0   (neighbour),
//   hardware float atomics (gx zeroed by the caller).
#include "tpu3_dev.h"

namespace {

constexpr int DT_C = 24, DT_G = 12, DT_K = 32;
constexpr int DT_THREADS = 256;             // 4 waves = 8 points per pass
constexpr int DT_TRS = 61;                  // floats per edge row of the transposition region (60 used)

struct DtArgs {
    long points;                    // P * N
    int n;                          // points per patch
    int idx_stride, idx_off;        // idx (P,N,idx_stride), neighbours at idx_off .. idx_off + 31
    const float *x;                 // (P,N,24)
    const int32_t *idx;
    const float *w0, *b0, *w1, *b1, *w2, *b2;      // (12,48), (12,36), (12,48) row-major + biases
    float *y;                       // fwd: (P,N,60)
    uint8_t *arg;                   // (P,N,36)
    const float *gy;                // bwd: (P,N,60) rows, `gys` floats apart
    float *gx;                      // (P,N,24), accumulated
    float *S;                       // (P*N, 36)
    float *wpart;                   // (gridDim.x, 64, 64): the workgroups' G^T Z blocks (rows 0 .. 35, columns 0 .. 47 used)
    int gys = 60;
};

// LDS image of the weights:
//   eh  [12][24]  W_0b                       (edge part of layer 0, acts on d_j)
//   h1w [12][12]  W_1h                       (acts on h_0)
//   h2w [12][24]  W_2h                       (acts on [h_1, h_0])
//   xw  [36][25]  rows 0-11 W_2x, 12-23 W_1x, 24-35 W_0a (x_i parts, padded rows: a lane per row reads conflict-free)
//   xb  [36]      b_2, b_1, b_0
//   stage [8][36] per half wave: the per-point terms (forward) / the summed gradients S (backward)
struct DtWeights {
    float eh[12 * 24];
    float h1w[12 * 12];
    float h2w[12 * 24];
    float xw[36 * 25];
    float xb[36];
    float stage[DT_THREADS / 32][40];
};
// (advisor, r4) the forward kernel declares only the image above (~7 KB); the backward kernel adds its transposition
// regions -- about 75 KB, two workgroups per compute unit of gfx950's 160 KB (it runs one wave per SIMD pair anyway)
struct DtLds : DtWeights {
    // backward: per wave, its 64 edges' rows [g (12, one of g_2 / g_1 / g_0 at a time) | h_1 | h_0 | d_j] with an odd
    // stride -- written a lane per edge, read back a lane per CHANNEL as the operands of the weight-gradient matrix
    // instructions (dt_wgrad_round); afterwards the same floats hold the 2 x 32 neighbour shares [edge][25] of the scatter
    float tr[DT_THREADS / 64][64 * DT_TRS + 8];
    int nbrow[DT_THREADS / 32][DT_K];       // rows of the neighbour shares
    float sgy[DT_THREADS / 32][64];         //           the point's incoming gradient row (60) ...
    int sarg[DT_THREADS / 32][40];          //           ... and its 36 arg-max slots, requested with the pass's first loads
};
static_assert(sizeof(DtLds) <= 80 * 1024, "dec_train_bwd_kernel: two workgroups per compute unit");

__device__ __forceinline__ void dt_load_weights(const DtArgs &a, DtWeights &s)
{
    // every global load is requested before the first LDS store (r4): as ten load -> store iterations the image cost
    // ten dependent round trips at the head of every workgroup
    const int tid = threadIdx.x;
    float r_eh[2], r_h2[2], r_h1 = 0.f, r_xw[4], r_xb = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + i * DT_THREADS, c = e / 24, d = e - c * 24;
        r_eh[i] = e < 12 * 24 ? a.w0[c * 48 + 24 + d] : 0.f;
        r_h2[i] = e < 12 * 24 ? a.w2[c * 48 + d] : 0.f;
    }
    if (tid < 12 * 12) {
        const int c = tid / 12, d = tid - c * 12;
        r_h1 = a.w1[c * 36 + d];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + i * DT_THREADS, r = e / 24, d = e - r * 24, c = r % 12;
        const float *src = r < 12 ? a.w2 + c * 48 + 24 + d : (r < 24 ? a.w1 + c * 36 + 12 + d : a.w0 + c * 48 + d);
        r_xw[i] = e < 36 * 24 ? *src : 0.f;
    }
    if (tid < 36)
        r_xb = tid < 12 ? a.b2[tid] : (tid < 24 ? a.b1[tid - 12] : a.b0[tid - 24]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + i * DT_THREADS;
        if (e < 12 * 24) {
            s.eh[e] = r_eh[i];
            s.h2w[e] = r_h2[i];
        }
    }
    if (tid < 12 * 12)
        s.h1w[tid] = r_h1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + i * DT_THREADS, r = e / 24, d = e - r * 24;
        if (e < 36 * 24)
            s.xw[r * 25 + d] = r_xw[i];
    }
    if (tid < 36)
        s.xb[tid] = r_xb;
    __syncthreads();
}

// max / sum over the 32 lanes of a half wave, result in every lane of the half
__device__ __forceinline__ float dt_half_max(float v)
{
#define DT_DPP(V, CTRL) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(V), CTRL, 0xF, 0xF, false))
    v = fmaxf(v, DT_DPP(v, 0xB1));
    v = fmaxf(v, DT_DPP(v, 0x4E));
    v = fmaxf(v, DT_DPP(v, 0x141));
    v = fmaxf(v, DT_DPP(v, 0x140));
    return fmaxf(v, __shfl_xor(v, 16, 64));
}
__device__ __forceinline__ float dt_half_sum(float v)
{
    v += DT_DPP(v, 0xB1);
    v += DT_DPP(v, 0x4E);
    v += DT_DPP(v, 0x141);
    v += DT_DPP(v, 0x140);
    return v + __shfl_xor(v, 16, 64);
#undef DT_DPP
}

// Per-point terms of point `pt` by its half wave: lane c < 32 computes term c, lanes 0-3 also terms 32-35; every
// lane then reads all 36 from `st`.  (pt may be a clamped duplicate for idle halves.)
__device__ __forceinline__ void dt_point_terms(const DtWeights &s, float *st, const float *xi, int hl)
{
    float t0 = s.xb[hl], t1 = s.xb[32 + (hl & 3)];
#pragma unroll
    for (int d = 0; d < DT_C; ++d) {
        t0 = __builtin_fmaf(s.xw[hl * 25 + d], xi[d], t0);
        t1 = __builtin_fmaf(s.xw[(32 + (hl & 3)) * 25 + d], xi[d], t1);
    }
    st[hl] = t0;
    if (hl < 4)
        st[32 + hl] = t1;
}

// ---- (r4) the edge chains on v_mfma_f32_4x4x1 ---------------------------------------------------------------------
// A lane owns an edge, and every product of the block is "12 or 24 outputs = W x (the lane's own vector)" with the same W
// in every lane -- the layout of dec_fused4_kernel (csrc/dense_edge_conv.hip): one v_mfma_f32_4x4x1 with B = component k
// of the lane's vector and A = W[4 rg .. 4 rg + 3][k] adds input k to outputs 4 rg .. 4 rg + 3 of 64 edges; the
// instruction's A-broadcast control (cbsz = 4, abid = b) takes A from block b (lanes 4 b .. 4 b + 3) for all blocks, so
// ONE register carries 16 (row group, k) operands and the six matrices of forward and backward live in 26 registers.
// As scalar fmaf chains with the weights as wave-uniform LDS reads the backward kernel issued ~380 LDS reads per pass,
// each waited for by the ONE wave per SIMD its 256 + 138 registers allowed: 20 us per pass.  The sums run in the same
// order as before (ascending k, seeded with the same value): same bits.
typedef float dt_f4 __attribute__((ext_vector_type(4)));

// operand register v of a matrix with K inputs: lane 4 b + i holds entry (row 4 rg + i, column k) of operand
// q = 16 v + b = rg K + k; `rs` / `cs` = strides of the matrix's rows / columns in `w` (cs = 1: as stored, rs = 1:
// transposed)
template <int NREG>
__device__ __forceinline__ void dt_load_operands(float (&op)[NREG], const float *w, int K, int count, int rs, int cs)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int v = 0; v < NREG; ++v) {
        const int q = 16 * v + (lane >> 2), i = lane & 3;
        const int rg = q / K, k = q - rg * K;
        op[v] = q < count ? w[(4 * rg + i) * rs + k * cs] : 0.f;
    }
}

template <int Q, int NREG>
__device__ __forceinline__ dt_f4 dt_mfma(const float (&op)[NREG], float b, dt_f4 c)
{
    return __builtin_amdgcn_mfma_f32_4x4x1f32(op[Q / 16], b, c, 4, Q % 16, 0);
}

template <int I, int N, typename F>
__device__ __forceinline__ void dt_static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        dt_static_for<I + 1, N>(f);
    }
}

// out[4 rg + i] += sum_k W[4 rg + i][k] in[k] for RG row groups, K inputs (ascending k: one fma chain per output)
template <int RG, int K, int NREG>
__device__ __forceinline__ void dt_matvec(const float (&op)[NREG], const float (&in)[K], float (&out)[4 * RG])
{
    dt_f4 acc[RG];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
        acc[rg] = (dt_f4){out[4 * rg], out[4 * rg + 1], out[4 * rg + 2], out[4 * rg + 3]};
    dt_static_for<0, K>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        dt_static_for<0, RG>([&](auto rc) __attribute__((always_inline)) {
            constexpr int rg = decltype(rc)::value;
            acc[rg] = dt_mfma<rg * K + k>(op, in[k], acc[rg]);
        });
    });
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
        out[4 * rg] = acc[rg][0]; out[4 * rg + 1] = acc[rg][1]; out[4 * rg + 2] = acc[rg][2]; out[4 * rg + 3] = acc[rg][3];
    }
}

struct DtFwdOps {
    float w0b[5], w1h[3], w2h[5];       // 3 x 24, 3 x 12, 3 x 24 operands
};

__device__ __forceinline__ void dt_load_fwd_ops(const DtArgs &a, DtFwdOps &o, bool with_h2)
{
    dt_load_operands(o.w0b, a.w0 + 24, 24, 72, 48, 1);
    dt_load_operands(o.w1h, a.w1, 12, 36, 36, 1);
    if (with_h2)
        dt_load_operands(o.w2h, a.w2, 24, 72, 48, 1);
}

// forward chain of one edge: a0, a1 (pre-activations) and (H2) h2
template <bool H2>
__device__ __forceinline__ void dt_edge_forward(const DtFwdOps &o, const float *st, const float (&dj)[DT_C],
                                                float (&a0)[DT_G], float (&a1)[DT_G], float (&h2)[DT_G])
{
#pragma unroll
    for (int c = 0; c < DT_G; ++c) {
        a0[c] = st[24 + c];
        a1[c] = st[12 + c];
    }
    dt_matvec<3, 24>(o.w0b, dj, a0);
    float r0[DT_G];
#pragma unroll
    for (int c = 0; c < DT_G; ++c)
        r0[c] = fmaxf(a0[c], 0.f);
    dt_matvec<3, 12>(o.w1h, r0, a1);
    if constexpr (H2) {
        float r10[24];
#pragma unroll
        for (int c = 0; c < DT_G; ++c) {
            h2[c] = st[c];
            r10[c] = fmaxf(a1[c], 0.f);
            r10[12 + c] = r0[c];
        }
        dt_matvec<3, 24>(o.w2h, r10, h2);
    }
}

__device__ __forceinline__ void dt_load_edge(const DtArgs &a, long pt, int hl, float (&xi)[DT_C], float (&dj)[DT_C],
                                             long &jrow)
{
    const long patch = pt / a.n;
    const float4 *X = (const float4 *)(a.x + pt * DT_C);
    int j = a.idx[pt * a.idx_stride + a.idx_off + hl];
    j = min(max(j, 0), a.n - 1);
    jrow = patch * a.n + j;
    const float4 *XJ = (const float4 *)(a.x + jrow * DT_C);
#pragma unroll
    for (int q = 0; q < DT_C / 4; ++q) {
        const float4 u = X[q], v = XJ[q];
        xi[4 * q] = u.x; xi[4 * q + 1] = u.y; xi[4 * q + 2] = u.z; xi[4 * q + 3] = u.w;
        dj[4 * q] = v.x - u.x; dj[4 * q + 1] = v.y - u.y; dj[4 * q + 2] = v.z - u.z; dj[4 * q + 3] = v.w - u.w;
    }
}

__global__ __launch_bounds__(DT_THREADS) void dec_train_fwd_kernel(DtArgs a)
{
    __shared__ DtWeights s;
    DtFwdOps ops;
    dt_load_fwd_ops(a, ops, true);
    dt_load_weights(a, s);
    const int half = threadIdx.x >> 5, hl = threadIdx.x & 31;
    float *st = s.stage[half];
    const long per_pass = (long)gridDim.x * (DT_THREADS / 32);
    const long passes = (a.points + per_pass - 1) / per_pass;
    for (long it = 0; it < passes; ++it) {
        const long pt0 = (it * gridDim.x + blockIdx.x) * (DT_THREADS / 32) + half;
        const bool live = pt0 < a.points;
        const long pt = live ? pt0 : a.points - 1;
        float xi[DT_C], dj[DT_C];
        long jrow;
        dt_load_edge(a, pt, hl, xi, dj, jrow);
        dt_point_terms(s, st, xi, hl);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (the half wave's own LDS writes, no barrier needed
        __builtin_amdgcn_wave_barrier();                            //  across waves: a half never spans two)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        float a0[DT_G], a1[DT_G], h2[DT_G];
        dt_edge_forward<true>(ops, st, dj, a0, a1, h2);
        // maxima over the 32 edges and the slot that attains each (lowest on ties)
        float m[36];
        uint32_t slot[36];
        const int hshift = (threadIdx.x & 32);                      // this half's bits in the wave's ballot
#pragma unroll
        for (int c = 0; c < 36; ++c) {
            const float v = c < 12 ? h2[c] : (c < 24 ? fmaxf(a1[c - 12], 0.f) : fmaxf(a0[c - 24], 0.f));
            m[c] = dt_half_max(v);
            const unsigned long long eq = __ballot(v == m[c]);
            slot[c] = (uint32_t)__builtin_ctz((uint32_t)(eq >> hshift) | 0x80000000u);
        }
        if (live && hl == 0) {
            float *yo = a.y + pt * 60;
#pragma unroll
            for (int c = 0; c < 36; ++c)
                yo[c] = m[c];
#pragma unroll
            for (int d = 0; d < DT_C; ++d)
                yo[36 + d] = xi[d];
            uint8_t *ao = a.arg + pt * 36;
#pragma unroll
            for (int c = 0; c < 36; ++c)
                ao[c] = (uint8_t)slot[c];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- (r4) the weight gradients of the edge parts, in the backward kernel ------------------------------------------------
// dW[o][c] = sum over edges of g[e][o] z[e][c] is a matrix product with the EDGES as the inner dimension, and the kernel
// holds an edge per lane: the wave's 64 rows [g | h_1 | h_0 | d_j] go through LDS once (a lane per edge writes its row, stride
// 61: conflict-free both ways) and come back a lane per channel as operands of v_mfma_f32_16x16x4_f32 -- A[o][k] = g of
// edge 4 t + k, B[k][c] = z of edge 4 t + k, 16 steps of four edges.  Three rounds (g_2 x [h_1 h_0], g_1 x h_0, g_0 x d_j),
// five 16 x 16 accumulators per wave that live across the passes: 80 matrix instructions per 64 edges.  Before, G and Z
// went to memory (107 MB per block, 21 store instructions per lane) and a second kernel streamed them back through LDS
// for the same product: 57 us of a block's 135.  Columns beyond a block's width and rows beyond 12 compute on whatever
// the neighbouring LDS words hold and are never stored.
typedef float dt_v4 __attribute__((ext_vector_type(4)));

template <int NT>
__device__ __forceinline__ void dt_wgrad_round(const float *tw, int bcol, dt_v4 (&acc)[NT])
{
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const float *r = tw + g * DT_TRS + i;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const float av = r[(4 * t) * DT_TRS];
#pragma unroll
        for (int n = 0; n < NT; ++n)
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, r[(4 * t) * DT_TRS + bcol + 16 * n], acc[n], 0, 0, 0);
    }
}

__device__ __forceinline__ void dt_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__global__ __launch_bounds__(DT_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void dec_train_bwd_kernel(DtArgs a)
{
    __shared__ DtLds s;
    DtFwdOps ops;
    dt_load_fwd_ops(a, ops, false);
    // transposed operands of the backward products: t = W_2h^T g_2 (24 outputs = 6 row groups, 12 inputs),
    // u = W_1h^T g_1 (12, 12), vb = W_0b^T g_0 (24, 12)
    float t2[5], t1[3], t0[5];
    dt_load_operands(t2, a.w2, 12, 72, 1, 48);
    dt_load_operands(t1, a.w1, 12, 36, 1, 36);
    dt_load_operands(t0, a.w0 + 24, 12, 72, 1, 48);
    dt_load_weights(a, s);
    const int half = threadIdx.x >> 5, hl = threadIdx.x & 31;
    float *st = s.stage[half];
    float *tw = s.tr[threadIdx.x >> 6];                             // the wave's transposition region
    float *trow = tw + (threadIdx.x & 63) * DT_TRS;                 // ... and the lane's (edge's) row in it
    dt_v4 wacc2[2], wacc1[1], wacc0[2];                             // G2^T [h1 h0] (cols 0-15, 16-31), G1^T h0, G0^T d_j (0-15, 16-31)
#pragma unroll
    for (int n = 0; n < 2; ++n)
        wacc2[n] = wacc0[n] = (dt_v4){0.f, 0.f, 0.f, 0.f};
    wacc1[0] = (dt_v4){0.f, 0.f, 0.f, 0.f};
    const long per_pass = (long)gridDim.x * (DT_THREADS / 32);
    const long passes = (a.points + per_pass - 1) / per_pass;
    for (long it = 0; it < passes; ++it) {
        const long pt0 = (it * gridDim.x + blockIdx.x) * (DT_THREADS / 32) + half;
        const bool live = pt0 < a.points;
        const long pt = live ? pt0 : a.points - 1;
        float xi[DT_C], dj[DT_C];
        long jrow;
        // the point's incoming gradient and arg-max record: requested together with the rows (behind other memory
        // operations their loads waited for everything before them: 45 % of the kernel once), handed over through LDS
        const float gy_a = a.gy[pt * a.gys + hl], gy_b = hl < 28 ? a.gy[pt * a.gys + 32 + hl] : 0.f;
        const int ar_a = a.arg[pt * 36 + hl], ar_b = hl < 4 ? a.arg[pt * 36 + 32 + hl] : 0;
        dt_load_edge(a, pt, hl, xi, dj, jrow);
        s.sgy[half][hl] = gy_a;
        s.sgy[half][32 + hl] = gy_b;
        s.sarg[half][hl] = ar_a;
        if (hl < 4)
            s.sarg[half][32 + hl] = ar_b;
        dt_point_terms(s, st, xi, hl);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        float a0[DT_G], a1[DT_G], h2[DT_G];
        dt_edge_forward<false>(ops, st, dj, a0, a1, h2);
        // Z = [h1 | h0 | d_j] into the edge's row of the transposition region (columns 12 .. 59)
#pragma unroll
        for (int c = 0; c < DT_G; ++c) {
            trow[12 + c] = fmaxf(a1[c], 0.f);
            trow[24 + c] = fmaxf(a0[c], 0.f);
        }
#pragma unroll
        for (int d = 0; d < DT_C; ++d)
            trow[36 + d] = dj[d];
        __builtin_amdgcn_sched_barrier(0);
        // incoming gradient of channel c goes to the edge that attained the maximum
        const float *gyp = s.sgy[half];
        const int *ap = s.sarg[half];
        float g2[DT_G], g1[DT_G], g0[DT_G];
#pragma unroll
        for (int c = 0; c < DT_G; ++c)
            g2[c] = (live && (int)ap[c] == hl) ? gyp[c] : 0.f;        // (idle halves: no contribution to anything)
#pragma unroll
        for (int c = 0; c < DT_G; ++c)
            trow[c] = g2[c];
        // S = sums over the point's edges, 12 at a time as the g's appear (`st`'s forward terms are all read by now):
        // taken at the end of the pass they kept all 36 values alive across the three rounds
#pragma unroll
        for (int c = 0; c < DT_G; ++c) {
            const float v = dt_half_sum(g2[c]);
            if (hl == 0)
                st[c] = v;
        }
        dt_wave_sync();
        dt_wgrad_round<2>(tw, 12, wacc2);
        float t[24];                                                // W_2h^T g_2: gradient w.r.t. [h_1, h_0]
#pragma unroll
        for (int d = 0; d < 24; ++d)
            t[d] = 0.f;
        dt_matvec<6, 12>(t2, g2, t);
#pragma unroll
        for (int c = 0; c < DT_G; ++c)
            g1[c] = (live && a1[c] > 0.f) ? ((int)ap[12 + c] == hl ? gyp[12 + c] : 0.f) + t[c] : 0.f;
        dt_wave_sync();                                             // (round 1's reads of column block 0 are done)
#pragma unroll
        for (int c = 0; c < DT_G; ++c)
            trow[c] = g1[c];
#pragma unroll
        for (int c = 0; c < DT_G; ++c) {
            const float v = dt_half_sum(g1[c]);
            if (hl == 0)
                st[12 + c] = v;
        }
        dt_wave_sync();
        dt_wgrad_round<1>(tw, 24, wacc1);
        float u[DT_G];                                              // W_1h^T g_1: gradient w.r.t. h_0
#pragma unroll
        for (int d = 0; d < DT_G; ++d)
            u[d] = 0.f;
        dt_matvec<3, 12>(t1, g1, u);
#pragma unroll
        for (int c = 0; c < DT_G; ++c)
            g0[c] = (live && a0[c] > 0.f) ? ((int)ap[24 + c] == hl ? gyp[24 + c] : 0.f) + t[12 + c] + u[c] : 0.f;
        dt_wave_sync();
#pragma unroll
        for (int c = 0; c < DT_G; ++c)
            trow[c] = g0[c];
#pragma unroll
        for (int c = 0; c < DT_G; ++c) {
            const float v = dt_half_sum(g0[c]);
            if (hl == 0)
                st[24 + c] = v;
        }
        dt_wave_sync();
        dt_wgrad_round<2>(tw, 36, wacc0);
        dt_wave_sync();                                             // (the region is free for the neighbour shares)
        __builtin_amdgcn_sched_barrier(0);
        // neighbour's share: W_0b^T g_0
        {
            float vb[DT_C];
#pragma unroll
            for (int d = 0; d < DT_C; ++d)
                vb[d] = 0.f;
            dt_matvec<6, 12>(t0, g0, vb);
            // scatter with a lane per CHANNEL: 24 lanes add one neighbour's 96 contiguous bytes per instruction (a
            // lane per edge would issue 64 separate 4-byte atomics per instruction: 0.46 ms per launch, 24x the
            // L2 transactions)
            float *nbs = tw + ((threadIdx.x & 63) >> 5) * (DT_K * 25) + hl * 25;   // [half of the wave][edge][25]
#pragma unroll
            for (int d = 0; d < DT_C; ++d)
                nbs[d] = vb[d];
            s.nbrow[half][hl] = (int)jrow;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (live && hl < DT_C) {
#pragma unroll
            for (int e = 0; e < DT_K; ++e)
                atomicAdd(a.gx + (long)s.nbrow[half][e] * DT_C + hl, tw[((threadIdx.x & 63) >> 5) * (DT_K * 25) + e * 25 + hl]);
        }
        // (S = the sums over the point's edges of [g_2 | g_1 | g_0] are in `st` by now, see above)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (live) {
            if (hl < 9)
                ((float4 *)(a.S + pt * 36))[hl] = make_float4(st[4 * hl], st[4 * hl + 1], st[4 * hl + 2], st[4 * hl + 3]);
            if (hl < DT_C) {
                // own point: gy_x + W_2x^T S_2 + W_1x^T S_1 + W_0a^T S_0 - W_0b^T S_0   (lane d < 24 owns channel d)
                float acc = gyp[36 + hl];
#pragma unroll
                for (int r = 0; r < 36; ++r)
                    acc = __builtin_fmaf(s.xw[r * 25 + hl], st[r], acc);
#pragma unroll
                for (int c = 0; c < DT_G; ++c)
                    acc = __builtin_fmaf(-s.eh[c * 24 + hl], st[24 + c], acc);
                atomicAdd(a.gx + pt * DT_C + hl, acc);
            }
        }
        dt_wave_sync();                                             // (the shares are read: the region takes rows again)
    }
    // the workgroup's G^T Z block: the four waves' accumulators added in a fixed order (through the transposition
    // regions), rows = G column (g_2 0-11, g_1 12-23, g_0 24-35), columns = Z column, (64, 64) per workgroup
    __syncthreads();
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        float *red = &s.tr[0][0] + wv * (5 * 4 * 64);               // [pair][q][lane]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            red[(0 * 4 + q) * 64 + lane] = wacc2[0][q];
            red[(1 * 4 + q) * 64 + lane] = wacc2[1][q];
            red[(2 * 4 + q) * 64 + lane] = wacc1[0][q];
            red[(3 * 4 + q) * 64 + lane] = wacc0[0][q];
            red[(4 * 4 + q) * 64 + lane] = wacc0[1][q];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 5 * 4 * 64; e += DT_THREADS) {
        const float *red = &s.tr[0][0];
        const float v = (red[e] + red[e + 1280]) + (red[e + 2560] + red[e + 3840]);
        const int pair = e >> 8, q = (e >> 6) & 3, lane = e & 63, i = lane & 15, o = 4 * (lane >> 4) + q;
        // pair 0 / 1: rows o, columns i / 16 + i (< 24); pair 2: rows 12 + o, columns 12 + i (< 24); pair 3 / 4: rows
        // 24 + o, columns 24 + i / 40 + i (< 48)
        const int row = pair < 2 ? o : (pair == 2 ? 12 + o : 24 + o);
        const int col = pair == 0 ? i : pair == 1 ? 16 + i : pair == 2 ? 12 + i : pair == 3 ? 24 + i : 40 + i;
        const bool ok = o < 12 && (pair == 1 ? i < 8 : pair == 2 ? i < 12 : pair == 4 ? i < 8 : true);
        if (ok)
            a.wpart[(size_t)blockIdx.x * 4096 + row * 64 + col] = v;
    }
}

int dt_check(long p, int n, int k, int idx_stride, int idx_off)
{
    if (p < 0 || n <= 0 || idx_stride <= 0 || idx_off < 0) return TPU3_EINVAL;
    if (k != DT_K || idx_off + k > idx_stride) return TPU3_ELIMIT;
    return TPU3_OK;
}

} // namespace

// workgroups of the backward launch = blocks of its weight-gradient output (tpu3_dec_train_wgrad_f32 adds them up)
int tpu3_dec_train_bwd_blocks(long points);

namespace {

// One workgroup per compute-unit slot, walking its passes (r4): a training batch is ~1250 passes of 8 points; launched as
// 1248 workgroups of one pass each (the cap used to be 2048) every workgroup paid the weight image and the operand
// registers for a single pass.  `per_cu` = workgroups a compute unit holds (two at ~200 registers); TPU3_DT_GRID
// overrides the cap (tuning hook).
unsigned dt_grid(long points, int per_cu)
{
    static const int cus = []() { int d = 0, v = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d); return v; }();
    static const long cap_env = getenv("TPU3_DT_GRID") ? atol(getenv("TPU3_DT_GRID")) : 0;
    long blocks = (points + DT_THREADS / 32 - 1) / (DT_THREADS / 32);
    const long cap = cap_env > 0 ? cap_env : (long)cus * per_cu;
    if (blocks > cap) blocks = cap;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

} // namespace

extern "C" int tpu3_dec_train_fwd_f32(tpu3_stream_t stream, long p, int n, int k, const float *x, const int32_t *idx,
                                      int idx_stride, int idx_off, const float *w0, const float *b0, const float *w1,
                                      const float *b1, const float *w2, const float *b2, float *y, uint8_t *arg)
{
    const int r = dt_check(p, n, k, idx_stride, idx_off);
    if (r) return r;
    if (p == 0) return TPU3_OK;
    if (!x || !idx || !w0 || !b0 || !w1 || !b1 || !w2 || !b2 || !y || !arg) return TPU3_EINVAL;
    if (((uintptr_t)x & 15) != 0) return TPU3_ELIMIT;
    DtArgs a{p * n, n, idx_stride, idx_off, x, idx, w0, b0, w1, b1, w2, b2, y, arg, nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(dec_train_fwd_kernel, dim3(dt_grid(a.points, 2)), dim3(DT_THREADS), 0, (hipStream_t)stream, a);
    return tpu3_launch_status();
}

int tpu3_dec_train_bwd_blocks(long points)
{
    return (int)dt_grid(points, 2);
}

extern "C" int tpu3_dec_train_bwd_f32(tpu3_stream_t stream, long p, int n, int k, const float *x, const int32_t *idx,
                                      int idx_stride, int idx_off, const float *w0, const float *b0, const float *w1,
                                      const float *b1, const float *w2, const float *b2, const uint8_t *arg,
                                      const float *gy, int gy_stride, float *gx, float *S, void *workspace,
                                      size_t workspace_bytes)
{
    const int r = dt_check(p, n, k, idx_stride, idx_off);
    if (gy_stride < 60) return TPU3_EINVAL;
    if (r) return r;
    if (p == 0) return TPU3_OK;
    if (!x || !idx || !w0 || !b0 || !w1 || !b1 || !w2 || !b2 || !arg || !gy || !gx || !S || !workspace) return TPU3_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)S | (uintptr_t)workspace) & 15) != 0) return TPU3_EINVAL;
    const unsigned grid = dt_grid(p * n, 2);
    if (workspace_bytes < (size_t)grid * 4096 * sizeof(float)) return TPU3_EINVAL;
    DtArgs a{p * n, n, idx_stride, idx_off, x, idx, w0, b0, w1, b1, w2, b2, nullptr, const_cast<uint8_t *>(arg), gy, gx,
             S, (float *)workspace, gy_stride};
    hipLaunchKernelGGL(dec_train_bwd_kernel, dim3(grid), dim3(DT_THREADS), 0, (hipStream_t)stream, a);
    return tpu3_launch_status();
}
