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
//   G (36 per edge) = [g_2 | g_1 | g_0],  Z (48 per edge) = [h_1 | h_0 | d_j]  -> weight gradients of the edge parts
//                                      (tpu3_dec_train_wgrad_f32); both as float4 PLANES over the edges (r4, see the stores)
//   S (points, 36) = sum over the point's edges of G                            -> weight gradients of the x_i parts
//                                                                                  (S^T X) and all bias gradients
//   gx (points, 24) += gy_x + [W_2x; W_1x; W_0a - W_0b]^T S   (own point)   and   gx[j] += W_0b^T g_0   (neighbour),
//   hardware float atomics (gx zeroed by the caller).
#include "tpu3_dev.h"

namespace {

constexpr int DT_C = 24, DT_G = 12, DT_K = 32;
constexpr int DT_THREADS = 256;             // 4 waves = 8 points per pass

struct DtArgs {
    long points;                    // P * N
    int n;                          // points per patch
    int idx_stride, idx_off;        // idx (P,N,idx_stride), neighbours at idx_off .. idx_off + 31
    const float *x;                 // (P,N,24)
    const int32_t *idx;
    const float *w0, *b0, *w1, *b1, *w2, *b2;      // (12,48), (12,36), (12,48) row-major + biases
    float *y;                       // fwd: (P,N,60)
    uint8_t *arg;                   // (P,N,36)
    const float *gy;                // bwd: (P,N,60)
    float *gx;                      // (P,N,24), accumulated
    float *G, *Z, *S;               // 9 / 12 float4 planes over the P*N*32 edges; (P*N, 36)
    long plane;                     // float4 entries between planes (tpu3_dec_train_plane_stride)
};

// LDS image of the weights:
//   eh  [12][24]  W_0b                       (edge part of layer 0, acts on d_j)
//   h1w [12][12]  W_1h                       (acts on h_0)
//   h2w [12][24]  W_2h                       (acts on [h_1, h_0])
//   xw  [36][25]  rows 0-11 W_2x, 12-23 W_1x, 24-35 W_0a (x_i parts, padded rows: a lane per row reads conflict-free)
//   xb  [36]      b_2, b_1, b_0
//   stage [8][36] per half wave: the per-point terms (forward) / the summed gradients S (backward)
struct DtLds {
    float eh[12 * 24];
    float h1w[12 * 12];
    float h2w[12 * 24];
    float xw[36 * 25];
    float xb[36];
    float stage[DT_THREADS / 32][40];
    float nb[DT_THREADS / 32][DT_K][25];    // backward: the 32 neighbour shares of a half wave, [edge][channel] (padded)
    int nbrow[DT_THREADS / 32][DT_K];       //           and their rows
    float sgy[DT_THREADS / 32][64];         //           the point's incoming gradient row (60) ...
    int sarg[DT_THREADS / 32][40];          //           ... and its 36 arg-max slots, requested with the pass's first loads
};

__device__ __forceinline__ void dt_load_weights(const DtArgs &a, DtLds &s)
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
__device__ __forceinline__ void dt_point_terms(const DtLds &s, float *st, const float *xi, int hl)
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
    __shared__ DtLds s;
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

__global__ __launch_bounds__(DT_THREADS) void dec_train_bwd_kernel(DtArgs a)
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
    const long per_pass = (long)gridDim.x * (DT_THREADS / 32);
    const long passes = (a.points + per_pass - 1) / per_pass;
    for (long it = 0; it < passes; ++it) {
        const long pt0 = (it * gridDim.x + blockIdx.x) * (DT_THREADS / 32) + half;
        const bool live = pt0 < a.points;
        const long pt = live ? pt0 : a.points - 1;
        float xi[DT_C], dj[DT_C];
        long jrow;
        // the point's incoming gradient and arg-max record: requested together with the rows (behind the Z / G stores
        // their loads waited for every store before them: 45 % of the kernel), handed to the half wave through LDS
        const float gy_a = a.gy[pt * 60 + hl], gy_b = hl < 28 ? a.gy[pt * 60 + 32 + hl] : 0.f;
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
        const long edge = pt * DT_K + hl;
        if (live) {                                                 // Z = [h1 | h0 | d_j], float4 planes over the edges
            float4 *zo = (float4 *)a.Z + edge;
            const size_t ps = (size_t)a.plane;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                zo[q * ps] = make_float4(fmaxf(a1[4 * q], 0.f), fmaxf(a1[4 * q + 1], 0.f), fmaxf(a1[4 * q + 2], 0.f),
                                         fmaxf(a1[4 * q + 3], 0.f));
#pragma unroll
            for (int q = 0; q < 3; ++q)
                zo[(3 + q) * ps] = make_float4(fmaxf(a0[4 * q], 0.f), fmaxf(a0[4 * q + 1], 0.f), fmaxf(a0[4 * q + 2], 0.f),
                                               fmaxf(a0[4 * q + 3], 0.f));
#pragma unroll
            for (int q = 0; q < 6; ++q)
                zo[(6 + q) * ps] = make_float4(dj[4 * q], dj[4 * q + 1], dj[4 * q + 2], dj[4 * q + 3]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // incoming gradient of channel c goes to the edge that attained the maximum
        const float *gyp = s.sgy[half];
        const int *ap = s.sarg[half];
        float g2[DT_G], g1[DT_G], g0[DT_G];
#pragma unroll
        for (int c = 0; c < DT_G; ++c)
            g2[c] = (int)ap[c] == hl ? gyp[c] : 0.f;
        float t[24];                                                // W_2h^T g_2: gradient w.r.t. [h_1, h_0]
#pragma unroll
        for (int d = 0; d < 24; ++d)
            t[d] = 0.f;
        dt_matvec<6, 12>(t2, g2, t);
#pragma unroll
        for (int c = 0; c < DT_G; ++c)
            g1[c] = a1[c] > 0.f ? ((int)ap[12 + c] == hl ? gyp[12 + c] : 0.f) + t[c] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
        float u[DT_G];                                              // W_1h^T g_1: gradient w.r.t. h_0
#pragma unroll
        for (int d = 0; d < DT_G; ++d)
            u[d] = 0.f;
        dt_matvec<3, 12>(t1, g1, u);
#pragma unroll
        for (int c = 0; c < DT_G; ++c)
            g0[c] = a0[c] > 0.f ? ((int)ap[24 + c] == hl ? gyp[24 + c] : 0.f) + t[12 + c] + u[c] : 0.f;
        if (live) {
            float4 *go = (float4 *)a.G + edge;
            const size_t ps = (size_t)a.plane;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                go[q * ps] = make_float4(g2[4 * q], g2[4 * q + 1], g2[4 * q + 2], g2[4 * q + 3]);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                go[(3 + q) * ps] = make_float4(g1[4 * q], g1[4 * q + 1], g1[4 * q + 2], g1[4 * q + 3]);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                go[(6 + q) * ps] = make_float4(g0[4 * q], g0[4 * q + 1], g0[4 * q + 2], g0[4 * q + 3]);
        }
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
            float *nbs = &s.nb[half][hl][0];
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
                atomicAdd(a.gx + (long)s.nbrow[half][e] * DT_C + hl, s.nb[half][e][hl]);
        }
        // S = sum over the point's edges of [g_2 | g_1 | g_0]; every lane of the half ends with all 36 in `st`
        __builtin_amdgcn_wave_barrier();                            // (`st` still holds the forward terms: all read by now)
#pragma unroll
        for (int c = 0; c < 36; ++c) {
            const float v = dt_half_sum(c < 12 ? g2[c] : (c < 24 ? g1[c - 12] : g0[c - 24]));
            if (hl == 0)
                st[c] = v;
        }
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
        __builtin_amdgcn_wave_barrier();
    }
}

int dt_check(long p, int n, int k, int idx_stride, int idx_off)
{
    if (p < 0 || n <= 0 || idx_stride <= 0 || idx_off < 0) return TPU3_EINVAL;
    if (k != DT_K || idx_off + k > idx_stride) return TPU3_ELIMIT;
    return TPU3_OK;
}

// One workgroup per compute-unit slot, walking its passes (r4).  The backward kernel holds 256 registers per lane --
// ONE workgroup per compute unit -- and a training batch is ~1250 passes of 8 points: launched as 1248 workgroups of one
// pass each (the cap used to be 2048) it ran five rounds of (weights into LDS, ~10 dependent load -> LDS-store round
// trips; then one pass) with nothing to hide either behind: 112 us per launch.  `per_cu` = workgroups a compute unit
// holds (backward 1, forward 2 at 188 registers); TPU3_DT_GRID overrides the cap (tuning hook).
} // namespace

// float4 entries between consecutive planes of G / Z: the edge count rounded up to 64 plus 17 -- with the bare count
// (a multiple of 2^13 for the training batch) the same edge of all 21 planes fell into the same memory channel
extern "C" long tpu3_dec_train_plane_stride(long points)
{
    const long e = points * DT_K;
    return (e + 63) / 64 * 64 + 17;
}

namespace {

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
    DtArgs a{p * n, n, idx_stride, idx_off, x, idx, w0, b0, w1, b1, w2, b2, y, arg, nullptr, nullptr, nullptr, nullptr,
             nullptr, 0};
    hipLaunchKernelGGL(dec_train_fwd_kernel, dim3(dt_grid(a.points, 2)), dim3(DT_THREADS), 0, (hipStream_t)stream, a);
    return tpu3_launch_status();
}

extern "C" int tpu3_dec_train_bwd_f32(tpu3_stream_t stream, long p, int n, int k, const float *x, const int32_t *idx,
                                      int idx_stride, int idx_off, const float *w0, const float *b0, const float *w1,
                                      const float *b1, const float *w2, const float *b2, const uint8_t *arg,
                                      const float *gy, float *gx, float *G, float *Z, float *S)
{
    const int r = dt_check(p, n, k, idx_stride, idx_off);
    if (r) return r;
    if (p == 0) return TPU3_OK;
    if (!x || !idx || !w0 || !b0 || !w1 || !b1 || !w2 || !b2 || !arg || !gy || !gx || !G || !Z || !S) return TPU3_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)G | (uintptr_t)Z | (uintptr_t)S) & 15) != 0) return TPU3_ELIMIT;
    DtArgs a{p * n, n, idx_stride, idx_off, x, idx, w0, b0, w1, b1, w2, b2, nullptr, const_cast<uint8_t *>(arg), gy, gx,
             G, Z, S, tpu3_dec_train_plane_stride(p * n)};
    hipLaunchKernelGGL(dec_train_bwd_kernel, dim3(dt_grid(a.points, 2)), dim3(DT_THREADS), 0, (hipStream_t)stream, a);
    return tpu3_launch_status();
}
