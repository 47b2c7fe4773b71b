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
//   G (edges, 36) = [g_2 | g_1 | g_0],  Z (edges, 48) = [h_1 | h_0 | d_j]      -> weight gradients of the edge
//                                                                                  parts (tpu3_linear_wgrad_f32)
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
    float *G, *Z, *S;               // (P*N*32, 36), (P*N*32, 48), (P*N, 36)
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

// forward chain of one edge: a0, a1 (pre-activations), h2
__device__ __forceinline__ void dt_edge_forward(const DtLds &s, const float *st, const float (&dj)[DT_C],
                                                float (&a0)[DT_G], float (&a1)[DT_G], float (&h2)[DT_G])
{
#pragma unroll
    for (int c = 0; c < DT_G; ++c) {
        float acc = st[24 + c];
#pragma unroll
        for (int d = 0; d < DT_C; ++d)
            acc = __builtin_fmaf(s.eh[c * 24 + d], dj[d], acc);
        a0[c] = acc;
        __builtin_amdgcn_sched_barrier(0);              // (a row of weight reads in flight at a time: registers)
    }
#pragma unroll
    for (int c = 0; c < DT_G; ++c) {
        float acc = st[12 + c];
#pragma unroll
        for (int d = 0; d < DT_G; ++d)
            acc = __builtin_fmaf(s.h1w[c * 12 + d], fmaxf(a0[d], 0.f), acc);
        a1[c] = acc;
    }
#pragma unroll
    for (int c = 0; c < DT_G; ++c) {
        float acc = st[c];
#pragma unroll
        for (int d = 0; d < DT_G; ++d)
            acc = __builtin_fmaf(s.h2w[c * 24 + d], fmaxf(a1[d], 0.f), acc);
#pragma unroll
        for (int d = 0; d < DT_G; ++d)
            acc = __builtin_fmaf(s.h2w[c * 24 + 12 + d], fmaxf(a0[d], 0.f), acc);
        h2[c] = acc;
        __builtin_amdgcn_sched_barrier(0);
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
        dt_edge_forward(s, st, dj, a0, a1, h2);
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
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        float a0[DT_G], a1[DT_G], h2[DT_G];
        dt_edge_forward(s, st, dj, a0, a1, h2);
        const long edge = pt * DT_K + hl;
        if (live) {                                                 // Z = [h1 | h0 | d_j]
            float4 *zo = (float4 *)(a.Z + edge * 48);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                zo[q] = make_float4(fmaxf(a1[4 * q], 0.f), fmaxf(a1[4 * q + 1], 0.f), fmaxf(a1[4 * q + 2], 0.f),
                                    fmaxf(a1[4 * q + 3], 0.f));
#pragma unroll
            for (int q = 0; q < 3; ++q)
                zo[3 + q] = make_float4(fmaxf(a0[4 * q], 0.f), fmaxf(a0[4 * q + 1], 0.f), fmaxf(a0[4 * q + 2], 0.f),
                                        fmaxf(a0[4 * q + 3], 0.f));
#pragma unroll
            for (int q = 0; q < 6; ++q)
                zo[6 + q] = make_float4(dj[4 * q], dj[4 * q + 1], dj[4 * q + 2], dj[4 * q + 3]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // incoming gradient of channel c goes to the edge that attained the maximum
        const float *gyp = a.gy + pt * 60;
        const uint8_t *ap = a.arg + pt * 36;
        float g2[DT_G], g1[DT_G], g0[DT_G];
#pragma unroll
        for (int c = 0; c < DT_G; ++c)
            g2[c] = (int)ap[c] == hl ? gyp[c] : 0.f;
        float t[24];                                                // W_2h^T g_2: gradient w.r.t. [h_1, h_0]
#pragma unroll
        for (int d = 0; d < 24; ++d)
            t[d] = 0.f;
#pragma unroll
        for (int c = 0; c < DT_G; ++c) {
#pragma unroll
            for (int d = 0; d < 24; ++d)
                t[d] = __builtin_fmaf(s.h2w[c * 24 + d], g2[c], t[d]);
            __builtin_amdgcn_sched_barrier(0);          // (six weight reads in flight, not all 72: registers)
        }
#pragma unroll
        for (int c = 0; c < DT_G; ++c)
            g1[c] = a1[c] > 0.f ? ((int)ap[12 + c] == hl ? gyp[12 + c] : 0.f) + t[c] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
        float u[DT_G];                                              // W_1h^T g_1: gradient w.r.t. h_0
#pragma unroll
        for (int d = 0; d < DT_G; ++d)
            u[d] = 0.f;
#pragma unroll
        for (int c = 0; c < DT_G; ++c) {
#pragma unroll
            for (int d = 0; d < DT_G; ++d)
                u[d] = __builtin_fmaf(s.h1w[c * 12 + d], g1[c], u[d]);
            if (c % 2 == 1)
                __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < DT_G; ++c)
            g0[c] = a0[c] > 0.f ? ((int)ap[24 + c] == hl ? gyp[24 + c] : 0.f) + t[12 + c] + u[c] : 0.f;
        if (live) {
            float4 *go = (float4 *)(a.G + edge * 36);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                go[q] = make_float4(g2[4 * q], g2[4 * q + 1], g2[4 * q + 2], g2[4 * q + 3]);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                go[3 + q] = make_float4(g1[4 * q], g1[4 * q + 1], g1[4 * q + 2], g1[4 * q + 3]);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                go[6 + q] = make_float4(g0[4 * q], g0[4 * q + 1], g0[4 * q + 2], g0[4 * q + 3]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // neighbour's share: W_0b^T g_0
        {
            float vb[DT_C];
#pragma unroll
            for (int d = 0; d < DT_C; ++d)
                vb[d] = 0.f;
#pragma unroll
            for (int c = 0; c < DT_G; ++c) {
#pragma unroll
                for (int d = 0; d < DT_C; ++d)
                    vb[d] = __builtin_fmaf(s.eh[c * 24 + d], g0[c], vb[d]);
                __builtin_amdgcn_sched_barrier(0);
            }
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
             nullptr};
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
             G, Z, S};
    hipLaunchKernelGGL(dec_train_bwd_kernel, dim3(dt_grid(a.points, 1)), dim3(DT_THREADS), 0, (hipStream_t)stream, a);
    return tpu3_launch_status();
}
