// skip.hip -- fused inter-level skip connection of a Level (inference) for gfx950.
//
// Replaces network/upsampler.py:317-347 of the reference: for every point of a patch, its K nearest
// points of the previous level's merged cloud (indices from the kNN kernel) contribute their
// features with bilateral weights
//     w_k  = exp(-|p_i - q_k|^2 / (h_s/2)) * exp(-|x_i - f_k|^2 / (h_f/2)),   h = mean_i min_k dist
//     w_k /= sum_k (w_k + 1e-5),        x_i += 0.2 * sum_k w_k f_k
// The reference gathers a (B,264,N,K) tensor (25 GB for the level-4 patches of 8 clouds), reduces it
// twice and re-reads it for the weighted sum through ~20 ATen kernels.  Here nothing of that size is
// materialised; two kernels read the K neighbour rows of each point straight from the previous
// level's feature table:
//   skip_dist_kernel   spatial and feature distances of every (point, neighbour) pair and their
//                      minima per point -> a small scratch (2K+2 floats per point);
//   skip_apply_kernel  h_s, h_f of the patch from the minima (fixed summation order), the weights, and
//                      x_i += 0.2 * sum_k w_k f_k in place in the level's feature buffer.
// The kernels are bound by where the gathered rows come from.  A wave owns a point (lanes across the
// 264 channels as float4, the K rows in flight together); a workgroup owns a SLICE of a patch's
// points, and workgroups are ordered so that each XCD works through one previous cloud at a time:
// its de-duplicated feature table (~1.5 MB) then stays in that XCD's 4 MB L2 while the level's own
// feature rows stream past it with non-temporal loads and stores.  (One workgroup per patch kept
// 6 clouds in flight per XCD and missed L2 on ~45 % of the requests: 4.8 TB/s of fabric traffic.)
#include "tpu3_dev.h"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

constexpr int SK_THREADS = 256;
constexpr int SK_KMAX = 8;
constexpr int SK_CPL = 5;            // channels per lane: C <= 320

struct SkipArgs {
    int n, k, c;
    int feat_stride;                 // row stride of feat (floats)
    int m;                           // rows of the previous cloud slab
    const float *xyz;                // (B,n,3)
    void *feat;                      // (B,n,feat_stride) in/out, first c channels; fp32 or (store = f16) fp16 rows
    const float *prev_xyz;           // (Bp,m,3)
    const void *prev_feat;           // (Bp,m,c), same element type as feat
    const int32_t *pts_of;           // (B) or null
    const void *idx;                 // (B,n,k)
    int idx64;
    float scale;                     // 0.2
    int per_cloud;                   // patches per previous cloud when they are contiguous, else 0
    int remap_blocks;                // blocks covered by the XCD-aware mapping (a multiple of 8 clouds)
    int slices, slice_len;           // workgroups per patch, points per workgroup
    float *dist;                     // scratch (B,n,2K): spatial then feature distances
    float *mins;                     // scratch (B,n,2): min_k of either
    float *wout;                     // training: (B,n,K) receives the normalised weights, else null
    int debug_phase;                 // (TPU3_DEBUG_SKIP_PHASE, timing probes only: 1 = the distance phase alone, 2 = the update alone)
};

// workgroup -> (patch, slice).  Workgroups are dealt round-robin to the 8 XCDs, each with its own
// L2; block i (XCD i % 8) takes cloud (i % 8) + 8 * (slot / items-per-cloud), so all slices of all
// patches of a previous cloud run on one XCD, one cloud after the other.
__device__ __forceinline__ void skip_item(const SkipArgs &a, int &b, int &slice)
{
    int id = blockIdx.x;
    if (id < a.remap_blocks) {
        const int per = a.per_cloud * a.slices;
        const int x = id & 7, slot = id >> 3;
        const int cl = slot / per;
        id = (x + 8 * cl) * per + (slot - cl * per);
    }
    b = id / a.slices;
    slice = id - b * a.slices;
}

typedef float sk_f4 __attribute__((ext_vector_type(4)));
typedef _Float16 sk_h4 __attribute__((ext_vector_type(4)));

// four channels of a feature row: fp32 rows as they are, fp16 rows (activation storage mode f16) widened on load
// and rounded on store -- the arithmetic in between is the same fp32 code
template <bool NT>
__device__ __forceinline__ sk_f4 sk_ld4(const float *row, int i4)
{
    return NT ? __builtin_nontemporal_load((const sk_f4 *)row + i4) : ((const sk_f4 *)row)[i4];
}
template <bool NT>
__device__ __forceinline__ sk_f4 sk_ld4(const _Float16 *row, int i4)
{
    const sk_h4 h = NT ? __builtin_nontemporal_load((const sk_h4 *)row + i4) : ((const sk_h4 *)row)[i4];
    return (sk_f4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
__device__ __forceinline__ void sk_st4(float *row, int i4, sk_f4 v) { __builtin_nontemporal_store(v, (sk_f4 *)row + i4); }
__device__ __forceinline__ void sk_st4(_Float16 *row, int i4, sk_f4 v)
{
    const sk_h4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    __builtin_nontemporal_store(h, (sk_h4 *)row + i4);
}
#define SK_DPP(V, CTRL, RM) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(V), CTRL, RM, 0xF, false))
// sum over the wave by DPP (no LDS traffic); every lane of the LAST row ends with the total
__device__ __forceinline__ float sk_wave_sum(float v)
{
    v += SK_DPP(v, 0xB1, 0xF);      // quad_perm [1,0,3,2]
    v += SK_DPP(v, 0x4E, 0xF);      // quad_perm [2,3,0,1]
    v += SK_DPP(v, 0x141, 0xF);     // row_half_mirror
    v += SK_DPP(v, 0x140, 0xF);     // row_mirror: every lane holds its row's sum
    v += SK_DPP(v, 0x142, 0xA);     // row_bcast15 -> rows 1, 3
    v += SK_DPP(v, 0x143, 0xC);     // row_bcast31 -> rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// sum / minimum over each aligned group of 8 lanes (fixed order), result in all 8
__device__ __forceinline__ float sk_group8_sum(float v)
{
    v += SK_DPP(v, 0xB1, 0xF);
    v += SK_DPP(v, 0x4E, 0xF);
    v += SK_DPP(v, 0x141, 0xF);
    return v;
}

__device__ __forceinline__ float sk_group8_min(float v)
{
    v = fminf(v, SK_DPP(v, 0xB1, 0xF));
    v = fminf(v, SK_DPP(v, 0x4E, 0xF));
    v = fminf(v, SK_DPP(v, 0x141, 0xF));
    return v;
}

// What a wave does per point is bound by the NUMBER of vector-memory instructions it issues (each row-wide
// float4 load keeps the CU's texture path busy for 16 cycles): K row loads + the point's own row, and ONE
// masked instruction of the first 16 lanes for everything small --
//   lanes 0 .. K-1     ("group 0", lane kk = neighbour kk): the neighbour's xyz (dist) / its two distances (apply)
//   lanes 8 .. 15      the point's own xyz (dist)
//   lane 8 j + kk      TAIL (64 < C/4 <= 72): float4 64 + j of neighbour kk's row and of the point's own row --
//                      the 8 channels beyond 256 of all K rows in one load (a second row-wide load per neighbour
//                      would carry 2 useful lanes of 64 at C = 264).
// K is a template parameter so that the K row loads of a point are ALL issued before the first one is consumed.
// VEC: rows are read as float4 (C % 4 == 0, C <= 288, 16-byte aligned slabs); otherwise scalar channel loops.
template <int K>
struct SkLanes {
    int tk, tjr, tj;        // lane = 8 tjr + tk; tj = tjr clamped to the tail's float4s
    bool small, tj_live, tact;
    __device__ __forceinline__ SkLanes(int lane, int C4, bool tail)
    {
        tk = lane & 7;
        tjr = lane >> 3;
        const int t = tail ? C4 - 64 : 0;
        tj = tail ? min(tjr, t - 1) : 0;
        tj_live = tjr < t;
        tact = tj_live && tk < K;
        small = lane < max(16, 8 * t);
    }
};

template <int K>
__device__ __forceinline__ void sk_neighbours(const SkipArgs &a, const void *idx, size_t io, int (&nbr)[K])
{
    if (a.idx64) {
#pragma unroll
        for (int kk = 0; kk < K; ++kk)
            nbr[kk] = (int)((const long long *)idx)[io + kk];
    } else {
#pragma unroll
        for (int kk = 0; kk < K; ++kk)
            nbr[kk] = ((const int *)idx)[io + kk];
    }
#pragma unroll
    for (int kk = 0; kk < K; ++kk)
        nbr[kk] = min(max(nbr[kk], 0), a.m - 1);
}

// the wave-uniform rows as a per-lane value: lane 8 j + kk -> neighbour kk
template <int K>
__device__ __forceinline__ int sk_lane_neighbour(const int (&nbr)[K], int tk)
{
    // (readfirstlane: the rows are wave-uniform anyway, and it keeps the compiler from re-forming the select chain
    // into a dynamically indexed array, which it then places in LDS)
    int nb = __builtin_amdgcn_readfirstlane(nbr[K - 1]);
#pragma unroll
    for (int kk = K - 2; kk >= 0; --kk) {
        const int t = __builtin_amdgcn_readfirstlane(nbr[kk]);
        nb = tk == kk ? t : nb;
    }
    return nb;
}

// (the bodies take their pointers as __restrict__ parameters: once inlined, the index loads are known not to be
// clobbered by the kernel's stores and stay scalar loads)
// (THREADS / FUSED: skip_fused_kernel -- a workgroup of THREADS owns a whole patch, the distance scratch is its LDS)
template <int K, bool VEC, bool TAIL, typename T, int THREADS = SK_THREADS, bool FUSED = false>
__device__ __forceinline__ void skip_dist_body(const SkipArgs &a, const void *__restrict__ idx_p,
                                               const T *__restrict__ feat_p, float *__restrict__ dist_p,
                                               float *__restrict__ mins_p)
{
    const int n = a.n, C = a.c;
    int b, slice;
    skip_item(a, b, slice);
    const int pb = a.pts_of ? a.pts_of[b] : b;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // i, the neighbour rows: SGPRs
    const int i_lo = slice * a.slice_len, i_hi = min(n, i_lo + a.slice_len);
    const float *XYZ = a.xyz + (size_t)b * n * 3;
    const T *F = feat_p + (size_t)b * n * a.feat_stride;
    const float *PX = a.prev_xyz + (size_t)pb * a.m * 3;
    const T *PF = (const T *)a.prev_feat + (size_t)pb * a.m * C;
    float2 *DS = (float2 *)dist_p + (FUSED ? 0 : (size_t)b * n * K);        // (spatial, feature) per neighbour
    float2 *MN = (float2 *)mins_p + (FUSED ? 0 : (size_t)b * n);
    const int C4 = C >> 2;
    const bool v0 = lane < C4;
    // loads are unconditional from clamped slots (a select on the result, not a branch around the load)
    const int l0 = VEC ? min(lane, C4 - 1) : 0;
    const SkLanes<K> L(lane, C4, TAIL);

    // The point's own row streams from HBM (the longest wait of an iteration) and its neighbour list is a scalar
    // load the row addresses depend on: both are fetched ONE ITERATION AHEAD (8 registers), so an iteration waits
    // for L2 (the gathered rows) only.
    constexpr int STEP = THREADS / 64;
    int nbr[K];
    sk_f4 x0 = {0.f, 0.f, 0.f, 0.f}, xt = {0.f, 0.f, 0.f, 0.f};
    if (i_lo + wave < i_hi) {
        const int i = i_lo + wave;
        sk_neighbours<K>(a, idx_p, ((size_t)b * n + i) * K, nbr);
        if (VEC) {
            const T *X4 = F + (size_t)i * a.feat_stride;
            x0 = sk_ld4<!FUSED>(X4, l0);
            if (TAIL && L.small)
                xt = sk_ld4<!FUSED>(X4, 64 + L.tj);
        }
    }
    for (int i = i_lo + wave; i < i_hi; i += STEP) {
        const int nb = sk_lane_neighbour<K>(nbr, L.tk);
        const int inx = i + STEP < i_hi ? i + STEP : i;         // next point of this wave (or this one again)
        int nbrn[K];
        sk_f4 x0n = {0.f, 0.f, 0.f, 0.f}, xtn = {0.f, 0.f, 0.f, 0.f};
        float p0 = 0.f, p1 = 0.f, p2 = 0.f;
        sk_f4 rt = {0.f, 0.f, 0.f, 0.f};
        float acc[K];
        if (VEC) {
            sk_f4 r[K];
#pragma unroll
            for (int kk = 0; kk < K; ++kk)
                r[kk] = sk_ld4<false>(PF + (size_t)nbr[kk] * C, l0);
            if (L.small) {
                const float *pp = L.tjr == 0 ? PX + (size_t)nb * 3 : XYZ + (size_t)i * 3;
                p0 = pp[0];
                p1 = pp[1];
                p2 = pp[2];
                if (TAIL)
                    rt = sk_ld4<false>(PF + (size_t)nb * C, 64 + L.tj);
            }
            {
                sk_neighbours<K>(a, idx_p, ((size_t)b * n + inx) * K, nbrn);
                const T *X4n = F + (size_t)inx * a.feat_stride;
                x0n = sk_ld4<!FUSED>(X4n, l0);
                if (TAIL && L.small)
                    xtn = sk_ld4<!FUSED>(X4n, 64 + L.tj);
            }
            float ts = 0.f;
            if (TAIL) {
                float d;
                d = xt.x - rt.x; ts = __builtin_fmaf(d, d, ts);
                d = xt.y - rt.y; ts = __builtin_fmaf(d, d, ts);
                d = xt.z - rt.z; ts = __builtin_fmaf(d, d, ts);
                d = xt.w - rt.w; ts = __builtin_fmaf(d, d, ts);
            }
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                float s = 0.f, d;
                d = x0.x - r[kk].x; s = __builtin_fmaf(d, d, s);
                d = x0.y - r[kk].y; s = __builtin_fmaf(d, d, s);
                d = x0.z - r[kk].z; s = __builtin_fmaf(d, d, s);
                d = x0.w - r[kk].w; s = __builtin_fmaf(d, d, s);
                acc[kk] = v0 ? s : 0.f;
                if (TAIL)
                    acc[kk] += L.tact && L.tk == kk ? ts : 0.f;
            }
        } else {
            sk_neighbours<K>(a, idx_p, ((size_t)b * n + inx) * K, nbrn);
            if (L.small) {
                const float *pp = L.tjr == 0 ? PX + (size_t)nb * 3 : XYZ + (size_t)i * 3;
                p0 = pp[0];
                p1 = pp[1];
                p2 = pp[2];
            }
            float xv[SK_CPL];
#pragma unroll
            for (int u = 0; u < SK_CPL; ++u) {
                const int c = lane + 64 * u;
                xv[u] = c < C ? (float)F[(size_t)i * a.feat_stride + c] : 0.f;
            }
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                acc[kk] = 0.f;
                const T *row = PF + (size_t)nbr[kk] * C;
#pragma unroll
                for (int u = 0; u < SK_CPL; ++u) {
                    const int c = lane + 64 * u;
                    const float d = c < C ? xv[u] - (float)row[c] : 0.f;
                    acc[kk] += d * d;
                }
            }
        }
        // spatial distance in lane kk: (dx^2 + dy^2) + dz^2, like torch.sum over 3 channels
        const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p0), 8));
        const float qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p1), 8));
        const float qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p2), 8));
        const float dx = qx - p0, dy = qy - p1, dz = qz - p2;
        const float sp = (dx * dx + dy * dy) + dz * dz;
        const float smin_ = sk_group8_min(lane < K ? sp : __builtin_inff());
        float fmin_ = 0.f, mine_f = 0.f;
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const float f = sk_wave_sum(acc[kk]);
            fmin_ = kk == 0 ? f : fminf(fmin_, f);
            mine_f = lane == kk ? f : mine_f;
        }
        if (lane < K)
            DS[(size_t)i * K + lane] = make_float2(sp, mine_f);
        if (lane == 0)
            MN[i] = make_float2(smin_, fmin_);
#pragma unroll
        for (int kk = 0; kk < K; ++kk)
            nbr[kk] = nbrn[kk];
        x0 = x0n;
        xt = xtn;
    }
}

template <int K, bool VEC, bool TAIL, typename T = float>
__global__ __launch_bounds__(SK_THREADS) void skip_dist_kernel(SkipArgs a)
{
    skip_dist_body<K, VEC, TAIL, T>(a, a.idx, (const T *)a.feat, a.dist, a.mins);
}

template <int K, bool VEC, bool TAIL, typename T, int THREADS = SK_THREADS, bool FUSED = false>
__device__ __forceinline__ void skip_apply_body(const SkipArgs &a, const void *__restrict__ idx_p,
                                                T *__restrict__ feat_p, const T *__restrict__ prev_p,
                                                const float *__restrict__ dist_p, const float *__restrict__ mins_p)
{
    const int n = a.n, C = a.c;
    int b, slice;
    skip_item(a, b, slice);
    const int pb = a.pts_of ? a.pts_of[b] : b;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // i, the neighbour rows: SGPRs
    const int i_lo = slice * a.slice_len, i_hi = min(n, i_lo + a.slice_len);
    T *F = feat_p + (size_t)b * n * a.feat_stride;
    const T *PF = prev_p + (size_t)pb * a.m * C;
    const float2 *DS = (const float2 *)dist_p + (FUSED ? 0 : (size_t)b * n * K);
    const float2 *MN = (const float2 *)mins_p + (FUSED ? 0 : (size_t)b * n);
    // ---- h = mean over the patch's points of the distance to the closest of the K neighbours; every wave of
    // every slice of a patch sums the same values in the same order (no barrier, no LDS) -----------------------
    float ms = 0.f, mf = 0.f;
    for (int i = lane; i < n; i += 64) {
        const float2 v = MN[i];
        ms += v.x;
        mf += v.y;
    }
    const float hs = sk_wave_sum(ms) / (float)n;
    const float hf = sk_wave_sum(mf) / (float)n;
    const float hs2 = hs / 2, hf2 = hf / 2;
    const int C4 = C >> 2;
    const bool v0 = lane < C4;
    const int l0 = VEC ? min(lane, C4 - 1) : 0;
    const SkLanes<K> L(lane, C4, TAIL);
    const int lk = min(L.tk, K - 1);                // (lanes 0 .. K-1: their own neighbour; K <= 8)

    // (own row and neighbour list one iteration ahead, as in skip_dist_kernel)
    constexpr int STEP = THREADS / 64;
    int nbr[K];
    sk_f4 x0 = {0.f, 0.f, 0.f, 0.f}, xt = {0.f, 0.f, 0.f, 0.f};
    if (i_lo + wave < i_hi) {
        const int i = i_lo + wave;
        sk_neighbours<K>(a, idx_p, ((size_t)b * n + i) * K, nbr);
        if (VEC) {
            const T *X4 = F + (size_t)i * a.feat_stride;
            x0 = sk_ld4<true>(X4, l0);
            if (TAIL && L.small)
                xt = sk_ld4<true>(X4, 64 + L.tj);
        }
    }
    for (int i = i_lo + wave; i < i_hi; i += STEP) {
        const int nb = sk_lane_neighbour<K>(nbr, L.tk);
        const int inx = i + STEP < i_hi ? i + STEP : i;
        int nbrn[K];
        sk_f4 x0n = {0.f, 0.f, 0.f, 0.f}, xtn = {0.f, 0.f, 0.f, 0.f};
        T *X4 = F + (size_t)i * a.feat_stride;
        sk_f4 r[K];
        if (VEC) {
#pragma unroll
            for (int kk = 0; kk < K; ++kk)
                r[kk] = sk_ld4<false>(PF + (size_t)nbr[kk] * C, l0);
        }
        float2 ds = make_float2(0.f, 0.f);
        sk_f4 rt = {0.f, 0.f, 0.f, 0.f};
        if (L.small) {
            ds = DS[(size_t)i * K + lk];
            if (VEC && TAIL)
                rt = sk_ld4<false>(PF + (size_t)nb * C, 64 + L.tj);
        }
        sk_neighbours<K>(a, idx_p, ((size_t)b * n + inx) * K, nbrn);
        if (VEC && inx != i) {      // (the row of `i` itself is about to be rewritten: never re-read it)
            const T *X4n = F + (size_t)inx * a.feat_stride;
            x0n = sk_ld4<true>(X4n, l0);
            if (TAIL && L.small)
                xtn = sk_ld4<true>(X4n, 64 + L.tj);
        }
        // weights (reference :340-342): w = ws*wf;  w /= sum_k (w + 1e-5).  Lane 8 j + kk evaluates neighbour kk
        // (two divisions, two exponentials, then one more division) and group 0's results are broadcast -- the
        // same operations on the same values as evaluating all K in every lane, a fifth of the VALU work
        // (IEEE divisions and expf are ~10 instructions each).
        float w[K], tot = 0.f;
        const float mine = expf(-ds.x / hs2) * expf(-ds.y / hf2);
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            w[kk] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), kk));
            tot += w[kk] + 1e-5f;
        }
        const float mine_w = mine / tot;
        if (a.wout && lane < K)
            a.wout[((size_t)b * n + i) * K + lane] = mine_w;
#pragma unroll
        for (int kk = 0; kk < K; ++kk)
            w[kk] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine_w), kk));
        if (VEC) {
            sk_f4 s0;
            s0.x = w[0] * r[0].x; s0.y = w[0] * r[0].y; s0.z = w[0] * r[0].z; s0.w = w[0] * r[0].w;
#pragma unroll
            for (int kk = 1; kk < K; ++kk) {
                s0.x = __builtin_fmaf(w[kk], r[kk].x, s0.x); s0.y = __builtin_fmaf(w[kk], r[kk].y, s0.y);
                s0.z = __builtin_fmaf(w[kk], r[kk].z, s0.z); s0.w = __builtin_fmaf(w[kk], r[kk].w, s0.w);
            }
            s0.x = __builtin_fmaf(a.scale, s0.x, x0.x); s0.y = __builtin_fmaf(a.scale, s0.y, x0.y);
            s0.z = __builtin_fmaf(a.scale, s0.z, x0.z); s0.w = __builtin_fmaf(a.scale, s0.w, x0.w);
            if (v0)
                sk_st4(X4, lane, s0);
            if (TAIL) {
                // lane 8 j + kk holds float4 64 + j of neighbour kk: weight it, add up the eight lanes of the group
                const float wt = L.tact ? mine_w : 0.f;
                sk_f4 t;
                t.x = sk_group8_sum(wt * rt.x); t.y = sk_group8_sum(wt * rt.y);
                t.z = sk_group8_sum(wt * rt.z); t.w = sk_group8_sum(wt * rt.w);
                t.x = __builtin_fmaf(a.scale, t.x, xt.x); t.y = __builtin_fmaf(a.scale, t.y, xt.y);
                t.z = __builtin_fmaf(a.scale, t.z, xt.z); t.w = __builtin_fmaf(a.scale, t.w, xt.w);
                if (L.tk == 0 && L.tj_live)
                    sk_st4(X4, 64 + L.tj, t);
            }
        } else {
            float s[SK_CPL];
#pragma unroll
            for (int u = 0; u < SK_CPL; ++u)
                s[u] = 0.f;
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                const T *row = PF + (size_t)nbr[kk] * C;
#pragma unroll
                for (int u = 0; u < SK_CPL; ++u) {
                    const int c = lane + 64 * u;
                    if (c < C)
                        s[u] = kk == 0 ? w[kk] * (float)row[c] : s[u] + w[kk] * (float)row[c];
                }
            }
#pragma unroll
            for (int u = 0; u < SK_CPL; ++u) {
                const int c = lane + 64 * u;
                if (c < C) {
                    T *p = F + (size_t)i * a.feat_stride + c;
                    *p = (T)(a.scale * s[u] + (float)*p);
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < K; ++kk)
            nbr[kk] = nbrn[kk];
        x0 = x0n;
        xt = xtn;
    }
}

template <int K, bool VEC, bool TAIL, typename T = float>
__global__ __launch_bounds__(SK_THREADS) void skip_apply_kernel(SkipArgs a)
{
    skip_apply_body<K, VEC, TAIL, T>(a, a.idx, (T *)a.feat, (const T *)a.prev_feat, a.dist, a.mins);
}

// (r5) ONE launch, a 16-wave workgroup per patch: distances and minima of all its points into LDS, a barrier, then the
// weights and the update.  The own rows are still read twice, but the second read follows the first by half a
// workgroup's life instead of a whole launch (1.3 GB of other rows): with ONE patch in flight per compute unit (84 MB
// on the device) it is served by the Infinity Cache, the (2K+2)-float scratch per point never reaches memory (every
// wave sums h from LDS) -- the same operations in the same order as the two kernels, bit for bit.  Measured on a
// 3840-patch chunk: two kernels 1.49 ms; fused with 4 / 8 / 12 / 16 waves per workgroup (8 / 4 / 2 / 1 patches per
// compute unit) 1.45 / 1.34 / 1.34 / 1.26 ms; 16 waves held to 64 registers (two patches per compute unit) 1.43 ms;
// with 137 of a patch's 312 own rows kept in the workgroup's LDS for the second phase 1.27 ms (no gain: dropped).
constexpr int SKF_THREADS = 1024;

template <int K, bool VEC, bool TAIL, typename T = float>
__global__ __launch_bounds__(SKF_THREADS) void skip_fused_kernel(SkipArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float sk_sm[];
    float *ds = sk_sm, *mn = sk_sm + (size_t)a.n * K * 2;
    if (a.debug_phase != 2)
        skip_dist_body<K, VEC, TAIL, T, SKF_THREADS, true>(a, a.idx, (const T *)a.feat, ds, mn);
    __syncthreads();
    if (a.debug_phase != 1)
        skip_apply_body<K, VEC, TAIL, T, SKF_THREADS, true>(a, a.idx, (T *)a.feat, (const T *)a.prev_feat, ds, mn);
}

// (r4, measured and removed -- commit cc6a076 has the kernels): a DPP ROW of 16 lanes per point, four points per wave
// step (a lane holds 4 + 1 float4 of each row; reductions are 4 DPP steps inside the rows for four points at once, lane
// k of a row evaluates neighbour k's spatial distance and weight, row_newbcast hands them round; loads stay coalesced,
// 256 contiguous bytes per point and instruction).  About half the wave instructions per point, within 1e-5 of the
// unfused formulation -- and SLOWER: 1.62 ms per 3840-patch chunk with the next neighbour's row in flight (106 / 136
// VGPRs, 4 / 3 waves per SIMD), 1.57 ms single-buffered (86 / 116 VGPRs), against 1.52 ms here.  These kernels are not
// bound by their instruction count: what limits them is the vector-memory path (K + 1 rows of 1056 B per point and
// kernel from L2 / HBM) and the waves available to hide it.

int g_skip_fused = getenv("TPU3_SKIP_FUSED") ? atoi(getenv("TPU3_SKIP_FUSED")) != 0 : 1;

template <int K>
int skip_launch(hipStream_t s, int blocks, const SkipArgs &a, bool vec, bool half)
{
    // inference, float4 lanes, the patch's scratch in LDS: the one-launch form (TPU3_SKIP_FUSED=0 /
    // tpu3_debug_skip_fused(0): the two kernels); fp32 and fp16 rows alike
    const size_t lds = (size_t)a.n * (K + 1) * 8;
    if (g_skip_fused && (vec || half) && !a.wout && lds <= 64 * 1024) {
        SkipArgs f = a;
        static const int dbg_phase = getenv("TPU3_DEBUG_SKIP_PHASE") ? atoi(getenv("TPU3_DEBUG_SKIP_PHASE")) : 0;
        f.debug_phase = dbg_phase;
        const int patches = blocks / a.slices;
        f.slices = 1;
        f.slice_len = a.n;
        f.remap_blocks = a.per_cloud ? (patches / a.per_cloud / 8) * 8 * a.per_cloud : 0;
        const dim3 g(patches), t(SKF_THREADS);
        if (half && a.c > 256)
            hipLaunchKernelGGL((skip_fused_kernel<K, true, true, _Float16>), g, t, lds, s, f);
        else if (half)
            hipLaunchKernelGGL((skip_fused_kernel<K, true, false, _Float16>), g, t, lds, s, f);
        else if (a.c > 256)
            hipLaunchKernelGGL((skip_fused_kernel<K, true, true>), g, t, lds, s, f);
        else
            hipLaunchKernelGGL((skip_fused_kernel<K, true, false>), g, t, lds, s, f);
        return tpu3_launch_status();
    }
    if (half) {         // fp16 rows: the float4-per-lane forms only (the caller checked the alignment)
        if (a.c > 256) {
            hipLaunchKernelGGL((skip_dist_kernel<K, true, true, _Float16>), dim3(blocks), dim3(SK_THREADS), 0, s, a);
            hipLaunchKernelGGL((skip_apply_kernel<K, true, true, _Float16>), dim3(blocks), dim3(SK_THREADS), 0, s, a);
        } else {
            hipLaunchKernelGGL((skip_dist_kernel<K, true, false, _Float16>), dim3(blocks), dim3(SK_THREADS), 0, s, a);
            hipLaunchKernelGGL((skip_apply_kernel<K, true, false, _Float16>), dim3(blocks), dim3(SK_THREADS), 0, s, a);
        }
        return tpu3_launch_status();
    }
    if (vec && a.c > 256) {
        hipLaunchKernelGGL((skip_dist_kernel<K, true, true>), dim3(blocks), dim3(SK_THREADS), 0, s, a);
        hipLaunchKernelGGL((skip_apply_kernel<K, true, true>), dim3(blocks), dim3(SK_THREADS), 0, s, a);
    } else if (vec) {
        hipLaunchKernelGGL((skip_dist_kernel<K, true, false>), dim3(blocks), dim3(SK_THREADS), 0, s, a);
        hipLaunchKernelGGL((skip_apply_kernel<K, true, false>), dim3(blocks), dim3(SK_THREADS), 0, s, a);
    } else {
        hipLaunchKernelGGL((skip_dist_kernel<K, false, false>), dim3(blocks), dim3(SK_THREADS), 0, s, a);
        hipLaunchKernelGGL((skip_apply_kernel<K, false, false>), dim3(blocks), dim3(SK_THREADS), 0, s, a);
    }
    return tpu3_launch_status();
}

constexpr int SK_SLICE_POINTS = 52;      // 312-point patches: 6 workgroups of 13 points per wave

// Training, backward.  The reference detaches both distances (network/upsampler.py:244-245), so the weights are
// constants of the step and x_out = x + scale * sum_k w_k f_k has two gradients: g itself for x, and for the previous
// level's features a scatter  gprev[nbr_k] += scale * w_k * g_i  -- a wave per point, lanes across the channels,
// hardware float atomics (the rows of a previous patch are hit ~5 times each, in no fixed order).
struct SkipBwdArgs {
    int n, k, c, m;
    long points;                     // b * n
    const float *g;                  // (b,n,c)
    const float *w;                  // (b,n,k)
    const int32_t *pts_of;           // (b) or null
    const void *idx;
    int idx64;
    float scale;
    float *gprev;                    // (bp,m,c), accumulated
};

__global__ __launch_bounds__(SK_THREADS) void skip_bwd_kernel(SkipBwdArgs a)
{
    const int lane = threadIdx.x & 63;
    const long p = (long)blockIdx.x * (SK_THREADS / 64) + (threadIdx.x >> 6);
    if (p >= a.points)
        return;
    const int b = (int)(p / a.n);
    const int pb = a.pts_of ? a.pts_of[b] : b;
    const float *G = a.g + (size_t)p * a.c;
    float gv[SK_CPL];
#pragma unroll
    for (int u = 0; u < SK_CPL; ++u)
        gv[u] = lane + 64 * u < a.c ? G[lane + 64 * u] : 0.f;
    for (int kk = 0; kk < a.k; ++kk) {
        const size_t e = (size_t)p * a.k + kk;
        const long nb = a.idx64 ? (long)((const int64_t *)a.idx)[e] : (long)((const int32_t *)a.idx)[e];
        const float w = a.scale * a.w[e];
        float *D = a.gprev + ((size_t)pb * a.m + nb) * a.c;
#pragma unroll
        for (int u = 0; u < SK_CPL; ++u)
            if (lane + 64 * u < a.c)
                atomicAdd(D + lane + 64 * u, w * gv[u]);
    }
}

int skip_forward(hipStream_t s, int b, int n, int k, int c, const float *xyz, void *feat, int feat_stride,
                 const float *prev_xyz, const void *prev_feat, int m, const int32_t *pts_of, const void *idx,
                 int idx_elem_size, float scale, int patches_per_cloud, void *workspace, size_t workspace_bytes,
                 float *wout, int store = TPU3_STORE_F32);

} // namespace

extern "C" size_t tpu3_interlevel_skip_workspace_bytes(int b, int n, int k)
{
    if (b <= 0 || n <= 0 || k <= 0)
        return 0;
    return (size_t)b * n * (2 * (size_t)k + 2) * sizeof(float);
}

extern "C" int tpu3_debug_skip_fused(int on)
{
    const int old = g_skip_fused;
    g_skip_fused = on != 0;
    return old;
}

extern "C" int tpu3_interlevel_skip_f32(tpu3_stream_t stream, int b, int n, int k, int c, const float *xyz,
                                        float *feat, int feat_stride, const float *prev_xyz,
                                        const float *prev_feat, int m, const int32_t *pts_of, const void *idx,
                                        int idx_elem_size, float scale, int patches_per_cloud, void *workspace,
                                        size_t workspace_bytes)
{
    return skip_forward((hipStream_t)stream, b, n, k, c, xyz, feat, feat_stride, prev_xyz, prev_feat, m, pts_of, idx,
                        idx_elem_size, scale, patches_per_cloud, workspace, workspace_bytes, nullptr);
}

extern "C" int tpu3_interlevel_skip_st_f32(tpu3_stream_t stream, int b, int n, int k, int c, const float *xyz,
                                           void *feat, int feat_stride, const float *prev_xyz, const void *prev_feat,
                                           int m, const int32_t *pts_of, const void *idx, int idx_elem_size,
                                           float scale, int patches_per_cloud, void *workspace,
                                           size_t workspace_bytes, int store)
{
    if (store != TPU3_STORE_F32 && store != TPU3_STORE_F16) return TPU3_EINVAL;
    return skip_forward((hipStream_t)stream, b, n, k, c, xyz, feat, feat_stride, prev_xyz, prev_feat, m, pts_of, idx,
                        idx_elem_size, scale, patches_per_cloud, workspace, workspace_bytes, nullptr, store);
}

extern "C" int tpu3_interlevel_skip_train_f32(tpu3_stream_t stream, int b, int n, int k, int c, const float *xyz,
                                              float *feat, int feat_stride, const float *prev_xyz,
                                              const float *prev_feat, int m, const int32_t *pts_of, const void *idx,
                                              int idx_elem_size, float scale, float *weights, void *workspace,
                                              size_t workspace_bytes)
{
    if (!weights) return TPU3_EINVAL;
    return skip_forward((hipStream_t)stream, b, n, k, c, xyz, feat, feat_stride, prev_xyz, prev_feat, m, pts_of, idx,
                        idx_elem_size, scale, 0, workspace, workspace_bytes, weights);
}

extern "C" int tpu3_interlevel_skip_bwd_f32(tpu3_stream_t stream, int b, int n, int k, int c, const float *g,
                                            const float *weights, int m, const int32_t *pts_of, const void *idx,
                                            int idx_elem_size, float scale, float *gprev)
{
    if (b < 0 || n <= 0 || k <= 0 || k > SK_KMAX || c <= 0 || c > 64 * SK_CPL || m <= 0) return TPU3_EINVAL;
    if (idx_elem_size != 4 && idx_elem_size != 8) return TPU3_EINVAL;
    if (b == 0) return TPU3_OK;
    if (!g || !weights || !idx || !gprev) return TPU3_EINVAL;
    const long points = (long)b * n;
    const long blocks = (points + SK_THREADS / 64 - 1) / (SK_THREADS / 64);
    if (blocks > 0x7FFFFFFF) return TPU3_ELIMIT;
    SkipBwdArgs a{n, k, c, m, points, g, weights, pts_of, idx, idx_elem_size == 8, scale, gprev};
    hipLaunchKernelGGL(skip_bwd_kernel, dim3((unsigned)blocks), dim3(SK_THREADS), 0, (hipStream_t)stream, a);
    return tpu3_launch_status();
}

namespace {

int skip_forward(hipStream_t s, int b, int n, int k, int c, const float *xyz, void *feat, int feat_stride,
                 const float *prev_xyz, const void *prev_feat, int m, const int32_t *pts_of, const void *idx,
                 int idx_elem_size, float scale, int patches_per_cloud, void *workspace, size_t workspace_bytes,
                 float *wout, int store)
{
    if (b < 0 || n <= 0 || k <= 0 || k > SK_KMAX || c <= 0 || c > 64 * SK_CPL || m <= 0) return TPU3_EINVAL;
    if (feat_stride < c || (idx_elem_size != 4 && idx_elem_size != 8)) return TPU3_EINVAL;
    if (b == 0) return TPU3_OK;
    if (!xyz || !feat || !prev_xyz || !prev_feat || !idx) return TPU3_EINVAL;
    const size_t need = tpu3_interlevel_skip_workspace_bytes(b, n, k);
    void *own = nullptr;
    if (!workspace || workspace_bytes < need) {     // caller gave no scratch: stream-ordered allocation
        const hipError_t e = hipMallocAsync(&own, need, s);
        if (e != hipSuccess) return (int)e;
        workspace = own;
    }
    const int slices = (n + SK_SLICE_POINTS - 1) / SK_SLICE_POINTS;
    const int slice_len = (n + slices - 1) / slices;
    if ((long)b * slices > 0x7FFFFFFF) return TPU3_ELIMIT;
    const int per = patches_per_cloud > 0 && b % patches_per_cloud == 0 ? patches_per_cloud : 0;
    const int remap = per ? (b / per / 8) * 8 * per * slices : 0;
    float *dist = (float *)workspace;
    SkipArgs a{n, k, c, feat_stride, m, xyz, feat, prev_xyz, prev_feat, pts_of, idx, idx_elem_size == 8, scale,
               per, remap, slices, slice_len, dist, dist + (size_t)b * n * 2 * k, wout, 0};
    // float4 rows: every row start must be 16-byte aligned (fp16 rows: four channels = 8 bytes per lane)
    const bool half = store == TPU3_STORE_F16;
    const uintptr_t amask = half ? 7 : 15;
    const bool vec = c % 4 == 0 && c <= 288 && feat_stride % 4 == 0 && ((uintptr_t)feat & amask) == 0 &&
                     ((uintptr_t)prev_feat & amask) == 0;
    if (half && !vec)
        return TPU3_ELIMIT;
    const int blocks = b * slices;
    int r;
    switch (k) {
    case 1: r = skip_launch<1>(s, blocks, a, vec, half); break;
    case 2: r = skip_launch<2>(s, blocks, a, vec, half); break;
    case 3: r = skip_launch<3>(s, blocks, a, vec, half); break;
    case 4: r = skip_launch<4>(s, blocks, a, vec, half); break;
    case 5: r = skip_launch<5>(s, blocks, a, vec, half); break;
    case 6: r = skip_launch<6>(s, blocks, a, vec, half); break;
    case 7: r = skip_launch<7>(s, blocks, a, vec, half); break;
    default: r = skip_launch<8>(s, blocks, a, vec, half); break;
    }
    if (own) {
        const hipError_t e = hipFreeAsync(own, s);
        if (!r) r = (int)e;
    }
    return r;
}

} // namespace

