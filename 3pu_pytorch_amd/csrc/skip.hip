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

namespace {

constexpr int SK_THREADS = 256;
constexpr int SK_KMAX = 8;
constexpr int SK_CPL = 5;            // channels per lane: C <= 320

struct SkipArgs {
    int n, k, c;
    int feat_stride;                 // row stride of feat (floats)
    int m;                           // rows of the previous cloud slab
    const float *xyz;                // (B,n,3)
    float *feat;                     // (B,n,feat_stride) in/out, first c channels
    const float *prev_xyz;           // (Bp,m,3)
    const float *prev_feat;          // (Bp,m,c)
    const int32_t *pts_of;           // (B) or null
    const void *idx;                 // (B,n,k)
    int idx64;
    float scale;                     // 0.2
    int per_cloud;                   // patches per previous cloud when they are contiguous, else 0
    int remap_blocks;                // blocks covered by the XCD-aware mapping (a multiple of 8 clouds)
    int slices, slice_len;           // workgroups per patch, points per workgroup
    float *dist;                     // scratch (B,n,2K): spatial then feature distances
    float *mins;                     // scratch (B,n,2): min_k of either
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

__device__ __forceinline__ float block_sum256(float v, float *red)
{
    v = tpu3_wave_sum_f32(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

typedef float sk_f4 __attribute__((ext_vector_type(4)));
// sum over the wave by DPP (no LDS traffic); every lane of the LAST row ends with the total
__device__ __forceinline__ float sk_wave_sum(float v)
{
#define SK_DPP_ADD(CTRL, RM)                                                                                 \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, RM, 0xF, false))
    SK_DPP_ADD(0xB1, 0xF);      // quad_perm [1,0,3,2]
    SK_DPP_ADD(0x4E, 0xF);      // quad_perm [2,3,0,1]
    SK_DPP_ADD(0x141, 0xF);     // row_half_mirror
    SK_DPP_ADD(0x140, 0xF);     // row_mirror: every lane holds its row's sum
    SK_DPP_ADD(0x142, 0xA);     // row_bcast15 -> rows 1, 3
    SK_DPP_ADD(0x143, 0xC);     // row_bcast31 -> rows 2, 3
#undef SK_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// K is a template parameter so that the K x NS row loads of a point are ALL issued before the first
// one is consumed (with a run-time K the compiler keeps one uniform branch per neighbour and the
// point costs K dependent memory round trips instead of one).
// VEC: rows are read as float4 (C % 4 == 0, 16-byte aligned slabs): lane l owns float4 l and 64 + l.
template <int K, bool VEC>
__global__ __launch_bounds__(SK_THREADS) void skip_dist_kernel(SkipArgs a)
{
    const int n = a.n, C = a.c;
    int b, slice;
    skip_item(a, b, slice);
    const int pb = a.pts_of ? a.pts_of[b] : b;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // i, the neighbour rows: SGPRs
    const int i_lo = slice * a.slice_len, i_hi = min(n, i_lo + a.slice_len);
    const float *XYZ = a.xyz + (size_t)b * n * 3;
    const float *F = a.feat + (size_t)b * n * a.feat_stride;
    const float *PX = a.prev_xyz + (size_t)pb * a.m * 3;
    const float *PF = a.prev_feat + (size_t)pb * a.m * C;
    float *DS = a.dist + (size_t)b * n * 2 * K;
    float *MN = a.mins + (size_t)b * n * 2;
    constexpr int NS = 2;
    const int C4 = C >> 2;
    const bool v0 = lane < C4, v1 = 64 + lane < C4;
    // loads are unconditional from clamped slots (a select on the result, not a branch around the load)
    const int l0 = VEC ? min(lane, C4 - 1) : 0, l1 = VEC ? min(64 + lane, C4 - 1) : 0;

    for (int i = i_lo + wave; i < i_hi; i += SK_THREADS / 64) {
        // neighbour rows (wave-uniform) and, in lanes < K, the spatial distance
        // ((dx^2 + dy^2) + dz^2, like torch.sum over 3 channels)
        int nbr[K];
        const size_t io = ((size_t)b * n + i) * K;
        if (a.idx64) {
#pragma unroll
            for (int kk = 0; kk < K; ++kk)
                nbr[kk] = (int)((const long long *)a.idx)[io + kk];
        } else {
#pragma unroll
            for (int kk = 0; kk < K; ++kk)
                nbr[kk] = ((const int *)a.idx)[io + kk];
        }
#pragma unroll
        for (int kk = 0; kk < K; ++kk)
            nbr[kk] = min(max(nbr[kk], 0), a.m - 1);
        float acc[K];
        if (VEC) {
            const sk_f4 *X4 = (const sk_f4 *)(F + (size_t)i * a.feat_stride);
            sk_f4 r[K][NS];
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                const sk_f4 *R4 = (const sk_f4 *)(PF + (size_t)nbr[kk] * C);
                r[kk][0] = R4[l0];
                r[kk][1] = R4[l1];
            }
            const sk_f4 x0 = __builtin_nontemporal_load(X4 + l0);
            const sk_f4 x1 = __builtin_nontemporal_load(X4 + l1);
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                float s = 0.f, t = 0.f, d;
                d = x0.x - r[kk][0].x; s += d * d;
                d = x0.y - r[kk][0].y; s += d * d;
                d = x0.z - r[kk][0].z; s += d * d;
                d = x0.w - r[kk][0].w; s += d * d;
                d = x1.x - r[kk][1].x; t += d * d;
                d = x1.y - r[kk][1].y; t += d * d;
                d = x1.z - r[kk][1].z; t += d * d;
                d = x1.w - r[kk][1].w; t += d * d;
                acc[kk] = (v0 ? s : 0.f) + (v1 ? t : 0.f);
            }
        } else {
            float xv[SK_CPL];
#pragma unroll
            for (int u = 0; u < SK_CPL; ++u) {
                const int c = lane + 64 * u;
                xv[u] = c < C ? F[(size_t)i * a.feat_stride + c] : 0.f;
            }
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                acc[kk] = 0.f;
                const float *row = PF + (size_t)nbr[kk] * C;
#pragma unroll
                for (int u = 0; u < SK_CPL; ++u) {
                    const int c = lane + 64 * u;
                    const float d = c < C ? xv[u] - row[c] : 0.f;
                    acc[kk] += d * d;
                }
            }
        }
        float fmin_ = __builtin_inff(), smin_ = __builtin_inff();
        float mine_f = 0.f, mine_s = 0.f;
        const float qx = XYZ[i * 3 + 0], qy = XYZ[i * 3 + 1], qz = XYZ[i * 3 + 2];
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const float f = sk_wave_sum(acc[kk]);
            const float dx = qx - PX[nbr[kk] * 3 + 0], dy = qy - PX[nbr[kk] * 3 + 1], dz = qz - PX[nbr[kk] * 3 + 2];
            const float sp = (dx * dx + dy * dy) + dz * dz;
            fmin_ = kk == 0 ? f : fminf(fmin_, f);
            smin_ = kk == 0 ? sp : fminf(smin_, sp);
            mine_f = lane == kk ? f : mine_f;
            mine_s = lane == kk ? sp : mine_s;
        }
        if (lane < K) {
            DS[(size_t)i * 2 * K + lane] = mine_s;
            DS[(size_t)i * 2 * K + K + lane] = mine_f;
        }
        if (lane == 0) {
            MN[i * 2 + 0] = smin_;
            MN[i * 2 + 1] = fmin_;
        }
    }
}

template <int K, bool VEC>
__global__ __launch_bounds__(SK_THREADS) void skip_apply_kernel(SkipArgs a)
{
    __shared__ float red[4];
    const int n = a.n, C = a.c;
    int b, slice;
    skip_item(a, b, slice);
    const int pb = a.pts_of ? a.pts_of[b] : b;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // i, the neighbour rows: SGPRs
    const int i_lo = slice * a.slice_len, i_hi = min(n, i_lo + a.slice_len);
    float *F = a.feat + (size_t)b * n * a.feat_stride;
    const float *PF = a.prev_feat + (size_t)pb * a.m * C;
    const float *DS = a.dist + (size_t)b * n * 2 * K;
    const float *MN = a.mins + (size_t)b * n * 2;
    // ---- h = mean over the patch's points of the distance to the closest of the K neighbours; every
    // slice of a patch sums the same values in the same order --------------------------------------------
    float ms = 0.f, mf = 0.f;
    for (int i = tid; i < n; i += SK_THREADS) {
        ms += MN[i * 2 + 0];
        mf += MN[i * 2 + 1];
    }
    const float hs = block_sum256(ms, red) / (float)n;
    const float hf = block_sum256(mf, red) / (float)n;
    const float hs2 = hs / 2, hf2 = hf / 2;
    constexpr int NS = 2;
    const int C4 = C >> 2;
    const bool v0 = lane < C4, v1 = 64 + lane < C4;
    const int l0 = VEC ? min(lane, C4 - 1) : 0, l1 = VEC ? min(64 + lane, C4 - 1) : 0;

    for (int i = i_lo + wave; i < i_hi; i += SK_THREADS / 64) {
        const size_t io = ((size_t)b * n + i) * K;
        int nbr[K];
        float w[K], tot = 0.f;
        if (a.idx64) {
#pragma unroll
            for (int kk = 0; kk < K; ++kk)
                nbr[kk] = (int)((const long long *)a.idx)[io + kk];
        } else {
#pragma unroll
            for (int kk = 0; kk < K; ++kk)
                nbr[kk] = ((const int *)a.idx)[io + kk];
        }
#pragma unroll
        for (int kk = 0; kk < K; ++kk)
            nbr[kk] = min(max(nbr[kk], 0), a.m - 1);
        // weights (reference :340-342): w = ws*wf;  w /= sum_k (w + 1e-5).  Lane kk evaluates neighbour kk
        // (two divisions, two exponentials, then one more division) and the results are broadcast -- the
        // same operations on the same values as evaluating all K in every lane, a fifth of the VALU work
        // (IEEE divisions and expf are ~10 instructions each and this loop is not memory-bound).
        {
            const int lk = min(lane, K - 1);
            const float mine = expf(-DS[(size_t)i * 2 * K + lk] / hs2) * expf(-DS[(size_t)i * 2 * K + K + lk] / hf2);
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                w[kk] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), kk));
                tot += w[kk] + 1e-5f;
            }
            const float mine_w = mine / tot;
#pragma unroll
            for (int kk = 0; kk < K; ++kk)
                w[kk] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine_w), kk));
        }
        if (VEC) {
            sk_f4 *X4 = (sk_f4 *)(F + (size_t)i * a.feat_stride);
            sk_f4 r[K][NS];
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                const sk_f4 *R4 = (const sk_f4 *)(PF + (size_t)nbr[kk] * C);
                r[kk][0] = R4[l0];
                r[kk][1] = R4[l1];
            }
            const sk_f4 x0 = __builtin_nontemporal_load(X4 + l0);
            const sk_f4 x1 = __builtin_nontemporal_load(X4 + l1);
            sk_f4 s0, s1;
            s0.x = w[0] * r[0][0].x; s0.y = w[0] * r[0][0].y; s0.z = w[0] * r[0][0].z; s0.w = w[0] * r[0][0].w;
            s1.x = w[0] * r[0][1].x; s1.y = w[0] * r[0][1].y; s1.z = w[0] * r[0][1].z; s1.w = w[0] * r[0][1].w;
#pragma unroll
            for (int kk = 1; kk < K; ++kk) {
                s0.x = s0.x + w[kk] * r[kk][0].x; s0.y = s0.y + w[kk] * r[kk][0].y;
                s0.z = s0.z + w[kk] * r[kk][0].z; s0.w = s0.w + w[kk] * r[kk][0].w;
                s1.x = s1.x + w[kk] * r[kk][1].x; s1.y = s1.y + w[kk] * r[kk][1].y;
                s1.z = s1.z + w[kk] * r[kk][1].z; s1.w = s1.w + w[kk] * r[kk][1].w;
            }
            s0.x = a.scale * s0.x + x0.x; s0.y = a.scale * s0.y + x0.y;
            s0.z = a.scale * s0.z + x0.z; s0.w = a.scale * s0.w + x0.w;
            s1.x = a.scale * s1.x + x1.x; s1.y = a.scale * s1.y + x1.y;
            s1.z = a.scale * s1.z + x1.z; s1.w = a.scale * s1.w + x1.w;
            if (v0)
                __builtin_nontemporal_store(s0, X4 + lane);
            if (v1)
                __builtin_nontemporal_store(s1, X4 + 64 + lane);
        } else {
            float s[SK_CPL];
#pragma unroll
            for (int u = 0; u < SK_CPL; ++u)
                s[u] = 0.f;
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                const float *row = PF + (size_t)nbr[kk] * C;
#pragma unroll
                for (int u = 0; u < SK_CPL; ++u) {
                    const int c = lane + 64 * u;
                    if (c < C)
                        s[u] = kk == 0 ? w[kk] * row[c] : s[u] + w[kk] * row[c];
                }
            }
#pragma unroll
            for (int u = 0; u < SK_CPL; ++u) {
                const int c = lane + 64 * u;
                if (c < C) {
                    float *p = F + (size_t)i * a.feat_stride + c;
                    *p = a.scale * s[u] + *p;
                }
            }
        }
    }
}

template <int K>
int skip_launch(hipStream_t s, int blocks, const SkipArgs &a, bool vec)
{
    if (vec) {
        hipLaunchKernelGGL((skip_dist_kernel<K, true>), dim3(blocks), dim3(SK_THREADS), 0, s, a);
        hipLaunchKernelGGL((skip_apply_kernel<K, true>), dim3(blocks), dim3(SK_THREADS), 0, s, a);
    } else {
        hipLaunchKernelGGL((skip_dist_kernel<K, false>), dim3(blocks), dim3(SK_THREADS), 0, s, a);
        hipLaunchKernelGGL((skip_apply_kernel<K, false>), dim3(blocks), dim3(SK_THREADS), 0, s, a);
    }
    return tpu3_launch_status();
}

constexpr int SK_SLICE_POINTS = 52;      // 312-point patches: 6 workgroups of 13 points per wave

} // namespace

extern "C" size_t tpu3_interlevel_skip_workspace_bytes(int b, int n, int k)
{
    if (b <= 0 || n <= 0 || k <= 0)
        return 0;
    return (size_t)b * n * (2 * (size_t)k + 2) * sizeof(float);
}

extern "C" int tpu3_interlevel_skip_f32(tpu3_stream_t stream, int b, int n, int k, int c, const float *xyz,
                                        float *feat, int feat_stride, const float *prev_xyz,
                                        const float *prev_feat, int m, const int32_t *pts_of, const void *idx,
                                        int idx_elem_size, float scale, int patches_per_cloud, void *workspace,
                                        size_t workspace_bytes)
{
    if (b < 0 || n <= 0 || k <= 0 || k > SK_KMAX || c <= 0 || c > 64 * SK_CPL || m <= 0) return TPU3_EINVAL;
    if (feat_stride < c || (idx_elem_size != 4 && idx_elem_size != 8)) return TPU3_EINVAL;
    if (b == 0) return TPU3_OK;
    if (!xyz || !feat || !prev_xyz || !prev_feat || !idx) return TPU3_EINVAL;
    const size_t need = tpu3_interlevel_skip_workspace_bytes(b, n, k);
    hipStream_t s = (hipStream_t)stream;
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
               per, remap, slices, slice_len, dist, dist + (size_t)b * n * 2 * k};
    // float4 rows: every row start must be 16-byte aligned
    const bool vec = c % 4 == 0 && c <= 512 && feat_stride % 4 == 0 && ((uintptr_t)feat & 15) == 0 &&
                     ((uintptr_t)prev_feat & 15) == 0;
    const int blocks = b * slices;
    int r;
    switch (k) {
    case 1: r = skip_launch<1>(s, blocks, a, vec); break;
    case 2: r = skip_launch<2>(s, blocks, a, vec); break;
    case 3: r = skip_launch<3>(s, blocks, a, vec); break;
    case 4: r = skip_launch<4>(s, blocks, a, vec); break;
    case 5: r = skip_launch<5>(s, blocks, a, vec); break;
    case 6: r = skip_launch<6>(s, blocks, a, vec); break;
    case 7: r = skip_launch<7>(s, blocks, a, vec); break;
    default: r = skip_launch<8>(s, blocks, a, vec); break;
    }
    if (own) {
        const hipError_t e = hipFreeAsync(own, s);
        if (!r) r = (int)e;
    }
    return r;
}
