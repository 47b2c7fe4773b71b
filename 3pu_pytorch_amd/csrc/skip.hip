// skip.hip -- fused inter-level skip connection of a Level (inference) for gfx950.
//
// Replaces network/upsampler.py:317-347 of the reference: for every point of a patch, its K nearest
// points of the previous level's merged cloud (indices from the kNN kernel) contribute their
// features with bilateral weights
//     w_k  = exp(-|p_i - q_k|^2 / (h_s/2)) * exp(-|x_i - f_k|^2 / (h_f/2)),   h = mean_i min_k dist
//     w_k /= sum_k (w_k + 1e-5),        x_i += 0.2 * sum_k w_k f_k
// The reference gathers a (B,264,N,K) tensor (25 GB for the level-4 patches of 8 clouds), reduces it
// twice and re-reads it for the weighted sum through ~20 ATen kernels.  Here one workgroup owns one
// patch and makes two passes over the K neighbour rows of each point straight from the previous
// level's feature table (L2 / MALL resident: 6.6 MB per cloud), nothing is materialised:
//   pass A  spatial and feature distances (a wave per point, lanes across the 264 channels, the K rows
//           of a point in flight together) -> LDS;  h_s, h_f by a block reduction;  weights -> LDS
//   pass B  x_i += 0.2 * sum_k w_k f_k, in place in the level's feature buffer.
#include "tpu3_dev.h"

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
};

__device__ __forceinline__ float block_sum256(float v, float *red)
{
    v = tpu3_wave_sum_f32(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(SK_THREADS) void skip_fused_kernel(SkipArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n = a.n, K = a.k, C = a.c;
    float *ds = lds;                 // n*K spatial distances, later the weights
    float *df = ds + n * K;          // n*K feature distances
    int *nb = (int *)(df + n * K);   // n*K neighbour rows
    float *red = (float *)(nb + n * K);
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  All patches of one
    // previous cloud gather from the same (de-duplicated: ~1.5 MB) slice of prev_feat, so they are
    // sent to the same XCD: block i -> XCD i % 8 handles cloud (i % 8) + 8 * (slot / per_cloud).
    int b = blockIdx.x;
    if (b < a.remap_blocks) {
        const int x = b & 7, slot = b >> 3;
        const int cl = slot / a.per_cloud;
        b = (x + 8 * cl) * a.per_cloud + (slot - cl * a.per_cloud);
    }
    const int pb = a.pts_of ? a.pts_of[b] : b;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *XYZ = a.xyz + (size_t)b * n * 3;
    float *F = a.feat + (size_t)b * n * a.feat_stride;
    const float *PX = a.prev_xyz + (size_t)pb * a.m * 3;
    const float *PF = a.prev_feat + (size_t)pb * a.m * C;

    // ---- neighbour rows + spatial distances ((dx^2 + dy^2) + dz^2, like torch.sum over 3 channels) --
    for (int t = tid; t < n * K; t += SK_THREADS) {
        const int i = t / K;
        const size_t io = ((size_t)b * n) * K + t;
        int j = a.idx64 ? (int)((const long long *)a.idx)[io] : ((const int *)a.idx)[io];
        j = min(max(j, 0), a.m - 1);
        nb[t] = j;
        const float dx = XYZ[i * 3 + 0] - PX[j * 3 + 0];
        const float dy = XYZ[i * 3 + 1] - PX[j * 3 + 1];
        const float dz = XYZ[i * 3 + 2] - PX[j * 3 + 2];
        ds[t] = (dx * dx + dy * dy) + dz * dz;
    }
    __syncthreads();
    // ---- pass A: feature distances, a wave per point, lanes across channels ------------------------------
    for (int i = wave; i < n; i += SK_THREADS / 64) {
        float xv[SK_CPL];
#pragma unroll
        for (int u = 0; u < SK_CPL; ++u) {
            const int c = lane + 64 * u;
            xv[u] = c < C ? F[(size_t)i * a.feat_stride + c] : 0.f;
        }
        float acc[SK_KMAX];
#pragma unroll
        for (int kk = 0; kk < SK_KMAX; ++kk) {
            acc[kk] = 0.f;
            if (kk < K) {
                const float *row = PF + (size_t)nb[i * K + kk] * C;
#pragma unroll
                for (int u = 0; u < SK_CPL; ++u) {
                    const int c = lane + 64 * u;
                    const float d = c < C ? xv[u] - row[c] : 0.f;
                    acc[kk] += d * d;
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < SK_KMAX; ++kk)
            if (kk < K) {
                const float s = tpu3_wave_sum_f32(acc[kk]);
                if (lane == 0)
                    df[i * K + kk] = s;
            }
    }
    __syncthreads();
    // ---- h = mean over points of the distance to the closest of the K neighbours --------------------------
    float ms = 0.f, mf = 0.f;
    for (int i = tid; i < n; i += SK_THREADS) {
        float a0 = ds[i * K], b0 = df[i * K];
        for (int kk = 1; kk < K; ++kk) {
            a0 = fminf(a0, ds[i * K + kk]);
            b0 = fminf(b0, df[i * K + kk]);
        }
        ms += a0;
        mf += b0;
    }
    const float hs = block_sum256(ms, red) / (float)n;
    const float hf = block_sum256(mf, red) / (float)n;
    const float hs2 = hs / 2, hf2 = hf / 2;
    __syncthreads();
    // ---- weights (reference :340-342): w = ws*wf;  w /= sum_k (w + 1e-5) ------------------------------------
    for (int i = tid; i < n; i += SK_THREADS) {
        float w[SK_KMAX], tot = 0.f;
        for (int kk = 0; kk < K; ++kk) {
            w[kk] = expf(-ds[i * K + kk] / hs2) * expf(-df[i * K + kk] / hf2);
            tot += w[kk] + 1e-5f;
        }
        for (int kk = 0; kk < K; ++kk)
            ds[i * K + kk] = w[kk] / tot;
    }
    __syncthreads();
    // ---- pass B: x_i += scale * sum_k w_k f_k ------------------------------------------------------------------
    for (int i = wave; i < n; i += SK_THREADS / 64) {
        float s[SK_CPL];
#pragma unroll
        for (int u = 0; u < SK_CPL; ++u)
            s[u] = 0.f;
#pragma unroll
        for (int kk = 0; kk < SK_KMAX; ++kk)
            if (kk < K) {
                const float wk = ds[i * K + kk];
                const float *row = PF + (size_t)nb[i * K + kk] * C;
#pragma unroll
                for (int u = 0; u < SK_CPL; ++u) {
                    const int c = lane + 64 * u;
                    if (c < C)
                        s[u] = kk == 0 ? wk * row[c] : s[u] + wk * row[c];
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

} // namespace

extern "C" int tpu3_interlevel_skip_f32(tpu3_stream_t stream, int b, int n, int k, int c, const float *xyz,
                                        float *feat, int feat_stride, const float *prev_xyz,
                                        const float *prev_feat, int m, const int32_t *pts_of, const void *idx,
                                        int idx_elem_size, float scale, int patches_per_cloud)
{
    if (b < 0 || n <= 0 || k <= 0 || k > SK_KMAX || c <= 0 || c > 64 * SK_CPL || m <= 0) return TPU3_EINVAL;
    if (feat_stride < c || (idx_elem_size != 4 && idx_elem_size != 8)) return TPU3_EINVAL;
    if (b == 0) return TPU3_OK;
    if (!xyz || !feat || !prev_xyz || !prev_feat || !idx) return TPU3_EINVAL;
    const size_t lds = ((size_t)3 * n * k + 16) * sizeof(float);
    if (lds > 150 * 1024) return TPU3_ELIMIT;
    int per = patches_per_cloud > 0 && b % patches_per_cloud == 0 ? patches_per_cloud : 0;
    const int remap = per ? (b / per / 8) * 8 * per : 0;
    SkipArgs a{n, k, c, feat_stride, m, xyz, feat, prev_xyz, prev_feat, pts_of, idx, idx_elem_size == 8, scale,
               per, remap};
    hipError_t e = hipFuncSetAttribute((const void *)skip_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(skip_fused_kernel, dim3(b), dim3(SK_THREADS), lds, (hipStream_t)stream, a);
    return tpu3_launch_status();
}
