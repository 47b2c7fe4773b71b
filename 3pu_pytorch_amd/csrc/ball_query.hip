// ball_query.hip -- radius neighbour query for gfx950.
// Replaces sampling.ball_query (reference: sampling/sampling_cuda.cu:269-317): for every query
// the first `nsample` point indices (in index order) with d2 < radius^2, the first hit
// replicated into the unused slots, zeros when nothing is in range.
// One lane per query; candidates are staged through LDS in coalesced tiles and read back as
// wave-uniform broadcasts, so xyz is fetched from HBM/L2 once per workgroup instead of once
// per query.  A wave leaves the tile loop as soon as all of its 64 queries are full.
#include "tpu3_dev.h"

namespace {

constexpr int BQ_THREADS = 256;
constexpr int BQ_TILE = 1024;

template <typename T>
__global__ __launch_bounds__(BQ_THREADS) void ball_query_kernel(int n, int m, float radius,
                                                                int nsample,
                                                                const T *__restrict__ query,
                                                                const T *__restrict__ xyz,
                                                                int32_t *__restrict__ idx)
{
    __shared__ T tile[BQ_TILE * 3];
    const int b = blockIdx.y;
    const T *X = xyz + (size_t)b * n * 3;
    const int j = blockIdx.x * BQ_THREADS + threadIdx.x;
    const bool live = j < m;
    T qx = 0, qy = 0, qz = 0;
    if (live) {
        const T *Q = query + ((size_t)b * m + j) * 3;
        qx = Q[0];
        qy = Q[1];
        qz = Q[2];
    }
    int32_t *O = idx + ((size_t)b * m + (live ? j : 0)) * nsample;
    const float radius2 = radius * radius;   // float even for double inputs (:282)
    int cnt = live ? 0 : nsample;
    for (int k0 = 0; k0 < n; k0 += BQ_TILE) {
        const int len = min(BQ_TILE, n - k0);
        __syncthreads();
        for (int i = threadIdx.x; i < len * 3; i += BQ_THREADS)
            tile[i] = X[(size_t)k0 * 3 + i];
        __syncthreads();
        if (__syncthreads_and(cnt >= nsample))
            break;
        for (int k = 0; k < len; ++k) {
            if (__all(cnt >= nsample))
                break;
            const T d2 = tpu3_sqdist3(qx - tile[k * 3 + 0], qy - tile[k * 3 + 1], qz - tile[k * 3 + 2]);
            if (cnt < nsample && d2 < (T)radius2) {
                if (cnt == 0)
                    for (int l = 0; l < nsample; ++l)
                        O[l] = k0 + k;
                O[cnt] = k0 + k;
                ++cnt;
            }
        }
    }
}

} // namespace

extern "C" int tpu3_ball_query(tpu3_stream_t stream, int b, int n, int m, float radius, int nsample,
                               int elem_size, const void *query, const void *xyz, int32_t *idx)
{
    if (b < 0 || n < 0 || m < 0 || nsample < 0) return TPU3_EINVAL;
    if (b == 0 || m == 0 || nsample == 0) return TPU3_OK;
    if (!idx || (n > 0 && (!query || !xyz))) return TPU3_EINVAL;
    if (b > 65535) return TPU3_ELIMIT;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(idx, 0, (size_t)b * m * nsample * sizeof(int32_t), s);
    if (e != hipSuccess) return (int)e;
    if (n == 0) return TPU3_OK;
    const dim3 g((m + BQ_THREADS - 1) / BQ_THREADS, b);
    if (elem_size == 4)
        hipLaunchKernelGGL(ball_query_kernel<float>, g, dim3(BQ_THREADS), 0, s, n, m, radius, nsample,
                           (const float *)query, (const float *)xyz, idx);
    else if (elem_size == 8)
        hipLaunchKernelGGL(ball_query_kernel<double>, g, dim3(BQ_THREADS), 0, s, n, m, radius, nsample,
                           (const double *)query, (const double *)xyz, idx);
    else
        return TPU3_EINVAL;
    return tpu3_launch_status();
}
