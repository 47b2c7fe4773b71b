// ball_query.hip -- radius neighbour query for gfx950.
// Replaces sampling.ball_query (reference: sampling/sampling_cuda.cu:269-317): for every query
// the first `nsample` point indices (in index order) with d2 < radius^2, the first hit
// replicated into the unused slots, zeros when nothing is in range.
// (r6) A WAVE per query: its 64 lanes test 64 candidates at a time -- ballot, then every hit writes itself to slot
// cnt + (number of hits in lower lanes): index order for free -- so a query is 79 wave-wide steps for 5000 candidates
// instead of 5000 dependent ones (one lane per query, rounds 1-5: 0.59 ms for 48 x 312 queries x 5000 points on 96
// workgroups).  Eight queries share a workgroup and the candidate tiles it stages through LDS (structure of arrays:
// lane i reads x[i], conflict-free), so xyz is read from L2 once per eight queries.  A workgroup leaves the tile loop
// as soon as all of its queries are full; the unused slots get the first hit (zeros when nothing is in range) in the
// kernel itself: no memset launch.
#include "tpu3_dev.h"

namespace {

constexpr int BQ_WAVES = 8;
constexpr int BQ_TILE = 1024;

template <typename T>
__global__ __launch_bounds__(64 * BQ_WAVES) void ball_query_kernel(int n, int m, float radius, int nsample,
                                                                  const T *__restrict__ query,
                                                                  const T *__restrict__ xyz,
                                                                  int32_t *__restrict__ idx)
{
    __shared__ T tx[BQ_TILE], ty[BQ_TILE], tz[BQ_TILE];
    const int b = blockIdx.y;
    const T *X = xyz + (size_t)b * n * 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * BQ_WAVES + wave;             // the wave's query
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
    int cnt = live ? 0 : nsample;            // wave-uniform
    int first = 0;
    for (int k0 = 0; k0 < n; k0 += BQ_TILE) {
        const int len = min(BQ_TILE, n - k0);
        __syncthreads();
        for (int i = threadIdx.x; i < len; i += 64 * BQ_WAVES) {
            tx[i] = X[(size_t)(k0 + i) * 3 + 0];
            ty[i] = X[(size_t)(k0 + i) * 3 + 1];
            tz[i] = X[(size_t)(k0 + i) * 3 + 2];
        }
        if (__syncthreads_and(cnt >= nsample))
            break;
        for (int k = 0; k < len && cnt < nsample; k += 64) {
            const int c = k + lane;
            bool hit = false;
            if (c < len) {
                const T d2 = tpu3_sqdist3(qx - tx[c], qy - ty[c], qz - tz[c]);
                hit = d2 < (T)radius2;
            }
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
            if (mask) {
                if (cnt == 0)
                    first = k0 + k + __builtin_ctzll(mask);
                const int slot = cnt + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                if (hit && slot < nsample)
                    O[slot] = k0 + c;
                cnt += __builtin_popcountll(mask);
            }
        }
    }
    // the unused slots: the first hit, or zeros when nothing is in range (sampling_cuda.cu:291-300)
    if (live && cnt < nsample)
        for (int l = cnt + lane; l < nsample; l += 64)
            O[l] = cnt > 0 ? first : 0;
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
    if (n == 0) {
        const hipError_t e = hipMemsetAsync(idx, 0, (size_t)b * m * nsample * sizeof(int32_t), s);
        return e == hipSuccess ? TPU3_OK : (int)e;
    }
    const dim3 g((m + BQ_WAVES - 1) / BQ_WAVES, b);
    if (elem_size == 4)
        hipLaunchKernelGGL(ball_query_kernel<float>, g, dim3(64 * BQ_WAVES), 0, s, n, m, radius, nsample,
                           (const float *)query, (const float *)xyz, idx);
    else if (elem_size == 8)
        hipLaunchKernelGGL(ball_query_kernel<double>, g, dim3(64 * BQ_WAVES), 0, s, n, m, radius, nsample,
                           (const double *)query, (const double *)xyz, idx);
    else
        return TPU3_EINVAL;
    return tpu3_launch_status();
}
