// normalize.hip -- per-patch normalisation for gfx950.
// Replaces network.operations.normalize_point_batch (reference: network/operations.py:12-30) on
// NCHW data: centroid = mean over points, pc -= centroid, radius = max_n sqrt(sum_c pc^2),
// pc /= radius.  The reference issues five small torch kernels per call and re-reads the patch
// each time; here one workgroup per patch keeps its points in registers between the three
// passes (a patch is 312..1024 points), so HBM sees one read and one write.
#include "tpu3_dev.h"

namespace {

constexpr int NZ_THREADS = 256;
constexpr int NZ_PPT = 8;       // register-resident up to 2048 points; longer patches re-read

__device__ __forceinline__ float block_sum(float v, float *red)
{
    v = tpu3_wave_sum_f32(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float block_max(float v, float *red)
{
    v = tpu3_wave_max_f32(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// CL: channel-last rows (b, n, 3) in and out instead of the reference's (b, 3, n); same operations in the same order
template <bool CL>
__global__ __launch_bounds__(NZ_THREADS) void normalize_kernel(int n_pad, const int32_t *__restrict__ n_arr,
                                                               const float *__restrict__ pc,
                                                               float *__restrict__ out,
                                                               float *__restrict__ centroid,
                                                               float *__restrict__ radius)
{
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int n = n_arr ? n_arr[b] : n_pad;
    constexpr int S = CL ? 3 : 1;                   // distance between consecutive points of one channel
    const float *X = pc + (size_t)b * 3 * n_pad, *Y = X + (CL ? 1 : n_pad), *Z = Y + (CL ? 1 : n_pad);
    float *OX = out + (size_t)b * 3 * n_pad, *OY = OX + (CL ? 1 : n_pad), *OZ = OY + (CL ? 1 : n_pad);
    const bool resident = n <= NZ_THREADS * NZ_PPT;
    float x[NZ_PPT], y[NZ_PPT], z[NZ_PPT];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    if (resident) {
#pragma unroll
        for (int j = 0; j < NZ_PPT; ++j) {
            const int k = threadIdx.x + j * NZ_THREADS;
            const bool live = k < n;
            x[j] = live ? X[k * S] : 0.f;
            y[j] = live ? Y[k * S] : 0.f;
            z[j] = live ? Z[k * S] : 0.f;
            sx += x[j];
            sy += y[j];
            sz += z[j];
        }
    } else {
        for (int k = threadIdx.x; k < n; k += NZ_THREADS) {
            sx += X[k * S];
            sy += Y[k * S];
            sz += Z[k * S];
        }
    }
    const float cx = block_sum(sx, red) / (float)n;
    const float cy = block_sum(sy, red) / (float)n;
    const float cz = block_sum(sz, red) / (float)n;
    float r2 = 0.f;
    if (resident) {
#pragma unroll
        for (int j = 0; j < NZ_PPT; ++j) {
            const int k = threadIdx.x + j * NZ_THREADS;
            x[j] -= cx;
            y[j] -= cy;
            z[j] -= cz;
            if (k < n)
                r2 = fmaxf(r2, (x[j] * x[j] + y[j] * y[j]) + z[j] * z[j]);
        }
    } else {
        for (int k = threadIdx.x; k < n; k += NZ_THREADS) {
            const float dx = X[k * S] - cx, dy = Y[k * S] - cy, dz = Z[k * S] - cz;
            r2 = fmaxf(r2, (dx * dx + dy * dy) + dz * dz);
        }
    }
    const float r = sqrtf(block_max(r2, red));      // sqrt is monotone: max sqrt == sqrt max
    if (resident) {
#pragma unroll
        for (int j = 0; j < NZ_PPT; ++j) {
            const int k = threadIdx.x + j * NZ_THREADS;
            if (k < n) {
                OX[k * S] = x[j] / r;
                OY[k * S] = y[j] / r;
                OZ[k * S] = z[j] / r;
            }
        }
    } else {
        for (int k = threadIdx.x; k < n; k += NZ_THREADS) {
            OX[k * S] = (X[k * S] - cx) / r;
            OY[k * S] = (Y[k * S] - cy) / r;
            OZ[k * S] = (Z[k * S] - cz) / r;
        }
    }
    if (threadIdx.x == 0) {
        centroid[b * 3 + 0] = cx;
        centroid[b * 3 + 1] = cy;
        centroid[b * 3 + 2] = cz;
        radius[b] = r;
    }
}

} // namespace

extern "C" int tpu3_normalize_f32(tpu3_stream_t stream, int b, int n, const int32_t *n_arr,
                                  const float *pc, float *out, float *centroid, float *radius)
{
    if (b < 0 || n < 0) return TPU3_EINVAL;
    if (b == 0 || n == 0) return TPU3_OK;
    if (!pc || !out || !centroid || !radius) return TPU3_EINVAL;
    hipLaunchKernelGGL(normalize_kernel<false>, dim3(b), dim3(NZ_THREADS), 0, (hipStream_t)stream, n, n_arr, pc,
                       out, centroid, radius);
    return tpu3_launch_status();
}

extern "C" int tpu3_normalize_cl_f32(tpu3_stream_t stream, int b, int n, const int32_t *n_arr,
                                     const float *pc, float *out, float *centroid, float *radius)
{
    if (b < 0 || n < 0) return TPU3_EINVAL;
    if (b == 0 || n == 0) return TPU3_OK;
    if (!pc || !out || !centroid || !radius) return TPU3_EINVAL;
    hipLaunchKernelGGL(normalize_kernel<true>, dim3(b), dim3(NZ_THREADS), 0, (hipStream_t)stream, n, n_arr, pc,
                       out, centroid, radius);
    return tpu3_launch_status();
}
