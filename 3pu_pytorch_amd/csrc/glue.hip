// glue.hip -- the data-movement steps BETWEEN the eval path's kernels, one launch each (gfx950, round 6).
//
// One cloud's 16x upsampling is a chain of ~390 dependent launches (reference: a Python loop over 48 patches x 4 levels
// of ~60 ATen kernels each, main.py:237-244, upsampler.py:107-189); 139 of them were ATen element-wise / reduce / sort /
// gather calls issued by this package's host code between the hand-written kernels -- 1.8 ms of kernel time per cloud,
// but a dispatch-after-drain gap each (7.4 ms of a 33 ms cloud, profiles/r05_kernel_stats_one_cloud.csv).  The steps
// they implement are fixed small functions of the reference:
//
//   repatch_filter   upsampler.py:63-77   d < 5 * mean(d) outlier mask, masked_select, N', patch_num = int(N'/k*5)
//   repatch_seeds    upsampler.py:78-79   the seeds' coordinates, patches beyond a cloud's count repeat its last one
//   gather_xyz       upsampler.py:158, main.py:380   rows of a cloud by int32 index (optionally written channel-first)
//   denormalize      upsampler.py:147, main.py:242   x * radius + centroid per patch
//   normalize (cl)   csrc/normalize.hip           the channel-last form of normalize_point_batch
//   fill_f32_i32     operations.fps           temp = 1e10 and idx = 0 in one launch
//
// Arithmetic notes.  mean(d): the sum runs in double over a fixed partition (thread chunks, wave butterflies, waves in
// order) and is rounded to float once -- torch.mean's float accumulation order is unspecified; the mask differs from it
// only for a distance within an ulp or two of 5 * mean.  patch_num is computed in double like Python's
// int(N' / k * 5).  x * radius + centroid is a multiplication and an addition, each rounded (the library is built with
// -ffp-contract=off), as the reference's two ATen kernels do.
#include "tpu3_dev.h"

namespace {

constexpr int RF_THREADS = 1024;

__global__ __launch_bounds__(RF_THREADS) void repatch_filter_kernel(int n, int k, int r, const float *__restrict__ dist,
                                                                   int dstride, const float *__restrict__ xyz,
                                                                   float *__restrict__ xyz_f, int32_t *__restrict__ count,
                                                                   int32_t *__restrict__ patch_num,
                                                                   int32_t *__restrict__ old_count,
                                                                   int32_t *__restrict__ m_count,
                                                                   unsigned long long *__restrict__ small)
{
    __shared__ double wsum[RF_THREADS / 64];
    __shared__ int wcnt[RF_THREADS / 64];
    __shared__ float thr_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *D = dist + (size_t)b * n * dstride;
    const float *X = xyz + (size_t)b * n * 3;
    float *Y = xyz_f + (size_t)b * n * 3;
    const int per = (n + RF_THREADS - 1) / RF_THREADS;
    const int i0 = min(n, tid * per), i1 = min(n, i0 + per);
    double acc = 0.0;
    for (int i = i0; i < i1; ++i)
        acc += (double)D[(size_t)i * dstride];
    for (int off = 32; off > 0; off >>= 1)
        acc += __shfl_xor(acc, off, 64);
    if (lane == 0)
        wsum[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < RF_THREADS / 64; ++w)
            s += wsum[w];
        thr_s = 5.f * (float)(s / (double)n);                   // 5 * torch.mean(d)  (upsampler.py:67-71)
    }
    __syncthreads();
    const float thr = thr_s;
    int kept = 0;
    for (int i = i0; i < i1; ++i)
        kept += D[(size_t)i * dstride] < thr ? 1 : 0;
    int inc = kept;                                             // inclusive scan over the workgroup's threads
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        inc += lane >= d ? o : 0;
    }
    if (lane == 63)
        wcnt[wave] = inc;
    __syncthreads();
    int before = inc - kept, total = 0;
    for (int w = 0; w < RF_THREADS / 64; ++w) {
        const int c = wcnt[w];
        before += w < wave ? c : 0;
        total += c;
    }
    // stable partition: the kept points first, in their order (masked_select), the dropped ones behind them, in theirs
    int kp = before, dp = total + (i0 - before);
    for (int i = i0; i < i1; ++i) {
        const bool keep = D[(size_t)i * dstride] < thr;
        const int pos = keep ? kp++ : dp++;
        Y[(size_t)pos * 3 + 0] = X[(size_t)i * 3 + 0];
        Y[(size_t)pos * 3 + 1] = X[(size_t)i * 3 + 1];
        Y[(size_t)pos * 3 + 2] = X[(size_t)i * 3 + 2];
    }
    if (tid == 0) {
        const int pn = max(1, (int)__builtin_floor((double)total / (double)k * 5.0));       // int(N' / k * 5), >= 1
        count[b] = total;
        patch_num[b] = pn;
        if (old_count) old_count[b] = pn * k;
        if (m_count) m_count[b] = pn * k * r;
        if (small && total < k)
            atomicAdd(small, 1ull);
    }
}

__global__ __launch_bounds__(256) void repatch_seeds_kernel(int b, int n, int p, const int32_t *__restrict__ seed_idx,
                                                           const int32_t *__restrict__ patch_num,
                                                           const float *__restrict__ xyz_f, float *__restrict__ seeds)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= b * p)
        return;
    const int e = i / p, j = i - e * p;
    const int slot = min(j, patch_num[e] - 1);
    int s = seed_idx[(size_t)e * p + slot];
    s = min(max(s, 0), n - 1);
    const float *src = xyz_f + ((size_t)e * n + s) * 3;
    seeds[(size_t)i * 3 + 0] = src[0];
    seeds[(size_t)i * 3 + 1] = src[1];
    seeds[(size_t)i * 3 + 2] = src[2];
}

template <bool NCHW_OUT>
__global__ __launch_bounds__(256) void gather_xyz_kernel(int n, int m, const float *__restrict__ x,
                                                        const int32_t *__restrict__ idx, float *__restrict__ out)
{
    const int b = blockIdx.y;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < m; j += gridDim.x * 256) {
        int s = idx[(size_t)b * m + j];
        s = min(max(s, 0), n - 1);
        const float *src = x + ((size_t)b * n + s) * 3;
        const float vx = src[0], vy = src[1], vz = src[2];
        if (NCHW_OUT) {
            float *o = out + (size_t)b * 3 * m + j;
            o[0] = vx; o[m] = vy; o[2 * (size_t)m] = vz;
        } else {
            float *o = out + ((size_t)b * m + j) * 3;
            o[0] = vx; o[1] = vy; o[2] = vz;
        }
    }
}

__global__ __launch_bounds__(256) void denormalize_kernel(long rows, int per, const float *__restrict__ x,
                                                         const float *__restrict__ radius,
                                                         const float *__restrict__ centroid, float *__restrict__ out)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * 3; i += (long)gridDim.x * 256) {
        const long row = i / 3;
        const int c = (int)(i - row * 3);
        const long p = row / per;
        const float v = x[i] * radius[p];
        out[i] = v + centroid[p * 3 + c];
    }
}

__global__ __launch_bounds__(256) void fill_f32_i32_kernel(float *__restrict__ a, long na, float va,
                                                          int32_t *__restrict__ c, long nc, int32_t vc)
{
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < na; i += stride)
        a[i] = va;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nc; i += stride)
        c[i] = vc;
}

} // namespace

extern "C" int tpu3_repatch_filter_f32(tpu3_stream_t stream, int b, int n, int k, int r, const float *dist, int dstride,
                                       const float *xyz, float *xyz_f, int32_t *count, int32_t *patch_num,
                                       int32_t *old_count, int32_t *m_count, unsigned long long *small_events)
{
    if (b < 0 || n < 0 || k <= 0 || r <= 0 || dstride <= 0) return TPU3_EINVAL;
    if (b == 0) return TPU3_OK;
    if (n == 0 || !dist || !xyz || !xyz_f || !count || !patch_num) return TPU3_EINVAL;
    if (b > 65535 * 32) return TPU3_ELIMIT;
    hipLaunchKernelGGL(repatch_filter_kernel, dim3(b), dim3(RF_THREADS), 0, (hipStream_t)stream, n, k, r, dist, dstride, xyz,
                       xyz_f, count, patch_num, old_count, m_count, small_events);
    return tpu3_launch_status();
}

extern "C" int tpu3_repatch_seeds_f32(tpu3_stream_t stream, int b, int n, int p, const int32_t *seed_idx,
                                      const int32_t *patch_num, const float *xyz_f, float *seeds)
{
    if (b < 0 || n <= 0 || p < 0) return TPU3_EINVAL;
    if (b == 0 || p == 0) return TPU3_OK;
    if (!seed_idx || !patch_num || !xyz_f || !seeds) return TPU3_EINVAL;
    hipLaunchKernelGGL(repatch_seeds_kernel, dim3((unsigned)(((long)b * p + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b, n,
                       p, seed_idx, patch_num, xyz_f, seeds);
    return tpu3_launch_status();
}

extern "C" int tpu3_gather_xyz_f32(tpu3_stream_t stream, int b, int n, int m, const float *x, const int32_t *idx,
                                   float *out, int nchw_out)
{
    if (b < 0 || n <= 0 || m < 0) return TPU3_EINVAL;
    if (b == 0 || m == 0) return TPU3_OK;
    if (!x || !idx || !out) return TPU3_EINVAL;
    if (b > 65535) return TPU3_ELIMIT;
    const dim3 g((unsigned)min((m + 255) / 256, 4096), b);
    if (nchw_out)
        hipLaunchKernelGGL(gather_xyz_kernel<true>, g, dim3(256), 0, (hipStream_t)stream, n, m, x, idx, out);
    else
        hipLaunchKernelGGL(gather_xyz_kernel<false>, g, dim3(256), 0, (hipStream_t)stream, n, m, x, idx, out);
    return tpu3_launch_status();
}

extern "C" int tpu3_denormalize_f32(tpu3_stream_t stream, long patches, int rows_per_patch, const float *x,
                                    const float *radius, const float *centroid, float *out)
{
    if (patches < 0 || rows_per_patch <= 0) return TPU3_EINVAL;
    if (patches == 0) return TPU3_OK;
    if (!x || !radius || !centroid || !out) return TPU3_EINVAL;
    const long rows = patches * rows_per_patch;
    const long blocks = (rows * 3 + 255) / 256;
    hipLaunchKernelGGL(denormalize_kernel, dim3((unsigned)(blocks > 65535 ? 65535 : blocks)), dim3(256), 0, (hipStream_t)stream,
                       rows, rows_per_patch, x, radius, centroid, out);
    return tpu3_launch_status();
}

extern "C" int tpu3_fill_f32_i32(tpu3_stream_t stream, float *a, long na, float va, int32_t *c, long nc, int32_t vc)
{
    if (na < 0 || nc < 0 || (na > 0 && !a) || (nc > 0 && !c)) return TPU3_EINVAL;
    if (na == 0 && nc == 0) return TPU3_OK;
    const long most = na > nc ? na : nc;
    const long blocks = (most + 255) / 256;
    hipLaunchKernelGGL(fill_f32_i32_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, a,
                       na, va, c, nc, vc);
    return tpu3_launch_status();
}
