// gather.hip -- index gather forward / backward for gfx950.
// Replaces sampling.gather_forward / gather_backward (reference: sampling/sampling_cuda.cu:28-100).
// Layout (b,c,n) channel-major as in the reference; one thread per output element, the m axis
// on threadIdx.x so idx reads and output writes are coalesced (the gathered reads are not, by
// nature -- they hit L2/MALL: a (b,c,n) slab of the sizes on this path is far below 4 MiB).
#include "tpu3_dev.h"
#include <hip/hip_fp16.h>

namespace {

template <typename T>
__global__ __launch_bounds__(256) void gather_fwd_kernel(int c, int n, int m,
                                                         const T *__restrict__ points,
                                                         const int32_t *__restrict__ idx,
                                                         T *__restrict__ out)
{
    const int b = blockIdx.z;
    const int32_t *I = idx + (size_t)b * m;
    for (int l = blockIdx.y; l < c; l += gridDim.y) {
        const T *src = points + ((size_t)b * c + l) * n;
        T *dst = out + ((size_t)b * c + l) * m;
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x)
            dst[j] = src[I[j]];
    }
}

template <typename T>
__device__ __forceinline__ void atomic_add_t(T *p, T v)
{
    atomicAdd(p, v);
}
template <>
__device__ __forceinline__ void atomic_add_t<__half>(__half *p, __half v)
{
    // 16-bit add through a 32-bit CAS on the containing word
    unsigned int *w = (unsigned int *)((uintptr_t)p & ~(uintptr_t)3);
    const bool hi = ((uintptr_t)p & 2) != 0;
    unsigned int old = *w, assumed;
    do {
        assumed = old;
        const unsigned short cur = hi ? (unsigned short)(assumed >> 16) : (unsigned short)(assumed & 0xFFFFu);
        const __half sum = __hadd(__ushort_as_half(cur), v);
        const unsigned int s = (unsigned int)__half_as_ushort(sum);
        const unsigned int repl = hi ? ((assumed & 0x0000FFFFu) | (s << 16)) : ((assumed & 0xFFFF0000u) | s);
        old = atomicCAS(w, assumed, repl);
    } while (old != assumed);
}

template <typename T>
__global__ __launch_bounds__(256) void gather_bwd_kernel(int c, int n, int m,
                                                         const T *__restrict__ grad_out,
                                                         const int32_t *__restrict__ idx,
                                                         T *__restrict__ grad_points)
{
    const int b = blockIdx.z;
    const int32_t *I = idx + (size_t)b * m;
    for (int l = blockIdx.y; l < c; l += gridDim.y) {
        const T *src = grad_out + ((size_t)b * c + l) * m;
        T *dst = grad_points + ((size_t)b * c + l) * n;
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x)
            atomic_add_t<T>(dst + I[j], src[j]);
    }
}

// ---- channel-last rows: out[b, j, :] = x[b, idx[b, j], :] and its transpose (atomic accumulation) -------------
// (the differentiable neighbour gather of group_knn, reference network/operations.py:209-211, on (B,N,C) rows:
// four channels per lane, a row of C <= 4 * 64 floats per wave pass)
template <typename I>
__global__ __launch_bounds__(256) void gather_rows_kernel(int n, long m, int c4, const float4 *__restrict__ x,
                                                          const I *__restrict__ idx, float4 *__restrict__ out)
{
    const int b = blockIdx.y;
    const float4 *X = x + (size_t)b * n * c4;
    const I *J = idx + (size_t)b * m;
    float4 *O = out + (size_t)b * m * c4;
    const long total = m * c4;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long j = t / c4;
        const int q = (int)(t - j * c4);
        const long src = min(max((long)J[j], 0L), (long)n - 1);
        O[t] = X[src * c4 + q];
    }
}

template <typename I>
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(int n, long m, int c, const float *__restrict__ g,
                                                               const I *__restrict__ idx, float *__restrict__ dx)
{
    const int b = blockIdx.y;
    const float *G = g + (size_t)b * m * c;
    const I *J = idx + (size_t)b * m;
    float *D = dx + (size_t)b * n * c;
    const long total = m * c;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long j = t / c;
        const int q = (int)(t - j * c);
        const long dst = min(max((long)J[j], 0L), (long)n - 1);
        atomicAdd(D + dst * c + q, G[t]);
    }
}

inline dim3 gather_grid(int b, int c, int m)
{
    int gx = (m + 255) / 256;
    if (gx > 1024) gx = 1024;
    int gy = c > 65535 ? 65535 : c;
    return dim3(gx, gy, b);
}

} // namespace

extern "C" int tpu3_gather_fwd(tpu3_stream_t stream, int b, int c, int n, int m, int elem_size,
                               const void *points, const int32_t *idx, void *out)
{
    if (b < 0 || c < 0 || n < 0 || m < 0) return TPU3_EINVAL;
    if (b == 0 || c == 0 || m == 0) return TPU3_OK;
    if (!points || !idx || !out || n == 0) return TPU3_EINVAL;
    if (b > 65535) return TPU3_ELIMIT;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g = gather_grid(b, c, m);
    switch (elem_size) {
    case 2:
        hipLaunchKernelGGL(gather_fwd_kernel<uint16_t>, g, dim3(256), 0, s, c, n, m,
                           (const uint16_t *)points, idx, (uint16_t *)out);
        break;
    case 4:
        hipLaunchKernelGGL(gather_fwd_kernel<uint32_t>, g, dim3(256), 0, s, c, n, m,
                           (const uint32_t *)points, idx, (uint32_t *)out);
        break;
    case 8:
        hipLaunchKernelGGL(gather_fwd_kernel<uint64_t>, g, dim3(256), 0, s, c, n, m,
                           (const uint64_t *)points, idx, (uint64_t *)out);
        break;
    default:
        return TPU3_EINVAL;
    }
    return tpu3_launch_status();
}

extern "C" int tpu3_gather_bwd(tpu3_stream_t stream, int b, int c, int n, int m, int elem_size,
                               const void *grad_out, const int32_t *idx, void *grad_points)
{
    if (b < 0 || c < 0 || n < 0 || m < 0) return TPU3_EINVAL;
    if (b == 0 || c == 0 || m == 0) return TPU3_OK;
    if (!grad_out || !idx || !grad_points || n == 0) return TPU3_EINVAL;
    if (b > 65535) return TPU3_ELIMIT;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g = gather_grid(b, c, m);
    switch (elem_size) {
    case 2:
        hipLaunchKernelGGL(gather_bwd_kernel<__half>, g, dim3(256), 0, s, c, n, m,
                           (const __half *)grad_out, idx, (__half *)grad_points);
        break;
    case 4:
        hipLaunchKernelGGL(gather_bwd_kernel<float>, g, dim3(256), 0, s, c, n, m,
                           (const float *)grad_out, idx, (float *)grad_points);
        break;
    case 8:
        hipLaunchKernelGGL(gather_bwd_kernel<double>, g, dim3(256), 0, s, c, n, m,
                           (const double *)grad_out, idx, (double *)grad_points);
        break;
    default:
        return TPU3_EINVAL;
    }
    return tpu3_launch_status();
}

extern "C" int tpu3_gather_rows_f32(tpu3_stream_t stream, int b, int n, long m, int c, const float *x, const void *idx,
                                    int idx_elem_size, float *out)
{
    if (b < 0 || n < 0 || m < 0 || c <= 0 || (idx_elem_size != 4 && idx_elem_size != 8)) return TPU3_EINVAL;
    if (b == 0 || m == 0) return TPU3_OK;
    if (!x || !idx || !out || n == 0) return TPU3_EINVAL;
    if (c % 4 || (((uintptr_t)x | (uintptr_t)out) & 15) != 0) return TPU3_ELIMIT;
    if (b > 65535) return TPU3_ELIMIT;
    long gx = (m * (c / 4) + 255) / 256;
    if (gx > 4096) gx = 4096;
    const dim3 g((unsigned)gx, b);
    if (idx_elem_size == 8)
        hipLaunchKernelGGL(gather_rows_kernel<long long>, g, dim3(256), 0, (hipStream_t)stream, n, m, c / 4,
                           (const float4 *)x, (const long long *)idx, (float4 *)out);
    else
        hipLaunchKernelGGL(gather_rows_kernel<int32_t>, g, dim3(256), 0, (hipStream_t)stream, n, m, c / 4,
                           (const float4 *)x, (const int32_t *)idx, (float4 *)out);
    return tpu3_launch_status();
}

extern "C" int tpu3_scatter_add_rows_f32(tpu3_stream_t stream, int b, int n, long m, int c, const float *g,
                                         const void *idx, int idx_elem_size, float *dx)
{
    if (b < 0 || n < 0 || m < 0 || c <= 0 || (idx_elem_size != 4 && idx_elem_size != 8)) return TPU3_EINVAL;
    if (b == 0 || m == 0) return TPU3_OK;
    if (!g || !idx || !dx || n == 0) return TPU3_EINVAL;
    if (b > 65535) return TPU3_ELIMIT;
    long gx = (m * c + 255) / 256;
    if (gx > 8192) gx = 8192;
    const dim3 grid((unsigned)gx, b);
    if (idx_elem_size == 8)
        hipLaunchKernelGGL(scatter_add_rows_kernel<long long>, grid, dim3(256), 0, (hipStream_t)stream, n, m, c, g,
                           (const long long *)idx, dx);
    else
        hipLaunchKernelGGL(scatter_add_rows_kernel<int32_t>, grid, dim3(256), 0, (hipStream_t)stream, n, m, c, g,
                           (const int32_t *)idx, dx);
    return tpu3_launch_status();
}
