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
