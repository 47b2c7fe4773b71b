// The round-3 judge's candidate for the inter-level skip connection, MEASURED: a LANE per point that walks its own
// 264-float row and its K = 5 gathered rows of the previous level with float4 loads and private fmaf chains (feature
// distances only: what skip_dist_kernel does with a wave per point and row-wide loads, 0.61 ms per 3840-patch launch).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/skip_lane_probe.hip -o /tmp/skip_lane_probe && /tmp/skip_lane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int C4 = 66, K = 5;       // 264 channels

__global__ __launch_bounds__(256) void lane_per_point(int n, const f4 *feat, const f4 *prev, const int *idx, float *dist)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const f4 *own = feat + (size_t)p * C4;
    const f4 *nb[K];
#pragma unroll
    for (int k = 0; k < K; ++k)
        nb[k] = prev + (size_t)idx[p * K + k] * C4;
    float d[K] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int c = 0; c < C4; ++c) {
        const f4 x = own[c];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const f4 y = nb[k][c];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float t = x[q] - y[q];
                d[k] = __builtin_fmaf(t, t, d[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
        dist[p * K + k] = d[k];
}

// the shape of the shipped kernel: a wave per point, lanes across the row (row-wide loads), DPP sums
__global__ __launch_bounds__(256) void wave_per_point(int n, const f4 *feat, const f4 *prev, const int *idx, float *dist)
{
    const int lane = threadIdx.x & 63;
    for (int p = blockIdx.x * 4 + (threadIdx.x >> 6); p < n; p += gridDim.x * 4) {
        const f4 x0 = feat[(size_t)p * C4 + lane], x1 = lane < 2 ? feat[(size_t)p * C4 + 64 + lane] : (f4){0, 0, 0, 0};
        float d[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const f4 *r = prev + (size_t)idx[p * K + k] * C4;
            const f4 y0 = r[lane], y1 = lane < 2 ? r[64 + lane] : (f4){0, 0, 0, 0};
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a = x0[q] - y0[q], b = x1[q] - y1[q];
                s = __builtin_fmaf(a, a, s);
                s = __builtin_fmaf(b, b, s);
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1)
                s += __shfl_xor(s, o, 64);
            d[k] = s;
        }
        if (lane < K)
            dist[p * K + lane] = d[lane];
    }
}

int main()
{
    const int n = 3840 * 312, m = 600000;
    std::vector<int> hidx((size_t)n * K);
    srand(1);
    for (int p = 0; p < n; ++p) {
        const int base = (int)((long)p * m / n);
        for (int k = 0; k < K; ++k) {
            int j = base + (rand() % 2000) - 1000;          // neighbours of a point sit in its outer patch's rows
            hidx[(size_t)p * K + k] = j < 0 ? 0 : (j >= m ? m - 1 : j);
        }
    }
    f4 *feat, *prev; int *idx; float *dist;
    hipMalloc(&feat, (size_t)n * C4 * 16); hipMalloc(&prev, (size_t)m * C4 * 16);
    hipMalloc(&idx, hidx.size() * 4); hipMalloc(&dist, (size_t)n * K * 4);
    hipMemset(feat, 0, (size_t)n * C4 * 16); hipMemset(prev, 0, (size_t)m * C4 * 16);
    hipMemcpy(idx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which)
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            if (which == 0)
                hipLaunchKernelGGL(lane_per_point, dim3((n + 255) / 256), dim3(256), 0, 0, n, feat, prev, idx, dist);
            else
                hipLaunchKernelGGL(wave_per_point, dim3(256 * 16), dim3(256), 0, 0, n, feat, prev, idx, dist);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 3)
                printf("%s: %.3f ms for %d points (K = 5 feature distances over 264 channels)\n",
                       which == 0 ? "lane per point, float4 loads, private chains" : "wave per point, row-wide loads, wave sums  ", ms, n);
        }
    return 0;
}
