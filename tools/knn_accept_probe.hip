// The round-3 judge's candidate for the feature kNN graph's selection, MEASURED on the device: per lane (query) an
// UNSORTED buffer of the 32 best keys and their maximum tau; a candidate costs one compare to reject, and only an
// accepted one replaces the maximum and re-derives it -- but a wave runs the accept path whenever ANY of its 64 lanes
// accepts.  Selection only (keys are hashes, no distance arithmetic): 312 candidates per query like a 312-point patch,
// candidates in order; compare with the shipped knn_graph_key_kernel's 1.56 ps per (query, candidate) pair INCLUDING
// its distances (tools/knn_graph_probe.py), i.e. ~1.1 ps for its sorting networks alone.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/knn_accept_probe.hip -o /tmp/knn_accept_probe && /tmp/knn_accept_probe
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int L = 32;

__device__ __forceinline__ unsigned key_of(unsigned q, unsigned j)
{
    unsigned h = q * 0x9E3779B1u ^ (j + 0x7F4A7C15u) * 0x85EBCA77u;
    h ^= h >> 15; h *= 0xC2B2AE3Du; h ^= h >> 13;
    return (h >> 1) | 1u;
}

__global__ __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(3, 3))) void accept_reject(int n, unsigned *out, unsigned long long *accepts)
{
    const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
    int buf[L];
#pragma unroll
    for (int i = 0; i < L; ++i)
        buf[i] = 0x7FFFFFFF;
    int tau = 0x7FFFFFFF;
    unsigned long long acc = 0;
    for (int j = 0; j < n; ++j) {
        const int key = (int)key_of(q, j);
        const bool take = key < tau;
        if (__builtin_amdgcn_ballot_w64(take)) {            // any lane accepts: the whole wave walks the buffer
            ++acc;
            bool done = !take;
            int m = (int)0x80000000;
#pragma unroll
            for (int i = 0; i < L; ++i) {                   // replace the (first) maximum, re-derive the maximum
                const bool hit = !done && buf[i] == tau;
                buf[i] = hit ? key : buf[i];
                done |= hit;
                m = max(m, buf[i]);
            }
            tau = take ? m : tau;
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < L; ++i)
        s += (unsigned)buf[i];
    out[q] = s;
    if ((threadIdx.x & 63) == 0)
        atomicAdd(accepts, acc);
}

int main()
{
    const int patches = 3840, n = 312;
    unsigned *out; unsigned long long *acc, h = 0;
    hipMalloc(&out, (size_t)patches * 320 * 4); hipMalloc(&acc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 6; ++rep) {
        hipMemset(acc, 0, 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL(accept_reject, dim3(patches), dim3(320), 0, 0, n, out, acc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&h, acc, 8, hipMemcpyDeviceToHost);
        if (rep >= 3)
            printf("accept / reject with an unsorted 32-slot buffer: %.3f ms per 3840 patches x 320 queries x %d candidates = %.2f ps per pair; "
                   "%.1f %% of a wave's candidates take the accept path\n", ms, n, ms * 1e9 / ((double)patches * 320 * n),
                   100.0 * h / ((double)patches * 5 * n));
    }
    return 0;
}
