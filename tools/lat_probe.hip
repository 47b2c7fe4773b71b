// lat_probe.hip -- development micro-benchmark (GPU box): latency of one wave's dependent, coalesced
// 1 KiB bucket reads (float4 per lane) from working sets of different sizes, with/without a store.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

__global__ void chase(float4 *data, const int *next, int steps, int do_store, unsigned long long *out, float *sink)
{
    int b = 0;
    float acc = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < steps; ++i) {
        float4 v = data[b * 64 + threadIdx.x];
        acc += v.x;
        if (do_store && v.y > 2.f) ((float *)(data + b * 64 + threadIdx.x))[3] = acc;
        // next bucket depends on the loaded data (v.z holds the next index as float bits)
        b = __builtin_amdgcn_readfirstlane(__float_as_int(v.z));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; }
    sink[threadIdx.x] = acc;
}

int main()
{
    for (size_t mb : {1, 2, 3, 4, 6, 8, 32, 512}) {
        const int nb = (int)(mb * 1024 * 1024 / 1024);
        std::vector<float4> h((size_t)nb * 64);
        std::vector<int> perm(nb);
        for (int i = 0; i < nb; ++i) perm[i] = i;
        srand(1);
        for (int i = nb - 1; i > 0; --i) { int j = rand() % (i + 1); std::swap(perm[i], perm[j]); }
        for (int i = 0; i < nb; ++i) {
            int nxt = perm[(i + 1) % nb];   // cycle through a random permutation
            for (int l = 0; l < 64; ++l) { h[(size_t)perm[i] * 64 + l] = make_float4(1.f, 0.f, 0.f, 0.f);
                                           ((int *)&h[(size_t)perm[i] * 64 + l])[2] = nxt; }
        }
        float4 *d; unsigned long long *out; float *sink;
        hipMalloc(&d, h.size() * sizeof(float4)); hipMalloc(&out, 64); hipMalloc(&sink, 256);
        hipMemcpy(d, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice);
        const int steps = 20000;
        for (int st = 0; st < 2; ++st) {
            hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, nullptr, steps, st, out, sink);   // warm
            hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, nullptr, steps, st, out, sink);
            unsigned long long c; hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
            printf("working set %4zu MiB store=%d : %.1f cycles per dependent 1 KiB wave read\n", mb, st, (double)c / steps);
        }
        hipFree(d); hipFree(out); hipFree(sink);
    }
    return 0;
}
