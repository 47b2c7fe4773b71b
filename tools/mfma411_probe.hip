// Issue rate of v_mfma_f32_4x4x1_16b_f32 (plain and with the A-broadcast control) against v_mfma_f32_16x16x4_f32:
// cycles per instruction per SIMD for 1 / 2 / 4 waves per SIMD and 1 / 3 / 6 independent accumulator chains.
// hipcc --offload-arch=gfx950 -O3 tools/mfma411_probe.hip -o /tmp/mfma411_probe && /tmp/mfma411_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int CH>
__global__ void probe(float *out, int iters, unsigned long long *cyc)
{
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if constexpr (MODE == 0) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
                if constexpr (MODE == 1) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 4, 5, 0);
                if constexpr (MODE == 2) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE, int CH>
void run(const char *name, int waves_per_simd)
{
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    probe<MODE, CH><<<256, 256 * waves_per_simd>>>(out, iters, cyc);      // one block per CU, waves_per_simd waves per SIMD
    probe<MODE, CH><<<256, 256 * waves_per_simd>>>(out, iters, cyc);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on gfx9? report raw ticks per instruction per SIMD and derive from wall time too
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<MODE, CH><<<256, 256 * waves_per_simd>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_per_simd = (double)iters * 8 * CH * waves_per_simd;
    printf("%-28s chains=%d waves/SIMD=%d: %.2f ns per instr per SIMD (%.1f cycles @2.4GHz), memtime ticks/instr %.2f\n", name, CH,
           waves_per_simd, ms * 1e6 / n_per_simd, ms * 1e6 / n_per_simd * 2.4, (double)h / n_per_simd * waves_per_simd);
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<0, 1>("4x4x1", 1); run<0, 3>("4x4x1", 1); run<0, 6>("4x4x1", 1); run<0, 3>("4x4x1", 2); run<0, 3>("4x4x1", 4);
    run<1, 1>("4x4x1 cbsz=4", 1); run<1, 3>("4x4x1 cbsz=4", 1); run<1, 6>("4x4x1 cbsz=4", 1); run<1, 3>("4x4x1 cbsz=4", 2); run<1, 3>("4x4x1 cbsz=4", 4);
    run<2, 1>("16x16x4", 1); run<2, 3>("16x16x4", 1); run<2, 3>("16x16x4", 2); run<2, 3>("16x16x4", 4);
    return 0;
}
