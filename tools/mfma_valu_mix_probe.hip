// What does a block of VALU instructions cost right after a run of v_mfma_f32_4x4x1 in the SAME wave?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_valu_mix_probe.hip -o tools/mfma_valu_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vadd(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// MODE bit0: MFMAs, bit1: 36 v_max on the MFMA results + others, bit2: v_add instead of v_max, bit3: the VALU block reads only non-MFMA registers
template <int MODE>
__global__ void probe(float *out, int iters, unsigned long long *cyc)
{
    f32x4 acc[3];
    for (int c = 0; c < 3; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m[36], o[12];
    for (int i = 0; i < 36; ++i) m[i] = -1e30f;
    for (int i = 0; i < 12; ++i) o[i] = threadIdx.x * 0.01f + i;
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE & 1) {
#pragma unroll
            for (int u = 0; u < 36; ++u)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 4, 5, 0);
        }
        if constexpr (MODE & 2) {
#pragma unroll
            for (int j = 0; j < 36; ++j) {
                const float src = (MODE & 8) ? o[j % 12] : (j < 12 ? acc[j / 4][j % 4] : o[j % 12]);
                m[j] = (MODE & 4) ? vadd(m[j], src) : vmax(m[j], src);
            }
        }
        asm volatile("" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < 3; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    for (int i = 0; i < 36; ++i) s += m[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
void run(const char *name, int wps)
{
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    probe<MODE><<<256, 256 * wps>>>(out, iters, cyc);
    probe<MODE><<<256, 256 * wps>>>(out, iters, cyc);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<MODE><<<256, 256 * wps>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s waves/SIMD=%d: %.0f ticks per iteration (wave 0), %.0f ns of SIMD time per wave-iteration\n", name, wps,
           (double)h / iters, ms * 1e6 / iters / wps);
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<1>("108 MFMA", 1); run<2>("36 v_max (12 on acc)", 1); run<3>("108 MFMA + 36 v_max (12 read MFMA results)", 1);
    run<11>("108 MFMA + 36 v_max (none reads MFMA results)", 1); run<7>("108 MFMA + 36 v_add (12 read MFMA results)", 1);
    run<3>("108 MFMA + 36 v_max (12 read MFMA results)", 2); run<3>("108 MFMA + 36 v_max (12 read MFMA results)", 4);
    run<1>("108 MFMA", 4); run<1>("108 MFMA", 2); run<2>("36 v_max (12 on acc)", 4); run<11>("108 MFMA + 36 v_max (none reads MFMA results)", 4);
    return 0;
}
