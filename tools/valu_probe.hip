// valu_probe.hip -- development micro-benchmark (GPU box): issue cost of the VALU instructions the kNN
// graph kernel is made of, with 1..8 waves per SIMD.  Cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float a[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = seed * (i + threadIdx.x);
    float d = seed;
    int cnt = 0;
    uint32_t w = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {                 // 32 independent fma
#pragma unroll
            for (int i = 0; i < 32; ++i) a[i] = __builtin_fmaf(a[i], d, 1.0f);
        } else if (MODE == 1) {          // the med3 insertion chain (each reads its lower neighbour)
#pragma unroll
            for (int i = 31; i > 0; --i) a[i] = __builtin_amdgcn_fmed3f(a[i - 1], d, a[i]);
            a[0] = fminf(a[0], d);
            d += 1.0f;
        } else if (MODE == 2) {          // 32 dependent fma (one chain)
#pragma unroll
            for (int i = 0; i < 32; ++i) d = __builtin_fmaf(a[i], d, 1.0f);
        } else if (MODE == 3) {          // v_cmp to SGPR + v_addc from SGPR, 8 each
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint64_t m = __builtin_amdgcn_ballot_w64(a[i] < d);
                uint64_t co;
                asm volatile("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(w), "=&s"(co) : "v"(w), "s"(m));
            }
            d += 1.0f;
        } else if (MODE == 4) {          // compare-exchange network layer: 16 x (v_min, v_max)
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                float lo, hi;
                asm("v_min_f32 %0, %1, %2" : "=v"(lo) : "v"(a[i]), "v"(a[i + 1]));
                asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(a[i]), "v"(a[i + 1]));
                a[i] = hi; a[i + 1] = lo;
            }
        } else if (MODE == 5) {          // v_cmp -> SGPR pair only (result or-ed on the scalar unit)
            uint64_t m = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                m |= __builtin_amdgcn_ballot_w64(a[i] < d);
            cnt += (int)(m & 1);
            d += 1.0f;
        } else if (MODE == 6) {          // v_fmac with three VGPR sources
#pragma unroll
            for (int i = 0; i < 32; ++i) a[i] = __builtin_fmaf(a[(i + 1) & 31], d, a[i]);
        } else if (MODE == 7) {          // v_cndmask + v_or on a VCC compare (3 VALU per bit)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                w |= a[i] < d ? (1u << i) : 0u;
            d += 1.0f;
        } else if (MODE == 8) {          // sign bit trick: v_sub + v_alignbit
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float r = a[i] - d;
                w = __builtin_amdgcn_alignbit(w, __float_as_uint(r), 31);
            }
            d += 1.0f;
        }
    }
    float s = d + w + cnt;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int per_iter)
{
    float *out;
    hipMalloc(&out, 256 * 8 * 1024 * 4 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps : {1, 2, 4, 8}) {       // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * 2.4e9;
        printf("%-28s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD\n", name, wps,
               cyc / ((double)iters * per_iter * wps));
    }
    hipFree(out);
}

int main()
{
    run<0>("v_fma_f32 independent", 32);
    run<1>("v_med3_f32 chain", 32);
    run<2>("v_fma_f32 dependent", 32);
    run<3>("v_cmp->sgpr + v_addc", 16);
    run<4>("v_min + v_max (asm) layer", 32);
    run<5>("v_cmp -> sgpr", 16);
    run<6>("v_fmac 3 vgpr", 32);
    run<7>("v_cmp vcc+cndmask+or", 24);
    run<8>("v_sub + v_alignbit", 32);
    return 0;
}
