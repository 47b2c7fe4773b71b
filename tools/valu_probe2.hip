// valu_probe2.hip -- development micro-benchmark (GPU box): steady-state issue cost of single VALU
// instruction forms on gfx950, hand-placed registers (.irp blocks of 16 independent instructions,
// 4 blocks per loop trip), 1/2/4/8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REGS "40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55"
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v60","v61","v62","v63","vcc","s20","s21","s22","s23"
#define BODY(INS) asm volatile(".irp r," REGS "\n " INS "\n .endr\n .irp r," REGS "\n " INS "\n .endr\n" \
                               ".irp r," REGS "\n " INS "\n .endr\n .irp r," REGS "\n " INS "\n .endr\n" ::: CLOB)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    asm volatile("v_mov_b32 v60, 1.0\n v_mov_b32 v61, 2.0\n v_mov_b32 v62, 3.0\n v_mov_b32 v63, 0" ::: CLOB);
    asm volatile(".irp r," REGS "\n v_mov_b32 v\\r, 1.0\n .endr" ::: CLOB);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) BODY("v_add_f32 v\\r, v60, v61");
        if (MODE == 1) BODY("v_min_f32 v\\r, v60, v61");
        if (MODE == 2) BODY("v_fma_f32 v\\r, v60, v61, v62");
        if (MODE == 3) BODY("v_fmac_f32 v\\r, v60, v61");
        if (MODE == 4) BODY("v_med3_f32 v\\r, v60, v61, v62");
        if (MODE == 5) BODY("v_med3_f32 v\\r, v60, v61, 1.0");
        if (MODE == 6) BODY("v_cmp_lt_f32 vcc, v\\r, v60");
        if (MODE == 7) BODY("v_cmp_lt_f32 s[20:21], v\\r, v60");
        if (MODE == 8) BODY("v_addc_co_u32 v\\r, s[22:23], v60, v61, s[20:21]");
        if (MODE == 9) BODY("v_mov_b32 v\\r, v60");
        if (MODE == 10) BODY("v_min_f32 v\\r, v\\r, v61");
        if (MODE == 11) BODY("v_med3_f32 v\\r, v\\r, v61, v62");
        if (MODE == 12) BODY("v_max_f32 v\\r, v60, v62");     // same-bank sources (60, 62 differ by 2)
        if (MODE == 13) BODY("v_max_f32 v\\r, v60, v60");
        if (MODE == 14) BODY("v_cndmask_b32 v\\r, v60, v61, vcc");
        if (MODE == 15) BODY("v_alignbit_b32 v\\r, v60, v61, 31");
        if (MODE == 16) BODY("v_pk_fma_f32 v[40:41], v[60:61], v[60:61], v[62:63]");
    }
    float r;
    asm volatile("v_mov_b32 %0, v40" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char *name)
{
    float *out;
    (void)hipMalloc(&out, 256 * 8 * 1024 * 4 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 10000;
    printf("%-44s", name);
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  %d w/SIMD: %5.2f", wps, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * wps));
    }
    printf("   cycles per wave-instruction\n");
    (void)hipFree(out);
}

int main()
{
    run<0>("v_add_f32 d, a, b");
    run<1>("v_min_f32 d, a, b");
    run<12>("v_max_f32 d, a, c (sources 2 regs apart)");
    run<13>("v_max_f32 d, a, a");
    run<10>("v_min_f32 d, d, b (own chain per reg)");
    run<2>("v_fma_f32 d, a, b, c");
    run<3>("v_fmac_f32 d, a, b");
    run<4>("v_med3_f32 d, a, b, c");
    run<5>("v_med3_f32 d, a, b, 1.0");
    run<11>("v_med3_f32 d, d, b, c");
    run<6>("v_cmp_lt_f32 vcc, d, a");
    run<7>("v_cmp_lt_f32 s[20:21], d, a");
    run<8>("v_addc_co_u32 d, s[22:23], a, b, s[20:21]");
    run<9>("v_mov_b32 d, a");
    run<14>("v_cndmask_b32 d, a, b, vcc");
    run<15>("v_alignbit_b32 d, a, b, 31");
    run<16>("v_pk_fma_f32 (2 lanes-ops per instr)");
    return 0;
}
