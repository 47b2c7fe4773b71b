// valu_probe3.hip -- development micro-benchmark (GPU box): issue cost of the INTEGER min / max / compare forms a
// compare-exchange network can be built from (same harness as valu_probe2.hip).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REGS "40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55"
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v60","v61","v62","v63","vcc","s20","s21","s22","s23"
#define BODY(INS) asm volatile(".irp r," REGS "\n " INS "\n .endr\n .irp r," REGS "\n " INS "\n .endr\n" \
                               ".irp r," REGS "\n " INS "\n .endr\n .irp r," REGS "\n " INS "\n .endr\n" ::: CLOB)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    asm volatile("v_mov_b32 v60, 1\n v_mov_b32 v61, 2\n v_mov_b32 v62, 3\n v_mov_b32 v63, 0" ::: CLOB);
    asm volatile(".irp r," REGS "\n v_mov_b32 v\\r, 1\n .endr" ::: CLOB);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) BODY("v_min_i32 v\\r, v60, v61");
        if (MODE == 1) BODY("v_max_i32 v\\r, v60, v61");
        if (MODE == 2) BODY("v_min_u32 v\\r, v60, v61");
        if (MODE == 3) BODY("v_max_u32 v\\r, v60, v61");
        if (MODE == 4) BODY("v_min3_i32 v\\r, v60, v61, v62");
        if (MODE == 5) BODY("v_med3_i32 v\\r, v60, v61, v62");
        if (MODE == 6) BODY("v_add_u32 v\\r, v60, v61");
        if (MODE == 7) BODY("v_sub_u32 v\\r, v60, v61");
        if (MODE == 8) BODY("v_and_b32 v\\r, v60, v61");
        if (MODE == 9) BODY("v_xor_b32 v\\r, v60, v61");
        if (MODE == 10) BODY("v_bfi_b32 v\\r, v60, v61, v62");
        if (MODE == 11) BODY("v_ashrrev_i32 v\\r, 31, v60");
        if (MODE == 12) BODY("v_pk_min_i16 v\\r, v60, v61");
        if (MODE == 13) BODY("v_pk_max_u16 v\\r, v60, v61");
        if (MODE == 14) BODY("v_min_i16 v\\r, v60, v61");
        if (MODE == 15) BODY("v_lshl_add_u32 v\\r, v60, 2, v61");
        if (MODE == 16) BODY("v_and_or_b32 v\\r, v60, v61, v62");
        if (MODE == 17) BODY("v_xad_u32 v\\r, v60, v61, v62");
        if (MODE == 18) BODY("v_add3_u32 v\\r, v60, v61, v62");
        if (MODE == 19) BODY("v_sub_f32 v\\r, v60, v61");
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
    printf("%-36s", name);
    for (int wps : {1, 4, 8}) {
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
    run<0>("v_min_i32"); run<1>("v_max_i32"); run<2>("v_min_u32"); run<3>("v_max_u32"); run<4>("v_min3_i32");
    run<5>("v_med3_i32"); run<6>("v_add_u32"); run<7>("v_sub_u32"); run<8>("v_and_b32"); run<9>("v_xor_b32");
    run<10>("v_bfi_b32"); run<11>("v_ashrrev_i32"); run<12>("v_pk_min_i16"); run<13>("v_pk_max_u16"); run<14>("v_min_i16");
    run<15>("v_lshl_add_u32"); run<16>("v_and_or_b32"); run<17>("v_xad_u32"); run<18>("v_add3_u32"); run<19>("v_sub_f32");
    return 0;
}
