// mfma_overlap_probe.hip -- development micro-benchmark (GPU box): do v_mfma_f32_4x4x1 and VALU
// instructions of the two issue classes (v_fma_f32: 2 cycles, v_min_f32: 4 cycles) overlap on one SIMD,
// (a) from different waves, (b) interleaved in one wave?  8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REGS "40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55"
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"
#define MF "v_mfma_f32_4x4x1_16b_f32 v[64:67], v60, v61, v[64:67]\n v_mfma_f32_4x4x1_16b_f32 v[68:71], v60, v61, v[68:71]\n v_mfma_f32_4x4x1_16b_f32 v[72:75], v60, v61, v[72:75]\n v_mfma_f32_4x4x1_16b_f32 v[76:79], v60, v61, v[76:79]\n"
#define MF16 MF MF MF MF
#define MFB(a) "v_mfma_f32_4x4x1_16b_f32 v[" #a ":" #a "+3], v60, v61, v[" #a ":" #a "+3]\n"
#define MF8 MFB(64) MFB(68) MFB(72) MFB(76) MFB(80) MFB(84) MFB(88) MFB(92)
#define MF16A MF8 MFB(96) MFB(100) MFB(104) MFB(108) MFB(112) MFB(116) MFB(120) MFB(124)
#define V16(INS) ".irp r," REGS "\n " INS "\n .endr\n"

// what: 0 = 64 MFMA, 1 = 64 v_min, 2 = 64 v_fma, 3 = 64 MFMA + 64 v_min interleaved 4:4, 4 = 64 MFMA + 64 v_fma interleaved
template <int A, int B>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    asm volatile("v_mov_b32 v60, 1.0\n v_mov_b32 v61, 2.0\n v_mov_b32 v62, 3.0\n v_mov_b32 v63, 0" ::: CLOB);
    asm volatile(".irp r," REGS ",64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79\n v_mov_b32 v\\r, 1.0\n .endr" ::: CLOB);
    const int what = ((blockIdx.x >> 3) & 1) ? B : A;      // blocks are dealt to the 8 XCDs round-robin: alternate per XCD-local block
    for (int it = 0; it < iters; ++it) {
        if (what == 0) asm volatile(MF16 MF16 MF16 MF16 ::: CLOB);
        if (what == 5) asm volatile(MF8 MF8 MF8 MF8 MF8 MF8 MF8 MF8 ::: CLOB);
        if (what == 6) asm volatile(MF16A MF16A MF16A MF16A ::: CLOB);
        if (what == 7) asm volatile(".rept 64\n v_mfma_f32_4x4x1_16b_f32 v[64:67], v60, v61, v[64:67]\n .endr\n" ::: CLOB);
        if (what == 8) asm volatile(".rept 4\n v_mfma_f32_16x16x4_f32 v[64:67], v60, v61, v[64:67]\n v_mfma_f32_16x16x4_f32 v[68:71], v60, v61, v[68:71]\n v_mfma_f32_16x16x4_f32 v[72:75], v60, v61, v[72:75]\n v_mfma_f32_16x16x4_f32 v[76:79], v60, v61, v[76:79]\n .endr\n" ::: CLOB);
        if (what == 9) asm volatile(".irp r,40,44,48,52,41,45,49,53,42,46,50,54,43,47,51,55\n v_mfma_f32_16x16x4_f32 v[64:67], v60, v61, v[64:67]\n v_min_f32 v\\r, v60, v61\n v_min_f32 v\\r, v60, v62\n v_min_f32 v\\r, v61, v62\n v_min_f32 v\\r, v62, v61\n .endr\n" ::: CLOB);
        if (what == 1) asm volatile(V16("v_min_f32 v\\r, v60, v61") V16("v_min_f32 v\\r, v60, v61") V16("v_min_f32 v\\r, v60, v61") V16("v_min_f32 v\\r, v60, v61") ::: CLOB);
        if (what == 2) asm volatile(V16("v_fma_f32 v\\r, v60, v61, v62") V16("v_fma_f32 v\\r, v60, v61, v62") V16("v_fma_f32 v\\r, v60, v61, v62") V16("v_fma_f32 v\\r, v60, v61, v62") ::: CLOB);
        if (what == 3) asm volatile(".irp r,40,44,48,52,41,45,49,53,42,46,50,54,43,47,51,55\n" MF " v_min_f32 v\\r, v60, v61\n v_min_f32 v\\r, v60, v62\n v_min_f32 v\\r, v61, v62\n v_min_f32 v\\r, v62, v61\n .endr\n" ::: CLOB);
        if (what == 4) asm volatile(".irp r,40,44,48,52,41,45,49,53,42,46,50,54,43,47,51,55\n" MF " v_fma_f32 v\\r, v60, v61, v62\n v_fma_f32 v\\r, v60, v62, v61\n v_fma_f32 v\\r, v61, v62, v60\n v_fma_f32 v\\r, v62, v61, v60\n .endr\n" ::: CLOB);
    }
    float r;
    asm volatile("s_nop 7\n s_nop 7\n v_add_f32 %0, v40, v64" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int A, int B>
void runw(const char *name)
{
    float *out;
    (void)hipMalloc(&out, 256 * 8 * 1024 * 4 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-44s", name);
    for (int wps : {1, 2, 4, 8}) {
        const int iters = 5000, blocks = 256 * wps;
        hipLaunchKernelGGL((k<A, B>), dim3(blocks), dim3(256), 0, 0, out, 100);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<A, B>), dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  %d w/SIMD: %6.2f", wps, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * wps));
    }
    printf("   cycles per MFMA per SIMD\n");
    (void)hipFree(out);
}

template <int A, int B>
void run(const char *name)
{
    float *out;
    (void)hipMalloc(&out, 256 * 8 * 1024 * 4 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 5000, wps = 8, blocks = 256 * wps;
    hipLaunchKernelGGL((k<A, B>), dim3(blocks), dim3(256), 0, 0, out, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<A, B>), dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-60s %8.1f cycles per SIMD per loop trip of all 8 waves\n", name, ms * 1e-3 * 2.4e9 / iters);
    (void)hipFree(out);
}

int main()
{
    runw<7, 7>("MFMA 4x4x1, 1 accumulator");
    runw<0, 0>("MFMA 4x4x1, 4 accumulators");
    runw<5, 5>("MFMA 4x4x1, 8 accumulators");
    runw<6, 6>("MFMA 4x4x1, 16 accumulators");
    run<8, 8>("8 waves x 16 MFMA 16x16x4");
    run<9, 9>("8 waves x (16 MFMA 16x16x4 + 64 v_min interleaved 1:4)");
    run<8, 1>("4 waves x 16 MFMA 16x16x4 | 4 waves x 64 v_min");
    run<0, 0>("8 waves x 64 MFMA 4x4x1");
    run<1, 1>("8 waves x 64 v_min");
    run<2, 2>("8 waves x 64 v_fma");
    run<0, 1>("4 waves x 64 MFMA | 4 waves x 64 v_min");
    run<0, 2>("4 waves x 64 MFMA | 4 waves x 64 v_fma");
    run<3, 3>("8 waves x (64 MFMA + 64 v_min interleaved)");
    run<4, 4>("8 waves x (64 MFMA + 64 v_fma interleaved)");
    return 0;
}
