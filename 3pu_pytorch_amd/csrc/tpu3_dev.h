// tpu3_dev.h -- device-side helpers shared by the gfx950 kernels of lib3pu_hip.so.
// CDNA4 only: 64-lane wavefronts, DPP row operations, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tpu3.h"

#define TPU3_WAVE 64

// The library is compiled with -ffp-contract=off: every fused multiply-add below is explicit,
// so the arithmetic is the oracle's (oracle/ref_kernels.c) operation for operation.

// Squared distance in the association nvcc's -fmad=true contraction produces for
// dx*dx + dy*dy + dz*dz (sampling_cuda.cu:143, nmdistance_cuda.cu:33): fma(dz,dz,fma(dx,dx,dy*dy)).
__device__ __forceinline__ float tpu3_sqdist3(float dx, float dy, float dz)
{
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
}
__device__ __forceinline__ double tpu3_sqdist3(double dx, double dy, double dz)
{
    return __builtin_fma(dz, dz, __builtin_fma(dx, dx, dy * dy));
}

// min / max of two floats as ONE instruction.  fminf() / fmaxf() lower to minnum / maxnum, and with the IEEE mode bit
// set the compiler quiets every operand it cannot prove canonical first (v_max_f32 v, v, v): a value that came out of a
// load, a lane exchange or a bit cast costs three half-rate instructions per minimum instead of one (round 5: 260 of
// the 335 v_max_f32 of the per-level FPS kernel were these).  v_min_f32 / v_max_f32 return the other operand for a
// quiet NaN like fminf / fmaxf do; results for non-NaN inputs are the same bits.
__device__ __forceinline__ float tpu3_min1(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float tpu3_max1(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Order-preserving float -> u32 map (ascending float order == ascending unsigned order),
// -0.0 is folded onto +0.0 first.  NaNs land above +inf.
__device__ __forceinline__ uint32_t tpu3_mono(float f)
{
    const uint32_t u = __float_as_uint(f + 0.0f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float tpu3_unmono(uint32_t m)
{
    return __uint_as_float(m ^ ((m >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// ---- DPP cross-lane primitives (gfx9 DPP encodings; wave64) --------------------------------
// quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror 0x141, row_mirror 0x140,
// row_bcast15 0x142 (lane 15 of a row -> next row), row_bcast31 0x143 (lane 31 -> rows 2,3).
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int tpu3_dpp(int v)
{
    // old = v: lanes without a valid source keep their own value (identity for min/max).
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);
}

// max over the 16 lanes of each row; every lane of the row ends with the row's result
__device__ __forceinline__ int tpu3_row_max_i32(int v)
{
    v = max(v, tpu3_dpp<0xB1>(v));
    v = max(v, tpu3_dpp<0x4E>(v));
    v = max(v, tpu3_dpp<0x141>(v));
    v = max(v, tpu3_dpp<0x140>(v));
    return v;
}
__device__ __forceinline__ uint32_t tpu3_row_min_u32(uint32_t v)
{
    v = min(v, (uint32_t)tpu3_dpp<0xB1>((int)v));
    v = min(v, (uint32_t)tpu3_dpp<0x4E>((int)v));
    v = min(v, (uint32_t)tpu3_dpp<0x141>((int)v));
    v = min(v, (uint32_t)tpu3_dpp<0x140>((int)v));
    return v;
}
__device__ __forceinline__ uint32_t tpu3_row_max_u32(uint32_t v)
{
    v = max(v, (uint32_t)tpu3_dpp<0xB1>((int)v));
    v = max(v, (uint32_t)tpu3_dpp<0x4E>((int)v));
    v = max(v, (uint32_t)tpu3_dpp<0x141>((int)v));
    v = max(v, (uint32_t)tpu3_dpp<0x140>((int)v));
    return v;
}

// whole-wave reductions; the result is returned wave-uniform (an SGPR after readlane)
__device__ __forceinline__ int tpu3_wave_max_i32(int v)
{
    v = tpu3_row_max_i32(v);
    v = max(v, tpu3_dpp<0x142, 0xA>(v));
    v = max(v, tpu3_dpp<0x143, 0xC>(v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t tpu3_wave_min_u32(uint32_t v)
{
    v = tpu3_row_min_u32(v);
    v = min(v, (uint32_t)tpu3_dpp<0x142, 0xA>((int)v));
    v = min(v, (uint32_t)tpu3_dpp<0x143, 0xC>((int)v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t tpu3_wave_max_u32(uint32_t v)
{
    v = tpu3_row_max_u32(v);
    v = max(v, (uint32_t)tpu3_dpp<0x142, 0xA>((int)v));
    v = max(v, (uint32_t)tpu3_dpp<0x143, 0xC>((int)v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ float tpu3_wave_max_f32(float v)
{
    // generic float max through the order-preserving map
    return tpu3_unmono(tpu3_wave_max_u32(tpu3_mono(v)));
}
__device__ __forceinline__ float tpu3_wave_sum_f32(float v)
{
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

// ---- latency-tuned reductions for the FPS round loops ----------------------------------------------
// hipcc lowers the update_dpp + max pairs above to v_mov_dpp / v_max / v_mov / s_nop (4 issue slots
// per step).  The FPS kernels are one long dependent chain per round, so here the DPP modifier is
// fused into the max itself (one instruction + the 2 wait states a DPP read of a freshly written
// VGPR needs), and a two-value form interleaves two independent chains so each fills the other's
// wait states.
#define TPU3_DPP_MAX_I32(reg, ctrl) "v_max_i32_dpp " reg ", " reg ", " reg " " ctrl "\n\t"
__device__ __forceinline__ int tpu3_wave_max_i32_fast(int v)
{
    asm volatile("s_nop 1\n\t"
                 TPU3_DPP_MAX_I32("%0", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") "s_nop 1\n\t"
                 TPU3_DPP_MAX_I32("%0", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf") "s_nop 1\n\t"
                 TPU3_DPP_MAX_I32("%0", "row_half_mirror row_mask:0xf bank_mask:0xf") "s_nop 1\n\t"
                 TPU3_DPP_MAX_I32("%0", "row_mirror row_mask:0xf bank_mask:0xf") "s_nop 1\n\t"
                 TPU3_DPP_MAX_I32("%0", "row_bcast:15 row_mask:0xa bank_mask:0xf") "s_nop 1\n\t"
                 TPU3_DPP_MAX_I32("%0", "row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1\n\t"
                 : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ void tpu3_wave_max_i32_fast_x2(int &a, int &b)
{
    asm volatile("s_nop 1\n\t"
                 TPU3_DPP_MAX_I32("%0", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 TPU3_DPP_MAX_I32("%1", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") "s_nop 0\n\t"
                 TPU3_DPP_MAX_I32("%0", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 TPU3_DPP_MAX_I32("%1", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf") "s_nop 0\n\t"
                 TPU3_DPP_MAX_I32("%0", "row_half_mirror row_mask:0xf bank_mask:0xf")
                 TPU3_DPP_MAX_I32("%1", "row_half_mirror row_mask:0xf bank_mask:0xf") "s_nop 0\n\t"
                 TPU3_DPP_MAX_I32("%0", "row_mirror row_mask:0xf bank_mask:0xf")
                 TPU3_DPP_MAX_I32("%1", "row_mirror row_mask:0xf bank_mask:0xf") "s_nop 0\n\t"
                 TPU3_DPP_MAX_I32("%0", "row_bcast:15 row_mask:0xa bank_mask:0xf")
                 TPU3_DPP_MAX_I32("%1", "row_bcast:15 row_mask:0xa bank_mask:0xf") "s_nop 0\n\t"
                 TPU3_DPP_MAX_I32("%0", "row_bcast:31 row_mask:0xc bank_mask:0xf")
                 TPU3_DPP_MAX_I32("%1", "row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1\n\t"
                 : "+v"(a), "+v"(b));
    a = __builtin_amdgcn_readlane(a, 63);
    b = __builtin_amdgcn_readlane(b, 63);
}
// four independent chains: each chain's DPP read is three instructions behind its write, no wait states needed
__device__ __forceinline__ void tpu3_wave_max_i32_fast_x4(int &a, int &b, int &c, int &d)
{
#define TPU3_X4(ctrl) TPU3_DPP_MAX_I32("%0", ctrl) TPU3_DPP_MAX_I32("%1", ctrl) TPU3_DPP_MAX_I32("%2", ctrl) TPU3_DPP_MAX_I32("%3", ctrl)
    asm volatile("s_nop 1\n\t"
                 TPU3_X4("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 TPU3_X4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 TPU3_X4("row_half_mirror row_mask:0xf bank_mask:0xf")
                 TPU3_X4("row_mirror row_mask:0xf bank_mask:0xf")
                 TPU3_X4("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 TPU3_X4("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1\n\t"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef TPU3_X4
    a = __builtin_amdgcn_readlane(a, 63);
    b = __builtin_amdgcn_readlane(b, 63);
    c = __builtin_amdgcn_readlane(c, 63);
    d = __builtin_amdgcn_readlane(d, 63);
}
// max over the 16 lanes of row 0 only (cross-wave slots); result valid in every lane of row 0
__device__ __forceinline__ int tpu3_row_max_i32_fast(int v)
{
    asm volatile("s_nop 1\n\t"
                 TPU3_DPP_MAX_I32("%0", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") "s_nop 1\n\t"
                 TPU3_DPP_MAX_I32("%0", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf") "s_nop 1\n\t"
                 TPU3_DPP_MAX_I32("%0", "row_half_mirror row_mask:0xf bank_mask:0xf") "s_nop 1\n\t"
                 TPU3_DPP_MAX_I32("%0", "row_mirror row_mask:0xf bank_mask:0xf") "s_nop 1\n\t"
                 : "+v"(v));
    return v;
}

// Wave arg-max with the FPS tie rule: value = signed-int order of the distance bits, ties (rare:
// duplicated points) to the smallest key.  Returns the maximum; `lane` = the one winning lane.
__device__ __forceinline__ int tpu3_wave_argmax(int bits, uint32_t key, int &lane)
{
    const int wmax = tpu3_wave_max_i32_fast(bits);
    unsigned long long tie = __ballot(bits == wmax);
    if (__builtin_popcountll(tie) != 1) {
        const uint32_t wkey = tpu3_wave_min_u32(bits == wmax ? key : 0xFFFFFFFFu);
        tie = __ballot(bits == wmax && key == wkey);
    }
    lane = __builtin_ctzll(tie);
    return wmax;
}

// FPS tie key (see tpu3.h): lexicographic (k mod bs, k div bs) packed so that unsigned order
// is the reference's winner order among equal distances; bs = 1 << lb <= 512.
__device__ __forceinline__ uint32_t tpu3_fps_tiekey(int k, int lb)
{
    return ((uint32_t)(k & ((1 << lb) - 1)) << 22) | (uint32_t)(k >> lb);
}
__device__ __forceinline__ int tpu3_fps_tiekey_to_index(uint32_t key, int lb)
{
    return (int)(((key & 0x3FFFFFu) << lb) | (key >> 22));
}
// log2 of the reference's block size for n points: largest power of two <= n, at most 512
// (sampling/cuda_utils.h:9-14; equal to the double-log formula for every n, tests pin it).
__host__ __device__ __forceinline__ int tpu3_fps_log2_bs(int n)
{
    int lb = 0;
    while ((2 << lb) <= n && lb < 9)
        ++lb;
    return lb;
}

static inline int tpu3_launch_status()
{
    return (int)hipGetLastError();
}
