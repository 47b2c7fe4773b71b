// fps_bucket.h -- declarations shared by the bucketed FPS kernels (fps_bucket.hip: setup kernels, the
// single-workgroup forms and the dispatcher; fps_cluster.hip: the multi-workgroup tile form).
#pragma once
#include "tpu3_dev.h"

#include <hip/hip_fp16.h>

namespace {

constexpr int FL_R = 16;                 // tile forms: points per bucket
constexpr int FL_TP = 64 * FL_R;         // tile forms: points per tile

// Arguments of batch element 0; fb_elem() derives element i.  User arrays are dense (b, n, ...)
// slabs, the per-element workspace arrays repeat every `per_elem` bytes, the sort arrays every
// `sort_stride` words.
struct FbArgs {
    int n, m, nb, nbpad, npad, ng;      // n, m: slab strides = upper bounds of the live sizes
    int ncell;                          // entries of the main kernel's LDS table: nbpad, or nbpad / 16 (three levels)
    int bsz, lb;                        // points per bucket; log2 of the tie-rule block size
    const int32_t *n_arr, *m_arr;       // live sizes per element, or null
    const float *xyz;     // (n,3) original order
    float *temp;          // (n)
    int32_t *idx;         // (m)
    float4 *sp;           // (npad) Morton order: x, y, z, running distance
    uint32_t *skey;       // (npad) tie key of the original index (0xFFFFFFFF = padding)
    uint32_t *ib;         // (9, nbpad) initial bucket table: max, key, x, y, z, box0, box1, box2, runner-up
    float *bbox;          // (8)
    size_t per_elem;
    size_t sort_stride;
    unsigned long long *prof;   // PROF builds only
    // (r3) tile form (fl_main_kernel): a bucket = 16 consecutive slots of sp / skey, a tile = 64 buckets; one record
    // per bucket and per tile
    int fl, ntile;              // fl != 0: tile form; tiles of the slab (upper bound of the live count)
    uint4 *rec;                 // (ntile * 64) bucket records: fp16 box (3 words) | runner-up distance bits
    int32_t *bm0;               // (ntile * 64) initial bucket maxima (distance bits)
    uint8_t *ba0;               // (ntile * 64) initial position (0..15) of the bucket's best point
    float *tt;                  // (ntile, 8) tile records: box lo.xyz hi.xyz, max bits, runner-up bits
};

// element i of the batch: pointers advanced, n / m / nb / lb replaced by the element's live values
__device__ __forceinline__ FbArgs fb_elem(const FbArgs &a0, int i)
{
    FbArgs a = a0;
    a.xyz = a0.xyz + (size_t)i * a0.n * 3;
    a.temp = a0.temp + (size_t)i * a0.n;
    a.idx = a0.idx + (size_t)i * a0.m;
    a.sp = (float4 *)((char *)a0.sp + (size_t)i * a0.per_elem);
    a.skey = (uint32_t *)((char *)a0.skey + (size_t)i * a0.per_elem);
    a.ib = (uint32_t *)((char *)a0.ib + (size_t)i * a0.per_elem);
    a.bbox = (float *)((char *)a0.bbox + (size_t)i * a0.per_elem);
    if (a0.fl) {
        a.rec = (uint4 *)((char *)a0.rec + (size_t)i * a0.per_elem);
        a.bm0 = (int32_t *)((char *)a0.bm0 + (size_t)i * a0.per_elem);
        a.ba0 = (uint8_t *)((char *)a0.ba0 + (size_t)i * a0.per_elem);
        a.tt = (float *)((char *)a0.tt + (size_t)i * a0.per_elem);
    }
    if (a0.n_arr) {
        a.n = min(max(a0.n_arr[i], 0), a0.n);
        a.nb = (a.n + a0.bsz - 1) / a0.bsz;
        a.lb = tpu3_fps_log2_bs(a.n);
    }
    if (a0.m_arr)
        a.m = min(max(a0.m_arr[i], 0), a0.m);
    return a;
}

__device__ __forceinline__ float fb_half_lo(uint32_t w) { return __half2float(__ushort_as_half((unsigned short)(w & 0xFFFFu))); }
__device__ __forceinline__ float fb_half_hi(uint32_t w) { return __half2float(__ushort_as_half((unsigned short)(w >> 16))); }

__device__ __forceinline__ float fb_dbox(float qx, float qy, float qz, float lx, float ly, float lz, float hx,
                                         float hy, float hz)
{
    const float dx = fmaxf(fmaxf(lx - qx, qx - hx), 0.f);
    const float dy = fmaxf(fmaxf(ly - qy, qy - hy), 0.f);
    const float dz = fmaxf(fmaxf(lz - qz, qz - hz), 0.f);
    return tpu3_sqdist3(dx, dy, dz);
}


} // namespace

// fps_cluster.hip: the tile form on `g` workgroups per element (g = 2, 4, 8 or 16; b * g workgroups that must all be
// resident: the caller keeps b * g small).  `mbox`: b * tpu3_fps_cluster_mailbox_bytes(g) bytes, ZEROED on the stream
// before the launch.  `stats`: optional 8 device words (rounds, samples, tie exchanges, poll sweeps, timeouts, ...).
size_t tpu3_fps_cluster_mailbox_bytes(int g);
size_t tpu3_fps_cluster_lds_bytes(int ntile, int g);
int tpu3_fps_cluster_launch(hipStream_t s, int b, int g, const void *fb_args, void *mbox, unsigned long long *stats);
