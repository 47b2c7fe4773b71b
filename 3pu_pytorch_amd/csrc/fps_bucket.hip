// fps_bucket.hip -- exact, work-skipping farthest-point sampling for large point sets (gfx950).
//
// The reference's FPS (sampling/sampling_cuda.cu:103-174) re-reads all n points every round:
// 239 616 points x 80 000 rounds for the final resample of a 5000 -> 80 000 cloud.  A round only
// CHANGES the running distance of points closer to the new sample than to every earlier one, and
// those are confined to a shrinking ball around it.  This kernel produces bit-identical indices
// (and identical final `temp`) while touching only that ball:
//
//   setup   points are sorted along a 30-bit Morton curve (rocPRIM radix sort) and cut into
//           BUCKETS of 64*PPL consecutive points; 16 consecutive buckets form a GROUP (one DPP row).
//           Every bucket / group knows an AABB and its current (max distance, tie key, xyz of that
//           point).  Bucket table + fp16 (outward-rounded) bucket boxes live in LDS, group boxes in
//           the owner lane's registers, the group table in LDS.
//   round   1. every lane tests the groups it owns:  dbox(sample, AABB) >= max  ==> nothing inside
//              can change (dbox uses the fp32 association of the point distance and fp32 rounding is
//              monotone, so dbox <= d(p) for every p inside: min(d(p), temp[p]) == temp[p] exactly);
//           2. the 16 children of each touched group are tested the same way by one DPP row (up to 4
//              groups per wave instruction);
//           3. touched buckets are re-scanned two at a time (64 lanes = 64 points; both buckets'
//              loads in flight together, the two reduction chains interleaved);
//           4. the touched groups' entries are rebuilt by a 16-lane row arg-max;
//           5. arg-max over the group table (fused-DPP wave reduction, one LDS hand-off across the
//              waves, ONE s_barrier per round) picks the next sample with the reference's tie rule.
//
// One workgroup of NW waves (one per SIMD: the per-round work is a dependent chain, extra waves only
// add issue pressure -- a 16-wave version of this kernel was issue-bound at 4400 cycles per round)
// per batch element.  Group g belongs to wave g % NW, so spatially adjacent groups are handled by
// different waves; every table entry is written and read by the same wave, hence no second barrier.
#include "tpu3_dev.h"

#include <hip/hip_fp16.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include <rocprim/rocprim.hpp>

namespace {

constexpr int FB_GS = 16;        // buckets per group = lanes per DPP row

// Arguments of batch element 0; fb_elem() derives element i.  User arrays are dense (b, n, ...)
// slabs, the per-element workspace arrays repeat every `per_elem` bytes, the sort arrays every
// `sort_stride` words.
struct FbArgs {
    int n, m, nb, nbpad, npad, ng;      // n, m: slab strides = upper bounds of the live sizes
    int bsz, lb;                        // points per bucket; log2 of the tie-rule block size
    const int32_t *n_arr, *m_arr;       // live sizes per element, or null
    const float *xyz;     // (n,3) original order
    float *temp;          // (n)
    int32_t *idx;         // (m)
    float4 *sp;           // (npad) Morton order: x, y, z, running distance
    uint32_t *skey;       // (npad) tie key of the original index (0xFFFFFFFF = padding)
    uint32_t *ib;         // (8, nbpad) initial bucket table: max, key, x, y, z, box0, box1, box2
    float *bbox;          // (8)
    size_t per_elem;
    size_t sort_stride;
    unsigned long long *prof;   // PROF builds only
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
    if (a0.n_arr) {
        a.n = min(max(a0.n_arr[i], 0), a0.n);
        a.nb = (a.n + a0.bsz - 1) / a0.bsz;
        a.lb = tpu3_fps_log2_bs(a.n);
    }
    if (a0.m_arr)
        a.m = min(max(a0.m_arr[i], 0), a0.m);
    return a;
}

__device__ __forceinline__ uint32_t spread10(uint32_t v)
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// bounding box of one cloud -> bbox[6] = lo.xyz, hi.xyz
__global__ __launch_bounds__(1024) void fb_bbox_kernel(FbArgs a0)
{
    __shared__ float red[6][16];
    const FbArgs a = fb_elem(a0, blockIdx.x);
    const int n = a.n;
    const float *__restrict__ xyz = a.xyz;
    float *__restrict__ bbox = a.bbox;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[(size_t)i * 3 + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    for (int a = 0; a < 3; ++a) {
        const float l = -tpu3_wave_max_f32(-lo[a]), h = tpu3_wave_max_f32(hi[a]);
        if ((threadIdx.x & 63) == 0) {
            red[a][threadIdx.x >> 6] = l;
            red[3 + a][threadIdx.x >> 6] = h;
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[threadIdx.x][0];
        for (int w = 1; w < 16; ++w)
            v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
        bbox[threadIdx.x] = v;
    }
}

// keys64 != null: ONE device-wide sort of the whole batch on (element << 32 | key) -- every slot of an
// element's stride gets a key, the padding behind n sorts last like the dead slots of a ragged element
__global__ __launch_bounds__(256) void fb_morton_kernel(FbArgs a0, uint32_t *__restrict__ keys0,
                                                        uint32_t *__restrict__ vals0,
                                                        unsigned long long *__restrict__ keys64)
{
    const FbArgs a = fb_elem(a0, blockIdx.y);
    const int n = a.n;
    const float *__restrict__ xyz = a.xyz;
    const float *__restrict__ bbox = a.bbox;
    const size_t base = (size_t)blockIdx.y * a0.sort_stride;
    uint32_t *__restrict__ vals = vals0 + base;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    auto put = [&](uint32_t key) {
        if (keys64)
            keys64[base + i] = ((unsigned long long)blockIdx.y << 32) | key;
        else
            keys0[base + i] = key;
        vals[i] = (uint32_t)i;
    };
    if (i >= (keys64 ? (int)a0.sort_stride : a0.n))
        return;
    if (i >= n) {               // slot beyond a ragged element's live size: sorts behind every point
        put(0x40000000u);
        return;
    }
    uint32_t code = 0;
    for (int a = 0; a < 3; ++a) {
        const float lo = bbox[a], ext = bbox[3 + a] - lo;
        float q = ext > 0.f ? (xyz[(size_t)i * 3 + a] - lo) / ext * 1023.0f : 0.f;
        q = fminf(fmaxf(q, 0.f), 1023.f);
        code |= spread10((uint32_t)q) << a;
    }
    put(code);
}

// Morton-ordered float4 (x,y,z,temp) + tie keys; slots past n repeat the last live point with temp = -1
__global__ __launch_bounds__(256) void fb_permute_kernel(FbArgs a0, const uint32_t *__restrict__ order0)
{
    const FbArgs a = fb_elem(a0, blockIdx.y);
    const uint32_t *__restrict__ order = order0 + (size_t)blockIdx.y * a0.sort_stride;
    const int lb = a.lb;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.npad || a.n <= 0)
        return;
    const bool live = i < a.n;
    const uint32_t o = order[live ? i : a.n - 1];
    a.sp[i] = make_float4(a.xyz[(size_t)o * 3 + 0], a.xyz[(size_t)o * 3 + 1], a.xyz[(size_t)o * 3 + 2],
                          live ? a.temp[o] : -1.0f);
    a.skey[i] = live ? tpu3_fps_tiekey((int)o, lb) : 0xFFFFFFFFu;
}

__global__ __launch_bounds__(256) void fb_writeback_kernel(FbArgs a0)
{
    const FbArgs a = fb_elem(a0, blockIdx.y);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n && a.m > 0)
        a.temp[tpu3_fps_tiekey_to_index(a.skey[i], a.lb)] = a.sp[i].w;
}

// lane-local best of one bucket (64*PPL points), optionally after folding sample q into it
struct FbCand {
    float t, x, y, z;
    uint32_t key;
};

// The re-scan of a bucket is split into load / apply / store so that a caller can put the loads of
// two buckets in flight together and issue the stores LAST: on gfx950 loads and stores share vmcnt
// but complete out of order with each other, so a value loaded before a store can only be waited
// for with vmcnt(0) once the store is in the queue -- a store between a load and its first use
// costs a full store round trip (measured: 2000 instead of 600 cycles per bucket pair).
template <int PPL>
struct FbBucket {
    float4 v[PPL];
    uint32_t key[PPL];
    float nt[PPL];          // updated running distances
};

template <int PPL>
__device__ __forceinline__ void fb_load(FbBucket<PPL> &b, const float4 *__restrict__ sp,
                                        const uint32_t *__restrict__ skey, int beta, int lane)
{
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        const int i = beta * (64 * PPL) + p * 64 + lane;
        b.v[p] = sp[i];
        b.key[p] = skey[i];
    }
}

template <int PPL>
__device__ __forceinline__ FbCand fb_apply(FbBucket<PPL> &b, float qx, float qy, float qz, bool update)
{
    FbCand c{-2.0f, 0.f, 0.f, 0.f, 0xFFFFFFFFu};
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        float t = b.v[p].w;
        if (update)
            t = fminf(tpu3_sqdist3(b.v[p].x - qx, b.v[p].y - qy, b.v[p].z - qz), t);
        b.nt[p] = t;
        if (t > c.t || (t == c.t && b.key[p] < c.key)) {
            c.t = t; c.key = b.key[p]; c.x = b.v[p].x; c.y = b.v[p].y; c.z = b.v[p].z;
        }
    }
    return c;
}

template <int PPL>
__device__ __forceinline__ void fb_store(const FbBucket<PPL> &b, float4 *__restrict__ sp, int beta, int lane)
{
#pragma unroll
    for (int p = 0; p < PPL; ++p)
        if (b.nt[p] != b.v[p].w)
            ((float *)(sp + beta * (64 * PPL) + p * 64 + lane))[3] = b.nt[p];
}

// initial bucket table (one wave per bucket, whole GPU): max / key / xyz and the fp16 box, rounded
// outward so that it still contains every point
template <int PPL>
__global__ __launch_bounds__(256) void fb_bucket_init_kernel(FbArgs a0)
{
    const FbArgs a = fb_elem(a0, blockIdx.y);
    const int lane = threadIdx.x & 63;
    const int beta = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (beta >= a.nbpad)
        return;
    uint32_t *ib = a.ib;
    const int S = a.nbpad;
    if (beta >= a.nb) {     // padding bucket: never wins, infinitely far away
        if (lane == 0) {
            const uint32_t pinf = 0x7C00u | (0x7C00u << 16);
            ib[0 * S + beta] = 0x80000000u; ib[1 * S + beta] = 0xFFFFFFFFu;
            ib[2 * S + beta] = 0; ib[3 * S + beta] = 0; ib[4 * S + beta] = 0;
            ib[5 * S + beta] = pinf; ib[6 * S + beta] = pinf; ib[7 * S + beta] = pinf;
        }
        return;
    }
    FbBucket<PPL> bk;
    fb_load<PPL>(bk, a.sp, a.skey, beta, lane);
    const FbCand c = fb_apply<PPL>(bk, 0.f, 0.f, 0.f, false);
    int wl;
    const int wmax = tpu3_wave_argmax(__float_as_int(c.t), c.key, wl);
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        const float4 v = bk.v[p];
        lo[0] = fminf(lo[0], v.x); hi[0] = fmaxf(hi[0], v.x);
        lo[1] = fminf(lo[1], v.y); hi[1] = fmaxf(hi[1], v.y);
        lo[2] = fminf(lo[2], v.z); hi[2] = fmaxf(hi[2], v.z);
    }
    uint32_t h[6];
    for (int c3 = 0; c3 < 3; ++c3) {
        h[c3] = __half_as_ushort(__float2half_rd(-tpu3_wave_max_f32(-lo[c3])));
        h[3 + c3] = __half_as_ushort(__float2half_ru(tpu3_wave_max_f32(hi[c3])));
    }
    if (lane == wl) {
        ib[0 * S + beta] = (uint32_t)wmax; ib[1 * S + beta] = c.key;
        ib[2 * S + beta] = __float_as_uint(c.x); ib[3 * S + beta] = __float_as_uint(c.y);
        ib[4 * S + beta] = __float_as_uint(c.z);
        ib[5 * S + beta] = h[0] | (h[1] << 16); ib[6 * S + beta] = h[2] | (h[3] << 16);
        ib[7 * S + beta] = h[4] | (h[5] << 16);
    }
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

struct FbSlots {        // cross-wave hand-off, up to 8 waves
    int d[2][8];
    uint32_t key[2][8];
    float x[2][8], y[2][8], z[2][8];
};

// LDS bytes of the main kernel: 8 words per bucket, 11 per group-table entry, the slots
constexpr size_t fb_lds_bytes(int nbpad, int nw, int ngpt)
{
    return (size_t)nbpad * 32 + (size_t)ngpt * nw * 64 * 11 * 4 + sizeof(FbSlots) + 64;
}

template <int NW, int NGPT, int PPL, bool PROF = false>
__global__ __launch_bounds__(NW * 64) void fb_main_kernel(FbArgs a0)
{
    constexpr int W = NW * 64;
    constexpr int GT = NGPT * W;                    // group-table entries (owner order)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FbArgs a = fb_elem(a0, blockIdx.x);
    const int nbpad = a.nbpad, ng = a.ng, lb = a.lb;
    if (a.n <= 0 || a.m <= 0)
        return;
    // Bucket i (64 * PPL consecutive Morton points) belongs to wave i % NW; a wave groups ITS buckets
    // 16 to a group in order.  Spatially adjacent buckets -- the handful a sample touches -- are thus
    // re-scanned by different waves in parallel (with groups of 16 consecutive buckets one wave did
    // most of a round's re-scans while the others waited at the barrier).  The LDS bucket table is
    // stored wave-major (slot = (i % NW) * Q + i / NW), so a group's children are 16 consecutive words.
    const int Q = nbpad / NW;
    // bucket table, indexed by bucket id (a DPP row reads 16 consecutive children)
    int *t_max = (int *)smem;
    uint32_t *t_key = (uint32_t *)(t_max + nbpad);
    float *t_x = (float *)(t_key + nbpad);
    float *t_y = t_x + nbpad;
    float *t_z = t_y + nbpad;
    uint32_t *t_b0 = (uint32_t *)(t_z + nbpad);     // fp16 boxes: lo.x|lo.y, lo.z|hi.x, hi.y|hi.z
    uint32_t *t_b1 = t_b0 + nbpad;
    uint32_t *t_b2 = t_b1 + nbpad;
    // group table in OWNER order (entry of the group owned by lane `l`, slot `j` at j*W + tid):
    // the per-round read of a wave is 64 consecutive words, no bank conflicts
    int *g_max = (int *)(t_b2 + nbpad);
    uint32_t *g_key = (uint32_t *)(g_max + GT);
    float *g_x = (float *)(g_key + GT);
    float *g_y = g_x + GT;
    float *g_z = g_y + GT;
    float *g_box = g_z + GT;                        // 6 x GT, setup only
    FbSlots &sl = *(FbSlots *)(g_box + 6 * GT);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = lane >> 4, col = lane & 15;
    float4 *__restrict__ sp = a.sp;
    const uint32_t *__restrict__ skey = a.skey;

    // group g  <->  wave g % NW, owner slot g / NW (lane slot % 64, register slot / 64)
    auto group_entry = [&](int slot) { return (slot >> 6) * W + wave * 64 + (slot & 63); };

    // Rebuild the entries of up to four groups at once: DPP row r handles the group in owner slot
    // `slot` (per lane; < 0 = row idle).  Row arg-max over the 16 children with the FPS tie rule.
    auto refresh_groups = [&](int slot, bool with_box) {
        const bool valid = slot >= 0 && slot * NW + wave < ng;
        const int beta = valid ? wave * Q + slot * FB_GS + col : 0;         // table index (see below)
        const int bits = valid ? t_max[beta] : (int)0x80000000;
        const uint32_t key = valid ? t_key[beta] : 0xFFFFFFFFu;
        const int rmax = tpu3_row_max_i32_fast(bits);
        unsigned long long tie = __ballot(valid && bits == rmax);
        const unsigned long long rows = __ballot(valid && col == 0);
        if (__builtin_popcountll(tie) != __builtin_popcountll(rows)) {     // duplicated points
            const uint32_t k = tpu3_row_min_u32(valid && bits == rmax ? key : 0xFFFFFFFFu);
            tie = __ballot(valid && bits == rmax && key == k);
        }
        const unsigned long long below = ((1ull << col) - 1ull) << (row * 16);
        if (valid && ((tie >> lane) & 1ull) && (tie & below) == 0) {
            const int e = group_entry(slot);
            g_max[e] = rmax; g_key[e] = key;
            g_x[e] = t_x[beta]; g_y[e] = t_y[beta]; g_z[e] = t_z[beta];
        }
        if (with_box) {     // setup: group AABB = union of the children's (outward-rounded) boxes
            const uint32_t w0 = valid ? t_b0[beta] : 0, w1 = valid ? t_b1[beta] : 0, w2 = valid ? t_b2[beta] : 0;
            float v[6] = {-fb_half_lo(w0), -fb_half_hi(w0), -fb_half_lo(w1), fb_half_hi(w1), fb_half_lo(w2),
                          fb_half_hi(w2)};
            for (int c3 = 0; c3 < 6; ++c3) {
                if (!valid)
                    v[c3] = -__builtin_inff();
                const float m = tpu3_unmono(tpu3_row_max_u32(tpu3_mono(v[c3])));
                if (valid && col == 0)
                    g_box[c3 * GT + group_entry(slot)] = c3 < 3 ? -m : m;
            }
        }
    };

    // ---- setup: bucket table from the init kernel's arrays, then every group's entry + AABB --------
    for (int i = tid; i < nbpad; i += W) {
        const int ti = (i % NW) * Q + i / NW;           // bucket i -> table slot
        t_max[ti] = (int)a.ib[0 * nbpad + i];
        t_key[ti] = a.ib[1 * nbpad + i];
        t_x[ti] = __uint_as_float(a.ib[2 * nbpad + i]);
        t_y[ti] = __uint_as_float(a.ib[3 * nbpad + i]);
        t_z[ti] = __uint_as_float(a.ib[4 * nbpad + i]);
        t_b0[ti] = a.ib[5 * nbpad + i];
        t_b1[ti] = a.ib[6 * nbpad + i];
        t_b2[ti] = a.ib[7 * nbpad + i];
    }
    for (int i = tid; i < GT; i += W) {
        g_max[i] = (int)0x80000000; g_key[i] = 0xFFFFFFFFu;
        g_x[i] = g_y[i] = g_z[i] = 0.f;
        for (int c3 = 0; c3 < 6; ++c3)
            g_box[c3 * GT + i] = __builtin_inff();          // lo = hi = +inf: infinitely far away
    }
    __syncthreads();
    for (int s0 = 0; s0 < NGPT * 64; s0 += 4)
        if ((s0 * NW + wave) < ng)
            refresh_groups(s0 + row, true);
    __syncthreads();
    float gbox[NGPT][6];
    int gmax[NGPT];
#pragma unroll
    for (int j = 0; j < NGPT; ++j) {
        for (int c3 = 0; c3 < 6; ++c3)
            gbox[j][c3] = g_box[c3 * GT + j * W + tid];
        gmax[j] = g_max[j * W + tid];
    }

    if (tid == 0)
        a.idx[0] = 0;
    float qx = a.xyz[0], qy = a.xyz[1], qz = a.xyz[2];

    unsigned long long pc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 1; r < a.m; ++r) {
        unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0;
        if (PROF) tk0 = __builtin_amdgcn_s_memtime();
        // ---- 1. group prune --------------------------------------------------------------------------
        unsigned long long gm[NGPT];
#pragma unroll
        for (int j = 0; j < NGPT; ++j)
            gm[j] = __ballot(fb_dbox(qx, qy, qz, gbox[j][0], gbox[j][1], gbox[j][2], gbox[j][3], gbox[j][4],
                                     gbox[j][5]) < __int_as_float(gmax[j]));
        unsigned long long ta = 0, tb = 0, tc = 0, td = 0;
        if (PROF) { ta = __builtin_amdgcn_s_memtime(); pc[8] += ta - tk0; }
#pragma unroll
        for (int j = 0; j < NGPT; ++j) {
            unsigned long long mask = gm[j];
            while (mask) {
                // up to four touched groups, one per DPP row
                int slot = -1;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    if (mask) {
                        const int l = __builtin_ctzll(mask);
                        mask &= mask - 1;
                        if (row == rr)
                            slot = j * 64 + l;
                    }
                // ---- 2. children test ------------------------------------------------------------------
                const bool valid = slot >= 0;
                const int beta = valid ? wave * Q + slot * FB_GS + col : 0;
                const uint32_t w0 = t_b0[beta], w1 = t_b1[beta], w2 = t_b2[beta];
                const float db = fb_dbox(qx, qy, qz, fb_half_lo(w0), fb_half_hi(w0), fb_half_lo(w1), fb_half_hi(w1),
                                         fb_half_lo(w2), fb_half_hi(w2));
                unsigned long long bt = __ballot(valid && db < __int_as_float(t_max[beta]));
                if (PROF) { tb = __builtin_amdgcn_s_memtime(); pc[9] += tb - ta; }
                // ---- 3. re-scan the touched buckets two at a time (an odd one out twice: idempotent) ---
                while (bt) {
                    const int p0 = __builtin_ctzll(bt);
                    bt &= bt - 1;
                    int p1 = p0;
                    if (bt) {
                        p1 = __builtin_ctzll(bt);
                        bt &= bt - 1;
                    }
                    const int b0 = __builtin_amdgcn_readlane(beta, p0), b1 = __builtin_amdgcn_readlane(beta, p1);
                    const int d0 = (b0 - wave * Q) * NW + wave, d1 = (b1 - wave * Q) * NW + wave;    // bucket ids
                    FbBucket<PPL> k0, k1;
                    fb_load<PPL>(k0, sp, skey, d0, lane);
                    fb_load<PPL>(k1, sp, skey, d1, lane);
                    const FbCand c0 = fb_apply<PPL>(k0, qx, qy, qz, true);
                    const FbCand c1 = fb_apply<PPL>(k1, qx, qy, qz, true);
                    int m0 = __float_as_int(c0.t), m1 = __float_as_int(c1.t);
                    if (PROF) { asm volatile("" :: "v"(m0), "v"(m1)); tc = __builtin_amdgcn_s_memtime(); pc[10] += tc - tb; }
                    tpu3_wave_max_i32_fast_x2(m0, m1);
                    unsigned long long t0 = __ballot(__float_as_int(c0.t) == m0);
                    unsigned long long t1 = __ballot(__float_as_int(c1.t) == m1);
                    if (__builtin_popcountll(t0) != 1) {      // duplicated points: smallest tie key
                        const uint32_t k = tpu3_wave_min_u32(__float_as_int(c0.t) == m0 ? c0.key : 0xFFFFFFFFu);
                        t0 = __ballot(__float_as_int(c0.t) == m0 && c0.key == k);
                    }
                    if (__builtin_popcountll(t1) != 1) {
                        const uint32_t k = tpu3_wave_min_u32(__float_as_int(c1.t) == m1 ? c1.key : 0xFFFFFFFFu);
                        t1 = __ballot(__float_as_int(c1.t) == m1 && c1.key == k);
                    }
                    if (lane == (int)__builtin_ctzll(t0)) {
                        t_max[b0] = m0; t_key[b0] = c0.key; t_x[b0] = c0.x; t_y[b0] = c0.y; t_z[b0] = c0.z;
                    }
                    if (lane == (int)__builtin_ctzll(t1)) {
                        t_max[b1] = m1; t_key[b1] = c1.key; t_x[b1] = c1.x; t_y[b1] = c1.y; t_z[b1] = c1.z;
                    }
                    fb_store<PPL>(k0, sp, d0, lane);        // stores last, off the dependent chain
                    if (b1 != b0)
                        fb_store<PPL>(k1, sp, d1, lane);
                    if (PROF) pc[5] += 1 + (b1 != b0);
                    if (PROF) { tb = __builtin_amdgcn_s_memtime(); pc[11] += tb - tc; }
                }
                // ---- 4. rebuild the touched groups' entries ------------------------------------------------
                if (PROF) td = __builtin_amdgcn_s_memtime();
                refresh_groups(slot, false);
                if (PROF) { ta = __builtin_amdgcn_s_memtime(); pc[12] += ta - td; }
                if (PROF) pc[6] += 1;
            }
        }
        if (PROF) tk1 = __builtin_amdgcn_s_memtime();
        // ---- 5. arg-max over the group table ----------------------------------------------------------
        int best = (int)0x80000000, bj = 0;
        uint32_t bkey = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < NGPT; ++j) {
            const int v = g_max[j * W + tid];
            const uint32_t k = g_key[j * W + tid];
            gmax[j] = v;
            if (v > best || (v == best && k < bkey)) {
                best = v; bkey = k; bj = j;
            }
        }
        const int par = r & 1;
        int wl;
        const int wmax = tpu3_wave_argmax(best, bkey, wl);
        if (lane == wl) {
            const int e = bj * W + tid;
            sl.d[par][wave] = wmax;
            sl.key[par][wave] = bkey;
            sl.x[par][wave] = g_x[e];
            sl.y[par][wave] = g_y[e];
            sl.z[par][wave] = g_z[e];
        }
        if (PROF) tk2 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (PROF) tk3 = __builtin_amdgcn_s_memtime();
        const int sd = lane < NW ? sl.d[par][lane] : (int)0x80000000;
        const uint32_t sk = lane < NW ? sl.key[par][lane] : 0xFFFFFFFFu;
        // lane l < NW holds wave l's whole slot: the winner's coordinates come by readlane, not by a
        // second LDS round trip
        const float sx = sl.x[par][lane & (NW - 1)], sy = sl.y[par][lane & (NW - 1)], sz = sl.z[par][lane & (NW - 1)];
        const int gbest = __builtin_amdgcn_readlane(tpu3_row_max_i32_fast(sd), 0);
        unsigned long long who = __ballot(lane < NW && sd == gbest);
        if (__builtin_popcountll(who) != 1) {
            const uint32_t rk = tpu3_row_min_u32(lane < NW && sd == gbest ? sk : 0xFFFFFFFFu);
            const uint32_t win = (uint32_t)__builtin_amdgcn_readlane((int)rk, 0);
            who = __ballot(lane < NW && sd == gbest && sk == win);
        }
        const int ww = __builtin_ctzll(who | (1ull << 63)) & 7;
        qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sx), ww));
        qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sy), ww));
        qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sz), ww));
        if (tid == 0)
            a.idx[r] = tpu3_fps_tiekey_to_index((uint32_t)__builtin_amdgcn_readlane((int)sk, ww), lb);
        if (PROF) {
            const unsigned long long tk4 = __builtin_amdgcn_s_memtime();
            pc[0] += tk1 - tk0; pc[1] += tk2 - tk1; pc[2] += tk3 - tk2; pc[3] += tk4 - tk3;
        }
    }
    if (PROF && lane == 0 && a.prof)
        for (int i = 0; i < 16; ++i)
            a.prof[wave * 16 + i] = pc[i];
}

// ---------------------------------------------------------------------------------------------
// Point sets that fit the register file (n <= 25 600): the same exact pruning with the points held
// in VGPRs.  One 16-wave workgroup per set; a ROW is 64 consecutive points of the Morton order (a
// compact region, = one bucket of the init kernel) and row r belongs to wave r % 16, slot r / 16
// (lane l holds point 64 r + l: x, y, z, running distance), so the handful of neighbouring rows a
// sample touches are re-scanned by different waves in parallel.  Per round lane j of a wave tests row j's
// AABB (kept in that lane's registers together with the row's current maximum); only touched rows
// are re-scanned -- the row index is wave-uniform, so the register array is addressed by scalar
// branches -- and publish (max, tie key, xyz of that point) to a row table in LDS.  The register-
// resident kernel of fps.hip spends 25 points x 10 VALU ops per lane and round on the same sets
// (2.0 us per round, all 16 waves busy); here a round touches ~2 rows of the whole set.
// ---------------------------------------------------------------------------------------------
template <int I, int N, typename F>
__device__ __forceinline__ void rb_static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        rb_static_for<I + 1, N>(f);
    }
}

struct RbSlots {
    int d[2][16];
    uint32_t key[2][16];
    float x[2][16], y[2][16], z[2][16];
};

constexpr size_t rb_lds_bytes(int r)
{
    return (size_t)16 * r * 64 * 4 + (size_t)16 * r * 5 * 4 + sizeof(RbSlots) + 64;
}

template <int R>
__global__ __launch_bounds__(1024) void rb_main_kernel(FbArgs a0)
{
    constexpr int NW = 16, ROWS = NW * R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: tie keys [wave][slot][lane] and row records [wave][slot]{max, key, x, y, z}: everything a
    // wave touches in the round loop is its own base address plus a compile-time offset (no
    // per-row address registers -- with 100 VGPRs of points there is no room for them)
    uint32_t *skl = (uint32_t *)smem;
    uint32_t *tbl = skl + ROWS * 64;
    RbSlots &sl = *(RbSlots *)(tbl + ROWS * 5);
    const FbArgs a = fb_elem(a0, blockIdx.x);
    if (a.n <= 0 || a.m <= 0)
        return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lb = a.lb;
    uint32_t *kw = skl + wave * R * 64 + lane;      // this lane's keys: kw[64 * j]
    uint32_t *tw = tbl + wave * R * 5;              // this wave's records: tw[5 * j + field]

    // row r = 16 * slot + wave (neighbouring rows go to different waves); lane l holds point 64 r + l
    float px[R], py[R], pz[R], pt[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int slot = (j * NW + wave) * 64 + lane;
        float4 v = make_float4(0.f, 0.f, 0.f, -1.0f);
        uint32_t key = 0xFFFFFFFFu;
        if (slot < a0.npad) {
            v = a.sp[slot];
            key = a.skey[slot];
        }
        px[j] = v.x; py[j] = v.y; pz[j] = v.z; pt[j] = v.w;
        kw[64 * j] = key;
    }
    // Row records and boxes come from the bucket-init kernel (a row is a 64-point bucket); lane j < R
    // keeps row j's AABB (fp16, rounded outward, packed) and current maximum in registers
    for (int i = tid; i < ROWS; i += 1024) {
        const int w = i / R, j = i - w * R, row = j * NW + w;
        const bool in = row < a0.nbpad;
        tbl[i * 5 + 0] = in ? a.ib[0 * a0.nbpad + row] : 0x80000000u;
        tbl[i * 5 + 1] = in ? a.ib[1 * a0.nbpad + row] : 0xFFFFFFFFu;
        tbl[i * 5 + 2] = in ? a.ib[2 * a0.nbpad + row] : 0u;
        tbl[i * 5 + 3] = in ? a.ib[3 * a0.nbpad + row] : 0u;
        tbl[i * 5 + 4] = in ? a.ib[4 * a0.nbpad + row] : 0u;
    }
    uint32_t bw0, bw1, bw2;
    int rowmax = (int)0x80000000;
    {
        const int row = min(lane, R - 1) * NW + wave;
        const bool in = lane < R && row < a0.nbpad;
        const uint32_t pinf = 0x7C00u | (0x7C00u << 16);
        bw0 = in ? a.ib[5 * a0.nbpad + row] : pinf;
        bw1 = in ? a.ib[6 * a0.nbpad + row] : pinf;
        bw2 = in ? a.ib[7 * a0.nbpad + row] : pinf;
        rowmax = in ? (int)a.ib[0 * a0.nbpad + row] : (int)0x80000000;
    }
    __syncthreads();

    float qx = a.xyz[0], qy = a.xyz[1], qz = a.xyz[2];
    // re-scan of slot j (compile-time j): fold the sample in, row arg-max with the FPS tie rule,
    // publish the record; returns the row's maximum (wave-uniform)
    auto rescan = [&](auto jc) -> int {
        constexpr int j = decltype(jc)::value;
        const float t = fminf(tpu3_sqdist3(px[j] - qx, py[j] - qy, pz[j] - qz), pt[j]);
        pt[j] = t;
        const int bits = __float_as_int(t);
        const int wmax = tpu3_wave_max_i32_fast(bits);
        unsigned long long tie = __ballot(bits == wmax);
        if (__builtin_popcountll(tie) != 1) {                    // duplicated points: smallest tie key
            const uint32_t k = kw[64 * j];
            const uint32_t kmin = tpu3_wave_min_u32(bits == wmax ? k : 0xFFFFFFFFu);
            tie = __ballot(bits == wmax && k == kmin);
        }
        if (lane == (int)__builtin_ctzll(tie)) {
            tw[5 * j + 0] = (uint32_t)wmax; tw[5 * j + 1] = kw[64 * j];
            tw[5 * j + 2] = __float_as_uint(px[j]); tw[5 * j + 3] = __float_as_uint(py[j]);
            tw[5 * j + 4] = __float_as_uint(pz[j]);
        }
        return wmax;
    };

    if (tid == 0)
        a.idx[0] = 0;
    for (int r = 1; r < a.m; ++r) {
        // ---- prune: which of this wave's rows can the sample change? ------------------------------
        const float db = fb_dbox(qx, qy, qz, fb_half_lo(bw0), fb_half_hi(bw0), fb_half_lo(bw1), fb_half_hi(bw1),
                                 fb_half_lo(bw2), fb_half_hi(bw2));
        const unsigned long long mask = __ballot(lane < R && db < __int_as_float(rowmax));
        if (mask)
            rb_static_for<0, R>([&](auto jc) {
                if ((mask >> decltype(jc)::value) & 1ull) {
                    const int wm = rescan(jc);
                    rowmax = lane == decltype(jc)::value ? wm : rowmax;
                }
            });
        // ---- this wave's best row: every lane fetches ITS row's record while the maxima are reduced, so
        // the winner can publish without a second LDS round trip ----------------------------------------------
        const int lj = min(lane, R - 1);
        const uint32_t rk = tw[5 * lj + 1];
        const float rx = __uint_as_float(tw[5 * lj + 2]), ry = __uint_as_float(tw[5 * lj + 3]);
        const float rz = __uint_as_float(tw[5 * lj + 4]);
        const int mine = lane < R ? rowmax : (int)0x80000000;
        const int wv = tpu3_wave_max_i32_fast(mine);
        unsigned long long tie = __ballot(mine == wv);
        if (__builtin_popcountll(tie) != 1) {
            const uint32_t kmin = tpu3_wave_min_u32(mine == wv ? rk : 0xFFFFFFFFu);
            tie = __ballot(mine == wv && rk == kmin);
        }
        const int par = r & 1;
        if (lane == (int)__builtin_ctzll(tie | (1ull << 63))) {
            sl.d[par][wave] = wv;
            sl.key[par][wave] = rk;
            sl.x[par][wave] = rx; sl.y[par][wave] = ry; sl.z[par][wave] = rz;
        }
        __syncthreads();
        // ---- arg-max over the 16 waves (one DPP row); lane l < 16 holds wave l's whole slot ----------------
        const int sd = lane < NW ? sl.d[par][lane] : (int)0x80000000;
        const uint32_t sk = lane < NW ? sl.key[par][lane] : 0xFFFFFFFFu;
        const float sx = sl.x[par][lane & (NW - 1)], sy = sl.y[par][lane & (NW - 1)], sz = sl.z[par][lane & (NW - 1)];
        const int gbest = __builtin_amdgcn_readlane(tpu3_row_max_i32_fast(sd), 0);
        unsigned long long who = __ballot(lane < NW && sd == gbest);
        if (__builtin_popcountll(who) != 1) {
            const uint32_t wk = tpu3_row_min_u32(lane < NW && sd == gbest ? sk : 0xFFFFFFFFu);
            const uint32_t win = (uint32_t)__builtin_amdgcn_readlane((int)wk, 0);
            who = __ballot(lane < NW && sd == gbest && sk == win);
        }
        const int ww = __builtin_ctzll(who | (1ull << 63)) & (NW - 1);
        // (uniform values: keep them in SGPRs)
        qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sx), ww));
        qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sy), ww));
        qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sz), ww));
        const uint32_t wkey = (uint32_t)__builtin_amdgcn_readlane((int)sk, ww);
        if (tid == 0)
            a.idx[r] = tpu3_fps_tiekey_to_index(wkey, lb);
    }
    // final running distances, back in the caller's order
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const uint32_t key = kw[64 * j];
        if (key != 0xFFFFFFFFu)
            a.temp[tpu3_fps_tiekey_to_index(key, lb)] = pt[j];
    }
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

constexpr int RB_MAX_N = 16 * 64 * 25;     // 25 rows per wave

struct FbPlan {
    int ppl, nw, ngpt, nb, nbpad, npad, ng;
    int rb_rows;          // > 0: register-resident kernel with this many rows per wave
    bool segmented;       // one segmented sort for the batch instead of a device sort per element
    bool global64;        // large sets, several elements: one device sort on (element << 32 | key)
    int ebits;            // bits of the element index in that key
    size_t ks, ps, bs;    // byte sizes: key array, per-point float array, bucket word array
    size_t per_elem;      // bytes of one batch element's arrays (sp, skey, ib, bbox)
    size_t sort_bytes;    // 4 key/value arrays x b
    size_t sort_temp;     // rocPRIM temporary storage
    size_t total;
};

constexpr int FB_NW = 4;            // waves per workgroup (one per SIMD)
constexpr int FB_NB_MAX = 4096;     // buckets: 32 B of LDS each
constexpr int FB_SORT_BITS = 31;    // 30 Morton bits + the dead-slot bit of ragged elements

// segment i of the batch-wide sort arrays: [i * stride, i * stride + n)
struct FbSegOffset {
    unsigned int stride, add;
    __host__ __device__ unsigned int operator()(unsigned int i) const { return i * stride + add; }
};
using FbCount = rocprim::counting_iterator<unsigned int>;
using FbOffsetIt = rocprim::transform_iterator<FbCount, FbSegOffset>;

bool fb_plan(int b, int n, FbPlan &p)
{
    p.rb_rows = 0;
    if (n <= RB_MAX_N) {
        const int rows = ((n + 63) / 64 + 15) / 16;
        for (int r : {4, 7, 10, 13, 16, 20, 25})
            if (r >= rows) {
                p.rb_rows = r;
                break;
            }
    }
    p.ppl = 0;
    for (int ppl : {1, 2, 4, 8, 16})
        if ((long)FB_NB_MAX * 64 * ppl >= n) {
            p.ppl = ppl;
            break;
        }
    if (!p.ppl)
        return false;
    const int bsz = 64 * p.ppl;
    p.nb = (n + bsz - 1) / bsz;
    p.nbpad = (p.nb + FB_GS * FB_NW - 1) / (FB_GS * FB_NW) * (FB_GS * FB_NW);       // whole groups per wave
    p.ng = p.nbpad / FB_GS;
    p.npad = p.nb * bsz;
    p.nw = FB_NW;
    p.ngpt = (p.ng + p.nw * 64 - 1) / (p.nw * 64);          // 1 for ng <= 256
    p.ks = align256(sizeof(uint32_t) * (size_t)n);
    p.ps = align256(sizeof(float) * (size_t)p.npad);
    p.bs = align256(sizeof(uint32_t) * (size_t)p.nbpad);
    p.per_elem = 5 * p.ps + 8 * p.bs + align256(8 * sizeof(float));
    p.segmented = b >= 4 && n <= 65536 && (size_t)b * (p.ks / 4) < 0x7FFFFFFFu;
    p.global64 = !p.segmented && b >= 2;
    p.ebits = 1;
    while ((1 << p.ebits) < b)
        ++p.ebits;
    p.sort_bytes = (p.global64 ? 6 : 4) * p.ks * (size_t)b;       // 64-bit keys in and out + two value arrays
    size_t tb = 0;
    if (p.global64) {
        (void)rocprim::radix_sort_pairs(nullptr, tb, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                        (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)b * (p.ks / 4), 0,
                                        32 + p.ebits, (hipStream_t)0);
    } else if (p.segmented) {
        const unsigned int stride = (unsigned int)(p.ks / 4);
        const FbOffsetIt bi(FbCount(0), FbSegOffset{stride, 0u});
        const FbOffsetIt ei(FbCount(0), FbSegOffset{stride, (unsigned int)n});
        (void)rocprim::segmented_radix_sort_pairs(nullptr, tb, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                                  (uint32_t *)nullptr, (uint32_t *)nullptr,
                                                  (unsigned int)((size_t)b * stride), (unsigned int)b, bi, ei, 0,
                                                  FB_SORT_BITS, (hipStream_t)0);
    } else {
        (void)rocprim::radix_sort_pairs(nullptr, tb, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                        (uint32_t *)nullptr, (size_t)n, 0, FB_SORT_BITS, (hipStream_t)0);
    }
    p.sort_temp = align256(tb);
    p.total = p.sort_bytes + (size_t)b * p.per_elem + p.sort_temp;
    return true;
}

// measurement hook (bench.py): events recorded on the launch stream immediately around the next
// fb_main_kernel launch, see tpu3_debug_fps_bucket_events
hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;

template <int PPL, bool PROF>
int fb_launch_main(hipStream_t s, int b, const FbArgs &a0, const FbPlan &p)
{
    if (p.ngpt != 1)
        return TPU3_ELIMIT;
    const size_t lds = fb_lds_bytes(p.nbpad, FB_NW, 1);
    auto kern = fb_main_kernel<FB_NW, 1, PPL, PROF>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess)
        return (int)e;
    const hipEvent_t e0 = g_ev_start, e1 = g_ev_stop;
    g_ev_start = g_ev_stop = nullptr;
    if (e0) (void)hipEventRecord(e0, s);
    hipLaunchKernelGGL(kern, dim3(b), dim3(FB_NW * 64), lds, s, a0);
    if (e1) (void)hipEventRecord(e1, s);
    return tpu3_launch_status();
}

int fb_run(hipStream_t s, int b, int n, int m, const int32_t *n_arr, const int32_t *m_arr, const float *xyz,
           float *temp, int32_t *idx, void *workspace, size_t workspace_bytes, unsigned long long *prof)
{
    FbPlan p;
    if (!fb_plan(b, n, p))
        return TPU3_ELIMIT;
    if (!workspace || workspace_bytes < p.total)
        return TPU3_EINVAL;
    if (b > 65535)
        return TPU3_ELIMIT;
    char *base = (char *)workspace;
    // workspace: [k_in | k_out | v_in | v_out] (b x ks each), b element slabs, sort temp
    uint32_t *k_in = (uint32_t *)base, *k_out = (uint32_t *)(base + (size_t)b * p.ks);
    uint32_t *v_in = (uint32_t *)(base + 2 * (size_t)b * p.ks), *v_out = (uint32_t *)(base + 3 * (size_t)b * p.ks);
    unsigned long long *k64_in = nullptr, *k64_out = nullptr;
    if (p.global64) {           // [k64_in | k64_out | v_in | v_out]
        k64_in = (unsigned long long *)base;
        k64_out = (unsigned long long *)(base + 2 * (size_t)b * p.ks);
        v_in = (uint32_t *)(base + 4 * (size_t)b * p.ks);
        v_out = (uint32_t *)(base + 5 * (size_t)b * p.ks);
    }
    char *slabs = base + p.sort_bytes;
    char *sort_tmp = slabs + (size_t)b * p.per_elem;
    FbArgs a0;
    a0.n = n; a0.m = m; a0.nb = p.nb; a0.nbpad = p.nbpad; a0.npad = p.npad; a0.ng = p.ng;
    a0.bsz = 64 * p.ppl; a0.lb = tpu3_fps_log2_bs(n);
    a0.n_arr = n_arr; a0.m_arr = m_arr;
    a0.xyz = xyz; a0.temp = temp; a0.idx = idx; a0.prof = prof;
    a0.sp = (float4 *)slabs;
    a0.skey = (uint32_t *)(slabs + 4 * p.ps);
    a0.ib = (uint32_t *)(slabs + 5 * p.ps);
    a0.bbox = (float *)(slabs + 5 * p.ps + 8 * p.bs);
    a0.per_elem = p.per_elem;
    a0.sort_stride = p.ks / 4;

    hipLaunchKernelGGL(fb_bbox_kernel, dim3(b), dim3(1024), 0, s, a0);
    hipLaunchKernelGGL(fb_morton_kernel, dim3((unsigned)((a0.sort_stride + 255) / 256), b), dim3(256), 0, s, a0, k_in,
                       v_in, k64_in);
    size_t tb = p.sort_temp;
    if (p.global64) {
        const hipError_t se = rocprim::radix_sort_pairs((void *)sort_tmp, tb, k64_in, k64_out, v_in, v_out,
                                                        (size_t)b * a0.sort_stride, 0, 32 + p.ebits, s);
        if (se != hipSuccess)
            return (int)se;
    } else if (p.segmented) {
        const unsigned int stride = (unsigned int)a0.sort_stride;
        const FbOffsetIt bi(FbCount(0), FbSegOffset{stride, 0u});
        const FbOffsetIt ei(FbCount(0), FbSegOffset{stride, (unsigned int)n});
        const hipError_t se = rocprim::segmented_radix_sort_pairs((void *)sort_tmp, tb, k_in, k_out, v_in, v_out,
                                                                  (unsigned int)((size_t)b * stride), (unsigned int)b,
                                                                  bi, ei, 0, FB_SORT_BITS, s);
        if (se != hipSuccess)
            return (int)se;
    } else {
        for (int i = 0; i < b; ++i) {
            const size_t o = (size_t)i * a0.sort_stride;
            tb = p.sort_temp;
            const hipError_t se = rocprim::radix_sort_pairs((void *)sort_tmp, tb, k_in + o, k_out + o, v_in + o,
                                                            v_out + o, (size_t)n, 0, FB_SORT_BITS, s);
            if (se != hipSuccess)
                return (int)se;
        }
    }
    hipLaunchKernelGGL(fb_permute_kernel, dim3((p.npad + 255) / 256, b), dim3(256), 0, s, a0, v_out);
    if (p.rb_rows && !prof && p.ppl == 1) {
        // the set fits the register file: rows (= 64-point buckets) in VGPRs, no write-back pass
        hipLaunchKernelGGL(fb_bucket_init_kernel<1>, dim3((p.nbpad + 3) / 4, b), dim3(256), 0, s, a0);
        const size_t lds = rb_lds_bytes(p.rb_rows);
        hipError_t e = hipSuccess;
#define RB_LAUNCH(RR)                                                                                    \
    e = hipFuncSetAttribute((const void *)rb_main_kernel<RR>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                            (int)lds);                                                                   \
    if (e != hipSuccess) return (int)e;                                                                  \
    hipLaunchKernelGGL(rb_main_kernel<RR>, dim3(b), dim3(1024), lds, s, a0)
        switch (p.rb_rows) {
        case 4: RB_LAUNCH(4); break;
        case 7: RB_LAUNCH(7); break;
        case 10: RB_LAUNCH(10); break;
        case 13: RB_LAUNCH(13); break;
        case 16: RB_LAUNCH(16); break;
        case 20: RB_LAUNCH(20); break;
        default: RB_LAUNCH(25); break;
        }
#undef RB_LAUNCH
        return tpu3_launch_status();
    }
    const dim3 gi((p.nbpad + 3) / 4, b);
    switch (p.ppl) {
    case 1: hipLaunchKernelGGL(fb_bucket_init_kernel<1>, gi, dim3(256), 0, s, a0); break;
    case 2: hipLaunchKernelGGL(fb_bucket_init_kernel<2>, gi, dim3(256), 0, s, a0); break;
    case 4: hipLaunchKernelGGL(fb_bucket_init_kernel<4>, gi, dim3(256), 0, s, a0); break;
    case 8: hipLaunchKernelGGL(fb_bucket_init_kernel<8>, gi, dim3(256), 0, s, a0); break;
    default: hipLaunchKernelGGL(fb_bucket_init_kernel<16>, gi, dim3(256), 0, s, a0); break;
    }
    int r;
    if (prof) {
        if (p.ppl != 1) return TPU3_EINVAL;
        r = fb_launch_main<1, true>(s, b, a0, p);
    } else {
        switch (p.ppl) {
        case 1: r = fb_launch_main<1, false>(s, b, a0, p); break;
        case 2: r = fb_launch_main<2, false>(s, b, a0, p); break;
        case 4: r = fb_launch_main<4, false>(s, b, a0, p); break;
        case 8: r = fb_launch_main<8, false>(s, b, a0, p); break;
        default: r = fb_launch_main<16, false>(s, b, a0, p); break;
        }
    }
    if (r)
        return r;
    hipLaunchKernelGGL(fb_writeback_kernel, dim3((n + 255) / 256, b), dim3(256), 0, s, a0);
    return tpu3_launch_status();
}

} // namespace

size_t tpu3_fps_bucket_workspace_bytes(int b, int n)
{
    FbPlan p;
    return fb_plan(b, n, p) ? p.total : 0;
}

// n_arr / m_arr: live sizes per element (device, may be null).  Returns TPU3_ELIMIT when n is beyond
// the bucket plan.
int tpu3_fps_bucket_launch(hipStream_t s, int b, int n, int m, const int32_t *n_arr, const int32_t *m_arr,
                           const float *xyz, float *temp, int32_t *idx, void *workspace, size_t workspace_bytes)
{
    return fb_run(s, b, n, m, n_arr, m_arr, xyz, temp, idx, workspace, workspace_bytes, nullptr);
}

// Measurement hook (not part of include/tpu3.h): the NEXT bucketed-FPS call records `start` / `stop`
// (hipEvent_t, created by the caller) on its stream right before / after fb_main_kernel, so that a
// caller can time exactly that kernel on whatever stream it runs.  One-shot; host-side state only.
extern "C" int tpu3_debug_fps_bucket_events(void *start, void *stop)
{
    g_ev_start = (hipEvent_t)start;
    g_ev_stop = (hipEvent_t)stop;
    return TPU3_OK;
}

// Development probe (not part of include/tpu3.h): the same kernel with per-phase cycle counters;
// prof = NW x 8 u64: [prune+rescan+refresh, argmax, barrier wait, broadcast, -, buckets, group batches, -].
extern "C" int tpu3_debug_fps_bucket_profile(void *stream, int n, int m, const float *xyz, float *temp,
                                             int32_t *idx, void *workspace, size_t workspace_bytes,
                                             unsigned long long *prof)
{
    return fb_run((hipStream_t)stream, 1, n, m, nullptr, nullptr, xyz, temp, idx, workspace, workspace_bytes, prof);
}
