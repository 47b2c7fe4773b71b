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
//   round   1. every lane tests the group it owns:  dbox(sample, AABB) >= max  ==> nothing inside
//              can change (dbox uses the fp32 association of the point distance and fp32 rounding is
//              monotone, so dbox <= d(p) for every p inside: min(d(p), temp[p]) == temp[p] exactly);
//           2. the 16 children of each touched group are tested the same way by one DPP row (up to 4
//              groups per wave instruction);
//           3. touched buckets are re-scanned (64 lanes = 64 points), several at a time with all their loads
//              in flight, the reduction chains interleaved;
//           4. the touched groups' entries are rebuilt by a 16-lane row arg-max;
//           5. the group table yields the next sampleS -- several per round, exactly (fm_main_kernel below).
//
// One workgroup of 4 waves (one per SIMD: the per-round work is a dependent chain, extra waves only
// add issue pressure -- a 16-wave version of the single-sample kernel was issue-bound at 4400 cycles per
// round) per batch element.  Group g belongs to wave g % 4; every table entry is written and read by the same
// wave, hence no barrier between update and selection.
#include "fps_bucket.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include <rocprim/rocprim.hpp>

namespace {

constexpr int FB_GS = 16;        // buckets per group = lanes per DPP row

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
    const int w = i;
    a.sp[w] = make_float4(a.xyz[(size_t)o * 3 + 0], a.xyz[(size_t)o * 3 + 1], a.xyz[(size_t)o * 3 + 2],
                          live ? a.temp[o] : -1.0f);
    a.skey[w] = live ? tpu3_fps_tiekey((int)o, lb) : 0xFFFFFFFFu;
}

// (r6) The four launches in front of the register-resident per-level kernels -- bounding box, Morton keys, a rocPRIM
// segmented radix sort (three kernels, 0.32 ms for one cloud's 48 sets of 24 960 points) and the permutation -- as ONE
// kernel: a 1024-thread workgroup per set bins its points by a 15-bit Morton cell (32 cells per axis of the set's box)
// with an LDS counting sort -- 32 768 counters packed two to a word (a set holds <= 25 600 points: 16 bits), the
// atomic's return value is the point's arrival rank inside its cell -- and writes the (x, y, z, running distance) rows
// and tie keys in that order.  The order INSIDE a cell is the arrival order (differs from run to run); the samples do
// not depend on it: the bucketed FPS is exact for any order, only its pruning quality depends on buckets being
// compact -- ~8 points per cell on a surface, so a lane's bucket of 7 - 25 consecutive points spans a few cells either
// way.  Slots past the live size repeat the last live row with temp = -1, as fb_permute_kernel leaves them.
constexpr int FB_BIN_PPT = 25;           // points per thread: 25 600 / 1024

__device__ __forceinline__ uint32_t fb_spread5(uint32_t v)
{
    v &= 0x1Fu;
    v = (v | (v << 8)) & 0x100Fu;
    v = (v | (v << 4)) & 0x10C3u;
    v = (v | (v << 2)) & 0x1249u;
    return v;
}

__global__ __launch_bounds__(1024) void fb_bin_kernel(FbArgs a0)
{
    __shared__ uint32_t cnt[16384];     // cell c: bits [16 (c & 1), +16) of word c >> 1
    __shared__ float red[6][16];
    __shared__ float box[6];
    __shared__ int wsum[16];
    const FbArgs a = fb_elem(a0, blockIdx.x);
    const int n = a.n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (n <= 0)
        return;
    const float *__restrict__ xyz = a.xyz;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = tid; i < n; i += 1024)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = xyz[(size_t)i * 3 + c];
            lo[c] = fminf(lo[c], v);
            hi[c] = fmaxf(hi[c], v);
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float l = -tpu3_wave_max_f32(-lo[c]), h = tpu3_wave_max_f32(hi[c]);
        if (lane == 0) {
            red[c][wave] = l;
            red[3 + c][wave] = h;
        }
    }
    for (int i = tid; i < 16384; i += 1024)
        cnt[i] = 0u;
    __syncthreads();
    if (tid < 6) {
        float v = red[tid][0];
        for (int w = 1; w < 16; ++w)
            v = tid < 3 ? fminf(v, red[tid][w]) : fmaxf(v, red[tid][w]);
        box[tid] = v;
        a.bbox[tid] = v;
    }
    __syncthreads();
    float sc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float ext = box[3 + c] - box[c];
        sc[c] = ext > 0.f ? 32.f / ext : 0.f;
    }
    uint32_t cr[FB_BIN_PPT];            // cell << 16 | rank
#pragma unroll
    for (int j = 0; j < FB_BIN_PPT; ++j) {
        const int i = tid + j * 1024;
        cr[j] = 0u;
        if (i < n) {
            uint32_t cell = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float q = (xyz[(size_t)i * 3 + c] - box[c]) * sc[c];
                q = fminf(fmaxf(q, 0.f), 31.f);
                cell |= fb_spread5((uint32_t)q) << c;
            }
            const uint32_t old = atomicAdd(&cnt[cell >> 1], (cell & 1u) ? 0x10000u : 1u);
            cr[j] = (cell << 16) | ((cell & 1u) ? old >> 16 : old & 0xFFFFu);
        }
    }
    __syncthreads();
    // exclusive scan of the 32 768 counts: thread t owns words 16 t .. 16 t + 15 (cells 32 t .. 32 t + 31)
    {
        uint32_t w[16];
        int sum = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            w[u] = cnt[16 * tid + u];
            sum += (int)(w[u] & 0xFFFFu) + (int)(w[u] >> 16);
        }
        int inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            inc += lane >= d ? o : 0;
        }
        if (lane == 63)
            wsum[wave] = inc;
        __syncthreads();
        int run = inc - sum;
        for (int k = 0; k < wave; ++k)
            run += wsum[k];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int c0 = (int)(w[u] & 0xFFFFu), c1 = (int)(w[u] >> 16);
            cnt[16 * tid + u] = (uint32_t)run | ((uint32_t)(run + c0) << 16);      // the two cells' first positions
            run += c0 + c1;
        }
    }
    __syncthreads();
    const int lb = a.lb;
#pragma unroll
    for (int j = 0; j < FB_BIN_PPT; ++j) {
        const int i = tid + j * 1024;
        if (i < n) {
            const uint32_t cell = cr[j] >> 16, word = cnt[cell >> 1];
            const int pos = (int)((cell & 1u) ? word >> 16 : word & 0xFFFFu) + (int)(cr[j] & 0xFFFFu);
            a.sp[pos] = make_float4(xyz[(size_t)i * 3 + 0], xyz[(size_t)i * 3 + 1], xyz[(size_t)i * 3 + 2], a.temp[i]);
            a.skey[pos] = tpu3_fps_tiekey(i, lb);
        }
    }
    __syncthreads();                    // (the workgroup's own global writes are visible to it behind the barrier)
    if (n < a.npad) {
        const float4 last = a.sp[n - 1];
        for (int i = n + tid; i < a.npad; i += 1024) {
            a.sp[i] = make_float4(last.x, last.y, last.z, -1.0f);
            a.skey[i] = 0xFFFFFFFFu;
        }
    }
}

__global__ __launch_bounds__(256) void fb_writeback_kernel(FbArgs a0)
{
    const FbArgs a = fb_elem(a0, blockIdx.y);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int w = i;
    if (i < a.n && a.m > 0)
        a.temp[tpu3_fps_tiekey_to_index(a.skey[w], a.lb)] = a.sp[w].w;
}

// lane-local best of one bucket (64*PPL points), optionally after folding sample q into it
struct FbCand {
    float t, x, y, z;
    uint32_t key;
};

// Second-largest running distance of a bucket, given every lane's best `c` and lane-local runner-up
// `second` (-2 when the lane holds a single point) and the winning lane: a wave max over the runners-up,
// the winner contributing its second instead of its best.  Exact (equal to the maximum when it is tied).
__device__ __forceinline__ int fb_runner_up(float best, float second, bool is_winner)
{
    return tpu3_wave_max_i32_fast(__float_as_int(is_winner ? second : best));
}

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

// Fold samples into a loaded bucket: lane i of (px, py, pz) holds sample i, `pmask` (wave-uniform) selects the
// samples to fold in (0: none, the bucket is only reduced).  Returns the lane-local best and runner-up.
template <int PPL>
__device__ __forceinline__ FbCand fm_apply(FbBucket<PPL> &b, uint32_t pmask, float px, float py, float pz,
                                           float &second)
{
    float t[PPL];
#pragma unroll
    for (int p = 0; p < PPL; ++p)
        t[p] = b.v[p].w;
    while (pmask) {
        const int i = __builtin_ctz(pmask);
        pmask &= pmask - 1;
        const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px), i));
        const float qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py), i));
        const float qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz), i));
#pragma unroll
        for (int p = 0; p < PPL; ++p)
            t[p] = tpu3_min1(tpu3_sqdist3(b.v[p].x - qx, b.v[p].y - qy, b.v[p].z - qz), t[p]);
    }
    FbCand c{-2.0f, 0.f, 0.f, 0.f, 0xFFFFFFFFu};
    second = -2.0f;
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
        b.nt[p] = t[p];
        if (t[p] > c.t || (t[p] == c.t && b.key[p] < c.key)) {
            second = c.t;
            c.t = t[p]; c.key = b.key[p]; c.x = b.v[p].x; c.y = b.v[p].y; c.z = b.v[p].z;
        } else {
            second = tpu3_max1(second, t[p]);
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
            ib[8 * S + beta] = 0x80000000u;
        }
        return;
    }
    FbBucket<PPL> bk;
    fb_load<PPL>(bk, a.sp, a.skey, beta, lane);
    float second;
    const FbCand c = fm_apply<PPL>(bk, 0u, 0.f, 0.f, 0.f, second);
    int wl;
    const int wmax = tpu3_wave_argmax(__float_as_int(c.t), c.key, wl);
    const int runner = fb_runner_up(c.t, second, lane == wl);
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
        ib[8 * S + beta] = (uint32_t)runner;
    }
}

// ---------------------------------------------------------------------------------------------
// Several samples per round, exactly.
//
// The loop above spends a full round (prune -> re-scan -> refresh -> arg-max -> barrier -> broadcast,
// ~2.8 us of dependent latency) per sample.  Most consecutive samples do not interact, though: late in
// the sampling the leading candidates are scattered over the whole cloud, far outside each other's
// update balls.  With ONE extra number per cell -- an upper bound R_c on the SECOND largest running
// distance inside cell c (cells = groups of 16 buckets) -- a whole prefix of the sampling order can
// be read off the group table at once:
//
//   let R* = max_c R_c.  Sort the cells with M_c > R* by (M_c descending, tie key): c_1, c_2, ...
//   The next samples are the best points p_1, p_2, ..., p_J of c_1, c_2, ... for the longest prefix in which
//   no p_j lies inside the update ball of an earlier member (|p_i - p_j|^2 >= M_j for i < j).
//
// Proof sketch: running distances only decrease, so a stale R_c stays an upper bound.  After c_1..c_{j-1}
// have been sampled and applied, every point other than the members' own best points has a distance
// <= max(M of an unsampled cell, R of a sampled cell) -- and R_c <= R* < M_j, M of the other unsampled
// cells <= M_j by the sort; p_j itself is out of reach of the earlier members, so it still holds M_j (the
// other points of its cell can only have fallen) and is the exact arg-max (ties: equal M are ordered by the reference's tie key, and strictness of M_j > R* keeps runner-ups
// out).  J >= 1 always: c_1 is the plain arg-max.  If no cell beats R* (ties at the top) the round falls
// back to the single arg-max sample.
//
// All J updates are applied in one pass: a touched bucket is re-scanned once with every sample that
// reaches it folded in (min is commutative), and bucket / group entries now also carry their runner-up.
// R* may be one round old (bounds only fall), so candidates, the waves' bests and the new R* travel through
// ONE barrier per round; every wave enters at most FM_WCAP candidates, the best one it leaves out caps what
// may be accepted.  The re-scans of a round are listed first and then run several buckets at a time with
// all their loads in flight.
// ---------------------------------------------------------------------------------------------
struct FmHeader {                   // one per wave and buffer
    int best;                       // best group entry of the wave (distance bits) ...
    uint32_t key;
    float x, y, z;
    int rmax;                       // largest runner-up bound among the wave's groups
    int count;                      // candidates entered (<= FM_WCAP)
    int drop;                       // best candidate NOT entered (INT_MIN if none)
};

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
        const float t = tpu3_min1(tpu3_sqdist3(px[j] - qx, py[j] - qy, pz[j] - qz), pt[j]);
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

// ---------------------------------------------------------------------------------------------
// Register-resident FPS with several samples per round (sets of 4097 .. 25 600 points: the per-level resampling)
// ---------------------------------------------------------------------------------------------
// The scheme of fm_main_kernel (exact: a candidate is a bucket whose maximum beats every bucket's runner-up
// bound R*; the candidates in descending order ARE the next samples as long as none lies inside an earlier one's
// update ball) with the whole set in the register file and a LANE per bucket: lane b of the 1024 owns the R
// Morton-consecutive points [b R, b R + R) -- x, y and the running distance in registers, beyond 13 points per
// lane z in LDS (read only: loads without a dependent store), like the 16-bit original indices (tie keys).  Bucket
// maximum, runner-up and the winner's coordinates are per-lane register state, a sample's box test covers 64
// buckets per instruction, a reached wave updates its R x 64 distances with plain per-lane arithmetic and
// re-derives the lanes' records once per round; only the wave's arg-max and runner-up bound need cross-lane
// reductions -- two per ROUND.  (Round 2's first form kept a 64-point row across the lanes of a wave: every row a
// sample reached cost two dependent wave-wide reductions and a record write, ~700 cycles of latency each, 4 - 5
// rows per wave and round one after the other; 25-point buckets also lower R*: 8.2 samples per round at 24 960
// points against 6.1.)  Up to RL_WCAP candidates per wave enter a round; one barrier, then wave 0 ranks them (a
// ranking repeated by all 16 waves would be issue-bound) while the others wait at a second barrier.
constexpr int RL_EW = 8;            // words per candidate entry (5 used)
constexpr int RL_CAP = 64;          // candidates per round
template <int R> constexpr bool rl_zl() { return R > 13; }       // z coordinates in LDS (registers: x, y, distance)

struct RlShared {
    FmHeader h[2][16];
    uint32_t cand[2][RL_CAP * RL_EW];
    float pick[2][RL_CAP][4];
    int npick[2];
    uint32_t pkey[2][RL_CAP];       // tie keys of the picks (rank order)
    int mrow[RL_CAP];               // the candidates' maxima, compact (wave 0's ranking reads them back as broadcasts)
    int ncand[2];                   // candidates appended this round (may exceed RL_CAP: the surplus is not stored ...)
    int drop[2];                    // ... and the best of the surplus caps what may be accepted
};

constexpr size_t rl_lds_bytes(int r, bool zl, int nw = 16)
{
    return (nw == 16 ? (size_t)1024 * r * 2 : 0) + (zl ? (size_t)nw * 64 * r * 4 : 0) + sizeof(RlShared) + 64;
}

// NW waves per set.  16: a set is a whole compute unit (25 x 1024 points).  The sets of a launch are independent
// dependent chains, each leaving most of its compute unit idle (a round is: a few of the waves update, one wave ranks),
// so SEVERAL sets per compute unit finish sooner than one after the other: sets of 7169 .. 12 800 points run as 8 waves
// of 25 points per lane (two sets per compute unit; their tie keys stay in memory -- read by ties and winners only --
// so that a set's LDS is its z coordinates alone), sets of <= 7168 points keep 16 waves of <= 7 points per lane with the
// registers capped at 64 (two per compute unit as well).  A 4-wave form (TPU3_RL_NW=4) exists for <= 6400 points.
template <int R, bool PROF = false, int NW = 16>
__global__ __launch_bounds__(NW * 64, ((R <= 7 && NW == 16) ? 8 : 4)) void rl_main_kernel(FbArgs a0)
{
    constexpr bool KG = NW < 16;                    // tie keys read from a.skey instead of an LDS copy
    constexpr bool ZL = rl_zl<R>();
    auto now = []() { return (unsigned long long)__builtin_amdgcn_s_memtime(); };
    unsigned long long pc[6] = {0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;     // PROF: apply, select, barrier 1, rank, barrier 2
    unsigned long long pr[5] = {0, 0, 0, 0, 0}, r0 = 0, r1 = 0;        // PROF, wave 0: ranking phases, candidates
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t *i16 = (uint16_t *)smem;                               // (!KG) original index of [wave][j][lane]
    float *zl = (float *)(smem + (KG ? 0 : (size_t)1024 * R * 2));  // (ZL) z of [wave][j][lane]
    RlShared &sh = *(RlShared *)(zl + (ZL ? NW * 64 * R : 0));
    const FbArgs a = fb_elem(a0, blockIdx.x);
    if (a.n <= 0 || a.m <= 0)
        return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lb = a.lb;
    uint16_t *kw = i16 + wave * R * 64 + lane;      // kw[64 * j]
    float *zw = zl + wave * R * 64 + lane;          // (ZL) zw[64 * j]
    auto key_at = [&](int j) __attribute__((always_inline)) -> uint32_t {
        if constexpr (KG) {
            const int slot = tid * R + j;
            return slot < a0.npad ? a.skey[slot] : 0xFFFFFFFFu;
        } else {
            const uint32_t v = kw[64 * j];
            return v == 0xFFFFu ? 0xFFFFFFFFu : tpu3_fps_tiekey((int)v, lb);
        }
    };

    // bucket `tid` = sorted points [tid R, tid R + R)
    float px[R], py[R], pz[ZL ? 1 : R], pt[R];
    float blx = __builtin_inff(), bly = blx, blz = blx, bhx = -blx, bhy = -blx, bhz = -blx;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int slot = tid * R + j;
        float4 v = make_float4(0.f, 0.f, 0.f, -1.0f);
        uint32_t key = 0xFFFFFFFFu;
        if (slot < a0.npad) {
            v = a.sp[slot];
            key = a.skey[slot];
        }
        px[j] = v.x; py[j] = v.y; pt[j] = v.w;
        if (ZL)
            zw[64 * j] = v.z;
        else
            pz[ZL ? 0 : j] = v.z;
        const bool alive = key != 0xFFFFFFFFu;
        if constexpr (!KG)
            kw[64 * j] = alive ? (uint16_t)tpu3_fps_tiekey_to_index(key, lb) : (uint16_t)0xFFFFu;
        blx = alive ? fminf(blx, v.x) : blx; bly = alive ? fminf(bly, v.y) : bly; blz = alive ? fminf(blz, v.z) : blz;
        bhx = alive ? fmaxf(bhx, v.x) : bhx; bhy = alive ? fmaxf(bhy, v.y) : bhy; bhz = alive ? fmaxf(bhz, v.z) : bhz;
    }
    auto ld_z = [&](auto jc) __attribute__((always_inline)) -> float {
        constexpr int j = decltype(jc)::value;
        if constexpr (ZL)
            return zw[64 * j];
        else
            return pz[j];
    };

    // the lane's record: bucket maximum (distance bits), runner-up, the winner's slot and coordinates
    int lmax = (int)0x80000000, lrun = (int)0x80000000, larg = 0;
    float lbx = 0.f, lby = 0.f, lbz = 0.f;
    auto lane_scan = [&]() __attribute__((always_inline)) {
        int best = (int)0x80000000, run = (int)0x80000000, arg = 0;
        float bx = 0.f, by = 0.f, bz = 0.f;
        // maximum and runner-up: two instructions per point (run = the median of (best, t, run) as long as
        // run <= best); then the winner's slot and coordinates by equality
        rb_static_for<0, R>([&](auto jc) __attribute__((always_inline)) {
            const int tb = __float_as_int(pt[decltype(jc)::value]);
            asm("v_med3_i32 %0, %1, %2, %0" : "+v"(run) : "v"(best), "v"(tb));
            best = max(best, tb);
        });
        rb_static_for<0, R>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const bool eq = __float_as_int(pt[j]) == best;
            arg = eq ? j : arg;
            bx = eq ? px[j] : bx; by = eq ? py[j] : by;
            if constexpr (!ZL)
                bz = eq ? pz[j] : bz;
        });
        // equal maxima inside the bucket (duplicated points): the smallest tie key wins
        if (__ballot(run == best && best >= 0)) {
            uint32_t bk = 0xFFFFFFFFu;
            rb_static_for<0, R>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                const int tb = __float_as_int(pt[j]);
                const uint32_t kj = key_at(j);
                const bool take = run == best && tb == best && kj < bk;
                bk = take ? kj : bk;
                arg = take ? j : arg;
                bx = take ? px[j] : bx; by = take ? py[j] : by;
                if constexpr (!ZL)
                    bz = take ? pz[j] : bz;
            });
        }
        if constexpr (ZL)
            bz = zw[64 * arg];
        lmax = best; lrun = run; larg = arg; lbx = bx; lby = by; lbz = bz;
    };
    lane_scan();
    if (tid < 2) {
        sh.ncand[tid] = 0;
        sh.drop[tid] = (int)0x80000000;
    }
    __syncthreads();
    // the wave's box (uniform) and an upper bound of its lanes' maxima: a sample is first tested against these --
    // all samples of the round in ONE evaluation, a lane per sample -- and only the few that may reach the wave
    // go through the per-bucket test
    auto uni = [](float v) __attribute__((always_inline)) {       // (wave-uniform: keep it in a scalar register)
        return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
    };
    const float wlx = uni(-tpu3_wave_max_f32(-blx)), wly = uni(-tpu3_wave_max_f32(-bly));
    const float wlz = uni(-tpu3_wave_max_f32(-blz)), whx = uni(tpu3_wave_max_f32(bhx));
    const float why = uni(tpu3_wave_max_f32(bhy)), whz = uni(tpu3_wave_max_f32(bhz));
    int wbound = 0x7F800000;                        // +inf until the first selection has reduced the maxima

    // pair (i < l) number `lane` of the l-major enumeration 0:(0,1) 1:(0,2) 2:(1,2) 3:(0,3) ... (wave 0's clearance test)
    if (tid == 0)
        a.idx[0] = 0;
    // current samples: lane i < J holds sample i; start with point 0
    float sx = a.xyz[0], sy = a.xyz[1], sz = a.xyz[2];
    int J = 1, r = 1, rstar = 0x7FFFFFFF;
    auto rl = [](float v, int i) __attribute__((always_inline)) {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i));
    };

    // One sample folded into the lane's R running distances.  (r6) The distance chain runs on PACKED fp32 -- two points per
    // v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32, each half the IEEE operation of the scalar instruction (same bits):
    // 3 R + R minima instead of 6 R + R instructions per sample and lane; the update is the phase in which the four waves
    // of a SIMD queue for its VALU (docs/kernels/fps_register_resident_levels.md).
    typedef float rl_f2 __attribute__((ext_vector_type(2)));
    auto fold_sample = [&](float qx, float qy, float qz) __attribute__((always_inline)) {
        const rl_f2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
        rb_static_for<0, R / 2>([&](auto pc_) __attribute__((always_inline)) {
            constexpr int j = 2 * decltype(pc_)::value;
            const rl_f2 X = {px[j], px[j + 1]}, Y = {py[j], py[j + 1]};
            const rl_f2 Z = {ld_z(std::integral_constant<int, j>{}), ld_z(std::integral_constant<int, j + 1>{})};
            const rl_f2 dx = X - qx2, dy = Y - qy2, dz = Z - qz2;
            const rl_f2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
            pt[j] = tpu3_min1(d.x, pt[j]);
            pt[j + 1] = tpu3_min1(d.y, pt[j + 1]);
            if constexpr (ZL && (j / 2) % 4 == 3)
                __builtin_amdgcn_sched_barrier(0);      // (eight LDS reads in flight, not all R: registers)
        });
        if constexpr (R % 2 == 1) {
            constexpr int j = R - 1;
            pt[j] = tpu3_min1(tpu3_sqdist3(px[j] - qx, py[j] - qy, ld_z(std::integral_constant<int, j>{}) - qz), pt[j]);
        }
    };

    // fold the first nj current samples into the buckets they reach.  A sample that does not reach a bucket (box
    // distance >= the bucket's maximum) cannot lower any of its distances, so a reached wave updates all of its
    // lanes unconditionally; a wave no sample reaches does nothing.
    auto apply = [&](int nj) __attribute__((always_inline)) {
        bool touched = false;
        unsigned long long sm = __ballot(lane < nj && fb_dbox(sx, sy, sz, wlx, wly, wlz, whx, why, whz) <
                                                          __int_as_float(wbound));
        while (sm) {
            const int i = __builtin_ctzll(sm);
            sm &= sm - 1;
            const float qx = rl(sx, i), qy = rl(sy, i), qz = rl(sz, i);
            if (!__ballot(fb_dbox(qx, qy, qz, blx, bly, blz, bhx, bhy, bhz) < __int_as_float(lmax)))
                continue;
            touched = true;
            if (PROF) pc[5] += 1;
            fold_sample(qx, qy, qz);
        }
        if (touched)
            lane_scan();
    };

    if (a.m > 1)
        for (int round = 0;; ++round) {
            if (PROF) t0 = now();
            apply(J);
            if (PROF) { t1 = now(); pc[0] += t1 - t0; t0 = t1; }
            // ---- select the next samples -------------------------------------------------------------
            // (r3) against a FRESH bound R*: the waves' runner-up maxima cross a barrier of their own before anybody
            // picks candidates (one more barrier per round; until round 2 the previous round's R* was used: valid, but
            // it sits just above the next candidates -- 9.3 samples per round against 14.2 on a level-4 set,
            // tools/fps_cells_sim.py).  Candidates are appended to ONE list through an LDS counter: no per-wave quota.
            const int par = round & 1;
            uint32_t *cl = sh.cand[par];
            const int mine = lmax;
            uint32_t rk = 0xFFFFFFFFu;
            {
                const int wv = tpu3_wave_max_i32_fast(mine);
                const int wr = tpu3_wave_max_i32_fast(lrun);
                wbound = wv;
                unsigned long long tie = __ballot(mine == wv);
                // the wave's best (ties: the smallest key) for the single-sample fall-back
                if (mine == wv)
                    rk = key_at(larg);
                if (__builtin_popcountll(tie) != 1) {
                    const uint32_t kmin = tpu3_wave_min_u32(mine == wv ? rk : 0xFFFFFFFFu);
                    tie = __ballot(mine == wv && rk == kmin);
                }
                if (lane == (int)__builtin_ctzll(tie)) {
                    FmHeader &h = sh.h[par][wave];
                    h.best = wv; h.key = rk; h.x = lbx; h.y = lby; h.z = lbz;
                    h.rmax = wr;
                }
            }
            if (PROF) { t1 = now(); pc[1] += t1 - t0; t0 = t1; }
            __syncthreads();
            {
                const int sr = lane < NW ? sh.h[par][lane].rmax : (int)0x80000000;
                rstar = __builtin_amdgcn_readlane(tpu3_row_max_i32_fast(sr), 0);
                const bool is_cand = mine > rstar;
                const unsigned long long cm = __ballot(is_cand);
                if (cm) {
                    int base = 0;
                    if (lane == 0)
                        base = atomicAdd(&sh.ncand[par], (int)__builtin_popcountll(cm));
                    base = __builtin_amdgcn_readfirstlane(base);
                    const int pos = base + __builtin_popcountll(cm & ((1ull << lane) - 1ull));
                    if (is_cand && pos < RL_CAP) {
                        if (mine != wbound)
                            rk = key_at(larg);
                        uint32_t *e = cl + pos * RL_EW;
                        e[0] = (uint32_t)mine; e[1] = rk;
                        e[2] = __float_as_uint(lbx); e[3] = __float_as_uint(lby); e[4] = __float_as_uint(lbz);
                    }
                    if (base + (int)__builtin_popcountll(cm) > RL_CAP) {
                        const int d = tpu3_wave_max_i32_fast(is_cand && pos >= RL_CAP ? mine : (int)0x80000000);
                        if (lane == 0)
                            atomicMax(&sh.drop[par], d);
                    }
                }
            }
            __syncthreads();
            if (PROF) { t1 = now(); pc[2] += t1 - t0; t0 = t1; }
            // wave 0 ranks the candidates; the others wait at a second barrier and read the round's samples from LDS
            const int left = a.m - r;
            if (wave == 0) {
                const FmHeader &hh = sh.h[par][lane & (NW - 1)];
                const int sd = lane < NW ? hh.best : (int)0x80000000;
                const uint32_t sk = lane < NW ? hh.key : 0xFFFFFFFFu;
                const float hx = hh.x, hy = hh.y, hz = hh.z;
                const int gbest = __builtin_amdgcn_readlane(tpu3_row_max_i32_fast(sd), 0);
                const int nc = __builtin_amdgcn_readfirstlane(sh.ncand[par]);
                const int gdrop = nc > RL_CAP ? __builtin_amdgcn_readfirstlane(sh.drop[par]) : (int)0x80000000;
                const int total = nc < RL_CAP ? nc : RL_CAP;
                const bool live = lane < total;
                const unsigned long long lm = __ballot(live);
                if (lane == 0) {                            // the other buffer's counters: next round starts from zero
                    sh.ncand[par ^ 1] = 0;
                    sh.drop[par ^ 1] = (int)0x80000000;
                }
                if (PROF) { r0 = now(); pr[0] += r0 - t0; pr[4] += (unsigned long long)total; }
                float qx, qy, qz;
                uint32_t okey;
                int nj;
                // (position-based surplus: if even the best listed candidate is below a dropped one, fall back)
                const int cM0 = live ? (int)cl[(lane & (RL_CAP - 1)) * RL_EW] : (int)0x80000000;
                const bool usable = total >= 1 && __ballot(live && cM0 > gdrop) != 0;
                if (!usable) {
                    // single sample: the plain arg-max over the waves' bests (the reference's tie rule)
                    unsigned long long who = __ballot(lane < NW && sd == gbest);
                    if (__builtin_popcountll(who) != 1) {
                        const uint32_t wk = tpu3_row_min_u32(lane < NW && sd == gbest ? sk : 0xFFFFFFFFu);
                        const uint32_t win = (uint32_t)__builtin_amdgcn_readlane((int)wk, 0);
                        who = __ballot(lane < NW && sd == gbest && sk == win);
                    }
                    const int ww = __builtin_ctzll(who | (1ull << 63)) & (NW - 1);
                    qx = rl(hx, ww); qy = rl(hy, ww); qz = rl(hz, ww);
                    okey = (uint32_t)__builtin_amdgcn_readlane((int)sk, ww);
                    nj = 1;
                } else {
                    const uint32_t *e = cl + (lane & (RL_CAP - 1)) * RL_EW;
                    const int cM = live ? (int)e[0] : (int)0x80000000;
                    const uint32_t cK = live ? e[1] : 0xFFFFFFFFu;
                    const float cx = __uint_as_float(e[2]), cy = __uint_as_float(e[3]), cz = __uint_as_float(e[4]);
                    // rank = number of candidates ahead of this one.  The list is read back as wave-uniform LDS
                    // broadcasts (the scalar bit scan -> readlane -> compare chain cost ~150 cycles per candidate in
                    // this lone wave: 2.7 k cycles per round at 18 candidates)
                    int rank = 0;
                    bool tie = false;
                    sh.mrow[lane & (RL_CAP - 1)] = cM;      // (dead lanes: INT_MIN, never ahead of anybody)
                    for (int c0 = 0; c0 < total; c0 += 16) {
                        int4 mv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            mv[u] = *(const int4 *)(sh.mrow + c0 + 4 * u);      // four broadcast reads in flight
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int m4[4] = {mv[u].x, mv[u].y, mv[u].z, mv[u].w};
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                rank += m4[v] > cM ? 1 : 0;
                                tie |= m4[v] == cM && c0 + 4 * u + v != lane;
                            }
                        }
                    }
                    if (__ballot(live && tie)) {            // equal maxima among candidates: order by the tie key
                        rank = 0;
                        for (int i = 0; i < total; ++i) {
                            const int mi = (int)cl[i * RL_EW];
                            const uint32_t ki = cl[i * RL_EW + 1];
                            rank += (mi > cM || (mi == cM && ki < cK)) ? 1 : 0;
                        }
                    }
                    if (PROF) { r1 = now(); pr[1] += r1 - r0; r0 = r1; }
                    // into rank order through LDS: pick[rank] = (x, y, z, M), pkey[rank]
                    if (live) {
                        *(float4 *)sh.pick[par][rank] = make_float4(cx, cy, cz, __int_as_float(cM));
                        sh.pkey[par][rank] = cK;
                    }
                    const float4 mine4 = *(const float4 *)sh.pick[par][lane & (RL_CAP - 1)];
                    qx = mine4.x; qy = mine4.y; qz = mine4.z;
                    okey = sh.pkey[par][lane & (RL_CAP - 1)];
                    const int sM = __float_as_int(mine4.w);
                    int jmax = __builtin_popcountll(__ballot(lane < total && sM > gdrop));
                    jmax = jmax < 1 ? 1 : jmax;
                    jmax = jmax < left ? jmax : left;
                    // longest prefix in which no member lies inside the update ball of an earlier member: the
                    // smallest l with d(sample i, sample l) < M_l for some i < l.  All pairs (i < l < jmax), 64 per
                    // pass, a lane per pair in l-major order (0:(0,1) 1:(0,2) 2:(1,2) 3:(0,3) ...): the first pass
                    // with a hit holds the smallest l.
                    {
                        const int npair = jmax * (jmax - 1) / 2;
                        for (int t0 = 0; t0 < npair; t0 += 64) {
                            // (the pair of a lane is derived here, every pass: kept from the kernel's start it was two
                            // registers of all 16 waves for wave 0's sake, in a kernel that spills at 128)
                            const int t = t0 + lane;
                            int pl = (int)((1.f + sqrtf(1.f + 8.f * (float)t)) * 0.5f);
                            pl -= pl * (pl - 1) / 2 > t ? 1 : 0;
                            pl += (pl + 1) * pl / 2 <= t ? 1 : 0;
                            const int pi = t - pl * (pl - 1) / 2;
                            const bool ok = pl < jmax;
                            const float4 L4 = *(const float4 *)sh.pick[par][ok ? pl : 0];
                            const float4 I4 = *(const float4 *)sh.pick[par][ok ? pi : 0];
                            const float d = tpu3_sqdist3(L4.x - I4.x, L4.y - I4.y, L4.z - I4.z);
                            const unsigned long long hit = __ballot(ok && d < L4.w);
                            if (hit) {
                                jmax = __builtin_amdgcn_readlane(pl, (int)__builtin_ctzll(hit));
                                break;
                            }
                        }
                    }
                    nj = jmax;
                    if (PROF) { r1 = now(); pr[2] += r1 - r0; r0 = r1; }
                }
                nj = nj < left ? nj : left;
                if (!usable && lane == 0) {
                    sh.pick[par][0][0] = qx; sh.pick[par][0][1] = qy; sh.pick[par][0][2] = qz;
                }
                if (lane < nj)
                    a.idx[r + lane] = tpu3_fps_tiekey_to_index(okey, lb);
                if (lane == 0)
                    sh.npick[par] = nj;
            }
            if (PROF) { t1 = now(); pc[3] += t1 - t0; t0 = t1; }
            __syncthreads();
            if (PROF) { t1 = now(); pc[4] += t1 - t0; t0 = t1; }
            J = sh.npick[par];
            sx = sh.pick[par][lane & (RL_CAP - 1)][0];
            sy = sh.pick[par][lane & (RL_CAP - 1)][1];
            sz = sh.pick[par][lane & (RL_CAP - 1)][2];
            r += J;
            if (r >= a.m) {
                if (a0.prof && blockIdx.x == 0 && tid == 0) {       // development probe: rounds, samples
                    a0.prof[0] = (unsigned long long)(round + 1);
                    a0.prof[1] = (unsigned long long)r;
                }
                if (PROF && a0.prof && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 1))
                    for (int i = 0; i < 6; ++i)
                        a0.prof[2 + wave * 6 + i] = pc[i];
                if (PROF && a0.prof && blockIdx.x == 0 && tid == 0)
                    for (int i = 0; i < 5; ++i)
                        a0.prof[46 + i] = pr[i];
                if (PROF && a0.prof && blockIdx.x == 0 && lane == 0) {      // every wave: apply cycles, updates
                    a0.prof[14 + wave * 2] = pc[0];
                    a0.prof[15 + wave * 2] = pc[5];
                }
                if (J > 1) {                            // every sample but the last one updates `temp`
                    for (int i = 0; i + 1 < J; ++i) {
                        fold_sample(rl(sx, i), rl(sy, i), rl(sz, i));
                    }
                }
                break;
            }
        }
    // final running distances, back in the caller's order
    rb_static_for<0, R>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if constexpr (KG) {
            const uint32_t key = key_at(j);
            if (key != 0xFFFFFFFFu)
                a.temp[tpu3_fps_tiekey_to_index(key, lb)] = pt[j];
        } else {
            const uint32_t v = kw[64 * j];
            if (v != 0xFFFFu)
                a.temp[v] = pt[j];
        }
    });
}

// ---------------------------------------------------------------------------------------------
// (r3) Memory-resident FPS with a LANE per bucket (25 601 .. 262 144 points: the metric's final 239 616 -> 80 000)
// ---------------------------------------------------------------------------------------------
// fm_main_kernel re-scans a reached 64-point bucket with a whole wave: two wave-wide reductions and a table write
// per bucket, four buckets in flight, 12 buckets per wave and round -- half of a round -- and its candidate cells are
// the 1024-point groups, whose runner-up bound R* admits ~17 samples per round.  rl_main_kernel showed what the
// lane-per-bucket layout buys on sets that fit the register file; this is the same scheme with the points in
// memory:
//   * a BUCKET is 16 Morton-consecutive points owned by a lane, a TILE 64 buckets (1024 points) owned by a wave
//     (tile t belongs to wave t % 16): a bucket is 256 contiguous bytes of the Morton-ordered slab;
//   * per bucket: the maximum running distance and the position of that point live in LDS (5 bytes per bucket,
//     80 KB at 262 144 points), the fp16 box and the runner-up in a 16-byte global record read when the bucket's
//     tile is reached; per tile: box, maximum and runner-up bound in the registers of four lanes of the owner wave,
//     which test the round's samples against it four at a time;
//   * the reached buckets of a round (a handful per tile, found by the owner wave with a lane per bucket) go on one
//     work list and are updated by a DPP ROW each, a lane per point, dealt over all waves; only when a sample's
//     ball covers most of a tile (the first rounds) the owner updates it on the spot with a lane per bucket;
//   * candidates are BUCKETS (bmax > R*, R* = the largest runner-up of any bucket, fresh): 16-point cells admit
//     ~35-40 samples per round on the metric's cloud (tools/fps_cells_sim.py: 20.8 on average over the first
//     12 000 samples, 40 in the last third) against 17 for 1024-point cells; they are appended to one list through
//     an LDS counter, wave 0 ranks them and finds the longest clear prefix as in rl_main_kernel.
// Exact for the same reason as the other bucketed kernels (box distance in the point distance's association is a
// lower bound of every computed distance; stale bounds stay bounds).
constexpr int FL_CAP = 64;               // samples per round
constexpr int FL_LIST = 512;             // candidates a round may list
constexpr int FL_EW = 8;                 // words per candidate entry (5 used)
constexpr int FL_WORK = 4096;            // work list entries: fewer than FL_DENSE reached buckets per tile x 256 tiles
constexpr int FL_DENSE = 16;             // a tile with this many reached buckets is updated on the spot
constexpr int FL_P2 = 3;                 // phase-2 steps of a wave whose points are fetched together
constexpr int FL_VIS = 4096;             // L3: entries of the visit list (reached tiles of a round: all of them at most)
constexpr bool FL_DEAL2 = false;         // two levels: the reached tiles dealt out over the waves too
constexpr int FL_CB = 8;                 // L3: tiles of a wave whose maxima are fetched together when candidates are listed
constexpr int FL_VB = 4;                 // L3: visits of a wave whose records are fetched together
constexpr int FL_TMAX = 4096;            // tiles of the three-level form (256 super-tiles of 16): 4 194 304 points

struct FlShared {
    FmHeader h[2][16];
    uint32_t cand[2][FL_LIST * 2 + 2 * FL_CAP];     // (maximum, slot) per candidate; the selected / ranked slots
    float pick[2][FL_CAP][4];       // the round's samples in rank order: x, y, z, distance
    uint32_t pkey[2][FL_CAP];
    int mrow[FL_CAP];
    int npick[2];
    int ncand[2];
    uint32_t minkey;                // arg-max with the tie rule (ties at the top)
    int nwork;                      // entries on the work list
    int nvis, vnext;                // L3: entries on the visit list, the next one to hand out
    int jclear;                     // length of the clear prefix of the round's ranked candidates
    unsigned long long stat[8];     // rounds, samples, overflow rounds, tie rounds; wave 0's cycles in apply,
                                    // collecting candidates (incl. its barriers), ranking; tile visits of wave 0
};

static_assert(offsetof(FlShared, cand) % 16 == 0 && offsetof(FlShared, mrow) % 16 == 0 && offsetof(FlShared, pick) % 16 == 0,
              "vector reads of the lists");
constexpr size_t fl_lds_bytes(int ntile, bool l3)
{
    return (l3 ? (size_t)FL_TMAX * 8 + FL_VIS * 12 : (((size_t)ntile * 64 * 5 + 15) & ~(size_t)15) + 256 * 12) + 512 * 4 +
           (size_t)FL_WORK * 12 + sizeof(FlShared) + 64;
}

// bucket / tile records of the initial state: one wave per tile
__global__ __launch_bounds__(64) void fl_init_kernel(FbArgs a0)
{
    const FbArgs a = fb_elem(a0, blockIdx.y);
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= a0.ntile)
        return;
    float bl[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float bh[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    int best = (int)0x80000000, run = (int)0x80000000, arg = 0;
    uint32_t bkey = 0xFFFFFFFFu;
    for (int j = 0; j < FL_R; ++j) {
        const int w = (t * 64 + lane) * FL_R + j;
        const float4 v = a.sp[w];
        const uint32_t key = a.skey[w];
        if (key != 0xFFFFFFFFu) {
            bl[0] = fminf(bl[0], v.x); bl[1] = fminf(bl[1], v.y); bl[2] = fminf(bl[2], v.z);
            bh[0] = fmaxf(bh[0], v.x); bh[1] = fmaxf(bh[1], v.y); bh[2] = fmaxf(bh[2], v.z);
        }
        const int tb = __float_as_int(v.w);
        if (tb > best || (tb == best && key < bkey)) {
            run = best; best = tb; bkey = key; arg = j;
        } else {
            run = max(run, tb);
        }
    }
    uint32_t h[6];
    for (int c = 0; c < 3; ++c) {
        h[c] = __half_as_ushort(__float2half_rd(bl[c]));
        h[3 + c] = __half_as_ushort(__float2half_ru(bh[c]));
    }
    const int b = t * 64 + lane;
    a.rec[b] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), (uint32_t)run);
    a.bm0[b] = best;
    a.ba0[b] = (uint8_t)arg;
    float tl[3], th[3];
    for (int c = 0; c < 3; ++c) {
        tl[c] = -tpu3_wave_max_f32(-bl[c]);
        th[c] = tpu3_wave_max_f32(bh[c]);
    }
    const int tm = tpu3_wave_max_i32_fast(best), tr = tpu3_wave_max_i32_fast(run);
    if (lane == 0) {
        float *o = a.tt + t * 8;
        o[0] = tl[0]; o[1] = tl[1]; o[2] = tl[2]; o[3] = th[0]; o[4] = th[1]; o[5] = th[2];
        o[6] = __int_as_float(tm); o[7] = __int_as_float(tr);
    }
}

// L3 (more than 256 tiles, up to 4096: config C5's 3.83 M points): one more level.  The unit a lane quad owns and
// tests the samples against is a SUPER-TILE of 16 tiles; a reached super-tile reads its 16 tile boxes (one lane
// each), a reached tile is visited as in the two-level form.  The buckets' maxima and arg-max positions do not
// fit LDS (1.2 MB) and live in memory next to the records; LDS keeps the maxima and runner-up bounds of the
// tiles (32 KB) and of the super-tiles.
template <bool PROF, bool L3>
__global__ __launch_bounds__(1024) void fl_main_kernel(FbArgs a0)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FbArgs a = fb_elem(a0, blockIdx.x);
    if (a.n <= 0 || a.m <= 0)
        return;
    const int ntile = (a.n + FL_TP - 1) / FL_TP;            // live tiles of this element
    // two levels: bucket maxima / positions [ntile * 64] | tile maxima, runner-up bounds [256] each | work list | shared
    // L3:         tile maxima, runner-up bounds [4096] each | super-tile maxima, bounds [256] each | work list |
    //             visit list | shared
    int *bmax = (int *)smem;
    uint8_t *barg = (uint8_t *)(bmax + a0.ntile * 64);
    int *tmx = L3 ? (int *)smem : (int *)(smem + (((size_t)a0.ntile * 64 * 5 + 15) & ~(size_t)15));
    int *trn = tmx + (L3 ? FL_TMAX : 256);
    int *gmx = L3 ? trn + FL_TMAX : tmx;            // the level the quads own: super-tiles (L3) or the tiles themselves
    int *grn = L3 ? gmx + 256 : trn;
    uint32_t *work = (uint32_t *)(grn + 256);                                           // [FL_WORK][3]
    uint32_t *vis = work + FL_WORK * 3;                                                 // L3: [FL_VIS][3], the visit list
    FlShared &sh = *(FlShared *)(vis + (L3 ? FL_VIS * 3 : 256 * 3));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lb = a.lb;
    float4 *__restrict__ TP = a.sp;
    const uint32_t *__restrict__ TK = a.skey;

    if constexpr (!L3)
        for (int i = tid; i < ntile * 64; i += 1024) {
            bmax[i] = a.bm0[i];
            barg[i] = a.ba0[i];
        }
    // a bucket's maximum / position of its best point: LDS, or (L3) the arrays the init kernel wrote, updated in place
    auto st_bm = [&](int b, int best, int arg) __attribute__((always_inline)) {
        if constexpr (L3) {
            a.bm0[b] = best;
            a.ba0[b] = (uint8_t)arg;
        } else {
            bmax[b] = best;
            barg[b] = (uint8_t)arg;
        }
    };
    // the unit of this lane's quad (a tile; L3: a super-tile of 16 tiles): slot = lane / 4, unit = slot * 16 + wave
    const int slot = lane >> 2, quad = lane & 3;
    const int tq = slot * 16 + wave;
    const int nunit = L3 ? (ntile + 15) >> 4 : ntile;
    const bool tvalid = tq < nunit;
    float tbx[6];
    for (int i = tid; i < (L3 ? FL_TMAX : 256); i += 1024) {
        tmx[i] = i < ntile ? __float_as_int(a.tt[i * 8 + 6]) : (int)0x80000000;
        trn[i] = i < ntile ? __float_as_int(a.tt[i * 8 + 7]) : (int)0x80000000;
    }
    if constexpr (L3) {
        // box = union of the super-tile's tile boxes, maxima over its tiles (every lane of the quad walks all 16)
        float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
        float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
        int um = (int)0x80000000, ur = (int)0x80000000;
        for (int l = 0; l < 16; ++l) {
            const int t = tq * 16 + l;
            if (tvalid && t < ntile) {
                const float *o = a.tt + t * 8;
                lo[0] = fminf(lo[0], o[0]); lo[1] = fminf(lo[1], o[1]); lo[2] = fminf(lo[2], o[2]);
                hi[0] = fmaxf(hi[0], o[3]); hi[1] = fmaxf(hi[1], o[4]); hi[2] = fmaxf(hi[2], o[5]);
                um = max(um, __float_as_int(o[6]));
                ur = max(ur, __float_as_int(o[7]));
            }
        }
        const bool ok = tvalid && lo[0] <= hi[0];
        tbx[0] = ok ? lo[0] : __builtin_inff(); tbx[1] = ok ? lo[1] : __builtin_inff(); tbx[2] = ok ? lo[2] : __builtin_inff();
        tbx[3] = ok ? hi[0] : __builtin_inff(); tbx[4] = ok ? hi[1] : __builtin_inff(); tbx[5] = ok ? hi[2] : __builtin_inff();
        if (quad == 0 && tq < 256) {
            gmx[tq] = um;
            grn[tq] = ur;
        }
        if (tid < 256 && tid >= nunit) {
            gmx[tid] = (int)0x80000000;
            grn[tid] = (int)0x80000000;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 6; ++c)
            tbx[c] = tvalid ? a.tt[tq * 8 + c] : __builtin_inff();
    }
    if (tid < 2)
        sh.ncand[tid] = 0;
    if (tid == 0) {
        sh.nwork = 0;
        sh.nvis = 0;
        sh.vnext = 0;
    }
    if (tid < 8)
        sh.stat[tid] = 0;
    const bool prof = PROF && a0.prof != nullptr && blockIdx.x == 0;
    unsigned long long c_apply = 0, c_coll = 0, c_rank = 0, c_vis = 0, c0 = 0, c1 = 0;
    unsigned long long c_p1 = 0, c_b1 = 0, c_p2 = 0, c_b2 = 0, d0 = 0, d1 = 0;
    unsigned long long c_pass = 0, c_tot = 0, c_nsel = 0, c_cut = 0;     // collect passes, listed / selected candidates, rounds cut by clearance
    if (tid == 0) {
        a.idx[0] = 0;
        sh.pick[1][0][0] = a.xyz[0]; sh.pick[1][0][1] = a.xyz[1]; sh.pick[1][0][2] = a.xyz[2];
        sh.pick[1][0][3] = 0.f;
    }
    __syncthreads();
    if (a.m <= 1)
        return;                                     // the reference's loop body never runs: temp untouched

    // pair (i < l) number `lane` of the l-major enumeration (wave 0's clearance test, first pass)
    int pair_l = (int)((1.f + sqrtf(1.f + 8.f * (float)lane)) * 0.5f);
    pair_l -= pair_l * (pair_l - 1) / 2 > lane ? 1 : 0;
    pair_l += (pair_l + 1) * pair_l / 2 <= lane ? 1 : 0;
    const int pair_i = lane - pair_l * (pair_l - 1) / 2;

    auto quad_or = [](uint32_t v) __attribute__((always_inline)) {
        v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
        v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);      // quad_perm [2,3,0,1]
        return v;
    };

    // ---- one bucket (the lane's): fold the samples of `sm` into its 16 distances, re-derive its record -----------
    // (wave-uniform control flow around it; `act` = this lane has a bucket to update)
    auto update_bucket = [&](bool act, int b, unsigned long long sm0, int cur, int &best, int &run)
        __attribute__((always_inline)) {
        // (two halves of 8 points: all 16 float4 at once cost 28 spilled registers in the whole kernel)
        float4 *__restrict__ base = TP + (size_t)b * FL_R;
        int arg = 0;
        best = (int)0x80000000; run = (int)0x80000000;
#pragma unroll 1
        for (int h = 0; h < FL_R; h += 8) {
            float4 pt[8];
            float nt[8];
            if (act) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    pt[j] = base[h + j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                nt[j] = act ? pt[j].w : 0.f;
            for (unsigned long long sm = sm0; sm; sm &= sm - 1) {
                const float4 p = *(const float4 *)sh.pick[cur][__builtin_ctzll(sm)];
                if (act) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        nt[j] = tpu3_min1(tpu3_sqdist3(pt[j].x - p.x, pt[j].y - p.y, pt[j].z - p.z), nt[j]);
                }
            }
            if (act) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int tb = __float_as_int(nt[j]);
                    asm("v_med3_i32 %0, %1, %2, %0" : "+v"(run) : "v"(best), "v"(tb));
                    arg = tb > best ? h + j : arg;          // (first of equal maxima; ties are settled below)
                    best = max(best, tb);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (nt[j] != pt[j].w)
                        ((float *)(base + h + j))[3] = nt[j];
            }
        }
        // equal maxima inside a bucket (duplicated points): the smallest tie key wins
        if (__ballot(act && run == best && best >= 0)) {
            if (act && run == best && best >= 0) {
                uint32_t bk = 0xFFFFFFFFu;
                for (int j = 0; j < FL_R; ++j) {
                    const uint32_t kj = TK[(size_t)b * FL_R + j];
                    const bool take = __float_as_int(base[j].w) == best && kj < bk;
                    bk = take ? kj : bk;
                    arg = take ? j : arg;
                }
            }
        }
        if (act) {
            st_bm(b, best, arg);
            ((uint32_t *)(a.rec + b))[3] = (uint32_t)run;
        }
    };

    // ---- fold the first nj samples of pick[cur] into every bucket they reach ---------------------------------
    // Two phases.  (1) every wave walks the tiles of its own that a sample may reach: bucket records, box tests;
    // the few reached buckets of a tile go on ONE work list (bucket, its samples); a tile with many reached buckets
    // -- the first rounds, when a sample's ball covers the whole cloud -- is updated on the spot, a lane per bucket.
    // (2) the list is worked off 64 entries per wave pass, a lane per entry: the lanes of a pass all have a bucket
    // to update.  (A first version updated every reached tile on the spot: 2-4 of 64 lanes had anything to do, and
    // the 70 tile visits of a round saturated the compute unit's VALU: 19-35 k cycles per round and wave.)
    auto apply = [&](int nj, int cur) __attribute__((always_inline)) {
        uint32_t mlo = 0, mhi = 0;
        if (prof) d0 = __builtin_amdgcn_s_memtime();
        if (tvalid) {
            const float tmf = __int_as_float(gmx[tq]);
            for (int i = quad; i < nj; i += 4) {
                const float4 p = *(const float4 *)sh.pick[cur][i];
                const bool hit = fb_dbox(p.x, p.y, p.z, tbx[0], tbx[1], tbx[2], tbx[3], tbx[4], tbx[5]) < tmf;
                mlo |= (hit && i < 32) ? (1u << i) : 0u;
                mhi |= (hit && i >= 32) ? (1u << (i - 32)) : 0u;
            }
        }
        mlo = quad_or(mlo);
        mhi = quad_or(mhi);
        // ---- one tile: which of its 64 buckets do the samples of `smt` reach?  few: on the work list; many: on the spot
        // (rc, bm: the record and maximum of this lane's bucket, loaded by the caller)
        auto visit_tile = [&](int t, unsigned long long smt, uint4 rc, int bm) __attribute__((always_inline)) {
            if (PROF) c_vis += 1;
            const int b = t * 64 + lane;
            const float lx = fb_half_lo(rc.x), ly = fb_half_hi(rc.x), lz = fb_half_lo(rc.y);
            const float hx = fb_half_hi(rc.y), hy = fb_half_lo(rc.z), hz = fb_half_hi(rc.z);
            unsigned long long mine = 0;            // the samples that reach THIS lane's bucket
            for (unsigned long long sm = smt; sm; sm &= sm - 1) {
                const int i = __builtin_ctzll(sm);
                const float4 p = *(const float4 *)sh.pick[cur][i];
                mine |= fb_dbox(p.x, p.y, p.z, lx, ly, lz, hx, hy, hz) < __int_as_float(bm) ? (1ull << i) : 0ull;
            }
            const bool reached = mine != 0;
            const unsigned long long rm = __ballot(reached);
            if (!rm)
                return;
            const int nreach = __builtin_popcountll(rm);
            int base = 0;
            bool dense = nreach >= FL_DENSE;
            if (!dense) {
                if (lane == 0)
                    base = atomicAdd(&sh.nwork, nreach);
                base = __builtin_amdgcn_readfirstlane(base);
                dense = base + nreach > FL_WORK;        // (list full -- L3, large balls: on the spot; the reserved slots are voided)
            }
            if (dense) {
                if (nreach < FL_DENSE && reached) {
                    const int pos = base + __builtin_popcountll(rm & ((1ull << lane) - 1ull));
                    if (pos < FL_WORK)
                        work[3 * pos] = 0xFFFFFFFFu;
                }
                int best, run;
                update_bucket(reached, b, smt, cur, best, run);
                int tm = reached ? best : bm, tr = reached ? run : (int)rc.w;
                tpu3_wave_max_i32_fast_x2(tm, tr);
                if (lane == 0) {
                    tmx[t] = tm;
                    trn[t] = tr;
                }
                return;
            }
            // the tile's maxima over the buckets NOT reached; phase 2 adds the reached ones' (atomic max)
            int um = reached ? (int)0x80000000 : bm, ur = reached ? (int)0x80000000 : (int)rc.w;
            tpu3_wave_max_i32_fast_x2(um, ur);
            if (lane == 0) {
                tmx[t] = um;
                trn[t] = ur;
            }
            if (reached) {
                uint32_t *e = work + 3 * (base + __builtin_popcountll(rm & ((1ull << lane) - 1ull)));
                e[0] = (uint32_t)b; e[1] = (uint32_t)mine; e[2] = (uint32_t)(mine >> 32);
            }
        };
        // Two levels: every wave visits the reached tiles of its own quads.  (Dealing them out over all waves through a
        // visit list, with the records of four visits in flight, was measured: 49.6 vs 45.1 ms -- with the maxima in
        // LDS the phase is bound by the compute unit's VALU throughput, ~100 instructions per visit and 70 visits per
        // round, not by which wave runs them.)
        // L3: the records AND the maxima come from memory, a visit is a dependent round trip (2 k cycles) in front of
        // 600 cycles of work, and the waves' shares differ by 2x: the reached tiles go on a visit list, a barrier, and
        // every wave takes every 16th entry, four visits' loads in flight.
        const unsigned long long touched0 = __ballot((mlo | mhi) != 0 && quad == 0);
        if constexpr (!L3 && !FL_DEAL2) {
            for (unsigned long long touched = touched0; touched; touched &= touched - 1) {
                const int L = __builtin_ctzll(touched);
                const int u = (L >> 2) * 16 + wave;
                const uint32_t slo = (uint32_t)__builtin_amdgcn_readlane((int)mlo, L);
                const uint32_t shi = (uint32_t)__builtin_amdgcn_readlane((int)mhi, L);
                visit_tile(u, ((unsigned long long)shi << 32) | slo, a.rec[u * 64 + lane], bmax[u * 64 + lane]);
            }
        } else if constexpr (!L3) {
            int base = 0;
            if (lane == 0 && touched0)
                base = atomicAdd(&sh.nvis, (int)__builtin_popcountll(touched0));
            base = __builtin_amdgcn_readfirstlane(base);
            if ((mlo | mhi) != 0 && quad == 0) {
                uint32_t *v = vis + 3 * (base + __builtin_popcountll(touched0 & ((1ull << lane) - 1ull)));
                v[0] = (uint32_t)tq; v[1] = mlo; v[2] = mhi;
            }
        } else {
            // the touched super-tiles' 16 tiles, a lane each and FOUR super-tiles per step (their tile boxes are one
            // round trip to memory): box and maximum, the samples that may reach the tile.  (The super-tiles and their
            // samples are handed to the lanes through this wave's slice of the idle candidate list.)
            uint32_t *ul = sh.cand[cur ^ 1] + wave * 48;
            const int ntouch = __builtin_popcountll(touched0);
            if ((mlo | mhi) != 0 && quad == 0) {
                uint32_t *e = ul + 3 * __builtin_popcountll(touched0 & ((1ull << lane) - 1ull));
                e[0] = (uint32_t)tq; e[1] = mlo; e[2] = mhi;
            }
            for (int e0 = 0; e0 < ntouch; e0 += 4) {
                const int ei = e0 + (lane >> 4);
                const bool ev = ei < ntouch;
                const uint32_t *e = ul + 3 * (ev ? ei : 0);
                const int t = (int)e[0] * 16 + (lane & 15);
                const bool tv = ev && t < ntile;
                const float *o = a.tt + (tv ? t : 0) * 8;
                const float4 o0 = *(const float4 *)o, o1 = *(const float4 *)(o + 4);
                const float tmf = tv ? __int_as_float(tmx[t]) : -1.f;
                unsigned long long minet = 0;
                for (unsigned long long sm = tv ? ((unsigned long long)e[2] << 32) | e[1] : 0ull; sm; sm &= sm - 1) {
                    const int i = __builtin_ctzll(sm);
                    const float4 p = *(const float4 *)sh.pick[cur][i];
                    minet |= fb_dbox(p.x, p.y, p.z, o0.x, o0.y, o0.z, o0.w, o1.x, o1.y) < tmf ? (1ull << i) : 0ull;
                }
                const unsigned long long tm = __ballot(minet != 0);
                int base = 0;
                if (lane == 0 && tm)
                    base = atomicAdd(&sh.nvis, (int)__builtin_popcountll(tm));
                base = __builtin_amdgcn_readfirstlane(base);
                if (minet != 0) {
                    uint32_t *v = vis + 3 * (base + __builtin_popcountll(tm & ((1ull << lane) - 1ull)));
                    v[0] = (uint32_t)t; v[1] = (uint32_t)minet; v[2] = (uint32_t)(minet >> 32);
                }
            }
        }
        if constexpr (L3 || FL_DEAL2) {
            __syncthreads();
            const int nv = sh.nvis;
            for (;;) {
                int i0 = 0;
                if (lane == 0)
                    i0 = atomicAdd(&sh.vnext, FL_VB);
                i0 = __builtin_amdgcn_readfirstlane(i0);
                if (i0 >= nv)
                    break;
                int tv[FL_VB], bmv[FL_VB];
                uint4 rcv[FL_VB];
#pragma unroll
                for (int k = 0; k < FL_VB; ++k) {
                    const int i = i0 + k;
                    tv[k] = (int)vis[3 * (i < nv ? i : i0)];
                    rcv[k] = a.rec[tv[k] * 64 + lane];
                    bmv[k] = L3 ? a.bm0[tv[k] * 64 + lane] : bmax[tv[k] * 64 + lane];
                }
                // (ONE copy of the visit: the batch rotates through the first slot)
#pragma unroll 1
                for (int i = i0; i < nv && i < i0 + FL_VB; ++i) {
                    const uint32_t *e = vis + 3 * i;
                    visit_tile(tv[0], ((unsigned long long)e[2] << 32) | e[1], rcv[0], bmv[0]);
#pragma unroll
                    for (int k = 0; k + 1 < FL_VB; ++k) {
                        tv[k] = tv[k + 1]; rcv[k] = rcv[k + 1]; bmv[k] = bmv[k + 1];
                    }
                }
            }
        }
        if (prof) { d1 = __builtin_amdgcn_s_memtime(); c_p1 += d1 - d0; d0 = d1; }
        __syncthreads();
        if (prof) { d1 = __builtin_amdgcn_s_memtime(); c_b1 += d1 - d0; d0 = d1; }
        // phase 2: a DPP row of 16 lanes per listed bucket, a lane per point -- one coalesced 256-byte read per
        // bucket, seven instructions per sample, the record by row reductions; four buckets per wave step, the
        // steps dealt over the 16 waves (a lane per bucket here took 800 instructions for ONE pass, 8 k cycles in a
        // lone wave: the sixteen points of a bucket are the parallelism a few reached buckets have)
        const int E = min(sh.nwork, FL_WORK);
        const int row = lane >> 4, col = lane & 15;
        for (int g0 = wave * 4; g0 < E; g0 += 64 * FL_P2) {
          // the points of up to FL_P2 steps of this wave in flight together (a step is one dependent trip to L2)
          float4 ptv[FL_P2];
          int bv[FL_P2];
#pragma unroll
          for (int k = 0; k < FL_P2; ++k) {
              const int ee = g0 + 64 * k + row;
              bv[k] = (int)work[3 * (ee < E ? ee : 0)];
              ptv[k] = TP[(size_t)(bv[k] >= 0 ? bv[k] : 0) * FL_R + col];
          }
#pragma unroll
          for (int k = 0; k < FL_P2; ++k) {
            const int e0 = g0 + 64 * k;
            if (e0 >= E)
                break;
            const bool inl = e0 + row < E;
            const uint32_t *e = work + 3 * (inl ? e0 + row : 0);
            const bool act = inl && bv[k] >= 0;                 // (a voided slot: its tile was updated on the spot)
            const int b = act ? bv[k] : 0;
            unsigned long long mine = act ? (((unsigned long long)e[2] << 32) | e[1]) : 0ull;
            float4 *__restrict__ pp = TP + (size_t)b * FL_R + col;
            const float4 pt = ptv[k];
            float nt = pt.w;
            while (__ballot(mine != 0)) {           // every row walks ITS bucket's samples
                const bool go = mine != 0;
                const float4 p = *(const float4 *)sh.pick[cur][go ? __builtin_ctzll(mine) : 0];
                mine &= mine - 1;
                const float d = tpu3_min1(tpu3_sqdist3(pt.x - p.x, pt.y - p.y, pt.z - p.z), nt);
                nt = go ? d : nt;
            }
            const int tb = act ? __float_as_int(nt) : (int)0x80000000;
            const int best = tpu3_row_max_i32_fast(tb);
            // the winner: the only lane at the maximum, or -- duplicated points -- the one with the smallest tie key
            unsigned long long tie = __ballot(act && tb == best);
            const unsigned long long rowm = 0xFFFFull << (row * 16);
            bool single = true;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                single &= __builtin_popcountll(tie & (0xFFFFull << (rr * 16))) <= 1;
            if (!single) {
                const uint32_t kj = act && tb == best ? TK[(size_t)b * FL_R + col] : 0xFFFFFFFFu;
                const uint32_t km = tpu3_row_min_u32(kj);
                tie = __ballot(act && tb == best && kj == km);
            }
            const bool winner = act && ((tie >> lane) & 1ull) != 0 && (tie & rowm & ((1ull << lane) - 1ull)) == 0;
            const int run = tpu3_row_max_i32_fast(winner || !act ? (int)0x80000000 : tb);
            if (act && nt != pt.w)
                ((float *)pp)[3] = nt;
            if (winner) {
                st_bm(b, best, col);
                ((uint32_t *)(a.rec + b))[3] = (uint32_t)run;
                atomicMax(&tmx[b >> 6], best);
                atomicMax(&trn[b >> 6], run);
            }
          }
        }
        if (prof) { d1 = __builtin_amdgcn_s_memtime(); c_p2 += d1 - d0; d0 = d1; }
        __syncthreads();
        if (prof) { d1 = __builtin_amdgcn_s_memtime(); c_b2 += d1 - d0; d0 = d1; }
        if (tid == 0) {
            sh.nwork = 0;
            sh.nvis = 0;
            sh.vnext = 0;
        }
        if constexpr (L3) {
            // the touched super-tiles' maxima from their tiles' (final behind the barrier above)
            for (unsigned long long touched = touched0; touched; touched &= touched - 1) {
                const int u = (__builtin_ctzll(touched) >> 2) * 16 + wave;
                const int t = u * 16 + (lane & 15);
                int vm = lane < 16 && t < ntile ? tmx[t] : (int)0x80000000;
                int vr = lane < 16 && t < ntile ? trn[t] : (int)0x80000000;
                tpu3_wave_max_i32_fast_x2(vm, vr);
                if (lane == 0) {
                    gmx[u] = vm;
                    grn[u] = vr;
                }
            }
            __syncthreads();
        }
    };

    int J = 1, r = 1;
    int thr_keep = (int)0x80000000;                 // the candidate threshold carried from round to round
    for (int round = 0;; ++round) {
        const int par = round & 1;
        if (prof) c0 = __builtin_amdgcn_s_memtime();
        apply(J, par ^ 1);
        if (prof) { c1 = __builtin_amdgcn_s_memtime(); c_apply += c1 - c0; c0 = c1; }
        // ---- select the next samples ---------------------------------------------------------------------
        uint32_t *cl = sh.cand[par];
        // (every wave reduces the 256 tile records itself: they are final behind apply()'s last barrier)
        int gbest, rstar;
        {
            int vm = (int)0x80000000, vr = (int)0x80000000;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                vm = max(vm, gmx[i * 64 + lane]);
                vr = max(vr, grn[i * 64 + lane]);
            }
            gbest = tpu3_wave_max_i32_fast(vm);
            rstar = tpu3_wave_max_i32_fast(vr);
        }
        const int tmax = tvalid ? gmx[tq] : (int)0x80000000;
        // the buckets of this wave's units in `ul` (a ballot over the quads' first lanes): body(bucket of this lane,
        // its maximum, its best point's position) per tile.  L3: only the tiles whose maximum passes `pred`, listed
        // first (a slice of the idle work list per wave) so that four tiles' loads are in flight together -- a
        // dependent round trip per tile was 44 k cycles of a 118 k round.
        auto for_tiles = [&](unsigned long long ul, auto pred, auto body) __attribute__((always_inline)) {
            if constexpr (!L3) {
                for (; ul; ul &= ul - 1) {
                    const int b = (((int)__builtin_ctzll(ul) >> 2) * 16 + wave) * 64 + lane;
                    body(b, bmax[b], (int)barg[b]);
                }
            } else {
                uint32_t *wl = work + wave * 256;
                int cnt = 0;
                for (; ul; ul &= ul - 1) {
                    const int t = (((int)__builtin_ctzll(ul) >> 2) * 16 + wave) * 16 + (lane & 15);
                    const bool ok = lane < 16 && t < ntile && pred(tmx[t]);
                    const unsigned long long om = __ballot(ok);
                    if (ok)
                        wl[cnt + __builtin_popcountll(om & ((1ull << lane) - 1ull))] = (uint32_t)t;
                    cnt += __builtin_popcountll(om);
                }
                for (int i0 = 0; i0 < cnt; i0 += FL_CB) {
                    int bv[FL_CB], mv[FL_CB], av[FL_CB];
#pragma unroll
                    for (int k = 0; k < FL_CB; ++k) {
                        bv[k] = (int)wl[i0 + k < cnt ? i0 + k : i0] * 64 + lane;
                        mv[k] = a.bm0[bv[k]];
                        av[k] = (int)a.ba0[bv[k]];
                    }
#pragma unroll 1
                    for (int i = i0; i < cnt && i < i0 + FL_CB; ++i) {
                        body(bv[0], mv[0], av[0]);
#pragma unroll
                        for (int k = 0; k + 1 < FL_CB; ++k) {
                            bv[k] = bv[k + 1]; mv[k] = mv[k + 1]; av[k] = av[k + 1];
                        }
                    }
                }
            }
        };
        bool ties_top = gbest <= rstar;             // no bucket beats every runner-up: plain arg-max, tie rule
        // Any threshold >= R* is valid, and the buckets above a FIXED threshold only get fewer: the last round's,
        // steered to list 100 .. 300 buckets, spares most rounds the bisection passes an overfull list costs (the
        // buckets above R* alone number thousands at 3.8 M points).
        int thr = L3 ? max(rstar, thr_keep < gbest ? thr_keep : rstar) : rstar;
        int total = 0;
        bool raised = false;
        for (;;) {
            // the buckets with bmax > thr (ties_top: == gbest) of this wave's tiles
            unsigned long long tl = __ballot(tvalid && quad == 0 && (ties_top ? tmax == gbest : tmax > thr));
            if (ties_top) {
                if (tid == 0)
                    sh.minkey = 0xFFFFFFFFu;
                __syncthreads();
            }
            if (PROF) c_pass += 1;
            auto tile_pred = [&](int v) __attribute__((always_inline)) { return ties_top ? v == gbest : v > thr; };
            for_tiles(tl, tile_pred, [&](int b, int bm, int ba) __attribute__((always_inline)) {
                const bool c = ties_top ? bm == gbest : bm > thr;
                const unsigned long long cm = __ballot(c);
                if (!cm)
                    return;
                if (ties_top) {
                    if (c)
                        atomicMin(&sh.minkey, TK[(size_t)b * FL_R + ba]);
                    return;
                }
                // (no memory access here: an entry is the maximum and the bucket; wave 0 fetches the coordinates of
                // the ranked list in one round trip -- with the point loads in this loop every wave paid a dependent
                // L2 trip per candidate tile, 33 k cycles of a 61 k round)
                int base = 0;
                if (lane == 0)
                    base = atomicAdd(&sh.ncand[par], (int)__builtin_popcountll(cm));
                base = __builtin_amdgcn_readfirstlane(base);
                const int pos = base + __builtin_popcountll(cm & ((1ull << lane) - 1ull));
                if (c && pos < FL_LIST) {
                    cl[pos * 2] = (uint32_t)bm;
                    cl[pos * 2 + 1] = ((uint32_t)b << 4) | (uint32_t)ba;
                }
            });
            __syncthreads();
            if (ties_top) {
                // second sweep: the bucket holding the smallest key among the maxima enters, alone
                const uint32_t mk = sh.minkey;
                unsigned long long t2 = __ballot(tvalid && quad == 0 && tmax == gbest);
                for_tiles(t2, tile_pred, [&](int b, int bm, int ba) __attribute__((always_inline)) {
                    if (bm == gbest) {
                        const size_t w = (size_t)b * FL_R + ba;
                        if (TK[w] == mk) {
                            cl[0] = (uint32_t)gbest;
                            cl[1] = ((uint32_t)b << 4) | (uint32_t)ba;
                        }
                    }
                });
                __syncthreads();
                total = 1;
                break;
            }
            total = sh.ncand[par];
            if (total <= FL_LIST) {
                if (total >= FL_CAP || thr <= rstar || raised)
                    break;
                // too few for a full round: lower the threshold and collect again
                __syncthreads();
                if (tid == 0)
                    sh.ncand[par] = 0;
                const int gap = gbest - thr;
                thr = gap < 0x20000000 ? max(rstar, thr - 3 * gap) : rstar;
                __syncthreads();
                continue;
            }
            // more candidates than the list holds (the first rounds): raise the threshold (any threshold >= R* is valid)
            // and collect again
            __syncthreads();
            if (tid == 0) {
                sh.ncand[par] = 0;
                sh.stat[2] += 1;
            }
            raised = true;
            if (gbest - thr <= 1)
                ties_top = true;
            else
                thr += (gbest - thr) >> 1;
            __syncthreads();
        }
        if (PROF) c_tot += total;
        if (!ties_top) {
            const int gap = gbest - thr;
            thr_keep = total >= 2 * FL_CAP + FL_CAP / 2 ? thr : (gap < 0x40000000 ? thr - gap : (int)0x80000000);
        }
        if (prof) { c1 = __builtin_amdgcn_s_memtime(); c_coll += c1 - c0; c0 = c1; }
        // wave 0: the FL_CAP best of the list (a few bisection steps on the values it holds in registers -- the list
        // usually holds more: 16-point cells qualify by the dozen), their rank order, ONE round trip for their
        // coordinates and keys; then ALL waves test the pairs for clearance (2016 pairs at 64 candidates: 32 passes
        // for a lone wave, 7 k cycles; two per wave here)
        const int left = a.m - r;
        uint32_t okey = 0;
        if (wave == 0) {
            int em[FL_LIST / 64];
            uint32_t eb[FL_LIST / 64];
#pragma unroll
            for (int u = 0; u < FL_LIST / 64; ++u) {
                const int i = u * 64 + lane;
                em[u] = i < total ? (int)cl[2 * i] : (int)0x80000000;
                eb[u] = cl[2 * (i < total ? i : 0) + 1];
            }
            int thr2 = (int)0x80000000, nsel = total;
            if (total > FL_CAP) {
                int lo = rstar, hi = gbest, chi = 0;                // count(> lo) > FL_CAP >= count(> hi) = chi
                for (int it = 0; it < 24 && hi - lo > 1; ++it) {
                    const int mid = lo + ((hi - lo) >> 1);
                    int c = 0;
#pragma unroll
                    for (int u = 0; u < FL_LIST / 64; ++u)
                        c += __builtin_popcountll(__ballot(em[u] > mid));
                    if (c > FL_CAP) {
                        lo = mid;
                    } else {
                        hi = mid; chi = c;
                        if (c >= FL_CAP - FL_CAP / 16)
                            break;
                    }
                }
                thr2 = hi; nsel = chi;
                if (lane == 0) sh.stat[2] += 1;
            }
            // (nsel == 0: more than FL_CAP buckets share the top value -- duplicated points; the exact arg-max
            // with the tie rule below settles it)
            int base = 0;
#pragma unroll
            for (int u = 0; u < FL_LIST / 64; ++u) {
                const bool sel = em[u] > thr2;
                const unsigned long long smk = __ballot(sel);
                if (sel) {
                    const int pos = base + __builtin_popcountll(smk & ((1ull << lane) - 1ull));
                    sh.mrow[pos] = em[u];
                    cl[2 * FL_LIST + pos] = eb[u];
                }
                base += __builtin_popcountll(smk);
            }
            const bool live = lane < nsel;
            const int cM = live ? sh.mrow[lane] : (int)0x80000000;
            const uint32_t cB = cl[2 * FL_LIST + (live ? lane : 0)];
            // ONE round trip for the coordinates and keys of the whole list, the ranking below in its shadow
            const float4 sp4 = TP[live ? cB : 0];
            const uint32_t cK = live ? TK[cB] : 0xFFFFFFFFu;
            if (!live)
                sh.mrow[lane] = (int)0x80000000;
            int rank = 0;
            bool tie = false;
            for (int c0 = 0; c0 < nsel; c0 += 16) {
                int4 mv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    mv[u] = *(const int4 *)(sh.mrow + c0 + 4 * u);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int m4[4] = {mv[u].x, mv[u].y, mv[u].z, mv[u].w};
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        rank += m4[v] > cM ? 1 : 0;
                        tie |= m4[v] == cM && c0 + 4 * u + v != lane;
                    }
                }
            }
            if (__ballot(live && tie)) {            // equal maxima among candidates: order by the tie key
                uint32_t *kt = cl + 2 * FL_LIST + FL_CAP;
                kt[lane] = cK;
                rank = 0;
                for (int c0 = 0; c0 < nsel; c0 += 8) {
                    const int4 m0 = *(const int4 *)(sh.mrow + c0), m1 = *(const int4 *)(sh.mrow + c0 + 4);
                    const uint4 k0 = *(const uint4 *)(kt + c0), k1 = *(const uint4 *)(kt + c0 + 4);
                    const int m8[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
                    const uint32_t k8[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
#pragma unroll
                    for (int v = 0; v < 8; ++v)
                        rank += (m8[v] > cM || (m8[v] == cM && k8[v] < cK)) ? 1 : 0;
                }
                if (lane == 0) sh.stat[3] += 1;
            }
            // into rank order
            if (live) {
                *(float4 *)sh.pick[par][rank] = sp4;            // (.w = the running distance = the bucket's maximum)
                sh.pkey[par][rank] = cK;
            }
            okey = sh.pkey[par][lane < nsel ? lane : 0];
            if (lane == 0) {
                const int jm = nsel < left ? nsel : left;
                sh.npick[par] = jm;
                sh.jclear = jm;
                sh.ncand[par ^ 1] = 0;
            }
        }
        __syncthreads();
        int jmax = sh.npick[par];
        if (jmax == 0) {
            // (only with > FL_CAP equal maxima at the top) one sample by the exact arg-max with the tie rule
            if (tid == 0)
                sh.minkey = 0xFFFFFFFFu;
            __syncthreads();
            unsigned long long t2 = __ballot(tvalid && quad == 0 && tmax == gbest);
            auto top_pred = [&](int v) __attribute__((always_inline)) { return v == gbest; };
            for_tiles(t2, top_pred, [&](int b, int bm, int ba) __attribute__((always_inline)) {
                if (bm == gbest)
                    atomicMin(&sh.minkey, TK[(size_t)b * FL_R + ba]);
            });
            __syncthreads();
            const uint32_t mk = sh.minkey;
            for_tiles(t2, top_pred, [&](int b, int bm, int ba) __attribute__((always_inline)) {
                if (bm == gbest) {
                    const size_t w = (size_t)b * FL_R + ba;
                    if (TK[w] == mk) {
                        *(float4 *)sh.pick[par][0] = TP[w];
                        sh.pkey[par][0] = mk;
                        sh.npick[par] = 1;
                        sh.jclear = 1;
                    }
                }
            });
            __syncthreads();
            jmax = 1;
            if (wave == 0)
                okey = sh.pkey[par][0];
        }
        // longest prefix in which no member lies inside the update ball of an earlier member: the smallest l with
        // d(sample i, sample l) < M_l for some i < l -- all pairs (i < l), 64 per pass in l-major order, the passes
        // dealt over the waves
        {
            const int npair = jmax * (jmax - 1) / 2;
            for (int t0 = wave * 64; t0 < npair; t0 += 1024) {
                int pl = pair_l, pi = pair_i;
                if (t0) {
                    const int t = t0 + lane;
                    pl = (int)((1.f + sqrtf(1.f + 8.f * (float)t)) * 0.5f);
                    pl -= pl * (pl - 1) / 2 > t ? 1 : 0;
                    pl += (pl + 1) * pl / 2 <= t ? 1 : 0;
                    pi = t - pl * (pl - 1) / 2;
                }
                const bool ok = pl < jmax;
                const float4 L4 = *(const float4 *)sh.pick[par][ok ? pl : 0];
                const float4 I4 = *(const float4 *)sh.pick[par][ok ? pi : 0];
                const float d = tpu3_sqdist3(L4.x - I4.x, L4.y - I4.y, L4.z - I4.z);
                const unsigned long long hit = __ballot(ok && d < L4.w);
                if (hit) {
                    const int first = __builtin_amdgcn_readlane(pl, (int)__builtin_ctzll(hit));
                    if (lane == 0)
                        atomicMin(&sh.jclear, first);
                    break;
                }
            }
        }
        __syncthreads();
        J = sh.jclear;
        if (PROF) { c_nsel += jmax; c_cut += J < jmax ? 1 : 0; }
        if (wave == 0) {
            if (lane < J)
                a.idx[r + lane] = tpu3_fps_tiekey_to_index(okey, lb);
            if (lane == 0) {
                sh.stat[0] += 1;
                sh.stat[1] += (unsigned long long)J;
            }
        }
        if (prof) { c1 = __builtin_amdgcn_s_memtime(); c_rank += c1 - c0; c0 = c1; }
        r += J;
        if (r >= a.m) {
            if (J > 1)
                apply(J - 1, par);                  // every sample but the last one updates `temp`
            break;
        }
    }
    if (prof && tid == 0) {
        sh.stat[4] = c_apply; sh.stat[5] = c_coll; sh.stat[6] = c_rank; sh.stat[7] = c_vis;
        for (int i = 0; i < 8; ++i)
            a0.prof[i] = sh.stat[i];
        a0.prof[8] = c_p1; a0.prof[9] = c_b1; a0.prof[10] = c_p2; a0.prof[11] = c_b2;
        a0.prof[12] = c_pass; a0.prof[13] = c_tot; a0.prof[14] = c_nsel; a0.prof[15] = c_cut;
    }
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

constexpr int RB_MAX_N = 16 * 64 * 25;     // 25 rows per wave

struct FbPlan {
    int ppl, nw, ngpt, nb, nbpad, npad, ng, ncell;
    bool l3;              // three levels: LDS cells of 16 leaf buckets, leaf table in global memory
    bool fl;              // tile form (fl_main_kernel): a lane per 16-point bucket, points in memory
    int ntile;            // tiles of 1024 points (fl)
    size_t fl_rec, fl_bm, fl_ba, fl_tt;     // byte sizes of the tile form's record arrays
    int rb_rows;          // > 0: register-resident kernel with this many rows per wave
    bool segmented;       // one segmented sort for the batch instead of a device sort per element
    bool global64;        // large sets, several elements: one device sort on (element << 32 | key)
    int ebits;            // bits of the element index in that key
    size_t ks, ps, bs;    // byte sizes: key array, per-point float array, bucket word array
    size_t per_elem;      // bytes of one batch element's arrays (sp, skey, ib, bbox)
    size_t sort_bytes;    // 4 key/value arrays x b
    size_t sort_temp;     // rocPRIM temporary storage
    size_t mbox;          // mailboxes of the multi-workgroup tile form (fps_cluster.hip), all elements
    int cluster;          // workgroups per element of that form (0: the single-workgroup kernels)
    size_t total;
};

// tpu3_debug_fps_cluster / TPU3_FPS_CLUSTER: -1 = the default policy of fb_plan, 0 = single-workgroup kernels only,
// 2 / 4 / 8 / 16 = that many workgroups per element whenever the size allows it
int g_cluster_force = getenv("TPU3_FPS_CLUSTER") ? atoi(getenv("TPU3_FPS_CLUSTER")) : -1;

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
    p.rb_rows = 0;                  // rows per wave (16 waves)
    // TPU3_FPS_FORCE_TILE=1: measurement hook (tools/fps_level_dispatch_probe.py) -- the per-level sets (4097 .. 25 600
    // points) on the tile form / the cluster form instead of the register-resident kernels
    static const bool force_tile = getenv("TPU3_FPS_FORCE_TILE") && atoi(getenv("TPU3_FPS_FORCE_TILE")) != 0;
    if (n <= RB_MAX_N && !(force_tile && n > 4096)) {
        const int rows = ((n + 63) / 64 + 15) / 16;
        for (int r : {4, 7, 10, 13, 16, 20, 25})
            if (r >= rows) {
                p.rb_rows = r;
                break;
            }
    }
    // 64-point buckets throughout.  Up to FB_NB_MAX of them the bucket table itself sits in LDS (two levels);
    // beyond, LDS holds cells of 16 leaf buckets and the leaf table stays in global memory (three levels).
    p.ppl = 1;
    const int bsz = 64;
    p.nb = (n + bsz - 1) / bsz;
    p.l3 = p.nb > FB_NB_MAX;
    if (p.l3 && p.nb > FB_NB_MAX * FB_GS)
        return false;
    // (bucket-table geometry of the register-resident kernels' setup: 64-point buckets, groups of 16)
    p.nw = 8;
    {
        const int unit = FB_GS * p.nw * (p.l3 ? FB_GS : 1);
        p.nbpad = (p.nb + unit - 1) / unit * unit;
        p.ncell = p.l3 ? p.nbpad / FB_GS : p.nbpad;
    }
    p.ng = p.ncell / FB_GS;
    p.npad = p.nb * bsz;
    // beyond the register-resident limit: the tile form (two levels up to 262 144 points, three up to 4 194 304)
    p.ntile = (n + FL_TP - 1) / FL_TP;
    p.fl = !p.rb_rows;
    if (p.fl && p.ntile > FL_TMAX)
        return false;
    p.fl_rec = p.fl_bm = p.fl_ba = p.fl_tt = 0;
    if (p.fl) {
        p.npad = p.ntile * FL_TP;
        p.fl_rec = align256((size_t)p.ntile * 64 * sizeof(uint4));
        p.fl_bm = align256((size_t)p.ntile * 64 * sizeof(int32_t));
        p.fl_ba = align256((size_t)p.ntile * 64);
        p.fl_tt = align256((size_t)p.ntile * 8 * sizeof(float));
    }
    p.ngpt = (p.ng + p.nw * 64 - 1) / (p.nw * 64);          // 1 for ng <= 256
    p.ks = align256(sizeof(uint32_t) * (size_t)n);
    p.ps = align256(sizeof(float) * (size_t)p.npad);
    p.bs = align256(sizeof(uint32_t) * (size_t)p.nbpad);
    p.per_elem = 5 * p.ps + 9 * p.bs + align256(8 * sizeof(float)) + p.fl_rec + p.fl_bm + p.fl_ba + p.fl_tt;
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
    // Several workgroups per element (fps_cluster.hip) when the launch is small.  All b * G workgroups of such a launch
    // must be resident, and four of them on four streams must still fit the 256 compute units together: b * G <= 64.
    // Up to 4 elements: 16 members each (239 616 -> 80 000 alone: 33.4 ms on one workgroup, 15.5 on 8, 14.4 on 16, 15.0 on 32);
    // up to 8: 8 members; beyond that the single-workgroup kernels -- a 32-cloud launch is hidden under the network
    // stages of the next batch either way, and spends fewer compute-unit-milliseconds on one unit per cloud (measured:
    // 9.09 vs 9.05 M points/s).  Beyond 256 tiles the two-level form NEEDS 16 members (config C5's 3744 tiles).
    // TPU3_FPS_CLUSTER = 0 (off) / 2, 4, 8, 16 (forced): tuning hook, also tpu3_debug_fps_cluster().
    p.cluster = 0;
    if (p.fl) {
        int gmin = 1;
        while (gmin * 256 < p.ntile)
            gmin *= 2;
        int gg = g_cluster_force >= 0 ? g_cluster_force : (b <= 4 ? 16 : (b <= 8 ? 8 : 1));
        if (g_cluster_force < 0 && b <= 2 && p.ntile >= 2048)
            gg = 32;                            // (config C5's 3744 tiles: 287 ms on 16 members, 248 on 32, 250 on 64)
        if (g_cluster_force < 0)
            while (gg > 1 && p.ntile < 4 * gg)
                gg /= 2;                        // fewer than four tiles per member: not worth an exchange per round
        if (gg > 1 && gg < gmin && (long)b * gmin <= 64)
            gg = gmin;
        // residency (advisor, r4): all b * G workgroups of a cluster launch take a whole compute unit each and spin on
        // each other -- forced values obey the bound too, and a device (partition, CU mask) with fewer than 64 units
        // lowers it; whatever does not fit runs on the single-workgroup kernel
        static const int ncu = []() {
            int d = 0, v = 256;
            (void)hipGetDevice(&d);
            (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d);
            return v;
        }();
        const long resident = ncu < 64 ? ncu : 64;
        if (gg >= 2 && gg >= gmin && gg <= 64 && (gg & (gg - 1)) == 0 && (long)b * gg <= resident &&
            tpu3_fps_cluster_lds_bytes(p.ntile, gg) <= 160 * 1024)
            p.cluster = gg;
    }
    p.mbox = p.fl ? align256((size_t)b * tpu3_fps_cluster_mailbox_bytes(64)) : 0;
    p.total = p.sort_bytes + (size_t)b * p.per_elem + p.sort_temp + p.mbox;
    return true;
}

// measurement hook (bench.py): events recorded on the launch stream immediately around the next
// fm_main_kernel launch, see tpu3_debug_fps_bucket_events
hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
// development probe: two device words that the register-resident multi-sample kernel of the NEXT call fills with
// (rounds, samples) of its first set, see tpu3_debug_fps_level_stats
unsigned long long *g_level_stats = nullptr;
// development probe: four device words the NEXT tile-form launch fills with (rounds, samples, overflow rounds, tie
// rounds) of its first set, see tpu3_debug_fps_tile_stats
unsigned long long *g_tile_stats = nullptr;

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
    a0.n = n; a0.m = m; a0.nb = p.nb; a0.nbpad = p.nbpad; a0.npad = p.npad; a0.ng = p.ng; a0.ncell = p.ncell;
    a0.bsz = 64 * p.ppl; a0.lb = tpu3_fps_log2_bs(n);
    a0.n_arr = n_arr; a0.m_arr = m_arr;
    a0.xyz = xyz; a0.temp = temp; a0.idx = idx; a0.prof = prof;
    a0.sp = (float4 *)slabs;
    a0.skey = (uint32_t *)(slabs + 4 * p.ps);
    a0.ib = (uint32_t *)(slabs + 5 * p.ps);
    a0.bbox = (float *)(slabs + 5 * p.ps + 9 * p.bs);
    a0.per_elem = p.per_elem;
    a0.sort_stride = p.ks / 4;
    a0.fl = p.fl ? 1 : 0;
    a0.ntile = p.ntile;
    {
        char *f = slabs + 5 * p.ps + 9 * p.bs + align256(8 * sizeof(float));
        a0.rec = (uint4 *)f;
        a0.bm0 = (int32_t *)(f + p.fl_rec);
        a0.ba0 = (uint8_t *)(f + p.fl_rec + p.fl_bm);
        a0.tt = (float *)(f + p.fl_rec + p.fl_bm + p.fl_ba);
    }

    // (r6) sets of the register-resident per-level kernels: bounding box, order and permutation in one launch
    static const bool bin_on = !(getenv("TPU3_FPS_BIN") && atoi(getenv("TPU3_FPS_BIN")) == 0);
    const bool binned = bin_on && p.rb_rows && n <= 1024 * FB_BIN_PPT;
    if (binned)
        hipLaunchKernelGGL(fb_bin_kernel, dim3(b), dim3(1024), 0, s, a0);
    size_t tb = p.sort_temp;
    if (!binned) {
    hipLaunchKernelGGL(fb_bbox_kernel, dim3(b), dim3(1024), 0, s, a0);
    hipLaunchKernelGGL(fb_morton_kernel, dim3((unsigned)((a0.sort_stride + 255) / 256), b), dim3(256), 0, s, a0, k_in,
                       v_in, k64_in);
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
    }
    if (p.rb_rows && n > 1024 * 4 && m >= 256) {
        // a lane per bucket of R Morton-consecutive points (rl_main_kernel)
        a0.prof = g_level_stats;
        g_level_stats = nullptr;
        hipError_t e = hipSuccess;
#define RL_LAUNCH(RR, PP, NWV)                                                                            \
    {                                                                                                     \
        const size_t lds = rl_lds_bytes(RR, rl_zl<RR>(), NWV);                                           \
        e = hipFuncSetAttribute((const void *)rl_main_kernel<RR, PP, NWV>,                                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
        if (e != hipSuccess) return (int)e;                                                               \
        hipLaunchKernelGGL((rl_main_kernel<RR, PP, NWV>), dim3(b), dim3(NWV * 64), lds, s, a0);           \
    }
        // waves per set (see rl_main_kernel).  TPU3_RL_NW=16: tuning hook, every set on 16 waves
        static const int rl_nw = getenv("TPU3_RL_NW") ? atoi(getenv("TPU3_RL_NW")) : 0;
        const int rr = (n + 1023) / 1024;
        // (measured, 1536 sets: 6240 points 2.91 ms on 16 waves with capped registers vs 3.01 ms on 4 waves;
        // 12 480 points 6.81 ms on 8 waves vs 7.66 ms on 16; 24 960 points need all 1024 lanes)
        if (rl_nw == 4 && !a0.prof && n <= 25 * 256) RL_LAUNCH(25, false, 4)
        else if (rr <= 7) RL_LAUNCH(7, false, 16)
        else if (rl_nw != 16 && !a0.prof && n <= 25 * 512) RL_LAUNCH(25, false, 8)
        else if (rr <= 13) RL_LAUNCH(13, false, 16)
        else if (rr <= 19) RL_LAUNCH(19, false, 16)
        else if (a0.prof) RL_LAUNCH(25, true, 16)
        else RL_LAUNCH(25, false, 16)
#undef RL_LAUNCH
        return tpu3_launch_status();
    }
    if (p.rb_rows > 7)
        return TPU3_ELIMIT;     // (up to 25 600 points with fewer than 256 samples: fps.hip keeps those on its
                                // plain register-resident kernel and never asks for a bucket plan)
    if (p.rb_rows) {
        // small sets, one sample per round: rows (= 64-point buckets) in VGPRs, no write-back pass
        hipLaunchKernelGGL(fb_bucket_init_kernel<1>, dim3((p.nbpad + 3) / 4, b), dim3(256), 0, s, a0);
        const size_t lds1 = rb_lds_bytes(p.rb_rows);
        hipError_t e = hipSuccess;
        if (p.rb_rows == 4) {
            e = hipFuncSetAttribute((const void *)rb_main_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(rb_main_kernel<4>, dim3(b), dim3(1024), lds1, s, a0);
        } else {
            e = hipFuncSetAttribute((const void *)rb_main_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(rb_main_kernel<7>, dim3(b), dim3(1024), lds1, s, a0);
        }
        return tpu3_launch_status();
    }
    if (a0.fl) {
        // a lane per 16-point bucket, 1024-point tiles (fl_main_kernel)
        a0.prof = g_tile_stats;
        g_tile_stats = nullptr;
        hipLaunchKernelGGL(fl_init_kernel, dim3(p.ntile, b), dim3(64), 0, s, a0);
        if (p.cluster) {
            // several workgroups per element (fps_cluster.hip)
            const hipEvent_t c0 = g_ev_start, c1 = g_ev_stop;
            g_ev_start = g_ev_stop = nullptr;
            unsigned long long *st = a0.prof;
            a0.prof = nullptr;
            if (c0) (void)hipEventRecord(c0, s);
            const int rc = tpu3_fps_cluster_launch(s, b, p.cluster, &a0, sort_tmp + p.sort_temp, st);
            if (c1) (void)hipEventRecord(c1, s);
            if (rc)
                return rc;
            hipLaunchKernelGGL(fb_writeback_kernel, dim3((n + 255) / 256, b), dim3(256), 0, s, a0);
            return tpu3_launch_status();
        }
        const size_t lds = fl_lds_bytes(p.ntile, p.l3);
        void (*kern)(FbArgs) = p.l3 ? (a0.prof ? fl_main_kernel<true, true> : fl_main_kernel<false, true>)
                                    : (a0.prof ? fl_main_kernel<true, false> : fl_main_kernel<false, false>);
        const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return (int)e;
        const hipEvent_t e0 = g_ev_start, e1 = g_ev_stop;
        g_ev_start = g_ev_stop = nullptr;
        if (e0) (void)hipEventRecord(e0, s);
        hipLaunchKernelGGL(kern, dim3(b), dim3(1024), lds, s, a0);
        if (e1) (void)hipEventRecord(e1, s);
        const int r = tpu3_launch_status();
        if (r)
            return r;
        hipLaunchKernelGGL(fb_writeback_kernel, dim3((n + 255) / 256, b), dim3(256), 0, s, a0);
        return tpu3_launch_status();
    }
    return TPU3_ELIMIT;         // (unreachable: every plan above returns)
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
// (hipEvent_t, created by the caller) on its stream right before / after fm_main_kernel, so that a
// caller can time exactly that kernel on whatever stream it runs.  One-shot; host-side state only.
extern "C" int tpu3_debug_fps_bucket_events(void *start, void *stop)
{
    g_ev_start = (hipEvent_t)start;
    g_ev_stop = (hipEvent_t)stop;
    return TPU3_OK;
}

extern "C" int tpu3_debug_fps_level_stats(unsigned long long *stats)
{
    g_level_stats = stats;
    return TPU3_OK;
}

// Tuning / test hook (not part of include/tpu3.h): workgroups per element of the tile-form FPS for the following calls
// (-1: default policy, 0: single-workgroup kernels only, 2 / 4 / 8 / 16).  Returns the previous setting.
extern "C" int tpu3_debug_fps_cluster(int g)
{
    const int old = g_cluster_force;
    g_cluster_force = g;
    return old;
}

// Which kernel family a call of this shape takes (the dispatch table of DESIGN section 4 as code; tests pin it):
// 0 plain register-resident / streaming (fps.hip), 1 rb_main (rows in registers, one sample per round), 2 rl_main (lane
// per bucket), 4 fl_main two levels, 5 fl_main three levels, 6 the cluster form (fps_cluster.hip); *cluster = its G.
extern "C" int tpu3_debug_fps_plan(int b, int n, int m, int *cluster)
{
    FbPlan p;
    if (cluster) *cluster = 0;
    // (the front of tpu3_fps_ragged_f32, csrc/fps.hip: up to 25 600 points the bucketed kernels are taken from 4096
    // points and 256 samples on; TPU3_FPS_BUCKET_MIN_N moves the first threshold)
    static const int min_n = getenv("TPU3_FPS_BUCKET_MIN_N") ? atoi(getenv("TPU3_FPS_BUCKET_MIN_N")) : 4096;
    if (n <= RB_MAX_N && !(n >= min_n && m >= 256))
        return 0;
    if (!fb_plan(b, n, p))
        return -1;
    if (p.rb_rows && n > 1024 * 4)
        return 2;
    if (p.rb_rows)
        return p.rb_rows <= 7 ? 1 : -1;
    if (cluster) *cluster = p.cluster;
    return p.cluster ? 6 : (p.l3 ? 5 : 4);
}

extern "C" int tpu3_debug_fps_tile_stats(unsigned long long *stats)
{
    g_tile_stats = stats;
    return TPU3_OK;
}
