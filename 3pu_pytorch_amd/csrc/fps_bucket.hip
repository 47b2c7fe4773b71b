// fps_bucket.hip -- exact, work-skipping farthest-point sampling for large point sets (gfx950).
//
// The reference's FPS (sampling/sampling_cuda.cu:103-174) re-reads all n points every round:
// 239 616 points x 80 000 rounds for the final resample of a 5000 -> 80 000 cloud.  A round only
// CHANGES the running distance of points closer to the new sample than to every earlier one, and
// those are confined to a shrinking ball around it.  This kernel produces bit-identical indices
// (and identical final `temp`) while touching only that ball:
//
//   setup   points are sorted along a 30-bit Morton curve (rocPRIM radix sort) and cut into
//           buckets of 64*PPL consecutive points; a bucket knows its tight AABB, and the table in
//           LDS holds for every bucket its current (max distance, tie key, xyz of that point).
//   round   every lane tests the buckets it owns:  dbox(sample, AABB) >= bucket max  ==> no
//           point of the bucket can change (dbox is computed with the same fp32 association as the
//           point distance, and fp32 rounding is monotone, so dbox <= d(p) for every p inside:
//           min(d(p), temp[p]) == temp[p] exactly) -- skip it.  Touched buckets are re-scanned by
//           the wave that owns them (64 lanes = 64 points, one coalesced read of x,y,z,temp,key).
//           Then the arg-max over the bucket table (DPP wave reduction + one LDS hand-off, ONE
//           s_barrier per round) picks the next sample with the reference's tie rule.
//
// One workgroup (1024 lanes) per batch element; bucket b is owned by wave b%16, lane (b/16)%64,
// so spatially adjacent (Morton-consecutive) buckets are re-scanned by different waves.  Every
// table entry is written and read by the same wave, hence no second barrier.
#include "tpu3_dev.h"

#include <cstring>
#include <rocprim/rocprim.hpp>
#include <vector>

namespace {

constexpr int FB_W = 1024;
constexpr int FB_NW = FB_W / 64;

struct FbArgs {
    int n, m, nb, npad;
    const float *xyz;     // (n,3) original order
    float *temp;          // (n)
    int32_t *idx;         // (m)
    float4 *sp;           // (npad) Morton order: x, y, z, running distance
    uint32_t *skey;       // (npad) tie key of the original index (0xFFFFFFFF = padding)
    unsigned long long *prof;   // PROF builds only: 16 waves x 8 cycle counters
};

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
__global__ __launch_bounds__(1024) void fb_bbox_kernel(int n, const float *__restrict__ xyz, float *__restrict__ bbox)
{
    __shared__ float red[6][16];
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

__global__ __launch_bounds__(256) void fb_morton_kernel(int n, const float *__restrict__ xyz,
                                                        const float *__restrict__ bbox,
                                                        uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    uint32_t code = 0;
    for (int a = 0; a < 3; ++a) {
        const float lo = bbox[a], ext = bbox[3 + a] - lo;
        float q = ext > 0.f ? (xyz[(size_t)i * 3 + a] - lo) / ext * 1023.0f : 0.f;
        q = fminf(fmaxf(q, 0.f), 1023.f);
        code |= spread10((uint32_t)q) << a;
    }
    keys[i] = code;
    vals[i] = (uint32_t)i;
}

// Morton-ordered float4 (x,y,z,temp) + tie keys; slots past n repeat the last live point with temp = -1
__global__ __launch_bounds__(256) void fb_permute_kernel(FbArgs a, const uint32_t *__restrict__ order, int lb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.npad)
        return;
    const bool live = i < a.n;
    const uint32_t o = order[live ? i : a.n - 1];
    a.sp[i] = make_float4(a.xyz[(size_t)o * 3 + 0], a.xyz[(size_t)o * 3 + 1], a.xyz[(size_t)o * 3 + 2],
                          live ? a.temp[o] : -1.0f);
    a.skey[i] = live ? tpu3_fps_tiekey((int)o, lb) : 0xFFFFFFFFu;
}

__global__ __launch_bounds__(256) void fb_writeback_kernel(FbArgs a, int lb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n)
        a.temp[tpu3_fps_tiekey_to_index(a.skey[i], lb)] = a.sp[i].w;
}

struct FbSlots {
    int d[2][FB_NW];
    uint32_t key[2][FB_NW];
    float x[2][FB_NW], y[2][FB_NW], z[2][FB_NW];
};

// arguments of batch element i from those of element 0: user arrays are dense (b, n, ...) slabs,
// workspace arrays repeat every `per_elem` bytes
__host__ __device__ inline FbArgs fb_elem(const FbArgs &a0, size_t per_elem, int i)
{
    FbArgs a = a0;
    a.xyz = a0.xyz + (size_t)i * a0.n * 3;
    a.temp = a0.temp + (size_t)i * a0.n;
    a.idx = a0.idx + (size_t)i * a0.m;
    a.sp = (float4 *)((char *)a0.sp + (size_t)i * per_elem);
    a.skey = (uint32_t *)((char *)a0.skey + (size_t)i * per_elem);
    return a;
}

template <int NBPT, int PPL, bool PROF = false>
__global__ __launch_bounds__(FB_W) void fb_main_kernel(FbArgs a0, size_t per_elem, int lb)
{
    constexpr int BS = 64 * PPL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FbArgs a = fb_elem(a0, per_elem, blockIdx.x);
    const int nb = a.nb;
    constexpr int TB = NBPT * FB_W;                 // table entries (owner order, see publish)
    int *t_max = (int *)smem;
    uint32_t *t_key = (uint32_t *)(t_max + TB);
    float *t_x = (float *)(t_key + TB);
    float *t_y = t_x + TB;
    float *t_z = t_y + TB;
    FbSlots &sl = *(FbSlots *)(t_z + TB);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float4 *__restrict__ sp = a.sp;
    const uint32_t *__restrict__ skey = a.skey;

    // lane-local best of one bucket after folding sample q into its points
    struct Cand { float t, x, y, z; uint32_t key; };
    auto fold = [&](int beta, float qx, float qy, float qz, bool update) {
        Cand c{-2.0f, 0.f, 0.f, 0.f, 0xFFFFFFFFu};
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
            const int i = beta * BS + p * 64 + lane;
            const float4 v = sp[i];
            const uint32_t key = skey[i];
            float t = v.w;
            if (update) {
                const float d = tpu3_sqdist3(v.x - qx, v.y - qy, v.z - qz);
                const float d2 = fminf(d, t);
                if (d2 != t)
                    ((float *)(sp + i))[3] = d2;
                t = d2;
            }
            if (t > c.t || (t == c.t && key < c.key)) {
                c.t = t; c.key = key; c.x = v.x; c.y = v.y; c.z = v.z;
            }
        }
        return c;
    };
    // The table is stored in OWNER order: the entry of bucket (j, l, wave) sits at j*1024 + wave*64 + l,
    // i.e. at j*1024 + tid of its owner, so the per-round table read of a wave is 64 consecutive
    // words (bucket-id order would put the 64 lanes 16 words apart: a 16-way LDS bank conflict
    // on every read of every wave, measured at ~4000 LDS cycles per round).
    auto publish = [&](int e, const Cand &c, int wmax, int win_lane) {
        if (lane == win_lane) {
            t_max[e] = wmax; t_key[e] = c.key;
            t_x[e] = c.x; t_y[e] = c.y; t_z[e] = c.z;
        }
    };

    // ---- setup: every wave scans the buckets its lanes own; the owner lane keeps the AABB --------
    float blo[NBPT][3], bhi[NBPT][3];
#pragma unroll
    for (int j = 0; j < NBPT; ++j) {
        for (int c = 0; c < 3; ++c) {           // an unowned slot is infinitely far away
            blo[j][c] = __builtin_inff();
            bhi[j][c] = __builtin_inff();
        }
        for (int l = 0; l < 64; ++l) {
            const int beta = j * FB_W + l * FB_NW + wave;
            if (beta >= nb)
                break;
            const Cand c = fold(beta, 0.f, 0.f, 0.f, false);
            int wl;
            const int wmax = tpu3_wave_argmax(__float_as_int(c.t), c.key, wl);
            publish(j * FB_W + wave * 64 + l, c, wmax, wl);
            float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
            float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
            for (int p = 0; p < PPL; ++p) {
                const float4 v = sp[beta * BS + p * 64 + lane];
                lo[0] = fminf(lo[0], v.x); hi[0] = fmaxf(hi[0], v.x);
                lo[1] = fminf(lo[1], v.y); hi[1] = fmaxf(hi[1], v.y);
                lo[2] = fminf(lo[2], v.z); hi[2] = fmaxf(hi[2], v.z);
            }
            for (int c3 = 0; c3 < 3; ++c3) {
                const float l3 = -tpu3_wave_max_f32(-lo[c3]), h3 = tpu3_wave_max_f32(hi[c3]);
                if (lane == l) {
                    blo[j][c3] = l3;
                    bhi[j][c3] = h3;
                }
            }
        }
    }

    int old = 0;
    if (tid == 0)
        a.idx[0] = 0;
    float qx = a.xyz[0], qy = a.xyz[1], qz = a.xyz[2];
    int cmax[NBPT];                 // this lane's buckets: current max bits (for the prune test)
#pragma unroll
    for (int j = 0; j < NBPT; ++j) {
        const int beta = j * FB_W + lane * FB_NW + wave;
        cmax[j] = beta < nb ? t_max[j * FB_W + tid] : (int)0x80000000;
    }

    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 1; r < a.m; ++r) {
        unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0, tk4 = 0;
        if (PROF) tk0 = __builtin_amdgcn_s_memtime();
        // ---- prune test: which of this wave's buckets can the new sample change? ------------------
        unsigned long long touched[NBPT];
#pragma unroll
        for (int j = 0; j < NBPT; ++j) {
            const float dx = fmaxf(fmaxf(blo[j][0] - qx, qx - bhi[j][0]), 0.f);
            const float dy = fmaxf(fmaxf(blo[j][1] - qy, qy - bhi[j][1]), 0.f);
            const float dz = fmaxf(fmaxf(blo[j][2] - qz, qz - bhi[j][2]), 0.f);
            touched[j] = __ballot(tpu3_sqdist3(dx, dy, dz) < __int_as_float(cmax[j]));
        }
        if (PROF) tk1 = __builtin_amdgcn_s_memtime();
        // ---- re-scan them two at a time: both buckets' loads are in flight together and the two
        //      reduction chains interleave; an odd one out is simply scanned twice (idempotent) -----
#pragma unroll
        for (int j = 0; j < NBPT; ++j) {
            unsigned long long mask = touched[j];
            while (mask) {
                const int l0 = __builtin_ctzll(mask);
                mask &= mask - 1;
                int l1 = l0;
                if (mask) {
                    l1 = __builtin_ctzll(mask);
                    mask &= mask - 1;
                }
                const int b0 = j * FB_W + l0 * FB_NW + wave, b1 = j * FB_W + l1 * FB_NW + wave;
                const Cand c0 = fold(b0, qx, qy, qz, true);
                const Cand c1 = fold(b1, qx, qy, qz, true);
                int m0 = __float_as_int(c0.t), m1 = __float_as_int(c1.t);
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
                publish(j * FB_W + wave * 64 + l0, c0, m0, __builtin_ctzll(t0));
                publish(j * FB_W + wave * 64 + l1, c1, m1, __builtin_ctzll(t1));
                if (PROF) pc[5] += 1 + (b1 != b0);
                if (PROF) pc[6] += 1;
            }
        }
        if (PROF) tk2 = __builtin_amdgcn_s_memtime();
        // ---- arg-max over the bucket table --------------------------------------------------------
        int best = (int)0x80000000, bj = 0;
        uint32_t bkey = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < NBPT; ++j) {
            const int beta = j * FB_W + lane * FB_NW + wave;
            if (beta < nb) {
                const int v = t_max[j * FB_W + tid];
                const uint32_t k = t_key[j * FB_W + tid];
                cmax[j] = v;
                if (v > best || (v == best && k < bkey)) {
                    best = v; bkey = k; bj = j;
                }
            }
        }
        const int par = r & 1;
        int wl;
        const int wmax = tpu3_wave_argmax(best, bkey, wl);
        if (lane == wl) {
            const int beta = bj * FB_W + lane * FB_NW + wave, e = bj * FB_W + tid;
            sl.d[par][wave] = wmax;
            sl.key[par][wave] = bkey;
            const bool ok = beta < nb;
            sl.x[par][wave] = ok ? t_x[e] : 0.f;
            sl.y[par][wave] = ok ? t_y[e] : 0.f;
            sl.z[par][wave] = ok ? t_z[e] : 0.f;
        }
        if (PROF) tk3 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (PROF) tk4 = __builtin_amdgcn_s_memtime();
        const int sd = lane < FB_NW ? sl.d[par][lane] : (int)0x80000000;
        const uint32_t sk = lane < FB_NW ? sl.key[par][lane] : 0xFFFFFFFFu;
        const int gmax = __builtin_amdgcn_readlane(tpu3_row_max_i32_fast(sd), 0);
        unsigned long long who = __ballot(lane < FB_NW && sd == gmax);
        if (__builtin_popcountll(who) != 1) {
            const uint32_t rk = tpu3_row_min_u32(lane < FB_NW && sd == gmax ? sk : 0xFFFFFFFFu);
            const uint32_t win = (uint32_t)__builtin_amdgcn_readlane((int)rk, 0);
            who = __ballot(lane < FB_NW && sd == gmax && sk == win);
        }
        const int ww = __builtin_ctzll(who | (1ull << 63)) & 15;
        qx = sl.x[par][ww];
        qy = sl.y[par][ww];
        qz = sl.z[par][ww];
        old = tpu3_fps_tiekey_to_index(sl.key[par][ww], lb);
        if (tid == 0)
            a.idx[r] = old;
        if (PROF) {
            const unsigned long long tk5 = __builtin_amdgcn_s_memtime();
            pc[0] += tk1 - tk0; pc[1] += tk2 - tk1; pc[2] += tk3 - tk2; pc[3] += tk4 - tk3; pc[4] += tk5 - tk4;
        }
    }
    if (PROF && lane == 0 && a.prof)
        for (int i = 0; i < 8; ++i)
            a.prof[wave * 8 + i] = pc[i];
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct FbPlan {
    int ppl, nbpt, nb, npad;
    size_t per_elem;      // bytes of one batch element's arrays
    size_t sort_temp;     // rocPRIM temporary storage
    size_t total;
};

bool fb_plan(int b, int n, FbPlan &p)
{
    // smallest bucket (64*PPL points) that keeps the table within LDS and 6 buckets per lane
    const int cap = 6 * FB_W;       // 6144 buckets: 5 words each = 120 KiB of LDS
    p.ppl = 0;
    for (int ppl : {1, 2, 4, 8, 16})
        if ((long)cap * 64 * ppl >= n) {
            p.ppl = ppl;
            break;
        }
    if (!p.ppl)
        return false;
    const int bs = 64 * p.ppl;
    p.nb = (n + bs - 1) / bs;
    p.npad = p.nb * bs;
    const int need = (p.nb + FB_W - 1) / FB_W;
    p.nbpt = need <= 1 ? 1 : (need <= 2 ? 2 : (need <= 4 ? 4 : 6));
    size_t e = 0;
    e += 4 * align256(sizeof(uint32_t) * (size_t)n);          // keys in/out, vals in/out
    e += 5 * align256(sizeof(float) * (size_t)p.npad);        // sx sy sz st skey
    e += align256(8 * sizeof(float));                         // bbox
    p.per_elem = e;
    size_t tb = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tb, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                              (uint32_t *)nullptr, (size_t)n, 0, 30, (hipStream_t)0);
    p.sort_temp = align256(tb);
    p.total = (size_t)b * p.per_elem + p.sort_temp;
    return true;
}

template <int NBPT, int PPL>
int fb_launch_main(hipStream_t s, int b, const FbArgs &a0, size_t per_elem, int nb, int lb)
{
    const size_t lds = (size_t)NBPT * FB_W * 20 + sizeof(FbSlots) + 16;
    auto kern = fb_main_kernel<NBPT, PPL>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess)
        return (int)e;
    hipLaunchKernelGGL(kern, dim3(b), dim3(FB_W), lds, s, a0, per_elem, lb);
    return tpu3_launch_status();
}

template <int PPL>
int fb_dispatch_nbpt(hipStream_t s, int b, const FbArgs &a0, const FbPlan &p, int lb)
{
    switch (p.nbpt) {
    case 1: return fb_launch_main<1, PPL>(s, b, a0, p.per_elem, p.nb, lb);
    case 2: return fb_launch_main<2, PPL>(s, b, a0, p.per_elem, p.nb, lb);
    case 4: return fb_launch_main<4, PPL>(s, b, a0, p.per_elem, p.nb, lb);
    default: return fb_launch_main<6, PPL>(s, b, a0, p.per_elem, p.nb, lb);
    }
}

} // namespace

size_t tpu3_fps_bucket_workspace_bytes(int b, int n)
{
    FbPlan p;
    return fb_plan(b, n, p) ? p.total : 0;
}

// Dense batch (no ragged sizes).  Returns TPU3_ELIMIT when n is beyond the bucket plan.
int tpu3_fps_bucket_launch(hipStream_t s, int b, int n, int m, const float *xyz, float *temp, int32_t *idx,
                           void *workspace, size_t workspace_bytes)
{
    FbPlan p;
    if (!fb_plan(b, n, p))
        return TPU3_ELIMIT;
    if (!workspace || workspace_bytes < p.total)
        return TPU3_EINVAL;
    char *base = (char *)workspace;
    char *sort_tmp = base + (size_t)b * p.per_elem;
    const int lb = tpu3_fps_log2_bs(n);
    const size_t ks = align256(sizeof(uint32_t) * (size_t)n), ps = align256(sizeof(float) * (size_t)p.npad);
    FbArgs a0;
    a0.n = n; a0.m = m; a0.nb = p.nb; a0.npad = p.npad;
    a0.xyz = xyz; a0.temp = temp; a0.idx = idx; a0.prof = nullptr;
    char *q0 = base + 4 * ks;
    a0.sp = (float4 *)q0;
    a0.skey = (uint32_t *)(q0 + 4 * ps);
    for (int i = 0; i < b; ++i) {
        char *e = base + (size_t)i * p.per_elem;
        uint32_t *k_in = (uint32_t *)e, *k_out = (uint32_t *)(e + ks);
        uint32_t *v_in = (uint32_t *)(e + 2 * ks), *v_out = (uint32_t *)(e + 3 * ks);
        float *bbox = (float *)(e + 4 * ks + 5 * ps);
        const FbArgs a = fb_elem(a0, p.per_elem, i);
        hipLaunchKernelGGL(fb_bbox_kernel, dim3(1), dim3(1024), 0, s, n, a.xyz, bbox);
        hipLaunchKernelGGL(fb_morton_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, a.xyz, bbox, k_in, v_in);
        size_t tb = p.sort_temp;
        hipError_t se = rocprim::radix_sort_pairs((void *)sort_tmp, tb, k_in, k_out, v_in, v_out, (size_t)n, 0, 30, s);
        if (se != hipSuccess)
            return (int)se;
        hipLaunchKernelGGL(fb_permute_kernel, dim3((p.npad + 255) / 256), dim3(256), 0, s, a, v_out, lb);
    }
    int r;
    switch (p.ppl) {
    case 1: r = fb_dispatch_nbpt<1>(s, b, a0, p, lb); break;
    case 2: r = fb_dispatch_nbpt<2>(s, b, a0, p, lb); break;
    case 4: r = fb_dispatch_nbpt<4>(s, b, a0, p, lb); break;
    case 8: r = fb_dispatch_nbpt<8>(s, b, a0, p, lb); break;
    default: r = fb_dispatch_nbpt<16>(s, b, a0, p, lb); break;
    }
    if (r)
        return r;
    for (int i = 0; i < b; ++i)
        hipLaunchKernelGGL(fb_writeback_kernel, dim3((n + 255) / 256), dim3(256), 0, s, fb_elem(a0, p.per_elem, i), lb);
    return tpu3_launch_status();
}

// Development probe (not part of include/tpu3.h): runs the <4,1> kernel with per-phase cycle
// counters; prof = 16 x 8 u64: [prune, rescan, argmax, barrier wait, broadcast, buckets, pairs, -].
extern "C" int tpu3_debug_fps_bucket_profile(void *stream, int n, int m, const float *xyz, float *temp,
                                             int32_t *idx, void *workspace, size_t workspace_bytes,
                                             unsigned long long *prof)
{
    hipStream_t s = (hipStream_t)stream;
    FbPlan p;
    if (!fb_plan(1, n, p) || p.ppl != 1 || p.nbpt != 4 || workspace_bytes < p.total)
        return TPU3_EINVAL;
    char *base = (char *)workspace;
    char *sort_tmp = base + p.per_elem;
    const int lb = tpu3_fps_log2_bs(n);
    const size_t ks = align256(sizeof(uint32_t) * (size_t)n), ps = align256(sizeof(float) * (size_t)p.npad);
    FbArgs a0;
    a0.n = n; a0.m = m; a0.nb = p.nb; a0.npad = p.npad;
    a0.xyz = xyz; a0.temp = temp; a0.idx = idx; a0.prof = prof;
    char *q0 = base + 4 * ks;
    a0.sp = (float4 *)q0;
    a0.skey = (uint32_t *)(q0 + 4 * ps);
    uint32_t *k_in = (uint32_t *)base, *k_out = (uint32_t *)(base + ks);
    uint32_t *v_in = (uint32_t *)(base + 2 * ks), *v_out = (uint32_t *)(base + 3 * ks);
    float *bbox = (float *)(base + 4 * ks + 5 * ps);
    hipLaunchKernelGGL(fb_bbox_kernel, dim3(1), dim3(1024), 0, s, n, xyz, bbox);
    hipLaunchKernelGGL(fb_morton_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, xyz, bbox, k_in, v_in);
    size_t tb = p.sort_temp;
    hipError_t se = rocprim::radix_sort_pairs((void *)sort_tmp, tb, k_in, k_out, v_in, v_out, (size_t)n, 0, 30, s);
    if (se != hipSuccess) return (int)se;
    hipLaunchKernelGGL(fb_permute_kernel, dim3((p.npad + 255) / 256), dim3(256), 0, s, a0, v_out, lb);
    const size_t lds = (size_t)4 * FB_W * 20 + sizeof(FbSlots) + 16;
    auto kern = fb_main_kernel<4, 1, true>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(1), dim3(FB_W), lds, s, a0, p.per_elem, lb);
    return tpu3_launch_status();
}
