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
    float *sx, *sy, *sz, *st;   // (npad) Morton order
    uint32_t *skey;       // (npad) tie key of the original index (0xFFFFFFFF = padding)
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

// Morton-ordered structure-of-arrays; slots past n repeat the last live point with temp = -1
__global__ __launch_bounds__(256) void fb_permute_kernel(FbArgs a, const uint32_t *__restrict__ order, int lb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.npad)
        return;
    const bool live = i < a.n;
    const uint32_t o = order[live ? i : a.n - 1];
    a.sx[i] = a.xyz[(size_t)o * 3 + 0];
    a.sy[i] = a.xyz[(size_t)o * 3 + 1];
    a.sz[i] = a.xyz[(size_t)o * 3 + 2];
    a.st[i] = live ? a.temp[o] : -1.0f;
    a.skey[i] = live ? tpu3_fps_tiekey((int)o, lb) : 0xFFFFFFFFu;
}

__global__ __launch_bounds__(256) void fb_writeback_kernel(FbArgs a, int lb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n)
        a.temp[tpu3_fps_tiekey_to_index(a.skey[i], lb)] = a.st[i];
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
    a.sx = (float *)((char *)a0.sx + (size_t)i * per_elem);
    a.sy = (float *)((char *)a0.sy + (size_t)i * per_elem);
    a.sz = (float *)((char *)a0.sz + (size_t)i * per_elem);
    a.st = (float *)((char *)a0.st + (size_t)i * per_elem);
    a.skey = (uint32_t *)((char *)a0.skey + (size_t)i * per_elem);
    return a;
}

template <int NBPT, int PPL>
__global__ __launch_bounds__(FB_W) void fb_main_kernel(FbArgs a0, size_t per_elem, int lb)
{
    constexpr int BS = 64 * PPL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const FbArgs a = fb_elem(a0, per_elem, blockIdx.x);
    const int nb = a.nb;
    int *t_max = (int *)smem;                       // bucket table
    uint32_t *t_key = (uint32_t *)(t_max + nb);
    float *t_x = (float *)(t_key + nb);
    float *t_y = t_x + nb;
    float *t_z = t_y + nb;
    FbSlots &sl = *(FbSlots *)(t_z + nb);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *__restrict__ sx = a.sx, *__restrict__ sy = a.sy, *__restrict__ sz = a.sz;
    float *__restrict__ st = a.st;
    const uint32_t *__restrict__ skey = a.skey;

    // re-scan one bucket against sample (qx,qy,qz); `first` = setup pass (no distance update,
    // also returns the bucket AABB)
    auto scan = [&](int beta, float qx, float qy, float qz, bool first, float (&lo)[3], float (&hi)[3]) {
        float best = -2.0f, bxv = 0.f, byv = 0.f, bzv = 0.f;
        uint32_t bkey = 0xFFFFFFFFu;
        float llo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
        float lhi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
            const int i = beta * BS + p * 64 + lane;
            const float x = sx[i], y = sy[i], z = sz[i];
            float t = st[i];
            const uint32_t key = skey[i];
            if (!first) {
                const float d = tpu3_sqdist3(x - qx, y - qy, z - qz);
                const float d2 = fminf(d, t);
                if (d2 != t)
                    st[i] = d2;
                t = d2;
            } else {
                llo[0] = fminf(llo[0], x); lhi[0] = fmaxf(lhi[0], x);
                llo[1] = fminf(llo[1], y); lhi[1] = fmaxf(lhi[1], y);
                llo[2] = fminf(llo[2], z); lhi[2] = fmaxf(lhi[2], z);
            }
            if (t > best || (t == best && key < bkey)) {
                best = t; bkey = key; bxv = x; byv = y; bzv = z;
            }
        }
        const int bits = __float_as_int(best);
        const int wmax = tpu3_wave_max_i32(bits);
        const uint32_t wkey = tpu3_wave_min_u32(bits == wmax ? bkey : 0xFFFFFFFFu);
        if (bits == wmax && bkey == wkey && (wkey != 0xFFFFFFFFu || lane == 0)) {
            t_max[beta] = wmax; t_key[beta] = wkey;
            t_x[beta] = bxv; t_y[beta] = byv; t_z[beta] = bzv;
        }
        if (first)
            for (int c = 0; c < 3; ++c) {
                lo[c] = -tpu3_wave_max_f32(-llo[c]);
                hi[c] = tpu3_wave_max_f32(lhi[c]);
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
            float lo[3], hi[3];
            scan(beta, 0.f, 0.f, 0.f, true, lo, hi);
            if (lane == l)
                for (int c = 0; c < 3; ++c) {
                    blo[j][c] = lo[c];
                    bhi[j][c] = hi[c];
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
        cmax[j] = beta < nb ? t_max[beta] : (int)0x80000000;
    }

    for (int r = 1; r < a.m; ++r) {
        // ---- prune test + re-scan of the touched buckets (each by its owning wave) -----------------
#pragma unroll
        for (int j = 0; j < NBPT; ++j) {
            const float dx = fmaxf(fmaxf(blo[j][0] - qx, qx - bhi[j][0]), 0.f);
            const float dy = fmaxf(fmaxf(blo[j][1] - qy, qy - bhi[j][1]), 0.f);
            const float dz = fmaxf(fmaxf(blo[j][2] - qz, qz - bhi[j][2]), 0.f);
            const float dbox = tpu3_sqdist3(dx, dy, dz);
            unsigned long long mask = __ballot(dbox < __int_as_float(cmax[j]));
            while (mask) {
                const int l = __builtin_ctzll(mask);
                mask &= mask - 1;
                float lo[3], hi[3];
                scan(j * FB_W + l * FB_NW + wave, qx, qy, qz, false, lo, hi);
            }
        }
        // ---- arg-max over the bucket table --------------------------------------------------------
        int best = (int)0x80000000, bj = 0;
        uint32_t bkey = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < NBPT; ++j) {
            const int beta = j * FB_W + lane * FB_NW + wave;
            if (beta < nb) {
                const int v = t_max[beta];
                const uint32_t k = t_key[beta];
                cmax[j] = v;
                if (v > best || (v == best && k < bkey)) {
                    best = v; bkey = k; bj = j;
                }
            }
        }
        const int par = r & 1;
        const int wmax = tpu3_wave_max_i32(best);
        const uint32_t wkey = tpu3_wave_min_u32(best == wmax ? bkey : 0xFFFFFFFFu);
        if (best == wmax && bkey == wkey && (wkey != 0xFFFFFFFFu || lane == 0)) {
            const int beta = bj * FB_W + lane * FB_NW + wave;
            sl.d[par][wave] = wmax;
            sl.key[par][wave] = wkey;
            const bool ok = beta < nb;
            sl.x[par][wave] = ok ? t_x[beta] : 0.f;
            sl.y[par][wave] = ok ? t_y[beta] : 0.f;
            sl.z[par][wave] = ok ? t_z[beta] : 0.f;
        }
        __syncthreads();
        const int sd = lane < FB_NW ? sl.d[par][lane] : (int)0x80000000;
        const uint32_t sk = lane < FB_NW ? sl.key[par][lane] : 0xFFFFFFFFu;
        const int rmax = tpu3_row_max_i32(sd);
        const uint32_t rk = tpu3_row_min_u32(sd == rmax ? sk : 0xFFFFFFFFu);
        const uint32_t win = (uint32_t)__builtin_amdgcn_readlane((int)rk, 0);
        const int gmax = __builtin_amdgcn_readlane(rmax, 0);
        const unsigned long long who = __ballot(lane < FB_NW && sd == gmax && sk == win);
        const int ww = __builtin_ctzll(who | (1ull << 63));
        qx = sl.x[par][ww & 15];
        qy = sl.y[par][ww & 15];
        qz = sl.z[par][ww & 15];
        old = tpu3_fps_tiekey_to_index(win, lb);
        if (tid == 0)
            a.idx[r] = old;
    }
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
    const size_t lds = (size_t)nb * 20 + sizeof(FbSlots) + 16;
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
    a0.xyz = xyz; a0.temp = temp; a0.idx = idx;
    char *q0 = base + 4 * ks;
    a0.sx = (float *)q0; a0.sy = (float *)(q0 + ps); a0.sz = (float *)(q0 + 2 * ps); a0.st = (float *)(q0 + 3 * ps);
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
