// knn.hip -- brute-force k-nearest-neighbour grouping for gfx950.
//
// Replaces network.operations.group_knn (reference: network/operations.py:151-216), which
// materialises the (B,M,N) distance matrix, de-duplicates on the HOST with np.unique and then
// runs torch.topk.  Here nothing but the k results per query ever reaches HBM.
//
//   D[q,p] = fmaf(-2, <q,p>, |q|^2) + |p|^2      (expanded form, like operations.py:158-161)
//   <q,p>, |.|^2 : ascending-channel fmaf chains from 0 (the oracle's order)
//   unique=True : D += max(D) * dup[p]            (operations.py:192-204)
//   result      : k smallest per query, ascending, ties to the lowest index
//
// Two kernels:
//   knn_insert_kernel  k <= 64 : one lane per query, candidates staged through LDS in
//                      coalesced tiles and consumed as wave-uniform ds_read_b128 broadcasts,
//                      the running top-k kept sorted in VGPRs (branch-free shift insertion,
//                      entered only when some lane's candidate beats its current k-th).
//   knn_sort_kernel    any k   : one workgroup per query; candidate keys
//                      (order-preserving distance bits << 32 | index) are bitonic-sorted in
//                      LDS, in chunks when n exceeds the LDS tile (the best k ride along).
#include "tpu3_dev.h"

namespace {

struct KnnArgs {
    int m, n, c, k;
    const float *query;      // (b,m,c)
    const float *points;     // (b,n,c)
    const int32_t *n_arr;    // (bp) live points per point set, or null
    const int32_t *m_arr;    // (b) live queries per query set, or null
    const int32_t *pts_of;   // (b) point set of each query set, or null (identity)
    const int32_t *grp;      // (b) unique-max group, or null (one group)
    const uint8_t *dup;      // (bp,n) or null
    const uint32_t *uws;     // [0] any-dup, [4+g] mono(max D) of group g
    void *idx;               // (b,m,k) i32 / i64
    int idx64;
    float *dist;             // (b,m,k) or null
};

__device__ __forceinline__ void store_idx(const KnnArgs &a, size_t off, int v)
{
    if (a.idx64)
        ((int64_t *)a.idx)[off] = (int64_t)v;
    else
        ((int32_t *)a.idx)[off] = v;
}

// LDS row of one candidate: C == 3 -> (x,y,z,|p|^2); otherwise C channels (zero padded from
// the runtime channel count) followed by (|p|^2, addend, 0, 0).  `addend` is max(D)*dup.
template <int C>
struct Row {
    static constexpr int F4 = (C == 3) ? 1 : (C / 4 + 1);
};

template <int C>
__device__ __forceinline__ void stage_row(float4 *row, const float *__restrict__ src, int c, float add)
{
    float v[C];
#pragma unroll
    for (int i = 0; i < C; ++i)
        v[i] = i < c ? src[i] : 0.f;
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < C; ++i)
        r = __builtin_fmaf(v[i], v[i], r);
    if (C == 3) {
        // the addend cannot ride along in a 4-float row; C == 3 callers fold it in below
        row[0] = make_float4(v[0], v[1], v[2], r);
    } else {
#pragma unroll
        for (int i = 0; i < C / 4; ++i)
            row[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
        row[C / 4] = make_float4(r, add, 0.f, 0.f);
    }
}

template <int C>
__device__ __forceinline__ float row_dist(const float4 *row, const float (&q)[C], float rq)
{
    float dot = 0.f, rp;
    if (C == 3) {
        const float4 p = row[0];
        dot = __builtin_fmaf(q[0], p.x, dot);
        dot = __builtin_fmaf(q[1], p.y, dot);
        dot = __builtin_fmaf(q[2], p.z, dot);
        rp = p.w;
    } else {
#pragma unroll
        for (int i = 0; i < C / 4; ++i) {
            const float4 p = row[i];
            dot = __builtin_fmaf(q[4 * i + 0], p.x, dot);
            dot = __builtin_fmaf(q[4 * i + 1], p.y, dot);
            dot = __builtin_fmaf(q[4 * i + 2], p.z, dot);
            dot = __builtin_fmaf(q[4 * i + 3], p.w, dot);
        }
        rp = row[C / 4].x;
    }
    return __builtin_fmaf(-2.f, dot, rq) + rp;
}

template <int C>
__device__ __forceinline__ void load_query(float (&q)[C], float &rq, const float *__restrict__ src, int c, bool live)
{
#pragma unroll
    for (int i = 0; i < C; ++i)
        q[i] = (live && i < c) ? src[i] : 0.f;
    rq = 0.f;
#pragma unroll
    for (int i = 0; i < C; ++i)
        rq = __builtin_fmaf(q[i], q[i], rq);
}

constexpr int tile_rows(int C) { return C == 3 ? 1024 : (C <= 8 ? 512 : (C <= 32 ? 256 : 128)); }

// ---------------------------------------------------------------------------------------------
// small k: lane-per-query register insertion
// ---------------------------------------------------------------------------------------------
template <int C, int KMAX>
__global__ __launch_bounds__(512) void knn_insert_kernel(KnnArgs a)
{
    constexpr int TILE = tile_rows(C);
    constexpr int F4 = Row<C>::F4;
    __shared__ float4 tile[TILE * F4];
    __shared__ float addend[C == 3 ? TILE : 1];
    const int b = blockIdx.y;
    const int pb = a.pts_of ? a.pts_of[b] : b;
    const int n = a.n_arr ? a.n_arr[pb] : a.n;
    const int m = a.m_arr ? a.m_arr[b] : a.m;
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = qi < m;
    const bool use_dup = a.dup != nullptr && a.uws[0] != 0;
    const float dmax = use_dup ? tpu3_unmono(a.uws[4 + (a.grp ? a.grp[b] : 0)]) : 0.f;

    float q[C], rq;
    load_query<C>(q, rq, a.query + ((size_t)b * a.m + (live ? qi : 0)) * a.c, a.c, live);

    float bd[KMAX];
    int bi[KMAX];
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
        bd[i] = __builtin_inff();
        bi[i] = 0x7FFFFFFF;
    }
    const float *P = a.points + (size_t)pb * a.n * a.c;
    const uint8_t *DUP = use_dup ? a.dup + (size_t)pb * a.n : nullptr;

    for (int j0 = 0; j0 < n; j0 += TILE) {
        const int len = min(TILE, n - j0);
        __syncthreads();
        for (int i = threadIdx.x; i < len; i += blockDim.x) {
            const float add = use_dup ? dmax * (float)DUP[j0 + i] : 0.f;
            stage_row<C>(tile + i * F4, P + (size_t)(j0 + i) * a.c, a.c, add);
            if (C == 3)
                addend[i] = add;
        }
        __syncthreads();
        if (live) {
            for (int j = 0; j < len; ++j) {
                float d = row_dist<C>(tile + j * F4, q, rq);
                if (use_dup)
                    d = d + (C == 3 ? addend[j] : tile[j * F4 + C / 4].y);
                if (d < bd[KMAX - 1]) {
                    // sorted insertion; an equal distance goes behind the (lower-index) holder
                    const int id = j0 + j;
#pragma unroll
                    for (int i = KMAX - 1; i > 0; --i) {
                        const bool up = bd[i - 1] > d;      // predecessor moves into slot i
                        const bool here = bd[i] > d;        // else the candidate lands here
                        bi[i] = up ? bi[i - 1] : (here ? id : bi[i]);
                        bd[i] = up ? bd[i - 1] : (here ? d : bd[i]);
                    }
                    if (bd[0] > d) {
                        bd[0] = d;
                        bi[0] = id;
                    }
                }
            }
        }
    }
    if (live) {
        const size_t o = ((size_t)b * a.m + qi) * a.k;
#pragma unroll
        for (int i = 0; i < KMAX; ++i)
            if (i < a.k) {
                store_idx(a, o + i, bi[i]);
                if (a.dist)
                    a.dist[o + i] = bd[i];
            }
    }
}

// ---------------------------------------------------------------------------------------------
// any k: workgroup-per-query bitonic selection
// ---------------------------------------------------------------------------------------------
template <int LOG2S>
__global__ __launch_bounds__(1024) void knn_sort_kernel(KnnArgs a)
{
    constexpr int S = 1 << LOG2S;
    __shared__ uint64_t keys[S];
    const int b = blockIdx.y, qi = blockIdx.x;
    const int pb = a.pts_of ? a.pts_of[b] : b;
    const int n = a.n_arr ? a.n_arr[pb] : a.n;
    const int m = a.m_arr ? a.m_arr[b] : a.m;
    if (qi >= m)
        return;
    const int W = blockDim.x, t = threadIdx.x;
    const int c = a.c, k = a.k;
    const bool use_dup = a.dup != nullptr && a.uws[0] != 0;
    const float dmax = use_dup ? tpu3_unmono(a.uws[4 + (a.grp ? a.grp[b] : 0)]) : 0.f;
    const float *P = a.points + (size_t)pb * a.n * c;
    const float *Q = a.query + ((size_t)b * a.m + qi) * c;
    const uint8_t *DUP = use_dup ? a.dup + (size_t)pb * a.n : nullptr;
    float rq = 0.f;
    for (int i = 0; i < c; ++i)
        rq = __builtin_fmaf(Q[i], Q[i], rq);

    int pos = 0;      // next unread candidate
    int base = 0;     // slots [0, base) hold the best-so-far (sorted)
    while (true) {
        const int take = min(n - pos, S - base);
        for (int s = base + t; s < S; s += W) {
            uint64_t key = ~0ull;
            const int j = pos + (s - base);
            if (s - base < take) {
                const float *p = P + (size_t)j * c;
                float dot = 0.f, rp = 0.f;
                for (int i = 0; i < c; ++i) {
                    const float v = p[i];
                    dot = __builtin_fmaf(Q[i], v, dot);
                    rp = __builtin_fmaf(v, v, rp);
                }
                float d = __builtin_fmaf(-2.f, dot, rq) + rp;
                if (use_dup)
                    d = d + dmax * (float)DUP[j];
                key = ((uint64_t)tpu3_mono(d) << 32) | (uint32_t)j;
            }
            keys[s] = key;
        }
        pos += take;
        // bitonic sort, ascending
        for (int size = 2; size <= S; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                __syncthreads();
                for (int u = t; u < S / 2; u += W) {
                    const int lo = 2 * u - (u & (stride - 1));
                    const int hi = lo + stride;
                    const bool asc = (lo & size) == 0;
                    const uint64_t x = keys[lo], y = keys[hi];
                    if ((x > y) == asc) {
                        keys[lo] = y;
                        keys[hi] = x;
                    }
                }
            }
        __syncthreads();
        if (pos >= n)
            break;
        base = k;
    }
    const size_t o = ((size_t)b * a.m + qi) * k;
    for (int i = t; i < k; i += W) {
        const uint64_t key = keys[i];
        store_idx(a, o + i, (int)(uint32_t)key);
        if (a.dist)
            a.dist[o + i] = tpu3_unmono((uint32_t)(key >> 32));
    }
}

// ---------------------------------------------------------------------------------------------
// unique=True pre-pass
// ---------------------------------------------------------------------------------------------
// dup[i] = 1 iff a row j < i is elementwise equal (float ==, so -0.0 == 0.0, NaN != NaN: the
// comparison np.unique's lexicographic sort performs).
__global__ __launch_bounds__(256) void knn_dup_kernel(int n_pad, int c, const float *__restrict__ points,
                                                      const int32_t *__restrict__ n_arr,
                                                      uint8_t *__restrict__ dup, uint32_t *__restrict__ uws)
{
    constexpr int TILE = 256;
    __shared__ float first[TILE];
    const int b = blockIdx.y;
    const int n = n_arr ? n_arr[b] : n_pad;
    const int i0 = blockIdx.x * blockDim.x;
    if (i0 >= n)
        return;
    const int i = i0 + threadIdx.x;
    const bool live = i < n;
    const float *P = points + (size_t)b * n_pad * c;
    const float mine = live ? P[(size_t)i * c] : 0.f;
    bool found = false;
    const int jend = min(n, i0 + (int)blockDim.x);
    for (int j0 = 0; j0 < jend; j0 += TILE) {
        const int len = min(TILE, jend - j0);
        __syncthreads();
        if ((int)threadIdx.x < len)
            first[threadIdx.x] = P[(size_t)(j0 + threadIdx.x) * c];
        __syncthreads();
        if (live && !found) {
            const int lim = min(len, i - j0);       // only j < i
            for (int j = 0; j < lim; ++j) {
                if (first[j] == mine) {
                    bool same = true;
                    for (int ch = 1; ch < c && same; ++ch)
                        same = P[(size_t)(j0 + j) * c + ch] == P[(size_t)i * c + ch];
                    if (same) {
                        found = true;
                        break;
                    }
                }
            }
        }
    }
    if (live) {
        dup[(size_t)b * n_pad + i] = found ? 1 : 0;
        if (found)
            uws[0] = 1u;
    }
}

// max over every (query, point) distance of the whole problem; skipped when nothing is dup
template <int C>
__global__ __launch_bounds__(512) void knn_dmax_kernel(KnnArgs a, uint32_t *uws)
{
    if (uws[0] == 0)
        return;
    constexpr int TILE = tile_rows(C);
    constexpr int F4 = Row<C>::F4;
    __shared__ float4 tile[TILE * F4];
    __shared__ uint32_t red[16];
    const int b = blockIdx.y;
    const int pb = a.pts_of ? a.pts_of[b] : b;
    const int n = a.n_arr ? a.n_arr[pb] : a.n;
    const int m = a.m_arr ? a.m_arr[b] : a.m;
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = qi < m;
    float q[C], rq;
    load_query<C>(q, rq, a.query + ((size_t)b * a.m + (live ? qi : 0)) * a.c, a.c, live);
    const float *P = a.points + (size_t)pb * a.n * a.c;
    uint32_t best = 0;          // mono() of anything is > 0 except -NaN patterns
    for (int j0 = 0; j0 < n; j0 += TILE) {
        const int len = min(TILE, n - j0);
        __syncthreads();
        for (int i = threadIdx.x; i < len; i += blockDim.x)
            stage_row<C>(tile + i * F4, P + (size_t)(j0 + i) * a.c, a.c, 0.f);
        __syncthreads();
        if (live)
            for (int j = 0; j < len; ++j)
                best = max(best, tpu3_mono(row_dist<C>(tile + j * F4, q, rq)));
    }
    best = tpu3_wave_max_u32(best);
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t r = 0;
        for (int w = 0; w < (int)(blockDim.x + 63) / 64; ++w)
            r = max(r, red[w]);
        atomicMax(uws + 4 + (a.grp ? a.grp[b] : 0), r);
    }
}

__global__ __launch_bounds__(256) void knn_dmax_generic_kernel(KnnArgs a, uint32_t *uws)
{
    // any channel count: one lane per query, points read straight from global memory (L2)
    if (uws[0] == 0)
        return;
    __shared__ uint32_t red[4];
    const int b = blockIdx.y;
    const int pb = a.pts_of ? a.pts_of[b] : b;
    const int n = a.n_arr ? a.n_arr[pb] : a.n;
    const int m = a.m_arr ? a.m_arr[b] : a.m;
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t best = 0;
    if (qi < m) {
        const float *Q = a.query + ((size_t)b * a.m + qi) * a.c;
        const float *P = a.points + (size_t)pb * a.n * a.c;
        float rq = 0.f;
        for (int i = 0; i < a.c; ++i)
            rq = __builtin_fmaf(Q[i], Q[i], rq);
        for (int j = 0; j < n; ++j) {
            float dot = 0.f, rp = 0.f;
            for (int i = 0; i < a.c; ++i) {
                const float v = P[(size_t)j * a.c + i];
                dot = __builtin_fmaf(Q[i], v, dot);
                rp = __builtin_fmaf(v, v, rp);
            }
            best = max(best, tpu3_mono(__builtin_fmaf(-2.f, dot, rq) + rp));
        }
    }
    best = tpu3_wave_max_u32(best);
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(uws + 4 + (a.grp ? a.grp[b] : 0), max(max(red[0], red[1]), max(red[2], red[3])));
}

// grouped[b,q,t,:] = points[b, idx[b,q,t], :]
__global__ __launch_bounds__(256) void knn_group_kernel(KnnArgs a, float *__restrict__ grouped, long total)
{
    const int c = a.c;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(e % c);
        const long r = e / c;                        // (b*m + q)*k + t
        const int b = (int)(r / ((long)a.m * a.k));
        const int j = a.idx64 ? (int)((const int64_t *)a.idx)[r] : ((const int32_t *)a.idx)[r];
        const int pb = a.pts_of ? a.pts_of[b] : b;
        const int n = a.n_arr ? a.n_arr[pb] : a.n;
        grouped[e] = (j >= 0 && j < n) ? a.points[((size_t)pb * a.n + j) * c + ch] : 0.f;
    }
}

template <int C, int KMAX>
int launch_insert(hipStream_t s, int b, const KnnArgs &a)
{
    int threads = ((a.m + 63) / 64) * 64;
    if (threads > 512)
        threads = 256;
    const dim3 g((a.m + threads - 1) / threads, b);
    hipLaunchKernelGGL((knn_insert_kernel<C, KMAX>), g, dim3(threads), 0, s, a);
    return tpu3_launch_status();
}

template <int C>
int dispatch_insert_k(hipStream_t s, int b, const KnnArgs &a)
{
    if (a.k <= 2) return launch_insert<C, 2>(s, b, a);
    if (a.k <= 8) return launch_insert<C, 8>(s, b, a);
    if (a.k <= 16) return launch_insert<C, 16>(s, b, a);
    if (a.k <= 33) return launch_insert<C, 33>(s, b, a);
    return launch_insert<C, 64>(s, b, a);
}

int dispatch_insert(hipStream_t s, int b, const KnnArgs &a)
{
    if (a.c == 3) return dispatch_insert_k<3>(s, b, a);
    if (a.c <= 8) return dispatch_insert_k<8>(s, b, a);
    if (a.c <= 16) return dispatch_insert_k<16>(s, b, a);
    if (a.c <= 24) return dispatch_insert_k<24>(s, b, a);
    if (a.c <= 32) return dispatch_insert_k<32>(s, b, a);
    return -100;    // not handled here
}

template <int LOG2S>
int launch_sort(hipStream_t s, int b, const KnnArgs &a)
{
    constexpr int S = 1 << LOG2S;
    const int threads = S / 2 > 1024 ? 1024 : (S / 2 < 64 ? 64 : S / 2);
    hipLaunchKernelGGL((knn_sort_kernel<LOG2S>), dim3(a.m, b), dim3(threads), 0, s, a);
    return tpu3_launch_status();
}

int dispatch_sort(hipStream_t s, int b, const KnnArgs &a)
{
    // slots: at least 2k (so every pass makes progress) and, if it fits, the whole candidate
    // set in one pass; capped at 8192 keys = 64 KiB of LDS
    long need = a.n < 2L * a.k ? 2L * a.k : a.n;
    if (need > 8192) need = 8192;
    if (2L * a.k > 8192) return TPU3_ELIMIT;
    if (need <= 128) return launch_sort<7>(s, b, a);
    if (need <= 256) return launch_sort<8>(s, b, a);
    if (need <= 512) return launch_sort<9>(s, b, a);
    if (need <= 1024) return launch_sort<10>(s, b, a);
    if (need <= 2048) return launch_sort<11>(s, b, a);
    if (need <= 4096) return launch_sort<12>(s, b, a);
    return launch_sort<13>(s, b, a);
}

bool bad_dims(int b, int m, int n, int c, int k)
{
    return b < 0 || m < 0 || n < 0 || c <= 0 || k < 0;
}

} // namespace

extern "C" int tpu3_knn_f32(tpu3_stream_t stream, int b, int m, int n, int c, int k,
                            const float *query, const float *points, const tpu3_knn_layout *layout,
                            const uint8_t *dup, const uint32_t *uws, void *idx, int idx_elem_size,
                            float *dist, float *grouped)
{
    if (bad_dims(b, m, n, c, k)) return TPU3_EINVAL;
    if (idx_elem_size != 4 && idx_elem_size != 8) return TPU3_EINVAL;
    if ((dup == nullptr) != (uws == nullptr)) return TPU3_EINVAL;
    if (b == 0 || m == 0 || k == 0) return TPU3_OK;
    if (k > n) return TPU3_EINVAL;             // operations.py:188 "points size must be >= k"
    if (!query || !points || !idx) return TPU3_EINVAL;
    if (b > 65535) return TPU3_ELIMIT;
    hipStream_t s = (hipStream_t)stream;
    const tpu3_knn_layout L = layout ? *layout : tpu3_knn_layout{nullptr, nullptr, nullptr, nullptr, b, 1};
    KnnArgs a{m, n, c, k, query, points, L.n_arr, L.m_arr, L.pts_of, L.grp, dup, uws, idx,
              idx_elem_size == 8, dist};
    int r = -100;
    if (k <= 64)
        r = dispatch_insert(s, b, a);
    if (r == -100)
        r = dispatch_sort(s, b, a);
    if (r) return r;
    if (grouped) {
        const long total = (long)b * m * k * c;
        long blocks = (total + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(knn_group_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, grouped, total);
        r = tpu3_launch_status();
    }
    return r;
}

extern "C" int tpu3_knn_unique_prepare_f32(tpu3_stream_t stream, int b, int m, int n, int c,
                                           const float *query, const float *points,
                                           const tpu3_knn_layout *layout, uint8_t *dup, uint32_t *uws)
{
    if (bad_dims(b, m, n, c, 0)) return TPU3_EINVAL;
    if (!dup || !uws) return TPU3_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const tpu3_knn_layout L = layout ? *layout : tpu3_knn_layout{nullptr, nullptr, nullptr, nullptr, b, 1};
    const int bp = L.pts_of ? L.bp : b;
    const int groups = L.grp ? L.groups : 1;
    if (bp < 0 || groups < 1) return TPU3_EINVAL;
    hipError_t e = hipMemsetAsync(uws, 0, (size_t)TPU3_KNN_UWS_WORDS(groups) * sizeof(uint32_t), s);
    if (e != hipSuccess) return (int)e;
    if (b == 0 || bp == 0 || n == 0) return TPU3_OK;
    if (!points) return TPU3_EINVAL;
    if (b > 65535 || bp > 65535) return TPU3_ELIMIT;
    hipLaunchKernelGGL(knn_dup_kernel, dim3((n + 255) / 256, bp), dim3(256), 0, s, n, c, points, L.n_arr,
                       dup, uws);
    int r = tpu3_launch_status();
    if (r || m == 0) return r;
    if (!query) return TPU3_EINVAL;
    KnnArgs a{m, n, c, 0, query, points, L.n_arr, L.m_arr, L.pts_of, L.grp, nullptr, nullptr, nullptr,
              0, nullptr};
    int threads = ((m + 63) / 64) * 64;
    if (threads > 512) threads = 256;
    const dim3 g((m + threads - 1) / threads, b);
    if (c == 3)
        hipLaunchKernelGGL(knn_dmax_kernel<3>, g, dim3(threads), 0, s, a, uws);
    else if (c <= 8)
        hipLaunchKernelGGL(knn_dmax_kernel<8>, g, dim3(threads), 0, s, a, uws);
    else if (c <= 16)
        hipLaunchKernelGGL(knn_dmax_kernel<16>, g, dim3(threads), 0, s, a, uws);
    else if (c <= 24)
        hipLaunchKernelGGL(knn_dmax_kernel<24>, g, dim3(threads), 0, s, a, uws);
    else if (c <= 32)
        hipLaunchKernelGGL(knn_dmax_kernel<32>, g, dim3(threads), 0, s, a, uws);
    else
        hipLaunchKernelGGL(knn_dmax_generic_kernel, dim3((m + 255) / 256, b), dim3(256), 0, s, a, uws);
    return tpu3_launch_status();
}
