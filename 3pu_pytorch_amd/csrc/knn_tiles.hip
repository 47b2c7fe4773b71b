// knn_tiles.hip -- small-k nearest neighbours of 3-d queries in a LARGE 3-d point set, pruned by space (gfx950).
//
// The inter-level skip connection of a Level (network/upsampler.py:324-325: group_knn(fm_knn = 5, xyz, previous_xyz,
// unique=True)) asks, for every point of every patch, for its 5 nearest points of the previous level's merged cloud:
// 99 840 queries against ~20 000 distinct points per cloud at level 4.  tpu3_knn_f32 answers it by brute force over the
// de-duplicated candidate list -- 82 G pair distances per 32-cloud step, 12.9 ms at a third of the vector peak.  The
// queries of a patch are neighbours in space, and so are the candidates that matter:
//
//   build (once per previous cloud, shared by all its patches):  the candidates ordered by a 12-bit Morton cell of the
//          cloud's box (an LDS counting sort, one workgroup per cloud), cut into TILES of 64 with their bounding boxes;
//          positions as (x, y, z, |p|^2) rows, original row indices beside them;
//   query: a WAVE per 64 consecutive queries of a patch.  The wave's bounding box against every tile box gives a lower
//          bound L of every pair distance; the nearest tile is searched first, its k-th distances give the bound U,
//          and only tiles with L <= U (plus the rounding margin below) are searched after it, U shrinking on the way.
//          ~8 of ~310 tiles per wave.
//
// Exactness.  The distances are the reference's own arithmetic (tpu3_knn_f32: D = fmaf(-2, <q,p>, |q|^2) + |p|^2, dot and
// norms as ascending-channel fmaf chains), compared as (D, row index) pairs -- the result does not depend on the order
// in which candidates are visited, ties go to the lower row index as in the brute-force kernel.  A tile is skipped only
// if NO computed distance of it can be below the wave's largest k-th distance: the computed D differs from the true
// squared distance by at most 2^-20 (|q|^2 + |p|^2) (three 3-term fmaf chains and two additions, each rounding relative
// 2^-24 on magnitudes <= 2 (|q|^2 + |p|^2)), and the box bound L by a relative 2^-21; both are in the margin.
// unique=True: the candidates are the first occurrences (tpu3_knn_unique_compact_i32); the verification that the
// duplicates stay out of the top k under the reference's D + max(D) * dup uses a LOWER bound of max(D) here (the
// largest distance seen and the bound L of the farthest tile), which keeps it sufficient; a query that cannot verify
// raises uws[1] and the caller's gated launches redo the call with the reference arithmetic, as before.  (When the
// list holds fewer than k rows, the k-th "distance" is NaN and the verification fails by construction.)
#include "tpu3_dev.h"

#include <cstdlib>

namespace {

constexpr int KT_TILE = 64;

struct KtBuildArgs {
    int bp, n, tiles;                // point sets, rows per set, tiles per set (= ceil(n / 64))
    const float *points;             // (bp,n,3)
    const int32_t *n_arr;            // (bp) live rows or null
    const int32_t *cand;             // (bp,n) first occurrences (valid when uws[0] != 0)
    const int32_t *cand_count;       // (bp)
    const uint32_t *uws;
    float *bbox;                     // (bp,8)
    unsigned long long *keys;        // (bp*n)
    int32_t *vals;                   // (bp*n)
    float4 *tile_pts;                // (bp, tiles*64)
    int32_t *tile_idx;               // (bp, tiles*64)
    float *tile_box;                 // (bp, tiles, 8): lo xyz, hi xyz, max |p|^2, live members
};

__device__ __forceinline__ int kt_count(const KtBuildArgs &a, int pb)
{
    return a.uws[0] != 0 ? a.cand_count[pb] : (a.n_arr ? a.n_arr[pb] : a.n);
}

__device__ __forceinline__ int kt_row(const KtBuildArgs &a, int pb, int j)
{
    return a.uws[0] != 0 ? a.cand[(size_t)pb * a.n + j] : j;
}

__device__ __forceinline__ float kt_wave_min_f32(float v) { return -tpu3_wave_max_f32(-v); }

// Spatial order of a set's candidates: one workgroup per set.  Bounding box, then a counting sort by a 12-bit Morton
// cell (16 cells per axis) entirely in LDS: histogram, scan, scatter through LDS atomics.  (A device radix sort of
// (set, 30-bit code) keys did the same in ~6 launches and 0.46 ms per call -- 24 calls per bench step; the order INSIDE a
// cell is irrelevant to the tiles' quality, and to the result: the search is exact for any tiling.)
constexpr int KT_CELLS = 4096;

__device__ __forceinline__ uint32_t kt_spread4(uint32_t v)
{
    v &= 0xFu;
    v = (v | (v << 4)) & 0x0C3u;
    v = (v | (v << 2)) & 0x249u;
    return v;
}

__global__ __launch_bounds__(1024) void kt_bin_kernel(KtBuildArgs a)
{
    __shared__ float red[16][6];
    __shared__ float box[6];
    __shared__ int hist[KT_CELLS];
    __shared__ int wsum[16];
    const int pb = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int count = kt_count(a, pb);
    const float inf = __builtin_inff();
    float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    for (int j = tid; j < count; j += 1024) {
        const float *p = a.points + ((size_t)pb * a.n + kt_row(a, pb, j)) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lo[c] = fminf(lo[c], p[c]);
            hi[c] = fmaxf(hi[c], p[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        lo[c] = kt_wave_min_f32(lo[c]);
        hi[c] = tpu3_wave_max_f32(hi[c]);
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            red[wave][c] = lo[c];
            red[wave][3 + c] = hi[c];
        }
    }
    for (int i = tid; i < KT_CELLS; i += 1024)
        hist[i] = 0;
    __syncthreads();
    if (tid < 6) {
        float v = red[0][tid];
        for (int w = 1; w < 16; ++w)
            v = tid < 3 ? fminf(v, red[w][tid]) : fmaxf(v, red[w][tid]);
        box[tid] = v;
    }
    __syncthreads();
    float sc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float ext = box[3 + c] - box[c];
        sc[c] = ext > 0.f ? 16.f / ext : 0.f;
    }
    auto cell_of = [&](const float *p) __attribute__((always_inline)) {
        uint32_t q[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int v = (int)((p[c] - box[c]) * sc[c]);
            q[c] = (uint32_t)(v < 0 ? 0 : (v > 15 ? 15 : v));
        }
        return (int)(kt_spread4(q[0]) | (kt_spread4(q[1]) << 1) | (kt_spread4(q[2]) << 2));
    };
    for (int j = tid; j < count; j += 1024)
        atomicAdd(&hist[cell_of(a.points + ((size_t)pb * a.n + kt_row(a, pb, j)) * 3)], 1);
    __syncthreads();
    // exclusive scan of the 4096 counters: four per thread, wave scans, the waves' totals
    int c4[4], run = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        c4[u] = hist[4 * tid + u];
        run += c4[u];
    }
    int inc = run;                                      // inclusive scan over the wave's lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        inc += lane >= d ? o : 0;
    }
    if (lane == 63)
        wsum[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w)
        base += wsum[w];
    int off = base + inc - run;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        hist[4 * tid + u] = off;
        off += c4[u];
    }
    __syncthreads();
    int32_t *out = a.vals + (size_t)pb * a.n;
    for (int j = tid; j < count; j += 1024) {
        const int row = kt_row(a, pb, j);
        const int pos = atomicAdd(&hist[cell_of(a.points + ((size_t)pb * a.n + row) * 3)], 1);
        out[pos] = row;
    }
    for (int j = count + tid; j < a.n; j += 1024)
        out[j] = -1;
}

// one wave per tile: its 64 (x, y, z, |p|^2) rows, the rows' indices, its box
__global__ __launch_bounds__(256) void kt_fill_kernel(KtBuildArgs a, const int32_t *__restrict__ sorted_rows)
{
    const int lane = threadIdx.x & 63;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= (long)a.bp * a.tiles)
        return;
    const int pb = (int)(tile / a.tiles), t = (int)(tile - (long)pb * a.tiles);
    const int i = t * KT_TILE + lane;                       // position in the set's sorted list
    const int row = i < a.n ? sorted_rows[(size_t)pb * a.n + i] : -1;
    const bool live = row >= 0;
    const float inf = __builtin_inff();
    float x = 0.f, y = 0.f, z = 0.f, r = inf;
    if (live) {
        const float *p = a.points + ((size_t)pb * a.n + row) * 3;
        x = p[0]; y = p[1]; z = p[2];
        r = __builtin_fmaf(x, x, 0.f);                      // |p|^2 as tpu3_knn_f32 forms it: ascending-channel chain
        r = __builtin_fmaf(y, y, r);
        r = __builtin_fmaf(z, z, r);
    }
    const size_t o = ((size_t)pb * a.tiles + t) * KT_TILE + lane;
    a.tile_pts[o] = make_float4(x, y, z, r);
    a.tile_idx[o] = row;
    const float lx = kt_wave_min_f32(live ? x : inf), ly = kt_wave_min_f32(live ? y : inf), lz = kt_wave_min_f32(live ? z : inf);
    const float hx = tpu3_wave_max_f32(live ? x : -inf), hy = tpu3_wave_max_f32(live ? y : -inf);
    const float hz = tpu3_wave_max_f32(live ? z : -inf), rm = tpu3_wave_max_f32(live ? r : 0.f);
    const int members = __builtin_popcountll(__ballot(live));
    if (lane == 0) {
        float *b = a.tile_box + ((size_t)pb * a.tiles + t) * 8;
        b[0] = lx; b[1] = ly; b[2] = lz; b[3] = hx; b[4] = hy; b[5] = hz; b[6] = rm; b[7] = (float)members;
    }
}

struct KtQueryArgs {
    int b, m, n, k, tiles;
    const float *query;              // (b,m,3)
    const int32_t *m_arr, *n_arr, *pts_of, *cand_count;
    uint32_t *uws;
    const float4 *tile_pts;
    const int32_t *tile_idx;
    const float *tile_box;
    void *idx;
    int idx64;
    float *dist;
    unsigned *dbg;                   // development probe: waves, tiles searched, tiles tested per query, tiles per set
};

// SORTED (r5): a workgroup per query set of up to KT_QMAX queries (a 312-point patch: five waves).  The queries of a
// patch arrive in kNN order from its seed, so 64 consecutive ones lie on a RING around the seed and their box covers
// most of the patch (6.4 of 10 / 9.0 of 20 tiles searched per wave on the pipeline's previous sets).  Here the
// workgroup first orders its queries by a 9-bit Morton cell of the set's own box (an LDS counting sort, four barriers)
// and wave w takes sorted positions 64 w ..: a compact blob.  Each result row is written to the query's ORIGINAL
// position; the search itself -- exact for any grouping of queries into waves -- is unchanged.
constexpr int KT_QMAX = 512;
constexpr int KT_QCELLS = 512;

template <int K, bool SORTED>
__global__ __launch_bounds__(SORTED ? KT_QMAX : 256) void kt_query_kernel(KtQueryArgs a)
{
    constexpr int NWV = SORTED ? KT_QMAX / 64 : 4;
    __shared__ float4 stage[NWV][KT_TILE];               // per wave: the tile being searched, (x, y, z, |p|^2) rows
    __shared__ int32_t srow[NWV][KT_TILE];               // ... and its members' row indices
    __shared__ float4 sq[SORTED ? KT_QMAX : 1];          // SORTED: the queries in Morton order, .w = original position
    __shared__ int qhist[SORTED ? KT_QCELLS : 1];
    __shared__ float qred[SORTED ? NWV : 1][6];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wpq = (a.m + 63) >> 6;
    int b, qi;
    if constexpr (SORTED) {
        b = blockIdx.x;
        qi = threadIdx.x;
    } else {
        const long w = (long)blockIdx.x * 4 + wv;
        if (w >= (long)a.b * wpq)
            return;
        b = (int)(w / wpq);
        qi = (int)(w - (long)b * wpq) * 64 + lane;
    }
    const int pb = a.pts_of ? a.pts_of[b] : b;
    const int m = a.m_arr ? a.m_arr[b] : a.m;
    const float inf = __builtin_inff();
    bool live = qi < m;
    float q0, q1, q2;
    {
        const float *qp = a.query + ((size_t)b * a.m + (live ? qi : 0)) * 3;
        q0 = live ? qp[0] : 0.f, q1 = live ? qp[1] : 0.f, q2 = live ? qp[2] : 0.f;
    }
    if constexpr (SORTED) {
        const int nthreads = blockDim.x, nwaves = nthreads >> 6;
        // the set's box
        {
            const float l0 = kt_wave_min_f32(live ? q0 : inf), l1 = kt_wave_min_f32(live ? q1 : inf), l2 = kt_wave_min_f32(live ? q2 : inf);
            const float h0 = tpu3_wave_max_f32(live ? q0 : -inf), h1 = tpu3_wave_max_f32(live ? q1 : -inf);
            const float h2 = tpu3_wave_max_f32(live ? q2 : -inf);
            if (lane == 0) {
                qred[wv][0] = l0; qred[wv][1] = l1; qred[wv][2] = l2;
                qred[wv][3] = h0; qred[wv][4] = h1; qred[wv][5] = h2;
            }
        }
        for (int i = threadIdx.x; i < KT_QCELLS; i += nthreads)
            qhist[i] = 0;
        __syncthreads();
        float lo[3], sc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float l = qred[0][c], h = qred[0][3 + c];
            for (int w = 1; w < nwaves; ++w) {
                l = fminf(l, qred[w][c]);
                h = fmaxf(h, qred[w][3 + c]);
            }
            lo[c] = l;
            sc[c] = h - l > 0.f ? 8.f / (h - l) : 0.f;
        }
        // 3 bits per axis (NaN coordinates fall into cell 0 through the clamps: any order is a valid order)
        auto bits3 = [](float v) __attribute__((always_inline)) {
            const int i = (int)fminf(fmaxf(v, 0.f), 7.f);
            return (uint32_t)((i & 1) | ((i & 2) << 2) | ((i & 4) << 4));
        };
        const int cell = (int)(bits3((q0 - lo[0]) * sc[0]) | (bits3((q1 - lo[1]) * sc[1]) << 1) | (bits3((q2 - lo[2]) * sc[2]) << 2));
        int slot = 0;
        if (live)
            slot = atomicAdd(&qhist[cell], 1);
        __syncthreads();
        if (wv == 0) {
            // exclusive offsets of the 512 cells: eight consecutive cells per lane
            int h[KT_QCELLS / 64], sum = 0;
#pragma unroll
            for (int u = 0; u < KT_QCELLS / 64; ++u) {
                h[u] = qhist[lane * (KT_QCELLS / 64) + u];
                sum += h[u];
            }
            int inc = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(inc, d, 64);
                inc += lane >= d ? o : 0;
            }
            int off = inc - sum;
#pragma unroll
            for (int u = 0; u < KT_QCELLS / 64; ++u) {
                qhist[lane * (KT_QCELLS / 64) + u] = off;
                off += h[u];
            }
        }
        __syncthreads();
        if (live)
            sq[qhist[cell] + slot] = make_float4(q0, q1, q2, __int_as_float(qi));
        __syncthreads();
        // sorted position threadIdx.x: the live queries fill positions 0 .. m - 1
        live = (int)threadIdx.x < m;
        const float4 mine = sq[live ? threadIdx.x : 0];
        q0 = live ? mine.x : 0.f, q1 = live ? mine.y : 0.f, q2 = live ? mine.z : 0.f;
        qi = live ? __float_as_int(mine.w) : 0;         // the row the result belongs to
    }
    if (__ballot(live) == 0)
        return;
    const int nlive = a.n_arr ? a.n_arr[pb] : a.n;
    const bool anydup = a.uws[0] != 0;
    const int count = anydup ? a.cand_count[pb] : nlive;
    const int ntile = (count + KT_TILE - 1) / KT_TILE;

    float rq = __builtin_fmaf(q0, q0, 0.f);
    rq = __builtin_fmaf(q1, q1, rq);
    rq = __builtin_fmaf(q2, q2, rq);
    // the wave's box and largest |q|^2
    const float wl0 = kt_wave_min_f32(live ? q0 : inf), wl1 = kt_wave_min_f32(live ? q1 : inf), wl2 = kt_wave_min_f32(live ? q2 : inf);
    const float wh0 = tpu3_wave_max_f32(live ? q0 : -inf), wh1 = tpu3_wave_max_f32(live ? q1 : -inf);
    const float wh2 = tpu3_wave_max_f32(live ? q2 : -inf), wrq = tpu3_wave_max_f32(live ? rq : 0.f);

    unsigned long long kb[K];            // (order-preserving bits of D) << 32 | row, ascending
#pragma unroll
    for (int i = 0; i < K; ++i)
        kb[i] = ~0ull;
    float dlast = inf;                   // D of kb[K - 1] (+inf while the list is not full): the cheap test per candidate

    const float *TB = a.tile_box + (size_t)pb * a.tiles * 8;
    const float4 *TP = a.tile_pts + (size_t)pb * a.tiles * KT_TILE;
    const int32_t *TI = a.tile_idx + (size_t)pb * a.tiles * KT_TILE;

    // lower bound of every pair distance (wave's queries x tile's members) and the margin that makes a skip safe
    auto tile_bound = [&](int t, float &L, float &E, float4 &lo, float4 &hi) __attribute__((always_inline)) {
        lo = *(const float4 *)(TB + (size_t)t * 8);
        hi = *(const float4 *)(TB + (size_t)t * 8 + 4);
        const float d0 = fmaxf(fmaxf(lo.x - wh0, wl0 - lo.w), 0.f);         // (lo.w = hi.x, hi = (hy, hz, max |p|^2, members))
        const float d1 = fmaxf(fmaxf(lo.y - wh1, wl1 - hi.x), 0.f);
        const float d2 = fmaxf(fmaxf(lo.z - wh2, wl2 - hi.y), 0.f);
        L = (d0 * d0 + d1 * d1 + d2 * d2) * (1.f - 0x1p-20f);
        E = 0x1p-20f * (wrq + hi.z);
    };
    int nsearch = 0, ntest = 0;
    auto search = [&](int t) __attribute__((always_inline)) {
        ++nsearch;
        const int base = t * KT_TILE;
        const int len = min(KT_TILE, count - base);
        // the tile through LDS: one coalesced 1 KB read, then wave-uniform 16-byte broadcasts
        stage[wv][lane] = TP[base + lane];
        srow[wv][lane] = TI[base + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // four members per step (slots beyond the list carry |p|^2 = +inf: their D is +inf and never enters): the four
        // broadcasts in flight together, ONE comparison on the smallest of the four distances
        auto enter = [&](float d, int j) __attribute__((always_inline)) {
            if (d <= dlast) {                            // (equal: the lower row may still go in front)
                const unsigned long long key = ((unsigned long long)tpu3_mono(d) << 32) | (uint32_t)srow[wv][j];
                if (key < kb[K - 1]) {
#pragma unroll
                    for (int i = K - 1; i > 0; --i) {
                        const bool up = kb[i - 1] > key;        // predecessor moves into slot i
                        const bool here = kb[i] > key;          // else the candidate lands here
                        kb[i] = up ? kb[i - 1] : (here ? key : kb[i]);
                    }
                    if (kb[0] > key)
                        kb[0] = key;
                    const uint32_t top = (uint32_t)(kb[K - 1] >> 32);
                    dlast = top == 0xFFFFFFFFu ? inf : tpu3_unmono(top);
                }
            }
        };
        (void)len;
#pragma unroll 2
        for (int j = 0; j < KT_TILE; j += 4) {
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 p = stage[wv][j + u];
                float dot = __builtin_fmaf(q0, p.x, 0.f);
                dot = __builtin_fmaf(q1, p.y, dot);
                dot = __builtin_fmaf(q2, p.z, dot);
                d[u] = __builtin_fmaf(-2.f, dot, rq) + p.w;
            }
            const float lo4 = fminf(fminf(d[0], d[1]), fminf(d[2], d[3]));
            if (lo4 <= dlast) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    enter(d[u], j + u);
            }
        }
        __builtin_amdgcn_wave_barrier();                 // (the next tile overwrites the stage)
    };
    auto rdl = [](float v, int l) __attribute__((always_inline)) {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
    };
    auto wave_bound = [&]() __attribute__((always_inline)) {
        // the largest K-th distance among the wave's live queries (+inf while some list is not full; for k < K the
        // K-th distance bounds the k-th from above: a looser bound, never a wrong one)
        return tpu3_wave_max_f32(live ? dlast : -inf);
    };

    // ---- the nearest tile first; then the tiles that overlap the wave's box (they hold most of the neighbours: the
    // bound is close to final after them); then whatever the bound still admits, the bound shrinking on the way
    float bestL = inf, farL = 0.f;
    int bestT = 0;
    for (int t0 = 0; t0 < ntile; t0 += 64) {
        const int t = t0 + lane;
        float L = inf, E = 0.f;
        float4 blo, bhi;
        if (t < ntile)
            tile_bound(t, L, E, blo, bhi);
        if (t < ntile)
            farL = fmaxf(farL, L - E);
        if (L < bestL) {
            bestL = L;
            bestT = t;
        }
    }
    {
        const float wmin = kt_wave_min_f32(bestL);
        const unsigned long long who = __ballot(bestL == wmin);
        bestT = __builtin_amdgcn_readlane(bestT, (int)__builtin_ctzll(who));
        farL = tpu3_wave_max_f32(farL);
    }
    search(bestT);
    float U = wave_bound();
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass)
        for (int t0 = 0; t0 < ntile; t0 += 64) {
            const int t = t0 + lane;
            float L = inf, E = 0.f;
            // pass 0: the tiles that touch the wave's box; pass 1: the others
            bool mine = t < ntile && t != bestT;         // (U may be +inf: the test below alone would pass anything)
            float4 blo = {0.f, 0.f, 0.f, 0.f}, bhi = {0.f, 0.f, 0.f, 0.f};
            if (mine)
                tile_bound(t, L, E, blo, bhi);
            mine = mine && ((L == 0.f) == (pass == 0));
            unsigned long long todo = __ballot(mine && L <= U + E);
            while (todo) {
                const int l = (int)__builtin_ctzll(todo);
                // the wave's box is mostly empty (64 consecutive points of a patch lie on a ring around its seed):
                // the tile is searched only if SOME query's own distance to the tile's box is within its own k-th
                ++ntest;
                const float lx = rdl(blo.x, l), ly = rdl(blo.y, l), lz = rdl(blo.z, l);
                const float hx = rdl(blo.w, l), hy = rdl(bhi.x, l), hz = rdl(bhi.y, l), rm = rdl(bhi.z, l);
                const float e0 = fmaxf(fmaxf(lx - q0, q0 - hx), 0.f), e1 = fmaxf(fmaxf(ly - q1, q1 - hy), 0.f);
                const float e2 = fmaxf(fmaxf(lz - q2, q2 - hz), 0.f);
                const float Lq = (e0 * e0 + e1 * e1 + e2 * e2) * (1.f - 0x1p-20f);
                if (__ballot(live && Lq <= dlast + 0x1p-20f * (rq + rm))) {
                    search(t0 + l);
                    U = wave_bound();
                }
                todo &= todo - 1;
                todo &= __ballot(mine && L <= U + E);
            }
        }
    if (a.dbg && lane == 0) {
        atomicAdd(a.dbg + 0, 1u);
        atomicAdd(a.dbg + 1, (unsigned)nsearch);
        atomicAdd(a.dbg + 2, (unsigned)ntest);
        atomicAdd(a.dbg + 3, (unsigned)ntile);
    }
    if (!live)
        return;
    const size_t o = ((size_t)b * a.m + qi) * a.k;
    // A list that is not full after the search (a query with a NaN coordinate fails every `lo4 <= dlast` test; a set
    // with fewer than k candidates fills up with the last tile's dead slots, row -1 at D = +inf) must never reach the
    // gathers behind this kernel as row -1: the exact path redoes the call, and the row written here stays in range.
    bool hole = false;
#pragma unroll
    for (int i = 0; i < K; ++i)
        hole |= i < a.k && (int)(uint32_t)kb[i] < 0;
    if (hole)
        a.uws[1] = 1u;
#pragma unroll
    for (int i = 0; i < K; ++i)
        if (i < a.k) {
            const int row = max((int)(uint32_t)kb[i], 0);
            if (a.idx64)
                ((int64_t *)a.idx)[o + i] = (int64_t)row;
            else
                ((int32_t *)a.idx)[o + i] = row;
            if (a.dist)
                a.dist[o + i] = tpu3_unmono((uint32_t)(kb[i] >> 32));
        }
    if (anydup && count < nlive) {
        // the duplicates' D' = D + max(D) >= (nearest distance) + (a lower bound of max D: the k-th distance itself --
        // its candidate was visited -- or the bound of the farthest tile): they stay behind the k-th
        unsigned long long kk = kb[0];
#pragma unroll
        for (int i = 1; i < K; ++i)
            kk = i == a.k - 1 ? kb[i] : kk;
        const float tk = tpu3_unmono((uint32_t)(kk >> 32)), dupmin = tpu3_unmono((uint32_t)(kb[0] >> 32));
        const float dq = fmaxf(tk, farL);
        if (!(tk < dupmin + dq))
            a.uws[1] = 1u;
    }
}

inline size_t kt_align(size_t v) { return (v + 255) & ~(size_t)255; }

} // namespace

extern "C" size_t tpu3_knn_tiles_workspace_bytes(int bp, int n)
{
    if (bp <= 0 || n <= 0) return 0;
    return kt_align((size_t)bp * n * sizeof(int32_t));
}

extern "C" int tpu3_knn_tiles_build_f32(tpu3_stream_t stream, int bp, int n, const float *points, const int32_t *n_arr,
                                        const int32_t *cand, const int32_t *cand_count, const uint32_t *uws,
                                        float *tile_pts, int32_t *tile_idx, float *tile_box, void *workspace,
                                        size_t workspace_bytes)
{
    if (bp < 0 || n < 0) return TPU3_EINVAL;
    if (bp == 0 || n == 0) return TPU3_OK;
    if (!points || !cand || !cand_count || !uws || !tile_pts || !tile_idx || !tile_box) return TPU3_EINVAL;
    if (bp > 65535 || (size_t)bp * n > 0x7FFFFFFFu) return TPU3_ELIMIT;
    if (!workspace || workspace_bytes < tpu3_knn_tiles_workspace_bytes(bp, n)) return TPU3_EINVAL;
    if ((((uintptr_t)tile_pts | (uintptr_t)tile_box) & 15) != 0) return TPU3_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int tiles = (n + KT_TILE - 1) / KT_TILE;
    KtBuildArgs a{bp, n, tiles, points, n_arr, cand, cand_count, uws, nullptr, nullptr, (int32_t *)workspace,
                  (float4 *)tile_pts, tile_idx, tile_box};
    hipLaunchKernelGGL(kt_bin_kernel, dim3(bp), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(kt_fill_kernel, dim3((unsigned)(((long)bp * tiles + 3) / 4)), dim3(256), 0, s, a,
                       (const int32_t *)workspace);
    return tpu3_launch_status();
}

// development probe (tools/knn_tiles_probe.py; not part of include/tpu3.h): the NEXT pruned search adds (waves, tiles
// searched, tiles tested query by query, tiles per set) to four device words.  One-shot; host-side state only.
static unsigned *g_kt_dbg = nullptr;
extern "C" int tpu3_debug_knn_tiles_stats(unsigned *words)
{
    g_kt_dbg = words;
    return TPU3_OK;
}

// the optimistic pass of tpu3_knn_f32 (c = 3, k <= 8) on the tiles of tpu3_knn_tiles_build_f32; called from there
extern "C" int tpu3_knn_tiles_query_f32(tpu3_stream_t stream, int b, int m, int n, int k, const float *query,
                                        const tpu3_knn_layout *layout, uint32_t *uws, void *idx, int idx_elem_size,
                                        float *dist)
{
    if (b < 0 || m < 0 || n <= 0 || k <= 0 || k > 8) return TPU3_EINVAL;
    if (!layout || !layout->tile_pts || !layout->tile_idx || !layout->tile_box || !layout->cand_count) return TPU3_EINVAL;
    if (idx_elem_size != 4 && idx_elem_size != 8) return TPU3_EINVAL;
    if (b == 0 || m == 0) return TPU3_OK;
    if (!query || !uws || !idx) return TPU3_EINVAL;
    const int tiles = (n + KT_TILE - 1) / KT_TILE;
    KtQueryArgs a{b, m, n, k, tiles, query, layout->m_arr, layout->n_arr, layout->pts_of, layout->cand_count, uws,
                  (const float4 *)layout->tile_pts, layout->tile_idx, layout->tile_box, idx, idx_elem_size == 8, dist,
                  g_kt_dbg};
    g_kt_dbg = nullptr;
    const long waves = (long)b * ((m + 63) / 64);
    const long blocks = (waves + 3) / 4;
    if (blocks > 0x7FFFFFFF) return TPU3_ELIMIT;
    hipStream_t s = (hipStream_t)stream;
    // query sets of 65 .. 512 points (the patches of a Level): a workgroup per set, its queries in Morton order
    // (TPU3_KNN_TILES_SORT=0: the waves take the queries in the order given, as before round 5)
    static const bool sort_on = !(getenv("TPU3_KNN_TILES_SORT") && atoi(getenv("TPU3_KNN_TILES_SORT")) == 0);
    if (sort_on && m > 64 && m <= KT_QMAX && b <= 0x7FFFFFFF) {
        const int threads = ((m + 63) / 64) * 64;
        if (k <= 5)
            hipLaunchKernelGGL((kt_query_kernel<5, true>), dim3((unsigned)b), dim3(threads), 0, s, a);
        else
            hipLaunchKernelGGL((kt_query_kernel<8, true>), dim3((unsigned)b), dim3(threads), 0, s, a);
        return tpu3_launch_status();
    }
    if (k <= 5)
        hipLaunchKernelGGL((kt_query_kernel<5, false>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((kt_query_kernel<8, false>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    return tpu3_launch_status();
}
