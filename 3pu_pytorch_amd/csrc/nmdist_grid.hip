// nmdist_grid.hip -- Chamfer "nm-distance" forward for LARGE clouds, pruned by space, same bits as the scan (gfx950).
//
// losses.nmdistance_forward (reference losses/nmdistance_cuda.cu:11-153) is a brute-force scan: every point of one
// set against every point of the other, 16 FLOP per pair.  csrc/nmdistance.hip does exactly that at the VALU rate the
// instruction mix allows (~0.3 of the packed-FMA peak) -- 2.7 ms for the evaluation metric's 80 000 x 80 000, ~0.5 s at
// config C5's 1.28 M x 1.28 M.  The RESULT, though, is a function of the two point sets only:
//
//     dist[j] = min_k d(j, k),   idx[j] = the smallest k attaining it        (strict '<' in scan order, :36, :125)
//     d(j, k) = fma(dz, dz, fma(dx, dx, dy * dy)),  (dx, dy, dz) = p_k - q_j (nvcc -fmad=true)
//
// so any search that (i) evaluates d with this very expression, (ii) compares (d, k) pairs lexicographically and
// (iii) skips a candidate only when a LOWER BOUND of its computed d is STRICTLY above the query's current best returns
// the same bits -- distance and index, exact ties included.  This file is that search:
//
//   build (both sets of a batch element on ONE grid over their common bounding box, G^3 cells in Morton order):
//       nmg_bbox -> nmg_hist (cell + arrival rank per point) -> nmg_scan (block-local exclusive scan) ->
//       nmg_scatter ((x, y, z, original index) rows in cell order) -> nmg_tilebox (TILES of 64 consecutive rows
//       with their boxes, SUPER-TILES of 64 tiles with theirs);
//   query (nmg_query): a WAVE per tile of the query set -- 64 neighbouring points, lane = point.  Box-to-box lower
//       bounds against the super-tiles, then against the tiles of the surviving ones, are compared with U = the largest
//       current best of the wave; a surviving tile is tested per lane (point-to-box bound against the lane's own best)
//       and searched -- 64 candidates streamed through scalar loads, 7 VALU instructions per pair -- only if some lane
//       may still improve.  The tile with the smallest bound goes first, so U is tight from the start.
//
// Why the bounds are safe in fp32: a box [lo, hi] holds actual coordinates.  For p >= lo and q <= hi',
// fl(p - q) >= fl(lo - hi') because rounding is monotone; likewise on the other side; so every computed |dx| is >= the
// computed gap g_x >= 0, and fma(g_z, g_z, fma(g_x, g_x, g_y * g_y)) <= the computed d because each operation is
// monotone in its non-negative arguments.  A bound that is NaN (Inf - Inf) never compares "above": the tile is searched.
// Non-finite inputs: a candidate whose distance is NaN is never selected by the reference unless it is candidate 0
// (`k == 0 ||`), in which case NaN stays for good -- reproduced by the fix-up at the end of the query kernel.
//
// The order of the rows inside a cell is the arrival order of the atomics and differs from run to run; the result does
// not depend on it.  Cost model: on a surface-like cloud a wave searches ~10-20 of the other set's tiles instead of all
// of them -- O(n) work, where the scan is O(n m).
#include "tpu3_dev.h"

#include <cstdlib>

namespace {

constexpr int NMG_TILE = 64;            // rows per tile = lanes per wave
constexpr int NMG_SUPER = 64;           // tiles per super-tile
constexpr int NMG_SCAN = 1024;          // cells per block of the scan
constexpr int NMG_MAX_PARTS = 2048;     // scan blocks per set: G <= 128

struct NmgArgs {
    int b, n, m;                        // batch elements, |set 0| = n (xyz1), |set 1| = m (xyz2)
    int G, cells, parts;                // cells per axis, G^3, cells / NMG_SCAN
    int pmax, tmax, smax;               // max(n, m), tiles and super-tiles per set (of the larger set)
    const float *xyz[2];                // (b, n, 3), (b, m, 3)
    float *dist[2];                     // (b, n), (b, m)
    int32_t *idx[2];
    uint32_t *bbox;                     // (b, 8)          mono-max encoded: [0..2] = ~mono(lo), [3..5] = mono(hi)
    int32_t *hist;                      // (2b, cells)     counts, then block-local exclusive offsets
    int32_t *part;                      // (2b, parts)     per scan block: total
    int2 *cellrank;                     // (2b, pmax)      cell, arrival rank inside the cell
    float4 *rows;                       // (2b, tmax * 64) x, y, z, original index (bits); pads are NaN
    float *tbox;                        // (2b, tmax, 8)   lo xyz, hi xyz, -, -
    uint32_t *sbox;                     // (2b, smax, 8)   encoded like bbox
    unsigned long long *stats;          // development probe (tpu3_debug_nmdist_grid_stats) or null: waves, super-tiles
                                        // and tiles that passed the wave's bound, tiles searched
};

__device__ __forceinline__ int nmg_count(const NmgArgs &a, int set) { return (set & 1) ? a.m : a.n; }

__device__ __forceinline__ const float *nmg_points(const NmgArgs &a, int set)
{
    const int e = set >> 1;
    return (set & 1) ? a.xyz[1] + (size_t)e * a.m * 3 : a.xyz[0] + (size_t)e * a.n * 3;
}

__device__ __forceinline__ uint32_t nmg_spread(uint32_t v)       // bit i of a <= 10-bit value -> bit 3 i
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// Morton cell of a point on the element's grid (clamped; NaN coordinates land in cell 0 of their axis)
__device__ __forceinline__ int nmg_cell(const NmgArgs &a, int e, float x, float y, float z)
{
    const uint32_t *bb = a.bbox + (size_t)e * 8;
    const float p[3] = {x, y, z};
    uint32_t code = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float lo = tpu3_unmono(~bb[c]), hi = tpu3_unmono(bb[3 + c]);
        const float ext = hi - lo;
        const float sc = ext > 0.f ? (float)a.G / ext : 0.f;
        float f = (p[c] - lo) * sc;
        f = fminf(fmaxf(f, 0.f), (float)(a.G - 1));
        code |= nmg_spread((uint32_t)(int)f) << c;
    }
    return (int)code;
}

// ---- build ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nmg_bbox_kernel(NmgArgs a)
{
    __shared__ uint32_t red[4][6];
    const int set = blockIdx.y, e = set >> 1;
    const int cnt = nmg_count(a, set);
    const float *P = nmg_points(a, set);
    const float inf = __builtin_inff();
    float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    for (int j = blockIdx.x * 256 + threadIdx.x; j < cnt; j += gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = P[(size_t)j * 3 + c];
            lo[c] = fminf(lo[c], v);
            hi[c] = fmaxf(hi[c], v);
        }
    }
    // (one atomic per word and WORKGROUP, few workgroups: 6144 atomics on six addresses took 140 us at 80 000 points)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t l = tpu3_wave_max_u32(~tpu3_mono(lo[c])), h = tpu3_wave_max_u32(tpu3_mono(hi[c]));
        if ((threadIdx.x & 63) == 0) {
            red[threadIdx.x >> 6][c] = l;
            red[threadIdx.x >> 6][3 + c] = h;
        }
    }
    __syncthreads();
    if (threadIdx.x < 6)
        atomicMax(a.bbox + (size_t)e * 8 + threadIdx.x,
                  max(max(red[0][threadIdx.x], red[1][threadIdx.x]), max(red[2][threadIdx.x], red[3][threadIdx.x])));
}

__global__ __launch_bounds__(256) void nmg_hist_kernel(NmgArgs a)
{
    const int set = blockIdx.y, e = set >> 1;
    const int cnt = nmg_count(a, set);
    const float *P = nmg_points(a, set);
    int32_t *H = a.hist + (size_t)set * a.cells;
    int2 *CR = a.cellrank + (size_t)set * a.pmax;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < cnt; j += gridDim.x * 256) {
        const int cell = nmg_cell(a, e, P[(size_t)j * 3], P[(size_t)j * 3 + 1], P[(size_t)j * 3 + 2]);
        CR[j] = make_int2(cell, atomicAdd(H + cell, 1));
    }
}

// exclusive scan of a block of NMG_SCAN counters in place; the block's total goes to part[]
__global__ __launch_bounds__(NMG_SCAN) void nmg_scan_kernel(NmgArgs a)
{
    __shared__ int wsum[NMG_SCAN / 64];
    const int set = blockIdx.y, blk = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t *H = a.hist + (size_t)set * a.cells + (size_t)blk * NMG_SCAN;
    const int v = H[tid];
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        inc += lane >= d ? o : 0;
    }
    if (lane == 63)
        wsum[wave] = inc;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NMG_SCAN / 64; ++w) {
        const int s = wsum[w];
        base += w < wave ? s : 0;
        total += s;
    }
    H[tid] = base + inc - v;
    if (tid == 0)
        a.part[(size_t)set * a.parts + blk] = total;
}

// rows in cell order: position = (rows of earlier scan blocks) + (block-local offset of the cell) + arrival rank
__global__ __launch_bounds__(256) void nmg_scatter_kernel(NmgArgs a)
{
    constexpr int PPT = NMG_MAX_PARTS / 256;   // parts per thread of the prefix
    __shared__ int pre[NMG_MAX_PARTS];  // exclusive prefix of part[]
    __shared__ int wtot[4];
    const int set = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        int v[PPT], sum = 0;
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            v[u] = tid * PPT + u < a.parts ? a.part[(size_t)set * a.parts + tid * PPT + u] : 0;
            sum += v[u];
        }
        int inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            inc += lane >= d ? o : 0;
        }
        if (lane == 63)
            wtot[wave] = inc;
        __syncthreads();
        int off = inc - sum;
        for (int w = 0; w < wave; ++w)
            off += wtot[w];
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            pre[tid * PPT + u] = off;
            off += v[u];
        }
        __syncthreads();
    }
    const int cnt = nmg_count(a, set);
    const float *P = nmg_points(a, set);
    const int32_t *H = a.hist + (size_t)set * a.cells;
    const int2 *CR = a.cellrank + (size_t)set * a.pmax;
    float4 *R = a.rows + (size_t)set * a.tmax * NMG_TILE;
    for (int j = blockIdx.x * 256 + tid; j < cnt; j += gridDim.x * 256) {
        const int2 cr = CR[j];
        const int pos = pre[cr.x / NMG_SCAN] + H[cr.x] + cr.y;
        R[pos] = make_float4(P[(size_t)j * 3], P[(size_t)j * 3 + 1], P[(size_t)j * 3 + 2], __int_as_float(j));
    }
}

// a wave per tile: NaN pads behind the set's last row, the tile's box, its share of the super-tile's box
__global__ __launch_bounds__(256) void nmg_tilebox_kernel(NmgArgs a)
{
    const int set = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int cnt = nmg_count(a, set);
    if (t * NMG_TILE >= cnt)
        return;
    float4 *R = a.rows + (size_t)set * a.tmax * NMG_TILE;
    const int i = t * NMG_TILE + lane;
    const bool live = i < cnt;
    const float inf = __builtin_inff(), nan = __builtin_nanf("");
    float4 p = make_float4(nan, nan, nan, __int_as_float(0x7FFFFFFF));
    if (live)
        p = R[i];
    else
        R[i] = p;
    // (NaN coordinates of live rows do not enter the box: fminf / fmaxf drop them, and such a row is never selected)
    const uint32_t lx = tpu3_wave_max_u32(~tpu3_mono(live ? fminf(p.x, inf) : inf));
    const uint32_t ly = tpu3_wave_max_u32(~tpu3_mono(live ? fminf(p.y, inf) : inf));
    const uint32_t lz = tpu3_wave_max_u32(~tpu3_mono(live ? fminf(p.z, inf) : inf));
    const uint32_t hx = tpu3_wave_max_u32(tpu3_mono(live ? fmaxf(p.x, -inf) : -inf));
    const uint32_t hy = tpu3_wave_max_u32(tpu3_mono(live ? fmaxf(p.y, -inf) : -inf));
    const uint32_t hz = tpu3_wave_max_u32(tpu3_mono(live ? fmaxf(p.z, -inf) : -inf));
    if (lane == 0) {
        float *tb = a.tbox + ((size_t)set * a.tmax + t) * 8;
        tb[0] = tpu3_unmono(~lx); tb[1] = tpu3_unmono(~ly); tb[2] = tpu3_unmono(~lz);
        tb[3] = tpu3_unmono(hx); tb[4] = tpu3_unmono(hy); tb[5] = tpu3_unmono(hz);
        tb[6] = 0.f; tb[7] = 0.f;
        uint32_t *sb = a.sbox + ((size_t)set * a.smax + t / NMG_SUPER) * 8;
        atomicMax(sb + 0, lx); atomicMax(sb + 1, ly); atomicMax(sb + 2, lz);
        atomicMax(sb + 3, hx); atomicMax(sb + 4, hy); atomicMax(sb + 5, hz);
    }
}

// ---- query ---------------------------------------------------------------------------------------------------------
struct NmgBox { float lx, ly, lz, hx, hy, hz; };

// lower bound of every computed distance between a point of box q and a point of box t (see the header)
__device__ __forceinline__ float nmg_box_box(const NmgBox &q, const NmgBox &t)
{
    const float gx = fmaxf(fmaxf(t.lx - q.hx, q.lx - t.hx), 0.f);
    const float gy = fmaxf(fmaxf(t.ly - q.hy, q.ly - t.hy), 0.f);
    const float gz = fmaxf(fmaxf(t.lz - q.hz, q.lz - t.hz), 0.f);
    return tpu3_sqdist3(gx, gy, gz);
}

__device__ __forceinline__ float nmg_point_box(float x, float y, float z, const NmgBox &t)
{
    const float gx = fmaxf(fmaxf(t.lx - x, x - t.hx), 0.f);
    const float gy = fmaxf(fmaxf(t.ly - y, y - t.hy), 0.f);
    const float gz = fmaxf(fmaxf(t.lz - z, z - t.hz), 0.f);
    return tpu3_sqdist3(gx, gy, gz);
}

// (d, index) of the lane's point against the 64 rows of one tile, folded into (best, besti) lexicographically.
// The tile's rows sit in the wave's own 1 KiB of LDS (a wave writes and reads only its own slice, in program order: no
// barrier); every ds_read_b128 is a broadcast.  A candidate costs 3 subtractions and the mul / fma / fma of the
// reference's expression; eight candidates share one minimum tree, one compare and one branch -- the lexicographic
// update of the eight runs only when some lane may take one of them.
// (First version: the rows through SCALAR loads straight from memory -- zero vector traffic, but eight dependent
// round trips per tile with nothing to hide them behind at 2.4 waves per SIMD: 7.6 us per tile, 671 us per 80 000^2.)
// (r6b) PACKED: the wave's LDS slice holds the tile as candidate PAIRS -- eight words per pair: x0 x1 y0 y1 z0 z1 i0 i1 --
// so that two ds_read_b128 deliver register pairs v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 can take as they are: the
// three subtractions and the mul / fma / fma of the reference's expression run on TWO candidates per instruction (each
// half is the IEEE operation of the scalar instruction: same bits), 3 + 0.5 instructions per pair instead of 6.6.
typedef float nmg_f2 __attribute__((ext_vector_type(2)));
typedef float nmg_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void nmg_search_tile(const nmg_f4 *pairs, float x, float y, float z, float &best, int &besti)
{
    const nmg_f2 xx = {x, x}, yy = {y, y}, zz = {z, z};
#pragma unroll 2
    for (int j = 0; j < NMG_TILE / 2; j += 4) {          // four pairs = eight candidates per step
        nmg_f4 xy[4], zi[4];
        nmg_f2 d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xy[u] = pairs[2 * (j + u)];
            zi[u] = pairs[2 * (j + u) + 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const nmg_f2 dx = xy[u].xy - xx, dy = xy[u].zw - yy, dz = zi[u].xy - zz;
            d[u] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
        }
        const float m8 = __builtin_fminf(__builtin_fminf(__builtin_fminf(d[0].x, d[0].y), __builtin_fminf(d[1].x, d[1].y)),
                                         __builtin_fminf(__builtin_fminf(d[2].x, d[2].y), __builtin_fminf(d[3].x, d[3].y)));
        if (__builtin_amdgcn_ballot_w64(m8 <= best)) {             // (NaN distances: dropped by fminf, never taken)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float dv = h ? d[u].y : d[u].x;
                    const int k = __float_as_int(h ? zi[u].w : zi[u].z);
                    const bool take = (dv < best) | ((dv == best) & (k < besti));
                    best = take ? dv : best;
                    besti = take ? k : besti;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void nmg_query_kernel(NmgArgs a)
{
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.y >> 1, dir = blockIdx.y & 1;
    const int sq = 2 * e + dir, sc = 2 * e + (dir ^ 1);          // query set, candidate set
    const int nq = nmg_count(a, sq), nc = nmg_count(a, sc);
    const int qt = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (qt * NMG_TILE >= nq)
        return;
    const bool live = qt * NMG_TILE + lane < nq;
    const float4 me = a.rows[((size_t)sq * a.tmax + qt) * NMG_TILE + lane];
    const float4 *__restrict__ CR = a.rows + (size_t)sc * a.tmax * NMG_TILE;
    const float *__restrict__ CB = a.tbox + (size_t)sc * a.tmax * 8;
    const uint32_t *__restrict__ SB = a.sbox + (size_t)sc * a.smax * 8;
    const int tiles = (nc + NMG_TILE - 1) / NMG_TILE, supers = (tiles + NMG_SUPER - 1) / NMG_SUPER;
    NmgBox qb;
    {
        const float *b = a.tbox + ((size_t)sq * a.tmax + qt) * 8;
        qb.lx = b[0]; qb.ly = b[1]; qb.lz = b[2]; qb.hx = b[3]; qb.hy = b[4]; qb.hz = b[5];
    }
    const float inf = __builtin_inff();
    auto super_bound = [&](int s) __attribute__((always_inline)) {
        if (s >= supers)
            return inf;
        const uint32_t *b = SB + (size_t)s * 8;
        NmgBox t;
        t.lx = tpu3_unmono(~b[0]); t.ly = tpu3_unmono(~b[1]); t.lz = tpu3_unmono(~b[2]);
        t.hx = tpu3_unmono(b[3]); t.hy = tpu3_unmono(b[4]); t.hz = tpu3_unmono(b[5]);
        return nmg_box_box(qb, t);
    };
    auto tile_bound = [&](int t) __attribute__((always_inline)) {
        if (t >= tiles)
            return inf;
        const float *b = CB + (size_t)t * 8;
        NmgBox bx;
        bx.lx = b[0]; bx.ly = b[1]; bx.lz = b[2]; bx.hx = b[3]; bx.hy = b[4]; bx.hz = b[5];
        return nmg_box_box(qb, bx);
    };
    float best = inf;
    int besti = 0x7FFFFFFF;
    uint32_t U = 0x7F800000u;                                   // bits of the wave's largest best (distances are >= 0)
    int st_super = 0, st_tile = 0, st_search = 0;
    __shared__ float stage[4][NMG_TILE * 4];
    float *mine = stage[threadIdx.x >> 6];
    int pre_t = -1;                                             // the tile whose rows are already on their way
    float4 pre_row = make_float4(0.f, 0.f, 0.f, 0.f);
    // search tile t; `next` (>= 0): the tile most likely to be searched after it -- its rows are requested now
    auto search = [&](int t, int next) __attribute__((always_inline)) {
        ++st_search;
        const float4 row = pre_t == t ? pre_row : CR[(size_t)t * NMG_TILE + lane];
        pre_t = next;
        if (next >= 0)
            pre_row = CR[(size_t)next * NMG_TILE + lane];
        {   // candidate `lane` of the tile into its half of pair lane / 2
            float *w = mine + (lane >> 1) * 8 + (lane & 1);
            w[0] = row.x; w[2] = row.y; w[4] = row.z; w[6] = row.w;
        }
        nmg_search_tile((const nmg_f4 *)mine, me.x, me.y, me.z, best, besti);
        U = tpu3_wave_max_u32(live ? __float_as_uint(best) : 0u);
    };
    // ---- seed: the tile with the smallest bound (any of them), so that U is tight before the sweep ----
    int seed;
    {
        uint32_t bs = 0xFFFFFFFFu;
        for (int s0 = 0; s0 < supers; s0 += 64) {
            const float lb = super_bound(s0 + lane);
            // key = (bound bits, super index): bounds are >= 0 or NaN (NaN: sorts last, still a valid choice)
            const uint32_t hi = __float_as_uint(lb) >> 12;
            bs = min(bs, s0 + lane < supers ? (hi << 12) | (uint32_t)min(s0 + lane, 4095) : 0xFFFFFFFFu);
        }
        // (supers <= 4096 is guaranteed by the launcher: 16.7 M rows per set)
        const int s = (int)(tpu3_wave_min_u32(bs) & 4095u);
        const float lb = tile_bound(s * NMG_SUPER + lane);
        const uint32_t key = s * NMG_SUPER + lane < tiles ? ((__float_as_uint(lb) >> 6) << 6) | (uint32_t)lane : 0xFFFFFFFFu;
        seed = s * NMG_SUPER + (int)(tpu3_wave_min_u32(key) & 63u);
        search(seed, -1);
    }
    // ---- sweep: super-tiles -> tiles -> lanes, in two passes ----
    // pass 0: the tiles whose box OVERLAPS the wave's box (bound exactly 0) -- after it every lane has seen its own
    // surroundings and its best is close to final; pass 1: the others, against a U that is tight by then.  (One pass in
    // index order searched 59 tiles per wave at 1.28 M points: the early tiles of the order are judged against the
    // seed tile's distances only.)  A NaN bound belongs to pass 1 and is never "above".
    auto lane_value = [](float v, int l) __attribute__((always_inline)) {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
    };
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll 1
        for (int s0 = 0; s0 < supers; s0 += 64) {
            const float lbs = super_bound(s0 + lane);
            // (the index tests matter: with U = Inf -- a lane whose distances are all NaN or Inf -- nothing is "above")
            const bool s_now = pass == 0 ? lbs <= 0.f : !(lbs > __uint_as_float(U));
            uint64_t smask = __builtin_amdgcn_ballot_w64(s0 + lane < supers && s_now);
            while (smask) {
                const int sb = __builtin_ctzll(smask);
                smask &= smask - 1;
                const int s = s0 + sb;
                ++st_super;
                // the lane's tile of this super-tile: its box stays in registers, the per-tile tests below read it
                // from the lane (v_readlane) instead of from memory
                const int tl = s * NMG_SUPER + lane;
                NmgBox tbx = {inf, inf, inf, -inf, -inf, -inf};
                if (tl < tiles) {
                    const float4 lo4 = *(const float4 *)(CB + (size_t)tl * 8);
                    const float2 hi2 = *(const float2 *)(CB + (size_t)tl * 8 + 4);
                    tbx.lx = lo4.x; tbx.ly = lo4.y; tbx.lz = lo4.z; tbx.hx = lo4.w; tbx.hy = hi2.x; tbx.hz = hi2.y;
                }
                const float lbt = nmg_box_box(qb, tbx);
                const bool t_now = pass == 0 ? lbt <= 0.f : (!(lbt <= 0.f) && !(lbt > __uint_as_float(U)));
                uint64_t tmask = __builtin_amdgcn_ballot_w64(tl < tiles && tl != seed && t_now);
                while (tmask) {
                    const int tb = __builtin_ctzll(tmask);
                    tmask &= tmask - 1;
                    const int t = s * NMG_SUPER + tb;
                    // U has shrunk since the ballot?  then the tile's own bound decides again, for the whole wave
                    if (lane_value(lbt, tb) > __uint_as_float(U))
                        continue;
                    ++st_tile;
                    NmgBox bx;
                    bx.lx = lane_value(tbx.lx, tb); bx.ly = lane_value(tbx.ly, tb); bx.lz = lane_value(tbx.lz, tb);
                    bx.hx = lane_value(tbx.hx, tb); bx.hy = lane_value(tbx.hy, tb); bx.hz = lane_value(tbx.hz, tb);
                    const float lb = nmg_point_box(me.x, me.y, me.z, bx);
                    if (!__builtin_amdgcn_ballot_w64(live && !(lb > best)))
                        continue;
                    // the next surviving tile of this super-tile is the likely successor: its rows are requested now
                    search(t, tmask ? s * NMG_SUPER + __builtin_ctzll(tmask) : -1);
                }
            }
        }
    }
    // ---- the reference's `k == 0 ||`: a NaN distance to candidate 0 is kept for good (nmdistance_cuda.cu:36) ----
    {
        const float *p0 = nmg_points(a, sc);
        const float d0 = tpu3_sqdist3(p0[0] - me.x, p0[1] - me.y, p0[2] - me.z);
        if (d0 != d0) {
            best = d0;
            besti = 0;
        }
    }
    if (a.stats && lane == 0) {
        atomicAdd(a.stats + 0, 1ull);
        atomicAdd(a.stats + 1, (unsigned long long)st_super);
        atomicAdd(a.stats + 2, (unsigned long long)st_tile);
        atomicAdd(a.stats + 3, (unsigned long long)st_search);
    }
    if (live) {
        const int j = __float_as_int(me.w);
        a.dist[dir][(size_t)e * nq + j] = best;
        a.idx[dir][(size_t)e * nq + j] = besti;
    }
}

unsigned long long *g_nmdist_stats = nullptr;
int g_nmdist_form = -1;                 // tpu3_debug_nmdist_form: -1 automatic, 0 scan, 1 grid
long g_nmdist_grid_calls = 0;

size_t nmg_align(size_t v) { return (v + 255) & ~(size_t)255; }

} // namespace

extern "C" int tpu3_debug_nmdist_form(int form)
{
    const int old = g_nmdist_form;
    if (form >= -1 && form <= 1)
        g_nmdist_form = form;
    return old;
}

extern "C" int tpu3_debug_nmdist_grid_stats(unsigned long long *stats)
{
    g_nmdist_stats = stats;
    return TPU3_OK;
}

extern "C" long tpu3_debug_nmdist_grid_calls(int reset)
{
    const long v = g_nmdist_grid_calls;
    if (reset)
        g_nmdist_grid_calls = 0;
    return v;
}

// Does a (b, n, m) forward call take the grid form?  Automatic: both sets large enough for tiles to mean something and
// enough pairs for the build (six small launches) to pay -- tools/chamfer_probe.py has the crossover.
bool tpu3_nmdist_takes_grid(int b, int n, int m)
{
    if (n < 1 || m < 1 || b < 1 || (long)b * 2 > 65535)
        return false;
    if ((long)max(n, m) > 64L * 64 * 4096)      // super-tile index bits of the seed key
        return false;
    if (g_nmdist_form >= 0)
        return g_nmdist_form == 1 && n >= 2 * NMG_TILE && m >= 2 * NMG_TILE;
    static const long min_pairs = getenv("TPU3_NMDIST_GRID_MIN_PAIRS") ? atol(getenv("TPU3_NMDIST_GRID_MIN_PAIRS")) : 16000000L;
    return n >= 2048 && m >= 2048 && (long)n * m >= min_pairs;
}

int tpu3_nmdist_grid_forward(hipStream_t s, int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1,
                             float *dist2, int32_t *idx1, int32_t *idx2)
{
    NmgArgs a;
    a.b = b; a.n = n; a.m = m;
    a.pmax = max(n, m);
    // cells per axis: ~n / 24 occupied cells on a surface-like cloud (3 G^2 of the G^3), i.e. a handful of rows per cell
    a.G = a.pmax >= 400000 ? 128 : (a.pmax >= 32768 ? 64 : (a.pmax >= 4096 ? 32 : 16));
    a.cells = a.G * a.G * a.G;
    a.parts = a.cells / NMG_SCAN;
    a.tmax = (a.pmax + NMG_TILE - 1) / NMG_TILE;
    a.smax = (a.tmax + NMG_SUPER - 1) / NMG_SUPER;
    a.xyz[0] = xyz1; a.xyz[1] = xyz2;
    a.dist[0] = dist1; a.dist[1] = dist2;
    a.idx[0] = idx1; a.idx[1] = idx2;
    const size_t sets = (size_t)2 * b;
    // zero-initialised head: bbox | sbox | hist   (the mono-max encoding makes 0 the neutral element of every box)
    const size_t o_bbox = 0;
    const size_t o_sbox = nmg_align(o_bbox + (size_t)b * 8 * 4);
    const size_t o_hist = nmg_align(o_sbox + sets * a.smax * 8 * 4);
    const size_t zero_bytes = nmg_align(o_hist + sets * a.cells * 4);
    const size_t o_part = zero_bytes;
    const size_t o_cr = nmg_align(o_part + sets * a.parts * 4);
    const size_t o_rows = nmg_align(o_cr + sets * a.pmax * 8);
    const size_t o_tbox = nmg_align(o_rows + sets * a.tmax * NMG_TILE * 16);
    const size_t total = nmg_align(o_tbox + sets * a.tmax * 8 * 4);
    char *ws = nullptr;
    hipError_t err = hipMallocAsync((void **)&ws, total, s);
    if (err != hipSuccess) return (int)err;
    err = hipMemsetAsync(ws, 0, zero_bytes, s);
    if (err != hipSuccess) { (void)hipFreeAsync(ws, s); return (int)err; }
    a.bbox = (uint32_t *)(ws + o_bbox);
    a.sbox = (uint32_t *)(ws + o_sbox);
    a.hist = (int32_t *)(ws + o_hist);
    a.part = (int32_t *)(ws + o_part);
    a.cellrank = (int2 *)(ws + o_cr);
    a.rows = (float4 *)(ws + o_rows);
    a.tbox = (float *)(ws + o_tbox);
    a.stats = g_nmdist_stats;
    const unsigned pblocks = (unsigned)min((a.pmax + 255) / 256, 2048);
    hipLaunchKernelGGL(nmg_bbox_kernel, dim3(min(pblocks, 64u), (unsigned)sets), dim3(256), 0, s, a);
    hipLaunchKernelGGL(nmg_hist_kernel, dim3(pblocks, (unsigned)sets), dim3(256), 0, s, a);
    hipLaunchKernelGGL(nmg_scan_kernel, dim3((unsigned)a.parts, (unsigned)sets), dim3(NMG_SCAN), 0, s, a);
    hipLaunchKernelGGL(nmg_scatter_kernel, dim3(pblocks, (unsigned)sets), dim3(256), 0, s, a);
    hipLaunchKernelGGL(nmg_tilebox_kernel, dim3((unsigned)((a.tmax + 3) / 4), (unsigned)sets), dim3(256), 0, s, a);
    hipLaunchKernelGGL(nmg_query_kernel, dim3((unsigned)((a.tmax + 3) / 4), (unsigned)sets), dim3(256), 0, s, a);
    ++g_nmdist_grid_calls;
    int r = tpu3_launch_status();
    err = hipFreeAsync(ws, s);
    return r ? r : (int)err;
}
