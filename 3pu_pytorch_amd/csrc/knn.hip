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
#include "knn_sortnet.h"

namespace {

struct KnnArgs {
    int m, n, c, k;
    int b, nblk_x;           // batch elements; query blocks per batch element (strided kernels)
    const float *query;      // (b,m,c)
    const float *points;     // (b,n,c)
    const int32_t *n_arr;    // (bp) live points per point set, or null
    const int32_t *m_arr;    // (b) live queries per query set, or null
    const int32_t *pts_of;   // (b) point set of each query set, or null (identity)
    const int32_t *grp;      // (b) unique-max group, or null (one group)
    const uint8_t *dup;      // (bp,n) or null
    uint32_t *uws;           // [0] any-dup, [1] optimistic pass failed, [4+g] mono(max D) of group g
    int mode;                // 0: D += max(D)*dup (reference arithmetic); 1: optimistic (see below)
    int gate;                // run only if uws[gate] != 0 (0 = always)
    void *idx;               // (b,m,k) i32 / i64
    int idx64;
    float *dist;             // (b,m,k) or null
    const int32_t *cand;     // (bp,n) ascending indices of the first occurrences, or null
    const int32_t *cand_count;   // (bp)
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
    if (C % 4 == 0 && c == C && ((uintptr_t)src & 15) == 0) {
        // full rows, 16-byte aligned: C/4 vector loads instead of C guarded scalar ones
#pragma unroll
        for (int i = 0; i < C / 4; ++i) {
            const float4 t = ((const float4 *)src)[i];
            v[4 * i] = t.x, v[4 * i + 1] = t.y, v[4 * i + 2] = t.z, v[4 * i + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < C; ++i)
            v[i] = i < c ? src[i] : 0.f;
    }
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
    if (C % 4 == 0 && c == C && ((uintptr_t)src & 15) == 0) {
#pragma unroll
        for (int i = 0; i < C / 4; ++i) {
            const float4 t = ((const float4 *)src)[i];      // src is a valid row also when !live
            q[4 * i] = live ? t.x : 0.f, q[4 * i + 1] = live ? t.y : 0.f;
            q[4 * i + 2] = live ? t.z : 0.f, q[4 * i + 3] = live ? t.w : 0.f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < C; ++i)
            q[i] = (live && i < c) ? src[i] : 0.f;
    }
    rq = 0.f;
#pragma unroll
    for (int i = 0; i < C; ++i)
        rq = __builtin_fmaf(q[i], q[i], rq);
}

// C = 24: 320 rows so that a whole 312-point patch is one tile (36 KiB of LDS)
constexpr int tile_rows(int C) { return C == 3 ? 1024 : (C <= 8 ? 512 : (C == 24 ? 320 : (C <= 32 ? 256 : 128))); }

typedef float kg_v4f __attribute__((ext_vector_type(4)));

// plain v_min_f32 (fminf() adds a canonicalising v_max; both are half-rate VALU instructions on gfx950)
__device__ __forceinline__ float kg_min(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---------------------------------------------------------------------------------------------
// small k: lane-per-query register insertion
// ---------------------------------------------------------------------------------------------
template <int C, int KMAX>
__global__ __launch_bounds__(512, (KMAX <= 33 ? 4 : 2)) void knn_insert_kernel(KnnArgs a)
{
    constexpr int TILE = tile_rows(C);
    constexpr int F4 = Row<C>::F4;
    __shared__ float4 tile[TILE * F4];
    __shared__ __attribute__((aligned(16))) float addend[C == 3 ? TILE : 1];
    if (a.gate && a.uws[a.gate < 0 ? 0 : a.gate] == 0)
        return;
    // work item = (batch element, block of queries); gated launches use a small grid and stride over
    // the items so that the (usual) early exit does not pay for dispatching tens of thousands of
    // workgroups (measured: 0.8 ms per empty 15 360-block launch)
    const int total = a.nblk_x * a.b;
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
    const int b = w / a.nblk_x;
    const int pb = a.pts_of ? a.pts_of[b] : b;
    const int n = a.n_arr ? a.n_arr[pb] : a.n;
    const int m = a.m_arr ? a.m_arr[b] : a.m;
    const int qi = (w - b * a.nblk_x) * blockDim.x + threadIdx.x;
    const bool live = qi < m;
    const bool use_dup = a.dup != nullptr && a.uws[0] != 0;
    // Optimistic mode: duplicate-flagged candidates are skipped instead of penalised.  With the
    // reference's D' = D + max(D)*dup every duplicate ranks behind every first occurrence unless
    // fewer than k first occurrences exist (or rounding puts a duplicate in front of the largest
    // distances); each query VERIFIES that its k-th distance is below a lower bound of every
    // duplicate's D' (fl(min dup D + max D seen) <= fl(D + true max), fp32 addition is monotone).
    // If any query cannot, uws[1] is raised and the caller's gated launches redo the call with the
    // reference arithmetic -- results are exact either way, the max(D) pass is skipped when it
    // cannot matter (it is a full extra distance pass over up to 12 480 candidates per query).
    const bool optimistic = use_dup && a.mode == 1;
    const float dmax = (use_dup && !optimistic) ? tpu3_unmono(a.uws[4 + (a.grp ? a.grp[b] : 0)]) : 0.f;
    float dupmin = __builtin_inff(), dqmax = -__builtin_inff();
    // Optimistic mode over a COMPACTED candidate list (the ascending indices of the first
    // occurrences): the duplicates are not even visited -- the merged cloud of overlapping patches
    // holds every point ~5 times.  Positions in the list are in index order, so ties resolve as
    // before; they are translated to row indices at the end.  A duplicate's distance equals its first
    // occurrence's bit for bit, so max D over the list is max D over all rows and the nearest
    // distance is a lower bound of every duplicate's: the same verification applies.
    const bool compact = optimistic && a.cand != nullptr;
    const int32_t *CAND = compact ? a.cand + (size_t)pb * a.n : nullptr;
    const int nscan = compact ? a.cand_count[pb] : n;

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

    for (int j0 = 0; j0 < nscan; j0 += TILE) {
        const int len = min(TILE, nscan - j0);
        __syncthreads();
        for (int i = threadIdx.x; i < len; i += blockDim.x) {
            // optimistic: the slot carries the dup flag itself; otherwise the addend max(D)*dup
            const int src = compact ? CAND[j0 + i] : j0 + i;
            const float add = compact ? 0.f
                                      : (use_dup ? (optimistic ? (float)DUP[src] : dmax * (float)DUP[src]) : 0.f);
            stage_row<C>(tile + i * F4, P + (size_t)src * a.c, a.c, add);
            if (C == 3)
                addend[i] = (compact || !use_dup) ? tile[i * F4].w : add;    // four-at-a-time loop: |p|^2, contiguous
        }
        __syncthreads();
        int jstart = 0;
        if constexpr (C == 3) {
            if (compact || !use_dup) {
                // first-occurrence lists (the inter-level kNN) and plain searches without duplicate handling
                // (the outlier filter): four candidates per step.  Lane l reads the
                // row of candidate j + l%4 (one 16-byte LDS read instead of four broadcasts), the three
                // products run as v_mfma_f32_4x4x1 (one fused multiply-add per output: the oracle's chain),
                // and ONE comparison on the minimum of the four decides whether any of them can enter the
                // list; the running maximum for the verification is two v_max3.  Runs on every lane of the
                // block (lanes beyond m carry a zero query): the MFMA wants the whole wave.
                const int len4 = len & ~3;
                for (int j = 0; j < len4; j += 4) {
                    const float4 p = tile[j + (int)(threadIdx.x & 3)];
                    kg_v4f acc;
                    // one accumulator: a dependent MFMA needs two wait states after its producer, and the
                    // result is not interlocked against VALU reads -- the compiler does not see inside the asm
                    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %4, 0\n\ts_nop 1\n\t"
                                 "v_mfma_f32_4x4x1_16b_f32 %0, %2, %5, %0\n\ts_nop 1\n\t"
                                 "v_mfma_f32_4x4x1_16b_f32 %0, %3, %6, %0\n\ts_nop 7"
                                 : "=&v"(acc)
                                 : "v"(p.x), "v"(p.y), "v"(p.z), "v"(q[0]), "v"(q[1]), "v"(q[2]));
                    const float4 rp = *(const float4 *)(addend + j);
                    const float d0 = __builtin_fmaf(-2.f, acc[0], rq) + rp.x;
                    const float d1 = __builtin_fmaf(-2.f, acc[1], rq) + rp.y;
                    const float d2 = __builtin_fmaf(-2.f, acc[2], rq) + rp.z;
                    const float d3 = __builtin_fmaf(-2.f, acc[3], rq) + rp.w;
                    float lo;
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(dqmax) : "v"(d0), "v"(d1));
                    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(dqmax) : "v"(d2), "v"(d3));
                    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(d0), "v"(d1), "v"(d2));
                    lo = kg_min(lo, d3);
                    if (lo < bd[KMAX - 1]) {
                        const float dd[4] = {d0, d1, d2, d3};
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const float d = dd[t];
                            if (d < bd[KMAX - 1]) {
                                const int id = j0 + j + t;
#pragma unroll
                                for (int i = KMAX - 1; i > 0; --i) {
                                    const bool up = bd[i - 1] > d;
                                    const bool here = bd[i] > d;
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
                jstart = len4;
            }
        }
        if (live) {
            for (int j = jstart; j < len; ++j) {
                float d = row_dist<C>(tile + j * F4, q, rq);
                if (use_dup) {
                    const float ad = compact ? 0.f : (C == 3 ? addend[j] : tile[j * F4 + C / 4].y);
                    if (optimistic) {
                        dqmax = fmaxf(dqmax, d);
                        if (ad != 0.f) {
                            dupmin = fminf(dupmin, d);
                            continue;
                        }
                    } else {
                        d = d + ad;
                    }
                }
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
        float tk = __builtin_inff();
        if (compact && nscan < n)
            dupmin = bd[0];
#pragma unroll
        for (int i = 0; i < KMAX; ++i)
            if (i < a.k) {
                store_idx(a, o + i, (compact && bi[i] < nscan) ? CAND[bi[i]] : bi[i]);
                if (a.dist)
                    a.dist[o + i] = bd[i];
                if (i == a.k - 1)
                    tk = bd[i];
            }
        if (optimistic && dupmin < __builtin_inff() && !(tk < dupmin + dqmax))
            a.uws[1] = 1u;      // cannot prove the duplicates stay out of the top k: redo exactly
    }
    __syncthreads();
    }   // work items
}

#ifndef KG_USE_MFMA
#define KG_USE_MFMA 1
#endif

// Distances of this lane's query to the candidates j .. j+3 of the staged tile.
// C % 4 == 0: the dot products run on v_mfma_f32_4x4x1_16b_f32 -- 16 blocks of (4 candidates x 1
// channel) x (1 channel x 4 queries): operand A of lane l is a channel of candidate j + l%4 (its
// 16-byte row reads are shared by the lanes of equal l%4), operand B is the lane's own query channel,
// and the four results of lane l are <q_l, p_j..j+3>.  One instruction is a single fused multiply-add
// per output, so 24 of them in ascending channel order are the oracle's fmaf chain bit for bit.  An fp32
// MFMA runs on the VALU datapath at the rate of v_fma_f32 (256 FMAs per 8 cycles, no overlap with other
// VALU work: tools/mfma_overlap_probe.hip); what it saves is LDS traffic and instruction issue.
template <int C, int G>
__device__ __forceinline__ void kg_dist(const float4 *tile, const float *rps, int j, const float (&q)[C], float rq,
                                        float *d)
{
    // distances to the candidates j .. j+4G-1: G accumulators (4 candidates each) advance together, so
    // consecutive MFMAs never wait for each other's result
    constexpr int F4 = Row<C>::F4;
    if constexpr (C % 4 == 0 && KG_USE_MFMA) {
        // The MFMAs are volatile asm in round-robin order over the G accumulators (the compiler's own
        // schedule ran each accumulator's chain back to back, with a wait on every LDS read); the
        // rows of channel quad i+1 are in flight while quad i is consumed.
        const float4 *row = tile + (j + (int)(threadIdx.x & 3)) * F4;
        kg_v4f acc[G];
        float4 p[G], pn[G];
#pragma unroll
        for (int g = 0; g < G; ++g)
            p[g] = row[g * 4 * F4];
#pragma unroll
        for (int i = 0; i < C / 4; ++i) {
            if (i + 1 < C / 4) {
#pragma unroll
                for (int g = 0; g < G; ++g)
                    pn[g] = row[g * 4 * F4 + i + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (i == 0)     // (s_nop: the compiler may have copied an operand into place with a v_mov just before)
                    asm volatile("s_nop 1\n\tv_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=&v"(acc[g]) : "v"(p[g].x), "v"(q[0]));
                else
                    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(p[g].x), "v"(q[4 * i]));
            }
#pragma unroll
            for (int g = 0; g < G; ++g)
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(p[g].y), "v"(q[4 * i + 1]));
#pragma unroll
            for (int g = 0; g < G; ++g)
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(p[g].z), "v"(q[4 * i + 2]));
#pragma unroll
            for (int g = 0; g < G; ++g)
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(p[g].w), "v"(q[4 * i + 3]));
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < C / 4) {
#pragma unroll
                for (int g = 0; g < G; ++g)
                    p[g] = pn[g];
            }
        }
        // MFMA results are not interlocked against VALU reads: 8 wait states before the epilogue
        asm volatile("s_nop 7" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float4 rp = *(const float4 *)(rps + j + 4 * g);
            d[4 * g + 0] = __builtin_fmaf(-2.f, acc[g][0], rq) + rp.x;
            d[4 * g + 1] = __builtin_fmaf(-2.f, acc[g][1], rq) + rp.y;
            d[4 * g + 2] = __builtin_fmaf(-2.f, acc[g][2], rq) + rp.z;
            d[4 * g + 3] = __builtin_fmaf(-2.f, acc[g][3], rq) + rp.w;
        }
    } else {
#pragma unroll
        for (int t = 0; t < 4 * G; ++t)
            d[t] = row_dist<C>(tile + (j + t) * F4, q, rq);
    }
}

// stage candidates [j0, j0+len) of P into the tile, padded to a multiple of 32 rows with |p|^2 = +inf
// (distance +inf: never among the k smallest, never selected)
template <int C>
__device__ __forceinline__ void kg_stage(float4 *tile, float *rps, const float *__restrict__ P, int j0, int len, int c)
{
    constexpr int F4 = Row<C>::F4;
    const int len32 = (len + 31) & ~31;
    for (int i = threadIdx.x; i < len32; i += blockDim.x) {
        if (i < len) {
            stage_row<C>(tile + i * F4, P + (size_t)(j0 + i) * c, c, 0.f);
            rps[i] = C == 3 ? tile[i * F4].w : tile[i * F4 + C / 4].x;
        } else {
#pragma unroll
            for (int t = 0; t < F4; ++t)
                tile[i * F4 + t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (C == 3)
                tile[i * F4].w = __builtin_inff();
            else
                tile[i * F4 + C / 4].x = __builtin_inff();
            rps[i] = __builtin_inff();
        }
    }
}

// compare-exchange on two registers: plain v_min_f32 / v_max_f32 (2-source VALU, full rate).  The
// insertion chain this replaces spent one v_med3_f32 per list slot and candidate, and a 3-source
// VALU instruction issues at half the rate of a 2-source one on gfx950 (tools/valu_probe.hip:
// 4.5 vs 2.3 cycles per wave-instruction).
__device__ __forceinline__ void kg_cx(float &a, float &b)
{
    float lo, hi;
    asm("v_min_f32 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    a = lo;
    b = hi;
}

// Fold L new distances into the running selection: `lst` = the L smallest so far (ascending), `e` = the
// next one (the (L+1)-th smallest).  The new values are sorted by a Batcher network, the L smallest of
// the 2L come out of one min(lst[i], nw[L-1-i]) layer as a bitonic sequence (log2 L clean-up layers sort
// it), and the smallest value that layer discards is the new candidate for `e`.
// About 20 2-source operations per candidate against 33 half-rate ones for the insertion chain.
template <int L>
__device__ __forceinline__ void kg_fold(float (&lst)[L], float &e, float (&nw)[L])
{
    static_assert(L == 16 || L == 32, "list length");
    if constexpr (L == 32) {
        KG_SORT32(nw, kg_cx)
    } else {
        KG_SORT16(nw, kg_cx)
    }
    float dm[4] = {e, __builtin_inff(), __builtin_inff(), __builtin_inff()};
#pragma unroll
    for (int i = 0; i < L; ++i) {
        kg_cx(lst[i], nw[L - 1 - i]);                   // lst[i] = min, nw[L-1-i] = max (discarded)
        dm[i & 3] = kg_min(dm[i & 3], nw[L - 1 - i]);
    }
    e = kg_min(kg_min(dm[0], dm[1]), kg_min(dm[2], dm[3]));
#pragma unroll
    for (int st = L / 2; st >= 1; st >>= 1)
#pragma unroll
        for (int i = 0; i < L; ++i)
            if ((i & st) == 0)
                kg_cx(lst[i], lst[i + st]);
}

// v = v + (lane's bit of mask);  w = 2 w + (lane's bit of mask): one v_addc_co_u32 each, the lane mask
// (a v_cmp result) is the carry-in
__device__ __forceinline__ int kg_add_bit(int v, uint64_t mask)
{
    int r;
    uint64_t co;
    asm("v_addc_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(r), "=&s"(co) : "v"(v), "s"(mask));
    return r;
}

__device__ __forceinline__ uint32_t kg_push_bit(uint32_t w, uint64_t mask)
{
    uint32_t r;
    uint64_t co;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(r), "=&s"(co) : "v"(w), "s"(mask));
    return r;
}

// ---------------------------------------------------------------------------------------------
// kNN *graph* for the fused DenseEdgeConv: the k nearest as a SET, nearest first
// ---------------------------------------------------------------------------------------------
// DenseEdgeConv drops the nearest neighbour (itself) and max-pools over the rest
// (network/layers.py:33-35,63), so the order of the other k-1 is irrelevant.  That allows a much
// cheaper exact selection than a sorted insertion with indices (~190 VALU ops per candidate):
//   pass 1  keeps only the k smallest DISTANCES, sorted, with one v_med3_f32 per slot
//           (new[i] = med3(old[i-1], d, old[i]); all slots independent);
//   pass 2  recomputes the distances (bit-identical) and collects the indices with
//           d < T_k, plus the first q candidates (lowest indices) with d == T_k, where q is how often
//           T_k occurs in the sorted list -- exactly the top-k under (distance, index) order.
// Slot 0 receives the nearest (lowest index among the minima), slots 1..k-1 the others in index
// order.  Runs only when the point sets hold no duplicated rows (uws[0] == 0); otherwise the exact
// sorted kernels above run instead (gated the other way).
template <int C, int K>
__global__ __launch_bounds__(512, 4) void knn_graph_kernel(KnnArgs a)
{
    constexpr int TILE = tile_rows(C);
    constexpr int F4 = Row<C>::F4;
    __shared__ float4 tile[TILE * F4];
    __shared__ __attribute__((aligned(16))) float rps[TILE];     // |p|^2 of the tile's rows, contiguous
    if (a.uws && a.uws[0] != 0)
        return;
    // a.mode == 2: self query (query set == point set) without a de-duplication pre-pass.  Identical
    // rows have D == 0 exactly (|q|^2, <q,p> and |p|^2 are then the same fmaf chain), so if no query
    // holds a second zero among its k smallest distances, no row is duplicated and the result is final; otherwise
    // uws[2] is raised and the caller's gated launches (hash de-duplication + exact kernels) redo it.
    const bool self_check = a.mode == 2;
    bool saw_zero = false;
    const int b = blockIdx.y;
    const int pb = a.pts_of ? a.pts_of[b] : b;
    const int n = a.n_arr ? a.n_arr[pb] : a.n;
    const int m = a.m_arr ? a.m_arr[b] : a.m;
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = qi < m;
    float q[C], rq;
    load_query<C>(q, rq, a.query + ((size_t)b * a.m + (live ? qi : 0)) * a.c, a.c, live);
    const float *P = a.points + (size_t)pb * a.n * a.c;

    // Both passes run on every lane of the block (lanes beyond m carry a zero query), so the loops
    // are wave-uniform: scalar trip counts, no exec juggling; only the stores are guarded.
    // ---- pass 1: the k smallest distances ---------------------------------------------------------
    constexpr int L = K - 1;
    float lst[L], kth = __builtin_inff();
#pragma unroll
    for (int i = 0; i < L; ++i)
        lst[i] = __builtin_inff();
    for (int j0 = 0; j0 < n; j0 += TILE) {
        const int len = __builtin_amdgcn_readfirstlane(min(TILE, n - j0));
        __syncthreads();
        kg_stage<C>(tile, rps, P, j0, len, a.c);
        __syncthreads();
        for (int j = 0; j < len; j += L) {
            float nw[L];
#pragma unroll
            for (int t = 0; t < L; t += 16)
                kg_dist<C, 4>(tile, rps, j + t, q, rq, nw + t);
            kg_fold<L>(lst, kth, nw);
        }
    }
    const float t1 = lst[0], tk = kth;
    int quota = 1;
#pragma unroll
    for (int i = 0; i < L; ++i)
        quota += lst[i] == tk ? 1 : 0;
    if (self_check && live) {
        // the query's own row is one zero; a second one (or a list so degenerate that zeros could
        // have been pushed out of it) asks for the exact path
        int zeros = tk == 0.f ? 1 : 0;
#pragma unroll
        for (int i = 0; i < L; ++i)
            zeros += lst[i] == 0.f ? 1 : 0;
        saw_zero = zeros >= 2 || !(tk > 0.f);
    }
    if (saw_zero)
        a.uws[2] = 1u;
    // ---- pass 2: the indices ---------------------------------------------------------------------------
    // Branch-free per candidate: the three comparisons land in lane masks (SGPR pairs), the mask logic
    // is scalar, and one v_addc per mask shifts the lane's bit into a 32-candidate word (w = 2w + bit).
    // After every 32 candidates the words are expanded into indices (about 3.4 bits per lane and word).
    int32_t *out = (int32_t *)a.idx + ((size_t)b * a.m + (live ? qi : 0)) * K;
    int cnt = 1, used = 0;
    bool found = false;
    for (int j0 = 0; j0 < n; j0 += TILE) {
        const int len = __builtin_amdgcn_readfirstlane(min(TILE, n - j0));
        if (n > TILE) {         // the tile still holds the whole set when it fits (the 312-point case)
            __syncthreads();
            kg_stage<C>(tile, rps, P, j0, len, a.c);
            __syncthreads();
        }
        for (int g0 = 0; g0 < len; g0 += 32) {
            constexpr int gl = 32;      // incl. pad rows (never selected)
            uint32_t w = 0, mw = 0;
            for (int t0 = 0; t0 < gl; t0 += 16) {
                float d[16];
                kg_dist<C, 4>(tile, rps, g0 + t0, q, rq, d);
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const uint64_t lt = __builtin_amdgcn_ballot_w64(d[t] < tk);
                    const uint64_t eq = __builtin_amdgcn_ballot_w64(d[t] == tk);
                    const uint64_t room = __builtin_amdgcn_ballot_w64(used < quota);
                    const uint64_t tie = eq & room;
                    used = kg_add_bit(used, tie);
                    w = kg_push_bit(w, lt | tie);
                    mw = kg_push_bit(mw, __builtin_amdgcn_ballot_w64(d[t] == t1));
                }
            }
            if (!live)
                w = 0;
            const int last = j0 + g0 + gl - 1;      // candidate of bit 0
            while (w) {
                const int hb = 31 - __clz(w);
                const uint32_t bit = 1u << hb;
                if (!found && (mw & bit)) {
                    out[0] = last - hb;
                    found = true;
                } else if (cnt < K) {
                    out[cnt++] = last - hb;
                }
                w &= ~bit;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Self kNN graph in ONE pass: the candidate index rides in the low mantissa bits
// ---------------------------------------------------------------------------------------------
// The two-pass kernel above computes every distance twice (pass 2 re-derives them to find the indices:
// ~35 % of its time).  For a SELF query the indices can travel through the sorting networks instead:
//     key = (bits(D) with the low IB bits cleared) | j          IB = ceil(log2 n), signed-int order
// The query's own row is left out (it is the nearest by definition: D == 0 exactly for identical rows), the
// list keeps the L = k - 1 smallest keys of the others and `e` the next one.  Truncation can only misorder
// candidates whose truncated distances are EQUAL, and that only matters at the boundary: if
// trunc(lst[L-1]) != trunc(e) the L keys hold exactly the L nearest others (ties to the lowest index, like the
// oracle).  Otherwise -- 0.4 % of the queries at n = 312 -- a second sweep (run by the wave only if one of its
// lanes needs it) re-derives the distances and picks, among the candidates with that truncated value, the
// ones with the smallest (D, j): up to two boundary slots per query are settled this way; more (or a negative
// or zero distance to another row: possibly duplicated rows) raise uws[2] = "use the exact path".
// Output: slot 0 = the query itself, slots 1 .. k-1 = the other members in no particular order (DenseEdgeConv
// drops the first and max-pools over the rest).
__device__ __forceinline__ void kg_cx_i32(int &a, int &b)
{
    int lo, hi;
    asm("v_min_i32 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_max_i32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    a = lo;
    b = hi;
}

// Fold L new keys into the running selection: `lst` = the L smallest so far (ascending), `e1` <= `e2` = the next
// two (the two smallest keys ever discarded).
template <int L>
__device__ __forceinline__ void kg_fold_i32(int (&lst)[L], int &e1, int &e2, int (&nw)[L])
{
    static_assert(L == 16 || L == 32, "list length");
    if constexpr (L == 32) {
        KG_SORT32(nw, kg_cx_i32)
    } else {
        KG_SORT16(nw, kg_cx_i32)
    }
    // The discarded keys x_i = max(lst[i], nw[L-1-i]) are the maximum of an ascending and a descending sequence:
    // V-shaped, so their two smallest are NEIGHBOURS -- smallest = min_i x_i, second = min_i max(x_i, x_{i+1})
    // (any pair's maximum is >= the second smallest; the pair around the valley attains it).  ~2 instructions
    // per key (v_max + v_min3 trees) instead of 4 for a running (min, second-min) pair.
#pragma unroll
    for (int i = 0; i < L; ++i)
        kg_cx_i32(lst[i], nw[L - 1 - i]);               // lst[i] = min, nw[L-1-i] = max (discarded)
    int pm[L];
#pragma unroll
    for (int i = 0; i + 1 < L; ++i)
        pm[i] = max(nw[i], nw[i + 1]);
    pm[L - 1] = e2;
    int c1 = nw[0], c2 = pm[0];
#pragma unroll
    for (int i = 1; i + 1 < L; i += 2) {
        c1 = min(c1, min(nw[i], nw[i + 1]));
        c2 = min(c2, min(pm[i], pm[i + 1]));
    }
    c1 = min(c1, nw[L - 1]);
    c2 = min(c2, pm[L - 1]);                            // ... and the old e2
    e2 = min(c2, max(e1, c1));
    e1 = min(e1, c1);
#pragma unroll
    for (int st = L / 2; st >= 1; st >>= 1)
#pragma unroll
        for (int i = 0; i < L; ++i)
            if ((i & st) == 0)
                kg_cx_i32(lst[i], lst[i + st]);
}

template <int C, int K>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(3, 3))) void knn_graph_key_kernel(KnnArgs a)
{
    constexpr int TILE = tile_rows(C);
    constexpr int F4 = Row<C>::F4;
    constexpr int L = K - 1;
    __shared__ float4 tile[TILE * F4];
    __shared__ __attribute__((aligned(16))) float rps[TILE];
    if (a.uws && a.uws[0] != 0)
        return;
    const int b = blockIdx.y;
    const int n = a.n;                              // dense self query: m == n, no layout
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = qi < n;
    float q[C], rq;
    load_query<C>(q, rq, a.query + ((size_t)b * n + (live ? qi : 0)) * a.c, a.c, live);
    const float *P = a.points + (size_t)b * n * a.c;
    const int IB = 32 - __clz(n - 1);               // index bits (n >= 2)
    const int keep = ~((1 << IB) - 1);
    const int q_first = __builtin_amdgcn_readfirstlane(qi);      // lane 0's query: qi - lane

    int lst[L], e = 0x7FFFFFFF, e2 = 0x7FFFFFFF;
#pragma unroll
    for (int i = 0; i < L; ++i)
        lst[i] = 0x7FFFFFFF;
    for (int j0 = 0; j0 < n; j0 += TILE) {
        const int len = __builtin_amdgcn_readfirstlane(min(TILE, n - j0));
        __syncthreads();
        kg_stage<C>(tile, rps, P, j0, len, a.c);
        __syncthreads();
        for (int j = 0; j < len; j += L) {
            float d[L];
#pragma unroll
            for (int t = 0; t < L; t += 16)
                kg_dist<C, 4>(tile, rps, j + t, q, rq, d + t);
            int nw[L];
            const int c0 = j0 + j;                  // wave-uniform
#pragma unroll
            for (int t = 0; t < L; ++t)
                nw[t] = (__float_as_int(d[t]) & keep) | (c0 + t);
            // the query's own row is left out; only the chunks that overlap this wave's 64 queries can hold it
            if (c0 + L > q_first && c0 < q_first + 64) {
#pragma unroll
                for (int t = 0; t < L; ++t)
                    nw[t] = c0 + t == qi ? 0x7FFFFFFF : nw[t];
            }
            kg_fold_i32<L>(lst, e, e2, nw);
        }
    }
    // a negative or (truncated) zero distance to ANOTHER row: rows may be duplicated -> the exact path decides
    bool redo = live && (lst[0] >> IB) <= 0;
    // boundary check: does a key OUTSIDE the list share the truncated distance T of the list's last member?
    const int T = lst[L - 1] >> IB;
    bool amb = live && !redo && (e >> IB) == T && e != 0x7FFFFFFF;
    if (n <= TILE && __builtin_amdgcn_ballot_w64(amb)) {
        // Usually only ONE outsider collides (the second does in ~0.06 % of the queries): then every candidate in
        // question is known by index -- the list's members with truncated value T (at its end) and e -- and the
        // lane re-derives just their distances (the same fmaf chain, rows still staged) and keeps the smallest
        // (D, j).  Settles up to two boundary slots; anything deeper goes to the sweep below.
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < L; ++i)
            cnt += (lst[i] >> IB) == T ? 1 : 0;
        const bool easy = amb && (e2 >> IB) != T && cnt <= 2;
        if (easy) {
            const int ja = lst[L - 1] & ~keep, jb = lst[L - 2] & ~keep, jc = e & ~keep;
            __builtin_amdgcn_sched_barrier(0);          // one row's loads in flight at a time (registers)
            const int da = __float_as_int(row_dist<C>(tile + ja * F4, q, rq));
            __builtin_amdgcn_sched_barrier(0);
            const int dc = __float_as_int(row_dist<C>(tile + jc * F4, q, rq));
            __builtin_amdgcn_sched_barrier(0);
            // (D, j) order; the members are kept unless the outsider is strictly better
            auto less = [](int d0, int j0, int d1, int j1) { return d0 < d1 || (d0 == d1 && j0 < j1); };
            if (cnt == 1) {
                if (less(dc, jc, da, ja))
                    lst[L - 1] = (dc & keep) | jc;
            } else {
                const int db = __float_as_int(row_dist<C>(tile + jb * F4, q, rq));
                // drop the worst of the three
                const bool a_worst = !less(da, ja, db, jb) && !less(da, ja, dc, jc);
                const bool b_worst = !a_worst && !less(db, jb, dc, jc);
                if (a_worst)
                    lst[L - 1] = (dc & keep) | jc;
                else if (b_worst)
                    lst[L - 2] = (dc & keep) | jc;
            }
            amb = false;
        }
    }
    // (the sweep re-stages tiles when the set spans several: then the whole workgroup must take it together)
    const bool sweep = n > TILE ? (bool)__syncthreads_or(amb ? 1 : 0) : __builtin_amdgcn_ballot_w64(amb) != 0;
    if (sweep) {
        // second sweep, wave-uniform: among the candidates with truncated distance T keep the two smallest
        // (D, j) -- candidates arrive in ascending j, so strict comparisons keep the lowest index on ties
        int b0 = 0x7FFFFFFF, b1 = 0x7FFFFFFF, j0b = 0, j1b = 0;
        for (int j0 = 0; j0 < n; j0 += TILE) {
            const int len = __builtin_amdgcn_readfirstlane(min(TILE, n - j0));
            if (n > TILE) {
                __syncthreads();
                kg_stage<C>(tile, rps, P, j0, len, a.c);
                __syncthreads();
            }
            for (int j = 0; j < len; j += 16) {
                float d[16];
                kg_dist<C, 4>(tile, rps, j, q, rq, d);
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int idx = j0 + j + t;
                    const int bits = __float_as_int(d[t]);
                    if ((bits >> IB) == T && idx != qi) {
                        if (bits < b0) {
                            b1 = b0; j1b = j0b; b0 = bits; j0b = idx;
                        } else if (bits < b1) {
                            b1 = bits; j1b = idx;
                        }
                    }
                }
            }
        }
        if (amb) {
            // the list's members with truncated value T sit at its end (sorted): replace them
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < L; ++i)
                cnt += (lst[i] >> IB) == T ? 1 : 0;
            if (cnt > 2) {
                redo = true;
            } else {
                lst[L - 1] = ((cnt == 1 ? b0 : b1) & keep) | (cnt == 1 ? j0b : j1b);
                if (cnt == 2)
                    lst[L - 2] = (b0 & keep) | j0b;
            }
        }
    }
    if (redo)
        a.uws[2] = 1u;
    if (live) {
        int32_t *out = (int32_t *)a.idx + ((size_t)b * n + qi) * K;
        out[0] = qi;
#pragma unroll
        for (int i = 0; i < L; ++i)
            out[1 + i] = lst[i] & ~keep;
    }
}

// ---------------------------------------------------------------------------------------------
// Self kNN graph of one patch, SLAB form (r5): a wave's queries are neighbours, far chunks are skipped
// ---------------------------------------------------------------------------------------------
// knn_graph_key_kernel runs every chunk of 32 candidates through the sorting network for every wave, because in
// patch order some lane of a wave always accepts something.  The top-k SET does not depend on the visiting order,
// so here the workgroup first orders its patch along ONE direction v of feature space:
//     v   = one power iteration of the centred second moment, started at (the row farthest from row 0) - row 0
//           (tools/knn_accept_sim.py: as good as the first principal axis for this purpose);
//     t_i = <x_i, v>, |v| < 1;   rows sorted by t (a binned counting sort: order inside a bin is arbitrary).
// A wave then owns 64 consecutive rows of that order -- a SLAB of the patch -- and visits the 32-row chunks outwards
// from its own, left and right alternately.  (t_q - t_c)^2 <= |x_q - x_c|^2 for every pair, so once every lane's
// gap to the t-range of the chunk AND ALL CHUNKS BEYOND IT on that side (r6: a suffix-min / prefix-max table -- the
// binned order is not monotone inside a bin, each chunk's own range is not enough), squared, exceeds the lane's
// current 32nd key (with a margin for the rounding of t and of the expanded-form distance, see E1 / E2), none of
// them can change any list: that side is closed without computing a single distance (~31 % of all chunks on the
// feature rows of a 16x run).  A chunk that survives the bound still skips the sorting network when no lane's smallest new key
// beats its 32nd (~10 %).  Keys carry the sorted POSITION in their low bits; truncation ties are only ever resolved
// at the list's boundary, and there by (distance, ORIGINAL index) exactly like the oracle.  A skipped key has a
// truncated distance strictly above the list's last one at that time (hence above the final one), so it can never
// be the boundary collision `e` / `e2` watch for.
// n <= 320 (one tile), one workgroup of ceil(n / 64) waves per patch.  Output as knn_graph_key_kernel.
// inclusive prefix sum over the 64 lanes of a wave: Hillis-Steele inside each row of 16 (row_shr), then the rows'
// totals handed on with row_bcast15 / row_bcast31
__device__ __forceinline__ int kg_wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);     // row_bcast15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);     // row_bcast31 -> rows 2, 3
    return v;
}

// Pre-pass of the slab form: the order of every patch along its direction v.  One workgroup per patch, a thread per
// row.  Cheap in registers, four workgroups per compute unit -- which makes it bound by instruction ISSUE (20 waves
// per compute unit), so every phase is written for few instructions: rows and directions are re-read from LDS rather
// than kept, the transposed sums run without per-lane predicates over zero-padded rows, and what is the same for
// every thread (the direction, its norm, the bin scale) is computed once by 24 lanes of wave 0.  (Fused into the graph
// kernel, whose 158 registers allow two workgroups per compute unit, the same work cost 13 % of that kernel's time.)
// Output, in the head of the patch's own (n, K) index rows -- scratch until the graph kernel overwrites them with
// the result: words [0, n) = t of the row at each sorted position, words [n, 2n) = that row's number.
// Rows are exactly C floats wide and 16-byte aligned (the launcher checks).
template <int C>
__global__ __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(5, 8))) void knn_slab_order_kernel(KnnArgs a)
{
    constexpr int TILE = 320;
    constexpr int NW = TILE / 64;
    constexpr int RMAX = 26;                        // rows per (channel, row group) thread of the transposed sums
    constexpr int ROWS = (TILE / C) * RMAX;         // 338: every row a group may touch exists (zero beyond n)
    constexpr int Q = C / 4;
    __shared__ __attribute__((aligned(16))) float tile[ROWS * C];      // the patch's rows, original order
    __shared__ float ts[ROWS];                      // s_i = <x_i, v0>
    __shared__ int hist[TILE];                      // bin counts, then exclusive offsets
    __shared__ __attribute__((aligned(16))) float vec[2 * C + 4];      // [A | S | sum of s]
    __shared__ __attribute__((aligned(16))) float dir[C + 4];          // v (|v| < 1), then mean of t, bin scale
    __shared__ int wfar[NW];                        // per wave: arg-max key of |x_i - x_0|^2
    if (a.uws && a.uws[0] != 0)
        return;
    const int b = blockIdx.y;
    const int n = a.n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = blockDim.x;                // ceil(n / 64) * 64
    const int nwaves = nthreads >> 6;
    const bool live = tid < n;
    const float4 *X4 = (const float4 *)(a.query + (size_t)b * n * C);
    float4 *tile4 = (float4 *)tile;
    const float4 *own4 = tile4 + tid * Q;
    {
        float4 x[Q], r0[Q];
        float d0 = 0.f;
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            x[i] = X4[(size_t)(live ? tid : 0) * Q + i];
            r0[i] = X4[i];
        }
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            if (!live)
                x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            tile4[tid * Q + i] = x[i];
            const float e0 = x[i].x - r0[i].x, e1 = x[i].y - r0[i].y, e2 = x[i].z - r0[i].z, e3 = x[i].w - r0[i].w;
            d0 = __builtin_fmaf(e0, e0, d0), d0 = __builtin_fmaf(e1, e1, d0);
            d0 = __builtin_fmaf(e2, e2, d0), d0 = __builtin_fmaf(e3, e3, d0);
        }
        // zero rows behind the workgroup's own (the transposed sums read up to ROWS rows)
        for (int r = nthreads * Q + tid; r < ROWS * Q; r += nthreads)
            tile4[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = nthreads + tid; r < ROWS; r += nthreads)
            ts[r] = 0.f;
        hist[tid] = 0;
        if (tid < 2 * C + 4)
            vec[tid] = 0.f;
        // the row farthest from row 0 (any of the farthest: the row number rides in the low bits; a NaN / Inf row may
        // win: the direction is then useless, never unsafe -- the graph kernel's bound uses the t values whatever they are)
        const int key = live ? (int)((__float_as_uint(d0) & 0x7FFFFE00u) | (uint32_t)tid) : 0;
        const int wmax = tpu3_wave_max_i32(key);
        if (lane == 0)
            wfar[wave] = wmax;
    }
    __syncthreads();
    float n0sq = 0.f;
    {
        int far = wfar[0];
        for (int w = 1; w < nwaves; ++w)
            far = max(far, wfar[w]);
        const float4 *rowp = tile4 + (far & 0x1FF) * Q;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            const float4 xo = own4[i], a0 = tile4[i], ap = rowp[i];
            const float v0 = ap.x - a0.x, v1 = ap.y - a0.y, v2 = ap.z - a0.z, v3 = ap.w - a0.w;
            s = __builtin_fmaf(xo.x, v0, s), s = __builtin_fmaf(xo.y, v1, s);
            s = __builtin_fmaf(xo.z, v2, s), s = __builtin_fmaf(xo.w, v3, s);
            n0sq = __builtin_fmaf(v0, v0, n0sq), n0sq = __builtin_fmaf(v1, v1, n0sq);
            n0sq = __builtin_fmaf(v2, v2, n0sq), n0sq = __builtin_fmaf(v3, v3, n0sq);
        }
        ts[tid] = s;                                // s_i = <x_i, v0>, v0 = x_far - x_0 (dead rows: 0)
    }
    __syncthreads();
    // [A | S | sum s] = [sum_i s_i x_i | sum_i x_i | sum_i s_i]: thread (channel k, row group g) adds its R rows (zero
    // rows beyond n: no predicates), one LDS float atomic per thread and sum adds the groups (their order does not
    // matter: v is a heuristic)
    {
        const int G = nthreads / C;
        const int k = tid % C, g = tid / C;
        if (g < G) {
            const int R = (n + G - 1) / G;          // <= RMAX, wave-uniform
            const float *tp = tile + (g * R) * C + k, *sp = ts + g * R;
            float accA = 0.f, accS = 0.f, accT = 0.f;
#pragma unroll
            for (int u = 0; u < RMAX; ++u) {
                if (u < R) {
                    const float xv = tp[u * C], sv = sp[u];
                    accA = __builtin_fmaf(sv, xv, accA);
                    accS += xv;
                    accT += sv;
                }
            }
            atomicAdd(&vec[k], accA);
            atomicAdd(&vec[C + k], accS);
            if (k == 0)
                atomicAdd(&vec[2 * C], accT);
        }
    }
    __syncthreads();
    if (wave == 0) {
        // v' = A - (sum s / n) S  (= the centred second moment times v0), scaled to |v| < 1; by C lanes, once
        const int k = lane < C ? lane : 0;
        const float sv = vec[2 * C] / (float)n;
        const float S = vec[C + k];
        float v = lane < C ? __builtin_fmaf(-sv, S, vec[k]) : 0.f;
        const float nrm2 = tpu3_wave_sum_f32(v * v);
        // (1 - 2^-10): the rounding of nrm2 and of the reciprocal square root stays far inside
        const bool ok = nrm2 > 0.f && nrm2 < __builtin_inff();
        v = ok ? v * (__builtin_amdgcn_rsqf(nrm2) * 0.9990234375f) : 0.f;
        const float mean = tpu3_wave_sum_f32(lane < C ? S * v : 0.f) / (float)n;
        if (lane < C)
            dir[lane] = v;
        if (lane == 0) {
            // nthreads bins over mean +- 2.5 sigma with sigma^2 ~ |C v0| / (|v0| n), the iteration's own estimate of
            // the variance along v (no min / max reduction); rows beyond land in the end bins.  The bins only make
            // the order GOOD; the graph kernel's bound uses each chunk's true t-range.
            const float sig = __builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(nrm2) * __builtin_amdgcn_rsqf(n0sq) / (float)n);
            dir[C] = mean;
            dir[C + 1] = 0.2f * (float)nthreads * __builtin_amdgcn_rcpf(sig);       // (nthreads / 2) / (2.5 sigma)
        }
    }
    __syncthreads();
    float t = 0.f;
    int bin, slot = 0;
    {
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            const float4 xo = own4[i], v4 = ((const float4 *)dir)[i];
            t = __builtin_fmaf(xo.x, v4.x, t), t = __builtin_fmaf(xo.y, v4.y, t);
            t = __builtin_fmaf(xo.z, v4.z, t), t = __builtin_fmaf(xo.w, v4.w, t);
        }
        // binned counting sort by t: position = (rows in lower bins) + (arrival order inside the bin)
        float f = __builtin_fmaf(t - dir[C], dir[C + 1], 0.5f * (float)nthreads);
        f = __builtin_fminf(__builtin_fmaxf(f, 0.f), (float)(nthreads - 1));   // (NaN -> bin 0 through the max)
        bin = (int)f;
        if (live)
            slot = atomicAdd(&hist[bin], 1);
    }
    __syncthreads();
    if (wave == 0) {
        // exclusive offsets of the bins, by one wave: five consecutive bins per lane
        int h[NW], sum = 0;
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            h[u] = lane * NW + u < nthreads ? hist[lane * NW + u] : 0;
            sum += h[u];
        }
        int off = kg_wave_incl_scan(sum) - sum;
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            if (lane * NW + u < nthreads)
                hist[lane * NW + u] = off;
            off += h[u];
        }
    }
    __syncthreads();
    if (live) {
        const int pos = hist[bin] + slot;
        int32_t *scr = (int32_t *)a.idx + (size_t)b * n * a.k;
        scr[pos] = __float_as_int(t);
        scr[n + pos] = tid;
    }
}

#ifdef KG_TRACE
// (-DKG_TRACE, tools/knn_trace.py: per wave the shader clocks of its slab loop and the chunks it took through the network)
__device__ long long *g_kg_trace;
#endif

// (r5, measured and dropped: two five-wave patches per workgroup -- ten waves are dealt 3 + 3 + 2 + 2 over the SIMDs
// whatever the start, where a second five-wave workgroup only fits beside the first when the dispatcher starts it on
// another SIMD: tools/knn_trace.py counts 1.40 patches resident per compute unit, 1.65 paired -- but 0.438 vs 0.407 ms
// per launch: the kernel is bound by VALU issue, more resident waves only slow each other, and a ten-wave workgroup's
// staging no longer overlaps the previous one's tail.)
template <int C, int K>
__global__ __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(3, 3))) void knn_graph_slab_kernel(KnnArgs a)
{
    constexpr int TILE = 320;
    constexpr int F4 = Row<C>::F4;
    constexpr int L = K - 1;
    constexpr int NCH = TILE / L;
    constexpr int NW = TILE / 64;
    static_assert(C % 4 == 0 && L == 32 && TILE % L == 0, "slab form: 4 | C, k = 33");
    __shared__ float4 tile[TILE * F4];              // the patch's rows in SORTED order
    __shared__ __attribute__((aligned(16))) float rps[TILE];
    __shared__ int orig[TILE];                      // original row of a sorted position
    __shared__ uint32_t crange[2 * NCH];            // t-range of each chunk, mono(): [c] = min, [NCH + c] = max
    __shared__ float cbound[NW][2 * NCH];           // per wave: [c] = min t over chunks >= c, [NCH + c] = max t over chunks <= c
    __shared__ uint32_t wrq[NW];                    // per wave: mono(max |x_i|^2)
#ifdef KG_TRACE
    const long long kt_entry = __builtin_readcyclecounter();
#endif
    if (a.uws && a.uws[0] != 0)
        return;
    const int b = blockIdx.y;
    const int n = a.n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwaves = blockDim.x >> 6;             // blockDim.x = ceil(n / 64) * 64
    const bool live = tid < n;
    const float *X = a.query + (size_t)b * n * a.c;
    const int IB = 32 - __clz(n - 1);
    const int keep = ~((1 << IB) - 1);

    // ---- the lane's query = the row at sorted position tid (knn_slab_order_kernel), staged at that position ----
    const int qi = tid;
    const int32_t *scr = (const int32_t *)a.idx + (size_t)b * n * K;
    const float tq = live ? __int_as_float(scr[tid]) : 0.f;
    const int my_row = live ? min(max(scr[n + tid], 0), n - 1) : 0;
    float q[C], rq;
    load_query<C>(q, rq, X + (size_t)my_row * a.c, a.c, live);
    {
#pragma unroll
        for (int i = 0; i < C / 4; ++i)
            tile[tid * F4 + i] = make_float4(q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]);   // (dead: zeros)
        const float r = live ? rq : __builtin_inff();       // pad rows n .. blockDim-1: never among the nearest
        tile[tid * F4 + C / 4] = make_float4(r, 0.f, 0.f, 0.f);
        rps[tid] = r;
        orig[tid] = my_row;
        // t-range of every chunk; max |x|^2 of the patch
        const uint32_t mt = tpu3_mono(tq);
        uint32_t lo = tpu3_row_min_u32(live ? mt : 0xFFFFFFFFu), hi = tpu3_row_max_u32(live ? mt : 0u);
        lo = min(lo, (uint32_t)tpu3_dpp<0x142, 0xA>((int)lo));     // rows 1, 3 += rows 0, 2: lanes 31 / 63 hold a chunk
        hi = max(hi, (uint32_t)tpu3_dpp<0x142, 0xA>((int)hi));
        if ((lane & 31) == 31) {
            crange[tid >> 5] = lo;                                 // (an all-pad chunk: neutral for the min / max below)
            crange[NCH + (tid >> 5)] = hi;
        }
        const uint32_t wm = tpu3_wave_max_u32(live ? tpu3_mono(rq) : 0u);
        if (lane == 0)
            wrq[wave] = wm;
    }
    __syncthreads();
    uint32_t mm = wrq[0];
    for (int w = 1; w < nwaves; ++w)
        mm = max(mm, wrq[w]);
    const float M = tpu3_unmono(mm);                // max |x_i|^2 over the patch
    const int nch = (n + L - 1) / L;
    // (r6) The closing test below shuts a side at chunk c for c AND every chunk beyond it, so the range it tests must
    // cover all of them.  The pre-pass only BINS the rows (order inside a bin = arrival order of its atomics): a chunk
    // that lies wholly inside one bin can have a larger minimum t than a later chunk of the same bin (a dense cluster
    // narrower than a bin; tools/knn_slab_bound_sim.py, test_knn_graph_slab_form_cluster_inside_one_bin).  So the table
    // holds, for the right side, the minimum over chunks c .. nch-1 and, for the left side, the maximum over chunks
    // 0 .. c: monotone whatever the order inside the bins is.  (For a chunk on the right of the wave's own the left-side
    // term t_q - max is <= 0, the own chunks being part of the prefix, and vice versa: the max of the two below is
    // always the bound of the side the walk is on.)  Each wave builds its own copy: no second workgroup barrier.
    {
        uint32_t slo = 0xFFFFFFFFu, phi = 0u;
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) {
            if (cc < nch) {
                const uint32_t l = crange[cc], h = crange[NCH + cc];
#ifdef KG_SLAB_OWN_TABLE        // (A/B builds only, tools/_ab: the r5 table -- each chunk's own range -- to show the test catches it)
                slo = min(slo, cc == lane ? l : 0xFFFFFFFFu);
                phi = max(phi, cc == lane ? h : 0u);
#else
                slo = min(slo, cc >= lane ? l : 0xFFFFFFFFu);
                phi = max(phi, cc <= lane ? h : 0u);
#endif
            }
        }
        if (lane < NCH) {
            cbound[wave][lane] = tpu3_unmono(slo);
            cbound[wave][NCH + lane] = tpu3_unmono(phi);
        }
        __builtin_amdgcn_wave_barrier();
    }
    const float *cb = cbound[wave];
    // margins of the bound (DESIGN / docs): |t - <x, v>| <= 2^-19.4 sqrt(M) per row, the expanded-form distance is
    // within 2^-17.3 M of the true one; E1, E2 are 10x / 5x those
    const float E1 = 3.0517578125e-05f * __builtin_amdgcn_sqrtf(M), E2 = 3.0517578125e-05f * M;

    int lst[L], e = 0x7FFFFFFF, e2 = 0x7FFFFFFF;
#pragma unroll
    for (int i = 0; i < L; ++i)
        lst[i] = 0x7FFFFFFF;
    int lo_c = 2 * wave - 1, hi_c = 2 * wave + 2;
    int side = 0;
#ifdef KG_TRACE
    const long long kt0 = __builtin_readcyclecounter();
    int kt_dist = 0, kt_net = 0;
#endif
    for (int step = 0;; ++step) {
        int c;
        const bool own = step < 2;
        bool try_skip = false;
        if (own) {
            c = 2 * wave + step;
        } else {
            const bool lopen = lo_c >= 0, hopen = hi_c < nch;
            if (!lopen && !hopen)
                break;
            const bool left = lopen && (!hopen || side == 0);
            side ^= 1;
            c = left ? lo_c : hi_c;
            // the bound: every row of chunk c (and beyond) is farther than the lane's 32nd key, for EVERY lane?
            const float gap = __builtin_fmaxf(cb[c] - tq, tq - cb[NCH + c]) - E1;
            const float tau = __int_as_float(lst[L - 1] | ~keep);      // top of the 32nd key's truncation bucket (NaN while the list is short)
            const bool out = !live || (gap > 0.f && __builtin_fmaf(gap, gap, -E2) > tau);
            const uint64_t outs = __builtin_amdgcn_ballot_w64(out);
            if (outs == ~0ull) {
                if (left) lo_c = -1; else hi_c = nch;
                continue;
            }
            if (left) --lo_c; else ++hi_c;
            // the key test below pays only where the bound already rules out a good share of the lanes (a quarter:
            // 55 % of the visited chunks, 96 % of the chunks it would skip -- tools/knn_accept_sim.py)
            try_skip = __builtin_popcountll(outs) >= 16;
        }
        const int j = c * L;
        float d[L];
#ifdef KG_TRACE
        ++kt_dist;
#endif
#pragma unroll
        for (int u = 0; u < L; u += 16)
            kg_dist<C, 4>(tile, rps, j + u, q, rq, d + u);
        int nw[L];
#pragma unroll
        for (int u = 0; u < L; ++u)
            nw[u] = (__float_as_int(d[u]) & keep) | (j + u);
        if (own) {
#pragma unroll
            for (int u = 0; u < L; ++u)
                nw[u] = j + u == qi ? 0x7FFFFFFF : nw[u];
        } else if (try_skip) {
            // no lane's smallest new key reaches its list: the sorting network is skipped
            int m0 = nw[0], m1 = nw[1], m2 = nw[2], m3 = nw[3];
#pragma unroll
            for (int u = 4; u < L; u += 4) {
                m0 = min(m0, nw[u]), m1 = min(m1, nw[u + 1]);
                m2 = min(m2, nw[u + 2]), m3 = min(m3, nw[u + 3]);
            }
            const int mn = min(min(m0, m1), min(m2, m3));
            if (__builtin_amdgcn_ballot_w64(!live || mn > (lst[L - 1] | ~keep)) == ~0ull)
                continue;
        }
        // (taking the wave's first chunk -- empty list -- through the sorting network alone was measured: no gain, the
        // second code path costs five spilled registers)
#ifdef KG_TRACE
        ++kt_net;
#endif
        kg_fold_i32<L>(lst, e, e2, nw);
    }
#ifdef KG_TRACE
    if (g_kg_trace && lane == 0) {
        long long *t = g_kg_trace + ((size_t)b * 8 + wave) * 4;
        t[0] = __builtin_readcyclecounter() - kt0; t[1] = kt_dist; t[2] = kt_net; t[3] = kt0 - kt_entry;
    }
    const long long kt_loop_end = __builtin_readcyclecounter();
#endif
    // ---- boundary collisions after truncation: as knn_graph_key_kernel, ties by the ORIGINAL index ------------
    bool redo = live && (lst[0] >> IB) <= 0;
    const int T = lst[L - 1] >> IB;
    bool amb = live && !redo && (e >> IB) == T && e != 0x7FFFFFFF;
    if (__builtin_amdgcn_ballot_w64(amb)) {
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < L; ++i)
            cnt += (lst[i] >> IB) == T ? 1 : 0;
        const bool easy = amb && (e2 >> IB) != T && cnt <= 2;
        if (easy) {
            const int ja = lst[L - 1] & ~keep, jb = lst[L - 2] & ~keep, jc = e & ~keep;
            const int oa = orig[ja], ob = orig[jb], oc = orig[jc];
            __builtin_amdgcn_sched_barrier(0);
            const int da = __float_as_int(row_dist<C>(tile + ja * F4, q, rq));
            __builtin_amdgcn_sched_barrier(0);
            const int dc = __float_as_int(row_dist<C>(tile + jc * F4, q, rq));
            __builtin_amdgcn_sched_barrier(0);
            auto less = [](int d0, int j0, int d1, int j1) { return d0 < d1 || (d0 == d1 && j0 < j1); };
            if (cnt == 1) {
                if (less(dc, oc, da, oa))
                    lst[L - 1] = (dc & keep) | jc;
            } else {
                const int db = __float_as_int(row_dist<C>(tile + jb * F4, q, rq));
                const bool a_worst = !less(da, oa, db, ob) && !less(da, oa, dc, oc);
                const bool b_worst = !a_worst && !less(db, ob, dc, oc);
                if (a_worst)
                    lst[L - 1] = (dc & keep) | jc;
                else if (b_worst)
                    lst[L - 2] = (dc & keep) | jc;
            }
            amb = false;
        }
    }
    if (__builtin_amdgcn_ballot_w64(amb)) {
        // second sweep, wave-uniform: among the rows with truncated distance T the two smallest (D, original index)
        int b0 = 0x7FFFFFFF, b1 = 0x7FFFFFFF, p0 = 0, p1 = 0, o0 = 0x7FFFFFFF, o1 = 0x7FFFFFFF;
        for (int j = 0; j < nch * L; j += 16) {
            float d[16];
            kg_dist<C, 4>(tile, rps, j, q, rq, d);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int pos = j + u;
                const int bits = __float_as_int(d[u]);
                if ((bits >> IB) == T && pos != qi) {
                    const int o = orig[pos];
                    if (bits < b0 || (bits == b0 && o < o0)) {
                        b1 = b0; p1 = p0; o1 = o0; b0 = bits; p0 = pos; o0 = o;
                    } else if (bits < b1 || (bits == b1 && o < o1)) {
                        b1 = bits; p1 = pos; o1 = o;
                    }
                }
            }
        }
        if (amb) {
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < L; ++i)
                cnt += (lst[i] >> IB) == T ? 1 : 0;
            if (cnt > 2) {
                redo = true;
            } else {
                lst[L - 1] = ((cnt == 1 ? b0 : b1) & keep) | (cnt == 1 ? p0 : p1);
                if (cnt == 2)
                    lst[L - 2] = (b0 & keep) | p0;
            }
        }
    }
    if (redo)
        a.uws[2] = 1u;
    if (live) {
        int32_t *out = (int32_t *)a.idx + ((size_t)b * n + my_row) * K;
        out[0] = my_row;
#pragma unroll
        for (int i = 0; i < L; ++i)
            out[1 + i] = orig[lst[i] & ~keep];
    }
#ifdef KG_TRACE
    if (g_kg_trace && lane == 0) {
        g_kg_trace[((size_t)b * 8 + 5 + (wave & 1)) * 4 + (wave >> 1)] = __builtin_readcyclecounter() - kt_loop_end;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        // second block of the trace buffer: [patch][wave] -> (entry clock, exit clock, hw id, xcc id)
        long long *t2 = g_kg_trace + (size_t)a.b * 32 + ((size_t)b * 8 + wave) * 4;
        t2[0] = kt_entry; t2[1] = __builtin_readcyclecounter(); t2[2] = hw; t2[3] = xcc;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// any k: workgroup-per-query bitonic selection
// ---------------------------------------------------------------------------------------------
template <int LOG2S>
__global__ __launch_bounds__(1024) void knn_sort_kernel(KnnArgs a)
{
    constexpr int S = 1 << LOG2S;
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];     // S keys (up to 128 KiB)
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
// 64 < k <= 512 of at most 3072 candidates: one WAVE per query, selection by bisection
// ---------------------------------------------------------------------------------------------
// The patch extraction of every level asks for the k = 312 nearest of 624 .. 2496 points (upsampler.py:59-86).
// Sorting all candidates (the kernel above: 78 compare-exchange stages over 4096 LDS slots per query) does ~15x
// the necessary work.  Here a wave keeps the query's candidate keys in registers (mono(D), candidate
// j = 64 p + lane), finds the k-th smallest key VALUE by bisection over the 32 key bits (count(key < trial) per
// step: a compare and an add per register, one DPP reduction), compacts the selected ones -- everything below
// the threshold plus the first ties in index order, the oracle's rule -- into LDS and sorts just those
// (S = 512 slots for k = 312: 45 stages over 256 pairs).  Same distance arithmetic, same (D, j) order, same
// treatment of dead slots as knn_sort_kernel.
__device__ __forceinline__ int ks_wave_sum_i32(int v)
{
#define KS_DPP(CTRL, RM) v += __builtin_amdgcn_update_dpp(0, v, CTRL, RM, 0xF, false)
    KS_DPP(0xB1, 0xF);
    KS_DPP(0x4E, 0xF);
    KS_DPP(0x141, 0xF);
    KS_DPP(0x140, 0xF);
    KS_DPP(0x142, 0xA);
    KS_DPP(0x143, 0xC);
#undef KS_DPP
    return __builtin_amdgcn_readlane(v, 63);
}

constexpr int KS_WAVES = 4;
constexpr int KS_SMAX = 512;

template <int PT>
__global__ __launch_bounds__(64 * KS_WAVES) void knn_select_kernel(KnnArgs a, int S)
{
    __shared__ uint64_t slots[KS_WAVES][KS_SMAX];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y, qi0 = blockIdx.x * KS_WAVES + wave;
    const int pb = a.pts_of ? a.pts_of[b] : b;
    const int n = a.n_arr ? a.n_arr[pb] : a.n;
    const int m = a.m_arr ? a.m_arr[b] : a.m;
    const bool live = qi0 < m;                      // (dead waves run along: the sort below uses barriers)
    const int qi = live ? qi0 : 0;
    const int c = a.c, k = a.k;
    const bool use_dup = a.dup != nullptr && a.uws[0] != 0;
    const float dmax = use_dup ? tpu3_unmono(a.uws[4 + (a.grp ? a.grp[b] : 0)]) : 0.f;
    const float *P = a.points + (size_t)pb * a.n * c;
    const float *Q = a.query + ((size_t)b * a.m + qi) * c;
    const uint8_t *DUP = use_dup ? a.dup + (size_t)pb * a.n : nullptr;
    uint64_t *K = slots[wave];
    float rq = 0.f;
    for (int i = 0; i < c; ++i)
        rq = __builtin_fmaf(Q[i], Q[i], rq);

    uint32_t key[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int j = p * 64 + lane;
        key[p] = 0xFFFFFFFFu;
        if (j < n) {
            const float *pr = P + (size_t)j * c;
            float dot = 0.f, rp = 0.f;
            for (int i = 0; i < c; ++i) {
                const float v = pr[i];
                dot = __builtin_fmaf(Q[i], v, dot);
                rp = __builtin_fmaf(v, v, rp);
            }
            float d = __builtin_fmaf(-2.f, dot, rq) + rp;
            if (use_dup)
                d = d + dmax * (float)DUP[j];
            key[p] = tpu3_mono(d);
        }
    }
    // T = the k-th smallest key: the largest value with fewer than k keys below it
    uint32_t T = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t trial = T | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int p = 0; p < PT; ++p)
            cnt += key[p] < trial ? 1 : 0;
        if (ks_wave_sum_i32(cnt) < k)
            T = trial;
    }
    int c_less = 0;
#pragma unroll
    for (int p = 0; p < PT; ++p)
        c_less += key[p] < T ? 1 : 0;
    c_less = ks_wave_sum_i32(c_less);
    const int need = k - c_less;                    // ties taken, lowest candidate index first
    int base_less = 0, base_tie = 0;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const bool lt = key[p] < T, eq = key[p] == T;
        const uint64_t ml = __builtin_amdgcn_ballot_w64(lt), me = __builtin_amdgcn_ballot_w64(eq);
        const int pl = __builtin_amdgcn_mbcnt_hi((uint32_t)(ml >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ml, 0));
        const int pe = __builtin_amdgcn_mbcnt_hi((uint32_t)(me >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)me, 0));
        const int j = p * 64 + lane;
        // dead slots (j >= n) carry the all-ones key of knn_sort_kernel: index -1, distance unmono(~0)
        const uint64_t packed = j < n ? ((uint64_t)key[p] << 32) | (uint32_t)j : ~0ull;
        if (lt)
            K[base_less + pl] = packed;
        if (eq && base_tie + pe < need)
            K[c_less + base_tie + pe] = packed;
        base_less += __builtin_popcountll(ml);
        base_tie += __builtin_popcountll(me);
    }
    for (int s0 = k + lane; s0 < S; s0 += 64)
        K[s0] = ~0ull;
    // bitonic sort of the wave's S slots, ascending (D, j)
    for (int size = 2; size <= S; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int u = lane; u < S / 2; u += 64) {
                const int lo = 2 * u - (u & (stride - 1));
                const int hi = lo + stride;
                const bool asc = (lo & size) == 0;
                const uint64_t x = K[lo], y = K[hi];
                if ((x > y) == asc) {
                    K[lo] = y;
                    K[hi] = x;
                }
            }
        }
    __syncthreads();
    if (!live)
        return;
    const size_t o = ((size_t)b * a.m + qi) * k;
    for (int i = lane; i < k; i += 64) {
        const uint64_t kk = K[i];
        store_idx(a, o + i, (int)(uint32_t)kk);
        if (a.dist)
            a.dist[o + i] = tpu3_unmono((uint32_t)(kk >> 32));
    }
}

template <int PT>
int launch_select(hipStream_t s, int b, const KnnArgs &a)
{
    int S = 128;
    while (S < a.k)
        S <<= 1;
    hipLaunchKernelGGL((knn_select_kernel<PT>), dim3((a.m + KS_WAVES - 1) / KS_WAVES, b), dim3(64 * KS_WAVES), 0, s,
                       a, S);
    return tpu3_launch_status();
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
    if (uws[a.gate] == 0)
        return;
    constexpr int TILE = tile_rows(C);
    constexpr int F4 = Row<C>::F4;
    __shared__ float4 tile[TILE * F4];
    __shared__ uint32_t red[16];
    const int total = a.nblk_x * a.b;
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
    const int b = w / a.nblk_x;
    const int pb = a.pts_of ? a.pts_of[b] : b;
    const int n = a.n_arr ? a.n_arr[pb] : a.n;
    const int m = a.m_arr ? a.m_arr[b] : a.m;
    const int qi = (w - b * a.nblk_x) * blockDim.x + threadIdx.x;
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
    __syncthreads();
    }   // work items
}

__global__ __launch_bounds__(256) void knn_dmax_generic_kernel(KnnArgs a, uint32_t *uws)
{
    // any channel count: one lane per query, points read straight from global memory (L2)
    if (uws[a.gate] == 0)
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

// ---- first-occurrence mask in O(n): open-addressing table of equivalence-class representatives -----
// Phase 1: every row probes from hash(row); an empty slot is claimed with atomicCAS, a slot whose
// representative equals the row is lowered to the smaller index with atomicMin.  Rows of one class
// share their hash, hence their probe sequence, hence their slot.  Phase 2 (next launch): the class
// slot holds the lowest index of the class; every other member is a duplicate.  Deterministic.
__device__ __forceinline__ bool knn_rows_equal(const float *__restrict__ a, const float *__restrict__ b, int c)
{
    for (int ch = 0; ch < c; ++ch)
        if (!(a[ch] == b[ch]))
            return false;
    return true;
}
__device__ __forceinline__ uint32_t knn_row_hash(const float *__restrict__ r, int c)
{
    uint32_t h = 0x9E3779B9u;
    for (int ch = 0; ch < c; ++ch) {
        uint32_t u = __float_as_uint(r[ch] + 0.0f);       // -0.0 == +0.0 must hash alike
        h ^= u + 0x9E3779B9u + (h << 6) + (h >> 2);
        h *= 0x85EBCA6Bu;
    }
    return h ^ (h >> 15);
}
// gate_word >= 0: the kernel runs only if uws[gate_word] != 0 (then launched with a small grid
// that strides over the (block, point set) work items)
__global__ __launch_bounds__(256) void knn_dup_hash_insert_kernel(int n_pad, int c, int tsize,
                                                                  const float *__restrict__ points,
                                                                  const int32_t *__restrict__ n_arr,
                                                                  uint32_t *__restrict__ table, int bp, int nblk,
                                                                  const uint32_t *__restrict__ uws, int gate_word)
{
    if (gate_word >= 0 && uws[gate_word] == 0)
        return;
    for (int w = blockIdx.x; w < bp * nblk; w += gridDim.x) {
    const int b = w / nblk;
    const int n = n_arr ? n_arr[b] : n_pad;
    const int i = (w - b * nblk) * blockDim.x + threadIdx.x;
    if (i >= n)
        continue;
    const float *P = points + (size_t)b * n_pad * c;
    uint32_t *T = table + (size_t)b * tsize;
    const float *row = P + (size_t)i * c;
    uint32_t s = knn_row_hash(row, c) & (uint32_t)(tsize - 1);
    for (int probes = 0; probes < tsize; ++probes) {
        uint32_t o = __hip_atomic_load(T + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (o == 0xFFFFFFFFu) {
            o = atomicCAS(T + s, 0xFFFFFFFFu, (uint32_t)i);
            if (o == 0xFFFFFFFFu)
                break;
        }
        if (knn_rows_equal(P + (size_t)o * c, row, c)) {
            atomicMin(T + s, (uint32_t)i);
            break;
        }
        s = (s + 1) & (uint32_t)(tsize - 1);
    }
    }   // work items
}
__global__ __launch_bounds__(256) void knn_dup_hash_lookup_kernel(int n_pad, int c, int tsize,
                                                                  const float *__restrict__ points,
                                                                  const int32_t *__restrict__ n_arr,
                                                                  const uint32_t *__restrict__ table,
                                                                  uint8_t *__restrict__ dup, uint32_t *__restrict__ uws,
                                                                  int bp, int nblk, int gate_word)
{
    if (gate_word >= 0 && uws[gate_word] == 0)
        return;
    for (int w = blockIdx.x; w < bp * nblk; w += gridDim.x) {
    const int b = w / nblk;
    const int n = n_arr ? n_arr[b] : n_pad;
    const int i = (w - b * nblk) * blockDim.x + threadIdx.x;
    if (i >= n)
        continue;
    const float *P = points + (size_t)b * n_pad * c;
    const uint32_t *T = table + (size_t)b * tsize;
    const float *row = P + (size_t)i * c;
    uint32_t s = knn_row_hash(row, c) & (uint32_t)(tsize - 1);
    uint8_t d = 0;
    for (int probes = 0; probes < tsize; ++probes) {
        const uint32_t o = T[s];
        if (o == 0xFFFFFFFFu)
            break;                                  // rows with a NaN never match anything, incl. themselves
        if (o == (uint32_t)i)
            break;
        if (knn_rows_equal(P + (size_t)o * c, row, c)) {
            d = 1;
            break;
        }
        s = (s + 1) & (uint32_t)(tsize - 1);
    }
    dup[(size_t)b * n_pad + i] = d;
    if (d)
        uws[0] = 1u;
    }   // work items
}

// Both phases for ONE point set per workgroup with the table in LDS (sets of up to 16 384 rows: the merged
// previous-level patches the inter-level search runs over have 624 .. 6240).  Same slots, same result as the two
// kernels above; their device-scope atomics on a table in global memory are executed memory-side on a
// multi-XCD part (~6 G/s for the 2.4 M of a level-4 call: 0.48 ms), LDS atomics are not.
__global__ __launch_bounds__(1024) void knn_dup_lds_kernel(int n_pad, int c, int tsize, const float *__restrict__ points,
                                                           const int32_t *__restrict__ n_arr,
                                                           uint8_t *__restrict__ dup, uint32_t *__restrict__ uws)
{
    extern __shared__ uint32_t lds_table[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = n_arr ? n_arr[b] : n_pad;
    const uint32_t mask = (uint32_t)(tsize - 1);
    for (int s = tid; s < tsize; s += 1024)
        lds_table[s] = 0xFFFFFFFFu;
    __syncthreads();
    const float *P = points + (size_t)b * n_pad * c;
    for (int i = tid; i < n; i += 1024) {
        const float *row = P + (size_t)i * c;
        uint32_t s = knn_row_hash(row, c) & mask;
        for (int probes = 0; probes < tsize; ++probes) {
            uint32_t o = __hip_atomic_load(lds_table + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (o == 0xFFFFFFFFu) {
                o = atomicCAS(lds_table + s, 0xFFFFFFFFu, (uint32_t)i);
                if (o == 0xFFFFFFFFu)
                    break;
            }
            if (knn_rows_equal(P + (size_t)o * c, row, c)) {
                atomicMin(lds_table + s, (uint32_t)i);
                break;
            }
            s = (s + 1) & mask;
        }
    }
    __syncthreads();
    bool any = false;
    for (int i = tid; i < n; i += 1024) {
        const float *row = P + (size_t)i * c;
        uint32_t s = knn_row_hash(row, c) & mask;
        uint8_t d = 0;
        for (int probes = 0; probes < tsize; ++probes) {
            const uint32_t o = lds_table[s];
            if (o == 0xFFFFFFFFu || o == (uint32_t)i)
                break;                              // (rows with a NaN never match anything, incl. themselves)
            if (knn_rows_equal(P + (size_t)o * c, row, c)) {
                d = 1;
                break;
            }
            s = (s + 1) & mask;
        }
        dup[(size_t)b * n_pad + i] = d;
        any = any || d;
    }
    if (any)
        uws[0] = 1u;
}

// cand[b, :count[b]] = ascending indices of the rows of point set b that are not duplicates
__global__ __launch_bounds__(256) void knn_compact_kernel(int n_pad, const int32_t *__restrict__ n_arr,
                                                          const uint8_t *__restrict__ dup,
                                                          const uint32_t *__restrict__ uws,
                                                          int32_t *__restrict__ cand, int32_t *__restrict__ count)
{
    __shared__ int wtot[4];
    if (uws[0] == 0)
        return;                 // nothing duplicated anywhere: the list is not consulted
    const int b = blockIdx.x;
    const int n = n_arr ? n_arr[b] : n_pad;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int i = j0 + threadIdx.x;
        const bool keep = i < n && dup[(size_t)b * n_pad + i] == 0;
        const unsigned long long mask = __ballot(keep);
        const int below = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0)
            wtot[wave] = __builtin_popcountll(mask);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w)
            off += wtot[w];
        if (keep)
            cand[(size_t)b * n_pad + off + below] = i;
        base += wtot[0] + wtot[1] + wtot[2] + wtot[3];
    }
    if (threadIdx.x == 0)
        count[b] = base;
}

__global__ __launch_bounds__(256) void knn_fill_gated_kernel(uint32_t *__restrict__ p, size_t words,
                                                             const uint32_t *__restrict__ uws, int gate_word)
{
    if (uws[gate_word] == 0)
        return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x)
        p[i] = 0xFFFFFFFFu;
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

constexpr int KNN_GATED_GRID = 1024;

template <int C, int KMAX>
int launch_insert(hipStream_t s, int b, const KnnArgs &a0)
{
    KnnArgs a = a0;
    int threads = ((a.m + 63) / 64) * 64;
    if (threads > 512)
        threads = 256;
    a.b = b;
    a.nblk_x = (a.m + threads - 1) / threads;
    long grid = (long)a.nblk_x * b;
    if (a.gate && grid > KNN_GATED_GRID)
        grid = KNN_GATED_GRID;
    hipLaunchKernelGGL((knn_insert_kernel<C, KMAX>), dim3((unsigned)grid), dim3(threads), 0, s, a);
    return tpu3_launch_status();
}

template <int C>
int dispatch_insert_k(hipStream_t s, int b, const KnnArgs &a)
{
    if (a.k <= 2) return launch_insert<C, 2>(s, b, a);
    if (C == 3 && a.k <= 5) return launch_insert<C, 5>(s, b, a);    // the inter-level search (fm_knn = 5)
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
    const size_t lds = (size_t)S * sizeof(uint64_t);
    const hipError_t e = hipFuncSetAttribute((const void *)knn_sort_kernel<LOG2S>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((knn_sort_kernel<LOG2S>), dim3(a.m, b), dim3(threads), lds, s, a);
    return tpu3_launch_status();
}

int dispatch_sort(hipStream_t s, int b, const KnnArgs &a)
{
    // slots: at least 2k (so every pass makes progress) and, if it fits, the whole candidate
    // set in one pass; 8192 keys = 64 KiB of LDS, 16 384 only when k asks for it (the training
    // data path extracts label patches of 16 x 312 = 4992 points)
    long need = a.n < 2L * a.k ? 2L * a.k : a.n;
    if (2L * a.k > 16384) return TPU3_ELIMIT;
    if (a.k <= KS_SMAX && a.n <= 64 * 48) {         // one wave per query (knn_select_kernel)
        if (a.n <= 64 * 10) return launch_select<10>(s, b, a);
        if (a.n <= 64 * 20) return launch_select<20>(s, b, a);
        if (a.n <= 64 * 32) return launch_select<32>(s, b, a);
        if (a.n <= 64 * 40) return launch_select<40>(s, b, a);
        return launch_select<48>(s, b, a);
    }
    if (2L * a.k > 8192) return launch_sort<14>(s, b, a);
    if (need > 8192) need = 8192;
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

constexpr int KNN_DUP_HASH_MIN_N = 128;     // below this the quadratic kernel needs no table and is as fast

int knn_dup_table_size(int n)
{
    int t = 64;
    while (t < 2 * n)
        t <<= 1;
    return t;
}

int launch_dmax(hipStream_t s, int b, const KnnArgs &a0, uint32_t *uws)
{
    KnnArgs a = a0;
    int threads = ((a.m + 63) / 64) * 64;
    if (threads > 512) threads = 256;
    a.b = b;
    a.nblk_x = (a.m + threads - 1) / threads;
    long grid = (long)a.nblk_x * b;
    if (grid > KNN_GATED_GRID)
        grid = KNN_GATED_GRID;
    const dim3 g((unsigned)grid);
    const int c = a.c;
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
        hipLaunchKernelGGL(knn_dmax_generic_kernel, dim3((a.m + 255) / 256, b), dim3(256), 0, s, a, uws);
    return tpu3_launch_status();
}

} // namespace

extern "C" size_t tpu3_knn_unique_workspace_bytes(int bp, int n)
{
    if (bp <= 0 || n < KNN_DUP_HASH_MIN_N)
        return 0;
    return (size_t)bp * (size_t)knn_dup_table_size(n) * sizeof(uint32_t);
}

extern "C" int tpu3_knn_f32(tpu3_stream_t stream, int b, int m, int n, int c, int k,
                            const float *query, const float *points, const tpu3_knn_layout *layout,
                            const uint8_t *dup, uint32_t *uws, void *idx, int idx_elem_size,
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
    KnnArgs a{m, n, c, k, b, 1, query, points, L.n_arr, L.m_arr, L.pts_of, L.grp, dup, uws, 0, 0, idx,
              idx_elem_size == 8, dist};
    if ((L.cand == nullptr) != (L.cand_count == nullptr)) return TPU3_EINVAL;
    int r = -100;
    if (k <= 64 && c <= 32) {
        if (dup) {
            // optimistic pass (duplicates skipped + verified), then -- only if some query could not
            // verify -- max(D) and the reference arithmetic; both follow-ups early-exit on uws[1] == 0
            a.mode = 1;
            a.cand = L.cand; a.cand_count = L.cand_count;
            if (c == 3 && k <= 8 && L.cand && L.tile_pts && L.tile_idx && L.tile_box)
                r = tpu3_knn_tiles_query_f32(stream, b, m, n, k, query, &L, uws, idx, idx_elem_size, dist);   // csrc/knn_tiles.hip
            else
                r = dispatch_insert(s, b, a);
            if (r) return r;
            a.cand = nullptr; a.cand_count = nullptr;
            KnnArgs d = a;
            d.mode = 0; d.gate = 1; d.dup = nullptr;
            r = launch_dmax(s, b, d, uws);
            if (r) return r;
            a.mode = 0; a.gate = 1;
            r = dispatch_insert(s, b, a);
        } else {
            r = dispatch_insert(s, b, a);
        }
    } else {
        if (dup) {      // selection by sorting: max(D) whenever anything is duplicated
            KnnArgs d = a;
            d.gate = 0; d.dup = nullptr;
            r = launch_dmax(s, b, d, uws);      // gate word 0 = the any-dup flag itself
            if (r) return r;
        }
        r = dispatch_sort(s, b, a);
    }
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

// kNN graph for the fused DenseEdgeConv (see knn_graph_kernel): idx (b,m,k) i32, slot 0 = nearest,
// slots 1..k-1 = the other members of the exact top-k set in index order.  unique=True semantics:
// dup/uws from tpu3_knn_unique_prepare_f32; when any row is duplicated the exact sorted kernels run
// instead (device-side gate, no host synchronisation).  Supported: c <= 32, k in {17, 33}; other
// sizes return TPU3_ELIMIT (callers then use tpu3_knn_f32).
// Self kNN graph without a de-duplication pre-pass (see knn_graph_kernel, mode 2): x (b,n,c) is both
// query and point set; dup (b,n) and uws are scratch owned by the call (uws is zeroed here); workspace
// = tpu3_knn_unique_workspace_bytes(b, n) bytes for the (rarely needed) hash tables.
namespace {
// The self graph's first (usually only) pass: one pass with the index in the key's low bits (n up to 2^13: >= 10
// mantissa bits stay), the two-pass kernel beyond that and for n == k.  Either raises uws[2] when rows may be
// duplicated.
// Threads per workgroup of the self graph: the patch's queries in one workgroup (312 points: five waves share one
// staged tile) when the launch fills the chip; ONE WAVE per workgroup (r4) when all waves of the launch find a SIMD
// of their own -- a training batch (32 patches) or one cloud's 48 outer patches used to be 32 / 48 workgroups whose
// five waves shared four SIMDs of one compute unit while most of the chip idled (82 -> 63 us per call).
// TPU3_KG_THREADS: tuning hook.
long g_kg_slab_launches = 0;     // tpu3_debug_knn_slab_launches: the tests assert that they exercise the slab form

int kg_graph_threads(int b, int n)
{
    static const int forced = getenv("TPU3_KG_THREADS") ? atoi(getenv("TPU3_KG_THREADS")) : 0;
    static const int cus = []() { int d = 0, v = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d); return v; }();
    if (forced > 0)
        return forced;
    int threads = ((n + 63) / 64) * 64;
    if (threads > 512) threads = 256;
    if ((long)b * ((n + 63) / 64) <= 8L * cus)      // (384 patches: 0.124 vs 0.142 ms; 768: 0.207 vs 0.173)
        threads = 64;
    return threads;
}

int launch_graph_first_pass(hipStream_t s, dim3 g, int threads, const KnnArgs &a)
{
    const int c = a.c, k = a.k;
#define KG(CC, KK) hipLaunchKernelGGL((knn_graph_kernel<CC, KK>), g, dim3(threads), 0, s, a)
#define KK1(CC, KK) hipLaunchKernelGGL((knn_graph_key_kernel<CC, KK>), g, dim3(threads), 0, s, a)
    const bool onepass = a.n > k && a.n <= 8192;
    // (r5) one patch per workgroup, ordered along its principal direction, far chunks skipped: see knn_graph_slab_kernel
    static const bool slab_on = !(getenv("TPU3_KG_SLAB") && atoi(getenv("TPU3_KG_SLAB")) == 0);
    if (slab_on && onepass && k == 33 && c == 24 && ((uintptr_t)a.query & 15) == 0 && a.n > 64 && a.n <= 320 && (int)g.x * threads >= a.n
        && threads == ((a.n + 63) / 64) * 64) {
        hipLaunchKernelGGL((knn_slab_order_kernel<24>), g, dim3(threads), 0, s, a);
        hipLaunchKernelGGL((knn_graph_slab_kernel<24, 33>), g, dim3(threads), 0, s, a);
        ++g_kg_slab_launches;
        return tpu3_launch_status();
    }
    if (k == 33) {
        if (c == 3) { if (onepass) KK1(3, 33); else KG(3, 33); }
        else if (c <= 8) { if (onepass) KK1(8, 33); else KG(8, 33); }
        else if (c <= 16) { if (onepass) KK1(16, 33); else KG(16, 33); }
        else if (c <= 24) { if (onepass) KK1(24, 33); else KG(24, 33); }
        else { if (onepass) KK1(32, 33); else KG(32, 33); }
    } else {
        if (c == 3) { if (onepass) KK1(3, 17); else KG(3, 17); }
        else if (c <= 8) { if (onepass) KK1(8, 17); else KG(8, 17); }
        else if (c <= 16) { if (onepass) KK1(16, 17); else KG(16, 17); }
        else if (c <= 24) { if (onepass) KK1(24, 17); else KG(24, 17); }
        else { if (onepass) KK1(32, 17); else KG(32, 17); }
    }
#undef KG
#undef KK1
    return tpu3_launch_status();
}
} // namespace

extern "C" long tpu3_debug_knn_slab_launches(int reset)
{
    const long v = g_kg_slab_launches;
    if (reset) g_kg_slab_launches = 0;
    return v;
}

extern "C" int tpu3_knn_graph_self_f32(tpu3_stream_t stream, int b, int n, int c, int k, const float *x,
                                       const tpu3_knn_layout *layout, uint8_t *dup, uint32_t *uws, int32_t *idx,
                                       void *workspace, size_t workspace_bytes)
{
    if (bad_dims(b, n, n, c, k)) return TPU3_EINVAL;
    if (c > 32 || (k != 17 && k != 33)) return TPU3_ELIMIT;
    if (b == 0 || n == 0) return TPU3_OK;
    if (k > n) return TPU3_EINVAL;
    if (!x || !idx || !dup || !uws) return TPU3_EINVAL;
    if (b > 65535) return TPU3_ELIMIT;
    const tpu3_knn_layout L = layout ? *layout : tpu3_knn_layout{nullptr, nullptr, nullptr, nullptr, b, 1};
    if (L.pts_of || L.n_arr || L.m_arr) return TPU3_EINVAL;      // dense self query only
    const int groups = L.grp ? L.groups : 1;
    const size_t need = tpu3_knn_unique_workspace_bytes(b, n);
    if (need == 0 || !workspace || workspace_bytes < need) return TPU3_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(uws, 0, (size_t)TPU3_KNN_UWS_WORDS(groups) * sizeof(uint32_t), s);
    if (e != hipSuccess) return (int)e;
    KnnArgs a{n, n, c, k, b, 1, x, x, nullptr, nullptr, nullptr, L.grp, dup, uws, 2, 0, idx, 0, nullptr};
    const int threads = kg_graph_threads(b, n);
    const dim3 g((n + threads - 1) / threads, b);
    int r = launch_graph_first_pass(s, g, threads, a);
    if (r) return r;
    // everything below runs only if some query saw a zero distance to another row (uws[2])
    const int tsize = knn_dup_table_size(n);
    const int nblk = (n + 255) / 256;
    long grid = (long)nblk * b;
    if (grid > KNN_GATED_GRID) grid = KNN_GATED_GRID;
    hipLaunchKernelGGL(knn_fill_gated_kernel, dim3(KNN_GATED_GRID), dim3(256), 0, s, (uint32_t *)workspace,
                       need / sizeof(uint32_t), (const uint32_t *)uws, 2);
    hipLaunchKernelGGL(knn_dup_hash_insert_kernel, dim3((unsigned)grid), dim3(256), 0, s, n, c, tsize, x, nullptr,
                       (uint32_t *)workspace, b, nblk, (const uint32_t *)uws, 2);
    hipLaunchKernelGGL(knn_dup_hash_lookup_kernel, dim3((unsigned)grid), dim3(256), 0, s, n, c, tsize, x, nullptr,
                       (const uint32_t *)workspace, dup, uws, b, nblk, 2);
    r = tpu3_launch_status();
    if (r) return r;
    // real duplicates (uws[0]): max(D) per group + the exact sorted kernel with the reference arithmetic
    KnnArgs d = a;
    d.mode = 0; d.gate = 0; d.dup = nullptr;
    r = launch_dmax(s, b, d, uws);
    if (r) return r;
    // (r5) the exact sorted kernel runs whenever the first pass raised uws[2] -- duplicated rows, but also a computed
    // distance <= 0 to a DIFFERENT row (features far from the origin: the oracle's slot 0 is then not the query
    // itself) or a truncation collision three slots deep -- not only when the hash pass found duplicates.
    // COST (advisor, r5): the flag is one word for the whole launch, so a single undecidable row makes the brute-force
    // kernel recompute EVERY patch of the launch (3840 patches: ~2 ms instead of 0.4), and the hash / dup passes above run
    // even when only the collision raised it.  Correct, and rare on feature rows (0 events in every bench step and in
    // the 112-patch chained replays); a per-patch flag array would confine the recomputation -- not built: the
    // inference path uses the optimistic entry point, whose caller recomputes the whole call anyway.
    a.mode = 0; a.gate = 2;
    return dispatch_insert(s, b, a);
}

// Optimistic form of the self graph: ONLY the two-pass kernel, no de-duplication state, no gated launches
// behind it (five tiny launches per call that, on a busy GPU, each wait for a compute unit: ~0.8 ms of
// stream time apiece under the bench's eight streams).  If some query saw a second zero distance -- rows
// may be duplicated -- the kernel raises events[2]; the result of that call must then be recomputed by
// tpu3_knn_graph_self_f32.  `events`: 4 u32 device words zeroed once by the caller, shared by any number of
// calls (events[0] must stay 0); the caller inspects events[2] at its own synchronisation point.
extern "C" int tpu3_knn_graph_self_optimistic_f32(tpu3_stream_t stream, int b, int n, int c, int k, const float *x,
                                                  const tpu3_knn_layout *layout, uint32_t *events, int32_t *idx)
{
    if (bad_dims(b, n, n, c, k)) return TPU3_EINVAL;
    if (c > 32 || (k != 17 && k != 33)) return TPU3_ELIMIT;
    if (b == 0 || n == 0) return TPU3_OK;
    if (k > n) return TPU3_EINVAL;
    if (!x || !idx || !events) return TPU3_EINVAL;
    if (b > 65535) return TPU3_ELIMIT;
    const tpu3_knn_layout L = layout ? *layout : tpu3_knn_layout{nullptr, nullptr, nullptr, nullptr, b, 1};
    if (L.pts_of || L.n_arr || L.m_arr) return TPU3_EINVAL;      // dense self query only
    hipStream_t s = (hipStream_t)stream;
    KnnArgs a{n, n, c, k, b, 1, x, x, nullptr, nullptr, nullptr, L.grp, nullptr, events, 2, 0, idx, 0, nullptr};
    const int threads = kg_graph_threads(b, n);
    const dim3 g((n + threads - 1) / threads, b);
    return launch_graph_first_pass(s, g, threads, a);
}

extern "C" int tpu3_knn_graph_f32(tpu3_stream_t stream, int b, int m, int n, int c, int k, const float *query,
                                  const float *points, const tpu3_knn_layout *layout, const uint8_t *dup,
                                  uint32_t *uws, int32_t *idx)
{
    if (bad_dims(b, m, n, c, k)) return TPU3_EINVAL;
    if ((dup == nullptr) != (uws == nullptr)) return TPU3_EINVAL;
    if (c > 32 || (k != 17 && k != 33)) return TPU3_ELIMIT;
    if (b == 0 || m == 0) return TPU3_OK;
    if (k > n) return TPU3_EINVAL;
    if (!query || !points || !idx) return TPU3_EINVAL;
    if (b > 65535) return TPU3_ELIMIT;
    hipStream_t s = (hipStream_t)stream;
    const tpu3_knn_layout L = layout ? *layout : tpu3_knn_layout{nullptr, nullptr, nullptr, nullptr, b, 1};
    KnnArgs a{m, n, c, k, b, 1, query, points, L.n_arr, L.m_arr, L.pts_of, L.grp, dup, uws, 0, 0, idx, 0, nullptr};
    int threads = ((m + 63) / 64) * 64;
    if (threads > 512) threads = 256;
    const dim3 g((m + threads - 1) / threads, b);
#define KG(CC, KK) hipLaunchKernelGGL((knn_graph_kernel<CC, KK>), g, dim3(threads), 0, s, a)
    if (k == 33) {
        if (c == 3) KG(3, 33); else if (c <= 8) KG(8, 33); else if (c <= 16) KG(16, 33);
        else if (c <= 24) KG(24, 33); else KG(32, 33);
    } else {
        if (c == 3) KG(3, 17); else if (c <= 8) KG(8, 17); else if (c <= 16) KG(16, 17);
        else if (c <= 24) KG(24, 17); else KG(32, 17);
    }
#undef KG
    int r = tpu3_launch_status();
    if (r || !dup) return r;
    // duplicated rows present (uws[0] != 0): exact path, everything gated on the any-dup flag
    KnnArgs d = a;
    d.gate = 0; d.dup = nullptr;
    r = launch_dmax(s, b, d, uws);                  // runs only if uws[0] != 0
    if (r) return r;
    a.mode = 0; a.gate = -1;                        // gate -1: run only if uws[0] != 0
    a.idx64 = 0;
    return dispatch_insert(s, b, a);
}

extern "C" int tpu3_knn_unique_compact_i32(tpu3_stream_t stream, int bp, int n, const int32_t *n_arr,
                                           const uint8_t *dup, const uint32_t *uws, int32_t *cand,
                                           int32_t *cand_count)
{
    if (bp < 0 || n < 0) return TPU3_EINVAL;
    if (bp == 0 || n == 0) return TPU3_OK;
    if (!dup || !uws || !cand || !cand_count) return TPU3_EINVAL;
    hipLaunchKernelGGL(knn_compact_kernel, dim3(bp), dim3(256), 0, (hipStream_t)stream, n, n_arr, dup, uws, cand,
                       cand_count);
    return tpu3_launch_status();
}

extern "C" int tpu3_knn_unique_prepare_f32(tpu3_stream_t stream, int b, int m, int n, int c,
                                           const float *query, const float *points,
                                           const tpu3_knn_layout *layout, uint8_t *dup, uint32_t *uws,
                                           void *workspace, size_t workspace_bytes)
{
    (void)query;
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
    const size_t need = tpu3_knn_unique_workspace_bytes(bp, n);
    if (need == 0) {
        hipLaunchKernelGGL(knn_dup_kernel, dim3((n + 255) / 256, bp), dim3(256), 0, s, n, c, points, L.n_arr,
                           dup, uws);
        return tpu3_launch_status();
    }
    const int tsize = knn_dup_table_size(n);
    if ((size_t)tsize * sizeof(uint32_t) <= 128 * 1024) {       // table in LDS: one workgroup per point set
        const int lds = tsize * (int)sizeof(uint32_t);
        e = hipFuncSetAttribute((const void *)knn_dup_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(knn_dup_lds_kernel, dim3(bp), dim3(1024), lds, s, n, c, tsize, points, L.n_arr, dup, uws);
        return tpu3_launch_status();
    }
    void *ws = workspace;
    const bool own = !(workspace && workspace_bytes >= need);
    if (own) {
        e = hipMallocAsync(&ws, need, s);
        if (e != hipSuccess) return (int)e;
    }
    e = hipMemsetAsync(ws, 0xFF, need, s);
    if (e != hipSuccess) return (int)e;
    const int nblk = (n + 255) / 256;
    const dim3 g((unsigned)((long)nblk * bp));
    hipLaunchKernelGGL(knn_dup_hash_insert_kernel, g, dim3(256), 0, s, n, c, tsize, points, L.n_arr, (uint32_t *)ws,
                       bp, nblk, (const uint32_t *)uws, -1);
    hipLaunchKernelGGL(knn_dup_hash_lookup_kernel, g, dim3(256), 0, s, n, c, tsize, points, L.n_arr,
                       (const uint32_t *)ws, dup, uws, bp, nblk, -1);
    int r = tpu3_launch_status();
    if (own) {
        e = hipFreeAsync(ws, s);
        if (!r && e != hipSuccess) r = (int)e;
    }
    return r;
}

#ifdef KG_TRACE
extern "C" int tpu3_debug_kg_trace(long long *buf)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_kg_trace), &buf, sizeof(buf));
}
#endif
