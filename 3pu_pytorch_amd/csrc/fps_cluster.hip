// fps_cluster.hip -- the tile-form farthest-point sampling (fps_bucket.hip, fl_main_kernel) on SEVERAL compute units
// per point set (gfx950).
//
// Reference: sampling/sampling_cuda.cu:103-174 (one thread block per batch element; main.py:379-380 calls it with ONE
// element of 239 616 points for 80 000 samples).  The single-workgroup tile form runs that call as ~2000 rounds of ~40
// exact samples each on one compute unit; a round is tile prune -> bucket records -> the reached buckets' points ->
// candidate list -> ranking -> clearance, two thirds of it the instruction throughput of that one unit (tools/
// fps_tile_probe.py: 27 k of 40 k cycles in the update phases).  Here a CLUSTER of G = 2 ... 64 workgroups (a power of
// two) shares a set:
//
//   * tile t of the Morton-ordered slab belongs to member t mod G (a sample's ball covers CONSECUTIVE tiles: the
//     interleaving spreads a round's visits evenly); the member keeps the maxima / arg-max positions of its tiles'
//     buckets in ITS LDS (1/G of the table: from 16 members on, 256 tiles each = 4.19 M points on two levels -- config
//     C5's 3.83 M point resample needs no third level) and is the only writer of its buckets' running distances and
//     records;
//   * per round every member applies the round's samples to its own tiles (prune: a wave per tile, a lane per
//     sample; visits; bucket updates), lists its own candidate buckets above ITS largest runner-up bound, and
//     PUBLISHES the best <= 62 of them (a histogram cut) with its maximum and the threshold T_g its list is complete
//     above: 8-byte {epoch, value} granules, one write-through (sc1) store each, no fence (MI355X guide, Guideline 16
//     form R2: the data is the flag);
//   * every member then reads ALL G mailboxes (a wave per source, relaxed agent-scope polls until every granule
//     carries the round's epoch) and derives the round's samples itself, on all its waves: T = max_g T_g, a thread per
//     listed candidate, the 64 best above T by a histogram threshold, seats, ranks as 16 partial sums, coordinates
//     and tie keys from the immutable part of the slab, the longest clear prefix.  All members compute the same list
//     from the same published words, so ONE exchange per round is the only inter-workgroup step -- no barrier, no
//     leader, no broadcast;
//   * equal maxima at the top (duplicated points) take the exact arg-max with the reference's tie rule through a
//     second, rare exchange of (smallest tie key, slot) per member.
//
// Exactness is the tile form's: any threshold >= R* = max of all runner-up bounds is valid; T >= R*
// because every T_g >= the member's own bound; the published lists are complete above T_g <= T; every cut keeps
// "everything above a threshold" (the histograms bin the maxima monotonically); rank order is (maximum descending, tie
// key ascending), independent of list order.  Bit-identical indices and final `temp`.
//
// Residency: the members of a cluster spin on each other, so all b * G workgroups of a launch must be resident.  The
// host keeps b * G <= 64 per launch (a whole compute unit each) and, through a ring of events across streams, at most
// (compute units / 64) = 4 cluster launches in flight per device (tpu3_fps_cluster_launch): everything else on the
// device terminates, so every admitted launch gets its units (tests/test_hip_kernels.py runs 640 such workgroups on
// ten streams at once).  Every poll is bounded all the same: a member that gives up counts a fault
// (tpu3_fps_cluster_faults, read by pipeline.upsample / bench.py at their synchronisation points; `stats[4]`) and all
// members of its cluster leave.
#include "fps_bucket.h"

#include <cstdlib>
#include <mutex>

namespace {

constexpr int FC_CAP = 64;              // samples per round
constexpr int FC_LCAP = 62;             // candidates a member publishes per round: 4 header + 2 x 62 granules = 2 sweeps
                                        // (31 / 15 / 7 on 16 / 32 / 64 members: the cluster's list holds FC_LIST entries)
constexpr int FC_LIST = 512;            // candidates a member may list per round
constexpr int FC_WORK = 2048;           // work list entries (reached buckets of a round, one member)
constexpr int FC_DENSE = 16;            // a tile with this many reached buckets is updated on the spot
constexpr int FC_P2 = 3;                // phase-2 steps of a wave whose points are fetched together
constexpr int FC_GMAX = 64;             // members per cluster at most (a wave polls sources wave, wave + 16, ...)
constexpr int FC_MB = 128;              // granules per mailbox: 4 header + 62 entries x 2 words
constexpr int FC_EWMAX = 2;             // words of a candidate entry: maximum, slot
constexpr unsigned FC_SPIN_MAX = 1u << 23;      // polls before a member gives up (~seconds)

typedef unsigned long long u64;

// granule traffic: relaxed agent-scope 8-byte accesses = global_{store,load}_dwordx2 sc1 (write-through / L1 bypass)
__device__ __forceinline__ void fc_put(u64 *p, unsigned epoch, uint32_t v)
{
    __hip_atomic_store(p, ((u64)epoch << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 fc_get(const u64 *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// faulted workgroups since the last reset (tpu3_fps_cluster_faults)
__device__ u64 fc_fault_count;
// fault injection (tpu3_debug_fps_cluster_absent, tests only): non-zero = member 1 of every cluster leaves at once, as
// if it had never become resident, and the others give up after 4096 polls instead of 2^23
__device__ unsigned fc_debug_absent;

struct FcShared {
    float pick[2][FC_CAP][4];           // the round's samples in rank order: x, y, z, distance
    uint32_t pkey[2][FC_CAP];
    int mrow[FC_CAP];                   // ranking: the selected maxima
    uint32_t msel[FC_CAP];              //          their slots (point index in the Morton slab)
    uint32_t kt[FC_CAP];                //          their tie keys (ties only)
    uint32_t cand[FC_LIST * FC_EWMAX];  // this member's candidates: (maximum, slot)
    uint32_t dl[FC_LIST * FC_EWMAX];    // what ALL members published, in arrival order (the pollers deposit while wave 0
                                        // may still be reading the local list: two buffers)
    uint32_t lsel[FC_CAP];              // the ones it publishes (positions in `cand`)
    int mcnt[FC_GMAX], mthr[FC_GMAX], mbest[FC_GMAX];
    float tbox[256 * 8];                // boxes of this member's tiles: lo.xyz, hi.xyz (+ 2 unused words)
    u64 tmc[FC_GMAX];                   // tie exchange: (key << 32 | slot) per member
    u64 tiekey;
    int npick[2];
    int ncand, nwork, jclear, ndense, nseat;
    alignas(16) int hist[256];          // merge: histogram of the candidates above T (monotone bins of the maxima)
    int rankeq[FC_CAP];                 //        per seat: larger maxima (low half), equal ones (high half)
    int lcount, lthr;
    int gbest;
    int fail;
    u64 stat[8];
};
static_assert(offsetof(FcShared, mrow) % 16 == 0 && offsetof(FcShared, cand) % 16 == 0 && offsetof(FcShared, dl) % 16 == 0 &&
                  offsetof(FcShared, hist) % 16 == 0 && offsetof(FcShared, tbox) % 16 == 0,
              "vector reads of the lists");

constexpr size_t fc_mbox_words(int g) { return (size_t)2 * g * FC_MB + (size_t)2 * g * 2 + 8; }

// (Measured and dropped: bucket records and the buckets' best points (x, y, z, tie key) in LDS with the candidates
// published WITH their coordinates -- no memory trip in a round but the reached buckets' points.  Bit-identical, and
// slower, 20.5 vs 17.2 ms for 239 616 -> 80 000 on eight members: six granules per candidate instead of two cost
// 2.2 k cycles more to publish and 1.3 k more to poll, against 0.4 k saved in the ranking and 0.2 k in the visits.)
constexpr size_t fc_lds_bytes(int ntile, int g)
{
    return ((((size_t)((ntile + g - 1) / g) * 64 * 5) + 15) & ~(size_t)15) + 512 * 4 + (size_t)FC_WORK * 12 +
           sizeof(FcShared) + 64;
}

template <bool PROF>
__global__ __launch_bounds__(1024) void fc_main_kernel(FbArgs a0, int lg, u64 *mbox0, u64 *stats)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EW = 2;                                   // words per candidate entry
    constexpr int NSW = (4 + FC_LCAP * EW + 63) / 64;       // 64-granule sweeps of a mailbox
    const int G = 1 << lg;
    const int lcap = G <= 8 ? FC_LCAP : FC_LIST / G - 1;    // (31 on 16 members, 15 on 32, 7 on 64)
    const int cl = blockIdx.x >> lg, g = blockIdx.x & (G - 1);
    const FbArgs a = fb_elem(a0, cl);
    if (a.n <= 0 || a.m <= 0)
        return;                                             // (the whole cluster leaves)
    const int ntile = (a.n + FL_TP - 1) / FL_TP;            // live tiles of this element
    const int nltmax = (a0.ntile + G - 1) >> lg;            // LDS stride: local tiles of a member at most
    const int nlt = (ntile - g + G - 1) >> lg;              // live local tiles of THIS member (0 for tiny sets)
    int *bmax = (int *)smem;                                // [nltmax * 64] maxima of this member's buckets
    uint8_t *barg = (uint8_t *)(bmax + nltmax * 64);        // [nltmax * 64] position of a bucket's best point
    int *tmx = (int *)(smem + (((size_t)nltmax * 64 * 5 + 15) & ~(size_t)15));      // [256] tile maxima
    int *trn = tmx + 256;                                   // [256] tile runner-up bounds
    uint32_t *work = (uint32_t *)(trn + 256);               // [FC_WORK][3]
    FcShared &sh = *(FcShared *)(work + FC_WORK * 3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lb = a.lb;
    float4 *__restrict__ TP = a.sp;
    const uint32_t *__restrict__ TK = a.skey;
    u64 *mb = mbox0 + (size_t)cl * fc_mbox_words(G);        // this cluster's mailboxes: [parity][member][FC_MB]
    u64 *tb = mb + (size_t)2 * G * FC_MB;                   // tie exchange: [parity][member][2]
    u64 *faultw = tb + (size_t)2 * G * 2;
    const unsigned absent = fc_debug_absent;
    const unsigned spin_max = absent ? 4096u : FC_SPIN_MAX;
    if (absent && g == 1)
        return;
    // a member gives up when its own polls run out, when another wave of its workgroup did, or when ANOTHER member
    // of the cluster has published its fault (looked at every 1024 polls): the cluster leaves together
    auto gave_up = [&](unsigned spins) __attribute__((always_inline)) {
        return spins > spin_max || __hip_atomic_load(&sh.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ||
               ((spins & 1023u) == 0 && fc_get(faultw) != 0);
    };

    // local bucket index <-> global bucket id (tile t = lt * G + g)
    auto gbucket = [&](int lbk) __attribute__((always_inline)) { return ((((lbk >> 6) << lg) | g) << 6) | (lbk & 63); };
    auto lbucket = [&](int gb) __attribute__((always_inline)) { return (((gb >> 6) >> lg) << 6) | (gb & 63); };

    for (int i = tid; i < nlt * 64; i += 1024) {
        const int gb = gbucket(i);
        bmax[i] = a.bm0[gb];
        barg[i] = a.ba0[gb];
    }
    for (int i = tid; i < 256; i += 1024) {
        const int t = (i << lg) | g;
        tmx[i] = i < nlt ? __float_as_int(a.tt[t * 8 + 6]) : (int)0x80000000;
        trn[i] = i < nlt ? __float_as_int(a.tt[t * 8 + 7]) : (int)0x80000000;
    }
    // A member's tiles are OWNED (visited, listed) by lane l < 16 of wave w: local tile l * 16 + w.
    const int ltq = lane * 16 + wave;
    const bool tvalid = lane < 16 && ltq < nlt;
    // The prune step: wave w tests ITS tiles (w, w + 16, ...) against the round's samples, a lane per SAMPLE, the
    // tile's box wave-uniform from LDS -- the ballot IS the tile's sample set, and the owner visits the tile at once:
    // no hand-off, no barrier between prune and visits.  (A quad per tile as in the single-workgroup form left 8 of a
    // wave's 64 lanes busy here, 7 k cycles of a round; tiles across the lanes with the samples dealt over the waves
    // needed LDS atomics and a barrier to assemble the sets.)
    float *tbox = sh.tbox;
    for (int i = tid; i < nlt * 8; i += 1024)
        tbox[i] = a.tt[(size_t)((((i >> 3) << lg) | g)) * 8 + (i & 7)];
    if (tid == 0) {
        sh.ncand = 0;
        sh.nwork = 0;
        sh.ndense = 0;
        sh.nseat = 0;
        sh.fail = 0;
        sh.npick[0] = sh.npick[1] = 0;
        if (g == 0)
            a.idx[0] = 0;
        sh.pick[1][0][0] = a.xyz[0]; sh.pick[1][0][1] = a.xyz[1]; sh.pick[1][0][2] = a.xyz[2];
        sh.pick[1][0][3] = 0.f;
    }
    if (tid < 8)
        sh.stat[tid] = 0;
    if (tid < 256)
        sh.hist[tid] = 0;
    if (tid < FC_CAP)
        sh.rankeq[tid] = 0;
    __syncthreads();
    if (a.m <= 1)
        return;                                     // the reference's loop body never runs: temp untouched

    // pair (i < l) number `lane` of the l-major enumeration (clearance test, first pass)
    int pair_l = (int)((1.f + sqrtf(1.f + 8.f * (float)lane)) * 0.5f);
    pair_l -= pair_l * (pair_l - 1) / 2 > lane ? 1 : 0;
    pair_l += (pair_l + 1) * pair_l / 2 <= lane ? 1 : 0;
    const int pair_i = lane - pair_l * (pair_l - 1) / 2;

    // PROF: wave 0's cycles per phase -- apply phase 1 (+ barrier), phase 2 (+ barrier), candidate collection, local
    // selection + publish, poll (+ barrier), merge / rank (+ barrier), clearance (+ barrier); listed / merged counts
    u64 pc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt = 0, pm = 0;
    auto mmark = [&](int k) __attribute__((always_inline)) {
        if (PROF) {
            const u64 now = __builtin_amdgcn_s_memtime();
            pc[k] += now - pm;
            pm = now;
        }
    };
    auto mark = [&](int k) __attribute__((always_inline)) {
        if (PROF) {
            const u64 now = __builtin_amdgcn_s_memtime();
            pc[k] += now - pt;
            pt = now;
        }
    };

    // ---- one bucket (the lane's): fold the samples of `sm0` into its 16 distances, re-derive its record ----------
    auto update_bucket = [&](bool act, int gb, int lbk, unsigned long long sm0, int cur, int &best, int &run)
        __attribute__((always_inline)) {
        float4 *__restrict__ base = TP + (size_t)gb * FL_R;
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
                    const int tbits = __float_as_int(nt[j]);
                    asm("v_med3_i32 %0, %1, %2, %0" : "+v"(run) : "v"(best), "v"(tbits));
                    arg = tbits > best ? h + j : arg;          // (first of equal maxima; ties are settled below)
                    best = max(best, tbits);
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
                    const uint32_t kj = TK[(size_t)gb * FL_R + j];
                    const bool take = __float_as_int(base[j].w) == best && kj < bk;
                    bk = take ? kj : bk;
                    arg = take ? j : arg;
                }
            }
        }
        if (act) {
            bmax[lbk] = best;
            barg[lbk] = (uint8_t)arg;
            ((uint32_t *)(a.rec + gb))[3] = (uint32_t)run;
        }
    };

    // ---- fold the first nj samples of pick[cur] into every bucket of THIS member they reach -------------------
    auto apply = [&](int nj, int cur) __attribute__((always_inline)) {
        // one tile (local index lt): which of its 64 buckets do the samples of `smt` reach?
        auto visit_tile = [&](int lt, unsigned long long smt) __attribute__((always_inline)) {
            const int lbk = lt * 64 + lane;
            const int gb = gbucket(lbk);
            const uint4 rc = a.rec[gb];
            const int bm = bmax[lbk];
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
            bool dense = nreach >= FC_DENSE;
            if (!dense) {
                if (lane == 0)
                    base = atomicAdd(&sh.nwork, nreach);
                base = __builtin_amdgcn_readfirstlane(base);
                dense = base + nreach > FC_WORK;        // (list full: on the spot; the reserved slots are voided)
            }
            if (dense) {
                if (nreach < FC_DENSE && reached) {
                    const int pos = base + __builtin_popcountll(rm & ((1ull << lane) - 1ull));
                    if (pos < FC_WORK)
                        work[3 * pos] = 0xFFFFFFFFu;
                }
                int best, run;
                update_bucket(reached, gb, lbk, smt, cur, best, run);
                int tm = reached ? best : bm, tr = reached ? run : (int)rc.w;
                tpu3_wave_max_i32_fast_x2(tm, tr);
                if (lane == 0) {
                    tmx[lt] = tm;
                    trn[lt] = tr;
                }
                return;
            }
            // the tile's maxima over the buckets NOT reached; phase 2 adds the reached ones' (atomic max)
            int um = reached ? (int)0x80000000 : bm, ur = reached ? (int)0x80000000 : (int)rc.w;
            tpu3_wave_max_i32_fast_x2(um, ur);
            if (lane == 0) {
                tmx[lt] = um;
                trn[lt] = ur;
            }
            if (reached) {
                uint32_t *e = work + 3 * (base + __builtin_popcountll(rm & ((1ull << lane) - 1ull)));
                e[0] = (uint32_t)gb; e[1] = (uint32_t)mine; e[2] = (uint32_t)(mine >> 32);
            }
        };
        {
            const bool has = lane < nj;
            const float4 ps = *(const float4 *)sh.pick[cur][has ? lane : 0];
            for (int lt = wave; lt < nlt; lt += 16) {
                const float4 b0 = *(const float4 *)(tbox + lt * 8), b1 = *(const float4 *)(tbox + lt * 8 + 4);
                const float tmf = __int_as_float(tmx[lt]);
                const unsigned long long m =
                    __ballot(has && fb_dbox(ps.x, ps.y, ps.z, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y) < tmf);
                if (m)
                    visit_tile(lt, m);
            }
        }
        __syncthreads();
        mark(0);
        // phase 2: a DPP row of 16 lanes per listed bucket, a lane per point; four buckets per wave step, the steps
        // dealt over the 16 waves
        const int E = min(sh.nwork, FC_WORK);
        const int row = lane >> 4, col = lane & 15;
        for (int g0 = wave * 4; g0 < E; g0 += 64 * FC_P2) {
            float4 ptv[FC_P2];
            int bv[FC_P2];
#pragma unroll
            for (int k = 0; k < FC_P2; ++k) {
                const int ee = g0 + 64 * k + row;
                bv[k] = (int)work[3 * (ee < E ? ee : 0)];
                ptv[k] = TP[(size_t)(bv[k] >= 0 ? bv[k] : 0) * FL_R + col];
            }
#pragma unroll
            for (int k = 0; k < FC_P2; ++k) {
                const int e0 = g0 + 64 * k;
                if (e0 >= E)
                    break;
                const bool inl = e0 + row < E;
                const uint32_t *e = work + 3 * (inl ? e0 + row : 0);
                const bool act = inl && bv[k] >= 0;                 // (a voided slot: its tile was updated on the spot)
                const int gb = act ? bv[k] : 0;
                unsigned long long mine = act ? (((unsigned long long)e[2] << 32) | e[1]) : 0ull;
                float4 *__restrict__ pp = TP + (size_t)gb * FL_R + col;
                const float4 pt = ptv[k];
                float nt = pt.w;
                while (__ballot(mine != 0)) {           // every row walks ITS bucket's samples
                    const bool go = mine != 0;
                    const float4 p = *(const float4 *)sh.pick[cur][go ? __builtin_ctzll(mine) : 0];
                    mine &= mine - 1;
                    const float d = tpu3_min1(tpu3_sqdist3(pt.x - p.x, pt.y - p.y, pt.z - p.z), nt);
                    nt = go ? d : nt;
                }
                const int tbits = act ? __float_as_int(nt) : (int)0x80000000;
                const int best = tpu3_row_max_i32_fast(tbits);
                // the winner: the only lane at the maximum, or -- duplicated points -- the one with the smallest tie key
                unsigned long long tie = __ballot(act && tbits == best);
                const unsigned long long rowm = 0xFFFFull << (row * 16);
                bool single = true;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    single &= __builtin_popcountll(tie & (0xFFFFull << (rr * 16))) <= 1;
                if (!single) {
                    const uint32_t kj = act && tbits == best ? TK[(size_t)gb * FL_R + col] : 0xFFFFFFFFu;
                    const uint32_t km = tpu3_row_min_u32(kj);
                    tie = __ballot(act && tbits == best && kj == km);
                }
                const bool winner = act && ((tie >> lane) & 1ull) != 0 && (tie & rowm & ((1ull << lane) - 1ull)) == 0;
                const int run = tpu3_row_max_i32_fast(winner || !act ? (int)0x80000000 : tbits);
                if (act && nt != pt.w)
                    ((float *)pp)[3] = nt;
                if (winner) {
                    const int lbk = lbucket(gb);
                    bmax[lbk] = best;
                    barg[lbk] = (uint8_t)col;
                    ((uint32_t *)(a.rec + gb))[3] = (uint32_t)run;
                    atomicMax(&tmx[lbk >> 6], best);
                    atomicMax(&trn[lbk >> 6], run);
                }
            }
        }
        __syncthreads();
        mark(1);
        if (tid == 0)
            sh.nwork = 0;
    };

    // the buckets of this wave's tiles in `ul` (a ballot over the owner lanes): body(local bucket of this lane,
    // its maximum, its best point's position) per tile
    auto for_tiles = [&](unsigned long long ul, auto body) __attribute__((always_inline)) {
        for (; ul; ul &= ul - 1) {
            const int lbk = ((int)__builtin_ctzll(ul) * 16 + wave) * 64 + lane;
            body(lbk, bmax[lbk], (int)barg[lbk]);
        }
    };

    int J = 1, r = 1;
    unsigned tepoch = 0;                            // tie exchanges so far
    unsigned long long n_tie = 0, n_sweep = 0;
    for (int round = 0;; ++round) {
        const int par = round & 1;
        const unsigned epoch = (unsigned)round + 1u;
        if (PROF) pt = __builtin_amdgcn_s_memtime();
        apply(J, par ^ 1);
        // ---- this member's candidates ------------------------------------------------------------------------
        int lbest, lrstar;
        {
            int vm = (int)0x80000000, vr = (int)0x80000000;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                vm = max(vm, tmx[i * 64 + lane]);
                vr = max(vr, trn[i * 64 + lane]);
            }
            tpu3_wave_max_i32_fast_x2(vm, vr);
            lbest = vm; lrstar = vr;
        }
        const int tmax = tvalid ? tmx[ltq] : (int)0x80000000;
        // every bucket above `thr` (any threshold >= this member's largest runner-up bound is valid); when more
        // than the list holds qualify the threshold is raised; when even the top value alone is shared by more
        // buckets than the list holds (duplicated points) nothing is listed and T_g = the top value itself
        int thr = lrstar, total = 0;
        bool none = lbest <= lrstar;
        while (!none) {
            const unsigned long long tl = __ballot(tvalid && tmax > thr);
            for_tiles(tl, [&](int lbk, int bm, int ba) __attribute__((always_inline)) {
                const bool c = bm > thr;
                const unsigned long long cm = __ballot(c);
                if (!cm)
                    return;
                int base = 0;
                if (lane == 0)
                    base = atomicAdd(&sh.ncand, (int)__builtin_popcountll(cm));
                base = __builtin_amdgcn_readfirstlane(base);
                const int pos = base + __builtin_popcountll(cm & ((1ull << lane) - 1ull));
                if (c && pos < FC_LIST) {
                    uint32_t *e = sh.cand + pos * EW;
                    e[0] = (uint32_t)bm;
                    e[1] = ((uint32_t)gbucket(lbk) << 4) | (uint32_t)ba;
                }
            });
            __syncthreads();
            total = sh.ncand;
            if (total <= FC_LIST)
                break;
            __syncthreads();
            if (tid == 0)
                sh.ncand = 0;
            if ((long)lbest - (long)thr <= 1) {
                none = true;
                total = 0;
            } else {
                thr = (int)(((long)thr + (long)lbest) >> 1);
            }
            __syncthreads();
        }
        if (none) {
            thr = max(lrstar, lbest);
            total = 0;
        }
        mark(2);
        if (PROF) pc[7] += total;
        // wave 0: the best `lcap` of the list (usually all of it: then the list is published as it stands), with the header
        if (wave == 0) {
            int thr2 = thr, nsel = total;
            const bool cut = total > lcap;
            if (cut) {
                int em[FC_LIST / 64];
#pragma unroll
                for (int u = 0; u < FC_LIST / 64; ++u) {
                    const int i = u * 64 + lane;
                    em[u] = (int)0x80000000;
                    if (u * 64 < total)
                        em[u] = i < total ? (int)sh.cand[EW * i] : (int)0x80000000;
                }
                // the cut by a 64-bin histogram of the listed maxima (a monotone binning, as in the merge below): the bins
                // from the top that hold <= lcap candidates are kept, and T_g = the largest maximum NOT kept -- the list
                // is complete above it.  (A bisection over the values took ~10 steps of 2-8 ballots: 1.5 k cycles of
                // every round of config C5, where ~70-100 buckets per member beat its bound.)
                const uint32_t Rl = (uint32_t)lbest - (uint32_t)thr;
                const int sftl = max(0, 9 - (int)__builtin_clz(Rl | 1u));
                const float scl = 64.f / ((float)(Rl >> sftl) + 1.f);
                int bn[FC_LIST / 64];
#pragma unroll
                for (int u = 0; u < FC_LIST / 64; ++u) {
                    bn[u] = -1;
                    if (u * 64 < total && em[u] > thr) {
                        bn[u] = min(63, (int)((float)(((uint32_t)em[u] - (uint32_t)thr) >> sftl) * scl));
                        atomicAdd(&sh.hist[bn[u]], 1);
                    }
                }
                int bcut;
                {
                    const int own = sh.hist[63 - lane];         // lane l: bin 63 - l; prefix over lanes = bins from the top
                    sh.hist[63 - lane] = 0;                     // (the merge expects the histogram empty)
                    int v = own;
                    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
                    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
                    const unsigned long long over = __ballot(v > lcap);      // (total > lcap: some lane overflows)
                    const int Lx = over ? (int)__builtin_ctzll(over) : 64;
                    bcut = 64 - Lx;                                          // keep bins >= bcut
                    nsel = Lx > 0 ? __builtin_amdgcn_readlane(v, Lx > 0 ? Lx - 1 : 0) : 0;
                }
                int left_out = (int)0x80000000;                 // the largest listed maximum that is not kept
#pragma unroll
                for (int u = 0; u < FC_LIST / 64; ++u)
                    if (u * 64 < total)
                        left_out = max(left_out, bn[u] < bcut ? em[u] : (int)0x80000000);
                thr2 = max(thr, tpu3_wave_max_i32_fast(left_out));
                int base = 0;
#pragma unroll
                for (int u = 0; u < FC_LIST / 64; ++u) {
                    if (u * 64 >= total)
                        break;
                    const bool sel = em[u] > thr2;
                    const unsigned long long smk = __ballot(sel);
                    if (sel)
                        sh.lsel[base + __builtin_popcountll(smk & ((1ull << lane) - 1ull))] = (uint32_t)(u * 64 + lane);
                    base += __builtin_popcountll(smk);
                }
            }
            // (the LDS list above is read back by other lanes of this wave only: program order within a wave)
            u64 *box = mb + ((size_t)par * G + g) * FC_MB;
            const int need = 4 + EW * nsel;
#pragma unroll
            for (int s = 0; s < NSW; ++s) {
                const int i = s * 64 + lane;
                if (s * 64 >= need)
                    break;
                uint32_t v = 0;
                if (i == 0) v = (uint32_t)lbest;
                else if (i == 1) v = (uint32_t)thr2;
                else if (i == 2) v = (uint32_t)nsel;
                else if (i >= 4 && i < need)
                    v = cut ? sh.cand[EW * sh.lsel[(i - 4) / EW] + (i - 4) % EW] : sh.cand[i - 4];
                if (i < need)
                    fc_put(box + i, epoch, v);
            }
            if (lane == 0)
                sh.ncand = 0;
        }
        mark(3);
        // ---- what every member published: one wave per source polls until each granule it needs carries the epoch
        for (int sw = wave; sw < G; sw += 16) {
            const u64 *src = mb + ((size_t)par * G + sw) * FC_MB;
            unsigned spins = 0;
            u64 x[NSW];
            int cnt = 0;
            for (;;) {
#pragma unroll
                for (int q = 0; q < NSW; ++q)
                    x[q] = fc_get(src + q * 64 + lane);
                const bool hdr = (__ballot((unsigned)(x[0] >> 32) == epoch) & 7ull) == 7ull;
                cnt = hdr ? __builtin_amdgcn_readlane((int)(uint32_t)x[0], 2) : 0;
                const int need = 4 + EW * cnt;
                bool miss = false;
#pragma unroll
                for (int q = 0; q < NSW; ++q)
                    miss |= q * 64 + lane < need && (unsigned)(x[q] >> 32) != epoch;
                if (hdr && !__ballot(miss))
                    break;
                if (gave_up(++spins)) {
                    if (lane == 0)
                        sh.fail = 1;
                    cnt = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (wave == 0)
                n_sweep += spins + 1;
            if (lane == 0) {
                sh.mbest[sw] = (int)(uint32_t)x[0];
                sh.mcnt[sw] = cnt;
            }
            if (lane == 1)
                sh.mthr[sw] = (int)(uint32_t)x[0];
            // the entries go straight onto the cluster's list (any order: the ranking does not depend on it), unfiltered
            // -- T0 is known only when all headers are in
            int base = 0;
            if (lane == 0 && cnt)
                base = atomicAdd(&sh.ndense, cnt);
            base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
            for (int q = 0; q < NSW; ++q) {
                const int i = q * 64 + lane;
                if (i >= 4 && i < 4 + EW * cnt)
                    sh.dl[EW * base + i - 4] = (uint32_t)x[q];
            }
        }
        __syncthreads();
        mark(4);
        if (sh.fail)
            break;
        // ---- the round's samples: every member derives the same list, ALL its waves at work --------------------------
        // (In a first form wave 0 did this alone -- filter, bisection for the 64 best, compaction, ranking: 6.3 k of a
        // round's ~19 k cycles while fifteen waves waited.)  A thread per listed candidate: T = max_g T_g and the
        // cluster's maximum from the headers; a 256-bin histogram of the candidates above T (LDS atomics) whose suffix
        // sums every wave scans itself gives the threshold bin that keeps <= 64 (a monotone binning of the maxima: the
        // kept set is "everything above a threshold", as exactness needs; up to one bin's worth of candidates fewer
        // than 64 may be kept); the kept ones take seats through an atomic counter; their ranks are counted as 16
        // partial sums (wave w against seats 4 w .. 4 w + 3) while wave 0's fetch of their coordinates is in flight.
        const int left = a.m - r;
        uint32_t okey = 0;
        int gbest, T0;
        {
            int gb_ = lane < G ? sh.mbest[lane] : (int)0x80000000;
            int t0_ = lane < G ? sh.mthr[lane] : (int)0x80000000;
            tpu3_wave_max_i32_fast_x2(gb_, t0_);
            gbest = gb_; T0 = t0_;
        }
        if (PROF) pm = __builtin_amdgcn_s_memtime();
        const int tot = sh.ndense;                              // (<= FC_LIST <= 1024 threads)
        const int my_bm = tid < tot ? (int)sh.dl[EW * tid] : (int)0x80000000;
        const bool cand_ok = my_bm > T0;
        int my_bin = -1;
        if (cand_ok) {
            // bin = floor(256 d / (R + 1)) with d = maximum - T0 <= R = cluster maximum - T0, as unsigned differences of
            // the distance bits (they may exceed 2^31: T0 can be negative), both shifted down to 23 bits so that the float
            // quotient is exact enough to be MONOTONE in d -- all the selection needs
            const uint32_t d = (uint32_t)my_bm - (uint32_t)T0, R = (uint32_t)gbest - (uint32_t)T0;
            const int sft = max(0, 9 - (int)__builtin_clz(R | 1u));
            my_bin = min(255, (int)((float)(d >> sft) * 256.f / ((float)(R >> sft) + 1.f)));
            atomicAdd(&sh.hist[my_bin], 1);
        }
        __syncthreads();
        mmark(8);
        int nsel, bstar;
        {
            // lane l holds bins 4 (63 - l) .. + 3, highest first: an inclusive PREFIX sum over the lanes is the number of
            // candidates in bins >= 4 (63 - l)
            const int4 h = *(const int4 *)(sh.hist + 4 * (63 - lane));
            const int own = h.x + h.y + h.z + h.w;
            int v = own;
            v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);         // row_shr:1
            v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);         // row_shr:2
            v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);         // row_shr:4
            v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);         // row_shr:8
            v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);         // row_bcast:15 -> rows 1, 3
            v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);         // row_bcast:31 -> rows 2, 3
            const unsigned long long over = __ballot(v > FC_CAP);
            if (!over) {
                bstar = 0;                                      // everything above T0 fits
                nsel = __builtin_amdgcn_readlane(v, 63);
            } else {
                const int Lx = (int)__builtin_ctzll(over);      // the first lane whose bins overflow the round
                int cum = __builtin_amdgcn_readlane(v - own, Lx);                   // candidates in the bins above its four
                const int h3 = __builtin_amdgcn_readlane(h.w, Lx), h2 = __builtin_amdgcn_readlane(h.z, Lx);
                const int h1 = __builtin_amdgcn_readlane(h.y, Lx);
                const int top = 4 * (63 - Lx);
                bstar = top + 4;
                if (cum + h3 <= FC_CAP) { cum += h3; bstar = top + 3;
                    if (cum + h2 <= FC_CAP) { cum += h2; bstar = top + 2;
                        if (cum + h1 <= FC_CAP) { cum += h1; bstar = top + 1; } } }
                nsel = cum;
            }
        }
        const bool kept = cand_ok && my_bin >= bstar;
        if (kept) {
            const int pos = atomicAdd(&sh.nseat, 1);
            sh.mrow[pos] = my_bm;
            sh.msel[pos] = (uint32_t)tid;                       // (position on the cluster's list)
        }
        if (tid >= nsel && tid < FC_CAP)
            sh.mrow[tid] = (int)0x80000000;
        __syncthreads();
        mmark(9);
        if (tid < 256)
            sh.hist[tid] = 0;                                   // (every wave has scanned it)
        // wave 0: ONE round trip for the coordinates and keys of the kept candidates (the IMMUTABLE words of the slab: any
        // member may read them; the running distance of a foreign bucket is taken from its published maximum), in flight
        // while the ranks are counted
        const bool live = wave == 0 && lane < nsel;
        float4 sp4 = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t cK = 0xFFFFFFFFu;
        if (live) {
            const uint32_t cB = sh.dl[EW * sh.msel[lane] + 1];
            sp4 = TP[cB];
            cK = TK[cB];
        }
        {
            // seat `lane` against seats 4 wave .. 4 wave + 3: larger maxima (low half) and equal ones (high half)
            const int mine = sh.mrow[lane];
            const int4 o = *(const int4 *)(sh.mrow + 4 * wave);
            const int gt = (o.x > mine) + (o.y > mine) + (o.z > mine) + (o.w > mine);
            const int eq = (o.x == mine) + (o.y == mine) + (o.z == mine) + (o.w == mine);
            if (lane < nsel && (gt | eq))
                atomicAdd(&sh.rankeq[lane], gt | (eq << 16));
        }
        __syncthreads();
        mmark(10);
        if (wave == 0) {
            const int cM = live ? sh.mrow[lane] : (int)0x80000000;
            const int re = live ? sh.rankeq[lane] : 0;
            int rank = re & 0xFFFF;
            const bool tie = live && (re >> 16) > 1;            // (its own seat counts once)
            sh.rankeq[lane] = 0;
            if (__ballot(tie)) {                    // equal maxima among candidates: order by the tie key
                sh.kt[lane] = cK;
                rank = 0;
                for (int c0 = 0; c0 < nsel; c0 += 8) {
                    const int4 m0 = *(const int4 *)(sh.mrow + c0), m1 = *(const int4 *)(sh.mrow + c0 + 4);
                    const uint4 k0 = *(const uint4 *)(sh.kt + c0), k1 = *(const uint4 *)(sh.kt + c0 + 4);
                    const int m8[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
                    const uint32_t k8[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
#pragma unroll
                    for (int v = 0; v < 8; ++v)
                        rank += (m8[v] > cM || (m8[v] == cM && k8[v] < cK)) ? 1 : 0;
                }
            }
            if (live) {
                *(float4 *)sh.pick[par][rank] = make_float4(sp4.x, sp4.y, sp4.z, __int_as_float(cM));
                sh.pkey[par][rank] = cK;
            }
            okey = sh.pkey[par][lane < nsel ? lane : 0];
            if (lane == 0) {
                const int jm = nsel < left ? nsel : left;
                sh.npick[par] = jm;
                sh.jclear = jm;
                sh.gbest = gbest;
                sh.ndense = 0;
                sh.nseat = 0;
            }
            mmark(11);
        }
        __syncthreads();
        mark(5);
        int jmax = sh.npick[par];
        if (jmax == 0) {
            // one sample by the exact arg-max with the tie rule: the smallest tie key among ALL buckets at the top
            // value -- every member finds its own, a second exchange finds the cluster's
            ++n_tie;
            const int gbest = sh.gbest;
            const unsigned te = ++tepoch;
            if (tid == 0)
                sh.tiekey = ~0ull;
            __syncthreads();
            const unsigned long long t2 = __ballot(tvalid && tmax == gbest);
            for_tiles(t2, [&](int lbk, int bm, int ba) __attribute__((always_inline)) {
                if (bm == gbest) {
                    const uint32_t w = ((uint32_t)gbucket(lbk) << 4) | (uint32_t)ba;
                    atomicMin(&sh.tiekey, ((u64)TK[w] << 32) | w);
                }
            });
            __syncthreads();
            if (wave == 0 && lane < 2) {
                const u64 k = sh.tiekey;
                fc_put(tb + ((size_t)(te & 1) * G + g) * 2 + lane, te, lane == 0 ? (uint32_t)(k >> 32) : (uint32_t)k);
            }
            for (int sw = wave; sw < G; sw += 16) {
                const u64 *src = tb + ((size_t)(te & 1) * G + sw) * 2;
                unsigned spins = 0;
                u64 x = 0;
                for (;;) {
                    x = fc_get(src + (lane & 1));
                    if ((__ballot((unsigned)(x >> 32) == te) & 3ull) == 3ull)
                        break;
                    if (gave_up(++spins)) {
                        if (lane == 0)
                            sh.fail = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                const uint32_t key = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, 0);
                const uint32_t slotw = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, 1);
                if (lane == 0)
                    sh.tmc[sw] = ((u64)key << 32) | slotw;
            }
            __syncthreads();
            if (sh.fail)
                break;
            if (wave == 0) {
                u64 best = ~0ull;
                for (int w = 0; w < G; ++w)
                    best = min(best, sh.tmc[w]);
                const uint32_t w = (uint32_t)best;
                if (lane == 0) {
                    const float4 p = TP[w];
                    *(float4 *)sh.pick[par][0] = make_float4(p.x, p.y, p.z, __int_as_float(gbest));
                    sh.pkey[par][0] = (uint32_t)(best >> 32);
                    sh.npick[par] = 1;
                    sh.jclear = 1;
                }
                okey = (uint32_t)(best >> 32);
            }
            __syncthreads();
            jmax = 1;
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
        mark(6);
        J = sh.jclear;
        if (wave == 0 && g == 0 && lane < J)
            a.idx[r + lane] = tpu3_fps_tiekey_to_index(okey, lb);
        r += J;
        if (r >= a.m) {
            if (J > 1)
                apply(J - 1, par);                  // every sample but the last one updates `temp`
            if (stats && tid == 0 && g == 0 && cl == 0) {
                stats[0] = (u64)round + 1; stats[1] = (u64)r - 1; stats[2] = n_tie; stats[3] = n_sweep;
                if (PROF)
                    for (int k = 0; k < 16; ++k)
                        stats[8 + k] = pc[k];
            }
            return;
        }
    }
    // a poll gave up (a member of this cluster never became resident, or died): tell the host and the partners;
    // member 0 fills the samples not taken with index 0, so that whatever gathers through idx stays in range
    if (g == 0)
        for (int i = r + tid; i < a.m; i += 1024)
            a.idx[i] = 0;
    if (tid == 0) {
        __hip_atomic_store(faultw, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(&fc_fault_count, 1ull);
        if (stats)
            atomicAdd(stats + 4, 1ull);
    }
}

} // namespace

size_t tpu3_fps_cluster_mailbox_bytes(int g)
{
    return fc_mbox_words(g) * sizeof(u64);
}

size_t tpu3_fps_cluster_lds_bytes(int ntile, int g)
{
    return fc_lds_bytes(ntile, g);
}

int tpu3_fps_cluster_launch(hipStream_t s, int b, int g, const void *fb_args, void *mbox, unsigned long long *stats)
{
    const FbArgs &a0 = *(const FbArgs *)fb_args;
    int lg = 0;
    while ((1 << lg) < g)
        ++lg;
    if ((1 << lg) != g || g < 2 || g > FC_GMAX || (a0.ntile + g - 1) / g > 256)
        return TPU3_EINVAL;
    const size_t lds = fc_lds_bytes(a0.ntile, g);
    if (lds > 160 * 1024)
        return TPU3_ELIMIT;
    hipError_t e = hipMemsetAsync(mbox, 0, (size_t)b * tpu3_fps_cluster_mailbox_bytes(g), s);
    if (e != hipSuccess)
        return (int)e;
    // (the per-phase cycle counters only when somebody asked for the statistics)
    void (*kern)(FbArgs, int, u64 *, u64 *) = stats ? fc_main_kernel<true> : fc_main_kernel<false>;
    e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess)
        return (int)e;
    // The members of a set spin on each other, so every workgroup of a launch must become resident.  One launch alone
    // always does: b g <= 64 workgroups of a whole compute unit each, and whatever else holds compute units drains.
    // Launches on DIFFERENT streams could each hold part of the chip and wait for the rest (until the bounded spins
    // give up and the fault counter says so): a ring of events keeps at most (compute units / 64) of them in flight
    // per device -- launch i on any stream waits for launch i - 4.  (Not under stream capture: a captured launch is
    // ordered by its graph.)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    if (cap != hipStreamCaptureStatusNone) {
        hipLaunchKernelGGL(kern, dim3((unsigned)b * g), dim3(1024), lds, s, a0, lg, (u64 *)mbox, (u64 *)stats);
        return tpu3_launch_status();
    }
    constexpr int MAXDEV = 16, RING = 8;
    static std::mutex mu;
    static hipEvent_t ring[MAXDEV][RING];
    static unsigned long long issued[MAXDEV];
    static int inflight[MAXDEV];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= MAXDEV)
        dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    if (!inflight[dev]) {
        int ncu = 256;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        inflight[dev] = ncu / 64 < 1 ? 1 : (ncu / 64 > RING ? RING : ncu / 64);
    }
    const int slot = (int)(issued[dev] % (unsigned)inflight[dev]);
    if (ring[dev][slot]) {
        e = hipStreamWaitEvent(s, ring[dev][slot], 0);
        if (e != hipSuccess)
            return (int)e;
    } else {
        e = hipEventCreateWithFlags(&ring[dev][slot], hipEventDisableTiming);
        if (e != hipSuccess)
            return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)b * g), dim3(1024), lds, s, a0, lg, (u64 *)mbox, (u64 *)stats);
    const int rc = tpu3_launch_status();
    if (rc)
        return rc;
    e = hipEventRecord(ring[dev][slot], s);
    ++issued[dev];
    return e == hipSuccess ? TPU3_OK : (int)e;
}

// Test hook (not part of the path): on != 0 makes member 1 of every cluster of the following launches leave at once
// (a workgroup that never became resident) and shortens the partners' patience to 4096 polls.  Returns 0 on success.
extern "C" int tpu3_debug_fps_cluster_absent(int on)
{
    const unsigned v = on ? 1u : 0u;
    return hipMemcpyToSymbol(HIP_SYMBOL(fc_debug_absent), &v, sizeof(v)) == hipSuccess ? TPU3_OK : TPU3_EINVAL;
}

extern "C" long tpu3_fps_cluster_faults(int reset)
{
    u64 v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(fc_fault_count), sizeof(v)) != hipSuccess)
        return -1;
    if (reset && v) {
        const u64 z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(fc_fault_count), &z, sizeof(z));
    }
    return (long)v;
}
