// fps.hip -- farthest-point sampling for gfx950 (MI355X).
//
// Replaces sampling.furthest_sampling (reference: sampling/sampling_cuda.cu:103-265).  The
// reference streams xyz and temp from global memory every round from ONE thread block per
// batch element.  Here the whole point set of a batch element lives in the VGPRs of one
// workgroup for the entire call (x, y, z, running distance per point: 16 B/point, up to
// 26 points per lane x 1024 lanes), so a round costs no memory traffic at all: a local
// arg-max scan, a DPP wave reduction, one LDS hand-off across the waves (one s_barrier per
// round, double-buffered slots) and a scalar load of the winner's coordinates.
//
// Result contract (bit-exact with the oracle, oracle/ref_kernels.c orc_fps_f32):
//   * d = fma(dz,dz,fma(dx,dx,dy*dy)), temp = fminf(d, temp)
//   * winner = max temp; among equals the smallest (k mod bs), then smallest k -- the outcome
//     of the reference's strided scan + left-biased tree.  Lane t owns k = t, t+W, t+2W, ...
//     with W a multiple of bs, so (k mod bs) is constant per lane and a strict '>' scan in
//     slot order already yields the lane's winner; waves then reduce (distance, tie key).
#include "tpu3_dev.h"

#include <cstdlib>

// fps_bucket.hip: exact work-skipping kernel for point sets beyond the register-resident limit
size_t tpu3_fps_bucket_workspace_bytes(int b, int n);
int tpu3_fps_bucket_launch(hipStream_t s, int b, int n, int m, const int32_t *n_arr, const int32_t *m_arr,
                           const float *xyz, float *temp, int32_t *idx, void *workspace, size_t workspace_bytes);

namespace {

constexpr int FPS_RESIDENT_MAX = 25600;

// Point sets of at least this size whose sample count is large enough take the pruned kernels of
// fps_bucket.hip (Morton order + exact AABB pruning: rows in registers up to 25 600 points, buckets
// in L2 beyond) instead of the plain register-resident kernel below, which updates every point in
// every round.  TPU3_FPS_BUCKET_MIN_N overrides the threshold (results are identical either way).
int fps_bucket_min_n()
{
    static const int v = [] {
        const char *e = getenv("TPU3_FPS_BUCKET_MIN_N");
        return e ? atoi(e) : 4096;
    }();
    return v;
}
constexpr int FPS_BUCKET_MIN_M = 256;

struct FpsArgs {
    int n, m;                 // padded sizes (strides)
    const int32_t *n_arr;     // optional live sizes
    const int32_t *m_arr;
    const float *xyz;         // (b,n,3)
    float *temp;              // (b,n)
    int32_t *idx;             // (b,m)
};

// Cross-wave arg-max hand-off.  slots[parity][wave] = {distance bits, tie key}.
template <int NW>
struct FpsShared {
    int d[2][NW];
    uint32_t key[2][NW];
};

// reduce (dbits, key) over the workgroup; returns the winning point index (uniform)
template <int NW>
__device__ __forceinline__ int fps_block_argmax(FpsShared<NW> &sh, int parity, int dbits,
                                                uint32_t key, int lb)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // fused-DPP max + ballot; the tie key is only reduced when several lanes share the maximum
    int wl;
    const int wmax = tpu3_wave_argmax(dbits, key, wl);
    const uint32_t wkey = (uint32_t)__builtin_amdgcn_readlane((int)key, wl);
    if (NW == 1)
        return tpu3_fps_tiekey_to_index(wkey, lb);
    if (lane == 0) {
        sh.d[parity][wave] = wmax;
        sh.key[parity][wave] = wkey;
    }
    __syncthreads();
    const int bd = lane < NW ? sh.d[parity][lane] : (int)0x80000000;
    const uint32_t bk = lane < NW ? sh.key[parity][lane] : 0xFFFFFFFFu;
    const int rmax = __builtin_amdgcn_readlane(tpu3_row_max_i32_fast(bd), 0);     // NW <= 16: one DPP row
    unsigned long long who = __ballot(lane < NW && bd == rmax);
    if (__builtin_popcountll(who) != 1) {
        const uint32_t rk = tpu3_row_min_u32(lane < NW && bd == rmax ? bk : 0xFFFFFFFFu);
        const uint32_t win = (uint32_t)__builtin_amdgcn_readlane((int)rk, 0);
        who = __ballot(lane < NW && bd == rmax && bk == win);
    }
    const int ww = __builtin_ctzll(who | (1ull << 63)) & (NW - 1);
    return tpu3_fps_tiekey_to_index(sh.key[parity][ww], lb);
}

// ---- register-resident kernel: n <= W * PPT ---------------------------------------------
template <int W, int PPT>
__global__ __launch_bounds__(W) void fps_resident_kernel(FpsArgs a)
{
    constexpr int NW = W / 64;
    __shared__ FpsShared<NW> sh;
    const int b = blockIdx.x;
    const int n = a.n_arr ? a.n_arr[b] : a.n;
    const int m = a.m_arr ? a.m_arr[b] : a.m;
    if (m <= 0 || n <= 0)
        return;
    const float *__restrict__ P = a.xyz + (size_t)b * a.n * 3;
    float *__restrict__ T = a.temp + (size_t)b * a.n;
    int32_t *__restrict__ I = a.idx + (size_t)b * a.m;
    const int t = threadIdx.x;
    const int lb = tpu3_fps_log2_bs(n);

    float px[PPT], py[PPT], pz[PPT], pt[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int k = t + j * W;
        if (k < n) {
            px[j] = P[k * 3 + 0];
            py[j] = P[k * 3 + 1];
            pz[j] = P[k * 3 + 2];
            pt[j] = T[k];
        } else {            // padding never wins: fminf(d, -1) == -1 sorts below every d >= 0
            px[j] = py[j] = pz[j] = 0.f;
            pt[j] = -1.0f;
        }
    }
    int old = 0;
    if (t == 0)
        I[0] = 0;
    for (int r = 1; r < m; ++r) {
        const float x1 = P[old * 3 + 0], y1 = P[old * 3 + 1], z1 = P[old * 3 + 2];
        float best = -1.0f;
        int bj = 0;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const float d = tpu3_sqdist3(px[j] - x1, py[j] - y1, pz[j] - z1);
            const float d2 = tpu3_min1(d, pt[j]);
            pt[j] = d2;
            if (d2 > best) {
                best = d2;
                bj = j;
            }
        }
        const uint32_t key = tpu3_fps_tiekey(t + bj * W, lb);
        old = fps_block_argmax<NW>(sh, r & 1, __float_as_int(best), key, lb);
        if (t == 0)
            I[r] = old;
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int k = t + j * W;
        if (k < n)
            T[k] = pt[j];
    }
}

// ---- streaming kernel: any n; xyz and temp stay in global memory (L2) -----------------------
// Correct for every size; used above the resident limit until the bucketed kernel takes over.
template <int W>
__global__ __launch_bounds__(W) void fps_stream_kernel(FpsArgs a)
{
    constexpr int NW = W / 64;
    __shared__ FpsShared<NW> sh;
    const int b = blockIdx.x;
    const int n = a.n_arr ? a.n_arr[b] : a.n;
    const int m = a.m_arr ? a.m_arr[b] : a.m;
    if (m <= 0 || n <= 0)
        return;
    const float *__restrict__ P = a.xyz + (size_t)b * a.n * 3;
    float *__restrict__ T = a.temp + (size_t)b * a.n;
    int32_t *__restrict__ I = a.idx + (size_t)b * a.m;
    const int t = threadIdx.x;
    const int lb = tpu3_fps_log2_bs(n);
    int old = 0;
    if (t == 0)
        I[0] = 0;
    for (int r = 1; r < m; ++r) {
        const float x1 = P[old * 3 + 0], y1 = P[old * 3 + 1], z1 = P[old * 3 + 2];
        float best = -1.0f;
        int bk = t;
        for (int k = t; k < n; k += W) {
            const float td = T[k];
            const float d = tpu3_sqdist3(P[k * 3 + 0] - x1, P[k * 3 + 1] - y1, P[k * 3 + 2] - z1);
            const float d2 = tpu3_min1(d, td);
            if (d2 != td)
                T[k] = d2;
            if (d2 > best) {
                best = d2;
                bk = k;
            }
        }
        old = fps_block_argmax<NW>(sh, r & 1, __float_as_int(best), tpu3_fps_tiekey(bk, lb), lb);
        if (t == 0)
            I[r] = old;
    }
}

template <int W, int PPT>
int launch_resident(hipStream_t s, int b, const FpsArgs &a)
{
    hipLaunchKernelGGL((fps_resident_kernel<W, PPT>), dim3(b), dim3(W), 0, s, a);
    return tpu3_launch_status();
}

} // namespace

extern "C" size_t tpu3_fps_workspace_bytes(int b, int n)
{
    if (b <= 0 || (n <= FPS_RESIDENT_MAX && n < fps_bucket_min_n()))
        return 0;
    return tpu3_fps_bucket_workspace_bytes(b, n);
}

extern "C" int tpu3_fps_ragged_f32(tpu3_stream_t stream, int b, int n, int m, const int32_t *n_arr,
                                   const int32_t *m_arr, const float *xyz, float *temp,
                                   int32_t *idx, void *workspace, size_t workspace_bytes)
{
    if (b < 0 || n < 0 || m < 0)
        return TPU3_EINVAL;
    if (b == 0 || n == 0 || m == 0)
        return TPU3_OK;
    if (!xyz || !temp || !idx)
        return TPU3_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    FpsArgs a{n, m, n_arr, m_arr, xyz, temp, idx};
    // W must be a multiple of bs = min(512, 2^floor(log2 n)) (tie rule, see header comment):
    // n < 512 -> bs <= 256 -> W = 256; otherwise W >= 512.
    if (n > FPS_RESIDENT_MAX || (n >= fps_bucket_min_n() && m >= FPS_BUCKET_MIN_M)) {
        // Morton buckets + exact pruning (fps_bucket.hip)
        const size_t need = tpu3_fps_bucket_workspace_bytes(b, n);
        if (need) {
            if (workspace && workspace_bytes >= need)
                return tpu3_fps_bucket_launch(s, b, n, m, n_arr, m_arr, xyz, temp, idx, workspace, workspace_bytes);
            void *ws = nullptr;                 // caller gave no scratch: stream-ordered allocation
            hipError_t e = hipMallocAsync(&ws, need, s);
            if (e != hipSuccess) return (int)e;
            const int r = tpu3_fps_bucket_launch(s, b, n, m, n_arr, m_arr, xyz, temp, idx, ws, need);
            e = hipFreeAsync(ws, s);
            return r ? r : (int)e;
        }
    }
    if (n <= 256) return launch_resident<256, 1>(s, b, a);
    if (n < 512) return launch_resident<256, 2>(s, b, a);
    if (n <= 1024) return launch_resident<512, 2>(s, b, a);
    if (n <= 2048) return launch_resident<512, 4>(s, b, a);
    if (n <= 3072) return launch_resident<1024, 3>(s, b, a);
    if (n <= 4096) return launch_resident<1024, 4>(s, b, a);
    if (n <= 6144) return launch_resident<1024, 6>(s, b, a);
    if (n <= 8192) return launch_resident<1024, 8>(s, b, a);
    if (n <= 12288) return launch_resident<1024, 12>(s, b, a);
    if (n <= 16384) return launch_resident<1024, 16>(s, b, a);
    if (n <= 20480) return launch_resident<1024, 20>(s, b, a);
    if (n <= FPS_RESIDENT_MAX) return launch_resident<1024, 25>(s, b, a);
    hipLaunchKernelGGL((fps_stream_kernel<1024>), dim3(b), dim3(1024), 0, s, a);
    return tpu3_launch_status();
}

extern "C" int tpu3_fps_f32(tpu3_stream_t stream, int b, int n, int m, const float *xyz,
                            float *temp, int32_t *idx)
{
    return tpu3_fps_ragged_f32(stream, b, n, m, nullptr, nullptr, xyz, temp, idx, nullptr, 0);
}
