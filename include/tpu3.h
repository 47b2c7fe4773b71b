/*
 * include/tpu3.h -- C ABI of lib3pu_hip.so, the MI355X (gfx950) implementation of the
 * 3PU patch-upsampling hot path: farthest-point sampling, index gather, ball query,
 * brute-force kNN grouping and the Chamfer "nm-distance" forward/backward.
 *
 * The entry points are what the reference's two pybind11 extension modules bind
 * (`sampling`: sampling/sampling.cpp:83-88, `losses`: losses/nmdistance.cpp:24-27) plus the
 * kNN grouping that the reference does in torch (network/operations.py:151-216).  Plain
 * pointers and sizes only -- no torch types.  All pointers are DEVICE pointers unless said
 * otherwise; `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).
 * Every function enqueues work on `stream` and returns immediately (no host synchronisation),
 * like the reference's launches.
 *
 * Return value: 0 on success, a negative TPU3_E* code for a rejected argument, or a positive
 * hipError_t when the launch failed (the reference prints and exit(-1)s instead,
 * sampling/cuda_utils.h:26-37).
 *
 * "Ragged" batches: functions that take `n_arr` / `m_arr` accept NULL (every batch element
 * uses the full n / m) or device arrays of `b` int32 holding the live point / query count of
 * each batch element inside its padded (n, m)-sized slab.  The reference has no such thing;
 * it is what lets the patch pipeline run all outer patches of a cloud in one launch after the
 * data-dependent outlier filter (network/upsampler.py:63-80) made their sizes differ.
 */
#ifndef TPU3_H
#define TPU3_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TPU3_OK 0
#define TPU3_EINVAL (-1)   /* bad size / NULL pointer / unsupported element size */
#define TPU3_ELIMIT (-2)   /* size beyond what this build supports             */

typedef void *tpu3_stream_t; /* hipStream_t */

/* arithmetic of the matrix-core (MFMA) kernels of the per-patch MLP stacks: an explicit argument of
 * every such entry point, never chosen silently */
#define TPU3_MFMA_F32 0    /* fp32 inputs, fp32 accumulate (v_mfma_f32_16x16x4_f32): the default path   */
#define TPU3_MFMA_F16 1    /* inputs rounded to fp16, fp32 accumulate (v_mfma_f32_16x16x{16,32}_f16)    */

/* Storage of a Level's feature buffer (the (B,N,264) dense concatenation, network/upsampler.py:293-311) for config C5
 * (BASELINE.json configs[4]: "fp16 feature MLPs"): the *_st_* entry points below read / write its rows in either type,
 * everything else about them is the entry point without the suffix.  Strides count elements of the stored type. */
#define TPU3_STORE_F32 0
#define TPU3_STORE_F16 1

/* Library identification: "3pu-hip <version> gfx950". */
const char *tpu3_version(void);

/* Human-readable text for a return code of this library (static storage). */
const char *tpu3_strerror(int code);

/* sampling.furthest_sampling  (sampling/sampling.cpp:26-35,85; kernel
 * sampling/sampling_cuda.cu:103-174).
 *   xyz  (b,n,3) f32 contiguous
 *   temp (b,n)   f32 in/out: running squared distance to the chosen set; the caller
 *                pre-fills it (1e10 in network/operations.py:291); on return it holds the
 *                distances after the last update, as the reference leaves them
 *   idx  (b,m)   i32 out; idx[:,0] = 0
 * Tie rule identical to the reference's block reduction: among equal maxima the smallest
 * (k mod bs), then the smallest k, with bs = largest power of two <= n clamped to [1,512]
 * (sampling/cuda_utils.h:9-14).  Unlike the reference (temp indexed by blockIdx.x,
 * sampling_cuda.cu:131,146) every batch element uses its own temp row, for any b. */
int tpu3_fps_f32(tpu3_stream_t stream, int b, int n, int m, const float *xyz, float *temp,
                 int32_t *idx);

/* Ragged form: element i samples m_arr[i] (<= m) of its first n_arr[i] (<= n) points.
 * `workspace` may be NULL or a device buffer of tpu3_fps_workspace_bytes(b, n) bytes; it is
 * only needed when n exceeds the register-resident limit. */
int tpu3_fps_ragged_f32(tpu3_stream_t stream, int b, int n, int m, const int32_t *n_arr,
                        const int32_t *m_arr, const float *xyz, float *temp, int32_t *idx,
                        void *workspace, size_t workspace_bytes);
size_t tpu3_fps_workspace_bytes(int b, int n);

/* sampling.gather_forward  (sampling/sampling.cpp:37-45,86; sampling_cuda.cu:28-41):
 * out[b,c,j] = points[b,c,idx[b,j]].  elem_size = 2, 4 or 8 bytes (half/float/double: the
 * reference dispatches on AT_DISPATCH_FLOATING_TYPES_AND_HALF, :48; the op is a copy). */
int tpu3_gather_fwd(tpu3_stream_t stream, int b, int c, int n, int m, int elem_size,
                    const void *points, const int32_t *idx, void *out);

/* sampling.gather_backward  (sampling/sampling.cpp:47-53,87; sampling_cuda.cu:66-80):
 * grad_points[b,c,idx[b,j]] += grad_out[b,c,j] (atomic); grad_points is zeroed by the
 * caller (network/operations.py:257-258).  elem_size = 2 (f16), 4 (f32) or 8 (f64). */
int tpu3_gather_bwd(tpu3_stream_t stream, int b, int c, int n, int m, int elem_size,
                    const void *grad_out, const int32_t *idx, void *grad_points);

/* sampling.ball_query  (sampling/sampling.cpp:59-81,88; sampling_cuda.cu:269-317).
 * query (b,m,3), xyz (b,n,3) f32 (elem_size 4) or f64 (elem_size 8); idx (b,m,nsample) i32
 * out.  The binding of the reference allocates idx as zeros (:69-71); this function zeroes
 * it on the stream before the kernel, so queries without a hit read 0. */
int tpu3_ball_query(tpu3_stream_t stream, int b, int n, int m, float radius, int nsample,
                    int elem_size, const void *query, const void *xyz, int32_t *idx);

/* losses.nmdistance_forward  (losses/nmdistance.cpp:12-14,25; nmdistance_cuda.cu:11-153).
 * xyz1 (b,n,3), xyz2 (b,m,3) f32 -> dist1 (b,n), idx1 (b,n) i32: squared distance to and
 * index of the nearest point of the other set (lowest index on exact ties); dist2/idx2 the
 * same for xyz2.  Returns 0 on success (the reference returns 1/0, ignored by its caller;
 * the Python mirror `losses.nmdistance_forward` maps 0 -> 1). */
int tpu3_nmdist_fwd_f32(tpu3_stream_t stream, int b, int n, int m, const float *xyz1,
                        const float *xyz2, float *dist1, float *dist2, int32_t *idx1,
                        int32_t *idx2);

/* Which kernel family tpu3_nmdist_fwd_f32 uses for the calls that follow: -1 = automatic (the brute-force scan of the
 * reference for small sets, the grid-pruned search of csrc/nmdist_grid.hip -- same distances, same indices, exact ties
 * included -- when both sets hold >= 2048 points and n * m >= 1.6e7), 0 = always the scan, 1 = the grid form whenever
 * both sets hold >= 128 points.  Returns the previous setting.  tpu3_debug_nmdist_grid_calls: forward calls that took
 * the grid form since the last reset (tests assert that they exercise it). */
int tpu3_debug_nmdist_form(int form);
/* tpu3_debug_nmdist_grid_stats: grid-form calls that follow ADD to four device u64 words: query waves, super-tiles and
 * tiles that passed a wave's bound, tiles searched (NULL switches the probe off; tools/chamfer_probe.py). */
int tpu3_debug_nmdist_grid_stats(unsigned long long *stats);
long tpu3_debug_nmdist_grid_calls(int reset);

/* losses.nmdistance_backward  (losses/nmdistance.cpp:17-21,26; nmdistance_cuda.cu:154-193).
 * Adds into caller-zeroed gradxyz1 (b,n,3), gradxyz2 (b,m,3). */
int tpu3_nmdist_bwd_f32(tpu3_stream_t stream, int b, int n, int m, const float *xyz1,
                        const float *xyz2, float *gradxyz1, float *gradxyz2,
                        const float *graddist1, const float *graddist2, const int32_t *idx1,
                        const int32_t *idx2);

/* ChamferLoss reduction (network/model_loss.py:64-84) next to the nm-distance call: from the two
 * distance rows of every batch element e
 *   cd[e] = forward_weight * mean_i(keep1 * dist1[e,i]) + mean_j(keep2 * dist2[e,j]),
 *   keep = 1, or with use_threshold != 0: dist < threshold * mean(dist row)   (:67-77),
 *   loss[0] = mean_e cd[e]                                                    (:80-84),
 * and, when gw1 (b,n) / gw2 (b,m) are not NULL, the derivative of loss with respect to every
 * distance (forward_weight * keep1 / (n b), keep2 / (m b)) -- what the backward pass multiplies into
 * tpu3_nmdist_bwd_f32.  Fixed summation order (deterministic); cd (b) f32 is scratch/out. */
int tpu3_chamfer_reduce_f32(tpu3_stream_t stream, int b, int n, int m, const float *dist1,
                            const float *dist2, int use_threshold, float threshold,
                            float forward_weight, float *loss, float *cd, float *gw1, float *gw2);

/* Optional batch layout of a kNN call (host struct, pointers inside are DEVICE pointers).
 * NULL layout = the reference's dense call: b query sets, b point sets, one group.
 *   n_arr   (bp)  live points of each point set inside its padded n-slab, or NULL
 *   m_arr   (b)   live queries of each query set, or NULL (padded queries give unspecified rows)
 *   pts_of  (b)   which point set each query set searches, or NULL (identity, bp == b).  This is
 *                 the reference's `previous_xyz.expand(batch_size, -1, -1)` (upsampler.py:319-323)
 *                 without materialising the copies
 *   grp     (b)   group of each query set, or NULL (one group).  unique=True adds max(D) taken
 *                 over one reference call's whole batch (operations.py:204); a launch that fuses
 *                 several reference calls keeps one maximum per group
 *   bp, groups    number of point sets / groups (ignored when the pointer is NULL) */
typedef struct {
    const int32_t *n_arr;
    const int32_t *m_arr;
    const int32_t *pts_of;
    const int32_t *grp;
    int bp;
    int groups;
    /* optional, unique=True only: per point set the ascending indices of its first occurrences
     * (bp,n) and their number (bp), from tpu3_knn_unique_compact_i32.  tpu3_knn_f32 then visits
     * only those rows (same result; the merged cloud of overlapping patches holds every point ~5x). */
    const int32_t *cand;
    const int32_t *cand_count;
    /* optional, with cand / cand_count, 3-d points and k <= 8 (the inter-level search, fm_knn = 5): the candidates in
     * Morton-ordered tiles of 64 with their boxes, from tpu3_knn_tiles_build_f32.  tpu3_knn_f32 then searches, per
     * wave of 64 queries, only the tiles whose box can hold a neighbour (same result, ~8 of ~310 tiles). */
    const float *tile_pts;          /* (bp, tiles*64, 4): x, y, z, |p|^2       tiles = ceil(n / 64) */
    const int32_t *tile_idx;        /* (bp, tiles*64): row of each member, -1 beyond the list */
    const float *tile_box;          /* (bp, tiles, 8): lo xyz, hi xyz, max |p|^2, members */
} tpu3_knn_layout;

/* Number of u32 words of the unique=True scratch `uws` for `groups` groups (>= 1). */
#define TPU3_KNN_UWS_WORDS(groups) (4 + (groups))

/* kNN grouping = network.operations.group_knn (network/operations.py:151-216) without the
 * (B,M,N) distance matrix.  Channel-last inputs: query (b,m,c), points (bp,n,c) f32.
 *   D[q,p] = fmaf(-2, <q,p>, |q|^2) + |p|^2, dot and norms as ascending-channel fmaf chains
 *   (the reference leaves torch.matmul's order unspecified; this is the oracle's order).
 * k smallest per query, ascending, ties to the lowest index.
 *   idx      (b,m,k) i32 or i64 (idx_elem_size 4 / 8; torch.topk returns int64)
 *   dist     (b,m,k) f32, may be NULL
 *   grouped  (b,m,k,c) f32 gathered neighbours, may be NULL (the reference returns this as a
 *            (b,c,m,k) permuted view of exactly this layout, :209-214)
 *   dup/uws  unique=True support (:192-204): NULL/NULL for unique=False, else the outputs of
 *            tpu3_knn_unique_prepare_f32 for the same query/points/layout.  D += max(D)*dup puts
 *            every duplicate behind every first occurrence unless fewer than k first occurrences
 *            exist; for k <= 64 the call first skips duplicates and lets every query verify that
 *            this is exact for it, and only otherwise (uws[1]) computes max(D) and applies the
 *            reference arithmetic.  Results are identical either way. */
int tpu3_knn_f32(tpu3_stream_t stream, int b, int m, int n, int c, int k, const float *query,
                 const float *points, const tpu3_knn_layout *layout, const uint8_t *dup,
                 uint32_t *uws, void *idx, int idx_elem_size, float *dist, float *grouped);

/* kNN graph for the fused DenseEdgeConv block: the exact top-k SET per query with the nearest
 * neighbour in slot 0 and the other k-1 members in ascending index order (DenseEdgeConv drops the
 * nearest and max-pools over the rest, network/layers.py:33-35,63, so their order is immaterial).
 * Two passes -- the k smallest distances by a v_med3 chain, then the indices by threshold -- instead
 * of a sorted insertion with indices.  idx (b,m,k) i32.  dup/uws as for tpu3_knn_f32; with
 * duplicated rows present the exact sorted kernels produce the (then ordered) result instead.
 * Supported: c <= 32 and k in {17, 33}, else TPU3_ELIMIT. */
int tpu3_knn_graph_f32(tpu3_stream_t stream, int b, int m, int n, int c, int k, const float *query,
                       const float *points, const tpu3_knn_layout *layout, const uint8_t *dup,
                       uint32_t *uws, int32_t *idx);

/* The same graph for a SELF query (x is both query and point set) without a de-duplication
 * pre-pass: identical rows have D == 0 exactly, so the kernel itself notices whether any row could
 * be duplicated; only then the hash de-duplication and the exact kernels run (device-side gates).
 * dup (b,n) u8 and uws (TPU3_KNN_UWS_WORDS(groups) u32) are scratch; workspace =
 * tpu3_knn_unique_workspace_bytes(b, n) bytes (n >= 128).  Slot 0 = the query's own row (the nearest), the other
 * k-1 members of the exact top-k set in no particular order (k < n <= 8192: one pass with the index in the key). */
int tpu3_knn_graph_self_f32(tpu3_stream_t stream, int b, int n, int c, int k, const float *x,
                            const tpu3_knn_layout *layout, uint8_t *dup, uint32_t *uws, int32_t *idx,
                            void *workspace, size_t workspace_bytes);

/* Optimistic form of the call above: only its first pass -- no scratch, no gated fallback
 * launches behind it.  If some query saw a second zero distance (rows may be duplicated) events[2] is set to 1
 * and the result of THAT call is not guaranteed: recompute it with tpu3_knn_graph_self_f32.  events: 4 u32
 * device words zeroed once by the caller and shared by any number of calls (events[0] must stay 0); the caller
 * reads events[2] at its own synchronisation point. */
int tpu3_knn_graph_self_optimistic_f32(tpu3_stream_t stream, int b, int n, int c, int k, const float *x,
                                       const tpu3_knn_layout *layout, uint32_t *events, int32_t *idx);

/* unique=True pre-pass: dup (bp,n) u8 = 1 iff an identical row exists at a smaller index of
 * the same point set (complement of np.unique(axis=0, return_index=True), operations.py:194-200).
 * O(n) per point set (open-addressing table of class representatives) above 1024 points,
 * a quadratic scan below.  uws = TPU3_KNN_UWS_WORDS(groups) device words, zeroed here:
 * [0] any-dup flag, [1] set by tpu3_knn_f32 when its optimistic pass had to be redone,
 * [2..3] reserved, [4+g] max(D) of group g as order-preserving bits (operations.py:204; filled by
 * tpu3_knn_f32 only when the result can depend on it).  `workspace`: NULL (stream-ordered
 * allocation inside) or tpu3_knn_unique_workspace_bytes(bp, n) device bytes. */
int tpu3_knn_unique_prepare_f32(tpu3_stream_t stream, int b, int m, int n, int c,
                                const float *query, const float *points,
                                const tpu3_knn_layout *layout, uint8_t *dup, uint32_t *uws,
                                void *workspace, size_t workspace_bytes);
size_t tpu3_knn_unique_workspace_bytes(int bp, int n);

/* Candidate list of a de-duplicated point set batch (see tpu3_knn_layout.cand): cand (bp,n) i32,
 * cand_count (bp) i32 from dup/uws of tpu3_knn_unique_prepare_f32; n_arr (bp) live sizes or NULL.
 * Left untouched (and never consulted) when no row is duplicated at all (uws[0] == 0). */
int tpu3_knn_unique_compact_i32(tpu3_stream_t stream, int bp, int n, const int32_t *n_arr,
                                const uint8_t *dup, const uint32_t *uws, int32_t *cand,
                                int32_t *cand_count);

/* Spatial tiles of the candidate lists above (see tpu3_knn_layout.tile_*), 3-d points only: per point set a bounding
 * box, 30-bit Morton codes, ONE device radix sort over all sets, then the tiles' rows and boxes.  points (bp,n,3);
 * cand / cand_count / uws as left by tpu3_knn_unique_prepare_f32 + tpu3_knn_unique_compact_i32 (uws[0] == 0: no
 * duplicates, every live row is a candidate); tile_pts (bp*tiles*64*4 f32, 16-byte aligned), tile_idx (bp*tiles*64
 * i32), tile_box (bp*tiles*8 f32, 16-byte aligned) with tiles = ceil(n / 64); workspace =
 * tpu3_knn_tiles_workspace_bytes(bp, n) device bytes.
 * tpu3_knn_tiles_query_f32 is the pruned search itself (the first pass of tpu3_knn_f32, which calls it when the layout
 * carries tiles): query (b,m,3), k <= 8; a query that cannot verify the unique=True rule raises uws[1]. */
int tpu3_knn_tiles_build_f32(tpu3_stream_t stream, int bp, int n, const float *points, const int32_t *n_arr,
                             const int32_t *cand, const int32_t *cand_count, const uint32_t *uws, float *tile_pts,
                             int32_t *tile_idx, float *tile_box, void *workspace, size_t workspace_bytes);
size_t tpu3_knn_tiles_workspace_bytes(int bp, int n);
int tpu3_knn_tiles_query_f32(tpu3_stream_t stream, int b, int m, int n, int k, const float *query,
                             const tpu3_knn_layout *layout, uint32_t *uws, void *idx, int idx_elem_size, float *dist);

/* Fused DenseEdgeConv block, inference (network/layers.py:44-64 for in_channels 24, growth 12,
 * 3 dense layers -- the configuration of every Level, network/upsampler.py:210-223):
 *   y_i = max_j [h2, h1, h0, x_i],  h0 = relu(W0 [x_i, x_j - x_i] + b0),  h1 = relu(W1 [h0, x_i] + b1),
 *   h2 = W2 [h1, h0, x_i] + b2,  j over the k neighbours idx[p, i, idx_off .. idx_off + k).
 *   x   (patches, n, 24) f32 channel-last;  idx (patches, n, idx_stride) i32 / i64, entries in [0, n)
 *   w0 (12,48) b0 (12)  w1 (12,36) b1 (12)  w2 (12,48) b2 (12): the nn.Conv2d weights, row-major
 *   out (patches, n, out_stride) f32: channels [0,60) of every row are written (out_stride >= 60,
 *       multiple of 4, base 16-byte aligned) -- lets the caller place y inside the level's
 *       concatenated feature buffer without a copy.  k must be a multiple of 16, at most 64.
 * Any n: the per-point table the kernel gathers from lives in LDS up to ~2700 points per patch and
 * in global memory beyond (the patch is then split over several workgroups).
 * mfma = TPU3_MFMA_F32: fp32 MFMA (exact fp32 fma chains; the summation order differs from a BLAS GEMM).
 * mfma = TPU3_MFMA_F16: weights and the activations entering a matrix instruction rounded to fp16, fp32
 * accumulate, everything else (bias, ReLU, max, tensors in memory) fp32 -- config C5's "fp16 feature MLPs
 * on MFMA"; results agree with the fp32 flavour to ~1e-2 relative (tests state the bound). */
int tpu3_dense_edge_conv_f32(tpu3_stream_t stream, int patches, int n, int k, const float *x,
                             const void *idx, int idx_elem_size, int idx_stride, int idx_off,
                             const float *w0, const float *b0, const float *w1, const float *b1,
                             const float *w2, const float *b2, float *out, int out_stride, int mfma);
/* The same block writing its rows into a feature buffer stored as `out_store` (TPU3_STORE_F16: mfma must be
 * TPU3_MFMA_F16, out 8-byte aligned, out_stride in halves). */
int tpu3_dense_edge_conv_st_f32(tpu3_stream_t stream, int patches, int n, int k, const float *x, const void *idx,
                                int idx_elem_size, int idx_stride, int idx_off, const float *w0, const float *b0,
                                const float *w1, const float *b1, const float *w2, const float *b2, void *out,
                                int out_stride, int mfma, int out_store);

/* The same block with the NEXT prep convolutions folded into its write-out (fp32, lane-per-point kernel; reference
 * network/upsampler.py:298-311: layer{2,3,4}_prep = Conv1d(84 / 144 / 204 -> 24) + ReLU over the level's dense
 * concatenation).  A prep convolution is linear in the concatenated row, so this block's 60 output channels
 * contribute fold_w . row to each later one while the row is still in registers, and the concatenation is never
 * re-read:
 *   t < fold_n:  s[t] = (fold_b ? fold_b[t] : acc[p, seed_off + t]) + sum_c fold_w[t, c] * y[p, c]      c < 60
 *   t < 24:      xnext[p, t] = max(s[t], 0)        -- the next block's input rows, (patches, n, 24) contiguous
 *   t >= 24:     acc[p, store_off + t - 24] = s[t] -- partial sums of the blocks after the next
 * fold_n in {24, 48, 72}; fold_w (fold_n, 60) row-major: the columns of this block's [y | x] channels in each later
 * prep convolution's weight (the caller adds the columns of a second copy of x, e.g. the buffer's x0 tail);
 * acc (patches, n, acc_stride).  TPU3_ELIMIT when the patch's tables do not fit LDS: run the layers unfolded. */
int tpu3_dense_edge_conv_fold_f32(tpu3_stream_t stream, int patches, int n, int k, const float *x,
                                  const void *idx, int idx_elem_size, int idx_stride, int idx_off,
                                  const float *w0, const float *b0, const float *w1, const float *b1,
                                  const float *w2, const float *b2, float *out, int out_stride, int fold_n,
                                  const float *fold_w, const float *fold_b, float *acc, int acc_stride,
                                  int seed_off, int store_off, float *xnext);

/* (r5) The same two launches with the weights PACKED once per set of weights instead of re-arranged by every
 * workgroup of every launch.  The lane-per-point kernel keeps its A operands in LDS tables and seven registers; built
 * in place that is ~1500 integer / LDS instructions per wave which, on a compute unit whose other workgroups are
 * inside their MFMA loops, take 35 - 45 k cycles of a wave's 180 k.  tpu3_dense_edge_conv_pack_f32 (one small launch)
 * writes them to `pack` (tpu3_dense_edge_conv_pack_floats(fold_n) floats, 16-byte aligned; fold_n = 0 for a block
 * without folded prep convolutions); the *_pk_* launches copy the blob and are otherwise the calls above, bit for
 * bit (fp32 lane-per-point form only: TPU3_ELIMIT when the patch's tables do not fit LDS -- pass the weights then).
 * The caller rebuilds the blob when a weight changes. */
size_t tpu3_dense_edge_conv_pack_floats(int fold_n);
int tpu3_dense_edge_conv_pack_f32(tpu3_stream_t stream, const float *w0, const float *b0, const float *w1,
                                  const float *b1, const float *w2, const float *b2, int fold_n,
                                  const float *fold_w, float *pack);
int tpu3_dense_edge_conv_pk_f32(tpu3_stream_t stream, int patches, int n, int k, const float *x,
                                const void *idx, int idx_elem_size, int idx_stride, int idx_off,
                                const float *pack, float *out, int out_stride);
int tpu3_dense_edge_conv_fold_pk_f32(tpu3_stream_t stream, int patches, int n, int k, const float *x,
                                     const void *idx, int idx_elem_size, int idx_stride, int idx_off,
                                     const float *pack, float *out, int out_stride, int fold_n,
                                     const float *fold_b, float *acc, int acc_stride, int seed_off,
                                     int store_off, float *xnext);

/* Fused inter-level skip connection of a Level, inference (network/upsampler.py:317-347):
 *   w_k = exp(-|p_i - q_k|^2 / (h_s/2)) * exp(-|x_i - f_k|^2 / (h_f/2)),  h = mean_i min_k (dist),
 *   w_k /= sum_k (w_k + 1e-5),  x_i += scale * sum_k w_k f_k          (scale = 0.2 in the reference)
 * xyz (b,n,3) patch coordinates; feat (b,n,feat_stride) in/out, the first c channels are x_i;
 * prev_xyz (bp,m,3), prev_feat (bp,m,c) previous level's merged cloud; pts_of (b) i32 maps a patch
 * to its previous cloud (NULL = identity); idx (b,n,k) i32/i64 neighbour rows (from tpu3_knn_f32,
 * k <= 8, c <= 320).  Nothing of size (b,n,k,c) is materialised.  patches_per_cloud > 0 states
 * that patches [i*ppc, (i+1)*ppc) share previous cloud pts_of[i*ppc] (a scheduling hint that keeps a
 * cloud's gathers in one XCD's L2; the result does not depend on it), 0 = unknown. */
int tpu3_interlevel_skip_f32(tpu3_stream_t stream, int b, int n, int k, int c, const float *xyz,
                             float *feat, int feat_stride, const float *prev_xyz,
                             const float *prev_feat, int m, const int32_t *pts_of, const void *idx,
                             int idx_elem_size, float scale, int patches_per_cloud, void *workspace,
                             size_t workspace_bytes);
/* scratch of the call above ((2k+2) floats per point); workspace may be NULL (stream-ordered
 * allocation inside) */
size_t tpu3_interlevel_skip_workspace_bytes(int b, int n, int k);

/* tpu3_interlevel_skip_f32 on a feature buffer stored as `store` (feat AND prev_feat; fp32 arithmetic on the widened
 * rows, the updated row rounded on its way back).  TPU3_STORE_F16: c % 4 == 0, c <= 288, 8-byte aligned rows (else
 * TPU3_ELIMIT). */
int tpu3_interlevel_skip_st_f32(tpu3_stream_t stream, int b, int n, int k, int c, const float *xyz, void *feat,
                                int feat_stride, const float *prev_xyz, const void *prev_feat, int m,
                                const int32_t *pts_of, const void *idx, int idx_elem_size, float scale,
                                int patches_per_cloud, void *workspace, size_t workspace_bytes, int store);

/* The same skip connection as ONE autograd node for training (model.py:53-66 runs network/upsampler.py:317-347 under
 * autograd).  The reference detaches both distances (:244-245), so the weights are constants of the step:
 *   forward  = tpu3_interlevel_skip_f32 that also stores the normalised weights, weights (b,n,k);
 *   backward : the gradient of x_i is g_i itself; gprev (bp,m,c), ZEROED BY THE CALLER, accumulates
 *              scale * weights[b,i,k] * g[b,i,:] at row idx[b,i,k] of previous cloud pts_of[b] (hardware float atomics,
 *              summation order not fixed).  g (b,n,c) contiguous. */
int tpu3_interlevel_skip_train_f32(tpu3_stream_t stream, int b, int n, int k, int c, const float *xyz, float *feat,
                                   int feat_stride, const float *prev_xyz, const float *prev_feat, int m,
                                   const int32_t *pts_of, const void *idx, int idx_elem_size, float scale,
                                   float *weights, void *workspace, size_t workspace_bytes);
int tpu3_interlevel_skip_bwd_f32(tpu3_stream_t stream, int b, int n, int k, int c, const float *g,
                                 const float *weights, int m, const int32_t *pts_of, const void *idx,
                                 int idx_elem_size, float scale, float *gprev);

/* Per-point linear layer with a small output width, inference (the "prep" convolutions of a
 * Level, network/upsampler.py:298,303,308: 84 / 144 / 204 -> 24 + ReLU; any kernel-size-1
 * nn.Conv1d / nn.Conv2d on channel-last data, network/layers.py:115-204):
 *   y[i, 0..cout) = act(W x[i, 0..cin) + bias),  x (m, x_stride) rows, y (m, y_stride) rows,
 *   w (cout, cin) row-major = the convolution weight, bias (cout) or NULL, relu != 0 -> ReLU.
 * cout <= 32 (TPU3_MFMA_F16: <= 128 -- in that mode the per-point half of up_layer1, 264 -> 128, runs here too);
 * cin <= 320; cin, cout and both strides multiples of 4, 16-byte aligned bases (else TPU3_ELIMIT: callers then use
 * their generic GEMM path).  mfma = TPU3_MFMA_F32 / TPU3_MFMA_F16 (fp16 operands, fp32 accumulate; rows stay fp32
 * in memory). */
int tpu3_linear_small_f32(tpu3_stream_t stream, long m, int cin, int cout, const float *x, int x_stride,
                          const float *w, const float *bias, int relu, float *y, int y_stride, int mfma);
/* The same layer reading its input rows from a feature buffer stored as `x_store` (TPU3_STORE_F16: mfma must be
 * TPU3_MFMA_F16 -- the fp16 value that enters the matrix instruction is the stored one; x 8-byte aligned, x_stride in
 * halves).  y stays fp32. */
int tpu3_linear_small_st_f32(tpu3_stream_t stream, long m, int cin, int cout, const void *x, int x_stride,
                             const float *w, const float *bias, int relu, float *y, int y_stride, int mfma,
                             int x_store);

/* Per-point linear layer with a WIDE output, inference: the per-point half of up_layer1
 * (network/upsampler.py:222 `up_layer1 = Conv2d(265, 128, ...)`, applied at :363 to [features ; code]: the first
 * 264 input channels are the same for the r replicas of a point, so W[:, :264] x_i + bias is computed once per
 * point and handed to tpu3_regress_tail_f32 as `a`):
 *   y[i, 0..cout) = W x[i, 0..cin) + bias,   w (cout, w_stride) row-major with w_stride >= cin (a column slice of
 *   the convolution weight is fine), bias (cout) or NULL.
 * Instantiated for cout = 128 and 256 < cin <= 272, cin and the row strides multiples of 4, x / y / bias 16-byte
 * aligned (else TPU3_ELIMIT: callers then use their library GEMM). */
int tpu3_linear_wide_f32(tpu3_stream_t stream, long m, int cin, int cout, const float *x, int x_stride,
                         const float *w, int w_stride, const float *bias, float *y, int y_stride);

/* (r6) The same layer with every fp32 operand as three bf16 terms, six partial products each on
 * v_mfma_f32_16x16x32_bf16 (the Python side's default since round 6; TPU3_SPLIT_BF16=0 / tpu3_split_bf16(0): off): as
 * accurate as the fp32 kernel against fp64, not bit-identical to it.  tpu3_linear_wide_split_bf16 writes the slab-major image of W[:, :cin] once per set of
 * weights (tpu3_linear_wide_split_bytes(cin) bytes, 16-byte aligned); tpu3_linear_wide_sb_f32 takes it in place of w.
 * cout = 128, cin and x_stride multiples of 8, x 32-byte aligned (else TPU3_ELIMIT). */
/* Arithmetic of the regressor's matrix layers (tpu3_regress_tail_f32 with mfma = TPU3_MFMA_F32, and -- on the Python side
 * -- the choice between tpu3_linear_wide_f32 and tpu3_linear_wide_sb_f32): on = 1 three-term bf16 operands on
 * v_mfma_f32_16x16x32_bf16, on = 0 the fp32 matrix instructions (bit-identical to an fmaf chain), on < 0 query only.
 * Returns the previous setting; the initial one comes from TPU3_SPLIT_BF16. */
int tpu3_split_bf16(int on);
size_t tpu3_linear_wide_split_bytes(int cin);
int tpu3_linear_wide_split_bf16(tpu3_stream_t stream, int cin, int cout, const float *w, int w_stride, void *ws);
int tpu3_linear_wide_sb_f32(tpu3_stream_t stream, long m, int cin, int cout, const float *x, int x_stride,
                            const void *ws, const float *bias, float *y, int y_stride);

/* Per-point linear layer with a handful of INPUT channels, inference: the 3 -> 24 coordinate lift that opens
 * every Level (network/upsampler.py:209 `layer0 = Conv2d(3, 24, [1, 1], activation=None)`, applied at :288):
 *   y[i, 0..cout) = act(W x[i, 0..cin) + bias); optionally the same row is also stored at y2 (the slice of the
 *   level's dense-concatenation buffer the reference builds with torch.cat, :293-311).
 * cin <= 8, cout <= 64 and a multiple of 4, output strides multiples of 4, 16-byte aligned outputs (else
 * TPU3_ELIMIT). */
int tpu3_linear_lift_f32(tpu3_stream_t stream, long m, int cin, int cout, const float *x, int x_stride,
                         const float *w, const float *bias, int relu, float *y, int y_stride, float *y2,
                         int y2_stride);

/* Regressor tail of a Level, inference (network/upsampler.py:363-372) for the reference's widths
 * 128 -> 128 -> 64 -> 3: for point i and replica j < r (r <= 4)
 *   out[i*r + j, 0..3) = W4 relu(W3 relu(W2 relu(a_i + c_j) + b2) + b3) + b4 + residual_i
 * a (m,128) = the per-point half of up_layer1 incl. its bias, c (r,128) = its per-replica (code)
 * half, w2 (128,128), w3 (64,128), w4 (3,64) row-major convolution weights, residual (m,3) the
 * normalised input coordinates, out (m*r,3).  One launch instead of three GEMMs and five
 * elementwise passes over (m*r,128) tensors.  mfma = TPU3_MFMA_F32 / TPU3_MFMA_F16 as above. */
int tpu3_regress_tail_f32(tpu3_stream_t stream, long m, int r, const float *a, const float *c,
                          const float *w2, const float *b2, const float *w3, const float *b3,
                          const float *w4, const float *b4, const float *residual, float *out, int mfma);

/* Training: weight gradient of a kernel-size-1 convolution with few outputs over very many rows
 * (the dense layers of DenseEdgeConv, network/layers.py:53-61: 48/36/48 -> 12 channels over
 * B*N*k rows; what autograd computes for nn.Conv2d there):
 *   dw[o][c] = sum_i dy[i][o] * x[i][c],  x (m, x_stride), dy (m, dy_stride), dw (cout, cin) row-major.
 * cout <= 16, cin <= 64 (else TPU3_ELIMIT).  Deterministic (two stages, no atomics); workspace =
 * tpu3_linear_wgrad_workspace_bytes(m) device bytes.  fp32 MFMA. */
int tpu3_linear_wgrad_f32(tpu3_stream_t stream, long m, int cin, int cout, const float *x, int x_stride,
                          const float *dy, int dy_stride, float *dw, void *workspace, size_t workspace_bytes);
size_t tpu3_linear_wgrad_workspace_bytes(long m);

/* Training: weight AND bias gradient of any kernel-size-1 convolution of a Level (network/upsampler.py:209-224:
 * layer0 3 -> 24, layerK_prep 84/144/204 -> 24, up_layer 265 -> 128 -> 128, fc_layer1 128 -> 64, fc_layer2 64 -> 3;
 * what autograd computes for their nn.Conv1d / nn.Conv2d, model.py:53-66):
 *   dw[o][c] = sum_i dy[i][o] * x[i][c],  db[o] = sum_i dy[i][o]   (db may be null)
 * x (m, x_stride), dy (m, dy_stride) rows with unit channel stride, dw (cout, cin) row-major.  cin <= 1023,
 * cout <= 1024 (else TPU3_ELIMIT).  Deterministic (two stages, no atomics); workspace =
 * tpu3_linear_wgrad_bias_workspace_bytes(m, cin, cout) device bytes.  fp32 MFMA. */
int tpu3_linear_wgrad_bias_f32(tpu3_stream_t stream, long m, int cin, int cout, const float *x, int x_stride,
                               const float *dy, int dy_stride, float *dw, float *db, void *workspace,
                               size_t workspace_bytes);
size_t tpu3_linear_wgrad_bias_workspace_bytes(long m, int cin, int cout);

/* Training: input gradient of a kernel-size-1 convolution with FEW outputs (layer0 and the layerK_prep convolutions,
 * 24 outputs; autograd's `grad_output @ weight`):  dx[i][c] = sum_o dy[i][o] * w[o][c],  dy (m, dy_stride), w
 * (cout, cin) row-major, dx (m, dx_stride).  cout <= 32, cin <= 320 (else TPU3_ELIMIT). */
int tpu3_linear_dgrad_f32(tpu3_stream_t stream, long m, int cin, int cout, const float *dy, int dy_stride,
                          const float *w, float *dx, int dx_stride);

/* network.operations.normalize_point_batch (network/operations.py:12-30) on NCHW data:
 * pc (b,3,n) f32 -> out (b,3,n), centroid (b,3), radius (b) ; ragged n_arr optional. */
int tpu3_normalize_f32(tpu3_stream_t stream, int b, int n, const int32_t *n_arr, const float *pc,
                       float *out, float *centroid, float *radius);
/* The same normalisation on channel-last rows: pc, out (b, n, 3); centroid (b, 3), radius (b).  Same operations in the
 * same order as tpu3_normalize_f32 (the eval path keeps its clouds channel-last: no transposes around the call). */
int tpu3_normalize_cl_f32(tpu3_stream_t stream, int b, int n, const int32_t *n_arr, const float *pc, float *out,
                          float *centroid, float *radius);

/* ---- (r6) the small steps between the eval path's kernels, one launch each (csrc/glue.hip) ----------------------------
 * tpu3_repatch_filter_f32: the outlier filter of the eval-mode patch extraction (network/upsampler.py:63-77) for b
 *   clouds of n points: dist = each point's distance to its closest neighbour (element i of cloud e at
 *   dist[(e * n + i) * dstride]); keeps d < 5 * mean(d); xyz_f (b, n, 3) receives the kept points first, in their
 *   order (masked_select), the dropped ones behind them; count (b) = N', patch_num (b) = max(1, int(N' / k * 5)),
 *   old_count = patch_num * k, m_count = patch_num * k * r (either may be NULL); *small_events (may be NULL) is
 *   incremented once per cloud with N' < k.
 * tpu3_repatch_seeds_f32: seeds (b, p, 3) = xyz_f[seed_idx[min(j, patch_num - 1)]] (upsampler.py:78-79 for padded patch
 *   slots: a slot beyond a cloud's patch count repeats its last live patch).
 * tpu3_gather_xyz_f32: out[e, j, :] = x[e, idx[e, j], :] for 3-channel rows and int32 indices; nchw_out != 0 writes
 *   (b, 3, m) instead of (b, m, 3)  (upsampler.py:158, main.py:380).
 * tpu3_denormalize_f32: out = x * radius[patch] + centroid[patch] on (patches, rows_per_patch, 3) rows, a multiplication
 *   and an addition, each rounded (upsampler.py:147, main.py:242).
 * tpu3_fill_f32_i32: a[0 .. na) = va and c[0 .. nc) = vc in one launch (FPS: temp = 1e10, idx = 0). */
int tpu3_repatch_filter_f32(tpu3_stream_t stream, int b, int n, int k, int r, const float *dist, int dstride,
                            const float *xyz, float *xyz_f, int32_t *count, int32_t *patch_num, int32_t *old_count,
                            int32_t *m_count, unsigned long long *small_events);
int tpu3_repatch_seeds_f32(tpu3_stream_t stream, int b, int n, int p, const int32_t *seed_idx, const int32_t *patch_num,
                           const float *xyz_f, float *seeds);
int tpu3_gather_xyz_f32(tpu3_stream_t stream, int b, int n, int m, const float *x, const int32_t *idx, float *out,
                        int nchw_out);
int tpu3_denormalize_f32(tpu3_stream_t stream, long patches, int rows_per_patch, const float *x, const float *radius,
                         const float *centroid, float *out);
int tpu3_fill_f32_i32(tpu3_stream_t stream, float *a, long na, float va, int32_t *c, long nc, int32_t vc);


/* DenseEdgeConv block for TRAINING (network/layers.py:44-64 under autograd, model.py:53-66) for the reference's
 * shape: 24 input channels, growth 12, three layers, k = 32 neighbours (else TPU3_ELIMIT: callers then use their
 * autograd formulation).  x (p,n,24), idx (p,n,idx_stride) i32 with the 32 neighbours of a point at
 * idx_off .. idx_off+31, w0 (12,48) / w1 (12,36) / w2 (12,48) = the convolution weights, b* (12).
 *   fwd: y (p,n,60) = [max_k h2 | max_k h1 | max_k h0 | x_i], arg (p,n,36) u8 = the neighbour slot attaining each max.
 *   bwd: gy (p,n,60) with rows gy_stride >= 60 floats apart (a channel slice of a wider gradient) -> gx (p,n,24) ACCUMULATED with hardware float atomics (zeroed by the caller),
 *        S (p*n, 36) = [g2 | g1 | g0] summed over a point's edges (weight gradients of the x_i parts = S^T X, bias
 *        gradients = column sums of S), and into `workspace` one block G^T Z per workgroup of the launch -- the weight
 *        gradients of the edge parts (G = [g2 | g1 | g0], Z = [h1 | h0 | x_j - x_i] per edge, accumulated on the matrix
 *        cores inside the kernel; the edge tensors themselves never reach memory).  `workspace` =
 *        tpu3_dec_train_wgrad_workspace_bytes(p*n) bytes, 16-byte aligned, handed on to tpu3_dec_train_wgrad_f32. */
int tpu3_dec_train_fwd_f32(tpu3_stream_t stream, long p, int n, int k, const float *x, const int32_t *idx,
                           int idx_stride, int idx_off, const float *w0, const float *b0, const float *w1,
                           const float *b1, const float *w2, const float *b2, float *y, uint8_t *arg);
int tpu3_dec_train_bwd_f32(tpu3_stream_t stream, long p, int n, int k, const float *x, const int32_t *idx,
                           int idx_stride, int idx_off, const float *w0, const float *b0, const float *w1,
                           const float *b1, const float *w2, const float *b2, const uint8_t *arg, const float *gy,
                           int gy_stride, float *gx, float *S, void *workspace, size_t workspace_bytes);
/* Weight and bias gradients of the block from what tpu3_dec_train_bwd_f32 leaves behind (what autograd computes for the
 * three nn.Conv2d of network/layers.py:53-61): gw0 (12,48), gw1 (12,36), gw2 (12,48) in the layers' own column order,
 * gb (36) = [b2 | b1 | b0].  points = p * n; `workspace` = the one the backward call filled.  Deterministic (the
 * workgroups' blocks are added in a fixed order). */
int tpu3_dec_train_wgrad_f32(tpu3_stream_t stream, long points, const float *x, const float *S, float *gw0, float *gw1,
                             float *gw2, float *gb, void *workspace, size_t workspace_bytes);
size_t tpu3_dec_train_wgrad_workspace_bytes(long points);

/* The differentiable neighbour gather of group_knn (network/operations.py:209-211: torch.gather over the expanded
 * point tensor; its backward is an index accumulation) on channel-last rows, training:
 *   tpu3_gather_rows_f32       out[b, j, :] = x[b, idx[b, j], :]        x (b,n,c), idx (b,m) i32 / i64, out (b,m,c)
 *   tpu3_scatter_add_rows_f32  dx[b, idx[b, j], :] += g[b, j, :]        (hardware float atomics; dx zeroed by the
 *                              caller; the summation order of a row's contributions is not fixed)
 * c % 4 == 0 and 16-byte aligned x / out for the gather (else TPU3_ELIMIT). */
int tpu3_gather_rows_f32(tpu3_stream_t stream, int b, int n, long m, int c, const float *x, const void *idx,
                         int idx_elem_size, float *out);
int tpu3_scatter_add_rows_f32(tpu3_stream_t stream, int b, int n, long m, int c, const float *g, const void *idx,
                              int idx_elem_size, float *dx);

/* ---- measurement hooks (bench.py and tools/ only; a caller of the path never needs them) ------------
 * tpu3_debug_fps_bucket_events: the NEXT bucketed-FPS call (n beyond the register-resident limit)
 * records `start` / `stop` (hipEvent_t created by the caller) on ITS stream immediately before / after
 * its main round-loop kernel, so that exactly that kernel can be timed on whatever stream it runs on.
 * One-shot, host-side state only.
 * tpu3_debug_fps_level_stats: the NEXT FPS call that takes the register-resident multi-sample kernel (per-level
 * resampling, 4096 < n <= 25 600) writes (rounds, samples) of its first set to stats[0..1] and, for the largest
 * sets, per-phase cycle counters of waves 0 and 1 to stats[2..13] and every wave's (update cycles, sample
 * updates) to stats[14..45], wave 0's ranking phases and candidate count to stats[46..50] (52 device words).
 * One-shot. */
int tpu3_debug_fps_bucket_events(void *start, void *stop);
int tpu3_debug_fps_level_stats(unsigned long long *stats);
/* tpu3_debug_fps_tile_stats: the NEXT FPS call that takes the tile-form kernel (25 601 .. 4 194 304 points: a lane
 * per 16-point bucket) writes 16 device words: rounds, samples, threshold bisections, rounds with equal maxima among
 * the candidates, wave 0's cycles in update / candidate collection / ranking, its tile visits, the update's split
 * (phase 1, barrier, phase 2, barrier), collection passes, candidates listed, candidates ranked, rounds cut by the
 * clearance test (first set of the batch).  One-shot; host-side state only. */
int tpu3_debug_fps_tile_stats(unsigned long long *stats);
/* tpu3_debug_knn_tiles_stats: the NEXT pruned inter-level search (tpu3_knn_tiles_query_f32) ADDS to four device u32
 * words: waves, tiles searched, tiles tested query by query, tiles per point set (summed over the waves).  One-shot;
 * host-side state only. */
int tpu3_debug_knn_tiles_stats(unsigned *words);
/* tpu3_debug_knn_slab_launches: how many self-graph calls took the SLAB form (knn_slab_order_kernel +
 * knn_graph_slab_kernel: k = 33, 24 channels, 64 < n <= 320, a launch large enough for one workgroup per patch) since
 * the last reset -- the dispatcher gives small launches a wave per workgroup and the one-pass kernel, so a test of the
 * slab form must check that it ran it (r6: the r5 slab tests did not). */
long tpu3_debug_knn_slab_launches(int reset);
/* tpu3_debug_fps_cluster: workgroups per point set of the tile-form FPS for the calls that follow: -1 = the default
 * policy (several compute units per set when the launch is small: 16 workgroups per set for up to 4 sets, 8 for up
 * to 8, 32 for one or two sets beyond 2 M points; b * G <= 64), 0 = single-workgroup kernels only, 2 / 4 / ... / 64 =
 * forced wherever the size allows.  Returns the previous
 * setting (also: environment TPU3_FPS_CLUSTER).  With the cluster form, tpu3_debug_fps_tile_stats receives rounds,
 * samples, tie exchanges, wave 0's poll sweeps and the launch's fault count in stats[0..4].
 * tpu3_debug_fps_plan: which FPS kernel family a (b, n, m) call takes -- 0 plain register-resident / streaming (up to
 * 25 600 points with fewer than 256 samples or fewer than 4096 points), 1 rows in registers with one sample per round
 * (exactly 4096 points), 2 a lane per bucket with several samples per round (4097 .. 25 600 points: the per-level
 * resampling), 4 / 5 the tile form on two / three levels, 6 the tile form on *cluster workgroups per set; -1 beyond
 * every plan (more than 4 194 304 points).  The dispatch table of DESIGN section 4 as code. */
int tpu3_debug_fps_cluster(int g);
int tpu3_debug_fps_plan(int b, int n, int m, int *cluster);
/* tpu3_debug_dec_split: the lane-per-point DenseEdgeConv kernel deals a patch's 64-point steps over four waves; steps
 * left over (a 312-point patch: the fifth) are split by neighbour slots over the waves and combined by an LDS maximum
 * (on = 1, the default) or taken whole by one wave (on = 0, the form before round 4; also TPU3_DEC_SPLIT=0).  Both
 * give the same bits (tests/test_hip_network.py).  Returns the previous setting; host-side state only. */
int tpu3_debug_dec_split(int on);
/* tpu3_debug_skip_fused: the inference skip on fp32 rows runs as ONE launch with a 16-wave workgroup per patch (on = 1,
 * the default: distances and minima in LDS, a barrier, weights and update; the second read of a patch's own rows
 * comes from the Infinity Cache) or as the two kernels with a global scratch (on = 0; also TPU3_SKIP_FUSED=0).  Both
 * give the same bits (tests/test_hip_network.py).  Returns the previous setting; host-side state only. */
int tpu3_debug_skip_fused(int on);

/* The multi-workgroup FPS spins on its partner workgroups with BOUNDED polls; a launch whose workgroups never all
 * became resident gives up, leaves its outputs incomplete and counts a fault.  Returns the number of faulted
 * workgroups since the last reset on the current device (synchronises the device: call at a synchronisation point,
 * as pipeline.upsample and bench.py do).  0 in every correct run. */
long tpu3_fps_cluster_faults(int reset);
/* Test hook: on != 0 makes member 1 of every cluster of the FOLLOWING multi-workgroup FPS launches leave at once (a
 * workgroup that never became resident) and shortens its partners' patience to 4096 polls, so that the fault path
 * (fault counted, members leave together, samples not taken filled with index 0, caller recomputes on the
 * single-workgroup kernel) can be exercised; 0 restores the product behaviour. */
int tpu3_debug_fps_cluster_absent(int on);

#ifdef __cplusplus
}
#endif
#endif /* TPU3_H */
