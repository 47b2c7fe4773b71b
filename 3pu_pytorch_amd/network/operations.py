"""Point-set operators of the patch-upsampling path -- the `network.operations` call sites of the
reference (network/operations.py:12-323), same names, argument order, defaults and return
tuples, running on the gfx950 kernels of lib3pu_hip.so.

What differs from the reference, by design:
  * group_knn never materialises the (B,M,N) distance matrix and never leaves the device
    (the reference copies the points to the host for np.unique on every unique=True call,
    operations.py:194-203);
  * every operator also exists in a *ragged / batched* form (``knn_query``, ``fps``) so that
    the patch pipeline can process all outer patches of a cloud in one launch;
  * device tensors only: there is no CPU path in this package (the reference's FPS / gather
    are CUDA-only as well, sampling.cpp:20-24; its group_knn also ran on CPU).
"""
import ctypes

import torch

from .. import _lib as L
from .. import sampling


import collections
import os
import warnings
import weakref
_DEBUG_UWS = bool(os.environ.get("TPU3_DEBUG_UWS"))
# split-bf16 arithmetic (tpu3_split_bf16 / TPU3_SPLIT_BF16, with the regressor tail in csrc/mlp.hip): up_layer1's per-point
# half on the bf16 matrix pipe with three-term operands; TPU3_SPLIT_BF16_WIDE=0 keeps this layer on the fp32 kernel
_SPLIT_BF16_WIDE = os.environ.get("TPU3_SPLIT_BF16_WIDE", "1") not in ("0", "")
_WIDE_SPLIT_CACHE = {}      # id(weight tensor) -> (weak reference, (version, address, row stride, cin), image, stream, event)

# Every place where the network leaves a hand-written kernel for the generic PyTorch-ROCm formulation
# (a shape the fused kernel does not cover: k not a multiple of 16, other channel counts, r > 4 ...)
# reports here: counted, and warned about once per reason, so that a slow configuration is never
# silent.  bench.py asserts that the measured configuration produced no such event.
GENERIC_PATH_EVENTS = collections.Counter()


def note_generic_path(reason):
    if getattr(BACKEND, "name", "") != "hip-gfx950":
        return                                   # a test stand-in backend is installed
    if not GENERIC_PATH_EVENTS[reason]:
        warnings.warn("3pu_pytorch_amd: generic (unfused) PyTorch path taken: " + reason, RuntimeWarning,
                      stacklevel=3)
    GENERIC_PATH_EVENTS[reason] += 1


class HipBackend(object):
    """The kernels, behind the small interface the operators below use.  Tests swap this object
    to exercise the host-side wiring without a GPU; the product never does."""

    name = "hip-gfx950"

    COMPACT_MIN_N = 1024      # unique=True: point sets this large get a first-occurrence list
    # ... and, for 3-d points and k <= 8 (the inter-level search, fm_knn = 5), spatial tiles of it (csrc/knn_tiles.hip):
    # a wave of queries then searches the few tiles of 64 near it instead of the whole list.  (r5) From 2048 rows on:
    # since a patch's queries are taken in Morton order (a wave = a compact blob instead of a ring around the seed) the
    # pruned search also wins on the batched pipeline's previous sets -- ONE outer patch's inner patches, 3120 / 6240
    # rows, 10 - 20 tiles -- where it lost in round 3 (6 - 9 tiles per wave then).  TPU3_KNN_TILES=0 /
    # TPU3_KNN_TILES_MIN_N: tuning hooks
    knn_tiles = os.environ.get("TPU3_KNN_TILES", "1") not in ("0", "")
    KNN_TILES_MIN_N = int(os.environ.get("TPU3_KNN_TILES_MIN_N", "2048"))

    # Self kNN graphs can run optimistically: only the one-pass kernel, which raises a device-side event when a
    # query saw a second zero distance (rows may be duplicated: the exact path is then required).  OFF by
    # default: direct callers (net(x) in eval mode, pipeline.pc_prediction, Model.test_model) get the exact
    # gated form, whose result needs no check.  A driver that owns a synchronisation point opts in, reads
    # `graph_dup_events()` there and recomputes with the exact form (pipeline.upsample does; bench.py opts in
    # and asserts the count is zero).
    optimistic_graph = False

    def _events(self, dev):
        ev = getattr(self, "_event_words", None)
        if ev is None:
            ev = self._event_words = {}
        key = (dev.type, dev.index)
        if key not in ev:
            ev[key] = torch.zeros((4,), dtype=torch.int32, device=dev)
            torch.cuda.synchronize(dev)         # kernels on OTHER streams will read these words: zero them first
        return ev[key]

    def graph_dup_events(self, reset=True):
        """Number of devices on which an optimistic graph call asked for the exact path since the last reset
        (synchronises)."""
        hits = 0
        for t in getattr(self, "_event_words", {}).values():
            hits += int(t[2].item() != 0)
            if reset:
                t.zero_()
        return hits

    def fps_cluster_faults(self, reset=True):
        """Workgroups of the multi-workgroup FPS (csrc/fps_cluster.hip) that gave up waiting for their partners since
        the last reset, on the current device (synchronises it).  Always 0 unless a cluster launch could not become
        resident (more than 256 cluster workgroups in flight at once); a non-zero count means some FPS result since
        the last reset is incomplete."""
        return int(L.lib().tpu3_fps_cluster_faults(1 if reset else 0))

    def fps_cluster(self, g):
        """Workgroups per point set of the tile-form FPS for the calls that follow (-1: the default policy, 0: the
        single-workgroup kernels only); returns the previous setting.  pipeline.upsample recomputes with 0 after a
        cluster launch reported a fault."""
        return int(L.lib().tpu3_debug_fps_cluster(int(g)))

    def knn(self, k, query, points, unique, layout=None, want_dist=True, want_grouped=True, unique_cache=None):
        """query (B,M,C), points (Bp,N,C) f32 contiguous device tensors ->
        idx int64 (B,M,k), dist f32 (B,M,k) | None, grouped f32 (B,M,k,C) | None.
        layout: None or dict(n_arr=, m_arr=, pts_of=, grp=, groups=) of int32 device tensors.
        unique_cache: optional dict owned by the caller; the de-duplication state of `points`
        (first-occurrence mask, flags, candidate lists) is kept in it and reused by later calls that
        search the SAME point sets (the chunks of one Level call), instead of being rebuilt."""
        L.require_device(query, "query")
        L.require_device(points, "points")
        L.require_dtype(query, torch.float32, "query")
        L.require_dtype(points, torch.float32, "points")
        b, m, c = query.shape
        bp, n, c2 = points.shape
        if c2 != c:
            raise RuntimeError("group_knn: query/points channel mismatch (%d vs %d)" % (c, c2))
        lay_ref = None
        lay = None
        keep = []
        groups = 1
        if layout is not None:
            lay = L.KnnLayout()
            for name in ("n_arr", "m_arr", "pts_of", "grp"):
                t = layout.get(name)
                if t is not None:
                    L.require_device(t, name)
                    L.require_dtype(t, torch.int32, name)
                    keep.append(t)
                setattr(lay, name, L.ptr(t))
            groups = int(layout.get("groups", 1)) if layout.get("grp") is not None else 1
            lay.bp, lay.groups = bp, groups
            lay_ref = ctypes.byref(lay)
            if layout.get("pts_of") is None and bp != b:
                raise RuntimeError("group_knn: %d point sets for %d query sets needs pts_of" % (bp, b))
        elif bp != b:
            raise RuntimeError("group_knn: batch mismatch (%d vs %d)" % (b, bp))
        dev = query.device
        idx = torch.empty((b, m, k), dtype=torch.int64, device=dev)
        dist = torch.empty((b, m, k), dtype=torch.float32, device=dev) if want_dist else None
        grouped = torch.empty((b, m, k, c), dtype=torch.float32, device=dev) if want_grouped else None
        lib = L.lib()
        with torch.cuda.device(dev):
            s = L.stream_of(query)
            dup = uws = None
            if unique:
                n_arr_t = layout.get("n_arr") if layout is not None else None
                key = (points.data_ptr(), tuple(points.shape), 0 if n_arr_t is None else n_arr_t.data_ptr(), groups)
                st = unique_cache.get("state") if unique_cache is not None else None
                if st is not None and st["key"] == key and st["points"] is points:
                    # the first-occurrence mask and the candidate lists depend on `points` only; the
                    # per-call words of uws ([1] "optimistic pass failed", [4+g] max(D) of the call's
                    # groups) must start from the state tpu3_knn_unique_prepare_f32 left them in
                    dup, cand, cand_count = st["dup"], st["cand"], st["cand_count"]
                    uws = st["uws"].clone()
                    tiles = st.get("tiles")
                else:
                    dup = torch.empty((bp, n), dtype=torch.uint8, device=dev)
                    uws = torch.empty((4 + groups,), dtype=torch.int32, device=dev)
                    need = lib.tpu3_knn_unique_workspace_bytes(bp, n)
                    ws = torch.empty((need,), dtype=torch.uint8, device=dev) if need else None
                    L.check(lib.tpu3_knn_unique_prepare_f32(s, b, m, n, c, L.ptr(query), L.ptr(points),
                                                            lay_ref, L.ptr(dup), L.ptr(uws), L.ptr(ws), need),
                            "tpu3_knn_unique_prepare_f32")
                    cand = cand_count = None
                    if n >= self.COMPACT_MIN_N and k <= 64 and c <= 32:
                        # list of first occurrences: the search then skips the duplicated rows entirely
                        cand = torch.empty((bp, n), dtype=torch.int32, device=dev)
                        cand_count = torch.empty((bp,), dtype=torch.int32, device=dev)
                        L.check(lib.tpu3_knn_unique_compact_i32(s, bp, n, L.ptr(n_arr_t), L.ptr(dup), L.ptr(uws),
                                                                L.ptr(cand), L.ptr(cand_count)),
                                "tpu3_knn_unique_compact_i32")
                    tiles = None
                    if cand is not None and c == 3 and k <= 8 and self.knn_tiles and n >= self.KNN_TILES_MIN_N:
                        # small k in a large 3-d set (the inter-level search): the candidates in Morton-ordered tiles
                        # of 64 with their boxes -- a wave of queries then searches only the tiles near it
                        nt = (n + 63) // 64
                        tiles = (torch.empty((bp, nt * 64, 4), dtype=torch.float32, device=dev),
                                 torch.empty((bp, nt * 64), dtype=torch.int32, device=dev),
                                 torch.empty((bp, nt, 8), dtype=torch.float32, device=dev))
                        tneed = lib.tpu3_knn_tiles_workspace_bytes(bp, n)
                        tws = torch.empty((tneed,), dtype=torch.uint8, device=dev)
                        L.check(lib.tpu3_knn_tiles_build_f32(s, bp, n, L.ptr(points), L.ptr(n_arr_t), L.ptr(cand),
                                                             L.ptr(cand_count), L.ptr(uws), L.ptr(tiles[0]),
                                                             L.ptr(tiles[1]), L.ptr(tiles[2]), L.ptr(tws), tneed),
                                "tpu3_knn_tiles_build_f32")
                    if unique_cache is not None:
                        # `points` itself is kept: the key holds its address, which must not be recycled
                        unique_cache["state"] = dict(key=key, points=points, dup=dup, uws=uws.clone(), cand=cand,
                                                     cand_count=cand_count, tiles=tiles)
                if cand is not None:
                    if lay is None:
                        lay = L.KnnLayout()
                        lay.bp, lay.groups = bp, 1
                        lay_ref = ctypes.byref(lay)
                    lay.cand, lay.cand_count = L.ptr(cand), L.ptr(cand_count)
                    if tiles is not None and c == 3 and k <= 8:
                        lay.tile_pts, lay.tile_idx, lay.tile_box = L.ptr(tiles[0]), L.ptr(tiles[1]), L.ptr(tiles[2])
            L.check(lib.tpu3_knn_f32(s, b, m, n, c, k, L.ptr(query), L.ptr(points), lay_ref, L.ptr(dup),
                                     L.ptr(uws), L.ptr(idx), 8, L.ptr(dist), L.ptr(grouped)),
                    "tpu3_knn_f32")
        if unique and _DEBUG_UWS:
            torch.cuda.synchronize()
            print("[uws] b=%d m=%d n=%d c=%d k=%d any_dup=%d redo=%d" % (b, m, n, c, k, int(uws[0]), int(uws[1])))
        return idx, dist, grouped

    def _layout(self, layout, b, bp):
        if layout is None:
            return None, 1, []
        lay = L.KnnLayout()
        keep = []
        for name in ("n_arr", "m_arr", "pts_of", "grp"):
            t = layout.get(name)
            if t is not None:
                L.require_device(t, name)
                L.require_dtype(t, torch.int32, name)
                keep.append(t)
            setattr(lay, name, L.ptr(t))
        groups = int(layout.get("groups", 1)) if layout.get("grp") is not None else 1
        lay.bp, lay.groups = bp, groups
        return lay, groups, keep

    def knn_graph(self, k, x, layout=None, optimistic=None):
        """Self kNN graph for the fused DenseEdgeConv: x (B,N,C) f32 -> idx int32 (B,N,k) holding the
        exact top-k set (unique=True semantics), nearest in slot 0, the rest in no particular order.
        Returns None when the size is not covered by the two-pass kernel.
        optimistic (default: self.optimistic_graph): only the two-pass kernel runs and possible duplicated
        rows are reported through graph_dup_events() instead of being handled by gated fallback launches."""
        b, n, c = x.shape
        if c > 32 or k not in (17, 33) or n < k:
            return None
        lay, groups, keep = self._layout(layout, b, b)
        lay_ref = ctypes.byref(lay) if lay is not None else None
        dev = x.device
        idx = torch.empty((b, n, k), dtype=torch.int32, device=dev)
        dense = layout is None or all(layout.get(nm) is None for nm in ("n_arr", "m_arr", "pts_of"))
        if dense and (self.optimistic_graph if optimistic is None else optimistic):
            with torch.cuda.device(dev):
                L.check(L.lib().tpu3_knn_graph_self_optimistic_f32(L.stream_of(x), b, n, c, k, L.ptr(x), lay_ref,
                                                                   L.ptr(self._events(dev)), L.ptr(idx)),
                        "tpu3_knn_graph_self_optimistic_f32")
            return idx
        dup = torch.empty((b, n), dtype=torch.uint8, device=dev)
        uws = torch.empty((4 + groups,), dtype=torch.int32, device=dev)
        lib = L.lib()
        with torch.cuda.device(dev):
            s = L.stream_of(x)
            need = lib.tpu3_knn_unique_workspace_bytes(b, n)
            ws = torch.empty((need,), dtype=torch.uint8, device=dev) if need else None
            if need and dense:
                # no de-duplication pre-pass: the kernel notices by itself whether one is needed
                L.check(lib.tpu3_knn_graph_self_f32(s, b, n, c, k, L.ptr(x), lay_ref, L.ptr(dup), L.ptr(uws),
                                                    L.ptr(idx), L.ptr(ws), need), "tpu3_knn_graph_self_f32")
                return idx
            L.check(lib.tpu3_knn_unique_prepare_f32(s, b, n, n, c, L.ptr(x), L.ptr(x), lay_ref, L.ptr(dup),
                                                    L.ptr(uws), L.ptr(ws), need), "tpu3_knn_unique_prepare_f32")
            L.check(lib.tpu3_knn_graph_f32(s, b, n, n, c, k, L.ptr(x), L.ptr(x), lay_ref, L.ptr(dup), L.ptr(uws),
                                           L.ptr(idx)), "tpu3_knn_graph_f32")
        return idx

    def fps(self, xyz, npoint, n_arr=None, m_arr=None):
        """xyz (B,N,3) f32 contiguous -> idx int32 (B,npoint).  Dense calls go through the
        drop-in `sampling.furthest_sampling` entry point, ragged ones through the C ABI."""
        b, n, _ = xyz.shape
        idx = torch.empty((b, npoint), dtype=torch.int32, device=xyz.device)
        lib = L.lib() if xyz.is_cuda else None
        need = lib.tpu3_fps_workspace_bytes(b, n) if lib is not None else 0
        dense = n_arr is None and m_arr is None and need == 0
        if lib is not None:
            # temp = 1e10 (the reference's protocol, operations.py:289) and, for ragged calls, idx = 0: ONE launch
            temp = torch.empty((b, n), dtype=torch.float32, device=xyz.device)
            with torch.cuda.device(xyz.device):
                L.check(lib.tpu3_fill_f32_i32(L.stream_of(xyz), L.ptr(temp), temp.numel(), 1e10,
                                              None if dense else L.ptr(idx), 0 if dense else idx.numel(), 0),
                        "tpu3_fill_f32_i32")
        else:
            temp = torch.full((b, n), 1e10, dtype=torch.float32, device=xyz.device)
            idx.zero_()
        if dense:
            sampling.furthest_sampling(b, n, npoint, xyz, temp, idx)
            return idx
        L.require_device(xyz, "xyz")
        L.require_dtype(xyz, torch.float32, "xyz")
        for t, nm in ((n_arr, "n_arr"), (m_arr, "m_arr")):
            if t is not None:
                L.require_device(t, nm)
                L.require_dtype(t, torch.int32, nm)
        # large point sets: scratch for the bucketed kernel comes from torch's caching allocator
        ws = torch.empty((need,), dtype=torch.uint8, device=xyz.device) if need else None
        with torch.cuda.device(xyz.device):
            L.check(L.lib().tpu3_fps_ragged_f32(L.stream_of(xyz), b, n, npoint, L.ptr(n_arr), L.ptr(m_arr),
                                                L.ptr(xyz), L.ptr(temp), L.ptr(idx), L.ptr(ws), need),
                    "tpu3_fps_ragged_f32")
        return idx

    def gather_forward(self, features, idx):
        b, c, n = features.shape
        npoint = idx.shape[1]
        out = torch.empty((b, c, npoint), dtype=features.dtype, device=features.device)
        return sampling.gather_forward(b, c, n, npoint, features, idx, out)

    def gather_backward(self, grad_out, idx, c, n):
        b, npoint = idx.shape
        grad = torch.zeros((b, c, n), dtype=grad_out.dtype, device=grad_out.device)
        return sampling.gather_backward(b, c, n, npoint, grad_out, idx, grad)

    def dense_edge_conv_pack(self, mlps, fold_w=None):
        """The block's operand tables (and those of the folded prep convolutions) written out once,
        tpu3_dense_edge_conv_pack_f32: a float32 blob the *_pk_* launches copy instead of building the tables in
        every workgroup.  The caller keeps it for as long as the weights are unchanged."""
        w = []
        for conv in mlps:
            w.append(conv.weight.detach().reshape(conv.weight.size(0), -1).contiguous())
            w.append(conv.bias.detach().contiguous())
        for t in w:
            L.require_device(t, "weights")
            L.require_dtype(t, torch.float32, "weights")
        fold_n = 0 if fold_w is None else fold_w.size(0)
        if fold_w is not None:
            fold_w = fold_w.contiguous()
        nf = L.lib().tpu3_dense_edge_conv_pack_floats(fold_n)
        if nf == 0:
            raise RuntimeError("dense_edge_conv_pack: fold_n must be 0, 24, 48 or 72")
        blob = torch.empty((nf,), device=w[0].device, dtype=torch.float32)
        with torch.cuda.device(w[0].device):
            L.check(L.lib().tpu3_dense_edge_conv_pack_f32(
                L.stream_of(w[0]), L.ptr(w[0]), L.ptr(w[1]), L.ptr(w[2]), L.ptr(w[3]), L.ptr(w[4]), L.ptr(w[5]),
                fold_n, L.ptr(fold_w), L.ptr(blob)), "tpu3_dense_edge_conv_pack_f32")
        return blob

    def dense_edge_conv(self, x, idx, idx_off, k, mlps, out, mfma=L.MFMA_F32, pack=None):
        """Fused DenseEdgeConv (inference): x (P,N,24) contiguous, idx (P,N,idx_stride) int64/int32,
        the k neighbours start at column idx_off; mlps = the block's three nn.Conv2d; `out` is a
        (P,N,>=60) view with unit channel stride whose channels [0,60) receive y.
        mfma: L.MFMA_F32 (exact fp32 chains) or L.MFMA_F16 (fp16 operands, fp32 accumulate)."""
        L.require_device(x, "x")
        L.require_dtype(x, torch.float32, "x")
        L.require_device(idx, "idx")
        P, N, _ = x.shape
        if out.stride(2) != 1 or out.stride(0) != N * out.stride(1):
            raise RuntimeError("dense_edge_conv: out must be a channel-slice of a contiguous (P,N,C) tensor")
        w = []
        for conv in mlps:
            w.append(conv.weight.detach().reshape(conv.weight.size(0), -1).contiguous())
            w.append(conv.bias.detach().contiguous())
        with torch.cuda.device(x.device):
            if out.dtype == torch.float16:      # the level's feature buffer stored as fp16 (activation storage "f16")
                L.check(L.lib().tpu3_dense_edge_conv_st_f32(
                    L.stream_of(x), P, N, k, L.ptr(x), L.ptr(idx), idx.element_size(), idx.size(2), idx_off,
                    L.ptr(w[0]), L.ptr(w[1]), L.ptr(w[2]), L.ptr(w[3]), L.ptr(w[4]), L.ptr(w[5]),
                    L.ptr(out), out.stride(1), int(mfma), L.STORE_F16), "tpu3_dense_edge_conv_st_f32")
                return out
            L.require_dtype(out, torch.float32, "out")
            if pack is not None and mfma == L.MFMA_F32:
                # packed operands (dense_edge_conv_pack): same launch, same bits, no table set-up per workgroup
                rc = L.lib().tpu3_dense_edge_conv_pk_f32(
                    L.stream_of(x), P, N, k, L.ptr(x), L.ptr(idx), idx.element_size(), idx.size(2), idx_off,
                    L.ptr(pack), L.ptr(out), out.stride(1))
                if rc != L.ELIMIT:                  # (patches beyond the lane-per-point kernel take the weights)
                    L.check(rc, "tpu3_dense_edge_conv_pk_f32")
                    return out
            L.check(L.lib().tpu3_dense_edge_conv_f32(
                L.stream_of(x), P, N, k, L.ptr(x), L.ptr(idx), idx.element_size(), idx.size(2), idx_off,
                L.ptr(w[0]), L.ptr(w[1]), L.ptr(w[2]), L.ptr(w[3]), L.ptr(w[4]), L.ptr(w[5]),
                L.ptr(out), out.stride(1), int(mfma)), "tpu3_dense_edge_conv_f32")
        return out

    def dense_edge_conv_fold(self, x, idx, idx_off, k, mlps, out, fold_w, fold_b, acc, seed_off, store_off, xnext,
                             pack=None):
        """dense_edge_conv (fp32) + the next prep convolutions folded into the write-out, see
        tpu3_dense_edge_conv_fold_f32: fold_w (fold_n, 60), fold_b (fold_n) or None (sums continue from
        acc[..., seed_off:]), acc (P,N,S) or None, xnext (P,N,24) receives the next block's input rows.
        Returns False when the patch size is beyond the kernel (the caller then runs the layers unfolded)."""
        L.require_device(x, "x")
        L.require_dtype(x, torch.float32, "x")
        P, N, _ = x.shape
        if out.stride(2) != 1 or out.stride(0) != N * out.stride(1):
            raise RuntimeError("dense_edge_conv: out must be a channel-slice of a contiguous (P,N,C) tensor")
        w = []
        for conv in mlps:
            w.append(conv.weight.detach().reshape(conv.weight.size(0), -1).contiguous())
            w.append(conv.bias.detach().contiguous())
        if pack is not None:        # packed operands of the block AND of fold_w (dense_edge_conv_pack(mlps, fold_w))
            with torch.cuda.device(x.device):
                rc = L.lib().tpu3_dense_edge_conv_fold_pk_f32(
                    L.stream_of(x), P, N, k, L.ptr(x), L.ptr(idx), idx.element_size(), idx.size(2), idx_off,
                    L.ptr(pack), L.ptr(out), out.stride(1), fold_w.size(0), L.ptr(fold_b), L.ptr(acc),
                    0 if acc is None else acc.stride(1), seed_off, store_off, L.ptr(xnext))
            if rc == L.ELIMIT:
                return False
            L.check(rc, "tpu3_dense_edge_conv_fold_pk_f32")
            return True
        fold_w = fold_w.contiguous()
        with torch.cuda.device(x.device):
            rc = L.lib().tpu3_dense_edge_conv_fold_f32(
                L.stream_of(x), P, N, k, L.ptr(x), L.ptr(idx), idx.element_size(), idx.size(2), idx_off,
                L.ptr(w[0]), L.ptr(w[1]), L.ptr(w[2]), L.ptr(w[3]), L.ptr(w[4]), L.ptr(w[5]),
                L.ptr(out), out.stride(1), fold_w.size(0), L.ptr(fold_w), L.ptr(fold_b), L.ptr(acc),
                0 if acc is None else acc.stride(1), seed_off, store_off, L.ptr(xnext))
        if rc == L.ELIMIT:
            return False
        L.check(rc, "tpu3_dense_edge_conv_fold_f32")
        return True

    def interlevel_skip(self, xyz, feat, prev_xyz, prev_feat, pts_of, idx, scale=0.2, per_cloud=0):
        """Fused skip connection (inference): feat (B,N,C) is updated in place
        (x_i += scale * sum_k w_k f_k with the reference's bilateral weights).  per_cloud: patches
        [i*per_cloud, (i+1)*per_cloud) share a previous cloud (scheduling hint, 0 = unknown)."""
        half = feat.dtype == torch.float16        # feature buffers stored as fp16 (activation storage "f16")
        for t, nm in ((xyz, "xyz"), (feat, "feat"), (prev_xyz, "prev_xyz"), (prev_feat, "prev_feat")):
            L.require_device(t, nm)
            L.require_dtype(t, torch.float16 if half and nm.endswith("feat") else torch.float32, nm)
        L.require_device(idx, "idx")
        B, N, C = feat.shape
        K = idx.size(2)
        need = L.lib().tpu3_interlevel_skip_workspace_bytes(B, N, K)
        ws = torch.empty((need,), dtype=torch.uint8, device=feat.device)
        with torch.cuda.device(feat.device):
            if half:
                L.check(L.lib().tpu3_interlevel_skip_st_f32(
                    L.stream_of(feat), B, N, K, C, L.ptr(xyz), L.ptr(feat), feat.stride(1), L.ptr(prev_xyz),
                    L.ptr(prev_feat), prev_xyz.size(1), L.ptr(pts_of), L.ptr(idx), idx.element_size(), float(scale),
                    int(per_cloud), L.ptr(ws), need, L.STORE_F16), "tpu3_interlevel_skip_st_f32")
                return feat
            L.check(L.lib().tpu3_interlevel_skip_f32(
                L.stream_of(feat), B, N, K, C, L.ptr(xyz), L.ptr(feat), feat.stride(1), L.ptr(prev_xyz),
                L.ptr(prev_feat), prev_xyz.size(1), L.ptr(pts_of), L.ptr(idx), idx.element_size(), float(scale),
                int(per_cloud), L.ptr(ws), need),
                "tpu3_interlevel_skip_f32")
        return feat

    def linear_dgrad(self, dy, weight):
        """dy (M, C_out) rows with unit channel stride, weight (C_out, C_in) contiguous -> dx (M, C_in) = dy W, or
        None when the shape is not covered (C_out <= 32, C_in <= 320): tpu3_linear_dgrad_f32."""
        m, cout = dy.shape
        cin = weight.size(1)
        if cout > 32 or cin > 320 or dy.stride(1) != 1 or dy.dtype != torch.float32 or not weight.is_contiguous():
            return None
        dx = torch.empty((m, cin), dtype=torch.float32, device=dy.device)
        with torch.cuda.device(dy.device):
            L.check(L.lib().tpu3_linear_dgrad_f32(L.stream_of(dy), m, cin, cout, L.ptr(dy), dy.stride(0), L.ptr(weight),
                                                  L.ptr(dx), cin), "tpu3_linear_dgrad_f32")
        return dx

    def interlevel_skip_train(self, xyz, feat, prev_xyz, prev_feat, pts_of, idx, scale=0.2):
        """Training forward of the skip connection: feat (B,N,C) updated in place, -> weights (B,N,K), the
        normalised bilateral weights the backward needs (tpu3_interlevel_skip_train_f32)."""
        B, N, C = feat.shape
        K = idx.size(2)
        need = L.lib().tpu3_interlevel_skip_workspace_bytes(B, N, K)
        ws = torch.empty((need,), dtype=torch.uint8, device=feat.device)
        weights = torch.empty((B, N, K), dtype=torch.float32, device=feat.device)
        with torch.cuda.device(feat.device):
            L.check(L.lib().tpu3_interlevel_skip_train_f32(
                L.stream_of(feat), B, N, K, C, L.ptr(xyz), L.ptr(feat), feat.stride(1), L.ptr(prev_xyz),
                L.ptr(prev_feat), prev_xyz.size(1), L.ptr(pts_of), L.ptr(idx), idx.element_size(), float(scale),
                L.ptr(weights), L.ptr(ws), need), "tpu3_interlevel_skip_train_f32")
        return weights

    def interlevel_skip_backward(self, g, weights, idx, pts_of, prev_shape, scale=0.2):
        """g (B,N,C) -> gradient of the previous level's features (Bp,M,C): scatter of scale * w_k * g_i
        (tpu3_interlevel_skip_bwd_f32; float atomics)."""
        B, N, C = g.shape
        gprev = torch.zeros(prev_shape, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            L.check(L.lib().tpu3_interlevel_skip_bwd_f32(
                L.stream_of(g), B, N, idx.size(2), C, L.ptr(g), L.ptr(weights), prev_shape[1], L.ptr(pts_of),
                L.ptr(idx), idx.element_size(), float(scale), L.ptr(gprev)), "tpu3_interlevel_skip_bwd_f32")
        return gprev

    def linear_small(self, x, weight, bias, relu, mfma=L.MFMA_F32):
        """Per-point linear layer with <= 32 outputs (fp16 operands: <= 128) on channel-last rows: x (..., C_in)
        with unit channel stride and ONE row stride (a channel slice of a contiguous buffer is fine),
        weight (C_out, C_in) -> (..., C_out), or None when the shape is not covered."""
        cin, cout = x.size(-1), weight.size(0)
        half = x.dtype == torch.float16           # rows of a feature buffer stored as fp16: fp16-operand kernels only
        if (cout > (128 if int(mfma) == int(L.MFMA_F16) else 32) or cin > 320 or cin % 4 or cout % 4
                or x.stride(-1) != 1 or x.dtype not in (torch.float32, torch.float16)
                or (half and int(mfma) != int(L.MFMA_F16))):
            return None
        rs = x.stride(-2)
        lead = x.shape[:-1]
        m = 1
        for d in lead:
            m *= d
        # all leading dims must collapse onto the row stride
        exp = rs
        for d, st in zip(reversed(lead), reversed(x.stride()[:-1])):
            if d != 1 and st != exp:
                return None
            exp *= d
        if rs % 4 or (x.data_ptr() & (7 if half else 15)) or (weight.data_ptr() & 15):
            return None
        w = weight.contiguous()
        y = torch.empty(lead + (cout,), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            if half:
                L.check(L.lib().tpu3_linear_small_st_f32(L.stream_of(x), m, cin, cout, L.ptr(x), rs, L.ptr(w),
                                                         L.ptr(bias), 1 if relu else 0, L.ptr(y), cout, int(mfma),
                                                         L.STORE_F16), "tpu3_linear_small_st_f32")
                return y
            L.check(L.lib().tpu3_linear_small_f32(L.stream_of(x), m, cin, cout, L.ptr(x), rs, L.ptr(w),
                                                  L.ptr(bias), 1 if relu else 0, L.ptr(y), cout, int(mfma)),
                    "tpu3_linear_small_f32")
        return y

    def dec_train_forward(self, x, idx, idx_off, weights):
        """Training forward of the (24, 12, 3, k = 32) DenseEdgeConv block: x (P,N,24), idx (P,N,S) int32 with the
        neighbours at [idx_off, idx_off + 32), weights = (w0, b0, w1, b1, w2, b2) -> y (P,N,60), arg (P,N,36) u8."""
        P, N, _ = x.shape
        y = torch.empty((P, N, 60), dtype=torch.float32, device=x.device)
        arg = torch.empty((P, N, 36), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            L.check(L.lib().tpu3_dec_train_fwd_f32(L.stream_of(x), P, N, 32, L.ptr(x), L.ptr(idx), idx.size(2), idx_off,
                                                   *[L.ptr(w) for w in weights], L.ptr(y), L.ptr(arg)),
                    "tpu3_dec_train_fwd_f32")
        return y, arg

    def dec_train_backward(self, x, idx, idx_off, weights, arg, gy):
        """-> gx (P,N,24), S (P*N, 36) and the workspace that holds the workgroups' weight-gradient blocks of the edge
        parts (for dec_train_wgrad); see tpu3_dec_train_bwd_f32."""
        P, N, _ = x.shape
        dev = x.device
        lib = L.lib()
        # gy may be a channel slice of the concatenated features' gradient: rows gy.stride(1) floats apart, no copy
        if not (gy.stride(2) == 1 and gy.stride(0) == N * gy.stride(1) and gy.stride(1) >= 60):
            gy = gy.contiguous()
        gx = torch.zeros((P, N, 24), dtype=torch.float32, device=dev)
        S = torch.empty((P * N, 36), dtype=torch.float32, device=dev)
        need = lib.tpu3_dec_train_wgrad_workspace_bytes(P * N)
        ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            L.check(lib.tpu3_dec_train_bwd_f32(L.stream_of(x), P, N, 32, L.ptr(x), L.ptr(idx), idx.size(2), idx_off,
                                               *[L.ptr(w) for w in weights], L.ptr(arg), L.ptr(gy), gy.stride(1),
                                               L.ptr(gx), L.ptr(S), L.ptr(ws), need), "tpu3_dec_train_bwd_f32")
        return gx, S, ws

    def dec_train_wgrad(self, x, S, ws):
        """The block's weight gradients (12,48), (12,36), (12,48) and bias gradients (36) = [b2 | b1 | b0] from what the
        backward kernel left behind: tpu3_dec_train_wgrad_f32 (two launches)."""
        points = S.size(0)
        dev = x.device
        lib = L.lib()
        gw0 = torch.empty((12, 48), dtype=torch.float32, device=dev)
        gw1 = torch.empty((12, 36), dtype=torch.float32, device=dev)
        gw2 = torch.empty((12, 48), dtype=torch.float32, device=dev)
        gb = torch.empty((36,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.check(lib.tpu3_dec_train_wgrad_f32(L.stream_of(x), points, L.ptr(x), L.ptr(S), L.ptr(gw0), L.ptr(gw1),
                                                 L.ptr(gw2), L.ptr(gb), L.ptr(ws), ws.numel()),
                    "tpu3_dec_train_wgrad_f32")
        return gw0, gw1, gw2, gb

    def gather_rows(self, x, idx):
        """x (B,N,C) f32 contiguous, idx (B,...) int32 / int64 -> (B,...,C) rows, or None when not covered."""
        B, N, C = x.shape
        if (C % 4 or x.dtype != torch.float32 or not x.is_contiguous() or not idx.is_contiguous()
                or idx.dtype not in (torch.int32, torch.int64) or idx.size(0) != B or (x.data_ptr() & 15)):
            return None
        m = idx.numel() // max(1, B)
        out = torch.empty(tuple(idx.shape) + (C,), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            L.check(L.lib().tpu3_gather_rows_f32(L.stream_of(x), B, N, m, C, L.ptr(x), L.ptr(idx), idx.element_size(),
                                                 L.ptr(out)), "tpu3_gather_rows_f32")
        return out

    def scatter_add_rows(self, g, idx, n):
        """Transpose of gather_rows: g (B,...,C) f32, idx (B,...) -> dx (B,n,C) with dx[b, idx[b,j]] += g[b,j]."""
        B, C = g.size(0), g.size(-1)
        g = g.contiguous()
        idx = idx.contiguous()
        m = idx.numel() // max(1, B)
        dx = torch.zeros((B, n, C), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            L.check(L.lib().tpu3_scatter_add_rows_f32(L.stream_of(g), B, n, m, C, L.ptr(g), L.ptr(idx),
                                                      idx.element_size(), L.ptr(dx)), "tpu3_scatter_add_rows_f32")
        return dx

    def linear_wide(self, x, weight, bias):
        """Per-point linear layer with 128 outputs (the per-point half of up_layer1): x (..., C_in) contiguous,
        weight (128, C_in) with unit column stride (a column slice of a wider matrix is fine) -> (..., 128),
        or None when the shape is not covered (the caller then uses the library GEMM)."""
        cin, cout = x.size(-1), weight.size(0)
        if (cout != 128 or cin % 4 or not 256 < cin <= 272 or x.dtype != torch.float32 or not x.is_contiguous()
                or weight.stride(1) != 1 or weight.dtype != torch.float32 or (x.data_ptr() & 15)):
            return None
        if bias is not None and (not bias.is_contiguous() or (bias.data_ptr() & 15)):
            return None
        m = x.numel() // cin
        y = torch.empty(x.shape[:-1] + (cout,), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            if _SPLIT_BF16_WIDE and cin % 8 == 0 and not (x.data_ptr() & 31) and L.lib().tpu3_split_bf16(-1):
                # fp32 operands as three bf16 terms (csrc/mlp.hip, linear_wide_sb_kernel)
                ws = self._wide_split(weight, cin)
                L.check(L.lib().tpu3_linear_wide_sb_f32(L.stream_of(x), m, cin, cout, L.ptr(x), cin, L.ptr(ws),
                                                        L.ptr(bias), L.ptr(y), cout), "tpu3_linear_wide_sb_f32")
                return y
            L.check(L.lib().tpu3_linear_wide_f32(L.stream_of(x), m, cin, cout, L.ptr(x), cin, L.ptr(weight),
                                                 weight.stride(0), L.ptr(bias), L.ptr(y), cout),
                    "tpu3_linear_wide_f32")
        return y

    def split_bf16(self, on=None):
        """Arithmetic of the regressor's matrix layers: True = three-term bf16 operands on the bf16 matrix pipe, False =
        fp32 matrix instructions; None = query.  Returns the previous setting (initially TPU3_SPLIT_BF16)."""
        return bool(L.lib().tpu3_split_bf16(-1 if on is None else int(bool(on))))

    def _wide_split(self, weight, cin):
        """The slab-major three-term bf16 image of weight[:, :cin] (tpu3_linear_wide_split_bf16), cached per WEIGHT
        TENSOR: the entry lives as long as the tensor the view comes from (a weak reference -- an address alone is
        reused by the allocator) and is rebuilt when its version counter, address or layout changed; built on one
        stream, other streams order themselves behind the build (as the fold plans and Level._code_term do)."""
        base = weight._base if weight._base is not None else weight
        key = (base._version, weight.data_ptr(), weight.stride(0), cin)
        here = torch.cuda.current_stream(weight.device)
        hit = _WIDE_SPLIT_CACHE.get(id(base))
        if hit is not None and hit[0]() is base and hit[1] == key:
            if hit[3] != here.cuda_stream:
                here.wait_event(hit[4])
                hit[2].record_stream(here)
            return hit[2]
        ws = torch.empty((int(L.lib().tpu3_linear_wide_split_bytes(cin)),), dtype=torch.uint8, device=weight.device)
        L.check(L.lib().tpu3_linear_wide_split_bf16(here.cuda_stream, cin, weight.size(0), L.ptr(weight), weight.stride(0),
                                                    L.ptr(ws)), "tpu3_linear_wide_split_bf16")
        done = torch.cuda.Event()
        done.record(here)
        ident = id(base)
        _WIDE_SPLIT_CACHE[ident] = (weakref.ref(base, lambda _r, i=ident: _WIDE_SPLIT_CACHE.pop(i, None)), key, ws,
                                    here.cuda_stream, done)
        return ws

    def invalidate_split_weights(self):
        """Drop the cached split images (after an edit of a weight the version counter cannot see: `param.data`)."""
        _WIDE_SPLIT_CACHE.clear()

    def linear_lift(self, x, weight, bias, relu, also=None):
        """Per-point linear layer with <= 8 input channels (the 3 -> 24 lift of a Level): x (..., C_in)
        contiguous rows, weight (C_out, C_in) -> (..., C_out) contiguous; `also`: a (..., C_out) view with unit
        channel stride and one row stride (a slice of the level's feature buffer) that receives the same rows.
        None when the shape is not covered."""
        cin, cout = x.size(-1), weight.size(0)
        if cin > 8 or cout > 64 or cout % 4 or x.dtype != torch.float32 or not x.is_contiguous():
            return None
        lead = x.shape[:-1]
        m = x.numel() // cin
        y2, y2s = None, 0
        if also is not None:
            if (tuple(also.shape) != tuple(lead) + (cout,) or also.stride(-1) != 1 or also.dtype != torch.float32
                    or also.stride(-2) % 4 or (also.data_ptr() & 15)):
                return None
            exp = also.stride(-2)
            for d, st in zip(reversed(lead), reversed(also.stride()[:-1])):
                if d != 1 and st != exp:
                    return None
                exp *= d
            y2, y2s = also, also.stride(-2)
        w = weight.contiguous()
        y = torch.empty(lead + (cout,), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            L.check(L.lib().tpu3_linear_lift_f32(L.stream_of(x), m, cin, cout, L.ptr(x), cin, L.ptr(w), L.ptr(bias),
                                                 1 if relu else 0, L.ptr(y), cout, L.ptr(y2), y2s),
                    "tpu3_linear_lift_f32")
        return y

    def linear_wgrad(self, x, dy):
        """x (M, C_in), dy (M, C_out) f32 rows with unit channel stride -> dW (C_out, C_in) =
        dy^T x, or None when the shape is not covered (C_out <= 16, C_in <= 64)."""
        m, cin = x.shape
        cout = dy.size(1)
        if cout > 16 or cin > 64 or x.stride(1) != 1 or dy.stride(1) != 1 or x.dtype != torch.float32:
            return None
        lib = L.lib()
        need = lib.tpu3_linear_wgrad_workspace_bytes(m)
        ws = torch.empty((need,), dtype=torch.uint8, device=x.device)
        dw = torch.empty((cout, cin), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            L.check(lib.tpu3_linear_wgrad_f32(L.stream_of(x), m, cin, cout, L.ptr(x), x.stride(0), L.ptr(dy),
                                              dy.stride(0), L.ptr(dw), L.ptr(ws), need), "tpu3_linear_wgrad_f32")
        return dw

    def linear_wgrad_bias(self, x, dy, want_bias=True):
        """x (M, C_in), dy (M, C_out) f32 rows with unit channel stride -> (dW (C_out, C_in) = dy^T x,
        db (C_out) = column sums of dy | None): tpu3_linear_wgrad_bias_f32, any layer width of a Level."""
        m, cin = x.shape
        cout = dy.size(1)
        if cin > 1023 or cout > 1024 or x.stride(1) != 1 or dy.stride(1) != 1 or x.dtype != torch.float32:
            return None
        lib = L.lib()
        need = lib.tpu3_linear_wgrad_bias_workspace_bytes(m, cin, cout)
        ws = torch.empty((need,), dtype=torch.uint8, device=x.device)
        dw = torch.empty((cout, cin), dtype=torch.float32, device=x.device)
        db = torch.empty((cout,), dtype=torch.float32, device=x.device) if want_bias else None
        with torch.cuda.device(x.device):
            L.check(lib.tpu3_linear_wgrad_bias_f32(L.stream_of(x), m, cin, cout, L.ptr(x), x.stride(0), L.ptr(dy),
                                                   dy.stride(0), L.ptr(dw), L.ptr(db), L.ptr(ws), need),
                    "tpu3_linear_wgrad_bias_f32")
        return dw, db

    def regress_tail(self, a, c, w2, b2, w3, b3, w4, b4, residual, mfma=L.MFMA_F32):
        """a (M,128), c (r,128), residual (M,3) -> (M*r, 3); see tpu3_regress_tail_f32."""
        m, r = a.size(0), c.size(0)
        out = torch.empty((m * r, 3), dtype=torch.float32, device=a.device)
        ts = [t.contiguous() for t in (a, c, w2, b2, w3, b3, w4, b4, residual)]
        with torch.cuda.device(a.device):
            L.check(L.lib().tpu3_regress_tail_f32(L.stream_of(a), m, r, *[L.ptr(t) for t in ts], L.ptr(out), int(mfma)),
                    "tpu3_regress_tail_f32")
        return out

    def ball_query(self, query, xyz, radius, nsample):
        """query (B,M,3), xyz (B,N,3) contiguous -> idx int32 (B,M,nsample) through the drop-in
        `sampling.ball_query` entry point."""
        return sampling.ball_query(query, xyz, radius, nsample)

    def normalize(self, pc, n_arr=None):
        """pc (B,3,N) f32 contiguous -> (out (B,3,N), centroid (B,3,1), radius (B,1,1))."""
        L.require_device(pc, "pc")
        L.require_dtype(pc, torch.float32, "pc")
        b, _, n = pc.shape
        out = torch.empty_like(pc)
        centroid = torch.empty((b, 3, 1), dtype=torch.float32, device=pc.device)
        radius = torch.empty((b, 1, 1), dtype=torch.float32, device=pc.device)
        with torch.cuda.device(pc.device):
            L.check(L.lib().tpu3_normalize_f32(L.stream_of(pc), b, n, L.ptr(n_arr), L.ptr(pc), L.ptr(out),
                                               L.ptr(centroid), L.ptr(radius)), "tpu3_normalize_f32")
        return out, centroid, radius


    # ---- (r6) the small steps between the eval path's kernels, one launch each (csrc/glue.hip) --------------------
    def normalize_cl(self, pc, n_arr=None):
        """pc (B,N,3) f32 contiguous channel-last -> (out (B,N,3), centroid (B,3), radius (B)): normalize() without
        the two transposes around it; same operations in the same order."""
        L.require_device(pc, "pc")
        L.require_dtype(pc, torch.float32, "pc")
        b, n, _ = pc.shape
        out = torch.empty_like(pc)
        centroid = torch.empty((b, 3), dtype=torch.float32, device=pc.device)
        radius = torch.empty((b,), dtype=torch.float32, device=pc.device)
        with torch.cuda.device(pc.device):
            L.check(L.lib().tpu3_normalize_cl_f32(L.stream_of(pc), b, n, L.ptr(n_arr), L.ptr(pc), L.ptr(out),
                                                  L.ptr(centroid), L.ptr(radius)), "tpu3_normalize_cl_f32")
        return out, centroid, radius

    def repatch_filter(self, closest, xyz, k, r, small_cell=None):
        """The outlier filter of the eval-mode patch extraction (reference upsampler.py:63-77) in one launch.
        closest (B,N,2) f32: kNN(k=2) distances (column 1 = the closest OTHER point); xyz (B,N,3).
        -> xyz_f (B,N,3) kept points first, count (B,) int32, patch_num, patch_num*k, patch_num*k*r (B,) int32;
        small_cell: optional 0-d int64 device tensor that counts clouds with fewer kept points than k."""
        B, N, _ = xyz.shape
        dev = xyz.device
        xyz_f = torch.empty_like(xyz)
        out = torch.empty((4, B), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            L.check(L.lib().tpu3_repatch_filter_f32(
                L.stream_of(xyz), B, N, k, r, closest.data_ptr() + 4, closest.stride(1), L.ptr(xyz), L.ptr(xyz_f),
                out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), L.ptr(small_cell)),
                "tpu3_repatch_filter_f32")
        return xyz_f, out[0], out[1], out[2], out[3]

    def repatch_seeds(self, seed_idx, patch_num, xyz_f):
        """seeds (B,P,3) = xyz_f[seed_idx[min(j, patch_num - 1)]] (upsampler.py:78-79 with padded patch slots)."""
        B, P = seed_idx.shape
        seeds = torch.empty((B, P, 3), dtype=torch.float32, device=xyz_f.device)
        with torch.cuda.device(xyz_f.device):
            L.check(L.lib().tpu3_repatch_seeds_f32(L.stream_of(xyz_f), B, xyz_f.size(1), P, L.ptr(seed_idx),
                                                   L.ptr(patch_num), L.ptr(xyz_f), L.ptr(seeds)), "tpu3_repatch_seeds_f32")
        return seeds

    def gather_xyz(self, x, idx, nchw_out=False):
        """x (B,N,3) f32 contiguous, idx (B,M) int32 -> rows (B,M,3), or (B,3,M) with nchw_out."""
        B, N, _ = x.shape
        M = idx.size(1)
        out = torch.empty((B, 3, M) if nchw_out else (B, M, 3), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            L.check(L.lib().tpu3_gather_xyz_f32(L.stream_of(x), B, N, M, L.ptr(x), L.ptr(idx), L.ptr(out),
                                                1 if nchw_out else 0), "tpu3_gather_xyz_f32")
        return out

    def denormalize(self, x, radius, centroid):
        """x (P,R,3) f32 contiguous, radius (P,), centroid (P,3) -> x * radius + centroid (two rounded operations)."""
        P, R, _ = x.shape
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            L.check(L.lib().tpu3_denormalize_f32(L.stream_of(x), P, R, L.ptr(x), L.ptr(radius), L.ptr(centroid),
                                                 L.ptr(out)), "tpu3_denormalize_f32")
        return out


BACKEND = HipBackend()


def normalize_point_batch(pc, NCHW=True):
    """normalize a batch of point clouds (operations.py:12-30)
    :param
        pc      [B, N, 3] or [B, 3, N]
        NCHW    if True, treat the second dimension as channel dimension
    :return
        pc      normalized point clouds, same shape as input
        centroid [B, 1, 3] or [B, 3, 1] center of point clouds
        furthest_distance [B, 1, 1] scale of point clouds
    One fused kernel for fp32 3-channel inference input; the differentiable torch formulation of
    the reference when a gradient is needed (training re-normalises network outputs)."""
    fused = (pc.dim() == 3 and pc.dtype == torch.float32
             and not (pc.requires_grad and torch.is_grad_enabled())
             and pc.size(1 if NCHW else 2) == 3)
    if fused:
        x = pc if NCHW else pc.transpose(2, 1)
        out, centroid, radius = BACKEND.normalize(x.contiguous())
        if not NCHW:
            out, centroid = out.transpose(2, 1).contiguous(), centroid.transpose(2, 1).contiguous()
        return out, centroid, radius
    point_axis = 2 if NCHW else 1
    dim_axis = 1 if NCHW else 2
    centroid = torch.mean(pc, dim=point_axis, keepdim=True)
    pc = pc - centroid
    furthest_distance, _ = torch.max(
        torch.sqrt(torch.sum(pc ** 2, dim=dim_axis, keepdim=True)), dim=point_axis, keepdim=True)
    pc = pc / furthest_distance
    return pc, centroid, furthest_distance


def knn_query(k, query, points, unique=True, layout=None, want_dist=True, want_grouped=True, unique_cache=None):
    """Channel-last kNN: query (B,M,C), points (Bp,N,C) -> (idx int64 (B,M,k), dist (B,M,k),
    grouped (B,M,k,C)).  The batched / ragged form of group_knn (see HipBackend.knn)."""
    if unique_cache is not None:
        return BACKEND.knn(k, query.contiguous(), points.contiguous(), unique, layout, want_dist, want_grouped,
                           unique_cache=unique_cache)
    return BACKEND.knn(k, query.contiguous(), points.contiguous(), unique, layout, want_dist, want_grouped)


def group_knn(k, query, points, unique=True, NCHW=True):
    """group batch of points to neighborhoods (operations.py:165-216)
    :param
        k: neighborhood size
        query: BxCxM or BxMxC
        points: BxCxN or BxNxC
        unique: neighborhood contains *unique* points
        NCHW: if true, the second dimension is the channel dimension
    :return
        neighbor_points BxCxMxk (if NCHW) or BxMxkxC (otherwise)
        index_batch     BxMxk   (int64, ascending distance, ties to the lowest index)
        distance_batch  BxMxk
    The neighbours are returned as the reference returns them: a (B,C,M,k) permuted view of
    (B,M,k,C) storage.  When `points` needs a gradient the gather is a differentiable
    torch.gather as in the reference (:209-211); indices and distances carry no gradient."""
    if NCHW:
        points_trans = points.transpose(2, 1).contiguous()
        query_trans = query.transpose(2, 1).contiguous()
    else:
        points_trans = points.contiguous()
        query_trans = query.contiguous()
    batch_size, num_points, _ = points_trans.size()
    assert(num_points >= k), "points size must be greater or equal to k"
    need_grad = points_trans.requires_grad and torch.is_grad_enabled()
    # Host tensors (the reference's data.py:135-139 calls group_knn on CPU tensors to cut training patches): the
    # search still runs on the HIP kernel -- inputs are staged to the current ROCm device and the results come back
    # as host tensors.  There is no CPU implementation behind this: without a device the call fails.
    staged = getattr(BACKEND, "name", "") == "hip-gfx950" and not points_trans.is_cuda
    if staged:
        if query_trans.is_cuda:
            raise RuntimeError("group_knn: query is on %s but points on the host" % query_trans.device)
        if not torch.cuda.is_available():
            raise RuntimeError("group_knn: host tensors are searched on the ROCm device, and none is available")
        dev = torch.device("cuda", torch.cuda.current_device())
        q_dev, p_dev = query_trans.detach().to(dev, torch.float32), points_trans.detach().to(dev, torch.float32)
    else:
        q_dev, p_dev = query_trans.detach(), points_trans.detach()
    with torch.no_grad():
        point_indices, distances, knn_trans = BACKEND.knn(k, q_dev, p_dev, unique, None, True, not need_grad)
    if staged:
        point_indices, distances = point_indices.cpu(), distances.to("cpu", points_trans.dtype)
        knn_trans = None if knn_trans is None else knn_trans.to("cpu", points_trans.dtype)
    if need_grad:
        knn_trans = torch.gather(points_trans.unsqueeze(1).expand(-1, query_trans.size(1), -1, -1), 2,
                                 point_indices.unsqueeze(-1).expand(-1, -1, -1, points_trans.size(-1)))
    if NCHW:
        knn_trans = knn_trans.permute(0, 3, 1, 2)
    return knn_trans, point_indices, distances


def group_ball(radius, nsample, query, points, NCHW=True):
    """Ball-query grouping -- the consumer the reference's `sampling.ball_query` export
    (sampling.cpp:59-81, sampling_cuda.cu:269-317) never got: for every query point the first
    `nsample` points within `radius` in index order, the first hit replicated into the unused slots
    (all zeros = point 0 when the ball is empty, as the kernel leaves them).
    :param
        radius, nsample   ball radius and neighbourhood size
        query   Bx3xM or BxMx3
        points  Bx3xN or BxNx3
    :return
        neighbor_points Bx3xMxnsample (if NCHW) or BxMxnsamplex3 (otherwise)
        index_batch     BxMxnsample int32
    The gather is differentiable with respect to `points` (torch.gather); indices carry no gradient."""
    if NCHW:
        points_trans = points.transpose(2, 1).contiguous()
        query_trans = query.transpose(2, 1).contiguous()
    else:
        points_trans = points.contiguous()
        query_trans = query.contiguous()
    assert(points_trans.size(2) == 3 and query_trans.size(2) == 3), "ball query is implemented for 3D points"
    with torch.no_grad():
        idx = BACKEND.ball_query(query_trans.detach(), points_trans.detach(), radius, nsample)
    grouped = torch.gather(points_trans.unsqueeze(1).expand(-1, query_trans.size(1), -1, -1), 2,
                           idx.long().unsqueeze(-1).expand(-1, -1, -1, points_trans.size(-1)))
    if NCHW:
        grouped = grouped.permute(0, 3, 1, 2)
    return grouped, idx


class GatherFunction(torch.autograd.Function):
    """features (B,C,N), idx (B,npoint) -> (B,C,npoint)   (operations.py:219-263)."""

    @staticmethod
    def forward(ctx, features, idx):
        features = features.contiguous()
        idx = idx.contiguous().to(dtype=torch.int32)
        _, C, N = features.size()
        output = BACKEND.gather_forward(features, idx)
        ctx.save_for_backward(idx)
        ctx.C = C
        ctx.N = N
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, = ctx.saved_tensors
        grad_features = BACKEND.gather_backward(grad_out.contiguous(), idx, ctx.C, ctx.N)
        return grad_features, None


gather_points = GatherFunction.apply


# Set by pipeline._upsample while it enqueues one sub-batch on its stream: called with "network_done" just before a level's
# resampling FPS is enqueued and with "resample_enqueued" just after (the stage boundaries the sub-batches are staggered at).
STAGE_HOOK = None


def fps(xyz, npoint, n_arr=None, m_arr=None):
    """Channel-last FPS: xyz (B,N,3) -> idx int32 (B,npoint); ragged counts optional."""
    return BACKEND.fps(xyz.contiguous(), npoint, n_arr, m_arr)


def furthest_point_sample(xyz, npoint, NCHW=True):
    """(operations.py:303-323)
    :param
        xyz (B, 3, N) or (B, N, 3)
        npoint a constant
    :return
        idx     (B, npoint) int32 indices of the sampled points (non-differentiable)
        points  (B, 3, npoint) or (B, npoint, 3) sampled point sets"""
    assert(xyz.dim() == 3), "input for furthest sampling must be a 3D-tensor, but xyz.size() is {}".format(xyz.size())
    if NCHW:
        xyz = xyz.transpose(2, 1).contiguous()
    assert(xyz.size(2) == 3), "furthest sampling is implemented for 3D points"
    with torch.no_grad():
        idx = BACKEND.fps(xyz.detach().contiguous(), npoint)
    sampled_pc = gather_points(xyz.transpose(2, 1).contiguous(), idx)
    if not NCHW:
        sampled_pc = sampled_pc.transpose(2, 1).contiguous()
    return idx, sampled_pc
