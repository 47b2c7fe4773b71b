"""Batched patch-upsampling pipeline: the inference driver of the reference
(main.py:214-246 `pc_prediction` + :375-380 concat and final FPS), re-designed for MI355X.

The reference walks the outer patches of ONE cloud in a Python loop at batch 1 (48 x
`net.forward`, each with ~4 host syncs).  Here every stage is one batched launch over all
patches of all clouds handed in:

    clouds (C,3,N)
      -> FPS seeds            (C, P)            P = int(N / num_point * patch_num_ratio)
      -> kNN patches          (C*P, 3, num_point)   group_knn(unique=True), one group per cloud
      -> normalise -> Net (all C*P patches advance through the levels together) -> de-normalise
      -> per-cloud concat in patch order (C, P*num_point*r, 3)
      -> final FPS            (C, 3, N*r)

Multi-GPU (one process per GPU, torch.distributed over RCCL): clouds -- or, for a single
cloud, its outer patches -- are split across ranks; the only communication is ONE all-gather
of fp32 xyz at the point where the reference concatenates (`shard="clouds"`: the finished
clouds; `shard="patches"`: the upsampled patches, after which the final FPS, which does not
shard, runs replicated).
"""
import os

import torch
import torch.distributed as dist

from .network import operations

# TPU3_FORCE_COLLECTIVES=1: take the sharded code paths -- shard_range, the MAX all-reduce of the event flags, the
# all-gather -- whenever a process group exists, even at world size 1.  A single-GPU box can then run every RCCL call
# of the N-rank path (tests/test_rccl_world1.py); at world size 1 the results are those of the unsharded call.
FORCE_COLLECTIVES = os.environ.get("TPU3_FORCE_COLLECTIVES", "0") not in ("0", "")


def _distributed(world):
    return world > 1 or (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


def num_outer_patches(num_shape_point, num_point, patch_num_ratio=3):
    """main.py:225."""
    return int(num_shape_point / num_point * patch_num_ratio)


@torch.no_grad()
def extract_outer_patches(clouds, num_point, patch_num_ratio=3):
    """main.py:225-235 for a batch of clouds: (C,3,N) -> seeds idx (C,P) int32,
    patches (C,P,num_point,3) channel-last, patch point indices (C,P,num_point) int64."""
    C, _, N = clouds.shape
    P = num_outer_patches(N, num_point, patch_num_ratio)
    cl = clouds.transpose(2, 1).contiguous()
    seed_idx = operations.fps(cl, P)
    be = operations.BACKEND
    if cl.is_cuda and hasattr(be, "gather_xyz"):
        seeds = be.gather_xyz(cl, seed_idx)
    else:
        seeds = torch.gather(cl, 1, seed_idx.long().unsqueeze(-1).expand(-1, -1, 3))
    layout = dict(grp=torch.arange(C, dtype=torch.int32, device=clouds.device), groups=C) if C > 1 else None
    idx, _, patches = operations.knn_query(num_point, seeds, cl, unique=True, layout=layout,
                                           want_dist=False)
    return seed_idx, patches, idx


@torch.no_grad()
def upsample_patches(net, patches_cl, up_ratio, levels_out=None):
    """main.py:237-244 for all patches at once: (Q,num_point,3) channel-last, un-normalised ->
    (Q, num_point*up_ratio, 3) de-normalised, and the normalised input patches (Q,3,num_point).
    levels_out: optional list; receives the cloud every patch holds after each level, de-normalised like the final
    output, (Q, 3, num_point * step^l) per level (parity diagnostics: tests/test_c2_parity.py)."""
    be = operations.BACKEND
    if (patches_cl.is_cuda and hasattr(be, "normalize_cl") and hasattr(net, "forward_eval_cl") and not net.training
            and patches_cl.dtype == torch.float32):
        # (r6) channel-last end to end: normalise, the net's eval path and the de-normalisation are three calls with
        # no transpose / multiply / add launches in between (they were nine ATen launches per call)
        norm_cl, centroid, radius = be.normalize_cl(patches_cl.contiguous())           # (Q,n,3), (Q,3), (Q,)
        if levels_out is not None:
            saved, net.trace = net.trace, []
        try:
            up_cl = net.forward_eval_cl(norm_cl, up_ratio)
            if levels_out is not None:
                for rec in net.trace:
                    levels_out.append(be.denormalize(rec["cloud"].contiguous(), radius, centroid).transpose(2, 1))
        finally:
            if levels_out is not None:
                net.trace = saved
        return be.denormalize(up_cl, radius, centroid), norm_cl.transpose(2, 1)
    patch = patches_cl.transpose(2, 1).contiguous()
    patch, centroid, radius = operations.normalize_point_batch(patch, NCHW=True)
    if levels_out is not None:
        saved, net.trace = net.trace, []
    try:
        up = net.forward(patch, ratio=up_ratio)
        if levels_out is not None:
            for rec in net.trace:
                levels_out.append(rec["cloud"].transpose(2, 1) * radius + centroid)
    finally:
        if levels_out is not None:
            net.trace = saved
    up = up * radius + centroid
    return up.transpose(2, 1).contiguous(), patch


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _all_gather_cat(t):
    """all-gather equal-shaped tensors of every rank and concatenate along dim 0 (rank order)."""
    rank, world = _world()
    if not _distributed(world):
        return t
    t = t.contiguous()
    out = torch.empty((world * t.size(0),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    try:
        dist.all_gather_into_tensor(out, t)
    except (RuntimeError, NotImplementedError):      # backends without the flat form
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        out = torch.cat(parts, dim=0)
    return out


def shard_range(total, rank, world):
    """Contiguous, padded split: every rank gets ceil(total/world) slots; slots past `total`
    repeat the last item (their results are dropped after the gather)."""
    per = (total + world - 1) // world
    ids = [min(rank * per + i, total - 1) for i in range(per)]
    return ids, per


def _any_rank(flag, device):
    """True on every rank if `flag` is true on any rank (one MAX all-reduce of a single word; no-op without a
    process group).  Decisions that change the NUMBER of collectives a rank issues must be taken on this."""
    rank, world = _world()
    if not _distributed(world):
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


@torch.no_grad()
def upsample(net, clouds, num_point, up_ratio, patch_num_ratio=3, shard=None, final_fps=True,
             timing=None, fps_stream=None, net_streams=None, sub_batch=4, fps_offset=0, check_small=None,
             optimistic_graph=None, stagger=None):
    """See _upsample.
    check_small: after enqueueing, synchronise, (i) recompute with the exact kNN-graph form if an optimistic graph
    call reported possibly duplicated feature rows, and (ii) raise if the outlier filter left some cloud with fewer
    points than a patch at some level (unreachable for finite input, see Net.small_cloud_events).  Default: on for
    plain calls, off when the caller overlaps work on side streams (fps_stream / net_streams) -- such callers
    (bench.py) read `net.small_cloud_events` and `BACKEND.graph_dup_events()` at their own synchronisation point.
    optimistic_graph: run the feature-space kNN graphs in their one-pass self-checking form (the backend's default
    is the exact gated form).  Default: exactly when this call also performs the check (check_small); a caller that
    passes True with check_small=False owns the check.
    Sharded calls: every rank takes the SAME decision (one MAX all-reduce of the event flags), so no rank issues
    collectives the others do not."""
    if check_small is None:
        check_small = fps_stream is None and not net_streams
    if optimistic_graph is None:
        optimistic_graph = bool(check_small)
    if check_small and hasattr(net, "reset_small_cloud_events"):
        net.reset_small_cloud_events()
    be = operations.BACKEND
    has_events = hasattr(be, "graph_dup_events")
    if check_small and has_events:
        be.graph_dup_events(reset=True)
    n_timing = len(timing) if timing is not None else 0

    def run(optimistic):
        saved = getattr(be, "optimistic_graph", None)
        if saved is not None:
            be.optimistic_graph = bool(optimistic)
        try:
            return _upsample(net, clouds, num_point, up_ratio, patch_num_ratio, shard, final_fps, timing, fps_stream,
                             net_streams, sub_batch, fps_offset, stagger)
        finally:
            if saved is not None:
                be.optimistic_graph = saved

    out = run(optimistic_graph)
    sharded = shard is not None
    if check_small and optimistic_graph and has_events:
        hit = bool(be.graph_dup_events(reset=True))
        if _any_rank(hit, clouds.device) if sharded else hit:
            # an optimistic feature-space kNN graph met (possibly) duplicated rows: recompute with the exact form
            if timing is not None:
                del timing[n_timing:]                   # the first pass's events are not this call's result
            out = run(False)
    if check_small and hasattr(be, "fps_cluster_faults"):
        # A multi-workgroup FPS launch whose members never all became resident gives up (bounded polls), counts a
        # fault and leaves its samples incomplete (filled with index 0).  Like the optimistic graph above, the call is
        # then recomputed -- on the single-workgroup kernels, which need no residency.  The decision is taken on EVERY
        # rank together (one MAX all-reduce): a rank that recomputed alone would issue collectives the others do not.
        hit = be.fps_cluster_faults(reset=True) > 0
        if _any_rank(hit, clouds.device) if sharded else hit:
            if timing is not None:
                del timing[n_timing:]
            saved = be.fps_cluster(0)
            try:
                out = run(False)
            finally:
                be.fps_cluster(saved)
            again = be.fps_cluster_faults(reset=True)
            if again:           # (cannot happen: no cluster launch was issued)
                raise RuntimeError("%d workgroups of an FPS launch gave up although the single-workgroup kernels "
                                   "were forced" % again)
    if check_small and hasattr(net, "small_cloud_events"):
        bad = net.small_cloud_events
        if _any_rank(bad, clouds.device) if sharded else bad:
            raise RuntimeError("%d cloud/level pairs on this rank were smaller than num_point=%d after the outlier "
                               "filter (NaN / Inf coordinates, or every point duplicated?); the batched pipeline does "
                               "not cover that case" % (bad, num_point))
    return out


# Staggering of concurrent sub-batches (upsample(net_streams=...)): on by default, TPU3_STAGGER=0 turns it off (A/B runs)
_STAGGER = int(os.environ.get("TPU3_STAGGER", "1")) != 0


@torch.no_grad()
def _upsample(net, clouds, num_point, up_ratio, patch_num_ratio=3, shard=None, final_fps=True,
              timing=None, fps_stream=None, net_streams=None, sub_batch=4, fps_offset=0, stagger=None):
    """Upsample a batch of clouds (C,3,N) -> (C,3,N*up_ratio)  [main.py test() :360-380 without
    the file I/O].  `shard`: None (no distribution), "clouds" or "patches" (see module doc).
    Every rank passes the same `clouds` and receives the full result.
    `timing`: optional list; a (start, end) pair of torch.cuda.Event bracketing the final-FPS
    launch on its stream is appended per call (bench.py's roofline measurement).
    `fps_stream`: optional side stream for the final FPS.  That kernel is one long dependent chain
    on ONE compute unit per cloud; on its own stream it overlaps with the network stages of the
    next call, which use the other 255 CUs.  The result is then only valid after that stream has
    been synchronised (the caller's job).  A LIST of side streams (together with `net_streams`)
    launches the final FPS per sub-batch, as soon as that sub-batch's network stages are done, on
    the list's streams in turn starting at `fps_offset`: fewer clouds per launch (one per XCD keeps
    a cloud's working set in that XCD's L2) and an earlier start.
    `net_streams`: optional list of streams; the clouds are then split into sub-batches of
    `sub_batch` clouds whose network stages run concurrently (round-robin over the streams).  The
    per-level resampling FPS is a latency chain on a few wavefronts per patch set while the kNN /
    DenseEdgeConv kernels are throughput-bound: one sub-batch's FPS hides under another's matrix
    work.  `stagger` (default: on, TPU3_STAGGER=0 off): the sub-batches enter a level one behind the
    other instead of in lockstep, and with a single `fps_stream` their join happens on THAT stream,
    so the caller's next call does not wait for this one's last sub-batch."""
    C, _, N = clouds.shape
    rank, world = _world()
    # main.py:379-380: the one big FPS down to N * up_ratio points per cloud
    def final(merged):
        if timing is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        idx = operations.fps(merged, N * up_ratio)
        if timing is not None:
            ev[1].record()
            timing.append(ev)
        if merged.is_cuda and hasattr(operations.BACKEND, "gather_xyz"):
            return operations.BACKEND.gather_xyz(merged.contiguous(), idx, nchw_out=True)   # (C,3,M) in one launch
        out = torch.gather(merged, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3))
        return out.transpose(2, 1).contiguous()

    if not _distributed(world):
        shard = None
    if shard is None and net_streams and len(net_streams) > 1 and C > 1:
        cur = torch.cuda.current_stream()
        parts = []
        per = max(1, min(int(sub_batch), -(-C // len(net_streams))))
        split_fps = final_fps and isinstance(fps_stream, (list, tuple)) and len(fps_stream) > 0
        result = clouds.new_empty((C, 3, N * up_ratio)) if split_fps else None
        for s in net_streams:
            s.wait_stream(cur)
        if split_fps:
            for f in fps_stream:
                f.wait_stream(cur)                      # `result` is allocated on the current stream
        mode = _STAGGER if stagger is None else bool(stagger)
        prev_ev = None
        for i, lo in enumerate(range(0, C, per)):
            s = net_streams[i % len(net_streams)]
            with torch.cuda.stream(s):
                my_ev = []
                if mode:
                    # STAGGER: a stage = a level's network kernels + its resampling FPS (operations.STAGE_HOOK marks
                    # the boundaries).  Sub-batch i+1 enters stage k only when sub-batch i has left it (events, no host
                    # synchronisation): the sub-batches spread over the stages instead of all reaching the same FPS
                    # at once.  (The per-level FPS is not a few idle units to hide under other work -- its 16-wave
                    # workgroups take a compute unit's whole register file each, 192 per 4-cloud sub-batch --, so the
                    # gain is the smoother mix and, below, no drain per call: ~1 % of the bench step.)
                    def hook(what, s=s, my_ev=my_ev, prev_ev=prev_ev):
                        if what == "resample_enqueued":
                            e = torch.cuda.Event()
                            e.record(s)                             # this sub-batch has left stage len(my_ev)
                            my_ev.append(e)
                            if prev_ev is not None and len(my_ev) < len(prev_ev):
                                s.wait_event(prev_ev[len(my_ev)])   # the next stage: after the sub-batch in front
                    if prev_ev:
                        s.wait_event(prev_ev[0])
                    operations.STAGE_HOOK = hook
                try:
                    sub = clouds[lo:lo + per]
                    _, patches, _ = extract_outer_patches(sub, num_point, patch_num_ratio)
                    P = patches.size(1)
                    up, _ = upsample_patches(net, patches.reshape(sub.size(0) * P, num_point, 3), up_ratio)
                    part = up.reshape(sub.size(0), P * up.size(1), 3)
                finally:
                    operations.STAGE_HOOK = None
                prev_ev = my_ev if mode else None
            if split_fps:
                f = fps_stream[(fps_offset + i) % len(fps_stream)]
                f.wait_stream(s)
                part.record_stream(f)
                with torch.cuda.stream(f):
                    result[lo:lo + per].copy_(final(part))
                result.record_stream(f)
            else:
                parts.append(part)
        if split_fps:
            return result
        if mode and final_fps and fps_stream is not None and not isinstance(fps_stream, (list, tuple)):
            # the join of the sub-batches on the FINAL-FPS stream, not on the caller's: the caller's next call then
            # finds its network streams free as each finishes -- no drain of the staggered pipeline per call
            for s in net_streams:
                fps_stream.wait_stream(s)
            for t in parts:
                t.record_stream(fps_stream)
            with torch.cuda.stream(fps_stream):
                return final(torch.cat(parts, dim=0))
        for s in net_streams:
            cur.wait_stream(s)
        for t in parts:
            t.record_stream(cur)
        merged = torch.cat(parts, dim=0)
    elif shard is None:
        _, patches, _ = extract_outer_patches(clouds, num_point, patch_num_ratio)
        P = patches.size(1)
        up, _ = upsample_patches(net, patches.reshape(C * P, num_point, 3), up_ratio)
        merged = up.reshape(C, P * up.size(1), 3)
    elif shard == "clouds":
        ids, per = shard_range(C, rank, world)
        mine = clouds[ids]
        local = _upsample(net, mine, num_point, up_ratio, patch_num_ratio, shard=None, final_fps=final_fps,
                          timing=timing, fps_stream=fps_stream, net_streams=net_streams, sub_batch=sub_batch,
                          fps_offset=fps_offset, stagger=stagger)
        for f in (fps_stream if isinstance(fps_stream, (list, tuple)) else [fps_stream]):
            if f is not None:
                torch.cuda.current_stream().wait_stream(f)
        out = _all_gather_cat(local)                         # (world*per, 3, N*r) in cloud order
        return out[:C]
    elif shard == "patches":
        # seeds + outer kNN are cheap and deterministic: every rank computes them redundantly
        _, patches, _ = extract_outer_patches(clouds, num_point, patch_num_ratio)
        P = patches.size(1)
        flat = patches.reshape(C * P, num_point, 3)
        ids, per = shard_range(C * P, rank, world)
        up_local, _ = upsample_patches(net, flat[ids], up_ratio)
        up = _all_gather_cat(up_local)[:C * P]               # patch order restored by rank order
        merged = up.reshape(C, P * up.size(1), 3)
    else:
        raise ValueError("shard must be None, 'clouds' or 'patches'")
    if not final_fps:
        return merged.transpose(2, 1).contiguous()
    if isinstance(fps_stream, (list, tuple)):
        fps_stream = fps_stream[fps_offset % len(fps_stream)] if fps_stream else None
    if fps_stream is None:
        return final(merged)
    fps_stream.wait_stream(torch.cuda.current_stream())
    merged.record_stream(fps_stream)
    with torch.cuda.stream(fps_stream):
        return final(merged)


class GraphedUpsample(object):
    """`upsample` for ONE input shape as a hipGraph (reference main.py:360-380 handles one cloud
    at a time, i.e. this is the reference's literal usage at its lowest latency).

    The eval path has no host synchronisation and fixed padded shapes, so the ~390 launches of a cloud are captured
    once and replayed: the ~5 ms of launch gaps of a 35 ms cloud go away.  The input is copied into a static buffer,
    the result is the graph's static output (valid until the next call; `clone=True` hands out a copy).

    The checks of `upsample(check_small=True)` run after every replay, at the synchronisation the caller needs
    anyway for the result: if an optimistic kNN graph asked for the exact form, a cluster-FPS launch faulted or a
    cloud fell below one patch, the call is answered by the eager `upsample` (which recomputes / raises as
    documented there), so a replayed result is never handed out unchecked.  `check=False` leaves result and check
    to the caller (it must then read BACKEND.graph_dup_events / fps_cluster_faults / net.small_cloud_events itself).

    Weights: the kernels read most parameters through their storage, but the DenseEdgeConv operand tables
    (layers.DenseEdgeConv._operand_pack) and the folded prep convolutions (Level._fold_plan) are DERIVED blobs built
    on the host side of a call -- a replay would keep using the blobs of the capture.  So every call compares the
    parameters' (version counter, address) key with the one recorded at capture and captures again when it changed
    (optimizer.step(), load_state_dict, `p.copy_()`, a replaced parameter tensor, set_mlp_precision).  Edits through
    `param.data` bump no counter (neither here nor in the eager caches): call `net.invalidate_weight_caches()` and
    `GraphedUpsample.invalidate()` after such an edit.

    A captured cluster-form final FPS (csrc/fps_cluster.hip) waits for its member workgroups to be co-resident; a
    replay that finds the device busy with other streams' work can fault like an eager launch can, which the check
    below turns into the eager recompute."""

    def __init__(self, net, shape, num_point, up_ratio, patch_num_ratio=3, device=None, final_fps=True):
        dev = torch.device(device) if device is not None else next(net.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("GraphedUpsample needs a ROCm device")
        self.net, self.num_point, self.up_ratio, self.patch_num_ratio = net, num_point, up_ratio, patch_num_ratio
        self.final_fps = final_fps
        self.shape = tuple(shape)
        self.static_in = torch.zeros(self.shape, dtype=torch.float32, device=dev)
        self.static_out = None
        self.graph = None
        self.weights_key = None
        self.captures = 0
        self.stream = torch.cuda.Stream(device=dev)

    def _weights_key(self):
        ps = list(self.net.parameters())
        be = operations.BACKEND
        return (tuple(p._version for p in ps), tuple(p.data_ptr() for p in ps),
                tuple(getattr(m, "mlp_precision", None) for m in self.net.modules() if hasattr(m, "mlp_precision")),
                be.split_bf16() if hasattr(be, "split_bf16") else None)     # (the capture bakes the kernel choice in)

    def invalidate(self):
        """Forget the captured graph (after a weight edit the version counters cannot see)."""
        self.graph = None
        self.static_out = None
        self.weights_key = None

    def _body(self):
        return _upsample(self.net, self.static_in, self.num_point, self.up_ratio, self.patch_num_ratio, None,
                         self.final_fps)

    @torch.no_grad()
    def _capture(self):
        be = operations.BACKEND
        saved = getattr(be, "optimistic_graph", None)
        if saved is not None:
            be.optimistic_graph = True
        try:
            # eager warm-up ON THE CAPTURE STREAM: allocator pools, the backend's event words, the levels' folded
            # weights (keyed to the stream that built them), kernel attributes -- nothing of that may happen inside
            # the capture
            self.stream.wait_stream(torch.cuda.current_stream(self.static_in.device))
            with torch.cuda.stream(self.stream):
                self._body()
            self.stream.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.static_out = self._body()
            self.weights_key = self._weights_key()
            self.captures += 1
        finally:
            if saved is not None:
                be.optimistic_graph = saved

    @torch.no_grad()
    def __call__(self, clouds, check=True, clone=False):
        if tuple(clouds.shape) != self.shape:
            raise ValueError("GraphedUpsample was built for %s, got %s" % (self.shape, tuple(clouds.shape)))
        be = operations.BACKEND

        def clean_counters():
            if hasattr(self.net, "reset_small_cloud_events"):
                self.net.reset_small_cloud_events()
            if hasattr(be, "graph_dup_events"):
                be.graph_dup_events(reset=True)
            if hasattr(be, "fps_cluster_faults"):       # (a fault left by ANY earlier call would send this one to eager)
                be.fps_cluster_faults(reset=True)
        if check:
            clean_counters()
        self.static_in.copy_(clouds)
        if self.graph is not None and self.weights_key != self._weights_key():
            self.graph = None                           # the packed / folded weight blobs of the capture are stale
        if self.graph is None:
            self._capture()
            if check:           # (the warm-up and the capture itself ran the kernels: start from clean counters)
                torch.cuda.synchronize(self.static_in.device)
                clean_counters()
        self.graph.replay()
        if check:
            torch.cuda.synchronize(self.static_in.device)
            bad = hasattr(be, "graph_dup_events") and be.graph_dup_events(reset=True)
            bad = bad or (hasattr(be, "fps_cluster_faults") and be.fps_cluster_faults(reset=True) > 0)
            bad = bad or (hasattr(self.net, "small_cloud_events") and self.net.small_cloud_events)
            if bad:
                return upsample(self.net, clouds, self.num_point, self.up_ratio, self.patch_num_ratio,
                                final_fps=self.final_fps, optimistic_graph=False)
        return self.static_out.clone() if clone else self.static_out


def pc_prediction(net, input_pc, num_point, up_ratio, patch_num_ratio=3):
    """Drop-in shaped like the reference's pc_prediction (main.py:214-246):
    input_pc 1x3xN -> (input_list of [1x3xM] normalised patches, up_point_list of [1x3xMr])."""
    _, patches, _ = extract_outer_patches(input_pc, num_point, patch_num_ratio)
    P = patches.size(1)
    up, patch = upsample_patches(net, patches.reshape(P, num_point, 3), up_ratio)
    up = up.transpose(2, 1).contiguous()
    return [patch[i:i + 1] for i in range(P)], [up[i:i + 1] for i in range(P)]
