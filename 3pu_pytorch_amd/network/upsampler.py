"""Progressive patch upsampler of 3PU for PyTorch-ROCm -- counterpart of the reference's
network/upsampler.py (Net :9-189, Level :192-374, and the never-instantiated AdaptiveLevel :377-512).  Same constructor arguments, module names and parameter shapes (reference checkpoints
load unchanged), same numerical definition of every step; the execution is re-designed:

  * `Net.forward` in eval mode accepts ANY batch of input patches (the reference asserts
    batch 1, :61) and runs all of them through each level together: the data-dependent sizes
    the outlier filter creates (:63-80) are kept on the device as per-patch counts next to
    fixed-size padded tensors, so a whole cloud (48 outer patches x 4 levels) costs a few dozen
    launches and zero host synchronisations instead of ~200 Level calls and ~400 syncs;
  * the inter-level skip searches the previous level's merged cloud through an index map
    (`pts_of`) instead of `expand`-ing it per patch (:319-323), and de-duplicates it once per
    cloud on the device instead of once per patch on the host;
  * activations are channel-last (see layers.py).
"""
from collections import OrderedDict
from math import log, sqrt

import os

import weakref

import torch

from . import layers
from . import operations


def _arange_like(n, t):
    return torch.arange(n, device=t.device)


class Net(torch.nn.Module):
    """3PU inter-level plus skip connection and dense layers (reference :9-189)."""

    def __init__(self, max_up_ratio=16, step_ratio=2, knn=16, growth_rate=12,
                 dense_n=3, max_num_point=312, fm_knn=3, **kwargs):
        super(Net, self).__init__()
        self.max_up_ratio = max_up_ratio
        self.step_ratio = step_ratio
        self.knn = knn
        self.growth_rate = growth_rate
        self.dense_n = dense_n
        self.fm_knn = fm_knn
        self.num_levels = int(log(max_up_ratio, step_ratio))
        self.levels = torch.nn.ModuleDict()
        self.max_num_point = max_num_point
        for l in range(1, self.num_levels + 1):
            # NB the reference does not forward fm_knn to Level (:25-26): every level uses 5
            self.levels['level_%d' % l] = Level(
                dense_n=dense_n, growth_rate=growth_rate, knn=knn, step_ratio=step_ratio)
        if self.training:
            for m in self.modules():
                if isinstance(m, (torch.nn.Conv2d, torch.nn.Conv1d)):
                    torch.nn.init.xavier_uniform_(m.weight)
                    torch.nn.init.zeros_(m.bias)
                elif isinstance(m, (torch.nn.InstanceNorm1d, torch.nn.InstanceNorm2d,
                                    torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                    torch.nn.init.zeros_(m.bias)
                    torch.nn.init.ones_(m.weight)
        # number of (cloud, level) pairs whose outlier-filtered size was below one patch: one device scalar per
        # (device, stream) that produced them, accumulated in place without a synchronisation (bounded by the
        # number of streams, however many eval calls are made; see small_cloud_events)
        self._small_cloud_counts = {}
        # optional list: when set, the eval path appends one dict per level
        # (patch_xyz (P,3,k) un-normalised inputs, out_norm (P,3,k*r) level output, patch_num,
        #  cloud (B,k*r^l,3) the cloud held after the level, reference :163 / :156-159)
        self.trace = None

    # ------------------------------------------------------------------------------------------
    # training: one random kNN patch per batch element and level (reference :52-58, :83-105, :126-147)
    # ------------------------------------------------------------------------------------------
    def extract_xyz_feature_patch(self, batch_xyz, k, gt_xyz=None, gt_k=None):
        """Training-mode patch extraction (the reference's method name, :44): every element of the batch draws
        one seed point; its k nearest points of the cloud are the patch, its gt_k nearest points of the
        ground truth the target.  Bx3xN -> Bx3xk, (Bx3xgt_k | None).
        The seed draw is the reference's (:53: one int32 randint of shape (B,1) on the cloud's device), so a
        recorded seed stream replays bit for bit."""
        count, _, size = batch_xyz.shape
        pick = torch.randint(0, size, (count, 1), dtype=torch.int32, device=batch_xyz.device)
        centre = operations.gather_points(batch_xyz, pick)                      # (B,3,1)

        def around(points, n_near, **layout):
            near, _, _ = operations.group_knn(n_near, centre, points, unique=False, **layout)     # (B,3,1,n_near)
            return near[:, :, 0, :]
        patch = around(batch_xyz, k, NCHW=True)
        target = around(gt_xyz, gt_k) if (gt_xyz is not None and gt_k is not None) else None
        return patch, target

    def _forward_train(self, cloud, ratio, truth):
        depth = int(log(ratio, self.step_ratio))
        cap = min(cloud.size(-1), self.max_num_point)       # points a level sees at most
        carry = None                # (what the previous level saw, its features), both channel-last: the (B,264,N)
        # layout of the reference's features exists only at Level.forward's boundary -- between the levels of one
        # training forward it would be two 10 MB transposes per level and direction
        for at in range(1, depth + 1):
            level = self.levels['level_%d' % at]
            seen = cloud
            if carry is None:
                seen_cl = seen.transpose(2, 1).contiguous()
                out_cl, feat_cl = level.forward_cl(seen_cl, seen_cl, None)
                cloud = out_cl.transpose(2, 1).contiguous()
            else:
                if cloud.size(-1) > cap:
                    # the ground truth shrinks with the patch: cap * ratio / step_ratio^(at-1) points
                    target_size = cap * ratio // self.step_ratio ** at * self.step_ratio
                    seen, truth = self.extract_xyz_feature_patch(cloud, cap, gt_xyz=truth, gt_k=target_size)
                unit, centre, scale = operations.normalize_point_batch(seen, NCHW=True)
                seen_cl = seen.transpose(2, 1).contiguous()
                out_cl, feat_cl = level.forward_cl(seen_cl, unit.transpose(2, 1).contiguous(), carry + (None,))
                cloud = out_cl.transpose(2, 1).contiguous() * scale + centre
            carry = (seen_cl, feat_cl)
        return cloud, truth

    # ------------------------------------------------------------------------------------------
    # inference: all patches of the batch advance level by level together
    # ------------------------------------------------------------------------------------------
    def _repatch(self, xyz_cl, k):
        """Eval-mode patch extraction (reference :59-86) for a batch of clouds at once.
        xyz_cl (B,N,3) -> patches (B,P,k,3) with P = int(N/k*5), live patch count (B,) int32
        (patches beyond a cloud's count repeat its last live patch and are never merged), and -- on the device path --
        (patch_num * k, patch_num * k * step_ratio), the two ragged counts the level needs, else None.
        On a device the steps between the three kernels (kNN k = 2, FPS, kNN k) are two launches of csrc/glue.hip
        (r6: they were 22 ATen launches per level -- mean, compare, sum, a stable radix sort, gathers, index
        arithmetic -- each a dispatch gap in the one-cloud latency)."""
        B, N, _ = xyz_cl.shape
        dev = xyz_cl.device
        be = operations.BACKEND
        # distance to the closest neighbour; points far from everything are outliers (:63-73)
        _, closest_d, _ = operations.knn_query(2, xyz_cl, xyz_cl, unique=False, want_grouped=False)
        # patch_num = int(num_point / k * 5) per cloud, in double like Python (:76)
        P = int(N / k * 5)
        kk = min(k, N)
        if xyz_cl.is_cuda and hasattr(be, "repatch_filter") and closest_d.is_contiguous() and closest_d.size(-1) == 2:
            xyz_f, count, patch_num, old_count, m_count = be.repatch_filter(
                closest_d, xyz_cl.contiguous(), k, self.step_ratio, self._small_cloud_cell(dev))
            seed_idx = operations.fps(xyz_f, P, n_arr=count, m_arr=patch_num)
            seeds = be.repatch_seeds(seed_idx, patch_num, xyz_f)
            counts = (old_count, m_count)
        else:
            closest_d = closest_d[:, :, 1]
            mask = closest_d < (5 * torch.mean(closest_d, dim=1, keepdim=True))
            count = mask.sum(dim=1).to(torch.int32)
            order = torch.argsort((~mask).to(torch.uint8), dim=1, stable=True)   # masked_select order
            xyz_f = torch.gather(xyz_cl, 1, order.unsqueeze(-1).expand(-1, -1, 3))
            patch_num = torch.floor(count.to(torch.float64) / k * 5).to(torch.int32).clamp_(min=1)
            self._note_small_clouds((count < k).sum())
            seed_idx = operations.fps(xyz_f, P, n_arr=count, m_arr=patch_num)
            slot = torch.minimum(_arange_like(P, xyz_cl).view(1, P), (patch_num - 1).view(B, 1).long())
            seed_idx = torch.gather(seed_idx.long(), 1, slot)
            seeds = torch.gather(xyz_f, 1, seed_idx.unsqueeze(-1).expand(-1, -1, 3))
            counts = None
        _, _, patches = operations.knn_query(kk, seeds, xyz_f, unique=False,
                                             layout=dict(n_arr=count), want_dist=False)
        return patches, patch_num, counts

    def _small_cloud_cell(self, dev):
        """The int64 event word of the current (device, stream), created on first use (see _note_small_clouds)."""
        key = (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream)
        cell = self._small_cloud_counts.get(key)
        if cell is None:
            cell = self._small_cloud_counts[key] = torch.zeros((), dtype=torch.int64, device=dev)
        return cell

    def _index_rows(self, B, P, dev):
        """(arange(B), repeat_interleave(arange(B), P)) as int32 device tensors, cached per shape: the owner tables of a
        level's launches are constants of the call shape (two launches per level otherwise).  The tables are BUILT by
        kernels on one stream; pipeline.upsample runs sub-batches of the same net on several streams, so every other
        stream orders itself behind the build (an event, as for the fold plans) -- a table read before it is written
        would be an out-of-range owner index."""
        cache = self.__dict__.setdefault("_owner_tables", {})
        key = (B, P, dev.type, dev.index)
        hit = cache.get(key)
        on_device = dev.type == "cuda"
        here = torch.cuda.current_stream(dev) if on_device else None
        if hit is None:
            each = torch.arange(B, dtype=torch.int32, device=dev)
            owner = torch.repeat_interleave(each, P) if P > 1 else each
            done = None
            if on_device:
                done = torch.cuda.Event()
                done.record(here)
            hit = cache[key] = (each, owner, here.cuda_stream if on_device else None, done)
            if len(cache) > 64:
                cache.pop(next(iter(cache)))
        elif on_device and hit[2] != here.cuda_stream:
            here.wait_event(hit[3])
            hit[0].record_stream(here)
            hit[1].record_stream(here)
        return hit[0], hit[1]

    def _note_small_clouds(self, n):
        """n: 0-d device tensor.  Added to the scalar of the current (device, stream): kernels of different
        streams never update the same word."""
        key = (n.device.type, n.device.index, torch.cuda.current_stream(n.device).cuda_stream if n.is_cuda else 0)
        cell = self._small_cloud_counts.get(key)
        if cell is None:
            self._small_cloud_counts[key] = n.to(torch.int64).clone()
        else:
            cell.add_(n)

    @property
    def small_cloud_events(self):
        """Number of (cloud, level) pairs since the last reset whose filtered cloud had fewer points
        than a patch.  The reference shrinks k to the filtered size there (:75-78); the batched path keeps
        k = num_point and reports.  The case is unreachable for finite input without wholesale duplication:
        the filter keeps d_i < 5 * mean(d), and by Markov's inequality at most N/5 of N non-negative values
        reach five times their mean, so N' >= 0.8 N >= 1.6 k (a level's input has step_ratio >= 2 times the
        patch size).  It takes NaN / Inf coordinates or a cloud in which every point has a twin (all d = 0:
        nothing passes `0 < 0`) -- inputs on which the reference itself fails (empty FPS / torch.cat).
        Callers that care (pipeline.upsample(check_small=True), bench.py) read this after their
        synchronisation and raise.  Reading synchronises the device."""
        if not self._small_cloud_counts:
            return 0
        cells = list(self._small_cloud_counts.values())
        if any(c.is_cuda for c in cells):
            torch.cuda.synchronize()
        return int(sum(int(c) for c in cells))

    def reset_small_cloud_events(self):
        # in place: a captured hipGraph (pipeline.GraphedUpsample) keeps adding to the cell of its stream
        for cell in self._small_cloud_counts.values():
            cell.zero_()

    def invalidate_weight_caches(self):
        """Drop every blob DERIVED from the weights: the packed DenseEdgeConv operand tables
        (layers.DenseEdgeConv._operand_pack), the folded prep convolutions (Level._fold_plan) and the split-bf16 images of
        up_layer1's weights (operations.HipBackend._wide_split).  All are keyed by the
        parameters' version counters and addresses, which `optimizer.step()`, `load_state_dict` and every in-place
        tensor method bump; an edit through `param.data` (p.data.copy_(), p.data.mul_()) bumps neither, so call this
        after one.  (The unpacked kernels read the weights at every launch; only the cached blobs can go stale.)"""
        for m in self.modules():
            if isinstance(m, layers.DenseEdgeConv):
                layers._PACK_CACHES.pop(m, None)
            elif isinstance(m, Level):
                _FOLD_CACHES.pop(m, None)
                _CODE_CACHES.pop(m, None)
        if hasattr(operations.BACKEND, "invalidate_split_weights"):
            operations.BACKEND.invalidate_split_weights()       # split-bf16 images of up_layer1's weights
        return self

    def set_mlp_precision(self, precision, activations=None):
        """Arithmetic of the matrix-core kernels of the per-patch feature stacks (inference):
        "f32" -- fp32 operands (default; what the parity tests pin), or "f16" -- operands rounded to fp16,
        fp32 accumulate (BASELINE config C5: "fp16 feature MLPs on MFMA").  FPS, every kNN, the Chamfer
        distance and the 3 -> 24 coordinate embedding (layer0) stay fp32 either way.  Explicit: a layer shape the f16 kernels do not cover raises.
        activations: storage of every Level's (B,N,264) feature buffer, "f32" (default) or -- with precision "f16"
        only -- "f16": the rows are written, gathered by the next level's skip connection and read by the prep
        convolutions / up_layer1 as fp16 (half the bytes of the HBM-bound kernels).  The matrix kernels see the same
        fp16 operands either way; only the skip connection's arithmetic sees rounded rows."""
        if precision not in ("f32", "f16"):
            raise ValueError("mlp precision must be 'f32' or 'f16'")
        activations = activations or "f32"
        if activations not in ("f32", "f16") or (activations == "f16" and precision != "f16"):
            raise ValueError("activation storage must be 'f32', or 'f16' together with mlp precision 'f16'")
        for m in self.modules():
            if isinstance(m, Level):
                m.activation_storage = activations
        for m in self.modules():
            if isinstance(m, (layers.Conv1d, layers.Conv2d)) and m.conv.in_channels < 16:
                continue            # layer0 embeds the xyz coordinates themselves (3 -> 24): kept in fp32
            if isinstance(m, (layers.DenseEdgeConv, layers.Conv1d, layers.Conv2d, Level)):
                m.mlp_precision = precision
        self.invalidate_weight_caches()
        return self

    def _forward_eval(self, xyz, ratio):
        return self.forward_eval_cl(xyz.transpose(2, 1).contiguous(), ratio).transpose(2, 1).contiguous()

    def forward_eval_cl(self, xyz_cl, ratio=None):
        """The eval path on channel-last clouds: (B,N,3) -> (B,N*ratio,3).  `forward` (the reference's Bx3xN interface)
        wraps it in two transposes; pipeline.upsample_patches calls it directly."""
        ratio = ratio or self.max_up_ratio
        B, num_point, _ = xyz_cl.size()
        dev = xyz_cl.device
        be = operations.BACKEND
        num_levels = int(log(ratio, self.step_ratio))
        max_num_point = min(num_point, self.max_num_point)
        for l in range(1, num_levels + 1):
            curr_ratio = self.step_ratio ** l
            level = self.levels['level_%d' % l]
            if l == 1:
                # every input patch is its own reference call: its own unique-max group
                each, _ = self._index_rows(B, 1, dev)
                old_xyz, old_count = xyz_cl, None
                if self.trace is not None:
                    self.trace.append(dict(patch_xyz=xyz_cl.transpose(2, 1)))
                xyz_cl, old_feat = level.forward_cl(xyz_cl, xyz_cl, None, owner=each, groups=B)
                if self.trace is not None:
                    self.trace[-1]["out_norm"] = xyz_cl.transpose(2, 1)
                    self.trace[-1]["cloud"] = xyz_cl
                continue
            counts = None
            if xyz_cl.size(1) > max_num_point:
                patches, patch_num, counts = self._repatch(xyz_cl, max_num_point)      # (B,P,k,3)
            else:
                patches = xyz_cl.unsqueeze(1)
                patch_num = torch.ones((B,), dtype=torch.int32, device=dev)
            P, k = patches.size(1), patches.size(2)
            patch_cl = patches.reshape(B * P, k, 3)
            on_device = patch_cl.is_cuda and hasattr(be, "normalize_cl")
            if on_device:
                norm_cl, centroid, radius = be.normalize_cl(patch_cl.contiguous())      # (BP,k,3), (BP,3), (BP,)
            else:
                norm, centroid, radius = operations.normalize_point_batch(
                    patch_cl.transpose(2, 1).contiguous(), NCHW=True)
                norm_cl = norm.transpose(2, 1).contiguous()
            _, owner = self._index_rows(B, P, dev)
            up_cl, feat = level.forward_cl(
                patch_cl, norm_cl, (old_xyz, old_feat, old_count), owner=owner, groups=B, per_owner=P)
            if self.trace is not None:
                self.trace.append(dict(patch_xyz=patch_cl.transpose(2, 1), out_norm=up_cl.transpose(2, 1),
                                       patch_num=patch_num))
            if on_device:
                up_cl = be.denormalize(up_cl.contiguous(), radius, centroid)
            else:
                up_cl = up_cl * radius.view(-1, 1, 1) + centroid.view(-1, 1, 3)
            r = up_cl.size(1) // k
            # merge the patches of each cloud (reference :149-155): patch-major concatenation
            merged = up_cl.reshape(B, P * k * r, 3)
            old_xyz = patch_cl.reshape(B, P * k, 3)
            old_feat = feat.reshape(B, P * k, feat.size(-1))
            if counts is not None and r == self.step_ratio:
                old_count, m_count = counts
            else:
                old_count = (patch_num * k).contiguous()
                m_count = (patch_num * (k * r)).contiguous()
            xyz_cl = merged
            if P > 1:
                # resample to num_point * curr_ratio points (reference :156-159)
                num_output_point = num_point * curr_ratio
                hook = operations.STAGE_HOOK
                if hook is not None:
                    hook("network_done")        # (pipeline._upsample: the stagger of concurrent sub-batches)
                idx = operations.fps(merged, num_output_point, n_arr=m_count, m_arr=None)
                if hook is not None:
                    hook("resample_enqueued")
                if merged.is_cuda and hasattr(be, "gather_xyz"):
                    xyz_cl = be.gather_xyz(merged, idx)
                else:
                    xyz_cl = torch.gather(merged, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3))
            if self.trace is not None:
                self.trace[-1]["cloud"] = xyz_cl            # (B, num_point * curr_ratio, 3): what the next level sees
        return xyz_cl

    def forward(self, xyz, ratio=None, gt=None, **kwargs):
        """
        :param
            xyz     Bx3xN
            ratio   upscaling factor (integer)
            gt      Bx3x(max_up_ratio*N)
        :return
            xyz     Bx3x(ratio*N)
            (during training:)
            gt      Bx3x(ratio*N)
        """
        ratio = ratio or self.max_up_ratio
        if self.training:
            assert(gt is not None)
            return self._forward_train(xyz, ratio, gt)
        return self._forward_eval(xyz, ratio)


class _SkipTrain(torch.autograd.Function):
    """Inter-level skip connection of a Level under autograd: x (B,N,C) is updated in place by the fused forward,
    which also leaves the weights (B,N,K); backward hands g to x and scatters 0.2 * w_k * g_i into the rows of the
    previous level's features the point interpolated from."""

    @staticmethod
    def forward(ctx, x, prev_feat, xyz, prev_xyz, pts_of, idx):
        weights = operations.BACKEND.interlevel_skip_train(xyz, x, prev_xyz, prev_feat, pts_of, idx)
        ctx.mark_dirty(x)
        ctx.save_for_backward(weights, idx, pts_of if pts_of is not None else idx)
        ctx.owned = pts_of is not None
        ctx.prev_shape = tuple(prev_feat.shape)
        return x

    @staticmethod
    def backward(ctx, g):
        weights, idx, pts_of = ctx.saved_tensors
        gprev = None
        if ctx.needs_input_grad[1]:
            gprev = operations.BACKEND.interlevel_skip_backward(g.contiguous(), weights, idx,
                                                                pts_of if ctx.owned else None, ctx.prev_shape)
        return g, gprev, None, None, None, None


# fold plans per Level (Level._fold_plan): held outside the modules -- a plan carries a stream event, which neither
# copy.deepcopy(net) nor torch.save(net) may meet
_FOLD_CACHES = weakref.WeakKeyDictionary()
_CODE_CACHES = weakref.WeakKeyDictionary()      # Level._code_term


class Level(torch.nn.Module):
    """3PU per-level network (reference :192-374)."""

    mlp_precision = "f32"       # see Net.set_mlp_precision
    activation_storage = "f32"  # see Net.set_mlp_precision(activations=...)
    # inference: fold layer{2,3,4}_prep into the write-out of the DenseEdgeConv blocks before them (fp32 kernels)
    fold_preps = os.environ.get("TPU3_FOLD_PREPS", "1") not in ("0", "")

    def __init__(self, dense_n=3, growth_rate=12, knn=16, fm_knn=5, step_ratio=2):
        super(Level, self).__init__()
        self.dense_n = dense_n
        self.fm_knn = fm_knn
        self.step_ratio = step_ratio
        # code for feature expansion (:200-206).  The reference keeps it as a plain CPU attribute and
        # copies it to the device in every forward (:354) -- a pageable H2D copy, i.e. a host
        # synchronisation per Level call (measured: 190 ms of stalls per 8-cloud step).  A
        # non-persistent buffer moves with the module and stays out of the state_dict.
        if step_ratio < 4:
            code = self.gen_1d_grid(step_ratio).unsqueeze(0).detach()
        else:
            expansion_ratio = round(sqrt(step_ratio)) ** 2
            code = self.gen_grid(expansion_ratio).unsqueeze(0).detach()
        self.register_buffer("code", code, persistent=False)

        self.layer0 = layers.Conv2d(3, 24, [1, 1], activation=None)
        self.layer1 = layers.DenseEdgeConv(24, growth_rate=growth_rate, n=dense_n, k=knn)
        in_channels = 84  # 24+(24+growth_rate*dense_n) = 24+(24+36) = 84
        self.layer2_prep = layers.Conv1d(in_channels, 24, 1, activation="relu")
        self.layer2 = layers.DenseEdgeConv(24, growth_rate=growth_rate, n=dense_n, k=knn)
        in_channels = 144  # 84+(24+36) = 144
        self.layer3_prep = layers.Conv1d(in_channels, 24, 1, activation="relu")
        self.layer3 = layers.DenseEdgeConv(24, growth_rate=growth_rate, n=dense_n, k=knn)
        in_channels = 204  # 144+(24+36) = 204
        self.layer4_prep = layers.Conv1d(in_channels, 24, 1, activation="relu")
        self.layer4 = layers.DenseEdgeConv(24, growth_rate=growth_rate, n=dense_n, k=knn)
        in_channels = 264  # 204+(24+36) = 264
        self.up_layer = torch.nn.Sequential(OrderedDict([
            ("up_layer1", layers.Conv2d(in_channels + self.code.size(1), 128, 1, activation="relu")),
            ("up_layer2", layers.Conv2d(128, 128, 1, activation="relu")), ]))
        self.fc_layer1 = layers.Conv2d(128, 64, 1, activation="relu")
        self.fc_layer2 = layers.Conv2d(64, 3, 1, activation=None)

    @staticmethod
    def exponential_distance_cl(points, knn_points):
        """Bilateral weight of the inter-level skip (reference :232-250), channel-last:
        points (B,N,C), knn_points (B,N,K,C) -> weight (B,N,K)."""
        distance = torch.sum((points.unsqueeze(2) - knn_points) ** 2, dim=-1).detach()
        # mean over points of the distance to the closest of the K neighbours
        h = torch.mean(torch.min(distance, dim=-1, keepdim=True)[0], dim=-2, keepdim=True)
        return torch.exp(-distance / (h / 2)).detach()

    def gen_grid(self, grid_size):
        """output [2, grid_size x grid_size] (reference :252-262)"""
        x = torch.linspace(-0.2, 0.2, grid_size, dtype=torch.float32)
        x, y = torch.meshgrid(x, x, indexing="ij")
        return torch.stack([x, y], dim=0).view([2, grid_size * grid_size])

    def gen_1d_grid(self, num_grid_point):
        """output [1, num_grid_point] (reference :264-270)"""
        return torch.linspace(-0.2, 0.2, num_grid_point).view(1, num_grid_point)

    # patches per launch group of forward_cl: bounds the (B,N,K,264) / (B,N*r,265) temporaries of the
    # skip connection and the regressor (25 GB / 10 GB for the 15 360 level-4 patches of 8 clouds)
    max_patches = int(os.environ.get("TPU3_MAX_PATCHES", "4096"))

    def _feat_dtype(self):
        return torch.float16 if getattr(self, "activation_storage", "f32") == "f16" else torch.float32

    def _fold_plan(self, blocks, widths, c0):
        """Per DenseEdgeConv block i = 0..2 the operands of HipBackend.dense_edge_conv_fold, or None when the layers
        are not the standard ones.  prep_j (j = i+1 .. 3) reads the concatenation [y_j-1 | ... | y_0 | x0]; block i's
        row [y_i (36) | x_i (24)] sits at columns (j-1-i) * 60 of it.  Block 0's input IS x0, whose second copy closes
        the concatenation: those 24 columns are added to the block's x part.  The first 24 outputs complete prep_i+1
        (bias, ReLU applied by the kernel), the others are partial sums kept in a (B,N,48) buffer."""
        preps = [p for _, p in blocks[1:]]
        if (getattr(self, "mlp_precision", "f32") != "f32" or any(w != 60 for w in widths) or c0 != 24
                or any(p.conv.out_channels != 24 or p.activation != "relu" or not p.pointwise()
                       for p in preps)):
            return None
        # (weights AND biases: block 0 bakes the three biases in; a bias-only in-place edit must rebuild too)
        key = tuple(t._version for p in preps for t in (p.conv.weight, p.conv.bias)) + \
            tuple(t.data_ptr() for p in preps for t in (p.conv.weight, p.conv.bias))
        cached = _FOLD_CACHES.get(self)
        on_device = preps[0].conv.weight.is_cuda
        here = torch.cuda.current_stream(preps[0].conv.weight.device) if on_device else None
        if cached is not None and cached[0] == key:
            # the plan was BUILT by kernels on one stream; pipeline.upsample runs sub-batches of the same net on
            # several streams: every other stream orders itself behind the build (advisor, round 3)
            built_on, done = cached[2], cached[3]
            if on_device and built_on != here.cuda_stream:
                here.wait_event(done)
                for e in cached[1]:
                    for t in (e["w"], e["b"]):
                        if t is not None:
                            t.record_stream(here)
            return cached[1]
        W = [p.conv.weight.detach().reshape(24, -1) for p in preps]            # (24, 84), (24, 144), (24, 204)
        bs = [p.conv.bias.detach() for p in preps]
        plan = []
        for i in range(3):
            rows = []
            for j in range(i, 3):                   # prep index j feeds block j + 1
                w = W[j][:, (j - i) * 60:(j - i) * 60 + 60].clone()
                if i == 0:
                    w[:, 36:60] += W[j][:, -24:]
                rows.append(w)
            entry = dict(w=torch.cat(rows, dim=0).contiguous(),
                         b=torch.cat(bs, dim=0).contiguous() if i == 0 else None,
                         seed_off=0 if i < 2 else 24, store_off=0 if i == 0 else 24)
            plan.append(entry)
        done = None
        if on_device:
            done = torch.cuda.Event()
            done.record(here)
        _FOLD_CACHES[self] = (key, plan, here.cuda_stream if on_device else None, done)
        return plan

    def forward_cl(self, xyz, xyz_normalized, previous=None, owner=None, groups=1, per_owner=0):
        """Channel-last level; large batches are processed in chunks of whole owner groups (every
        patch is independent apart from the per-group unique-max, so chunks are exact).
        per_owner > 0: owner is `per_owner` consecutive patches per id (scheduling hint)."""
        B = xyz_normalized.size(0)
        if B <= self.max_patches or torch.is_grad_enabled():
            return self._forward_cl(xyz, xyz_normalized, previous, owner, groups, per_owner)
        if owner is None:
            bounds = list(range(0, B, self.max_patches)) + [B]
        else:
            # owner ids are non-decreasing (patches of one cloud are contiguous): cut between owners
            # A chunk is always a whole number of owners: unique=True adds max(D) over one reference
            # call = one owner's patches (operations.py:204), so an owner must never straddle chunks.
            per = B // groups if groups > 0 and B % groups == 0 else None
            if per is None:
                bounds = [0, B]
                per_owner = 0
            else:
                owners = max(1, self.max_patches // per)
                if owners > 8:
                    owners -= owners % 8        # whole sets of 8 clouds: one per XCD (skip kernel)
                step = owners * per
                bounds = list(range(0, B, step)) + [B]
        # every chunk writes its features straight into its rows of one (B,N,264) buffer
        blocks = (self.layer1, self.layer2, self.layer3, self.layer4)
        total = self.layer0.conv.out_channels + sum(b.in_channels + b.n * b.growth_rate for b in blocks)
        feat_all = xyz_normalized.new_empty((B, xyz_normalized.size(1), total), dtype=self._feat_dtype())
        outs = []
        ucache = {} if hasattr(operations.BACKEND, "knn_graph") else None     # de-dup state of `previous`
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            own = None if owner is None else owner[lo:hi].contiguous()
            o, _ = self._forward_cl(xyz[lo:hi], xyz_normalized[lo:hi], previous, own, groups, per_owner,
                                    feat_buf=feat_all[lo:hi], unique_cache=ucache)
            outs.append(o)
        return torch.cat(outs, dim=0), feat_all

    def _forward_cl(self, xyz, xyz_normalized, previous=None, owner=None, groups=1, per_owner=0, feat_buf=None,
                    unique_cache=None):
        """Channel-last level:
            xyz, xyz_normalized  (B,N,3)
            previous             None or (prev_xyz (Bp,M,3), prev_feat (Bp,M,C), prev_count (Bp,)|None)
            owner                (B,) int32: which previous cloud / unique-group each patch
                                 belongs to (None: identity, one group -- the reference's call)
        -> xyz_up (B, N*r, 3) normalised coordinates, features (B,N,264)."""
        B, N, _ = xyz_normalized.shape
        graph_layout = None if owner is None else dict(grp=owner, groups=groups)

        # One (B,N,264) buffer holds the level's dense concatenation [y4 | y3 | y2 | y1 | x0]
        # (reference: four torch.cat, :293-311); every block writes its slice in place and the next
        # prep convolution reads the tail slice, so nothing is copied.
        blocks = ((self.layer1, None), (self.layer2, self.layer2_prep), (self.layer3, self.layer3_prep),
                  (self.layer4, self.layer4_prep))
        widths = [blk.in_channels + blk.n * blk.growth_rate for blk, _ in blocks]
        c0 = self.layer0.conv.out_channels
        total = c0 + sum(widths)
        if torch.is_grad_enabled():
            x0 = self.layer0.forward_cl(xyz_normalized)
            x = x0
            for blk, prep in blocks:
                y, _ = blk.forward_cl(x if prep is None else prep.forward_cl(x), layout=graph_layout)
                x = torch.cat([y, x], dim=-1)
        else:
            feat = feat_buf if feat_buf is not None else xyz_normalized.new_empty((B, N, total), dtype=self._feat_dtype())
            lo = total - c0
            if feat.dtype == torch.float32:
                x0 = self.layer0.forward_cl(xyz_normalized, also=feat[..., lo:])     # x0 and its slice in one pass
            else:
                x0 = self.layer0.forward_cl(xyz_normalized)
                feat[..., lo:].copy_(x0)                                             # (rounded to the stored type)
            plan = self._fold_plan(blocks, widths, c0) if self.fold_preps and x0.is_cuda else None
            acc = xyz_normalized.new_empty((B, N, 48)) if plan is not None else None
            inp, folded = x0, False
            for i, ((blk, prep), wdt) in enumerate(zip(blocks, widths)):
                if i > 0 and not folded:
                    inp = prep.forward_cl(feat[..., lo:])
                fold = None
                if plan is not None and i < len(plan):
                    # the later prep convolutions' share of this block's rows is added in the block's write-out
                    # (csrc/dense_edge_conv.hip): the buffer is not re-read, the next block's input arrives directly
                    fold = dict(plan[i], acc=acc, xnext=xyz_normalized.new_empty((B, N, c0)))
                blk.forward_cl(inp, layout=graph_layout, out=feat[..., lo - wdt:lo], fold=fold)
                folded = bool(fold is not None and fold.get("done"))
                if fold is not None and not folded:
                    plan = None                     # (shape beyond the kernel: the remaining layers run unfolded)
                if folded:
                    inp = fold["xnext"]
                lo -= wdt
            x = feat

        # interlevel skip connection (:317-347)
        if previous is not None and self.fm_knn > 0:
            prev_xyz, prev_feat, prev_count = previous
            layout = None
            if owner is not None or prev_count is not None or prev_xyz.size(0) != B:
                if owner is None:
                    if prev_xyz.size(0) == B:
                        owner_ = torch.arange(B, dtype=torch.int32, device=xyz.device)
                    else:   # the reference's expand of ONE previous cloud over the batch (:319-323)
                        assert prev_xyz.size(0) == 1
                        owner_ = torch.zeros(B, dtype=torch.int32, device=xyz.device)
                    layout = dict(pts_of=owner_, n_arr=prev_count)
                else:
                    layout = dict(pts_of=owner, n_arr=prev_count, grp=owner, groups=groups)
                pts_of = layout["pts_of"]             # (converted to int64 only where the unfused path indexes with it)
            else:
                pts_of = None
            covered = x.is_cuda and x.is_contiguous() and self.fm_knn <= 8 and x.size(-1) <= 320
            fused = not torch.is_grad_enabled() and hasattr(operations.BACKEND, "interlevel_skip") and covered
            fused_train = (torch.is_grad_enabled() and hasattr(operations.BACKEND, "interlevel_skip_train")
                           and covered and x.dtype == torch.float32)
            if not fused and not torch.is_grad_enabled():
                operations.note_generic_path("inter-level skip with fm_knn=%d, %d channels (fused kernel: "
                                             "fm_knn <= 8, <= 320 channels)" % (self.fm_knn, x.size(-1)))
            with torch.no_grad():
                knn_idx, _, knn_points = operations.knn_query(
                    self.fm_knn, xyz.detach(), prev_xyz.detach(), unique=True, layout=layout,
                    want_dist=False, want_grouped=not (fused or fused_train), unique_cache=unique_cache)
            if fused_train:
                # one autograd node (csrc/skip.hip): the weights are constants of the step (the reference detaches
                # both distances, :244-245), so x gets g and the previous features a weighted scatter of g
                x = _SkipTrain.apply(x, prev_feat.contiguous(), xyz.detach().contiguous(),
                                     prev_xyz.detach().contiguous(),
                                     None if pts_of is None else layout["pts_of"], knn_idx)
                return self._regress(x, xyz_normalized, B, N)
            if fused:
                operations.BACKEND.interlevel_skip(
                    xyz.contiguous(), x, prev_xyz.contiguous(), prev_feat.contiguous(),
                    None if pts_of is None else layout["pts_of"], knn_idx, per_cloud=per_owner)
                return self._regress(x, xyz_normalized, B, N)
            bsel = (torch.arange(B, device=xyz.device) if pts_of is None else pts_of.long()).view(-1, 1, 1)
            knn_feats = prev_feat[bsel, knn_idx]                              # (B,N,K,C)
            s_weight = self.exponential_distance_cl(xyz, knn_points)
            f_weight = self.exponential_distance_cl(x, knn_feats)
            weight = s_weight * f_weight
            weight = weight / torch.sum(weight + 1e-5, dim=-1, keepdim=True)
            x = 0.2 * torch.sum(weight.unsqueeze(-1) * knn_feats, dim=2) + x

        return self._regress(x, xyz_normalized, B, N)

    def _code_term(self, code, w, cin):
        """W_c code_j of the regressor's first layer for a 1-d code: (r,1) * (1,128), a constant of the weights --
        cached per Level like the fold plan (weight version + address; built on one stream, other streams order
        themselves behind the build), one launch per set of weights instead of one per Level call."""
        if not code.is_cuda:
            return code[0].t() * w[:, cin:].t()
        key = (w._version, w.data_ptr(), code.data_ptr(), code.dtype, cin)
        here = torch.cuda.current_stream(code.device)
        hit = _CODE_CACHES.get(self)
        if hit is not None and hit[0] == key:
            if hit[2] != here.cuda_stream:
                here.wait_event(hit[3])
                hit[1].record_stream(here)
            return hit[1]
        with torch.no_grad():
            c = (code[0].t() * w[:, cin:].t()).contiguous()
        done = torch.cuda.Event()
        done.record(here)
        _CODE_CACHES[self] = (key, c, here.cuda_stream, done)
        return c

    def _regress(self, x, xyz_normalized, B, N):
        point_features = x
        # feature expansion: every point r times, followed by its 1-d / 2-d code (:350-361)
        _, code_length, ratio = self.code.size()
        # (fp16 feature buffers: only the rows are fp16, everything computed from them is fp32)
        code = self.code.to(device=x.device, dtype=torch.float32 if x.dtype == torch.float16 else x.dtype)   # (1,L,r)
        up1 = self.up_layer.up_layer1
        if not torch.is_grad_enabled() and up1.pointwise() and up1.activation == "relu":
            # inference: W [x_i ; code_j] = W_x x_i + W_c code_j -- the 264-channel part is the same
            # for the r replicas of a point, so it is computed once per point and the (B, N*r, 265)
            # concatenation of the reference is never built (half the FLOPs of this layer)
            w = up1.conv.weight.view(up1.conv.weight.size(0), -1)
            cin = x.size(-1)
            f16 = getattr(self, "mlp_precision", "f32") == "f16"
            if f16:     # the per-point half of up_layer1 on fp16-operand MFMA (csrc/mlp.hip, linear_wide_f16_kernel)
                a = None
                if x.is_cuda and hasattr(operations.BACKEND, "linear_small"):
                    a = operations.BACKEND.linear_small(x, w[:, :cin], up1.conv.bias, False, mfma=operations.L.MFMA_F16)
                if a is None:
                    raise RuntimeError("mlp_precision='f16': up_layer1 with %d inputs / %d outputs is not covered by "
                                       "the fp16-operand kernel" % (cin, w.size(0)))
            else:
                a = None
                if x.is_cuda and hasattr(operations.BACKEND, "linear_wide"):
                    a = operations.BACKEND.linear_wide(x, w[:, :cin], up1.conv.bias)    # (B,N,128), csrc/mlp.hip
                if a is None:
                    a = torch.nn.functional.linear(x, w[:, :cin], up1.conv.bias)
            if code_length == 1:
                # a 1-d code: W_c code_j is ONE product per entry -- an outer product, bit for bit what the (r,1) x
                # (1,128) GEMM gives, without a vendor GEMM launch per Level call on the inference path
                c = self._code_term(code, w, cin)                                         # (r,1) * (1,128)
            else:
                c = torch.nn.functional.linear(code[0].t().contiguous(), w[:, cin:])      # (r,128)
            up2, fc1, fc2 = self.up_layer.up_layer2, self.fc_layer1, self.fc_layer2
            be = operations.BACKEND
            if (hasattr(be, "regress_tail") and x.is_cuda and ratio <= 4 and a.size(-1) == 128
                    and all(l.pointwise() for l in (up2, fc1, fc2))
                    and (up2.activation, fc1.activation, fc2.activation) == ("relu", "relu", None)
                    and (up2.conv.out_channels, fc1.conv.out_channels, fc2.conv.out_channels) == (128, 64, 3)):
                # relu(a_i + c_j) -> 128 -> 64 -> 3 + residual, register to register (csrc/mlp.hip)
                flat = lambda l: l.conv.weight.view(l.conv.weight.size(0), -1)
                out = be.regress_tail(a.reshape(B * N, 128), c, flat(up2), up2.conv.bias, flat(fc1),
                                      fc1.conv.bias, flat(fc2), fc2.conv.bias,
                                      xyz_normalized.reshape(B * N, 3),
                                      mfma=operations.L.MFMA_F16 if f16 else operations.L.MFMA_F32)
                return out.view(B, N * ratio, 3), point_features
            if f16:
                raise RuntimeError("mlp_precision='f16': the fused regressor tail does not cover step ratio %d" % ratio)
            operations.note_generic_path("regressor tail with step ratio %d / widths %s (fused kernel: ratio <= 4, "
                                         "128 -> 128 -> 64 -> 3)" % (ratio, (a.size(-1), up2.conv.out_channels,
                                                                            fc1.conv.out_channels, fc2.conv.out_channels)))
            x = torch.relu_(a.unsqueeze(2) + c.view(1, 1, ratio, -1)).reshape(B, N * ratio, -1)
        elif (torch.is_grad_enabled() and x.is_cuda and up1.pointwise() and up1.activation == "relu"
              and up1.conv.bias is not None):
            # training on a device: the same split of up_layer1 -- the per-point half once per point (half the rows
            # in its forward, input- and weight-gradient products), the code half once per replica, and the
            # (B, N*r, 265) concatenation with its backward never exists
            w = up1.conv.weight.view(up1.conv.weight.size(0), -1)
            cin = x.size(-1)
            a = layers.pointwise_linear_train(x, w[:, :cin], up1.conv.bias)                     # (B,N,128)
            c = torch.nn.functional.linear(code[0].t().contiguous(), w[:, cin:])                # (r,128)
            x = torch.relu_(a.unsqueeze(2) + c.view(1, 1, ratio, -1)).reshape(B, N * ratio, -1)
        else:
            code = code.permute(0, 2, 1).reshape(1, 1, ratio, code_length).expand(B, N, -1, -1)
            x = torch.cat([x.unsqueeze(2).expand(-1, -1, ratio, -1), code], dim=-1)
            x = x.reshape(B, N * ratio, x.size(-1))
            # coordinate regression (:363-369) + residual (:371-372)
            x = up1.forward_cl(x)
        x = self.up_layer.up_layer2.forward_cl(x)
        x = self.fc_layer1.forward_cl(x)
        x = self.fc_layer2.forward_cl(x)
        x = x + xyz_normalized.unsqueeze(2).expand(-1, -1, ratio, -1).reshape(B, N * ratio, 3)
        return x, point_features

    def forward(self, xyz, xyz_normalized, previous_level4=None, **kwargs):
        """
        :param
            xyz             Bx3xN input xyz, unnormalized
            xyz_normalized  Bx3xN input xyz, normalized
            previous_level4 tuple of the xyz and feature of the final feature
                            in the previous level (Bx3xM, BxCxM)
        :return
            xyz             Bx3xNr output xyz, normalized
            l4_features     BxCxN feature of the input points
        """
        previous = None
        if previous_level4 is not None:
            pxyz, pfeat = previous_level4
            previous = (pxyz.transpose(2, 1).contiguous(), pfeat.transpose(2, 1).contiguous(), None)
        x, feat = self.forward_cl(xyz.transpose(2, 1).contiguous(),
                                  xyz_normalized.transpose(2, 1).contiguous(), previous)
        return x.transpose(2, 1).contiguous(), feat.transpose(2, 1).contiguous()


class AdaptiveLevel(Level):
    """Upsampling unit with a free target point number (reference :377-512).  Nothing in the reference
    instantiates it; it is mirrored for completeness of the module API: layer1 is a DenseEdgeConv on
    all points, layers 2-4 are SampledDenseEdgeConvs on 48 / 16 / 1 sampled points, each followed by
    an interpolation of the previous features onto the sampled points; the single global feature is
    expanded over a round(sqrt(target))^2 grid code and regressed to coordinates.

    As in the reference, layer4 asks for knn + 1 neighbours among the 16 points layer3 kept, so the
    module only runs for knn <= 15 (the reference's torch.topk raises for more; so does the kNN here).
    FPS, gathers and every kNN run on the HIP kernels through `operations`; the small dense layers
    are ordinary convolutions."""

    def __init__(self, dense_n=3, growth_rate=12, knn=16, fm_knn=5):
        super(Level, self).__init__()
        self.dense_n = dense_n
        self.fm_knn = fm_knn
        self.layer0 = layers.Conv2d(3, 24, [1, 1], activation=None)
        self.layer1 = layers.DenseEdgeConv(24, growth_rate=growth_rate, n=dense_n, k=knn)
        in_channels = 84
        self.layer2_prep = layers.Conv1d(in_channels, 24, 1, activation="relu")
        self.layer2 = layers.SampledDenseEdgeConv(24, growth_rate=growth_rate, n=dense_n, k=knn)
        in_channels = 144
        self.layer3_prep = layers.Conv1d(in_channels, 24, 1, activation="relu")
        self.layer3 = layers.SampledDenseEdgeConv(24, growth_rate=growth_rate, n=dense_n, k=knn)
        in_channels = 204
        self.layer4_prep = layers.Conv1d(in_channels, 24, 1, activation="relu")
        self.layer4 = layers.SampledDenseEdgeConv(24, growth_rate=growth_rate, n=dense_n, k=knn)
        in_channels = 264
        self.up_layer = torch.nn.Sequential(OrderedDict([
            ("up_layer1", layers.Conv2d(in_channels + 2, 128, 1, activation="relu")),
            ("up_layer2", layers.Conv2d(128, 128, 1, activation="relu")), ]))
        self.fc_layer1 = layers.Conv2d(128, 64, 1, activation="relu")
        self.fc_layer2 = layers.Conv2d(64, 3, 1, activation=None)

    def exponential_distance(self, points, knnIdx_points):
        """points (B,C,N[,1]), knnIdx_points (B,C,N,K) -> distance, weight (B,1,N,K)   (:409-427;
        unlike Level's, the bandwidth carries + 1e-5)"""
        if points.dim() == 3:
            points = points.unsqueeze(dim=-1)
        distance = torch.sum((points - knnIdx_points) ** 2, dim=1, keepdim=True).detach()
        h = torch.mean(torch.min(distance, dim=-1, keepdim=True)[0], dim=-2, keepdim=True) + 1e-5
        weight = torch.exp(-distance / (h / 2)).detach()
        return distance, weight

    def gen_grid(self, grid_size):
        """output [2, grid_size x grid_size] over [-1, 1]^2   (:429-439)"""
        x = torch.linspace(-1.0, 1.0, grid_size, dtype=torch.float32)
        x, y = torch.meshgrid(x, x, indexing="ij")
        return torch.stack([x, y], dim=0).view([2, grid_size * grid_size])

    def interpolate(self, previous_xyz, xyz, previous_feat):
        """previous_feat (B,C,M) at previous_xyz (B,3,M) -> (B,C,N') at xyz (B,3,N'): weighted mean over
        the fm_knn nearest previous points (:441-465).  The reference repeats previous_feat N' times
        before its gather; here the neighbours' rows are gathered directly."""
        knn_points, knn_idx, _ = operations.group_knn(self.fm_knn, xyz, previous_xyz, unique=True, NCHW=True)
        B, C, _ = previous_feat.shape
        Np, K = knn_idx.size(1), knn_idx.size(2)
        feats = torch.gather(previous_feat, 2, knn_idx.reshape(B, 1, Np * K).expand(-1, C, -1)).view(B, C, Np, K)
        _, weight = self.exponential_distance(xyz, knn_points)
        weight = weight / torch.sum(weight + 1e-5, dim=-1, keepdim=True)
        return torch.sum(weight * feats, dim=-1)

    def forward(self, xyz, target_n_point):
        """xyz (B,3,N) -> xyz (B,3,round(sqrt(target_n_point))^2), global feature (B,264,1)   (:467-512)"""
        code = self.gen_grid(round(sqrt(target_n_point))).to(device=xyz.device)
        code = code.expand(xyz.size(0), -1, -1)
        xyz_normalized, centroid, radius = operations.normalize_point_batch(xyz, NCHW=True)
        x = self.layer0(xyz_normalized.unsqueeze(dim=-1)).squeeze(dim=-1)
        y, _ = self.layer1(x)
        x = torch.cat([y, x], dim=1)
        sampled_xyz = xyz_normalized
        for prep, layer, nsample in ((self.layer2_prep, self.layer2, 48), (self.layer3_prep, self.layer3, 16),
                                     (self.layer4_prep, self.layer4, 1)):
            y, new_xyz, _ = layer(prep(x), nsample, sampled_xyz)
            x = torch.cat([y, self.interpolate(sampled_xyz, new_xyz, x)], dim=1)
            sampled_xyz = new_xyz
        global_features = x
        x = x.expand(-1, -1, code.size(-1))
        x = torch.cat([x, code], dim=1).unsqueeze(-1)
        x = self.fc_layer2(self.fc_layer1(self.up_layer(x))).squeeze(-1)
        x = (x * radius.detach()) + centroid.detach()
        return x, global_features
