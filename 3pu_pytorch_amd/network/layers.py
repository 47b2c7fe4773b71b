"""Per-patch feature stacks of 3PU for PyTorch-ROCm -- counterpart of the reference's
network/layers.py (DenseEdgeConv :6-64, Conv2d :115-158, Conv1d :161-204).

Same classes, constructor arguments, parameter names and shapes (so reference checkpoints load:
`layerK.mlps.{0,1,2}.{weight,bias}`, `*.conv.{weight,bias}`), different execution:

  * activations are kept channel-last, (B, N, C) / (B, N, k, C): every 1x1 convolution of the
    reference is then one dense row-major GEMM [B*N(*k), C_in] x [C_in, C_out] that rocBLAS /
    hipBLASLt maps onto MFMA, instead of an NCHW convolution with 12 output channels;
  * the k-NN graph comes from the fused HIP kernel (indices + gathered neighbours in one pass,
    no (B,N,N) distance matrix, no host round trip for unique=True);
  * the dense concatenations write into one pre-allocated (B,N,k,C_total) buffer slice by slice
    instead of re-copying the growing tensor at every torch.cat.

The NCHW `forward` keeps the reference's calling convention; `forward_cl` is the channel-last
entry the Level uses.
"""
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import operations
from .. import _lib as L


class _SkinnyLinear(torch.autograd.Function):
    """y = x W^T + b for few outputs over very many rows (the dense layers of DenseEdgeConv in
    training: B*N*k rows, 12 outputs).  Forward and dX are ordinary GEMMs; dW = dy^T x is a 48 x 12
    reduction over ~3e5 rows that the vendor GEMM runs at 0.6 TFLOP/s -- it goes through the
    streaming MFMA kernel tpu3_linear_wgrad_f32 (deterministic)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gy2 = gy.reshape(-1, gy.size(-1))
        gx = gy.matmul(weight) if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1]:
            gw = operations.BACKEND.linear_wgrad(x.reshape(-1, x.size(-1)), gy2)
            if gw is None:
                gw = gy2.t().matmul(x.reshape(-1, x.size(-1)))
        if ctx.needs_input_grad[2]:
            gb = gy2.sum(dim=0)
        return gx, gw, gb


class _PointwiseLayer(torch.autograd.Function):
    """y = act(x W^T + b), act = identity | ReLU, for the per-point layers of a Level in training (lift, prep
    convolutions, regressor).  Forward and dX are vendor GEMMs; dW and db come from ONE streaming pass over
    (x, dy) (tpu3_linear_wgrad_bias_f32, deterministic) instead of autograd's 65 .. 150 us skinny GEMM plus a
    column reduction, and the ReLU mask is applied to dy once for all three."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        y = F.linear(x, weight, bias)
        if relu:
            y = torch.relu_(y)
        ctx.relu = relu
        ctx.save_for_backward(x, weight, y if relu else weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        g = gy.contiguous()
        if ctx.relu:
            g = torch.ops.aten.threshold_backward(g, y, 0)
        g2 = g.reshape(-1, g.size(-1))
        gx = None
        if ctx.needs_input_grad[0]:
            if weight.size(0) <= 32 and hasattr(operations.BACKEND, "linear_dgrad"):
                gx = operations.BACKEND.linear_dgrad(g2, weight.contiguous())
                gx = gx.view(x.shape) if gx is not None else None
            if gx is None:
                gx = g.matmul(weight)
        gw = gb = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            res = operations.BACKEND.linear_wgrad_bias(x.reshape(-1, x.size(-1)), g2)
            if res is None:
                gw, gb = g2.t().matmul(x.reshape(-1, x.size(-1))), g2.sum(dim=0)
            else:
                gw, gb = res
        return gx, gw, gb, None


def pointwise_linear_train(x, weight, bias):
    """y = x W^T + b through _PointwiseLayer when the device kernels apply (weight may be a column slice of a
    parameter), else plain F.linear."""
    if (x.is_cuda and x.dtype == torch.float32 and x.numel() // x.size(-1) >= 1024
            and hasattr(operations.BACKEND, "linear_wgrad_bias")):
        return _PointwiseLayer.apply(x, weight, bias, False)
    return F.linear(x, weight, bias)


def pointwise_train(layer, x):
    """Training shortcut of a pointwise Conv1d / Conv2d (activation None / ReLU, bias, no normalisation) on a
    device: one autograd node, see _PointwiseLayer.  None = not applicable."""
    conv = layer.conv
    if (not torch.is_grad_enabled() or not x.is_cuda or layer.activation not in (None, "relu") or conv.bias is None
            or not conv.weight.requires_grad or x.dtype != torch.float32 or x.numel() // x.size(-1) < 1024
            or not hasattr(operations.BACKEND, "linear_wgrad_bias")):
        return None
    w = conv.weight
    return _PointwiseLayer.apply(x, w.view(w.size(0), w.size(1)), conv.bias, layer.activation == "relu")


class _GatherRows(torch.autograd.Function):
    """knn_point = x[b, idx] (group_knn's differentiable gather, reference operations.py:209-211) with the backward
    as ONE atomic scatter-add launch: torch's index backward sorts the 3e5 indices of every DenseEdgeConv block
    first (nine rocPRIM merge passes + the accumulation kernel per call)."""

    @staticmethod
    def forward(ctx, x, idx):
        out = operations.BACKEND.gather_rows(x, idx)
        ctx.save_for_backward(idx)
        ctx.n = x.size(1)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return operations.BACKEND.scatter_add_rows(g, idx, ctx.n), None


def gather_neighbours(x, idx):
    """x (B,N,C), idx (B,N,k) -> (B,N,k,C), differentiable in x."""
    be = operations.BACKEND
    if (x.is_cuda and hasattr(be, "gather_rows") and x.dtype == torch.float32 and x.size(-1) % 4 == 0
            and x.is_contiguous() and idx.is_contiguous() and (x.data_ptr() & 15) == 0):
        return _GatherRows.apply(x, idx)
    b = torch.arange(x.size(0), device=x.device).view(-1, 1, 1)
    return x[b, idx]


class _FusedDECTrain(torch.autograd.Function):
    """The (24, 12, 3, k = 32) DenseEdgeConv block as ONE autograd node (csrc/dec_train.hip): forward and backward
    are one launch each; the backward kernel accumulates the weight gradients of the edge parts on the matrix cores (one
    block per workgroup) and leaves the per-point sums S; tpu3_dec_train_wgrad_f32 adds a streaming pass S^T [x | 1] and
    one kernel that sums the blocks and writes the layers' own layouts."""

    @staticmethod
    def forward(ctx, x, idx, idx_off, w0, b0, w1, b1, w2, b2):
        weights = tuple(t.detach().contiguous().view(t.size(0), -1) if t.dim() > 1 else t.detach().contiguous()
                        for t in (w0, b0, w1, b1, w2, b2))
        x = x.contiguous()
        y, arg = operations.BACKEND.dec_train_forward(x, idx, idx_off, weights)
        ctx.save_for_backward(x, idx, arg, *weights)
        ctx.idx_off = idx_off
        ctx.shapes = (w0.shape, w1.shape, w2.shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, idx, arg = ctx.saved_tensors[:3]
        weights = ctx.saved_tensors[3:]
        be = operations.BACKEND
        gx, S, ws = be.dec_train_backward(x, idx, ctx.idx_off, weights, arg, gy)
        gw0, gw1, gw2, gb = be.dec_train_wgrad(x.view(-1, x.size(-1)), S, ws)
        return (gx, None, None, gw0.view(ctx.shapes[0]), gb[24:36], gw1.view(ctx.shapes[1]), gb[12:24],
                gw2.view(ctx.shapes[2]), gb[0:12])


def linear_1x1(conv, x):
    """Apply a kernel-size-1 nn.Conv1d / nn.Conv2d to channel-last activations (..., C_in)."""
    w = conv.weight
    w2 = w.view(w.size(0), w.size(1))
    if (torch.is_grad_enabled() and w.requires_grad and x.is_cuda and w.size(0) <= 16 and w.size(1) <= 64
            and x.numel() // x.size(-1) >= 16384 and x.is_contiguous() and conv.bias is not None
            and hasattr(operations.BACKEND, "linear_wgrad")):
        return _SkinnyLinear.apply(x, w2, conv.bias)
    return F.linear(x, w2, conv.bias)


# packed operand blobs of the fused DenseEdgeConv kernel per module (layers.DenseEdgeConv._operand_pack); held outside
# the modules so that copy.deepcopy(net) / torch.save(net) never meet a stream event
_PACK_CACHES = weakref.WeakKeyDictionary()


class DenseEdgeConv(nn.Module):
    """Dense edge convolution block (reference layers.py:6-64): feature-space kNN graph, edge
    feature [x_i, x_j - x_i], `n` 1x1 convolutions with dense concatenation, max over k."""

    def __init__(self, in_channels, growth_rate, n, k, **kwargs):
        super(DenseEdgeConv, self).__init__()
        self.growth_rate = growth_rate
        self.n = n
        self.k = k
        self.in_channels = in_channels
        self.mlps = torch.nn.ModuleList()
        self.mlps.append(torch.nn.Conv2d(2 * in_channels, growth_rate, 1, bias=True))
        for i in range(1, n):
            in_channels += growth_rate
            self.mlps.append(torch.nn.Conv2d(in_channels, growth_rate, 1, bias=True))

    def get_local_graph_cl(self, x, k, idx=None, layout=None):
        """x (B,N,C) -> edge feature (B,N,k,2C) = [x_i, x_j - x_i], idx (B,N,k).
        The first of the k+1 neighbours is dropped as in the reference (:33-35)."""
        need_grad = x.requires_grad and torch.is_grad_enabled()
        if idx is None:
            full = None
            if need_grad and x.is_cuda and x.dtype == torch.float32 and hasattr(operations.BACKEND, "knn_graph"):
                # training: only the neighbour SET matters (the nearest is dropped, the rest is max-pooled),
                # so the graph kernel serves here too; None = configuration it does not cover
                with torch.no_grad():      # (exact form: nobody inspects the optimistic events in a training loop)
                    full = operations.BACKEND.knn_graph(k + 1, x.detach().contiguous(), layout, optimistic=False)
            if full is not None:
                idx, knn_point = full.long(), None
            else:
                with torch.no_grad():
                    idx, _, knn_point = operations.knn_query(
                        k + 1, x.detach(), x.detach(), unique=True, layout=layout,
                        want_dist=False, want_grouped=not need_grad)
            idx = idx[:, :, 1:]
            if knn_point is not None and not need_grad:
                knn_point = knn_point[:, :, 1:, :]
        else:
            knn_point = None
        if knn_point is None or need_grad:
            knn_point = gather_neighbours(x, idx.contiguous())       # differentiable gather (B,N,k,C)
        center = x.unsqueeze(2).expand_as(knn_point)
        return torch.cat([center, knn_point - center], dim=-1), idx

    # arithmetic of the fused block's matrix instructions: "f32" (default: exact fp32 chains) or "f16"
    # (fp16 operands, fp32 accumulate -- config C5); set through Net.set_mlp_precision
    mlp_precision = "f32"

    def fused_reason(self, x):
        """None when the hand-written MFMA kernel covers this call, else why it does not."""
        if not hasattr(operations.BACKEND, "dense_edge_conv"):
            return "stand-in backend"
        if torch.is_grad_enabled():
            return "training (autograd formulation)"
        if not x.is_cuda or x.dtype != torch.float32:
            return "DenseEdgeConv input is %s on %s" % (x.dtype, x.device.type)
        if (self.in_channels, self.growth_rate, self.n) != (24, 12, 3):
            return "DenseEdgeConv(in_channels=%d, growth_rate=%d, n=%d): the fused kernel covers (24, 12, 3)" % (
                self.in_channels, self.growth_rate, self.n)
        if self.k % 16 or self.k > 64:
            return "DenseEdgeConv k=%d: the fused kernel covers k in {16, 32, 48, 64}" % self.k
        return None

    # training: evaluate everything that depends on ONE point per point (see _forward_train_hoisted)
    hoist_train = True
    # training on the device for the reference's shape: one launch per direction (csrc/dec_train.hip)
    fused_train = True

    def _fused_train_ok(self, x):
        return (hasattr(operations.BACKEND, "dec_train_forward") and x.is_cuda and x.dtype == torch.float32
                and (self.in_channels, self.growth_rate, self.n, self.k) == (24, 12, 3, 32)
                and all(m.bias is not None for m in self.mlps))

    def _forward_train_fused(self, x, idx=None, layout=None):
        k = self.k
        if idx is None:
            full = None
            if hasattr(operations.BACKEND, "knn_graph"):
                with torch.no_grad():
                    full = operations.BACKEND.knn_graph(k + 1, x.detach().contiguous(), layout, optimistic=False)
            if full is None:
                with torch.no_grad():
                    full, _, _ = operations.knn_query(k + 1, x.detach(), x.detach(), unique=True, layout=layout,
                                                      want_dist=False, want_grouped=False)
                full = full.to(torch.int32)
            # the kernels read the int32 buffer; the idx handed back is int64 like the sibling training paths' (and the
            # reference's), so that it can be fed back through `idx=` / torch.gather (advisor, r4)
            idx32, off, idx = full.contiguous(), 1, full[:, :, 1:].long()
        else:
            idx32, off = idx.to(torch.int32).contiguous(), 0
        m = self.mlps
        y = _FusedDECTrain.apply(x, idx32, off, m[0].weight, m[0].bias, m[1].weight, m[1].bias, m[2].weight, m[2].bias)
        return y, idx

    def _forward_train_hoisted(self, x, idx=None, layout=None):
        """The block of reference :44-64 for autograd, with the same hoisting as the fused inference kernel: a
        layer's input is [h_{i-1}, ..., h_0, x_i] (layer 0: [x_i, x_j - x_i]), so
            W_0 [x_i, x_j - x_i] = (W_0a - W_0b) x_i + W_0b x_j        W_i [h.., x_i] = W_i^h [h..] + W_i^x x_i
        and every x_i term (and z_j = W_0b x_j) is ONE (B*N, C) x (C, (n+1) g) product per block instead of
        C-wide columns of the B*N*k edge tensors: the edge tensors are g .. (n-1) g channels wide instead of
        2C .. C + (n-1) g, nothing of width C is concatenated per edge, and max over k commutes with the
        channel concatenation.  Same mathematics, different summation order (fp32 reassociation)."""
        B, N, C = x.shape
        g, n, k = self.growth_rate, self.n, self.k
        if idx is None:
            _, idx = self.get_local_graph_idx(x, k, layout)
        w = [m.weight.view(m.weight.size(0), -1) for m in self.mlps]
        cols = [w[0][:, :C] - w[0][:, C:], w[0][:, C:]] + [w[i][:, i * g:] for i in range(1, n)]
        bias = [self.mlps[0].bias, torch.zeros_like(self.mlps[0].bias)] + [self.mlps[i].bias for i in range(1, n)]
        P = F.linear(x, torch.cat(cols, dim=0), torch.cat(bias, dim=0))          # (B,N,(n+1) g)
        zg = gather_neighbours(P[..., g:2 * g].contiguous(), idx.contiguous())    # (B,N,k,g)
        hs = [F.relu(P[..., :g].unsqueeze(2) + zg)]                               # layer 0: ReLU (:57)
        for i in range(1, n):
            inp = hs[0] if i == 1 else torch.cat(hs, dim=-1)                      # newest first, like the reference
            wh = w[i][:, :i * g]
            if (wh.size(0) <= 16 and wh.size(1) <= 64 and inp.is_cuda and inp.numel() // inp.size(-1) >= 16384
                    and hasattr(operations.BACKEND, "linear_wgrad")):
                h = _SkinnyLinear.apply(inp.contiguous(), wh.contiguous(), None)
            else:
                h = F.linear(inp, wh)
            h = h + P[..., (i + 1) * g:(i + 2) * g].unsqueeze(2)
            hs.insert(0, h if i == n - 1 else F.relu(h))                          # last layer: no ReLU (:59)
        y = torch.cat([torch.max(h, dim=2)[0] for h in hs] + [x], dim=-1)
        return y, idx

    def get_local_graph_idx(self, x, k, layout=None):
        """Neighbour indices only (the first of the k+1 dropped, reference :33-35): (None, idx (B,N,k))."""
        full = None
        if x.is_cuda and x.dtype == torch.float32 and hasattr(operations.BACKEND, "knn_graph"):
            with torch.no_grad():
                full = operations.BACKEND.knn_graph(k + 1, x.detach().contiguous(), layout, optimistic=False)
        if full is None:
            with torch.no_grad():
                full, _, _ = operations.knn_query(k + 1, x.detach(), x.detach(), unique=True, layout=layout,
                                                  want_dist=False, want_grouped=False)
        return None, full.long()[:, :, 1:]

    @property
    def _pack_cache(self):
        """{fold_n: (key, blob, stream, event)} of this block's packed operands (see _operand_pack)."""
        return _PACK_CACHES.get(self, {})

    def _operand_pack(self, fold_w=None):
        """The fused fp32 kernel's operand tables for this block's CURRENT weights (and for `fold_w`, the folded prep
        convolutions' columns), HipBackend.dense_edge_conv_pack, cached until a weight changes (version counters and
        addresses, as Level._fold_plan -- an edit through `param.data` bumps neither: Net.invalidate_weight_caches()).  The blob is built by a launch on one stream; a call on another stream orders
        itself behind that launch.  None when the backend has no packed launch."""
        if not hasattr(operations.BACKEND, "dense_edge_conv_pack"):
            return None
        ps = [t for conv in self.mlps for t in (conv.weight, conv.bias)]
        if not ps[0].is_cuda:
            return None
        key = tuple(t._version for t in ps) + tuple(t.data_ptr() for t in ps) + \
            ((fold_w.data_ptr(), fold_w._version, fold_w.size(0)) if fold_w is not None else (0, 0, 0))
        cache = _PACK_CACHES.setdefault(self, {})      # (not an attribute: events and blobs must not be pickled / deep-copied)
        slot = 0 if fold_w is None else fold_w.size(0)
        here = torch.cuda.current_stream(ps[0].device)
        hit = cache.get(slot)
        if hit is not None and hit[0] == key:
            if hit[2] != here.cuda_stream:
                here.wait_event(hit[3])
                hit[1].record_stream(here)
            return hit[1]
        with torch.no_grad():
            blob = operations.BACKEND.dense_edge_conv_pack(self.mlps, fold_w)
        done = torch.cuda.Event()
        done.record(here)
        cache[slot] = (key, blob, here.cuda_stream, done)
        return blob

    def forward_cl(self, x, idx=None, layout=None, out=None, fold=None):
        """x (B,N,C) channel-last -> y (B,N,C + n*growth_rate), idx (B,N,k).
        `out`: optional (B,N,C + n*growth_rate) view (unit channel stride) that receives y -- the
        Level passes a slice of its concatenated feature buffer so that no copy is needed.
        `fold` (inference, fused fp32 kernel only): dict(w (F,60), b (F,)|None, acc (B,N,S)|None, seed_off,
        store_off, xnext (B,N,24)) -- the later prep convolutions' share of this block's rows is added while the
        rows are in registers (HipBackend.dense_edge_conv_fold); fold["done"] is set when that happened, the
        caller runs the prep convolution itself otherwise."""
        B, N, C = x.shape
        g, n, k = self.growth_rate, self.n, self.k
        why = self.fused_reason(x) if idx is None else "caller-supplied neighbour indices"
        if why is None:
            # kNN graph + gather + 3 dense layers + max in two launches, no (B,N,k,C) tensors
            x = x.contiguous()
            full_idx = operations.BACKEND.knn_graph(k + 1, x, layout) \
                if hasattr(operations.BACKEND, "knn_graph") else None
            if full_idx is None:
                full_idx, _, _ = operations.knn_query(k + 1, x, x, unique=True, layout=layout,
                                                      want_dist=False, want_grouped=False)
            if out is None:
                out = x.new_empty((B, N, C + n * g))
            packable = self.mlp_precision == "f32" and (C, g, n) == (24, 12, 3) and out.dtype == torch.float32
            if (fold is not None and self.mlp_precision == "f32" and (C, g, n) == (24, 12, 3)
                    and hasattr(operations.BACKEND, "dense_edge_conv_fold")):
                fold["done"] = operations.BACKEND.dense_edge_conv_fold(
                    x, full_idx, 1, k, self.mlps, out, fold["w"], fold.get("b"), fold.get("acc"),
                    fold.get("seed_off", 0), fold.get("store_off", 0), fold["xnext"],
                    pack=self._operand_pack(fold["w"]) if packable else None)
                if fold["done"]:
                    return out, full_idx[:, :, 1:]
            operations.BACKEND.dense_edge_conv(x, full_idx, 1, k, self.mlps, out,
                                               mfma=L.MFMA_F16 if self.mlp_precision == "f16" else L.MFMA_F32,
                                               pack=self._operand_pack() if packable else None)
            return out, full_idx[:, :, 1:]
        if self.mlp_precision != "f32":
            raise RuntimeError("mlp_precision=%r needs the fused DenseEdgeConv kernel, which does not cover this "
                               "call: %s" % (self.mlp_precision, why))
        if not torch.is_grad_enabled():
            operations.note_generic_path(why)
        if torch.is_grad_enabled() and self.fused_train and self._fused_train_ok(x):
            y, idx = self._forward_train_fused(x, idx, layout)
            if out is not None:
                out.copy_(y)
                y = out
            return y, idx
        if torch.is_grad_enabled() and self.hoist_train:
            y, idx = self._forward_train_hoisted(x, idx, layout)
            if out is not None:
                out.copy_(y)
                y = out
            return y, idx
        edge, idx = self.get_local_graph_cl(x, k, idx, layout)
        if torch.is_grad_enabled():
            # training: autograd-friendly concatenations, exactly the reference's dataflow (:53-61)
            y = torch.cat([F.relu(linear_1x1(self.mlps[0], edge)),
                           x.unsqueeze(2).expand(-1, -1, k, -1)], dim=-1)
            for i in range(1, n):
                h = linear_1x1(self.mlps[i], y)
                y = torch.cat([h if i == n - 1 else F.relu(h), y], dim=-1)
        else:
            # inference: y = [h_{n-1}, ..., h_1, h_0, x_i] along channels, one buffer filled from
            # the back (no re-copy of the growing tensor per concatenation)
            total = C + n * g
            y = x.new_empty((B, N, k, total))
            y[..., total - C:] = x.unsqueeze(2)
            lo = total - C - g
            y[..., lo:lo + g] = F.relu_(linear_1x1(self.mlps[0], edge))    # layer 0: ReLU (:57)
            for i in range(1, n):
                h = linear_1x1(self.mlps[i], y[..., lo:])
                if i != n - 1:
                    h = F.relu_(h)                                          # last layer: none (:59)
                lo -= g
                y[..., lo:lo + g] = h
        y, _ = torch.max(y, dim=2)
        if out is not None:
            out.copy_(y)
            y = out
        return y, idx

    def forward(self, x, idx=None):
        """x (B,C,N) -> y (B,C',N), idx (B,N,k)   (reference :44-64)."""
        y, idx = self.forward_cl(x.transpose(2, 1).contiguous(), idx)
        return y.transpose(2, 1).contiguous(), idx


class SampledDenseEdgeConv(DenseEdgeConv):
    """Dense edge convolution evaluated at a sampled subset of the points (reference
    layers.py:67-112; used only by AdaptiveLevel).  The subset is `nsample` furthest-point samples of
    `xyz`, or -- for nsample == 1 -- the point closest to the centroid; each sampled point's graph is
    its k nearest rows of the FULL feature set (the nearest of the k+1 dropped, as in the parent).

    Device work goes through the same kernels as the parent (FPS, gather, kNN with unique=True);
    the dense layers run channel-last on (B, nsample, k, C)."""

    def get_local_graph_cl(self, query, x, k, layout=None):
        """query (B,S,C), x (B,N,C) -> edge feature (B,S,k,2C) = [q_i, x_j - q_i], idx (B,S,k)."""
        need_grad = (x.requires_grad or query.requires_grad) and torch.is_grad_enabled()
        with torch.no_grad():
            idx, _, knn_point = operations.knn_query(k + 1, query.detach(), x.detach(), unique=True,
                                                     layout=layout, want_dist=False,
                                                     want_grouped=not need_grad)
        idx = idx[:, :, 1:]
        if need_grad:
            b = torch.arange(x.size(0), device=x.device).view(-1, 1, 1)
            knn_point = x[b, idx]
        else:
            knn_point = knn_point[:, :, 1:, :]
        center = query.unsqueeze(2).expand_as(knn_point)
        return torch.cat([center, knn_point - center], dim=-1), idx

    def sample(self, nsample, xyz):
        """xyz (B,3,N) -> sampled_xyz (B,3,nsample), sampled_idx (B,nsample)   (reference :92-100)"""
        if nsample == 1:
            center = torch.mean(xyz, dim=-1, keepdim=True)
            sampled_xyz, sampled_idx, _ = operations.group_knn(1, center, xyz, unique=False)
            return sampled_xyz.squeeze(2), sampled_idx.squeeze(1)
        sampled_idx, sampled_xyz = operations.furthest_point_sample(xyz, nsample, NCHW=True)
        return sampled_xyz, sampled_idx

    def forward(self, x, nsample, xyz):
        """x (B,C,N), xyz (B,3,N) -> y (B,C + n*growth_rate, nsample), sampled_xyz (B,3,nsample),
        sampled_idx (B,nsample)   (reference :91-112)."""
        sampled_xyz, sampled_idx = self.sample(nsample, xyz)
        sampled_x = operations.gather_points(x, sampled_idx)               # (B,C,nsample)
        q = sampled_x.transpose(2, 1).contiguous()
        edge, _ = self.get_local_graph_cl(q, x.transpose(2, 1).contiguous(), self.k)
        k, n = self.k, self.n
        y = torch.cat([F.relu(linear_1x1(self.mlps[0], edge)), q.unsqueeze(2).expand(-1, -1, k, -1)], dim=-1)
        for i in range(1, n):
            h = linear_1x1(self.mlps[i], y)
            y = torch.cat([h if i == n - 1 else F.relu(h), y], dim=-1)
        y, _ = torch.max(y, dim=2)
        return y.transpose(2, 1).contiguous(), sampled_xyz, sampled_idx


def _fused_linear(layer, x, also=None):
    """Inference shortcut of a pointwise Conv1d / Conv2d with at most 32 outputs (the prep
    convolutions of a Level): linear + bias + ReLU in one MFMA kernel that reads the input rows in
    place (e.g. a channel slice of the level's feature buffer); layers with <= 8 INPUT channels (the 3 -> 24
    lift) go to the streaming kernel, which can store the rows a second time into `also`.
    None = not applicable (`also` is then untouched)."""
    be = operations.BACKEND
    if (torch.is_grad_enabled() or not hasattr(be, "linear_small") or not x.is_cuda
            or layer.activation not in (None, "relu")):
        return None
    w = layer.conv.weight
    if x.size(-1) <= 8 and hasattr(be, "linear_lift"):
        return be.linear_lift(x, w.view(w.size(0), -1), layer.conv.bias, layer.activation == "relu", also=also)
    if layer.conv.out_channels > 32 or also is not None:
        return None
    f16 = getattr(layer, "mlp_precision", "f32") == "f16"
    y = be.linear_small(x, w.view(w.size(0), -1), layer.conv.bias, layer.activation == "relu",
                        mfma=L.MFMA_F16 if f16 else L.MFMA_F32)
    if y is None and f16:
        raise RuntimeError("mlp_precision='f16': the fused per-point kernel does not cover this layer "
                           "(%d -> %d channels)" % (x.size(-1), w.size(0)))
    return y


def _make_norm(normalization, out_channels, momentum, dims):
    if normalization == 'batch':
        cls = nn.BatchNorm2d if dims == 2 else nn.BatchNorm1d
    elif normalization == 'instance':
        cls = nn.InstanceNorm2d if dims == 2 else nn.InstanceNorm1d
    else:
        raise ValueError("only \"batch/instance\" normalization permitted.")
    return cls(out_channels, affine=True, eps=0.001, momentum=momentum)


def _make_act(activation):
    if activation == 'relu':
        return nn.ReLU()
    if activation == 'elu':
        return nn.ELU(alpha=1.0)
    if activation == 'lrelu':
        return nn.LeakyReLU(0.1)
    raise ValueError("only \"relu/elu/lrelu\" allowed")


class Conv2d(nn.Module):
    """2d convolution with custom normalization and activation (reference layers.py:115-158)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True,
                 activation=None, normalization=None, momentum=0.01):
        super(Conv2d, self).__init__()
        self.activation = activation
        self.normalization = normalization
        bias = not normalization and bias
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size,
                              stride=stride, padding=padding, bias=bias)
        if normalization is not None:
            self.norm = _make_norm(normalization, out_channels, momentum, 2)
        if activation is not None:
            self.act = _make_act(activation)

    def pointwise(self):
        """True when the layer is a pure per-point linear map (+activation): the only form the
        hot path uses (every call site passes kernel size 1 and normalization=None)."""
        c = self.conv
        return (self.normalization is None and tuple(c.kernel_size) == (1, 1)
                and tuple(c.stride) == (1, 1) and tuple(c.padding) == (0, 0))

    def forward_cl(self, x, also=None):
        """channel-last (..., C_in) -> (..., C_out); `also`: a view that receives a copy of the result."""
        assert self.pointwise()
        y = _fused_linear(self, x, also)
        if y is not None:
            return y
        y = pointwise_train(self, x)
        if y is not None:
            if also is not None:
                also.copy_(y)
            return y
        x = linear_1x1(self.conv, x)
        if self.activation is not None:
            x = self.act(x)
        if also is not None:
            also.copy_(x)
        return x

    def forward(self, x, epoch=None):
        x = self.conv(x)
        if self.normalization is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.act(x)
        return x


class Conv1d(nn.Module):
    """1d convolution with custom normalization and activation (reference layers.py:161-204)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True,
                 activation=None, normalization=None, momentum=0.01):
        super(Conv1d, self).__init__()
        self.activation = activation
        self.normalization = normalization
        bias = not normalization and bias
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size,
                              stride=stride, padding=padding, bias=bias)
        if normalization is not None:
            self.norm = _make_norm(normalization, out_channels, momentum, 1)
        if activation is not None:
            self.act = _make_act(activation)

    def pointwise(self):
        c = self.conv
        return (self.normalization is None and tuple(c.kernel_size) == (1,)
                and tuple(c.stride) == (1,) and tuple(c.padding) == (0,))

    def forward_cl(self, x):
        assert self.pointwise()
        y = _fused_linear(self, x)
        if y is not None:
            return y
        y = pointwise_train(self, x)
        if y is not None:
            return y
        x = linear_1x1(self.conv, x)
        if self.activation is not None:
            x = self.act(x)
        return x

    def forward(self, x, epoch=None):
        x = self.conv(x)
        if self.normalization is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.act(x)
        return x
