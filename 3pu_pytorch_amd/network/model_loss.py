"""Chamfer loss of 3PU on the gfx950 nm-distance kernels.

Public names and semantics follow the reference's network/model_loss.py (`NmDistanceFunction`
:5-28, `nndistance` :30, `ChamferLoss` :33-85: threshold rule :67-77, `forward_weight`, NCHW inputs
transposed on entry); the construction is this build's own:

  * `ChamferLoss.forward` is ONE autograd node (`_ChamferNode`): the nm-distance launch pair, then
    `tpu3_chamfer_reduce_f32`, which turns the two distance rows into the loss scalar AND into
    d loss / d distance for every point (the keep-mask of the threshold rule times 1/(n B)) in a
    single pass.  Backward is then one scaling and the scatter kernel -- no mean/where/mul graph is
    recorded (the reference builds ~12 elementwise / reduction nodes per loss evaluation);
  * the reference's `NmDistanceFunction.backward` cannot run (it reads undefined names and the removed
    `ctx.saved_variables`, :22-24); the gradient implemented here is the one its kernel computes
    (nmdistance_cuda.cu:154-173).

Device tensors only: every entry point goes through `losses` (the drop-in extension-module mirror),
which raises for CPU tensors or a missing HIP library.
"""
import torch

from .. import _lib as L
from .. import losses


def _as_point_rows(t, name):
    """(B,n,3) or (B,3,n) -> contiguous (B,n,3).  A (B,3,3) tensor is taken as rows, like the
    reference (:55-63 only transposes when the last dimension is not 3)."""
    if t.dim() != 3:
        raise AssertionError("input for ChamferLoss must be a 3D-tensor, but %s.size() is %s"
                             % (name, tuple(t.size())))
    if t.size(2) != 3:
        if t.size(1) != 3:
            raise AssertionError("ChamferLoss is implemented for 3D points")
        t = t.transpose(2, 1)
    return t.contiguous()


def _nm_forward(xyz1, xyz2):
    """One launch pair: (dist1 (B,n), idx1 (B,n) i32, dist2 (B,m), idx2 (B,m) i32)."""
    b, n, m = xyz1.size(0), xyz1.size(1), xyz2.size(1)
    dev = xyz1.device
    dist1 = torch.empty((b, n), dtype=xyz1.dtype, device=dev)
    dist2 = torch.empty((b, m), dtype=xyz2.dtype, device=dev)
    idx = torch.empty((b, n + m), dtype=torch.int32, device=dev)
    idx1 = idx.view(-1)[:b * n].view(b, n)
    idx2 = idx.view(-1)[b * n:].view(b, m)
    losses.nmdistance_forward(xyz1, xyz2, dist1, dist2, idx1, idx2)
    return dist1, idx1, dist2, idx2


def _nm_backward(xyz1, xyz2, g1, g2, idx1, idx2):
    gx1 = torch.zeros_like(xyz1)
    gx2 = torch.zeros_like(xyz2)
    losses.nmdistance_backward(xyz1, xyz2, gx1, gx2, g1.contiguous(), g2.contiguous(), idx1, idx2)
    return gx1, gx2


class NmDistanceFunction(torch.autograd.Function):
    """3D point set to 3D point set distance: xyz1 (B,n,3), xyz2 (B,m,3) ->
    dist1 (B,n), idx1 (B,n) int32, dist2 (B,m), idx2 (B,m) int32; differentiable in both sets."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
        dist1, idx1, dist2, idx2 = _nm_forward(xyz1, xyz2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, idx1, dist2, idx2

    @staticmethod
    def backward(ctx, g_dist1, _g_idx1, g_dist2, _g_idx2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        return _nm_backward(xyz1, xyz2, g_dist1, g_dist2, idx1, idx2)


nndistance = NmDistanceFunction.apply


class _ChamferNode(torch.autograd.Function):
    """pred (B,n,3), gt (B,m,3) -> scalar Chamfer loss; see the module docstring."""

    @staticmethod
    def forward(ctx, pred, gt, threshold, forward_weight):
        dist1, idx1, dist2, idx2 = _nm_forward(pred, gt)
        b, n, m = pred.size(0), pred.size(1), gt.size(1)
        dev = pred.device
        out = torch.empty((1 + b,), dtype=torch.float32, device=dev)       # [loss | cd per element]
        want_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        gw1 = torch.empty_like(dist1) if want_grad else None
        gw2 = torch.empty_like(dist2) if want_grad else None
        with torch.cuda.device(dev):
            L.check(L.lib().tpu3_chamfer_reduce_f32(
                L.stream_of(pred), b, n, m, L.ptr(dist1), L.ptr(dist2),
                0 if threshold is None else 1, 0.0 if threshold is None else float(threshold),
                float(forward_weight), L.ptr(out), out.data_ptr() + 4, L.ptr(gw1), L.ptr(gw2)),
                "tpu3_chamfer_reduce_f32")
        if want_grad:
            ctx.save_for_backward(pred, gt, idx1, idx2, gw1, gw2)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        pred, gt, idx1, idx2, gw1, gw2 = ctx.saved_tensors
        gx1, gx2 = _nm_backward(pred, gt, gw1 * g, gw2 * g, idx1, idx2)
        return (gx1 if ctx.needs_input_grad[0] else None, gx2 if ctx.needs_input_grad[1] else None,
                None, None)


class ChamferLoss(torch.nn.Module):
    """chamfer loss: mean over the batch of forward_weight * mean_i d(pred_i, gt) + mean_j d(gt_j, pred),
    d = squared distance to the nearest point of the other set.  With a threshold, distances of
    threshold * (row mean) or more are ignored (strong outliers), the divisor stays the row length."""

    def __init__(self, threshold=None, forward_weight=1.0):
        super(ChamferLoss, self).__init__()
        self._threshold = threshold
        self.forward_weight = forward_weight

    def set_threshold(self, value):
        self._threshold = value

    def unset_threshold(self):
        self._threshold = None

    def forward(self, pred, gt):
        pred = _as_point_rows(pred, "pred")
        gt = _as_point_rows(gt, "gt")
        if pred.size(2) != 3 or gt.size(2) != 3:
            raise AssertionError("ChamferLoss is implemented for 3D points")
        return _ChamferNode.apply(pred, gt, self._threshold, self.forward_weight)
