"""Drop-in for the reference's `losses` extension module (losses/nmdistance.cpp:24-27): the two
Chamfer "nm-distance" entry points with the reference's positional signatures and int return
value (1 = launched, 0 = launch failed; the reference's caller ignores it, model_loss.py:15,27).
Unlike the reference (no checks at all) wrong devices / dtypes raise RuntimeError.
"""
import torch

from . import _lib as L


def _check_f32(t, name):
    L.require_device(t, name)
    L.require_dtype(t, torch.float32, name)


def _check_i32(t, name):
    L.require_device(t, name)
    L.require_dtype(t, torch.int32, name)


def nmdistance_forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
    """(xyz1[b,n,3] f32, xyz2[b,m,3] f32, dist1[b,n] f32, dist2[b,m] f32, idx1[b,n] i32,
    idx2[b,m] i32) -> int   (nmdistance.cpp:12-14)."""
    for t, nm in ((xyz1, "xyz1"), (xyz2, "xyz2"), (dist1, "dist1"), (dist2, "dist2")):
        _check_f32(t, nm)
    _check_i32(idx1, "idx1")
    _check_i32(idx2, "idx2")
    b, n, m = xyz1.size(0), xyz1.size(1), xyz2.size(1)
    if xyz2.size(0) != b or dist1.numel() != b * n or dist2.numel() != b * m \
            or idx1.numel() != b * n or idx2.numel() != b * m:
        raise RuntimeError("nmdistance_forward: tensor sizes do not match")
    with torch.cuda.device(xyz1.device):
        rc = L.lib().tpu3_nmdist_fwd_f32(L.stream_of(xyz1), b, n, m, L.ptr(xyz1), L.ptr(xyz2),
                                         L.ptr(dist1), L.ptr(dist2), L.ptr(idx1), L.ptr(idx2))
    if rc < 0:
        L.check(rc, "tpu3_nmdist_fwd_f32")
    return 1 if rc == 0 else 0


def nmdistance_backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
    """(xyz1, xyz2, gradxyz1[b,n,3] zeros, gradxyz2[b,m,3] zeros, graddist1[b,n], graddist2[b,m],
    idx1, idx2) -> int   (nmdistance.cpp:17-21)."""
    for t, nm in ((xyz1, "xyz1"), (xyz2, "xyz2"), (gradxyz1, "gradxyz1"), (gradxyz2, "gradxyz2"),
                  (graddist1, "graddist1"), (graddist2, "graddist2")):
        _check_f32(t, nm)
    _check_i32(idx1, "idx1")
    _check_i32(idx2, "idx2")
    b, n, m = xyz1.size(0), xyz1.size(1), xyz2.size(1)
    if gradxyz1.numel() != b * n * 3 or gradxyz2.numel() != b * m * 3 \
            or graddist1.numel() != b * n or graddist2.numel() != b * m:
        raise RuntimeError("nmdistance_backward: tensor sizes do not match")
    with torch.cuda.device(xyz1.device):
        rc = L.lib().tpu3_nmdist_bwd_f32(L.stream_of(xyz1), b, n, m, L.ptr(xyz1), L.ptr(xyz2),
                                         L.ptr(gradxyz1), L.ptr(gradxyz2), L.ptr(graddist1),
                                         L.ptr(graddist2), L.ptr(idx1), L.ptr(idx2))
    if rc < 0:
        L.check(rc, "tpu3_nmdist_bwd_f32")
    return 1 if rc == 0 else 0
