"""Drop-in for the reference's `sampling` extension module (sampling/sampling.cpp:83-88):
same four names, same positional signatures, same in-place/aliasing behaviour, backed by the
gfx950 kernels of lib3pu_hip.so.  Stricter than the reference where that is invisible to a
correct caller: dtype checks, a device guard and launches on torch's current stream
(the reference launches on the legacy default stream with no guard).
"""
import torch

from . import _lib as L

_ELEM = {torch.float16: 2, torch.float32: 4, torch.float64: 8}


def furthest_sampling(b, n, m, input, temp, idx):
    """(int b, int n, int m, Tensor input[b,n,3] f32, Tensor temp[b,n] f32 (=1e10),
    Tensor idx[b,m] i32) -> idx   (sampling.cpp:26-35)."""
    L.require_device(input, "input")
    L.require_device(temp, "temp")
    L.require_device(idx, "idx")
    L.require_dtype(input, torch.float32, "input")
    L.require_dtype(temp, torch.float32, "temp")
    L.require_dtype(idx, torch.int32, "idx")
    if input.numel() != b * n * 3 or temp.numel() != b * n or idx.numel() != b * m:
        raise RuntimeError("furthest_sampling: tensor sizes do not match (b,n,m)=(%d,%d,%d)" % (b, n, m))
    with torch.cuda.device(input.device):
        L.check(L.lib().tpu3_fps_f32(L.stream_of(input), b, n, m, L.ptr(input), L.ptr(temp), L.ptr(idx)),
                "tpu3_fps_f32")
    return idx


def gather_forward(b, c, n, npoints, points, idx, out):
    """(int b, int c, int n, int npoints, Tensor points[b,c,n], Tensor idx[b,npoints] i32,
    Tensor out[b,c,npoints]) -> out   (sampling.cpp:37-45)."""
    L.require_device(points, "points_tensor")
    L.require_device(idx, "idx_tensor")
    L.require_device(out, "out_tensor")
    L.require_dtype(idx, torch.int32, "idx_tensor")
    if points.dtype not in _ELEM or out.dtype != points.dtype:
        raise RuntimeError("gather_forward: points/out must share a floating dtype")
    if points.numel() != b * c * n or idx.numel() != b * npoints or out.numel() != b * c * npoints:
        raise RuntimeError("gather_forward: tensor sizes do not match")
    with torch.cuda.device(points.device):
        L.check(L.lib().tpu3_gather_fwd(L.stream_of(points), b, c, n, npoints, _ELEM[points.dtype],
                                        L.ptr(points), L.ptr(idx), L.ptr(out)), "tpu3_gather_fwd")
    return out


def gather_backward(b, c, n, npoints, grad_out, idx, grad_points):
    """(int b, int c, int n, int npoints, Tensor grad_out[b,c,npoints], Tensor idx i32,
    Tensor grad_points[b,c,n] zeros) -> grad_points   (sampling.cpp:47-53)."""
    L.require_device(grad_out, "grad_out_tensor")
    L.require_device(idx, "idx_tensor")
    L.require_device(grad_points, "grad_points_tensor")
    L.require_dtype(idx, torch.int32, "idx_tensor")
    if grad_out.dtype not in _ELEM or grad_points.dtype != grad_out.dtype:
        raise RuntimeError("gather_backward: grad tensors must share a floating dtype")
    if grad_out.numel() != b * c * npoints or idx.numel() != b * npoints or grad_points.numel() != b * c * n:
        raise RuntimeError("gather_backward: tensor sizes do not match")
    with torch.cuda.device(grad_out.device):
        L.check(L.lib().tpu3_gather_bwd(L.stream_of(grad_out), b, c, n, npoints, _ELEM[grad_out.dtype],
                                        L.ptr(grad_out), L.ptr(idx), L.ptr(grad_points)),
                "tpu3_gather_bwd")
    return grad_points


def ball_query(query, xyz, radius, nsample):
    """(Tensor query[b,m,3], Tensor xyz[b,n,3], float radius, int nsample) -> idx[b,m,nsample] i32
    (sampling.cpp:59-81).  CPU tensors are rejected like the reference ("CPU not supported")."""
    if not query.is_cuda:
        raise RuntimeError("query must be a CUDA tensor")
    L.require_device(query, "query")
    L.require_device(xyz, "xyz")
    if xyz.dtype not in (torch.float32, torch.float64) or query.dtype != xyz.dtype:
        raise RuntimeError("ball_query: query/xyz must both be float32 or float64")
    b, m = query.size(0), query.size(1)
    n = xyz.size(1)
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=query.device)
    with torch.cuda.device(query.device):
        L.check(L.lib().tpu3_ball_query(L.stream_of(query), xyz.size(0), n, m, float(radius), int(nsample),
                                        _ELEM[xyz.dtype], L.ptr(query), L.ptr(xyz), L.ptr(idx)),
                "tpu3_ball_query")
    return idx
