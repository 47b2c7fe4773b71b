"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front end of ``liboracle.so`` (the C restatement in ``ref_kernels.c``) plus
numpy/torch-CPU restatements of the reference's torch-level operators for the
patch-upsampling path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; nothing under
``3pu_pytorch_amd/`` does.

Every function cites the reference lines it follows (paths relative to the
reference checkout).  Kernel-level parity is otherwise unpinned (the reference
ships no tests); the torch-level functions are pinned by the
fixtures in ``tests/golden/`` that ``oracle/make_golden.py`` produced by importing
the reference's own Python in the build container.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ORC_FMA = 1
ORC_GRID32_BUG = 2


def build(force=False):
    """Compile ref_kernels.c -> liboracle.so with gcc (see oracle/Makefile)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "ref_kernels.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.orc_opt_n_threads.restype = ctypes.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def opt_n_threads(work_size):
    """sampling/cuda_utils.h:9-14."""
    return int(lib().orc_opt_n_threads(int(work_size)))


def fps(xyz, m, temp=None, flags=ORC_FMA):
    """sampling_cuda.cu:103-265.  xyz (B,N,3) f32 -> idx (B,m) i32, temp (B,N) f32."""
    xyz = _f32(xyz)
    b, n, _ = xyz.shape
    if temp is None:
        temp = np.full((b, n), 1e10, dtype=np.float32)
    else:
        temp = _f32(temp).copy()
    idx = np.zeros((b, m), dtype=np.int32)
    lib().orc_fps_f32(b, n, m, _p(xyz), _p(temp), _p(idx), flags)
    return idx, temp


def gather_fwd(points, idx):
    """sampling_cuda.cu:28-41.  points (B,C,N) any float dtype, idx (B,m) -> (B,C,m)."""
    points = np.ascontiguousarray(points)
    idx = _i32(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.empty((b, c, m), dtype=points.dtype)
    lib().orc_gather_fwd(b, c, n, m, points.dtype.itemsize, _p(points), _p(idx), _p(out))
    return out


def gather_bwd(grad_out, idx, n):
    """sampling_cuda.cu:66-80.  grad_out (B,C,m) f32/f64 -> grad_points (B,C,N)."""
    grad_out = np.ascontiguousarray(grad_out)
    idx = _i32(idx)
    b, c, m = grad_out.shape
    gp = np.zeros((b, c, n), dtype=grad_out.dtype)
    fn = {np.dtype(np.float32): lib().orc_gather_bwd_f32,
          np.dtype(np.float64): lib().orc_gather_bwd_f64}[grad_out.dtype]
    fn(b, c, n, m, _p(grad_out), _p(idx), _p(gp))
    return gp


def ball_query(query, xyz, radius, nsample, flags=ORC_FMA):
    """sampling_cuda.cu:269-317 + sampling.cpp:59-81.  (B,M,3),(B,N,3) -> (B,M,nsample) i32."""
    dt = np.float64 if np.asarray(xyz).dtype == np.float64 else np.float32
    query = np.ascontiguousarray(query, dtype=dt)
    xyz = np.ascontiguousarray(xyz, dtype=dt)
    b, m, _ = query.shape
    n = xyz.shape[1]
    idx = np.zeros((b, m, nsample), dtype=np.int32)
    fn = lib().orc_ball_query_f64 if dt == np.float64 else lib().orc_ball_query_f32
    fn(b, n, m, ctypes.c_float(radius), nsample, _p(query), _p(xyz), _p(idx), flags)
    return idx


def nmdistance_fwd(xyz1, xyz2, flags=ORC_FMA):
    """nmdistance_cuda.cu:11-153.  (B,n,3),(B,m,3) -> dist1,idx1,dist2,idx2."""
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.empty((b, n), np.float32)
    d2 = np.empty((b, m), np.float32)
    i1 = np.empty((b, n), np.int32)
    i2 = np.empty((b, m), np.int32)
    lib().orc_nmdistance_fwd(b, n, m, _p(xyz1), _p(xyz2), _p(d1), _p(d2), _p(i1), _p(i2), flags)
    return d1, i1, d2, i2


def nmdistance_bwd(xyz1, xyz2, graddist1, graddist2, idx1, idx2):
    """nmdistance_cuda.cu:154-193 -> gradxyz1 (B,n,3), gradxyz2 (B,m,3)."""
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = np.zeros((b, n, 3), np.float32)
    g2 = np.zeros((b, m, 3), np.float32)
    lib().orc_nmdistance_bwd(b, n, m, _p(xyz1), _p(xyz2), _p(g1), _p(g2), _p(_f32(graddist1)),
                             _p(_f32(graddist2)), _p(_i32(idx1)), _p(_i32(idx2)))
    return g1, g2


def first_occurrence_dup(points):
    """operations.py:194-200 -- dup[b,n] = 1 iff an identical row exists at a smaller index
    (the complement of np.unique(axis=0, return_index=True)).  points (B,N,C) f32."""
    points = _f32(points)
    b, n, c = points.shape
    dup = np.zeros((b, n), np.uint8)
    lib().orc_first_occurrence_dup(b, n, c, _p(points), _p(dup))
    return dup


def knn(k, query, points, unique=True):
    """operations.py:151-216 on channel-last arrays: query (B,M,C), points (B,N,C) f32 ->
    idx (B,M,k) i32, dist (B,M,k) f32 ascending; fixed summation order, ties to lowest index
    (see ref_kernels.c)."""
    query, points = _f32(query), _f32(points)
    b, m, c = query.shape
    n = points.shape[1]
    assert n >= k, "points size must be greater or equal to k"
    idx = np.empty((b, m, k), np.int32)
    dist = np.empty((b, m, k), np.float32)
    lib().orc_knn_f32(b, m, n, c, k, _p(query), _p(points), int(bool(unique)), _p(idx), _p(dist))
    return idx, dist


# ---------------------------------------------------------------------------------------------
# torch-level operators (numpy, NCHW like the reference)
# ---------------------------------------------------------------------------------------------

def normalize_point_batch(pc, NCHW=True):
    """operations.py:12-30 (fp32; numpy's pairwise mean may differ from torch's in the last
    ulp -- compared with tolerance)."""
    pc = np.asarray(pc, np.float32)
    point_axis = 2 if NCHW else 1
    dim_axis = 1 if NCHW else 2
    centroid = pc.mean(axis=point_axis, keepdims=True, dtype=np.float32)
    pc = pc - centroid
    fd = np.sqrt((pc ** 2).sum(axis=dim_axis, keepdims=True, dtype=np.float32)).max(
        axis=point_axis, keepdims=True)
    return pc / fd, centroid, fd


def group_knn(k, query, points, unique=True, NCHW=True):
    """operations.py:165-216 -> (neighbours (B,C,M,k)|(B,M,k,C), idx int64 (B,M,k), dist)."""
    query = np.asarray(query, np.float32)
    points = np.asarray(points, np.float32)
    if NCHW:
        q = np.ascontiguousarray(query.transpose(0, 2, 1))
        p = np.ascontiguousarray(points.transpose(0, 2, 1))
    else:
        q, p = query, points
    idx, dist = knn(k, q, p, unique)
    b = np.arange(p.shape[0])[:, None, None]
    nb = p[b, idx]  # (B,M,k,C)
    if NCHW:
        nb = nb.transpose(0, 3, 1, 2)
    return nb, idx.astype(np.int64), dist


def furthest_point_sample(xyz, npoint, NCHW=True):
    """operations.py:303-323 -> (idx int32 (B,npoint), sampled (B,3,npoint)|(B,npoint,3))."""
    xyz = np.asarray(xyz, np.float32)
    pts = np.ascontiguousarray(xyz.transpose(0, 2, 1)) if NCHW else xyz
    idx, _ = fps(pts, npoint)
    sampled = gather_fwd(np.ascontiguousarray(pts.transpose(0, 2, 1)), idx)
    if not NCHW:
        sampled = np.ascontiguousarray(sampled.transpose(0, 2, 1))
    return idx, sampled


def chamfer_loss(pred, gt, threshold=None, forward_weight=1.0):
    """model_loss.py:50-85 (forward only).  pred, gt (B,n,3) or (B,3,n)."""
    pred = np.asarray(pred, np.float32)
    gt = np.asarray(gt, np.float32)
    if pred.shape[2] != 3:
        pred = pred.transpose(0, 2, 1)
    if gt.shape[2] != 3:
        gt = gt.transpose(0, 2, 1)
    p2g, _, g2p, _ = nmdistance_fwd(pred, gt)
    if threshold is not None:
        ft = p2g.mean(axis=1, keepdims=True, dtype=np.float32) * np.float32(threshold)
        bt = g2p.mean(axis=1, keepdims=True, dtype=np.float32) * np.float32(threshold)
        p2g = np.where(p2g < ft, p2g, np.zeros_like(p2g))
        g2p = np.where(g2p < bt, g2p, np.zeros_like(g2p))
    cd = np.float32(forward_weight) * p2g.mean(axis=1, dtype=np.float32) + g2p.mean(
        axis=1, dtype=np.float32)
    return cd.mean(dtype=np.float32)
