"""oracle/refshim.py -- TEST INFRASTRUCTURE (build container only).

Makes the reference's *Python* layer importable on CPU so that golden vectors can be
generated from it.  The reference imports three modules that do not
exist here -- ``faiss`` (unused at run time), and its two CUDA extensions ``sampling`` and
``losses`` -- so stand-ins are injected into ``sys.modules`` before
``network.operations`` is imported.  The stand-ins are backed by the CPU oracle
(``oracle/ref_kernels.c``); every fixture produced through them says so in its metadata.

Nothing here is used on the GPU box: ``/root/reference`` does not exist there.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

from . import oracle as orc

REFERENCE_ROOT = os.environ.get("TPU3_REFERENCE_ROOT", "/root/reference")


def _np(t):
    return t.detach().cpu().numpy()


def _make_sampling():
    m = types.ModuleType("sampling")

    def furthest_sampling(b, n, npoint, xyz, temp, idx):
        i, t = orc.fps(_np(xyz), npoint, temp=_np(temp))
        idx.copy_(torch.from_numpy(i))
        temp.copy_(torch.from_numpy(t))
        return idx

    def gather_forward(b, c, n, npoints, points, idx, out):
        out.copy_(torch.from_numpy(orc.gather_fwd(_np(points), _np(idx))))
        return out

    def gather_backward(b, c, n, npoints, grad_out, idx, grad_points):
        grad_points.add_(torch.from_numpy(orc.gather_bwd(_np(grad_out), _np(idx), n)))
        return grad_points

    def ball_query(query, xyz, radius, nsample):
        return torch.from_numpy(orc.ball_query(_np(query), _np(xyz), radius, nsample))

    m.furthest_sampling = furthest_sampling
    m.gather_forward = gather_forward
    m.gather_backward = gather_backward
    m.ball_query = ball_query
    return m


def _make_losses():
    m = types.ModuleType("losses")

    def nmdistance_forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
        d1, i1, d2, i2 = orc.nmdistance_fwd(_np(xyz1), _np(xyz2))
        dist1.copy_(torch.from_numpy(d1))
        dist2.copy_(torch.from_numpy(d2))
        idx1.copy_(torch.from_numpy(i1))
        idx2.copy_(torch.from_numpy(i2))
        return 1

    def nmdistance_backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
        g1, g2 = orc.nmdistance_bwd(_np(xyz1), _np(xyz2), _np(graddist1), _np(graddist2),
                                    _np(idx1), _np(idx2))
        gradxyz1.add_(torch.from_numpy(g1))
        gradxyz2.add_(torch.from_numpy(g2))
        return 1

    m.nmdistance_forward = nmdistance_forward
    m.nmdistance_backward = nmdistance_backward
    return m


def import_reference():
    """Return the reference's (operations, layers, upsampler, model_loss) modules, imported
    from REFERENCE_ROOT with the three stand-in modules in place.  Build container only."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference checkout not present at %s" % REFERENCE_ROOT)
    saved = {k: sys.modules.get(k) for k in ("faiss", "sampling", "losses", "network")}
    sys.modules["faiss"] = types.ModuleType("faiss")
    sys.modules["sampling"] = _make_sampling()
    sys.modules["losses"] = _make_losses()
    for k in [k for k in sys.modules if k == "network" or k.startswith("network.")]:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        ops = importlib.import_module("network.operations")
        layers = importlib.import_module("network.layers")
        ups = importlib.import_module("network.upsampler")
        loss = importlib.import_module("network.model_loss")
    finally:
        sys.path.remove(REFERENCE_ROOT)
        # leave the reference's `network` package importable under a private alias only
        for k in [k for k in list(sys.modules) if k == "network" or k.startswith("network.")]:
            sys.modules["_ref_" + k] = sys.modules.pop(k)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
            elif k in sys.modules and k != "network":
                del sys.modules[k]
    return ops, layers, ups, loss


class _FakeH5File:
    """h5py.File stand-in over a dict of arrays (dataset access `f[name][...]` and `close`)."""

    def __init__(self, store):
        self._store = store

    def __getitem__(self, name):
        return np.array(self._store[name])          # a fresh array, like a dataset read

    def close(self):
        pass


def import_reference_data(store):
    """Import the reference's data.py (H5Dataset) with `h5py` served from `store` (a dict
    dataset-name -> array standing for the HDF5 file) and `plyfile` stubbed; its group_knn is the
    reference's own (through import_reference).  Build container only."""
    ops, layers, ups, loss = import_reference()
    saved = {k: sys.modules.get(k) for k in ("h5py", "plyfile", "network", "network.operations", "misc",
                                              "misc.logger", "utils", "utils.pc_utils", "data")}
    h5 = types.ModuleType("h5py")
    h5.File = lambda path, mode="r": _FakeH5File(store)
    sys.modules["h5py"] = h5
    sys.modules["plyfile"] = types.ModuleType("plyfile")
    net_pkg = types.ModuleType("network")
    net_pkg.operations = ops
    sys.modules["network"] = net_pkg
    sys.modules["network.operations"] = ops
    for k in ("misc", "misc.logger", "utils", "utils.pc_utils", "data"):
        sys.modules.pop(k, None)
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        data = importlib.import_module("data")
        pcu = importlib.import_module("utils.pc_utils")
    finally:
        sys.path.remove(REFERENCE_ROOT)
        for k in ("misc", "misc.logger", "utils", "utils.pc_utils", "data"):
            if k in sys.modules:
                sys.modules["_ref_" + k] = sys.modules.pop(k)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
            else:
                sys.modules.pop(k, None)
    return data, pcu
