"""not-gpu: the training data path (3pu_pytorch_amd/data.py) against the fixture produced by the
reference's own data.py (oracle/make_golden.py section 7): loading / normalisation, the sequence of
items drawn from numpy's global RNG (ratio, seeds, rotation), the curriculum switches; plus the
augmentation helpers and the npz storage."""
import os

import numpy as np
import pytest
import torch

from conftest import golden, pkg
from oracle.backend import OracleBackend


@pytest.fixture()
def mods(orc, monkeypatch):
    ops = pkg("network.operations")
    monkeypatch.setattr(ops, "BACKEND", OracleBackend())
    return pkg("data"), pkg("utils.pc_utils")


def _dataset(data, g, **kw):
    store = {k[len("store_"):]: g[k] for k in g.files if k.startswith("store_")}
    return data.H5Dataset(str(g["file_name"]), num_shape_point=int(g["num_shape_point"]),
                          num_patch_point=int(g["num_patch_point"]), up_ratio=int(g["up_ratio"]), step_ratio=2,
                          batch_size=int(g["batch_size"]), store=store, **kw)


def test_loading_and_normalisation_match_reference(mods):
    data, _ = mods
    g = golden("data_path.npz")
    ds = _dataset(data, g)
    np.testing.assert_array_equal(ds.input_array.numpy(), g["input_array"])
    np.testing.assert_array_equal(ds.label_array["x2"].numpy(), g["label_x2"])
    np.testing.assert_array_equal(ds.label_array["x4"].numpy(), g["label_x4"])
    assert len(ds) == 300 * 4 and ds.all_scales == [2, 4] and not ds.is_2D


def test_item_sequence_matches_reference(mods):
    """Same RNG draws in the same order => the reference's ratios, seeds and rotations.  Patch
    membership is exact (same kNN definition); values differ only by the rounding of the mean /
    matmul (torch vs numpy): 1e-6."""
    data, _ = mods
    g = golden("data_path.npz")
    ds = _dataset(data, g)
    for i in range(6):
        if i == 4:
            ds.unset_combined()
            ds.set_max_ratio(2)
        a, b, r = ds[i]
        assert r == int(g["item%d_ratio" % i])
        assert a.shape == g["item%d_input" % i].shape and b.shape == g["item%d_label" % i].shape
        np.testing.assert_allclose(a.numpy(), g["item%d_input" % i], atol=1e-6, rtol=0)
        np.testing.assert_allclose(b.numpy(), g["item%d_label" % i], atol=1e-6, rtol=0)
    # label patches are normalised to the unit ball, input patches share their frame
    assert abs(float(b.norm(dim=1).max()) - 1.0) < 1e-5


def test_eval_phase_jitter_and_dropout(mods):
    data, _ = mods
    g = golden("data_path.npz")
    ds = _dataset(data, g, phase="test")
    a, b, r = ds[1]
    assert a.shape == (4, 3, 64) and b.shape == (4, 3, 64 * r)
    assert abs(float(b.norm(dim=1).max()) - 1.0) < 1e-5          # normalised, not rotated
    ds = _dataset(data, g, jitter=True, jitter_max=0.01, jitter_sigma=0.005, drop_out=0.5)
    a, b, r = ds[0]
    assert a.shape == (4, 3, 32) and b.shape == (4, 3, 64 * r) and torch.isfinite(a).all()


def test_augmentation_helpers(mods):
    _, pcu = mods
    np.random.seed(3)
    R = pcu.rotation_matrices(5)
    eye = np.einsum("bij,bkj->bik", R, R)
    np.testing.assert_allclose(eye, np.broadcast_to(np.eye(3), (5, 3, 3)), atol=1e-6)
    np.testing.assert_allclose(np.linalg.det(R.astype(np.float64)), 1.0, atol=1e-6)
    x = np.random.rand(2, 10, 6).astype(np.float32)
    y = np.random.rand(2, 20, 6).astype(np.float32)
    x0, y0 = x.copy(), y.copy()
    xr, yr = pcu.rotate_point_cloud_and_gt(x, y)
    np.testing.assert_allclose(np.linalg.norm(xr[..., :3], axis=-1), np.linalg.norm(x0[..., :3], axis=-1), atol=1e-5)
    np.testing.assert_allclose(np.linalg.norm(yr[..., 3:], axis=-1), np.linalg.norm(y0[..., 3:], axis=-1), atol=1e-5)
    xs, ys, s = pcu.random_scale_point_cloud_and_gt(x0.copy(), y0.copy(), 0.8, 1.2)
    assert s.shape == (2,) and ((s >= 0.8) & (s <= 1.2)).all()
    np.testing.assert_allclose(xs[..., :3], x0[..., :3] * s[:, None, None], rtol=1e-6)
    np.testing.assert_array_equal(ys[..., 3:], y0[..., 3:])


def test_npz_storage_and_missing_h5py(mods, tmp_path):
    data, _ = mods
    path = data.write_synthetic(str(tmp_path), num_shapes=2, points=(312, 624, 1248))
    assert os.path.basename(path) == "train_poisson_312_poisson_624_poisson_1248.npz"
    ds = data.H5Dataset(path, num_shape_point=300, num_patch_point=32, up_ratio=4, batch_size=3)
    assert tuple(ds.input_array.shape) == (2, 312, 3)              # smallest set >= num_shape_point
    a, b, r = ds[0]
    assert a.shape == (3, 3, 32) and b.shape == (3, 3, 32 * r) and r in (2, 4)
    with pytest.raises(RuntimeError, match="h5py"):
        data.H5Dataset(str(tmp_path / "train_poisson_312_poisson_624.hdf5"), 312, 32, up_ratio=2)
