"""Training data path of 3PU for MI355X -- counterpart of the reference's data.py (H5Dataset :16-196).

Same class name, constructor arguments, curriculum methods and item format
(`input (B,3,M)`, `label (B,3,r*M)`, `ratio`), same use of numpy's global RNG in the same order
(ratio choice -> patch seeds -> [jitter] -> rotation angles), so a seeded run draws the reference's
patches.  What differs is where the work happens: the reference extracts the patches with
`group_knn` on CPU tensors inside DataLoader workers; here the whole data set is resident in HBM
(a few hundred MB against 288 GB) and one item = one batch of patches comes out of the device kNN
(`tpu3_knn_f32`, k up to 16 x 312 of 80 000 points through the sort kernel), the normalisation
and the rotation as device tensors, ready for `Model.set_input`.

Storage: the reference reads one HDF5 file (`h5py`, not in this image).  `.npz` files with the same
data-set names (`{tag}_{points}`) are read directly; `.hdf5` / `.h5` needs `h5py`.
`write_synthetic(path, ...)` produces a small Poisson-sphere data set in that layout.

Reference defects handled consciously: `augment` jitters an undefined name when `jitter=True`
(:158-160) -- the input patches are jittered; the `drop_out < 1` branch calls `.value` on an int and
indexes with the `None` that `np.random.shuffle` returns (:176-180) -- a random subset of
`int(M * drop_out)` input points is kept; the `phase != "train"` branch uses an undefined `B`
(:201) -- that dead assignment is dropped.
"""
import os
import re
from math import log

import numpy as np
import torch
import torch.utils.data as data

from .network import operations
from .utils import pc_utils


def _open_store(path):
    """name -> array mapping of the data-set file (npz directly, HDF5 through h5py when present)."""
    if path.endswith(".npz"):
        f = np.load(path)
        return {k: f[k] for k in f.files}
    try:
        import h5py
    except ImportError:
        raise RuntimeError("reading %s needs h5py, which is not installed; convert the file to .npz with the "
                           "same data-set names ({tag}_{points}) or use data.write_synthetic()" % path)
    with h5py.File(path, "r") as f:
        return {k: f[k][...] for k in f.keys()}


def _stem(path):
    base = os.path.basename(path)
    return base[:base.rfind(".")] if "." in base else base


class H5Dataset(data.Dataset):
    """All shapes of the file in memory (here: device memory); an item is a batch of patches."""

    def __init__(self, h5_path, num_shape_point, num_patch_point,
                 phase="train",
                 up_ratio=16, step_ratio=2,
                 jitter=False, jitter_max=0.01, jitter_sigma=0.005,
                 batch_size=16, drop_out=1.0, device=None, store=None):
        super(H5Dataset, self).__init__()
        np.random.seed(0)
        self.phase = phase
        self.is_2D = False
        self.batch_size = batch_size
        self.num_patch_point = num_patch_point
        self.num_shape_point = num_shape_point
        self.jitter = jitter
        self.jitter_max = jitter_max
        self.jitter_sigma = jitter_sigma
        self.drop_out = drop_out
        self.step_ratio = step_ratio
        if device is None:
            # the patch sampler (shape_to_patch -> operations.group_knn) runs on the HIP kernels only:
            # unlike the reference's CPU sampler there is nothing to fall back to
            if torch.cuda.is_available():
                device = torch.device("cuda", torch.cuda.current_device())
            elif getattr(operations.BACKEND, "name", "") == "hip-gfx950":
                raise RuntimeError("H5Dataset needs a ROCm device (its kNN patch sampler runs on the gfx950 "
                                   "kernels); pass device=... or make one visible")
            else:
                device = torch.device("cpu")          # a test stand-in backend is installed
        self.device = torch.device(device)
        input_array, label_array = self.load_patch_data(h5_path, up_ratio, step_ratio, num_shape_point, store)
        self.input_array = torch.from_numpy(np.ascontiguousarray(input_array)).to(self.device)
        self.label_array = {k: torch.from_numpy(np.ascontiguousarray(v)).to(self.device)
                            for k, v in label_array.items()}
        self.all_scales = [step_ratio ** r for r in range(1, int(log(up_ratio, step_ratio)) + 1)]
        self.curr_scales = [step_ratio ** r for r in range(1, int(log(up_ratio, step_ratio)) + 1)]
        self.__combined = True

    def __len__(self):
        return 300 * self.batch_size

    # ---- curriculum (reference :47-62) ----------------------------------------------------------
    def add_next_ratio(self):
        self.curr_scales = self.all_scales[:min(len(self.curr_scales) + 1, len(self.all_scales))]

    def set_combined(self):
        self.__combined = True

    def unset_combined(self):
        self.__combined = False

    def set_max_ratio(self, ratio):
        self.curr_scales = [self.step_ratio ** r for r in range(1, int(log(ratio, self.step_ratio)) + 1)]

    # ---- loading (reference :64-117) --------------------------------------------------------------
    def load_patch_data(self, h5_path, up_ratio, step_ratio, num_point, store=None):
        """File name `train_{tag}_{n1}_{tag}_{n2}_....hdf5|npz`, data sets `{tag}_{n}` of shape
        (S, n, >=3).  Input = the smallest n >= num_point, labels x{r} = the smallest n >= r * input
        points; every shape is centred and scaled by the input cloud's centroid / furthest distance.
        -> data (S,N,3) float32, label dict "x{r}" -> (S, r*N, 3)."""
        stem = _stem(h5_path)
        num_points = np.asarray(sorted(map(int, re.findall(r"\d+", stem))))
        num_in_point = num_points[np.searchsorted(num_points, num_point)]
        tag = re.findall("_([A-Za-z]+)_", os.path.basename(h5_path))[-1]
        f = store if store is not None else _open_store(h5_path)
        data_ = np.array(f[tag + "_%d" % num_in_point][:, :, 0:3])
        centroid = np.mean(data_[:, :, 0:3], axis=1, keepdims=True)
        data_[:, :, 0:3] = data_[:, :, 0:3] - centroid
        furthest_distance = np.amax(np.sqrt(np.sum(data_[:, :, 0:3] ** 2, axis=-1)), axis=1, keepdims=True)
        data_[:, :, 0:3] = data_[:, :, 0:3] / np.expand_dims(furthest_distance, axis=-1)
        label = {}
        for x in range(1, int(log(up_ratio, step_ratio) + 1)):
            r = step_ratio ** x
            closest_larger_equal = num_points[np.searchsorted(num_points, num_in_point * r)]
            lab = np.array(f[tag + "_%d" % closest_larger_equal][:, :, :3])
            lab[:, :, 0:3] = lab[:, :, 0:3] - centroid
            lab[:, :, 0:3] = lab[:, :, 0:3] / np.expand_dims(furthest_distance, axis=-1)
            label["x%d" % r] = lab
        if np.all(data_[:, :, 2] == 0):
            self.is_2D = True
        return data_, label

    # ---- patches (reference :119-140) -----------------------------------------------------------
    def shape_to_patch(self, input_pc, label_pc, ratio):
        """input_pc (1,N,3), label_pc (1,r*N,3) device tensors -> input patches (B,M,3), label
        patches (B,r*M,3): kNN patches around `batch_size` random input points (unique=True, the
        default of the reference's call)."""
        rnd = np.random.randint(0, input_pc.shape[1], [self.batch_size])
        seeds = input_pc[:, torch.from_numpy(rnd).to(input_pc.device), :]            # (1,B,3)
        label_patches = operations.group_knn(self.num_patch_point * ratio, seeds, label_pc, NCHW=False)[0][0]
        input_patches = operations.group_knn(self.num_patch_point, seeds, input_pc, NCHW=False)[0][0]
        return input_patches, label_patches

    @staticmethod
    def _normalize_by_label(input_patches, label_patches):
        """pc_utils.normalize_point_cloud on the label patches, the same centroid / radius applied to
        the input patches (reference :162-165)."""
        centroid = label_patches.mean(dim=1, keepdim=True)
        label_patches = label_patches - centroid
        furthest = label_patches.pow(2).sum(dim=-1, keepdim=True).sqrt().amax(dim=1, keepdim=True)
        return (input_patches - centroid) / furthest, label_patches / furthest

    def augment(self, input_patches, label_patches):
        """noise, common normalisation, random rotation, optional input drop-out (reference :142-182)"""
        if self.jitter:
            noise = np.clip(self.jitter_sigma * np.random.randn(*input_patches.shape).astype(np.float32),
                            -self.jitter_max, self.jitter_max)
            if self.is_2D:
                noise[:, :, 2:] = 0
            input_patches = input_patches + torch.from_numpy(noise).to(input_patches.device)
        input_patches, label_patches = self._normalize_by_label(input_patches, label_patches)
        R = torch.from_numpy(pc_utils.rotation_matrices(input_patches.shape[0], np.float32)).to(input_patches.device)
        input_patches = torch.bmm(input_patches, R)
        label_patches = torch.bmm(label_patches, R)
        if self.drop_out < 1:
            keep = int(input_patches.shape[1] * self.drop_out)
            idx = np.random.permutation(input_patches.shape[1])[:keep]
            input_patches = input_patches[:, torch.from_numpy(idx).to(input_patches.device), :]
        return input_patches, label_patches

    def __getitem__(self, index):
        if self.__combined:
            ratio = self.curr_scales[np.random.randint(len(self.curr_scales))]
        else:
            ratio = self.curr_scales[-1]
        index = index % self.input_array.shape[0]
        input_patches, label_patches = self.shape_to_patch(
            self.input_array[index:index + 1], self.label_array["x%d" % ratio][index:index + 1], ratio)
        if self.phase == "train":
            input_patches, label_patches = self.augment(input_patches, label_patches)
        else:
            input_patches, label_patches = self._normalize_by_label(input_patches, label_patches)
        return (input_patches.transpose(2, 1).contiguous(), label_patches.transpose(2, 1).contiguous(), ratio)


def write_synthetic(path, num_shapes=4, points=(312, 624, 1248, 2496, 4992), tag="poisson", seed=0):
    """A small data set in the reference's layout (`{tag}_{n}` arrays of (S,n,3) float32): every
    shape is a randomly squashed sphere, sampled uniformly and independently at each resolution.
    The file name the loader parses is derived from `points`; returns the path written."""
    rng = np.random.default_rng(seed)
    arrays = {}
    axes = rng.uniform(0.6, 1.0, size=(num_shapes, 1, 3)).astype(np.float32)
    for n in points:
        cand = rng.standard_normal((num_shapes, n, 3)).astype(np.float32)
        cand /= np.linalg.norm(cand, axis=2, keepdims=True)
        arrays["%s_%d" % (tag, n)] = (cand * axes).astype(np.float32)
    base = "train_" + "_".join("%s_%d" % (tag, n) for n in points) + ".npz"
    out = os.path.join(path, base) if os.path.isdir(path) else path
    np.savez(out, **arrays)
    return out
