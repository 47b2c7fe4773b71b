"""oracle/cpu_baseline.py -- TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.

The `cpu_baseline` leg of bench.py: the CPU restatement of the hot path (the C oracle for FPS /
kNN / gather, torch-CPU for the conv stacks, driven by the same host logic as the product through
oracle/backend.py) timed on the host cores of the box bench.py runs on, on a BOUNDED sample of
config C2 (5000 -> 80000 points, 16x, 48 outer patches).  kind = "port": the reference has no CPU
path for FPS / gather / Chamfer (CUDA-only extensions), so this is the build's own restatement.
"""
import importlib
import os
import time

import numpy as np
import torch


def measure(num_shape_point=5000, num_point=312, up_ratio=16, sample_patches=1, fps_rounds=1500):
    from . import oracle as orc
    from .backend import OracleBackend
    ops = importlib.import_module("3pu_pytorch_amd.network.operations")
    ups = importlib.import_module("3pu_pytorch_amd.network.upsampler")
    pipe = importlib.import_module("3pu_pytorch_amd.pipeline")
    # the conv stacks of a 312-point patch are tiny GEMMs: more than ~32 threads only adds
    # synchronisation cost (256 threads measured 2.5x slower than 8); `cores` reports what is used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    saved = ops.BACKEND
    ops.BACKEND = OracleBackend()
    try:
        rng = np.random.default_rng(0)
        cloud = rng.standard_normal((1, num_shape_point, 3)).astype(np.float32)
        cloud /= np.linalg.norm(cloud, axis=2, keepdims=True)
        x = torch.from_numpy(np.ascontiguousarray(cloud.transpose(0, 2, 1)))
        torch.manual_seed(0)
        net = ups.Net(max_up_ratio=up_ratio, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).eval()
        t0 = time.perf_counter()
        _, patches, _ = pipe.extract_outer_patches(x, num_point, 3)
        t_outer = time.perf_counter() - t0
        P = patches.size(1)
        t0 = time.perf_counter()
        with torch.no_grad():
            up, _ = pipe.upsample_patches(net, patches[0, :sample_patches], up_ratio)
        t_patch = (time.perf_counter() - t0) / sample_patches
        # final FPS: the merged cloud has P * num_point * up_ratio points; time `fps_rounds` rounds
        n_merged = P * num_point * up_ratio
        merged = rng.standard_normal((1, n_merged, 3)).astype(np.float32)
        merged /= np.linalg.norm(merged, axis=2, keepdims=True)
        t0 = time.perf_counter()
        orc.fps(merged, fps_rounds)
        t_round = (time.perf_counter() - t0) / max(1, fps_rounds - 1)
        m_out = num_shape_point * up_ratio
        t_fps = t_round * (m_out - 1)
        total = t_outer + P * t_patch + t_fps
        return {
            "value": m_out / total, "unit": "points/s", "cores": cores, "kind": "port",
            "sample": ("C2 cloud: outer FPS+kNN measured in full (%.3f s); %d of %d outer patches through "
                       "all %d levels (%.2f s/patch, torch-CPU %d threads + C oracle kernels); final FPS "
                       "%d of %d rounds over %d points (%.2f ms/round, single-thread C); whole-cloud time "
                       "extrapolated = %.1f s" % (t_outer, sample_patches, P, int(np.log2(up_ratio)), t_patch,
                                                  cores, fps_rounds, m_out, n_merged, t_round * 1e3, total)),
        }
    finally:
        ops.BACKEND = saved
