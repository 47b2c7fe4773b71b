"""DIAGNOSTIC: the bench with the per-level resampling FPS replaced by a strided pick (wrong results,
right shapes) -- an upper bound on what a faster level FPS could buy.  Never a measurement."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ops = bench.pkg("network.operations")
orig = ops.BACKEND.fps
def fake(xyz, npoint, n_arr=None, m_arr=None):
    if 6000 <= xyz.size(1) <= 25600 and npoint >= 600:
        step = xyz.size(1) // npoint
        return (torch.arange(npoint, device=xyz.device, dtype=torch.int32) * step).unsqueeze(0).expand(xyz.size(0), -1).contiguous()
    return orig(xyz, npoint, n_arr, m_arr)
ops.BACKEND.fps = fake
bench.main()
