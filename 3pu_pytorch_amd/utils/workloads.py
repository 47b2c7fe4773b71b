"""Synthetic inputs of the BASELINE configurations, generated on the device."""
import torch

from ..network import operations


def poisson_sphere(seed, n, dev, ops=operations):
    """Config C2's cloud: a blue-noise-like sphere -- 8n uniform S^2 candidates drawn from torch.Generator(seed) on
    the host, thinned to n by (this library's) FPS -> (1,3,n) f32 on `dev`.  Stands in for the `poisson_5000` data
    the reference's Readme names; oracle/make_golden.py::c2_cloud is the same construction over the C oracle."""
    g = torch.Generator().manual_seed(seed)
    cand = torch.randn(1, 8 * n, 3, generator=g)
    cand = (cand / cand.norm(dim=2, keepdim=True)).to(dev)
    idx = ops.fps(cand, n)
    return torch.gather(cand, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).transpose(2, 1).contiguous()
