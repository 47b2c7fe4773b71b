"""Timing probe for the FPS kernels (GPU box): time vs number of rounds / points."""
import importlib
import sys
import os
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
dev = torch.device("cuda", 0)


def sphere(seed, n, b=1):
    g = torch.Generator().manual_seed(seed)
    p = torch.randn(b, n, 3, generator=g)
    return (p / p.norm(dim=2, keepdim=True)).to(dev)


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts)


if __name__ == "__main__":
    x = sphere(0, 239616)
    prev = 0.0
    for m in (2, 500, 2000, 5000, 20000, 40000, 80000):
        t = timeit(lambda: ops.fps(x, m))
        print("bucket n=239616 m=%6d  %9.3f ms   %.3f us/round (marginal)" % (m, t, (t - prev) * 1e3 / max(1, m)))
    for n, m, b in ((312, 33, 48), (624, 10, 48), (1248, 20, 48), (2496, 40, 48), (5000, 48, 1), (6240, 1248, 48),
                    (12480, 2496, 48), (24960, 4992, 48), (24960, 4992, 1)):
        x = sphere(1, n, b)
        t = timeit(lambda: ops.fps(x, m))
        print("resident b=%2d n=%6d m=%5d  %9.3f ms   %.3f us/round" % (b, n, m, t, t * 1e3 / m))
