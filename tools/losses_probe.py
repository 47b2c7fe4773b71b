"""The point-set kernels OUTSIDE the inference step, launched a fixed number of times each so that a profiler run of
this script has something to average: nm-distance forward in both forms (scan = the reference's algorithm,
csrc/nmdistance.hip; grid = the pruned exact search, csrc/nmdist_grid.hip) and backward, ball query, gather forward /
backward -- the sizes bench.py's `rooflines_other` quotes.

usage (tools/collect_profiles.sh):
    rocprofv3 --kernel-trace --stats ... -- python tools/losses_probe.py          -> profiles/r06_kernel_stats_losses.csv
    rocprofv3 --pmc SQ_INSTS_VALU ...    -- python tools/losses_probe.py --reps 3 -> profiles/r06_pmc_*_losses_by_kernel.txt"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("3pu_pytorch_amd")
losses = importlib.import_module("3pu_pytorch_amd.losses")
sampling = importlib.import_module("3pu_pytorch_amd.sampling")
lib = importlib.import_module("3pu_pytorch_amd._lib").lib()

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)


def sphere(b, n, scale=1.0):
    x = torch.randn((b, n, 3), device=dev, generator=g)
    return (x / x.norm(dim=2, keepdim=True) * scale).contiguous()


for b, n, m in ((32, 624, 624), (32, 4992, 4992), (1, 80000, 80000), (1, 1280000, 1280000)):
    x1, x2 = sphere(b, n), sphere(b, m, 1.01)
    d1, d2 = torch.empty((b, n), device=dev), torch.empty((b, m), device=dev)
    i1 = torch.empty((b, n), dtype=torch.int32, device=dev)
    i2 = torch.empty((b, m), dtype=torch.int32, device=dev)
    for form in (0, 1):
        if form == 0 and n > 400000:
            continue                        # (0.5 s per call: the scan's rate is the 80 000 x 80 000 one)
        lib.tpu3_debug_nmdist_form(form)
        for _ in range(args.reps):
            losses.nmdistance_forward(x1, x2, d1, d2, i1, i2)
    lib.tpu3_debug_nmdist_form(-1)
    g1, g2 = torch.ones_like(d1), torch.ones_like(d2)
    gx1, gx2 = torch.zeros_like(x1), torch.zeros_like(x2)
    for _ in range(args.reps):
        losses.nmdistance_backward(x1, x2, gx1, gx2, g1, g2, i1, i2)
    torch.cuda.synchronize()
xyz, q = sphere(48, 5000), sphere(48, 312)
for _ in range(args.reps):
    sampling.ball_query(q, xyz, 0.1, 32)
for b, c, n, m in ((1, 3, 239616, 80000), (48, 3, 24960, 4992)):
    pts = torch.randn((b, c, n), device=dev, generator=g)
    idx = torch.randint(0, n, (b, m), device=dev, generator=g, dtype=torch.int32)
    out = torch.empty((b, c, m), device=dev)
    go, gp = torch.randn((b, c, m), device=dev, generator=g), torch.zeros((b, c, n), device=dev)
    for _ in range(args.reps):
        sampling.gather_forward(b, c, n, m, pts, idx, out)
        sampling.gather_backward(b, c, n, m, go, idx, gp)
torch.cuda.synchronize()
print("done")
