"""ONE cloud per call through pipeline.upsample exactly as bench.py's latency measurement issues it (optimistic feature
graphs, no small-cloud check) -- for `rocprofv3 --kernel-trace --stats`: kernels and launches per cloud.
usage: rocprofv3 --kernel-trace --stats ... -- python tools/one_cloud_profile.py [iterations]   (GPU box)"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
pipe = importlib.import_module("3pu_pytorch_amd.pipeline")
ups = importlib.import_module("3pu_pytorch_amd.network.upsampler")
work = importlib.import_module("3pu_pytorch_amd.utils.workloads")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev).eval()
cloud = work.poisson_sphere(0, 5000, dev, ops)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ts = []
with torch.no_grad():
    for it in range(iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.upsample(net, cloud, 312, 16, 3, check_small=False, optimistic_graph=True)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
print("ITERS %d  one cloud: median %.2f ms  (%s)" % (iters, sorted(ts)[len(ts) // 2], " ".join("%.1f" % t for t in ts[:8])))
