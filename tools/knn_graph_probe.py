"""Time the kNN-graph kernel on one DenseEdgeConv-shaped chunk (3840 patches x 312 points x 24 channels, k=33)."""
import importlib, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("3pu_pytorch_amd")
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
dev = torch.device("cuda:0")
B = int(os.environ.get("PATCHES", "3840"))
n, C, k = int(os.environ.get("N", "312")), 24, 33
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand((B, n, C), device=dev, generator=g)
owner = torch.repeat_interleave(torch.arange(B // 40, dtype=torch.int32, device=dev), 40)
lay = dict(grp=owner, groups=B // 40)
ts = []
for it in range(int(os.environ.get("ITERS", "6"))):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    idx = ops.BACKEND.knn_graph(k, x, layout=lay)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("knn_graph ms: %s   min %.3f median %.3f   %.2f ns per query" % (" ".join("%.3f" % t for t in ts[:6]), min(ts), sorted(ts)[len(ts) // 2], sorted(ts)[len(ts) // 2] * 1e6 / (B * n)))
