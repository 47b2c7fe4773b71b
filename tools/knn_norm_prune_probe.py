"""How selective would norm-difference pruning be for the feature kNN graphs of a real step?
For every knn_graph call: fraction of (query, candidate) pairs with (|p| - |q|)^2 <= d_k(q)."""
import importlib, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ops, pipe, ups = bench.pkg("network.operations"), bench.pkg("pipeline"), bench.pkg("network.upsampler")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev).eval()
clouds = torch.cat([bench.poisson_sphere(i, 5000, dev, ops) for i in range(2)], dim=0)
orig = ops.BACKEND.knn_graph
stats = []
def hook(k, x, layout=None):
    idx = orig(k, x, layout)
    xs = x[:64].double()                                  # a sample of patches
    d = torch.cdist(xs, xs) ** 2
    dk = d.topk(k, dim=2, largest=False).values[:, :, -1]            # (P, n)
    nrm = xs.norm(dim=2)
    lb = (nrm.unsqueeze(2) - nrm.unsqueeze(1)) ** 2                   # (P, n, n)
    frac = (lb <= dk.unsqueeze(2)).double().mean().item()
    stats.append((tuple(x.shape), frac))
    return idx
ops.BACKEND.knn_graph = hook
with torch.no_grad():
    pipe.upsample(net, clouds, 312, 16, 3, final_fps=False)
for i, (shp, f) in enumerate(stats):
    print(i, shp, "fraction of pairs surviving the norm bound: %.3f" % f)
