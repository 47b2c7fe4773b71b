"""Time the fused inter-level skip kernel on level-4-shaped synthetic data (one 3840-patch chunk)."""
import importlib, sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tpu = importlib.import_module("3pu_pytorch_amd")
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
dev = torch.device("cuda:0")
clouds = int(os.environ.get("CLOUDS", "96"))
P, n, K, C, m, uniq = 40, 312, 5, 264, 6240, 1400
B = clouds * P
g = torch.Generator(device=dev).manual_seed(0)
xyz = torch.rand((B, n, 3), device=dev, generator=g)
feat = torch.rand((B, n, C), device=dev, generator=g)
pxyz = torch.rand((clouds, m, 3), device=dev, generator=g)
pfeat = torch.rand((clouds, m, C), device=dev, generator=g)
# neighbours: spatially coherent within a patch -- a window of the unique rows
base = torch.randint(0, uniq - 400, (B, 1, 1), device=dev, generator=g)
idx = (base + torch.randint(0, 400, (B, n, K), device=dev, generator=g)).to(torch.int64)
if os.environ.get("COHERENT"):      # neighbours vary smoothly along the point index: points i .. i + 15 share most rows
    ii = torch.arange(n, device=dev).view(1, n, 1)
    idx = (base + ii // 2 + torch.arange(K, device=dev).view(1, 1, K) * 3).clamp_(max=m - 1).to(torch.int64)
if os.environ.get("SAME_ROW"):
    idx = torch.zeros_like(idx) + torch.arange(K, device=dev).view(1, 1, K)
owner = torch.repeat_interleave(torch.arange(clouds, dtype=torch.int32, device=dev), P)
for per in (0, P):
    ts = []
    for it in range(6):
        f = feat.clone()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.BACKEND.interlevel_skip(xyz, f, pxyz, pfeat, owner, idx, per_cloud=per)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("per_cloud=%d  ms: %s" % (per, " ".join("%.3f" % t for t in ts)))
