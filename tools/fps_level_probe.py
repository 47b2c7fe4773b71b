"""Time the per-level resampling FPS shapes of the 8-cloud bench (b = 384 merged patch sets)."""
import importlib, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
def cloud(b, n):
    p = torch.randn(b, n, 3, generator=g)
    return (p / p.norm(dim=2, keepdim=True)).to(dev)
for b, n, m in ((384, 24960, 2496), (384, 12480, 1248), (384, 6240, 624), (256, 24960, 2496), (128, 24960, 2496), (1, 24960, 2496)):
    x = cloud(b, n)
    ts = []
    for it in range(4):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.fps(x, m)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("b=%3d n=%5d m=%4d  %8.3f ms  %.3f us/round" % (b, n, m, min(ts), min(ts) * 1e3 / m))
