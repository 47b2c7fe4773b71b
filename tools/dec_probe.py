"""Time the fused DenseEdgeConv kernel on one level-sized chunk (3840 patches x 312 points, k = 32)."""
import importlib, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("3pu_pytorch_amd")
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
layers = importlib.import_module("3pu_pytorch_amd.network.layers")
dev = torch.device("cuda:0")
B, n, C, k = int(os.environ.get("PATCHES", "3840")), int(os.environ.get("N", "312")), 24, 32
torch.manual_seed(0)
conv = layers.DenseEdgeConv(24, growth_rate=12, n=3, k=k).to(dev).eval()
x = torch.rand((B, n, C), device=dev)
idx = torch.randint(0, n, (B, n, k + 1), device=dev, dtype=torch.int32)
out = torch.empty((B, n, 60), device=dev)
ts = []
with torch.no_grad():
    for it in range(int(os.environ.get("ITERS", "6"))):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.BACKEND.dense_edge_conv(x, idx, 1, k, conv.mlps, out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
flop = B * n * k * 3168.0
print("dec ms: %s   min %.3f median %.3f   (%.0f TFLOP/s of the block's 3168 FLOP per edge)"
      % (" ".join("%.3f" % t for t in ts[:6]), min(ts), sorted(ts)[len(ts) // 2], flop / min(ts) / 1e9))
