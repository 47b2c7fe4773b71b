"""Time the fused DenseEdgeConv kernel on one level-sized chunk (3840 patches x 312 points, k = 32)."""
import importlib, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("3pu_pytorch_amd")
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
layers = importlib.import_module("3pu_pytorch_amd.network.layers")
dev = torch.device("cuda:0")
B, n, C, k = int(os.environ.get("PATCHES", "3840")), int(os.environ.get("N", "312")), 24, int(os.environ.get("K", "32"))
torch.manual_seed(0)
conv = layers.DenseEdgeConv(24, growth_rate=12, n=3, k=k).to(dev).eval()
x = torch.rand((B, n, C), device=dev)
idx = torch.randint(0, n, (B, n, k + 1), device=dev, dtype=torch.int32)
out = torch.empty((B, n, 60), device=dev)
PACK = os.environ.get("PACK", "1") != "0"      # packed operand tables (the network's path) or built per workgroup
pack = ops.BACKEND.dense_edge_conv_pack(conv.mlps) if PACK else None
ts = []
with torch.no_grad():
    for it in range(int(os.environ.get("ITERS", "6"))):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.BACKEND.dense_edge_conv(x, idx, 1, k, conv.mlps, out, pack=pack)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
if os.environ.get("FOLD"):      # the fold form: FOLD = 24 / 48 / 72 outputs of the next prep convolutions
    fn = int(os.environ["FOLD"])
    fw, fb = torch.randn((fn, 60), device=dev) * 0.1, torch.randn((fn,), device=dev) * 0.1
    acc = torch.zeros((B, n, 48), device=dev)
    xn = torch.empty((B, n, 24), device=dev)
    fpack = ops.BACKEND.dense_edge_conv_pack(conv.mlps, fw) if PACK else None
    tf = []
    with torch.no_grad():
        for it in range(int(os.environ.get("ITERS", "6"))):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.BACKEND.dense_edge_conv_fold(x, idx, 1, k, conv.mlps, out, fw, fb, acc, 0, 0, xn, pack=fpack)
            e1.record()
            torch.cuda.synchronize()
            tf.append(e0.elapsed_time(e1))
    print("dec fold %d ms: %s   min %.3f median %.3f" % (fn, " ".join("%.3f" % t for t in tf[:6]), min(tf),
                                                        sorted(tf)[len(tf) // 2]))
    print("digest fold: out %.9e xn %.9e acc %.9e" % (out.double().sum().item(), xn.double().sum().item(),
                                                       acc.double().sum().item()))
flop = B * n * k * 3168.0
print("digest: %.9e" % out.double().sum().item())
print("dec ms: %s   min %.3f median %.3f   (%.0f TFLOP/s of the block's 3168 FLOP per edge)"
      % (" ".join("%.3f" % t for t in ts[:6]), min(ts), sorted(ts)[len(ts) // 2], flop / min(ts) / 1e9))
