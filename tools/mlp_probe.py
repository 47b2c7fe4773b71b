"""Time the regressor-tail and prep-convolution kernels on one level-4 chunk (3840 patches x 312 points)."""
import importlib, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
dev = torch.device("cuda:0")
M = 3840 * 312
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn((M, 128), device=dev, generator=g)
c = torch.randn((2, 128), device=dev, generator=g)
w2 = torch.randn((128, 128), device=dev, generator=g) / 11
w3 = torch.randn((64, 128), device=dev, generator=g) / 11
w4 = torch.randn((3, 64), device=dev, generator=g) / 8
b2, b3, b4 = (torch.randn((n,), device=dev, generator=g) for n in (128, 64, 3))
res = torch.randn((M, 3), device=dev, generator=g)
feat = torch.randn((M, 264), device=dev, generator=g)
wp = torch.randn((24, 204), device=dev, generator=g) / 14
bp = torch.randn((24,), device=dev, generator=g)
def timeit(fn, name, flop):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = min(ts)
    print("%-28s %7.3f ms  %6.1f TFLOP/s" % (name, t, flop / t / 1e9))
timeit(lambda: ops.BACKEND.regress_tail(a, c, w2, b2, w3, b3, w4, b4, res), "regress_tail", M * 2 * 49536.0)
timeit(lambda: ops.BACKEND.linear_small(feat[:, 60:], wp, bp, True), "linear_small 204->24", M * 204 * 24 * 2.0)
timeit(lambda: torch.relu_(torch.nn.functional.linear(feat[:, 60:], wp, bp)), "torch 204->24 + relu", M * 204 * 24 * 2.0)
timeit(lambda: feat[:, 60:].contiguous(), "slice copy (0.98 GB r + w)", 0.0)
timeit(lambda: feat.clone(), "full clone (1.27 GB r + w)", 0.0)
full = torch.randn((M, 204), device=dev, generator=g)
timeit(lambda: ops.BACKEND.linear_small(full, wp, bp, True), "linear_small contiguous 204", M * 204 * 24 * 2.0)
