"""linear_wide (up_layer1's per-point half, 264 -> 128) on one level-4 chunk: time and error against fp64.
Run with TPU3_SPLIT_BF16=0 and =1 (the flag is read once per process)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
L = importlib.import_module("3pu_pytorch_amd._lib")
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
w = torch.randn((128, 265), device=dev, generator=g) / 16
b = torch.randn((128,), device=dev, generator=g)
print("TPU3_SPLIT_BF16=%s XS=%s" % (os.environ.get("TPU3_SPLIT_BF16", "0"), os.environ.get("XS", "264")))


def wide(x, w, b):
    """the C ABI directly: rows with a stride (XS != 264: a padded feature buffer)"""
    if x.is_contiguous():
        return ops.BACKEND.linear_wide(x, w, b)
    m = x.size(0)
    y = torch.empty((m, 128), dtype=torch.float32, device=x.device)
    L.check(L.lib().tpu3_linear_wide_f32(L.stream_of(x), m, 264, 128, L.ptr(x), x.stride(0), L.ptr(w), w.stride(0),
                                         L.ptr(b), L.ptr(y), 128), "tpu3_linear_wide_f32")
    return y

for M in (1, 17, 129, 4992, 3840 * 312 + 5):
    XS = int(os.environ.get("XS", "264"))
    x = (torch.randn((M, XS), device=dev, generator=g) * torch.rand((M, 1), device=dev, generator=g) * 4)[:, :264]
    x[:, 100:120] = 0
    y = wide(x, w[:, :264], b)
    ref = (x.double() @ w[:, :264].double().t() + b.double())
    err = (y.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    y32 = torch.nn.functional.linear(x, w[:, :264], b)
    err32 = (y32.double() - ref).abs().max().item()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); wide(x, w[:, :264], b); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = min(ts)
    print("M %8d  max err vs fp64 %.3e (torch fp32 GEMM %.3e; |y| max %.1f)   %.3f ms  %.1f TF/s  %.0f GB/s"
          % (M, err, err32, scale, t, M * 264 * 128 * 2 / t / 1e9, M * (264 + 128) * 4 / t / 1e6))
