"""fp16-operand DenseEdgeConv against the fp32 flavour over shapes (GPU box): max / mean |diff| and timing."""
import importlib, sys, time, torch
sys.path.insert(0, ".")
importlib.import_module("3pu_pytorch_amd")
layers = importlib.import_module("3pu_pytorch_amd.network.layers")
dev = torch.device("cuda", 0)
for P, N, k in [(1, 312, 16), (1, 312, 32), (1, 2700, 16), (1, 3000, 32), (1, 3000, 16), (2, 2800, 48), (3840, 312, 32), (234, 1024, 32)]:
    torch.manual_seed(1)
    blk = layers.DenseEdgeConv(24, growth_rate=12, n=3, k=k).to(dev)
    for m in blk.mlps:
        torch.nn.init.xavier_uniform_(m.weight); torch.nn.init.uniform_(m.bias, -0.5, 0.5)
    x = torch.randn(P, N, 24, device=dev)
    with torch.no_grad():
        res = {}
        for prec in ("f32", "f16"):
            blk.mlp_precision = prec
            y, idx = blk.forward_cl(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                blk.forward_cl(x)
            torch.cuda.synchronize(); res[prec] = (y, (time.perf_counter() - t0) / 5 * 1e3)
        d = (res["f16"][0] - res["f32"][0]).abs()
        print("P=%d N=%d k=%d: max %.4g mean %.4g  (graph+block: f32 %.3f ms, f16 %.3f ms)" % (
            P, N, k, float(d.max()), float(d.mean()), res["f32"][1], res["f16"][1]))
