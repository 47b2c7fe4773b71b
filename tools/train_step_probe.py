"""Config C3 training step as bench.py times it (B = 32 patches of 312 points, inputs resident): eager and
hipGraph, ratio 16 and 4 (GPU box)."""
import importlib, os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ups = importlib.import_module("3pu_pytorch_amd.network.upsampler")
model_mod = importlib.import_module("3pu_pytorch_amd.model")
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(7)
inp = torch.randn(32, 312, 3, generator=g)
inp = (inp / inp.norm(dim=2, keepdim=True)).transpose(2, 1).contiguous().to(dev)
for ratio in [int(v) for v in os.environ.get("RATIOS", "16,4").split(",")]:
    lab = torch.randn(32, 312 * ratio, 3, generator=g)
    lab = (lab / lab.norm(dim=2, keepdim=True)).transpose(2, 1).contiguous().to(dev)
    for mode in os.environ.get("MODES", "eager,graph").split(","):
        torch.manual_seed(0)
        tnet = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev)
        model = model_mod.Model(tnet, "train", types.SimpleNamespace(lr_init=1e-3, ckpt=None, graph_steps=(mode == "graph")))
        for _ in range(3):
            model.set_input(inp, ratio, label_pc=lab); model.optimize()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = int(os.environ.get("STEPS", "10"))
        for _ in range(n):
            model.set_input(inp, ratio, label_pc=lab); model.optimize()
        torch.cuda.synchronize()
        print("ratio %2d %-5s: %.2f ms per step" % (ratio, mode, (time.perf_counter() - t0) / n * 1e3), flush=True)
