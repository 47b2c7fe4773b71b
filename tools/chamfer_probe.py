"""nm-distance (Chamfer) forward / backward kernels: time and fraction of the fp32 vector peak
(16 FLOP per pair and direction pair = 2*B*n*m*8)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("3pu_pytorch_amd")
ml = importlib.import_module("3pu_pytorch_amd.network.model_loss")
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
for B, n, m in ((32, 624, 624), (32, 4992, 4992), (1, 80000, 80000), (1, 1280000, 1280000)):
    a = torch.rand((B, n, 3), device=dev, generator=g, requires_grad=True)
    b = torch.rand((B, m, 3), device=dev, generator=g)
    def fwd():
        return ml.nndistance(a, b)
    def timeit(fn, reps=3):
        fn(); torch.cuda.synchronize(); ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        return min(ts)
    with torch.no_grad():
        tf = timeit(fwd)
    def fb():
        d1, _, d2, _ = ml.nndistance(a, b)
        (d1.mean() + d2.mean()).backward()
    tb = timeit(fb)
    flop = 2.0 * B * n * m * 8
    print("B=%2d n=m=%7d  forward %8.3f ms = %5.1f TFLOP/s (%.2f of 157.3)   forward+backward %8.3f ms"
          % (B, n, tf, flop / tf / 1e9, flop / tf / 1e9 / 157.3, tb))
