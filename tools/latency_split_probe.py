"""One cloud through pipeline.upsample: host enqueue time vs device time (is the 1-cloud latency launch-bound?)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
pipe = importlib.import_module("3pu_pytorch_amd.pipeline")
ups = importlib.import_module("3pu_pytorch_amd.network.upsampler")
wl = importlib.import_module("3pu_pytorch_amd.utils.workloads")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev).eval()
cloud = wl.poisson_sphere(0, 5000, dev)
for final in (False, True):
    for it in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        with torch.no_grad():
            pipe.upsample(net, cloud, 312, 16, 3, final_fps=final, check_small=False, optimistic_graph=True)
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print("final_fps=%s: host enqueue %.1f ms, device span %.1f ms, wall %.1f ms" % (final, (t1 - t0) * 1e3, e0.elapsed_time(e1), (t2 - t0) * 1e3))
