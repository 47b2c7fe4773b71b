"""One cloud through pipeline.upsample (the reference's test() loop handles one cloud at a time): wall time per call
and, under rocprofv3 --kernel-trace --stats, where it goes (GPU box)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ops = bench.pkg("network.operations")
pipe = bench.pkg("pipeline")
ups = bench.pkg("network.upsampler")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev).eval()
cloud = bench.poisson_sphere(0, 5000, dev, ops)
for final in (True, False):
    ts = []
    for it in range(int(os.environ.get("ITERS", "5"))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            pipe.upsample(net, cloud, 312, 16, 3, final_fps=final, check_small=False)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("1 cloud, final_fps=%s: %s ms" % (final, " ".join("%.1f" % t for t in ts)))
# stage times of the final resampling alone (synchronised between stages)
with torch.no_grad():
    merged = pipe.upsample(net, cloud, 312, 16, 3, final_fps=False, check_small=False).transpose(2, 1).contiguous()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx = ops.fps(merged, 80000)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out = torch.gather(merged, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).transpose(2, 1).contiguous()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("final FPS of the real merged cloud %s: fps %.1f ms, gather %.2f ms" % (tuple(merged.shape), (t1 - t0) * 1e3, (t2 - t1) * 1e3))
