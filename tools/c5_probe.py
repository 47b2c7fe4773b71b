"""Config C5 (stress): one 80 000-point cloud, num_point = 1024 (234 outer patches), 16x -> 1.28 M points."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ops, pipe, ups = bench.pkg("network.operations"), bench.pkg("pipeline"), bench.pkg("network.upsampler")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev).eval()
cloud = bench.poisson_sphere(0, 80000, dev, ops)
mode = os.environ.get("C5_MODE", "f32")             # f32 | f16 (fp16 operands) | f16a (+ fp16 feature buffers)
if mode != "f32":
    net.set_mlp_precision("f16", activations="f16" if mode == "f16a" else "f32")
print("mode", mode)
for final in (False, True):
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = pipe.upsample(net, cloud, 1024, 16, 3, final_fps=final)
        torch.cuda.synchronize(); t = time.perf_counter() - t0
    print("C5 final_fps=%s: %s in %.1f ms -> %.2f M points/s, peak %.1f GB"
          % (final, tuple(out.shape), t * 1e3, 80000 * 16 / t / 1e6, torch.cuda.max_memory_allocated() / 2**30))
assert torch.isfinite(out).all()
