"""Time every FPS call of one real 8-cloud network step (real merged patch sets, ragged sizes)."""
import importlib, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ops, pipe, ups = bench.pkg("network.operations"), bench.pkg("pipeline"), bench.pkg("network.upsampler")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev).eval()
clouds = torch.cat([bench.poisson_sphere(i, 5000, dev, ops) for i in range(8)], dim=0)
orig = ops.BACKEND.fps
log = []
def timed(xyz, npoint, n_arr=None, m_arr=None):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(xyz, npoint, n_arr, m_arr)
    e1.record()
    torch.cuda.synchronize()
    log.append((tuple(xyz.shape), npoint, n_arr is not None, m_arr is not None, e0.elapsed_time(e1)))
    if os.environ.get("DUMP") and xyz.shape[1] == 24960:
        torch.save(dict(xyz=xyz.cpu(), n_arr=None if n_arr is None else n_arr.cpu()), os.environ["DUMP"])
    return r
ops.BACKEND.fps = timed
with torch.no_grad():
    for it in range(2):
        log.clear()
        pipe.upsample(net, clouds, 312, 16, 3, final_fps=False)
for l in log:
    print("xyz %-18s m=%5d ragged_n=%d ragged_m=%d  %8.3f ms" % l)
