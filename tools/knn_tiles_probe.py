"""Pruned inter-level kNN (csrc/knn_tiles.hip) on the bench's own level inputs: tiles searched per wave, time against
the brute-force kernel (GPU box).  The Level calls are intercepted, so the data are the real patches / previous clouds."""
import ctypes, importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ops, pipe, ups = bench.pkg("network.operations"), bench.pkg("pipeline"), bench.pkg("network.upsampler")
L = importlib.import_module("3pu_pytorch_amd._lib")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev).eval()
clouds = torch.cat([bench.poisson_sphere(i, 5000, dev, ops) for i in range(4)])
calls = []
real = ops.BACKEND.knn
def spy(k, q, p, unique, layout=None, want_dist=True, want_grouped=True, unique_cache=None):
    if k == 5 and q.size(-1) == 3:
        calls.append((q.clone(), p.clone(), {kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in (layout or {}).items()}))
    return real(k, q, p, unique, layout, want_dist, want_grouped, unique_cache=unique_cache)
ops.BACKEND.knn = spy
with torch.no_grad():
    pipe.upsample(net, clouds, 312, 16, 3, final_fps=False)
del ops.BACKEND.knn
print("%d inter-level searches captured" % len(calls))
for q, p, lay in calls:
    res = {}
    for tiles in (True, False):
        ops.BACKEND.knn_tiles = tiles
        cache = {}
        real(5, q, p, True, lay, False, False, unique_cache=cache)          # builds the de-dup state (+ tiles)
        st = torch.zeros(4, dtype=torch.int32, device=dev)
        if tiles:
            L.lib().tpu3_debug_knn_tiles_stats(ctypes.c_void_p(st.data_ptr()))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx, _, _ = real(5, q, p, True, lay, False, False, unique_cache=cache)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[tiles] = (idx, dt, st.cpu().numpy())
    ops.BACKEND.knn_tiles = True
    w = res[True][2]
    print("queries %s  points %s: tiles %.2f ms (%.1f of %.0f tiles searched per wave, %.1f tested per query), brute force %.2f ms, equal %s"
          % (tuple(q.shape), tuple(p.shape), res[True][1] * 1e3, w[1] / max(1, w[0]), w[3] / max(1, w[0]), w[2] / max(1, w[0]),
             res[False][1] * 1e3, bool(torch.equal(res[True][0], res[False][0]))))

# a whole cloud's previous level, as Net.forward (not the patch pipeline) searches it: 4 clouds, 49 920 rows each (every
# point ~2.5 times), 99 840 queries per cloud in 312-point patches
g = torch.Generator().manual_seed(1)
base = torch.randn(4, 20000, 3, generator=g)
base = base / base.norm(dim=2, keepdim=True)
pts = torch.gather(base, 1, torch.randint(0, 20000, (4, 49920, 1), generator=g).expand(-1, -1, 3)).to(dev)
qc = torch.randn(4, 99840, 3, generator=g)
qc = (qc / qc.norm(dim=2, keepdim=True)).to(dev)
seeds = qc[:, ::312]
_, _, patches = real(312, seeds.contiguous(), qc, False, None, False, True)       # (4,320,312,3): kNN-ordered patches
q = patches.reshape(4 * 320, 312, 3).contiguous()
lay = dict(pts_of=torch.arange(4, dtype=torch.int32, device=dev).repeat_interleave(320),
           grp=torch.arange(4, dtype=torch.int32, device=dev).repeat_interleave(320), groups=4)
res = {}
for tiles in (True, False):
    ops.BACKEND.knn_tiles = tiles
    cache = {}
    real(5, q, pts, True, lay, False, False, unique_cache=cache)
    st = torch.zeros(4, dtype=torch.int32, device=dev)
    if tiles:
        L.lib().tpu3_debug_knn_tiles_stats(ctypes.c_void_p(st.data_ptr()))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx, _, _ = real(5, q, pts, True, lay, False, False, unique_cache=cache)
    torch.cuda.synchronize(); res[tiles] = (idx, time.perf_counter() - t0, st.cpu().numpy())
ops.BACKEND.knn_tiles = True
w = res[True][2]
print("whole clouds: queries %s  points %s: tiles %.2f ms (%.1f of %.0f tiles searched per wave), brute force %.2f ms, equal %s"
      % (tuple(q.shape), tuple(pts.shape), res[True][1] * 1e3, w[1] / max(1, w[0]), w[3] / max(1, w[0]), res[False][1] * 1e3,
         bool(torch.equal(res[True][0], res[False][0]))))
