"""Per-level resampling FPS: the dispatcher's choice (register-resident rl_main_kernel) against the tile form
(fl_main_kernel) and the cluster form (fc_main_kernel) on REAL level data -- VERDICT r4 item 2.

One 16x pass over 4 clouds is run with the backend's fps() spied; the merged sets of every level (6240 / 12 480 /
24 960 points per outer patch, 48 sets per cloud) are replayed as 48 sets (one cloud), 192 (a bench sub-batch) and
1536 (the bench's 32 clouds).  Run once per configuration (the switches are read once per process):
    python tools/fps_level_dispatch_probe.py                                  # shipped dispatch
    TPU3_FPS_FORCE_TILE=1 TPU3_FPS_CLUSTER=0 python tools/fps_level_dispatch_probe.py     # fl_main, one workgroup per set
    TPU3_FPS_FORCE_TILE=1 TPU3_FPS_CLUSTER=2 SETS=48 python tools/...                     # two workgroups per set (48 sets only)
"""
import ctypes, hashlib, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
L = bench.pkg("_lib")
ops = bench.pkg("network.operations")
pipe = bench.pkg("pipeline")
ups = bench.pkg("network.upsampler")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev).eval()
g = torch.Generator().manual_seed(0)
p = torch.randn(4, 5000, 3, generator=g)
clouds = (p / p.norm(dim=2, keepdim=True)).transpose(2, 1).contiguous().to(dev)
seen = {}
orig = ops.BACKEND.fps
force = os.environ.get("TPU3_FPS_FORCE_TILE", "0")
os.environ["TPU3_FPS_FORCE_TILE"] = force


def spy(xyz, npoint, n_arr=None, m_arr=None):
    if 6000 <= xyz.size(1) <= 25600 and xyz.size(1) not in seen:
        seen[xyz.size(1)] = (xyz.clone(), npoint, None if n_arr is None else n_arr.clone(), None if m_arr is None else m_arr.clone())
    return orig(xyz, npoint, n_arr, m_arr)


ops.BACKEND.fps = spy
with torch.no_grad():
    pipe.upsample(net, clouds, 312, 16, 3, final_fps=False, check_small=False)
ops.BACKEND.fps = orig
torch.cuda.synchronize()
sets = [int(v) for v in os.environ.get("SETS", "48,192,1536").split(",")]
print("TPU3_FPS_FORCE_TILE=%s TPU3_FPS_CLUSTER=%s" % (force, os.environ.get("TPU3_FPS_CLUSTER", "(default)")))
for n, (x, m, na, ma) in sorted(seen.items()):
    for nb in sets:
        reps = (nb + x.size(0) - 1) // x.size(0)
        xs = x.repeat(reps, 1, 1)[:nb].contiguous()
        nas = None if na is None else na.repeat(reps)[:nb].contiguous()
        mas = None if ma is None else ma.repeat(reps)[:nb].contiguous()
        cl = ctypes.c_int(0)
        plan = L.lib().tpu3_debug_fps_plan(nb, n, m, ctypes.byref(cl))
        ts = []
        for it in range(4):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            idx = orig(xs, m, nas, mas)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        digest = hashlib.sha256(idx[:48].cpu().numpy().tobytes()).hexdigest()[:12]
        faults = L.lib().tpu3_fps_cluster_faults(1)
        print("n=%5d m=%4d sets=%4d plan %d (G=%d): %8.3f ms = %.3f us per sample and set  [idx digest %s, faults %d]"
              % (n, m, nb, plan, cl.value, min(ts), min(ts) * 1e3 / m, digest, faults))
