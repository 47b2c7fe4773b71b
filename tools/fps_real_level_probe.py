"""Per-level resampling FPS on REAL level data: one bench step is run with the backend's fps() spied, the
merged sets of every level (6240 / 12 480 / 24 960 points per outer patch) are replayed alone with the
rounds / samples probe of the register-resident kernel (GPU box)."""
import importlib, os, sys
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
clouds = bench.make_clouds(4, 5000, dev) if hasattr(bench, "make_clouds") else None
if clouds is None:
    g = torch.Generator().manual_seed(0)
    p = torch.randn(4, 5000, 3, generator=g)
    clouds = (p / p.norm(dim=2, keepdim=True)).transpose(2, 1).contiguous().to(dev)
seen = {}
orig = ops.BACKEND.fps
def spy(xyz, npoint, n_arr=None, m_arr=None):
    if 6000 <= xyz.size(1) <= 25600 and xyz.size(1) not in seen:
        seen[xyz.size(1)] = (xyz.clone(), npoint, None if n_arr is None else n_arr.clone(), None if m_arr is None else m_arr.clone())
    return orig(xyz, npoint, n_arr, m_arr)
ops.BACKEND.fps = spy
with torch.no_grad():
    pipe.upsample(net, clouds, 312, 16, 3, final_fps=False, check_small=False)
ops.BACKEND.fps = orig
torch.cuda.synchronize()
stats = torch.zeros(52, dtype=torch.int64, device=dev)
for n, (x, m, na, ma) in sorted(seen.items()):
    for nb in (x.size(0), 1):
        xs = x[:nb].contiguous()
        ts = []
        for it in range(3):
            stats.zero_()
            L.lib().tpu3_debug_fps_level_stats(stats.data_ptr())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(xs, m, None if na is None else na[:nb].contiguous(), None if ma is None else ma[:nb].contiguous())
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        r, s = int(stats[0]), int(stats[1])
        print("n=%5d m=%4d sets=%3d: %7.3f ms  rounds %d samples %d (%.2f per round, %.2f us per round)"
              % (n, m, nb, min(ts), r, s, s / max(1, r), min(ts) * 1e3 / max(1, r)))
        pc = stats.cpu().numpy()
        if pc[14:].any():
            print("    apply cycles per round, waves 0..15: " + " ".join("%.0f" % (pc[14 + 2 * w] / max(1, r)) for w in range(16)))
            print("    sample updates per round, waves 0..15: " + " ".join("%.2f" % (pc[15 + 2 * w] / max(1, r)) for w in range(16)))
        if pc[46:].any():
            print("    wave 0 ranking per round: headers+live %.0f  rank loops %.0f  permute+clearance %.0f  | candidates %.1f"
                  % (pc[46] / max(1, r), pc[47] / max(1, r), pc[48] / max(1, r), pc[50] / max(1, r)))
        if pc[2:14].any():
            for w in range(2):
                print("    wave %d cycles per round: " % w + "  ".join(
                    "%s %.0f" % (nm, pc[2 + w * 6 + i] / max(1, r)) for i, nm in enumerate(("apply", "select", "barrier1", "rank", "barrier2", "-"))))
