"""Multi-workgroup tile-form FPS (csrc/fps_cluster.hip) against the single-workgroup kernel: bit-exact picks and final
temp, rounds / samples / tie exchanges / poll sweeps, timing per G (GPU box).
usage: fps_cluster_probe.py [n] [m]    REAL=1: the reference's merged C2 cloud    GS="2,4,8,16"   B=<clouds>"""
import ctypes, importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("3pu_pytorch_amd._lib")
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
lib = L.lib()
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 239616
m = int(sys.argv[2]) if len(sys.argv) > 2 else 80000
B = int(os.environ.get("B", "1"))
g = torch.Generator().manual_seed(0)
x = torch.randn(B, n, 3, generator=g)
x = (x / x.norm(dim=2, keepdim=True)).to(dev)
if os.environ.get("REAL"):
    gg = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "c2_x16.npz"))
    x = torch.from_numpy(np.ascontiguousarray(gg["pred_concat"].transpose(0, 2, 1))).to(dev).repeat(B, 1, 1)
    n = x.shape[1]
if os.environ.get("DUP"):          # duplicated points: the tie paths
    x[:, n // 2:] = x[:, :n - n // 2]


def run(G, reps=3):
    lib.tpu3_debug_fps_cluster(G)
    cl = ctypes.c_int(0)
    kind = lib.tpu3_debug_fps_plan(B, n, m, ctypes.byref(cl))
    stats = torch.zeros(32, dtype=torch.int64, device=dev)
    lib.tpu3_debug_fps_tile_stats(ctypes.c_void_p(stats.data_ptr()))
    idx = ops.fps(x, m)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        i2 = ops.fps(x, m)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return kind, cl.value, idx, stats.cpu().numpy(), min(ts), bool(torch.equal(i2, idx))


kind, _, ref, st, t_ref, det = run(0)
print("single workgroup (plan %d): %.2f ms, rounds %d samples %d" % (kind, t_ref, st[0], st[1]))
for G in [int(v) for v in os.environ.get("GS", "2,4,8,16").split(",")]:
    kind, cl, idx, st, t, det = run(G)
    if kind != 6:
        print("G=%d: plan %d (cluster form not taken)" % (G, kind))
        continue
    bad = (idx != ref).nonzero()
    print("G=%2d: %.2f ms (%.2fx)  rounds %d samples %d (%.1f/round) tie exchanges %d, %.2f sweeps/round, faults %d; "
          "bit-exact %s%s, deterministic %s, cluster faults %d"
          % (cl, t, t_ref / t, st[0], st[1], st[1] / max(1, st[0]), st[2], st[3] / max(1, st[0]), st[4],
             bad.numel() == 0, "" if bad.numel() == 0 else " first mismatch at %s" % bad[0].tolist(), det,
             lib.tpu3_fps_cluster_faults(1)))
    R = max(1, st[0])
    print("        wave 0 cycles per round: phase 1 %.0f | phase 2 %.0f | collect %.0f | local select + publish %.0f | poll %.0f "
          "| merge + rank %.0f | clearance %.0f ; listed locally %.1f" % tuple(st[8 + k] / R for k in range(8)))
    print("        merge split (wave 0): histogram + barrier %.0f | scan + seats + barrier %.0f | partial ranks + barrier %.0f | ties + picks %.0f"
          % tuple(st[16 + k] / R for k in range(4)))
lib.tpu3_debug_fps_cluster(-1)
