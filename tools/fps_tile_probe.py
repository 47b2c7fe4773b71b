"""Tile-form FPS kernel (fl_main_kernel): rounds, samples per round, timing, bit-exact check of a prefix against the
oracle (GPU box).  usage: fps_tile_probe.py [n] [m]   ORACLE=<m'> checks the first m' picks."""
import ctypes, importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("3pu_pytorch_amd._lib")
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
lib = L.lib()
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 239616
m = int(sys.argv[2]) if len(sys.argv) > 2 else 80000
g = torch.Generator().manual_seed(0)
x = torch.randn(1, n, 3, generator=g)
x = (x / x.norm(dim=2, keepdim=True)).to(dev)
if os.environ.get("REAL"):
    gg = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "c2_x16.npz"))
    x = torch.from_numpy(np.ascontiguousarray(gg["pred_concat"].transpose(0, 2, 1))).to(dev)
    n = x.shape[1]
stats = torch.zeros(16, dtype=torch.int64, device=dev)
lib.tpu3_debug_fps_tile_stats(ctypes.c_void_p(stats.data_ptr()))
idx = ops.fps(x, m)
torch.cuda.synchronize()
st = stats.cpu().numpy()
print("rounds %d samples %d (%.2f per round), overflow rounds %d, tie rounds %d" % (st[0], st[1], st[1] / max(1, st[0]), st[2], st[3]))
print("wave 0 per round: apply %.0f cycles (%.1f tile visits), collect %.0f, rank %.0f" % (st[4] / max(1, st[0]), st[7] / max(1, st[0]), st[5] / max(1, st[0]), st[6] / max(1, st[0])))
print("   apply split: phase 1 %.0f, barrier %.0f, phase 2 %.0f, barrier %.0f" % tuple(st[8 + i] / max(1, st[0]) for i in range(4)))
print("   per round: %.2f collect passes, %.0f listed, %.1f selected, %.2f of the rounds cut by clearance" % tuple(st[12 + i] / max(1, st[0]) for i in range(4)))
print("distinct picks:", int(idx.unique().numel()), "of", m)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    i2 = ops.fps(x, m)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("ops.fps %d -> %d: %.2f ms  (%.3f us per sample)" % (n, m, dt * 1e3, dt * 1e6 / m))
print("deterministic:", bool(torch.equal(i2, idx)))
if os.environ.get("ORACLE"):
    from oracle import oracle as orc
    mm = int(os.environ["ORACLE"])
    ref, _ = orc.fps(x.cpu().numpy(), mm)
    bad = np.where(i2[0, :mm].cpu().numpy() != ref[0])[0]
    print("first %d picks bit-exact vs oracle: %s%s" % (mm, len(bad) == 0, "" if len(bad) == 0 else " first mismatch at %d" % bad[0]))
