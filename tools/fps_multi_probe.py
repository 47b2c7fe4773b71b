"""Multi-sample FPS kernel (fm_main_kernel): samples per round, timing, and a bit-exact prefix check against
the plain resident algorithm's result on a smaller m (GPU box).  usage: fps_multi_probe.py [n] [m]"""
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
need = lib.tpu3_fps_workspace_bytes(1, n)
ws = torch.empty(need, dtype=torch.uint8, device=dev)
temp = torch.full((1, n), 1e10, device=dev)
idx = torch.zeros((1, m), dtype=torch.int32, device=dev)
prof = torch.zeros(64, dtype=torch.int64, device=dev)
rc = lib.tpu3_debug_fps_bucket_profile(None, n, m, x.data_ptr(), temp.data_ptr(), idx.data_ptr(), ws.data_ptr(), need,
                                       prof.data_ptr())
torch.cuda.synchronize()
p = prof.cpu().numpy()
print("rc %d  rounds %d  samples %d  (%.2f per round)  candidate overflows %d" % (rc, p[0], p[1], p[1] / max(1, p[0]), p[2]))
names = ["prune", "children", "flush", "refresh", "select-pre", "barrier", "post", "buckets"]
for w in range(4):
    print("wave %d per round: " % w + "  ".join("%s %.0f" % (names[i], p[8 + w * 8 + i] / max(1, p[0])) for i in range(8)))
print("capped rounds %d, tie rounds %d, mean candidates %.1f" % (p[2], p[3], p[4] / max(1, p[0])))
print("distinct picks:", int(idx.unique().numel()), "of", m)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    i2 = ops.fps(x, m)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("ops.fps %d -> %d: %.2f ms  (%.3f us per sample)" % (n, m, dt * 1e3, dt * 1e6 / m))
print("profile run == plain run:", bool(torch.equal(i2, idx)))
if os.environ.get("ORACLE"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle as orc
    mm = int(os.environ["ORACLE"])
    t0 = time.perf_counter(); ref, _ = orc.fps(x.cpu().numpy(), mm); print("oracle %.1f s" % (time.perf_counter() - t0))
    print("first %d picks bit-exact vs oracle:" % mm, bool((i2[:, :mm].cpu().numpy() == ref).all()))
