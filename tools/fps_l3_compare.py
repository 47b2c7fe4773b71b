"""Three-level tile-form FPS against the 64-point-bucket three-level kernel (TPU3_FL=1), all picks (GPU box).
usage: fps_l3_compare.py [n] [m]; run once per TPU3_FL value, the picks go to gpurun_out/l3_<FL>.npy"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3833856
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1280000
g = torch.Generator().manual_seed(1)
x = torch.rand(1, n, 3, generator=g)
x[:, :, 2] *= 0.3 + 0.2 * torch.sin(6 * x[:, :, 0])        # a curved sheet of varying thickness
x = x.to(dev)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx = ops.fps(x, m)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("TPU3_FL=%s fps %d -> %d: %.1f ms" % (os.environ.get("TPU3_FL", "default"), n, m, dt * 1e3), flush=True)
tag = os.environ.get("TPU3_FL", "default")
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/l3_%s.npy" % tag, idx.cpu().numpy())
other = "gpurun_out/l3_%s.npy" % ("1" if tag == "default" else "default")
if os.path.exists(other):
    o = np.load(other)
    bad = np.where(o != idx.cpu().numpy())
    print("all %d picks equal: %s%s" % (m, len(bad[0]) == 0, "" if len(bad[0]) == 0 else " first mismatch at %d" % bad[1][0]))
