"""Three-level tile-form FPS against the 64-point-bucket kernel on awkward inputs (GPU box): run once with TPU3_FL=1 and
once without; the second run compares.  Clouds: quadruplicated points, Gaussian blobs, a noisy line, a lattice."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
n, m = 300000, 90000
clouds = {}
base = torch.rand(1, n // 4, 3, generator=g)
clouds["dup4"] = base.repeat(1, 4, 1)[:, torch.randperm(n, generator=g)]
c = torch.randn(8, 3, generator=g)
clouds["blobs"] = (c[torch.randint(0, 8, (n,), generator=g)] + 0.05 * torch.randn(n, 3, generator=g)).unsqueeze(0)
t = torch.rand(n, 1, generator=g)
clouds["line"] = (t * torch.tensor([[1.0, 2.0, -0.5]]) + 1e-3 * torch.randn(n, 3, generator=g)).unsqueeze(0)
k = int(round(n ** (1 / 3))) + 1
gg = torch.stack(torch.meshgrid(*[torch.arange(k)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n].float() * 0.01
clouds["lattice"] = gg[torch.randperm(gg.size(0), generator=g)].unsqueeze(0)
tag = os.environ.get("TPU3_FL", "default")
os.makedirs("/tmp/l3", exist_ok=True)
for name, x in clouds.items():
    x = x.to(dev).contiguous()
    mm = min(m, x.size(1) // 3)
    idx = ops.fps(x, mm).cpu().numpy()
    np.save("/tmp/l3/%s_%s.npy" % (name, tag), idx)
    other = "/tmp/l3/%s_%s.npy" % (name, "1" if tag == "default" else "default")
    if os.path.exists(other):
        o = np.load(other)
        print("%-8s n=%d m=%d: equal %s" % (name, x.size(1), mm, bool((o == idx).all())), flush=True)
    else:
        print("%-8s n=%d m=%d: saved (%s)" % (name, x.size(1), mm, tag), flush=True)
