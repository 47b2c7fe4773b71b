"""Which ATen kernels are left in the C3 training step, and which Python lines launch them (GPU box).
torch.profiler over eager steps of tools/train_step_probe.py's setup: device time per (op, source line)."""
import importlib, os, re, sys, types
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ups = importlib.import_module("3pu_pytorch_amd.network.upsampler")
model_mod = importlib.import_module("3pu_pytorch_amd.model")
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(7)
ratio = int(os.environ.get("RATIO", "16"))
inp = torch.randn(32, 312, 3, generator=g)
inp = (inp / inp.norm(dim=2, keepdim=True)).transpose(2, 1).contiguous().to(dev)
lab = torch.randn(32, 312 * ratio, 3, generator=g)
lab = (lab / lab.norm(dim=2, keepdim=True)).transpose(2, 1).contiguous().to(dev)
torch.manual_seed(0)
tnet = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev)
model = model_mod.Model(tnet, "train", types.SimpleNamespace(lr_init=1e-3, ckpt=None, graph_steps=False))
for _ in range(3):
    model.set_input(inp, ratio, label_pc=lab); model.optimize()
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=False) as prof:
    for _ in range(N):
        model.set_input(inp, ratio, label_pc=lab); model.optimize()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=int(os.environ.get("DEPTH", "6")))
rows = []
for e in ka:
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt <= 0:
        continue
    if not (e.key.startswith("aten::") or e.key.startswith("Mem") or "Optimizer" in e.key):
        continue                                    # (kernel rows and autograd nodes repeat their ops' time)
    here = [s for s in e.stack if "3pu_pytorch_amd" in s or "tools/" in s]
    rows.append((dt / N, e.count / N, e.key, here[:3]))
rows.sort(key=lambda r: -r[0])
tot = sum(r[0] for r in rows)
print("device time per step: %.3f ms in %d (op, site) groups" % (tot / 1e3, len(rows)))
for dt, cnt, key, here in rows[:int(os.environ.get("TOP", "90"))]:
    print("%8.1f us %6.1f x  %-28s %s" % (dt, cnt, key[:28], " <- ".join(re.sub(r".*3pu_pytorch_amd/", "", h)[-60:] for h in here)))
