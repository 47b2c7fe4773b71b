"""Which Python lines of the package issue the ATen calls of one C3 training step (GPU box): a TorchDispatchMode
counts (op, innermost package frame) over one eager step."""
import collections, importlib, os, sys, traceback, types
import torch
from torch.utils._python_dispatch import TorchDispatchMode
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
for _ in range(2):
    model.set_input(inp, ratio, label_pc=lab); model.optimize()
torch.cuda.synchronize()
VIEW = ("view", "reshape", "expand", "transpose", "permute", "slice", "select", "unsqueeze", "squeeze", "detach",
        "alias", "as_strided", "t.default", "_unsafe_view", "unbind", "split", "empty", "size", "stride", "is_", "numel",
        "narrow", "sym_", "_local_scalar", "lift_fresh", "item", "chunk", "unfold", "new_empty", "result_type")
counts = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types_, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEW):
            site = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "3pu_pytorch_amd" in fr.filename:
                    site = "%s:%d %s" % (fr.filename.split("3pu_pytorch_amd/")[-1], fr.lineno, fr.name)
                    break
            shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), None)
            counts[(name.replace("aten.", ""), site, shp)] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    model.set_input(inp, ratio, label_pc=lab); model.optimize()
torch.cuda.synchronize()
by_site = collections.Counter()
for (name, site, shp), c in counts.items():
    by_site[(site, name)] += c
print("%d device-touching ATen calls in one step" % sum(counts.values()))
for (site, name), c in sorted(by_site.items(), key=lambda kv: -kv[1])[:int(os.environ.get("TOP", "120"))]:
    shapes = sorted({str(s) for (n, st, s), _ in counts.items() if n == name and st == site})[:3]
    print("%4d x %-26s %-52s %s" % (c, name[:26], site[-52:], " ".join(shapes)[:70]))
