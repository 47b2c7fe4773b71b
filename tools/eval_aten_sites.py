"""Which Python lines of the package issue the ATen calls of ONE cloud's 16x upsampling (eval path, GPU box): a
TorchDispatchMode counts (op, innermost package frame) over one eager `pipeline.upsample` call of one C2 cloud -- the
launches between the hand-written kernels that the one-cloud latency pays a dispatch gap for."""
import collections
import importlib
import os
import sys
import traceback

import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ups = importlib.import_module("3pu_pytorch_amd.network.upsampler")
pipe = importlib.import_module("3pu_pytorch_amd.pipeline")
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
work = importlib.import_module("3pu_pytorch_amd.utils.workloads")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev).eval()
cloud = work.poisson_sphere(0, 5000, dev, ops)
for _ in range(2):
    pipe.upsample(net, cloud, 312, 16, 3, check_small=False, optimistic_graph=True)
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


with Mode(), torch.no_grad():
    pipe.upsample(net, cloud, 312, 16, 3, check_small=False, optimistic_graph=True)
torch.cuda.synchronize()
by_site = collections.Counter()
for (name, site, shp), c in counts.items():
    by_site[(site, name)] += c
print("%d device-touching ATen calls in one cloud's upsampling" % sum(counts.values()))
for (site, name), c in sorted(by_site.items(), key=lambda kv: (kv[0][0], -kv[1])):
    shapes = sorted({str(s) for (n, st, s), _ in counts.items() if n == name and st == site})[:3]
    print("%4d x %-30s %-56s %s" % (c, name[:30], site[-56:], " ".join(shapes)[:60]))
