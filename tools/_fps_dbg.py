import sys, importlib, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import sphere
from oracle import oracle as orc
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
dev = torch.device("cuda", 0)
n, m = 239616, 20000
xyz = sphere(2000 + n, n, 1)
ref, reft = orc.fps(xyz, m)
x = torch.from_numpy(xyz).to(dev)
outs = []
for rep in range(3):
    i = ops.fps(x, m).cpu().numpy()
    bad = np.where(i[0] != ref[0])[0]
    print("rep", rep, "mismatches", len(bad), "first", bad[:5], "mine", i[0][bad[:3]], "ref", ref[0][bad[:3]])
    outs.append(i)
print("deterministic:", (outs[0] == outs[1]).all(), (outs[1] == outs[2]).all())
