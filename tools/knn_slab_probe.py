"""The slab form of the feature kNN graph (csrc/knn.hip, knn_graph_slab_kernel) on REAL feature rows.

Runs the 16x network once on the C2 fixture cloud (one cloud, HIP path), records the input of every feature graph,
and times each recorded tensor (patches repeated to PATCHES rows) through BACKEND.knn_graph.  Run it twice, with
TPU3_KG_SLAB=0 and =1, to compare the two kernels on the same rows; CHECK=1 also compares the sets with the oracle
on the first 4 patches of every graph.

usage (GPU box): TPU3_KG_SLAB=0 python tools/knn_slab_probe.py; TPU3_KG_SLAB=1 CHECK=1 python tools/knn_slab_probe.py"""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("3pu_pytorch_amd")
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
ups = importlib.import_module("3pu_pytorch_amd.network.upsampler")
pipe = importlib.import_module("3pu_pytorch_amd.pipeline")
dev = torch.device("cuda:0")
B = int(os.environ.get("PATCHES", "3840"))

g = np.load(os.path.join(ROOT, "tests", "golden", "c2_x16.npz"))
state = np.load(os.path.join(ROOT, "tests", "golden", "net16_state.npz"))
net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5)
net.load_state_dict({k: torch.from_numpy(state[k]) for k in state.files if k != "meta"})
net = net.to(dev).eval()
rows = []
real = ops.BACKEND.knn_graph


def spy(k, x, layout=None, optimistic=None):
    rows.append(x.detach().clone())
    return real(k, x, layout=layout, optimistic=optimistic)


ops.BACKEND.knn_graph = spy
with torch.no_grad():
    pipe.upsample(net, torch.from_numpy(g["cloud"]).to(dev), 312, 16, 3)
ops.BACKEND.knn_graph = real
torch.cuda.synchronize()
print("TPU3_KG_SLAB=%s: %d feature graphs recorded, shapes %s" % (os.environ.get("TPU3_KG_SLAB", "(default 1)"), len(rows),
                                                                  sorted({tuple(r.shape) for r in rows})))
if os.environ.get("CHECK"):
    from oracle import oracle as orc
    bad = 0
    for x in rows:
        xs = x[:4].contiguous()
        ri, _ = orc.knn(33, xs.cpu().numpy(), xs.cpu().numpy(), True)
        ops.BACKEND.graph_dup_events(reset=True)
        idx = ops.BACKEND.knn_graph(33, xs, optimistic=True).cpu().numpy()
        ev = ops.BACKEND.graph_dup_events(reset=True)
        ok = ev == 0 and (idx[:, :, 0] == ri[:, :, 0]).all() and (np.sort(idx[:, :, 1:], -1) == np.sort(ri[:, :, 1:], -1)).all()
        bad += 0 if ok else 1
    print("oracle check on 4 patches of each graph: %d of %d graphs differ" % (bad, len(rows)))
tot = 0.0
per = []
for gi, x in enumerate(rows):
    reps = (B + x.size(0) - 1) // x.size(0)
    xb = x.repeat(reps, 1, 1)[:B].contiguous()
    ts = []
    for it in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.BACKEND.knn_graph(33, xb, optimistic=True)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    med = sorted(ts[2:])[len(ts[2:]) // 2]
    per.append(med)
print("ms per launch of %d patches, by graph (level-major, 4 blocks per level): %s" % (B, " ".join("%.3f" % p for p in per)))
print("mean %.3f ms = %.2f ps per (query, candidate) pair" % (np.mean(per), np.mean(per) * 1e9 / (B * 312 * 312)))
