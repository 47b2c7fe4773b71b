"""Device gradients of the training forward against the reference fixture (tests/golden/net_train_grad.npz), with the
fused skip-connection node on and off, and the two device runs against each other (GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from conftest import pkg
import test_host_network_cpu as host
ops, ups, layers = pkg("network.operations"), pkg("network.upsampler"), pkg("network.layers")
dev = torch.device("cuda", 0)

def run():
    g, pred, gt, ginp, grads = host._train_grads(ups, dev)
    ref = {k[5:]: g[k] for k in g.files if k.startswith("grad_level")}
    return ref, {n: v.cpu().numpy() for n, v in grads.items()}

def table(tag, ref, got):
    rows = []
    for n, r in ref.items():
        sc = max(1e-6, float(np.abs(r).max()))
        rows.append((float(np.abs(got[n] - r).max()) / sc, n))
    for lv in (1, 2, 3):
        sub = sorted(v for v in rows if "level_%d" % lv in v[1])
        print("%-22s level %d: worst %.2e (%s), median %.2e" % (tag, lv, sub[-1][0], sub[-1][1].split("level_%d." % lv)[1], sub[len(sub) // 2][0]))

ref, fused = run()
table("fused vs reference", ref, fused)
cls = type(ops.BACKEND)
fn = cls.interlevel_skip_train
del cls.interlevel_skip_train
_, plain = run()
cls.interlevel_skip_train = fn
table("plain vs reference", ref, plain)
table("fused vs plain", plain, fused)
_, fused2 = run()
table("fused vs fused again", fused, fused2)

real = ups._SkipTrain
class Formula(object):
    """the autograd formulation of the skip connection through the node's call site"""
    @staticmethod
    def apply(x, pf, xyz, pxyz, pts_of, idx):
        bsel = torch.arange(x.size(0), device=x.device).view(-1, 1, 1)
        kf = pf[bsel, idx.long()]
        w = ups.Level.exponential_distance_cl(xyz, pxyz[bsel, idx.long()]) * ups.Level.exponential_distance_cl(x, kf)
        w = w / torch.sum(w + 1e-5, dim=-1, keepdim=True)
        return 0.2 * torch.sum(w.unsqueeze(-1) * kf, dim=2) + x

# Where the two runs part: every DenseEdgeConv block's input, graph, output and gradients.  (Measured: the skip node's
# output differs from the formula's by 6e-8 at level 2; that flips near-tied neighbours in level 3's feature-space
# graphs of blocks 3 and 4, and the gradients follow the other neighbour: 1e-2 of the largest gradient there.)
def traced_dec(kind):
    rec = []
    fd = layers._FusedDECTrain
    class D(object):
        @staticmethod
        def apply(x, idx, idx_off, *w):
            slot = {"x": x.detach().clone(), "idx": idx.detach().clone().float()}
            x.register_hook(lambda g, s=slot: s.__setitem__("gx", g.detach().clone()))
            y = fd.apply(x, idx, idx_off, *w)
            slot["y"] = y.detach().clone()
            y.register_hook(lambda g, s=slot: s.__setitem__("gy", g.detach().clone()))
            rec.append(slot)
            return y
    layers._FusedDECTrain = D
    ups._SkipTrain = real if kind == "node" else Formula
    run()
    ups._SkipTrain = real
    layers._FusedDECTrain = fd
    return rec
ra, rb = traced_dec("node"), traced_dec("formula")
for i, (a, b) in enumerate(zip(ra, rb)):
    print("level %d block %d: " % (i // 4 + 1, i % 4 + 1) + ", ".join("%s %.3g (of %.3g)" % (k, float((a[k] - b[k]).abs().max()), float(b[k].abs().max())) for k in ("x", "idx", "y", "gy", "gx")))
