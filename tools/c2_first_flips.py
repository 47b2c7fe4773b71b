"""Per outer patch of the recorded C2 clouds (tests/golden/c2_chain_all.npz: all 48 of seed 0; c2_chain_all_seed1.npz: 16 of
seed 1): the first discrete choice in which the HIP path departs from the reference's run, and max |dx| of the cloud
after each level -- on its own choices and with the reference's choices replayed (tests/chain_replay.py, ChainAll).

usage (GPU box): python tools/c2_first_flips.py > profiles/r06_c2_first_flips.txt"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from chain_replay import first_flip_all, run_chain_all        # noqa: E402

ops = importlib.import_module("3pu_pytorch_amd.network.operations")
ups = importlib.import_module("3pu_pytorch_amd.network.upsampler")
dev = torch.device("cuda", 0)


def net_with(weights):
    state = np.load(os.path.join(ROOT, "tests", "golden", weights))
    net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5)
    net.load_state_dict({k: torch.from_numpy(state[k]) for k in state.files if k != "meta"})
    return net.to(dev).eval()


def errors(g, ids, levels, x16):
    err = np.zeros((len(ids), 4))
    for i, q in enumerate(ids):
        for l in (1, 2, 3, 4):
            ref = g["p%d_l%d_out" % (q, l)]
            mine = levels[l - 1][i].T if l < 4 else x16[i]
            err[i, l - 1] = np.abs(mine - ref).max()
    return err


for name in ("c2_chain_all.npz", "c2_chain_all_seed1.npz", "c2_chain_all_trained.npz"):
    g = np.load(os.path.join(ROOT, "tests", "golden", name))
    ids = [int(q) for q in g["patch_ids"]]
    weights = str(g["weights"]) if "weights" in g.files else "net16_state.npz"
    net = net_with(weights)
    print("== %s (cloud seed %d, %d outer patches, weights %s)" % (name, int(g["cloud_seed"]), len(ids), weights))
    counts = {}
    for s in range(0, len(ids), 16):
        part = ids[s:s + 16]
        own, lv, x16 = run_chain_all(ops, net, g, part, dev, "record")
        e_own = errors(g, part, lv, x16)
        rep, lv, x16 = run_chain_all(ops, net, g, part, dev, "replay")
        e_rep = errors(g, part, lv, x16)
        assert rep.unexplained == [], rep.unexplained[:5]
        for i, q in enumerate(part):
            flip = first_flip_all(own, g, i, q)
            counts[flip] = counts.get(flip, 0) + 1
            print("outer patch %2d: first differing choice %-10s | own choices, max |dx| after level 1..4: %s | replayed: %s"
                  % (q, flip, " ".join("%.1e" % v for v in e_own[i]), " ".join("%.1e" % v for v in e_rep[i])))
        forced = {k: v for k, v in sorted(rep.forced.items()) if v}
        print("   (patches %d..%d replayed: rows that took the reference's tight set: %s)" % (part[0], part[-1], forced))
    print("first differing choice, count of outer patches: %s" % dict(sorted(counts.items(), key=lambda kv: str(kv[0]))))
