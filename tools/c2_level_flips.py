"""Which outer patches of the C2 fixture leave the 1e-5 band at which level on the HIP path (GPU box)."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = lambda s: importlib.import_module("3pu_pytorch_amd." + s)
par, ups = pkg("utils.parity"), pkg("network.upsampler")
dev = torch.device("cuda:0")
g = np.load(os.path.join(ROOT, "tests/golden/c2_x16.npz"))
state = np.load(os.path.join(ROOT, "tests/golden/net16_state.npz"))
net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5)
net.load_state_dict({k: torch.from_numpy(state[k]) for k in state.files if k != "meta"})
net = net.to(dev).eval()
mine = par.run_c2(net, torch.from_numpy(g["cloud"]).to(dev))
ref = {k: g[k] for k in par.KEYS}
for lv, (x, y) in enumerate(zip(par.level_clouds(mine, dev), par.level_clouds(ref, dev)), 1):
    err = (x - y).abs().reshape(48, -1).amax(dim=1).cpu().numpy()
    print("level %d: patches beyond 1e-5: %s" % (lv, [int(i) for i in np.where(err > 1e-5)[0]]))
    print("   max err per patch: " + " ".join("%.1e" % e for e in err))
