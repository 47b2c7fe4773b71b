import sys, numpy as np, torch, importlib
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from conftest import golden, pkg
ups = pkg("network.upsampler")
dev = torch.device("cuda",0)
net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5)
state = golden("net16_state.npz")
net.load_state_dict({k: torch.from_numpy(state[k]) for k in state.files if k!="meta"}); net.to(dev).eval()
g = golden("net_eval.npz"); lv = golden("net_levels_x16.npz")
net.trace = []
with torch.no_grad():
    y = net(torch.from_numpy(g["patch"]).to(dev), ratio=16)
for l,t in enumerate(net.trace,1):
    ref_in, ref_out = lv["l%d_patch_xyz"%l], lv["l%d_out_norm"%l]
    P = ref_in.shape[0]
    mi = t["patch_xyz"][:P].cpu().numpy(); mo = t["out_norm"][:P].cpu().numpy()
    pn = t.get("patch_num")
    print("level",l,"P",P,"patch_num", None if pn is None else pn.cpu().numpy(), "in shape", tuple(t["patch_xyz"].shape),
          "in err frac>1e-5", (np.abs(mi-ref_in).max(axis=1)>1e-5).mean(), "out err frac>1e-5", (np.abs(mo-ref_out).max(axis=1)>1e-5).mean(), "out maxerr", np.abs(mo-ref_out).max())
from oracle import oracle as orc
ref = g["x16"]; yy = y.cpu().numpy()
d1,_,d2,_ = orc.nmdistance_fwd(np.ascontiguousarray(yy.transpose(0,2,1)), np.ascontiguousarray(ref.transpose(0,2,1)))
print("x16 chamfer", d1.mean()+d2.mean(), "nn dist pct 50/90/99/max", [float(np.sqrt(np.percentile(d1,p))) for p in (50,90,99,100)])
# self spacing of reference cloud
dd,_,_,_ = orc.nmdistance_fwd(np.ascontiguousarray(ref.transpose(0,2,1)), np.ascontiguousarray(ref.transpose(0,2,1))[:, ::2])
print("ref self-spacing (to half of itself) median", float(np.sqrt(np.median(dd[dd>0]))))
