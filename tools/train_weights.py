"""Non-random weights for the parity tests: the published checkpoints (reference Readme.md:37) are
not in this image, so the product's Net is TRAINED for a few hundred config-C3 steps on the synthetic data path
(B = 32 patches of 312 points, Chamfer loss, Adam 1e-3, ratios 2 -> 4 -> 8 like the reference's progressive schedule,
main.py:118-124; at ratio 16 the reference's loss weight log2(16/16) is 0, model.py:72, so level 4 keeps its
initialisation there as well) and the state dict is written as an .npz: tests/golden/net16_trained.npz.  The reference's
own Python is then run under these weights in the build container (oracle/make_golden.py chainall ... trained).

usage (GPU box): python tools/train_weights.py gpurun_out/r6/net16_trained.npz [steps_per_ratio]"""
import importlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("3pu_pytorch_amd")
data = importlib.import_module("3pu_pytorch_amd.data")
ups = importlib.import_module("3pu_pytorch_amd.network.upsampler")
Model = importlib.import_module("3pu_pytorch_amd.model").Model
out = sys.argv[1]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
path = data.write_synthetic(tempfile.mkdtemp(), num_shapes=8, points=(5000, 10000, 20000, 40000, 80000))
ds = data.H5Dataset(path, num_shape_point=5000, num_patch_point=312, batch_size=32, up_ratio=16, device=dev)
torch.manual_seed(0)
np.random.seed(0)
net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev)
init = {k: v.detach().clone() for k, v in net.state_dict().items()}
model = Model(net, "train", types.SimpleNamespace(lr_init=1e-3, ckpt=None))
step = 0
for r in (2, 4, 8):
    ds.unset_combined()
    ds.set_max_ratio(r)
    for i in range(per):
        a, b, rr = ds[step]
        model.set_input(a, rr, label_pc=b)
        model.optimize()
        step += 1
        if i in (0, per - 1) or i % 50 == 0:
            torch.cuda.synchronize()
            print("ratio %2d step %4d: Chamfer loss %.6f" % (r, i, float(model.error_log["cd_loss_x%d" % r])), flush=True)
state = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
moved = {k: float((v.detach() - init[k]).abs().max()) for k, v in net.state_dict().items()}
print("largest parameter change: %.3f; tensors that moved by more than 1e-3: %d of %d"
      % (max(moved.values()), sum(v > 1e-3 for v in moved.values()), len(moved)))
assert all(np.isfinite(v).all() for v in state.values())
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
np.savez_compressed(out, meta=np.array("3pu_pytorch_amd Net(16x) after %d training steps per ratio 2/4/8 on the synthetic C3 data "
                                       "path (tools/train_weights.py), seed 0" % per), **state)
print("wrote", out)
