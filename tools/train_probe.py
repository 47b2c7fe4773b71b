"""Config C3: one training step (forward, Chamfer, backward, clip, Adam) at B = 32, 312-point input
patches, for every ratio, fed by the device-resident data path; rocprof-friendly (a few steps)."""
import importlib, os, sys, time, tempfile, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tpu = importlib.import_module("3pu_pytorch_amd")
data = importlib.import_module("3pu_pytorch_amd.data")
ups = importlib.import_module("3pu_pytorch_amd.network.upsampler")
Model = importlib.import_module("3pu_pytorch_amd.model").Model
dev = torch.device("cuda", 0)
path = data.write_synthetic(tempfile.mkdtemp(), num_shapes=8, points=(5000, 10000, 20000, 40000, 80000))
ds = data.H5Dataset(path, num_shape_point=5000, num_patch_point=312, batch_size=32, up_ratio=16, device=dev)
torch.manual_seed(0)
net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev)
model = Model(net, "train", types.SimpleNamespace(lr_init=1e-3, ckpt=None))
STEPS = int(os.environ.get("STEPS", "5"))
for r in [int(v) for v in os.environ.get("RATIOS", "2,4,8,16").split(",")]:
    ds.unset_combined(); ds.set_max_ratio(r)
    for i in range(2):
        a, b, rr = ds[i]; model.set_input(a, rr, label_pc=b); model.optimize()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = STEPS
    for i in range(n):
        a, b, rr = ds[i]; model.set_input(a, rr, label_pc=b); model.optimize()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
    print("ratio %2d: %.1f ms per training step (B=32 x 312 points -> %d), loss %.5f"
          % (r, t * 1e3, 312 * r, model.error_log["cd_loss_x%d" % r]))
print("peak memory %.1f GB" % (torch.cuda.max_memory_allocated() / 2**30))
