"""Items per second of the device-resident training data path at the reference's training sizes
(input 5000 points, labels up to 80 000, 312-point patches, batch 32)."""
import importlib, os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
data = importlib.import_module("3pu_pytorch_amd.data")
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
path = data.write_synthetic(tmp, num_shapes=8, points=(5000, 10000, 20000, 40000, 80000))
ds = data.H5Dataset(path, num_shape_point=5000, num_patch_point=312, batch_size=32, up_ratio=16, device=dev)
for r in (2, 4, 8, 16):
    ds.unset_combined(); ds.set_max_ratio(r)
    for i in range(3):
        ds[i]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for i in range(n):
        a, b, rr = ds[i]
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / n
    print("ratio %2d: %.2f ms per item (batch of 32 patch pairs, label patches %d points) = %.0f patch pairs/s"
          % (r, t * 1e3, b.shape[2], 32 / t))
