"""Per-phase cycle counters of the bucketed FPS kernel (development probe, GPU box)."""
import ctypes
import importlib
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("3pu_pytorch_amd._lib")
lib = ctypes.CDLL(L.LIB_PATH)
dev = torch.device("cuda", 0)
n, m = 239616, int(sys.argv[1]) if len(sys.argv) > 1 else 40000
g = torch.Generator().manual_seed(0)
x = torch.randn(1, n, 3, generator=g)
x = (x / x.norm(dim=2, keepdim=True)).to(dev)
L.lib()
need = L.lib().tpu3_fps_workspace_bytes(1, n)
ws = torch.empty(need, dtype=torch.uint8, device=dev)
temp = torch.full((1, n), 1e10, device=dev)
idx = torch.zeros((1, m), dtype=torch.int32, device=dev)
NWV = 4
prof = torch.zeros(NWV * 16, dtype=torch.int64, device=dev)
fn = lib.tpu3_debug_fps_bucket_profile
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_void_p]
rc = fn(None, n, m, x.data_ptr(), temp.data_ptr(), idx.data_ptr(), ws.data_ptr(), need, prof.data_ptr())
torch.cuda.synchronize()
print("rc", rc)
p = prof.view(NWV, 16).cpu().numpy().astype(float)
names = ["phase1", "argmax", "barrier", "bcast", "-", "buckets", "groupbatches", "-", "gprune", "childtest", "load+apply", "reduce+publish", "refresh"]
print("per round, shader cycles:")
for w in range(NWV):
    print("wave %2d " % w + "  ".join("%s %8.1f" % (names[i], p[w, i] / (m - 1)) for i in range(13)))
tot = p[:, :4].sum(1) / (m - 1)
print("sum of phases per round:", tot)
