"""Phase timeline of dec_fused4_kernel from shader-clock marks (a -DDEC4_TRACE build of csrc/dense_edge_conv.hip).

  python tools/dec_trace.py build     # here: hipcc -DDEC4_TRACE -> tools/_ab/lib3pu_hip_trace.so (other objects as built)
  python tools/dec_trace.py           # on the GPU box, with that library copied over 3pu_pytorch_amd/lib3pu_hip.so

Per wave: the time between consecutive marks (median / mean / p90 over the waves of ONE launch in steady state), and
how many waves of the launch are in which phase over time."""
import ctypes, importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "3pu_pytorch_amd")
OUT = os.path.join(ROOT, "tools", "_ab", "lib3pu_hip_trace.so")
NAMES = ["raw weights -> LDS / packed tables (+barrier)", "tables, packed operands (+barrier) / packed fold table",
         "fold table, comb init", "phase A (z table)",
         "barrier 1", "centre terms", "slots, whole step", "finish, whole step", "centre + slots, split step",
         "barrier 2", "finish, split step"]

if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.path.insert(0, ROOT)
    build = importlib.import_module("3pu_pytorch_amd.build")
    build.build()
    obj = os.path.join("/tmp", "dense_edge_conv_trace.o")
    flags = [f for f in build.HIPCC_FLAGS if f != "-shared"]
    subprocess.check_call([build.hipcc_path()] + flags + ["-DDEC4_TRACE", "-c", os.path.join(build.CSRC, "dense_edge_conv.hip"),
                                                           "-o", obj], cwd=build.CSRC)
    objs = [os.path.join(build.OBJ, f) for f in sorted(os.listdir(build.OBJ)) if f.endswith(".o") and f != "dense_edge_conv.o"]
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call([build.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, obj] + objs)
    print(OUT)
    sys.exit(0)

import numpy as np
import torch
sys.path.insert(0, ROOT)
importlib.import_module("3pu_pytorch_amd")
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
layers = importlib.import_module("3pu_pytorch_amd.network.layers")
L = importlib.import_module("3pu_pytorch_amd._lib")
dev = torch.device("cuda:0")
B, n, C, k = int(os.environ.get("PATCHES", "3840")), int(os.environ.get("N", "312")), 24, int(os.environ.get("K", "32"))
fold = int(os.environ.get("FOLD", "0"))
torch.manual_seed(0)
conv = layers.DenseEdgeConv(24, growth_rate=12, n=3, k=k).to(dev).eval()
x = torch.rand((B, n, C), device=dev)
idx = torch.randint(0, n, (B, n, k + 1), device=dev, dtype=torch.int32)
out = torch.empty((B, n, 60), device=dev)
fw, fb = torch.randn((max(fold, 24), 60), device=dev) * 0.1, torch.randn((max(fold, 24),), device=dev) * 0.1
acc, xn = torch.zeros((B, n, 48), device=dev), torch.empty((B, n, 24), device=dev)


PACK = os.environ.get("PACK", "1") != "0"
pack = ops.BACKEND.dense_edge_conv_pack(conv.mlps, fw if fold else None) if PACK else None


def launch():
    if fold:
        ops.BACKEND.dense_edge_conv_fold(x, idx, 1, k, conv.mlps, out, fw, fb, acc, 0, 0, xn, pack=pack)
    else:
        ops.BACKEND.dense_edge_conv(x, idx, 1, k, conv.mlps, out, pack=pack)


h = ctypes.CDLL(L.LIB_PATH)
h.tpu3_debug_dec_trace.argtypes = [ctypes.c_void_p]
trace = torch.zeros((B * 4, 16), dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(int(os.environ.get("WARM", "1500"))):
        launch()
    torch.cuda.synchronize()
    assert h.tpu3_debug_dec_trace(ctypes.c_void_p(trace.data_ptr())) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        launch()
    e0.record()
    launch()
    e1.record()
    torch.cuda.synchronize()
    h.tpu3_debug_dec_trace(ctypes.c_void_p(0))
t = trace.cpu().numpy()
ok = (t[:, :12] > 0).all(axis=1)
print("waves with all marks: %d of %d" % (ok.sum(), len(t)))
t = t[ok]
ms = e0.elapsed_time(e1)
t0 = t[:, 0].min()
span = t[:, 11].max() - t0
rt = t[:, 14]
print("launch %.3f ms; first mark to last mark %d ticks of the shader clock counter (%.1f MHz if they span the launch); "
      "s_memrealtime span %d" % (ms, span, span / ms / 1e3, rt.max() - rt.min()))
d = np.diff(t[:, :12], axis=1).astype(np.float64)
print("%-40s %10s %10s %10s" % ("phase", "median", "mean", "p90"))
for i, nm in enumerate(NAMES):
    print("%-40s %10.0f %10.0f %10.0f" % (nm, np.median(d[:, i]), d[:, i].mean(), np.percentile(d[:, i], 90)))
life = (t[:, 11] - t[:, 0]).astype(np.float64)
print("%-40s %10.0f %10.0f %10.0f" % ("wave life (first to last mark)", np.median(life), life.mean(), np.percentile(life, 90)))
# occupancy over time: waves alive, waves inside the slot loops, waves waiting at a barrier
T = np.linspace(t0, t[:, 11].max(), 41)[:-1]
print("time slice: waves alive / in a slot loop / at a barrier (of %d wave slots: 1024 SIMDs x 3)" % (1024 * 3))
for a0 in T[::2]:
    alive = ((t[:, 0] <= a0) & (t[:, 11] > a0)).sum()
    loop = (((t[:, 6] <= a0) & (t[:, 7] > a0)) | ((t[:, 8] <= a0) & (t[:, 9] > a0))).sum()
    bar = (((t[:, 4] <= a0) & (t[:, 5] > a0)) | ((t[:, 9] <= a0) & (t[:, 10] > a0))).sum()
    print("  %8d  %5d %5d %5d" % (a0 - t0, alive, loop, bar))
