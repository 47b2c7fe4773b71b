"""Per-wave balance of knn_graph_slab_kernel on real feature rows (a -DKG_TRACE build of csrc/knn.hip).

  python tools/knn_trace.py build     # here: tools/_ab/lib3pu_hip_kgtrace.so
  python tools/knn_trace.py           # on the GPU box, with that library copied over 3pu_pytorch_amd/lib3pu_hip.so"""
import ctypes, importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_ab", "lib3pu_hip_kgtrace.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.path.insert(0, ROOT)
    build = importlib.import_module("3pu_pytorch_amd.build")
    build.build()
    obj = "/tmp/knn_trace.o"
    flags = [f for f in build.HIPCC_FLAGS if f != "-shared"]
    subprocess.check_call([build.hipcc_path()] + flags + ["-DKG_TRACE", "-c", os.path.join(build.CSRC, "knn.hip"), "-o", obj],
                          cwd=build.CSRC)
    objs = [os.path.join(build.OBJ, f) for f in sorted(os.listdir(build.OBJ)) if f.endswith(".o") and f != "knn.o"]
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call([build.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, obj] + objs)
    print(OUT)
    sys.exit(0)
import numpy as np
import torch
sys.path.insert(0, ROOT)
importlib.import_module("3pu_pytorch_amd")
ops = importlib.import_module("3pu_pytorch_amd.network.operations")
ups = importlib.import_module("3pu_pytorch_amd.network.upsampler")
pipe = importlib.import_module("3pu_pytorch_amd.pipeline")
L = importlib.import_module("3pu_pytorch_amd._lib")
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
h = ctypes.CDLL(L.LIB_PATH)
h.tpu3_debug_kg_trace.argtypes = [ctypes.c_void_p]
tot = np.zeros((5, 3))
for gi, x in enumerate(rows):
    reps = (B + x.size(0) - 1) // x.size(0)
    xb = x.repeat(reps, 1, 1)[:B].contiguous()
    trace = torch.zeros((B, 8, 4), dtype=torch.int64, device=dev)
    for _ in range(30):
        ops.BACKEND.knn_graph(33, xb, optimistic=True)
    torch.cuda.synchronize()
    assert h.tpu3_debug_kg_trace(ctypes.c_void_p(trace.data_ptr())) == 0
    ops.BACKEND.knn_graph(33, xb, optimistic=True)
    torch.cuda.synchronize()
    h.tpu3_debug_kg_trace(ctypes.c_void_p(0))
    t = trace.cpu().numpy()[:, :5, :3].astype(np.float64)
    m = t.mean(0)
    tot += m
    if gi % 4 == 0:
        print("graph %2d: per slab 0..4  cycles %s  chunks with distances %s  through the network %s   workgroup: max/mean cycles %.2f"
              % (gi, np.round(m[:, 0]).astype(int), np.round(m[:, 1], 2), np.round(m[:, 2], 2),
                 (t[:, :, 0].max(1) / t[:, :, 0].mean(1)).mean()))
tot /= len(rows)
print("all %d graphs: per slab cycles %s  distances %s  network %s" % (len(rows), np.round(tot[:, 0]).astype(int), np.round(tot[:, 1], 2), np.round(tot[:, 2], 2)))
