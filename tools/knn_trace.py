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
    trace = torch.zeros((2, B, 8, 4), dtype=torch.int64, device=dev)
    for _ in range(30):
        ops.BACKEND.knn_graph(33, xb, optimistic=True)
    torch.cuda.synchronize()
    assert h.tpu3_debug_kg_trace(ctypes.c_void_p(trace.data_ptr())) == 0
    ops.BACKEND.knn_graph(33, xb, optimistic=True)
    torch.cuda.synchronize()
    h.tpu3_debug_kg_trace(ctypes.c_void_p(0))
    both = trace.cpu().numpy()
    full = both[0].astype(np.float64)
    t = full[:, :5, :3]
    m = t.mean(0)
    stage = full[:, :5, 3].mean()
    tail = np.concatenate([full[:, 5, :3].ravel(), full[:, 6, :2].ravel()]).mean()
    if gi == 0:
        w = both[1][:, :5, :]                                   # (B, 5 waves, [entry, exit, hw, xcc])
        hw, xcc = w[:, :, 2], w[:, :, 3] & 0xF
        simd, cu, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 13) & 7
        print("   SIMD of wave 0..4 (share of workgroups): " + "  ".join(
            "w%d: %s" % (k, np.round(np.bincount(simd[:, k], minlength=4) / len(simd), 2)) for k in range(5)))
        print("   waves 0 and 4 on the same SIMD: %.2f of the workgroups" % (simd[:, 0] == simd[:, 4]).mean())
        key = (xcc[:, 0] * 8 + se[:, 0]) * 16 + cu[:, 0]
        conc = []
        for kk in np.unique(key)[:64]:
            sel = key == kk
            st, en = w[sel][:, :, 0].min(1), w[sel][:, :, 1].max(1)
            ts = np.linspace(st.min(), en.max(), 200)
            conc.append(np.mean([((st <= x) & (en > x)).sum() for x in ts]))
        print("   compute units seen %d; patches resident per compute unit over its busy time: mean %.2f" % (len(np.unique(key)), np.mean(conc)))
    if gi % 4 == 0:
        print("   staging (entry -> slab loop) %.0f cycles, after the loop (collisions, write-out) %.0f" % (stage, tail))
    tot += m
    if gi % 4 == 0:
        print("graph %2d: per slab 0..4  cycles %s  chunks with distances %s  through the network %s   workgroup: max/mean cycles %.2f"
              % (gi, np.round(m[:, 0]).astype(int), np.round(m[:, 1], 2), np.round(m[:, 2], 2),
                 (t[:, :, 0].max(1) / t[:, :, 0].mean(1)).mean()))
tot /= len(rows)
print("all %d graphs: per slab cycles %s  distances %s  network %s" % (len(rows), np.round(tot[:, 0]).astype(int), np.round(tot[:, 1], 2), np.round(tot[:, 2], 2)))
