"""oracle/cpu_baseline.py -- TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.

The `cpu_baseline` leg of bench.py: config C1 of BASELINE.json ("CPU reference: 1 random 5000-pt
cloud, num_point=312, up_ratio=2, single upsample stage, PyTorch-CPU FPS/kNN/Chamfer") run IN FULL on
the host cores of the box bench.py runs on -- the CPU restatement of the hot path (C oracle with
OpenMP for FPS / kNN / nm-distance, torch-CPU for the conv stacks), driven by the SAME host logic as
the product (pipeline.py, network/*) through the stand-in backend of oracle/backend.py.
kind = "port": the reference has no CPU path for FPS / gather / Chamfer (CUDA-only extensions), so
this is the build's own restatement.  Median of `repeats` runs after one warm-up, per-stage times.

`measure_c1` also returns the upsampled cloud, so bench.py can put the "Chamfer vs ref" half of the
metric (HIP output against this oracle-driven output of the same cloud) into its line.
"""
import importlib
import os
import time

import numpy as np
import torch


def c1_cloud(seed=0, n=5000):
    """normalised 3-D Gaussians (uniform on S^2), (1,3,n) f32."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, n, 3, generator=g)
    return (x / x.norm(dim=2, keepdim=True)).transpose(2, 1).contiguous()


def c1_net(ups):
    torch.manual_seed(0)
    return ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).eval()


def _lscpu():
    try:
        import subprocess
        out = subprocess.check_output(["lscpu"], text=True)
        keep = {}
        for line in out.splitlines():
            k, _, v = line.partition(":")
            if k.strip() in ("Model name", "Socket(s)", "Core(s) per socket", "Thread(s) per core", "CPU(s)"):
                keep[k.strip()] = v.strip()
        return keep
    except Exception:
        return {}


def measure_c1(num_shape_point=5000, num_point=312, up_ratio=2, repeats=5, threads=None):
    from . import oracle as orc
    from .backend import OracleBackend
    ops = importlib.import_module("3pu_pytorch_amd.network.operations")
    ups = importlib.import_module("3pu_pytorch_amd.network.upsampler")
    pipe = importlib.import_module("3pu_pytorch_amd.pipeline")
    orc.build()
    ncpu = os.cpu_count() or 1
    # torch-CPU side: the conv stacks of a 312-point patch are small GEMMs; beyond ~32 threads the
    # synchronisation costs more than it buys (measured: 256 threads 2.5x slower than 8).  The C oracle
    # (OpenMP) uses every core the runtime gives it (OMP_NUM_THREADS unset = all).
    t_torch = threads or min(ncpu, 32)
    torch.set_num_threads(t_torch)
    saved = ops.BACKEND
    ops.BACKEND = OracleBackend()
    try:
        x = c1_cloud(0, num_shape_point)
        target = c1_cloud(1, num_shape_point * up_ratio).transpose(2, 1).contiguous().numpy()
        net = c1_net(ups)
        runs = []
        out = None
        for it in range(repeats + 1):
            st = {}
            t0 = time.perf_counter()
            cl = x.transpose(2, 1).contiguous()
            P = pipe.num_outer_patches(num_shape_point, num_point, 3)
            seed_idx = ops.fps(cl, P)
            st["seed_fps"] = time.perf_counter() - t0
            t1 = time.perf_counter()
            seeds = torch.gather(cl, 1, seed_idx.long().unsqueeze(-1).expand(-1, -1, 3))
            _, _, patches = ops.knn_query(num_point, seeds, cl, unique=True, want_dist=False)
            st["outer_knn"] = time.perf_counter() - t1
            t1 = time.perf_counter()
            with torch.no_grad():
                up, _ = pipe.upsample_patches(net, patches.reshape(P, num_point, 3), up_ratio)
            st["levels"] = time.perf_counter() - t1
            t1 = time.perf_counter()
            merged = up.reshape(1, P * up.size(1), 3)
            idx = ops.fps(merged, num_shape_point * up_ratio)
            out = torch.gather(merged, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3))
            st["final_fps"] = time.perf_counter() - t1
            st["total"] = time.perf_counter() - t0
            t1 = time.perf_counter()
            cd = float(orc.chamfer_loss(out.numpy(), target))
            st["chamfer_vs_seed1_sphere"] = time.perf_counter() - t1
            if it:                                   # run 0 = warm-up
                runs.append(st)
        med = {k: float(np.median([r[k] for r in runs])) for k in runs[0]}
        m_out = num_shape_point * up_ratio
        return {
            "value": m_out / med["total"], "unit": "points/s", "cores": ncpu, "kind": "port",
            "sample": ("config C1 in full: 1 cloud x %d pts, num_point=%d, up_ratio=%d (one level), %d outer "
                       "patches -> %d -> FPS %d; median of %d runs after 1 warm-up; C oracle kernels on OpenMP "
                       "(all %d hardware threads), torch-CPU conv stacks on %d threads"
                       % (num_shape_point, num_point, up_ratio, P, P * num_point * up_ratio, m_out, repeats,
                          ncpu, t_torch)),
            "stage_s": med, "chamfer_vs_seed1_sphere": cd, "lscpu": _lscpu(),
        }, out.contiguous()                                  # (1, m_out, 3) channel-last
    finally:
        ops.BACKEND = saved


def measure(num_shape_point=5000, num_point=312, up_ratio=2):
    return measure_c1(num_shape_point, num_point, up_ratio)[0]


def run_c2(golden_dir, threads=None):
    """Config C2 (the metric's: 5000 -> 80 000 points, 16x, 48 outer patches) through the oracle-driven CPU path
    -- the product's HOST logic (pipeline.py, network/*: every patch of a level in one batched call) over the C
    oracle and torch-CPU convolutions -- on the cloud and weights of tests/golden/c2_x16.npz.  Returns a "run"
    (oracle/parity_np.py) and the wall time.  Minutes on 8 cores."""
    from . import oracle as orc
    from .backend import OracleBackend
    ops = importlib.import_module("3pu_pytorch_amd.network.operations")
    ups = importlib.import_module("3pu_pytorch_amd.network.upsampler")
    pipe = importlib.import_module("3pu_pytorch_amd.pipeline")
    orc.build()
    torch.set_num_threads(threads or min(os.cpu_count() or 1, 32))
    g = np.load(os.path.join(golden_dir, "c2_x16.npz"))
    state = np.load(os.path.join(golden_dir, "net16_state.npz"))
    net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5)
    net.load_state_dict({k: torch.from_numpy(state[k]) for k in state.files if k != "meta"})
    net.eval()
    saved = ops.BACKEND
    ops.BACKEND = OracleBackend()
    try:
        t0 = time.perf_counter()
        cloud = torch.from_numpy(g["cloud"])
        seed_idx, patches, pidx = pipe.extract_outer_patches(cloud, 312, 3)
        P = patches.size(1)
        levels = []
        with torch.no_grad():
            up, _ = pipe.upsample_patches(net, patches.reshape(P, 312, 3), 16, levels_out=levels)
        merged = up.reshape(1, P * up.size(1), 3)
        idx = ops.fps(merged, 80000)
        final = torch.gather(merged, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3))
        dt = time.perf_counter() - t0
    finally:
        ops.BACKEND = saved
    run = {"seed_idx": seed_idx.numpy(), "patch_idx": pidx.numpy()[0],
           "lv1": levels[0].numpy(), "lv2": levels[1].numpy(), "lv3": levels[2].numpy(),
           "pred_concat": merged.transpose(2, 1).contiguous().numpy(), "final": final.transpose(2, 1).contiguous().numpy()}
    return run, dt


def measure_c2(golden_dir, out_path=None):
    """`python -m oracle.cpu_baseline --c2`: the oracle-driven CPU path against the reference driver's fixture and
    against the two reference-vs-reference controls -> profiles/r04_c2_cpu_vs_ref.json."""
    import json
    from . import parity_np as pn
    run, dt = run_c2(golden_dir)
    ref = np.load(os.path.join(golden_dir, "c2_x16.npz"))
    out = {"what": "config C2 through the oracle-driven CPU path (product host logic over oracle/ref_kernels.c + "
                   "torch-CPU convolutions) against tests/golden/c2_x16.npz (the reference's own Python driver)",
           "wall_s": dt, "torch": torch.__version__, "numpy": np.__version__,
           "outer_seeds_bit_exact": bool((run["seed_idx"] == ref["seed_idx"]).all()),
           "outer_patch_idx_bit_exact": bool((run["patch_idx"] == ref["patch_idx"]).all()),
           "cpu_path_vs_ref": pn.compare_runs(run, ref)}
    for name in ("c2_x16_alt.npz", "c2_x16_alt2.npz", "c2_x16_alt3.npz"):
        path = os.path.join(golden_dir, name)
        if os.path.exists(path):
            alt = np.load(path)
            out["ref_vs_" + name[7:-4]] = dict(pn.compare_runs(ref, alt), variant=str(alt["variant"]))
            out["cpu_path_vs_" + name[7:-4]] = pn.compare_runs(run, alt)
    if out_path:
        with open(out_path, "w") as f:
            json.dump(out, f, indent=1)
    return out


if __name__ == "__main__":
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    if "--c2" in sys.argv:
        r = measure_c2(os.path.join(root, "tests", "golden"), os.path.join(root, "profiles", "r04_c2_cpu_vs_ref.json"))
        import json
        print(json.dumps(r, indent=1))
    else:
        print(measure())
