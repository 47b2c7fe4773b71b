"""nm-distance (Chamfer) forward / backward kernels: time per call in BOTH forms -- the reference's scan
(csrc/nmdistance.hip) and the grid-pruned search with the same bits (csrc/nmdist_grid.hip, r6) -- and the fraction of
the fp32 vector peak on SURVEY 8d's model (16 FLOP per pair and direction pair = 2*B*n*m*8; the pruned search skips most
of those pairs, so its "model fraction" may exceed 1: it is a work-skipping exact kernel, reported as such).

usage: python tools/chamfer_probe.py [--sphere]      (--sphere: points on S^2 like the pipeline's clouds instead of
                                                      uniform in the unit cube)"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("3pu_pytorch_amd")
ml = importlib.import_module("3pu_pytorch_amd.network.model_loss")
lib = importlib.import_module("3pu_pytorch_amd._lib").lib()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
SPHERE = "--sphere" in sys.argv


def cloud(B, n):
    if SPHERE:
        p = torch.randn((B, n, 3), device=dev, generator=g)
        return (p / p.norm(dim=2, keepdim=True)).contiguous()
    return torch.rand((B, n, 3), device=dev, generator=g)


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


print("points: %s" % ("uniform on S^2" if SPHERE else "uniform in the unit cube"))
for B, n, m in ((32, 624, 624), (32, 2496, 2496), (32, 4992, 4992), (1, 5000, 5000), (1, 20000, 20000), (1, 80000, 80000),
                (1, 320000, 320000), (1, 1280000, 1280000)):
    a = cloud(B, n).requires_grad_(True)
    b = cloud(B, m)
    flop = 2.0 * B * n * m * 8
    row = "B=%2d n=m=%7d" % (B, n)
    for form, name in ((0, "scan"), (1, "grid")):
        if form == 0 and n > 400000:
            with torch.no_grad():            # one run only: ~0.5 s
                lib.tpu3_debug_nmdist_form(0)
                t = timeit(lambda: ml.nndistance(a, b), reps=1)
        else:
            lib.tpu3_debug_nmdist_form(form)
            with torch.no_grad():
                t = timeit(lambda: ml.nndistance(a, b))
        row += "   %s %9.3f ms (%6.1f TFLOP/s on the 16 n m model = %5.2f of 157.3)" % (name, t, flop / t / 1e9, flop / t / 1e9 / 157.3)
    if n >= 2048:
        st = torch.zeros(4, dtype=torch.int64, device=dev)
        lib.tpu3_debug_nmdist_form(1)
        lib.tpu3_debug_nmdist_grid_stats(st.data_ptr())
        with torch.no_grad():
            ml.nndistance(a, b)
        torch.cuda.synchronize()
        lib.tpu3_debug_nmdist_grid_stats(None)
        w, ss, tt, sr = [int(v) for v in st.cpu()]
        row += "   [per query wave: %.1f super-tiles, %.1f tiles pass the wave bound, %.1f searched]" % (ss / w, tt / w, sr / w)
    lib.tpu3_debug_nmdist_form(-1)
    lib.tpu3_debug_nmdist_grid_calls(1)

    def fb():
        d1, _, d2, _ = ml.nndistance(a, b)
        (d1.mean() + d2.mean()).backward()
    tb = timeit(fb)
    row += "   automatic (%s) forward+backward %9.3f ms" % ("grid" if lib.tpu3_debug_nmdist_grid_calls(1) else "scan", tb)
    print(row, flush=True)
