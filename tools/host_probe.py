"""How long does the host take to ENQUEUE one pipeline step (GPU box)?"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ops, pipe, ups = bench.pkg("network.operations"), bench.pkg("pipeline"), bench.pkg("network.upsampler")
dev = torch.device("cuda", 0)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev).eval()
clouds = torch.cat([bench.poisson_sphere(i, 5000, dev, ops) for i in range(C)], 0)
for final in (False, True):
    for _ in range(2):
        out = pipe.upsample(net, clouds, 312, 16, 3, final_fps=final); torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipe.upsample(net, clouds, 312, 16, 3, final_fps=final)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("clouds=%d final_fps=%s: host enqueue %.1f ms, until GPU done %.1f ms" % (C, final, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
out = pipe.upsample(net, clouds, 312, 16, 3, final_fps=False)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
