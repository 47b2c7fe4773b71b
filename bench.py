"""bench.py -- headline benchmark of the hot path (contract in the task statement).

Metric (BASELINE.json): upsampled points/sec, 16x, 312-point patches, 5000 -> 80000 points per
cloud, config C2 (1 x MI355X, num_point 312, num_shape_point 5000, up_ratio 16, random-init
weights, synthetic Poisson-sphere inputs).  A "step" is one pass of the whole pipeline
(seeds, outer patches, 4 progressive levels incl. every inner FPS / kNN, concat, final FPS) over
`--clouds` clouds per GPU, inputs resident in HBM.

Multi-GPU (launched by torch.distributed.run, one rank per GPU, RCCL):
  python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 --clouds 8
      = config C4 exactly (64 clouds x 5000 points, whole clouds per rank, ONE all-gather of the
      upsampled xyz per step over xGMI; weak scaling: per-GPU work fixed as N grows);
  ... bench.py --gpus 8 --shard patches
      = ONE cloud whose 48 outer patches are split across the ranks (all-gather of the upsampled
      patches, final FPS replicated; strong scaling, the final FPS is the Amdahl term).
The default (--gpus N, 32 clouds per GPU) is the throughput configuration of the 1-GPU line.

One JSON line is printed by rank 0.  Besides the contract's keys it carries
  roofline          the kernel that BOUNDS the step: the largest entry of `rooflines_other` by ms_per_step (round 4:
                    dec_fused4_kernel, fp32 MFMA) with per-launch averages and the PMC traffic per launch
  roofline_step     the whole step against the chip: executed MFMA FLOP and counter bytes / ms_per_step
  rooflines_other   (r6: followed by the point-set kernels OUTSIDE the step -- nm-distance forward in both forms / backward,
                    ball query, gather forward / backward -- as `ms_per_call` entries with `ms_per_step` null)
                    every hand-written kernel of a step, timed with events on its launch stream: MFMA kernels on
                    EXECUTED matrix-core FLOPs, the kNN graph on the (2C+3) VALU model, HBM kernels on algorithmic
                    bytes; the final FPS as a LATENCY entry (rounds, us_per_round, x_over_floor -- it runs hidden on
                    a side stream and bounds nothing)
  rooflines_train   config C3 (one optimisation step, B = 32, ratio 16, eager): the training kernels by time, the
                    dominant one against its bound
  cpu_baseline      config C1 in full on the host cores (oracle/cpu_baseline.py), per-stage times
  parity            config C1 through the HIP path and through the oracle-driven CPU path on the same cloud:
                    chamfer_vs_oracle, set_close_1e-5  ("Chamfer vs ref" half of the metric)
  parity_c2         the metric's OWN configuration (C2, 16x, 5000 -> 80000) through the HIP path against the output of
                    the reference's own Python driver for the same cloud and weights (tests/golden/c2_x16.npz)
  value_1cloud      BASELINE's C2 read literally: ONE cloud at a time (the faster of eager / hipGraph replay) -> points/s
  value_8clouds     config C4's per-rank share: 8 clouds per GPU per step -> points/s
  value_fp32_matrix_instructions   the same step with the regressor's two matrix kernels on the fp32 matrix instructions
                    (TPU3_SPLIT_BF16=0) instead of the default three-term bf16 operands: what `dtype` names, side by side
  extras            latency_ms_1cloud (+ _eager), ms_per_step_8clouds, train_step_ms (C3: B = 32, ratio 16),
                    chamfer_80k_ms, c5 (stress) ...
"""
import argparse
import importlib
import json
import os
import sys
import time

# The step keeps ~9 HIP streams busy (network sub-batches + the final-FPS launches of several
# steps).  The HIP runtime multiplexes streams onto 4 hardware queues by default, and a stream that
# lands behind a 300 ms final-FPS kernel on the same queue stalls (measured: 302 vs 190 ms/step).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_PEAK_TF = 157.3           # fp32 vector = fp32 MFMA dense peak
F16_MFMA_PEAK_TF = 2500.0      # dense fp16/bf16 MFMA peak
PROFILE_TRAFFIC = [os.path.join(ROOT, "profiles", n) for n in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json")]


def pkg(sub=None):
    return importlib.import_module("3pu_pytorch_amd" + ("." + sub if sub else ""))


def poisson_sphere(seed, n, dev, ops=None):
    """Config C2's input (3pu_pytorch_amd/utils/workloads.py)."""
    return pkg("utils.workloads").poisson_sphere(seed, n, dev)


class KernelTimer(object):
    """Wraps methods of the HIP backend with events on the launch stream (torch's current stream IS
    the stream the library launches on) and records (ms, shape info) per call."""

    def __init__(self, backend):
        self.be = backend
        self.reps = []          # one {name: [(e0, e1, info), ...]} per repetition (= one untimed step)
        self.saved = {}

    def begin_rep(self):
        self.reps.append({})

    def wrap(self, name, info):
        fn = getattr(self.be, name)
        self.saved[name] = fn

        def wrapper(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            self.reps[-1].setdefault(name, []).append((e0, e1, info(*a, **kw)))
            return out
        setattr(self.be, name, wrapper)

    def restore(self):
        for name in self.saved:
            try:
                delattr(self.be, name)          # instance attribute shadows the class method
            except AttributeError:
                pass

    def total(self, name, pred=None):
        """(median over the repetitions of the per-step sum of the kernel's launch times, the launch shapes of a
        step, [min, max] of the per-step sums).  `name`: one wrapped method or a tuple of them."""
        names = name if isinstance(name, tuple) else (name,)
        sums, shapes = [], []
        for rep in self.reps:
            sel = [(a.elapsed_time(b), s) for nm in names for a, b, s in rep.get(nm, []) if pred is None or pred(s)]
            if sel:
                sums.append(sum(ms for ms, _ in sel))
                shapes = [s for _, s in sel]
        if not sums:
            return 0.0, [], None
        return float(np.median(sums)), shapes, [float(min(sums)), float(max(sums))]


def other_rooflines(ops, pipe, net, clouds, npnt, r, traffic, reps=5):
    """`reps` extra UNTIMED single-stream steps (after one discarded warm-up step: the first step after the
    multi-stream timed region runs with cold caches and clocks) with events around every hand-written network
    kernel; every ms_per_step is the MEDIAN over the steps of that kernel's per-step sum, the spread is reported."""
    be = ops.BACKEND
    kt = KernelTimer(be)
    kt.wrap("dense_edge_conv", lambda x, idx, off, k, mlps, out, **kw: (x.shape[0], x.shape[1], k, 0))
    # (blocks 1-3 of a Level: the same kernel with the later prep convolutions folded into its write-out -- fold_n =
    # 72 / 48 / 24 more outputs over the block's 60-channel row)
    kt.wrap("dense_edge_conv_fold", lambda x, idx, off, k, mlps, out, fold_w, *rest, **kw:
            (x.shape[0], x.shape[1], k, fold_w.shape[0]))
    kt.wrap("knn_graph", lambda k, x, layout=None: (x.shape[0], x.shape[1], x.shape[2], k))
    kt.wrap("regress_tail", lambda a, c, *rest, **kw: (a.shape[0], c.shape[0]))
    kt.wrap("linear_small", lambda x, w, b, relu, **kw: (x.numel() // x.shape[-1], x.shape[-1], w.shape[0]))
    kt.wrap("interlevel_skip", lambda xyz, feat, pxyz, pfeat, pts_of, idx, **kw: (feat.shape[0], feat.shape[1],
                                                                                  idx.shape[2], feat.shape[2]))
    kt.wrap("linear_wide", lambda x, w, b: (x.numel() // x.shape[-1], x.shape[-1], w.shape[0]))
    kt.wrap("linear_lift", lambda x, w, b, relu, **kw: (x.numel() // x.shape[-1], x.shape[-1], w.shape[0],
                                                        kw.get("also") is not None))
    kt.wrap("fps", lambda xyz, npoint, n_arr=None, m_arr=None: (xyz.shape[0], xyz.shape[1], npoint))
    kt.wrap("knn", lambda k, q, p, unique, *a, **kw: (q.shape[0], q.shape[1], p.shape[1], q.shape[2], k))
    try:
        for rep in range(reps + 1):
            kt.begin_rep()
            pipe.upsample(net, clouds, npnt, r, 3, final_fps=False, check_small=False, optimistic_graph=True)
            torch.cuda.synchronize()
        del kt.reps[0]
    finally:
        kt.restore()
    out = []

    def tr(key):
        v = (traffic or {}).get("others_per_step", {}).get(key)
        return None if v is None else v.get("traffic_bytes_per_step")

    ms, shp, spread = kt.total(("dense_edge_conv", "dense_edge_conv_fold"))
    if shp:
        # executed matrix-core work of the lane-per-point kernel (csrc/dense_edge_conv.hip, dec_fused4_kernel): per
        # 64-point step 108 v_mfma_f32_4x4x1 (512 FLOP each) per neighbour slot + 288 per-point ones (centre terms, z
        # and c2 tables); no padded rows or k slots -- the only padding is the lanes beyond n in a patch's last step
        # + (folded launches) 60 input channels x fold_n / 4 output groups more of them for the prep convolutions
        steps = lambda n: -(-n // 64)
        ex = sum(p * steps(n) * (108 * k + 288 + 15 * f) * 512.0 for p, n, k, f in shp)
        useful = sum(p * n * (108 * k + 288 + 15 * f) * 8.0 for p, n, k, f in shp)        # 512 / 64 lanes = 8 FLOP per lane
        alg = sum(p * n * (k * 3168.0 + 120.0 * f) for p, n, k, f in shp)   # SURVEY 8a a9 (un-hoisted) + the prep convolutions
        ach = ex / (ms * 1e-3) / 1e12
        out.append({"kernel": "dec_fused4_kernel (DenseEdgeConv, fp32 MFMA 4x4x1, lane per point), %d launches/step" % len(shp),
                    "bound": "mfma", "achieved": ach, "peak": FP32_PEAK_TF, "unit": "TFLOP/s", "frac": ach / FP32_PEAK_TF,
                    "useful_frac": useful / ex * ach / FP32_PEAK_TF,
                    "basis": "executed v_mfma_f32_4x4x1 FLOPs: the block (hoisted formulation) + the prep convolutions folded "
                             "into 3 launches of 4 (60 x fold_n products per point); useful_frac discounts the idle lanes "
                             "of a patch's last 64-point step (312 of 320)",
                    "ms_per_step": ms, "ms_per_step_min_max": spread, "executed_flop_per_step": ex,
                    "survey_model_flop_per_step": alg, "traffic": tr("dec_fused")})
    ms, shp, spread = kt.total("knn_graph")
    if shp:
        flop = sum(p * n * n * (2.0 * c + 3.0) for p, n, c, k in shp)   # B*M*N*(2C+3), M = N
        ach = flop / (ms * 1e-3) / 1e12
        t_graph = [tr(kk) for kk in ("knn_graph_slab_kernel", "knn_slab_order_kernel")]
        t_graph = sum(t_graph) if all(v is not None for v in t_graph) else tr("knn_graph_key_kernel")
        out.append({"kernel": "knn_slab_order_kernel + knn_graph_slab_kernel (feature kNN graph k=33, slab form, exact) incl. "
                              "the output allocation, %d launches/step" % len(shp),
                    "bound": "valu", "achieved": ach, "peak": FP32_PEAK_TF, "unit": "TFLOP/s", "frac": ach / FP32_PEAK_TF,
                    "basis": "SURVEY 8d compute model B*M*N*(2C+3) FLOP against the fp32 vector peak",
                    "ms_per_step": ms, "ms_per_step_min_max": spread, "model_flop_per_step": flop, "traffic": t_graph})
    ms, shp, spread = kt.total("regress_tail")
    if shp:
        ex = sum(m * rr * 2.0 * (128 * 128 + 128 * 64 + 64 * 16) for m, rr in shp)
        ach = ex / (ms * 1e-3) / 1e12
        split = bool(ops.BACKEND.split_bf16())
        if split:
            # six v_mfma_f32_16x16x32_bf16 per (output tile, slab pair) in layers 2 and 3, fp32 MFMAs in layer 4: the
            # EXECUTED matrix FLOP are 6x the layers' 2 * cin * cout per row, priced on the bf16 peak
            ex_b = sum(m * rr * 2.0 * 6 * (128 * 128 + 128 * 64) for m, rr in shp)
            out.append({"kernel": "regress_tail_sb_kernel (128->128->64->3, fp32 operands as 3 bf16 terms, 6 partial products on "
                                  "v_mfma_f32_16x16x32_bf16; two replicas per weight operand), %d launches/step" % len(shp),
                        "bound": "mfma", "achieved": ex_b / (ms * 1e-3) / 1e12, "peak": F16_MFMA_PEAK_TF, "unit": "TFLOP/s",
                        "frac": ex_b / (ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TF,
                        "basis": "executed bf16 MFMA FLOPs (6 partial products per fp32 product) against the dense bf16 peak; "
                                 "on the fp32 model (2 * cin * cout per row) the kernel delivers %.1f TFLOP/s = %.2f of the fp32 "
                                 "MFMA peak" % (ach, ach / FP32_PEAK_TF),
                        "ms_per_step": ms, "ms_per_step_min_max": spread, "executed_flop_per_step": ex_b,
                        "traffic": tr("regress_tail_sb_kernel")})
        else:
            out.append({"kernel": "regress_tail_kernel (128->128->64->3, fp32 MFMA; TPU3_SPLIT_BF16=0), %d launches/step" % len(shp),
                        "bound": "mfma", "achieved": ach, "peak": FP32_PEAK_TF, "unit": "TFLOP/s", "frac": ach / FP32_PEAK_TF,
                        "basis": "executed MFMA FLOPs (last layer padded 3 -> 16 rows)", "ms_per_step": ms,
                        "ms_per_step_min_max": spread, "executed_flop_per_step": ex, "traffic": tr("regress_tail_kernel")})
    ms, shp, spread = kt.total("linear_wide")
    if shp:
        # executed: per 16 rows 17 slabs x 8 output tiles x 4 MFMAs (264 channels padded to 272)
        ex = sum(-(-m // 16) * 17 * 8 * 4 * 2048.0 for m, cin, cout in shp)
        ach = ex / (ms * 1e-3) / 1e12
        if bool(ops.BACKEND.split_bf16()) and os.environ.get("TPU3_SPLIT_BF16_WIDE", "1") not in ("0", ""):
            # per 16 rows: 9 slabs of 32 channels (264 padded to 288) x 8 output tiles x 6 partial products of 16384 FLOP
            ex_b = sum(-(-m // 16) * 9 * 8 * 6 * 16384.0 for m, cin, cout in shp)
            alg = sum(m * 2.0 * cin * cout for m, cin, cout in shp)
            byt = sum(m * 4.0 * (cin + cout) for m, cin, cout in shp)
            out.append({"kernel": "linear_wide_sb_kernel (up_layer1 per point, 264 -> 128, fp32 operands as 3 bf16 terms, 6 partial "
                                  "products on v_mfma_f32_16x16x32_bf16, weight slabs through an LDS ring), %d launches/step" % len(shp),
                        "bound": "mfma", "achieved": ex_b / (ms * 1e-3) / 1e12, "peak": F16_MFMA_PEAK_TF, "unit": "TFLOP/s",
                        "frac": ex_b / (ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TF,
                        "basis": "executed bf16 MFMA FLOPs (6 partial products per fp32 product, 264 of 288 k slots useful) against "
                                 "the dense bf16 peak; on the fp32 model (2 * cin * cout per row) %.1f TFLOP/s = %.2f of the fp32 MFMA "
                                 "peak; its rows in and out are %.0f GB/s = %.2f of the HBM peak" % (
                                     alg / (ms * 1e-3) / 1e12, alg / (ms * 1e-3) / 1e12 / FP32_PEAK_TF,
                                     byt / (ms * 1e-3) / 1e9, byt / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS),
                        "ms_per_step": ms, "ms_per_step_min_max": spread, "executed_flop_per_step": ex_b,
                        "traffic": tr("linear_wide_sb_kernel")})
            ms = None
    if shp and ms is not None:
        out.append({"kernel": "linear_wide_kernel (up_layer1 per point, 264 -> 128, fp32 MFMA; TPU3_SPLIT_BF16=0), %d launches/step" % len(shp),
                    "bound": "mfma", "achieved": ach, "peak": FP32_PEAK_TF, "unit": "TFLOP/s", "frac": ach / FP32_PEAK_TF,
                    "basis": "executed v_mfma_f32_16x16x4 FLOPs (264 of 272 k slots useful)", "ms_per_step": ms, "ms_per_step_min_max": spread,
                    "executed_flop_per_step": ex, "traffic": tr("linear_wide_kernel")})
    ms, shp, spread = kt.total("linear_lift")
    if shp:
        byt = sum(m * 4.0 * (cin + cout * (2 if also else 1)) for m, cin, cout, also in shp)
        ach = byt / (ms * 1e-3) / 1e9
        out.append({"kernel": "linear_lift_kernel (layer0, 3 -> 24, also stored into the feature buffer), %d launches/step"
                              % len(shp),
                    "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "basis": "4*C_in B read, 4*C_out B written per destination and row", "ms_per_step": ms, "ms_per_step_min_max": spread,
                    "algorithmic_bytes_per_step": byt, "traffic": tr("linear_lift_kernel")})
    ms, shp, spread = kt.total("linear_small")
    if shp:
        byt = sum(m * 4.0 * (cin + cout) for m, cin, cout in shp)
        ach = byt / (ms * 1e-3) / 1e9
        out.append({"kernel": "linear_small_kernel (prep convolutions 84/144/204 -> 24), %d launches/step" % len(shp),
                    "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "basis": "4*(C_in + C_out) B per row read once / written once", "ms_per_step": ms, "ms_per_step_min_max": spread,
                    "algorithmic_bytes_per_step": byt, "traffic": tr("linear_small_kernel")})
    ms, shp, spread = kt.total("interlevel_skip")
    if shp:
        byt = sum(b * n * (2.0 * 4 * c) for b, n, k, c in shp)           # own row read once, written once
        ach = byt / (ms * 1e-3) / 1e9
        out.append({"kernel": "skip_fused_kernel (inter-level skip, one launch, a 16-wave workgroup per patch), %d launches/step"
                              % len(shp),
                    "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "basis": "2 x 4C B per point: the own row read once and written once (its second read, half a workgroup's "
                             "life after the first, comes from the Infinity Cache; the 2K gathered neighbour rows per point "
                             "from L2).  Rounds 1-4 ran two kernels = 3 x 4C B per point: on that basis this entry would read "
                             "%.2f" % (1.5 * ach / HBM_PEAK_GBS),
                    "ms_per_step": ms, "ms_per_step_min_max": spread, "algorithmic_bytes_per_step": byt, "traffic": tr("skip_")})
    ms, shp, spread = kt.total("fps", lambda s: s[2] >= 256)
    if shp:
        rounds = sum(m - 1 for _, _, m in shp)
        entry = {"kernel": "rl_main_kernel (per-level resampling FPS, several samples per round), %d launches/step"
                           % len(shp),
                 "bound": "latency", "us_per_sample": ms * 1e3 / max(1, rounds), "ms_per_step": ms,
                 "ms_per_step_min_max": spread,
                 "sets_per_launch": [b for b, _, _ in shp],
                 "basis": "dependent chain: one workgroup (= one compute unit) per set, 256 sets at a time; "
                          "us_per_sample = launch time / samples per SET (all sets of a launch share it)"}
        # rounds of the LARGEST level set (24 960 -> 4992), one wave of sets (<= 256: every set has its compute unit),
        # from the kernel's own counters; the floor is the round's update phase as pure instruction issue
        try:
            import ctypes
            seen = {}
            real = be.fps

            def spy(xyz, npoint, n_arr=None, m_arr=None):
                if npoint >= 256 and xyz.size(1) <= 25600 and xyz.size(1) > seen.get("n", 0):
                    seen.update(n=xyz.size(1), args=(xyz[:192].clone(), npoint, None if n_arr is None else n_arr[:192].clone(),
                                                    None if m_arr is None else m_arr[:192].clone()))
                return real(xyz, npoint, n_arr, m_arr)
            be.fps = spy
            try:
                pipe.upsample(net, clouds[:4], npnt, r, 3, final_fps=False, check_small=False, optimistic_graph=True)
            finally:
                del be.fps
            if seen:
                st = torch.zeros(52, dtype=torch.int64, device=clouds.device)
                lib = pkg("_lib").lib()
                best = None
                for it in range(3):
                    st.zero_()
                    lib.tpu3_debug_fps_level_stats(ctypes.c_void_p(st.data_ptr()))
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    real(*seen["args"])
                    e1.record()
                    torch.cuda.synchronize()
                    best = e0.elapsed_time(e1) if best is None else min(best, e0.elapsed_time(e1))
                rnd, smp = int(st[0]), int(st[1])
                if rnd:
                    entry.update({"largest_set": "%d -> %d points, %d sets in one wave of workgroups (operator time incl. "
                                                 "Morton sort and set-up)" % (seen["n"], seen["args"][1], seen["args"][0].size(0)),
                                  "rounds": rnd, "samples_per_round": smp / rnd, "us_per_round": best * 1e3 / rnd,
                                  "us_per_round_floor": 3.3, "x_over_floor": best * 1e3 / rnd / 3.3,
                                  "floor_model": "a round's update phase alone: ~10 sample-updates of 25 x 64 points per SIMD "
                                                 "= 7.9 k cycles of VALU issue on every SIMD of the set's compute unit = 3.3 us "
                                                 "at 2.4 GHz (a model of THIS algorithm, not a hardware bound)"})
        except Exception as e:                                           # noqa: BLE001 (reported, not hidden)
            entry["rounds"] = "failed: %s" % (str(e).splitlines()[0][:120])
        out.append(entry)
    ms, shp, spread = kt.total("knn")
    if shp:
        out.append({"kernel": "knn_insert / knn_select / knn_sort kernels (patch extraction, outlier filter, inter-level k=5), "
                              "%d launches/step" % len(shp), "bound": "valu", "ms_per_step": ms, "ms_per_step_min_max": spread})
    return out


def extras_block(args, ops, pipe, ups, net, clouds, dev, N, npnt, r, nets=None, sides=None):
    """Secondary numbers the judge asked for next to the headline (all untimed w.r.t. `value`)."""
    ex = {}
    # config C4's per-rank share: 8 clouds per GPU per step, same stream arrangement as the timed region
    try:
        sub = clouds[:8]
        # (the timed region's own streams: a second set would share hardware queues with the first -- 24 streams on 16
        # queues -- and two sub-batches that land on one queue run one after the other: 86 instead of 74 ms per step)
        nets8 = nets if nets is not None else (
            [torch.cuda.Stream(device=dev) for _ in range(args.net_streams)] if args.net_streams > 1 else None)
        side8 = sides if sides is not None or args.no_overlap else [torch.cuda.Stream(device=dev) for _ in range(args.fps_streams)]
        def step8(i):
            return pipe.upsample(net, sub, npnt, r, 3, fps_stream=None if side8 is None else side8[i % len(side8)],
                                 net_streams=nets8, sub_batch=args.sub_batch, check_small=False, optimistic_graph=True)
        for i in range(3):
            step8(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K8 = 30     # (as many steps as it takes to amortise the last step's un-overlapped final FPS, like the timed region's 20)
        for i in range(K8):
            step8(i)
        torch.cuda.synchronize()
        ex["ms_per_step_8clouds"] = (time.perf_counter() - t0) / K8 * 1e3
    except Exception as e:                                               # noqa: BLE001
        ex["ms_per_step_8clouds"] = "failed: %s" % (str(e).splitlines()[0][:120])
    # 1-cloud latency: the reference's test() loop handles one cloud at a time (main.py:340-389)
    one = clouds[:1]
    ts = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.upsample(net, one, npnt, r, 3, check_small=False, optimistic_graph=True)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ex["latency_ms_1cloud_eager"] = float(np.median(ts[1:]))
    # ... and the same call as a hipGraph (pipeline.GraphedUpsample: captured once, checked after every replay)
    try:
        fast = pipe.GraphedUpsample(net, tuple(one.shape), npnt, r, 3)
        ref = pipe.upsample(net, one, npnt, r, 3)
        same = bool(torch.equal(fast(one, clone=True), ref))
        ts = []
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fast(one)                                  # (check=True: synchronises and reads the event words)
            ts.append((time.perf_counter() - t0) * 1e3)
        ex["latency_ms_1cloud_graph"] = float(np.median(ts[1:]))
        ex["latency_1cloud_graph_note"] = ("hipGraph replay incl. input copy and post-replay checks; result == eager: %s.  "
                                           "(r5: no gain -- the ~390 kernels of a cloud are dependent, the gaps are the "
                                           "GPU's own dispatch-after-drain, not host launches)" % same)
        ex["latency_ms_1cloud"] = min(ex["latency_ms_1cloud_eager"], ex["latency_ms_1cloud_graph"])
        del fast
    except Exception as e:                                               # noqa: BLE001 (reported, not hidden)
        ex["latency_ms_1cloud"] = ex["latency_ms_1cloud_eager"]
        ex["latency_1cloud_graph_note"] = "graph capture failed: %s" % (str(e).splitlines()[0][:120])
    # Chamfer between two 80 000-point clouds (the evaluation metric's kernel)
    ml = pkg("network.model_loss")
    a = poisson_sphere(1000, N, dev, ops).transpose(2, 1).contiguous().repeat(1, r, 1)
    a = a + 0.001 * torch.randn_like(a)
    b = a.flip(1).contiguous() + 0.0005
    crit = ml.ChamferLoss()
    ts = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        crit(a, b)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ex["chamfer_80k_ms"] = float(np.median(ts[1:]))
    # config C3: one optimisation step, B = 32 patches, ratio 16 (and 4), eager and as a hipGraph
    try:
        import types
        model_mod = pkg("model")
        g = torch.Generator().manual_seed(7)
        inp = torch.randn(32, 312, 3, generator=g)
        inp = (inp / inp.norm(dim=2, keepdim=True)).transpose(2, 1).contiguous().to(dev)
        train = {}
        for ratio in (16, 4):
            lab = torch.randn(32, 312 * ratio, 3, generator=g)
            lab = (lab / lab.norm(dim=2, keepdim=True)).transpose(2, 1).contiguous().to(dev)
            for mode in ("eager", "graph"):
                torch.manual_seed(0)
                tnet = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev)
                model = model_mod.Model(tnet, "train", types.SimpleNamespace(lr_init=1e-3, ckpt=None,
                                                                             graph_steps=(mode == "graph")))
                try:
                    for _ in range(3):
                        model.set_input(inp, ratio, label_pc=lab)
                        model.optimize()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(10):
                        model.set_input(inp, ratio, label_pc=lab)
                        model.optimize()
                    torch.cuda.synchronize()
                    train["x%d_%s" % (ratio, mode)] = (time.perf_counter() - t0) * 100.0
                except Exception as e:                                   # noqa: BLE001 (reported, not hidden)
                    train["x%d_%s" % (ratio, mode)] = "failed: %s" % (str(e).splitlines()[0][:120])
        ex["train_step_ms"] = train
    except Exception as e:                                               # noqa: BLE001
        ex["train_step_ms"] = "failed: %s" % (str(e).splitlines()[0][:120])
    # config C5 (stress): one 80 000-point cloud, num_point = 1024 (234 outer patches), 16x -> 1.28 M points,
    # feature MLPs on fp16-operand MFMA; FPS / kNN stay fp32.  One warm-up, one timed run.
    try:
        c5 = poisson_sphere(5, 80000, dev, ops)
        net.set_mlp_precision("f16", activations="f16")     # fp16 operands AND fp16 feature buffers (half the HBM bytes)
        res = {}
        for final in (False, True):
            for it in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = pipe.upsample(net, c5, 1024, 16, 3, final_fps=final, check_small=False, optimistic_graph=True)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            res["network_stages_ms" if not final else "total_ms"] = dt * 1e3
        assert tuple(out.shape) == (1, 3, 1280000) and bool(torch.isfinite(out).all())
        res["points_per_s"] = 1280000 / (res["total_ms"] * 1e-3)
        # the fp16-operand cloud against the fp32 cloud of the same weights (every 16th point of the 3.83 M merged ones)
        m16 = pipe.upsample(net, c5, 1024, 16, 3, final_fps=False, check_small=False, optimistic_graph=True)
        # (the same operands with fp32 feature buffers: what the fp16 storage costs / buys on its own)
        net.set_mlp_precision("f16", activations="f32")
        for it in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipe.upsample(net, c5, 1024, 16, 3, final_fps=False, check_small=False, optimistic_graph=True)
            torch.cuda.synchronize()
            res["network_stages_ms_f32_buffers"] = (time.perf_counter() - t0) * 1e3
        net.set_mlp_precision("f32")
        m32 = pipe.upsample(net, c5, 1024, 16, 3, final_fps=False, check_small=False, optimistic_graph=True)
        net.set_mlp_precision("f16", activations="f16")
        s16 = m16.transpose(2, 1)[:, ::16].contiguous()
        s32 = m32.transpose(2, 1)[:, ::16].contiguous()
        d1, _, d2, _ = pkg("network.model_loss").nndistance(s16, s32)
        _, dself, _ = ops.knn_query(2, s32[:, :20000].contiguous(), s32[:, :20000].contiguous(), unique=False,
                                    want_grouped=False)
        res["chamfer_f16_vs_f32"] = float(d1.mean() + d2.mean())
        res["f32_cloud_spacing_sq_median"] = float(dself[:, :, 1].clamp_min(0).median())
        del m16, m32
        res["config"] = ("C5: 1 cloud x 80000 pts, num_point=1024, up_ratio=16, fp16-operand MFMA feature MLPs, feature "
                         "buffers stored as fp16, 1 GPU")
        ex["c5_stress"] = res
    except Exception as e:                                               # noqa: BLE001
        ex["c5_stress"] = "failed: %s" % (str(e).splitlines()[0][:160])
    finally:
        net.set_mlp_precision("f32")
    return ex


def point_kernel_rooflines(dev, reps=7):
    """(r6) The point-set kernels of the C ABI that are NOT launched by an inference step -- nm-distance forward /
    backward (the metric's Chamfer and the training loss), ball query, gather forward / backward -- each at the sizes
    SURVEY 8(a) names, timed with HIP events on the launch stream (median of `reps` calls after a warm-up) and priced
    with SURVEY 8(d)'s formulas: nm-distance forward against the 157.3 TF fp32 vector peak on 2*B*n*m*8 FLOP (for the
    grid-pruned form also the EXECUTED pair evaluations from the kernel's own counters: it is a work-skipping exact
    search, so its fraction of the scan's model may exceed 1), the others against 8 TB/s on their algorithmic bytes.
    Several of these are a few microseconds long: launch-bound, and reported as such."""
    import ctypes
    sampling, losses, tlib = pkg("sampling"), pkg("losses"), pkg("_lib").lib()
    g = torch.Generator(device=dev).manual_seed(3)

    def sphere(b, n, scale=1.0):
        x = torch.randn((b, n, 3), device=dev, generator=g)
        return (x / x.norm(dim=2, keepdim=True) * scale).contiguous()

    def timed(fn):
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
        return float(np.median(ts)), [float(min(ts)), float(max(ts))]

    out = []
    note = "not part of an inference step: ms_per_call of one C-ABI call on an otherwise idle device"
    # ---- nm-distance forward: both forms at every size --------------------------------------------------------------
    for b, n, m in ((32, 624, 624), (32, 4992, 4992), (1, 80000, 80000)):
        x1, x2 = sphere(b, n), sphere(b, m, 1.01)
        d1, d2 = torch.empty((b, n), device=dev), torch.empty((b, m), device=dev)
        i1 = torch.empty((b, n), dtype=torch.int32, device=dev)
        i2 = torch.empty((b, m), dtype=torch.int32, device=dev)
        flop = 2.0 * b * n * m * 8
        auto_grid = None
        for form in (-1, 0, 1):
            tlib.tpu3_debug_nmdist_form(form)
            tlib.tpu3_debug_nmdist_grid_calls(1)
            try:
                ms, mm = timed(lambda: losses.nmdistance_forward(x1, x2, d1, d2, i1, i2))
                took_grid = tlib.tpu3_debug_nmdist_grid_calls(1) > 0
                if form == -1:
                    auto_grid = took_grid
                    continue
                if form == 1 and not took_grid:
                    continue
                ent = {"kernel": ("nmg_query_kernel + 5 build kernels (csrc/nmdist_grid.hip: grid-pruned exact search)"
                                  if took_grid else "nmdist_fwd_kernel / nmdist_fwd_split_kernel (csrc/nmdistance.hip: the reference's scan)"),
                       "call": "tpu3_nmdist_fwd_f32(b=%d, n=%d, m=%d), both directions" % (b, n, m),
                       "form": "grid" if took_grid else "scan", "automatic_choice": auto_grid == took_grid,
                       "ms_per_call": ms, "ms_per_call_min_max": mm, "ms_per_step": None, "scope": note,
                       "bound": "valu", "unit": "TFLOP/s", "peak": FP32_PEAK_TF,
                       "model_flop_per_call": flop, "achieved": flop / (ms * 1e-3) / 1e12,
                       "frac": flop / (ms * 1e-3) / 1e12 / FP32_PEAK_TF,
                       "algorithmic_bytes_per_call": 20.0 * b * (n + m),
                       "basis": "SURVEY 8(d): 2*B*n*m*8 FLOP (3 sub, mul, 2 fma, compare per pair and direction) / call time"}
                if took_grid:
                    st = torch.zeros(4, dtype=torch.int64, device=dev)
                    tlib.tpu3_debug_nmdist_grid_stats(ctypes.c_void_p(st.data_ptr()))
                    losses.nmdistance_forward(x1, x2, d1, d2, i1, i2)
                    torch.cuda.synchronize()
                    tlib.tpu3_debug_nmdist_grid_stats(None)
                    waves, _, passed, searched = [int(v) for v in st.cpu()]
                    ex = searched * 64.0 * 64.0 * 8
                    ent.update({"tiles_searched_per_query_wave": searched / max(waves, 1),
                                "tiles_tested_per_query_wave": passed / max(waves, 1),
                                "executed_flop_per_call": ex, "executed_frac": ex / (ms * 1e-3) / 1e12 / FP32_PEAK_TF,
                                "frac_note": "`frac` prices the call on the SCAN's model (what the reference executes); "
                                             "the search evaluates %.2f %% of those pairs -- same distances and indices"
                                             % (100.0 * ex / flop)})
                out.append(ent)
            finally:
                tlib.tpu3_debug_nmdist_form(-1)
    # ---- nm-distance backward ----------------------------------------------------------------------------------------
    for b, n, m in ((32, 624, 624), (1, 80000, 80000)):
        x1, x2 = sphere(b, n), sphere(b, m, 1.01)
        d1, d2 = torch.empty((b, n), device=dev), torch.empty((b, m), device=dev)
        i1 = torch.empty((b, n), dtype=torch.int32, device=dev)
        i2 = torch.empty((b, m), dtype=torch.int32, device=dev)
        losses.nmdistance_forward(x1, x2, d1, d2, i1, i2)
        g1, g2 = torch.ones_like(d1), torch.ones_like(d2)
        gx1, gx2 = torch.zeros_like(x1), torch.zeros_like(x2)
        ms, mm = timed(lambda: losses.nmdistance_backward(x1, x2, gx1, gx2, g1, g2, i1, i2))
        byts = float(b) * (n + m) * (12 + 12 + 4 + 4 + 24)
        out.append({"kernel": "nmdist_bwd_kernel x 2 (csrc/nmdistance.hip)", "call": "tpu3_nmdist_bwd_f32(b=%d, n=%d, m=%d)" % (b, n, m),
                    "ms_per_call": ms, "ms_per_call_min_max": mm, "ms_per_step": None, "scope": note, "bound": "hbm",
                    "unit": "GB/s", "peak": HBM_PEAK_GBS, "algorithmic_bytes_per_call": byts,
                    "achieved": byts / (ms * 1e-3) / 1e9, "frac": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "basis": "SURVEY 8(d): B*(n+m)*(12+12+4+4 + 24 atomic) B / call time (fp32 atomics into L2; two launches)"})
    # ---- ball query (exported by the reference, called by nothing in it: sampling.cpp:59-81) --------------------------------
    b, m, n, ns = 48, 312, 5000, 32
    xyz, q = sphere(b, n), sphere(b, m)
    ms, mm = timed(lambda: sampling.ball_query(q, xyz, 0.1, ns))
    byts = float(b) * (12 * n + 12 * m + 4 * m * ns)
    out.append({"kernel": "ball_query_kernel (csrc/ball_query.hip)", "call": "tpu3_ball_query(b=%d, m=%d, n=%d, r=0.1, nsample=%d)" % (b, m, n, ns),
                "ms_per_call": ms, "ms_per_call_min_max": mm, "ms_per_step": None, "scope": note, "bound": "hbm", "unit": "GB/s",
                "peak": HBM_PEAK_GBS, "algorithmic_bytes_per_call": byts, "achieved": byts / (ms * 1e-3) / 1e9,
                "frac": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "basis": "SURVEY 8(d): B*(12N + 12M + 4*M*nsample) B / call time (incl. the result tensor's allocation; "
                         "the kernel itself scans N candidates per query: %.1f G pair tests" % (b * m * n / 1e9)})
    # ---- gather forward / backward (sampling.cpp:37-53) -------------------------------------------------------------
    for b, c, n, m in ((1, 3, 239616, 80000), (48, 3, 24960, 4992)):
        pts = torch.randn((b, c, n), device=dev, generator=g)
        idx = torch.randint(0, n, (b, m), device=dev, generator=g, dtype=torch.int32)
        outp = torch.empty((b, c, m), device=dev)
        ms, mm = timed(lambda: sampling.gather_forward(b, c, n, m, pts, idx, outp))
        byts = float(b) * c * (4 * m + 4 * m) + 4.0 * b * m
        out.append({"kernel": "gather_fwd_kernel (csrc/gather.hip)", "call": "tpu3_gather_fwd(b=%d, c=%d, n=%d, npoints=%d)" % (b, c, n, m),
                    "ms_per_call": ms, "ms_per_call_min_max": mm, "ms_per_step": None, "scope": note, "bound": "hbm", "unit": "GB/s",
                    "peak": HBM_PEAK_GBS, "algorithmic_bytes_per_call": byts, "achieved": byts / (ms * 1e-3) / 1e9,
                    "frac": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "basis": "SURVEY 8(d): B*C*(4m read-gather + 4m write) + 4*B*m B / call time"})
        go = torch.randn((b, c, m), device=dev, generator=g)
        gp = torch.zeros((b, c, n), device=dev)
        ms, mm = timed(lambda: sampling.gather_backward(b, c, n, m, go, idx, gp))
        out.append({"kernel": "gather_bwd_kernel (csrc/gather.hip)", "call": "tpu3_gather_bwd(b=%d, c=%d, n=%d, npoints=%d)" % (b, c, n, m),
                    "ms_per_call": ms, "ms_per_call_min_max": mm, "ms_per_step": None, "scope": note, "bound": "hbm", "unit": "GB/s",
                    "peak": HBM_PEAK_GBS, "algorithmic_bytes_per_call": byts, "achieved": byts / (ms * 1e-3) / 1e9,
                    "frac": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "basis": "SURVEY 8(d): the forward's bytes, the writes as fp32 atomics into the zeroed gradient / call time"})
    return out


def train_rooflines(ops, ups, dev, ratio=16, reps=3):
    """Config C3 (BASELINE: one training step, batch 32 patches of 312 points, up_ratio 16, Chamfer fwd + bwd): an
    EAGER step with events around the hand-written training kernels (a hipGraph replay cannot be bracketed per
    kernel) -> each kernel's ms per step and the dominant one against its bound."""
    import types
    model_mod = pkg("model")
    be = ops.BACKEND
    g = torch.Generator().manual_seed(7)
    inp = torch.randn(32, 312, 3, generator=g)
    inp = (inp / inp.norm(dim=2, keepdim=True)).transpose(2, 1).contiguous().to(dev)
    lab = torch.randn(32, 312 * ratio, 3, generator=g)
    lab = (lab / lab.norm(dim=2, keepdim=True)).transpose(2, 1).contiguous().to(dev)
    torch.manual_seed(0)
    tnet = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev)
    model = model_mod.Model(tnet, "train", types.SimpleNamespace(lr_init=1e-3, ckpt=None, graph_steps=False))
    for _ in range(2):
        model.set_input(inp, ratio, label_pc=lab)
        model.optimize()
    kt = KernelTimer(be)
    shape0 = lambda t, *a, **kw: tuple(t.shape)
    kt.wrap("dec_train_forward", lambda x, idx, off, w: tuple(x.shape))
    kt.wrap("dec_train_backward", lambda x, idx, off, w, arg, gy: tuple(x.shape))
    kt.wrap("dec_train_wgrad", lambda x, S, ws: (S.shape[0], ws.shape[0]))
    kt.wrap("knn_graph", lambda k, x, layout=None, optimistic=None: tuple(x.shape))
    kt.wrap("linear_wgrad_bias", lambda x, dy, want_bias=True: (x.shape[0], x.shape[1], dy.shape[1]))
    kt.wrap("linear_dgrad", lambda dy, w: (dy.shape[0], w.shape[0], w.shape[1]))
    kt.wrap("interlevel_skip_train", lambda xyz, feat, *a, **kw: tuple(feat.shape))
    kt.wrap("interlevel_skip_backward", lambda gg, *a, **kw: tuple(gg.shape))
    walls = []
    try:
        for _ in range(reps):
            kt.begin_rep()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.set_input(inp, ratio, label_pc=lab)
            model.optimize()
            torch.cuda.synchronize()
            walls.append((time.perf_counter() - t0) * 1e3)
    finally:
        kt.restore()
    names = {"dec_train_backward": "dec_train_bwd_kernel (DenseEdgeConv block backward: recompute, route to the arg-max edges, "
                                   "three layers back, the edge parts' weight gradients G^T Z accumulated on the matrix cores)",
             "dec_train_forward": "dec_train_fwd_kernel", "dec_train_wgrad": "linear_wgrad_all_kernel (S^T [x | 1]) + "
             "dec_wgrad_assemble_kernel (adds the backward workgroups' blocks)", "knn_graph": "knn_graph kernels (feature graphs, exact form)",
             "linear_wgrad_bias": "linear_wgrad_all_kernel (per-point layers: dW and db in one streaming pass)",
             "linear_dgrad": "linear_dgrad_small_kernel", "interlevel_skip_train": "skip_dist + skip_apply (weights kept)",
             "interlevel_skip_backward": "skip_bwd_kernel"}
    rows = []
    for nm, label in names.items():
        ms, shp, spread = kt.total(nm)
        if shp:
            rows.append({"kernel": label, "launches_per_step": len(shp), "ms_per_step": ms, "ms_per_step_min_max": spread,
                         "_name": nm, "_shapes": shp})
    rows.sort(key=lambda r: -r["ms_per_step"])
    dom = None
    for r in rows:
        shp = r.pop("_shapes")
        nm = r.pop("_name")
        if nm == "dec_train_backward":
            # executed matrix-core work per 64 edges: 288 v_mfma_f32_4x4x1 (512 FLOP: the recomputed forward chains and
            # the three transposed products) + 80 v_mfma_f32_16x16x4_f32 (2048 FLOP: G^T Z); the kernel is bound by
            # its dependent loads and the 32 x 24-lane float atomics of the neighbour shares, not by this
            flop = sum(p * n * 32 / 64.0 * (288 * 512.0 + 80 * 2048.0) for p, n, _ in shp)
            ach = flop / (r["ms_per_step"] * 1e-3) / 1e12
            r.update({"bound": "mfma", "achieved": ach, "peak": FP32_PEAK_TF, "unit": "TFLOP/s",
                      "frac": ach / FP32_PEAK_TF, "algorithmic_flop_per_step": flop,
                      "basis": "executed fp32 matrix-core FLOP per launch / launch time (HIP events); latency-bound: a pass "
                               "of 8 points is index -> row loads, ~370 matrix instructions, 32 atomics per lane group"})
        if dom is None:
            dom = dict(r)
    return {"config": "C3: B=32 patches x 312 pts, up_ratio=%d (%d levels), Chamfer fwd+bwd, clip, Adam; eager step with events "
                      "around the hand-written training kernels" % (ratio, int(np.log2(ratio))),
            "step_ms_eager_with_events": float(np.median(walls)), "kernels": rows, "dominant": dom,
            "kernels_ms_sum": float(sum(r["ms_per_step"] for r in rows))}


def parity_block(ops, pipe, ups, dev, cpu_out, N=5000, npnt=312):
    """Config C1 on the device against the oracle-driven CPU output of the same cloud and weights."""
    from oracle import cpu_baseline
    return pkg("utils.parity").c1_parity(dev, cpu_baseline.c1_net(ups).to(dev), cpu_baseline.c1_cloud(0, N), cpu_out, npnt)


def parity_c2_block(dev):
    """The metric's own configuration against the reference driver's fixture AND the reference-vs-reference controls
    (tests/golden/c2_x16*.npz are data files; tests/test_c2_controls_cpu.py pins the controls on CPU)."""
    gdir = os.path.join(ROOT, "tests", "golden")
    fixtures = {}
    for k in ("ref", "alt", "alt2", "alt3"):
        path = os.path.join(gdir, "c2_x16%s.npz" % ("" if k == "ref" else "_" + k))
        if os.path.exists(path):
            fixtures[k] = np.load(path, allow_pickle=False)
    ups = pkg("network.upsampler")
    net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5)
    state = np.load(os.path.join(gdir, "net16_state.npz"), allow_pickle=False)
    net.load_state_dict({k: torch.from_numpy(state[k]) for k in state.files if k != "meta"})
    return pkg("utils.parity").c2_parity(dev, net.to(dev).eval(), fixtures)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clouds", type=int, default=32,
                    help="clouds per GPU per step (config C4 = --gpus 8 --clouds 8; more clouds in flight amortise "
                         "the final-FPS latency chain)")
    ap.add_argument("--shard", choices=["clouds", "patches"], default="clouds",
                    help="multi-GPU partition: whole clouds per rank (weak scaling), or the outer patches of ONE "
                         "cloud split across ranks (strong scaling; implies --clouds 1)")
    ap.add_argument("--fps_streams", type=int, default=4, help="side streams for the final FPS")
    ap.add_argument("--fps_per_sub_batch", action="store_true",
                    help="one final-FPS launch per network sub-batch instead of ONE per step")
    ap.add_argument("--net_streams", type=int, default=8,
                    help="sub-batches of clouds whose network stages run on concurrent streams")
    ap.add_argument("--sub_batch", type=int, default=4, help="clouds per network sub-batch")
    ap.add_argument("--no_overlap", action="store_true",
                    help="run the final FPS on the main stream instead of a side stream")
    ap.add_argument("--num_shape_point", type=int, default=5000)
    ap.add_argument("--num_point", type=int, default=312)
    ap.add_argument("--up_ratio", type=int, default=16)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_extras", action="store_true", help="skip rooflines_other / extras / parity (profiling runs)")
    ap.add_argument("--digest", action="store_true",
                    help="add result_digest = sha256 of every cloud of the last step's (gathered) result to the line "
                         "(functional checks of the N > 1 path: tests/test_bench_multirank.py)")
    ap.add_argument("--diag_skip_final_fps", action="store_true",
                    help="DIAGNOSTIC ONLY (the printed line is not a valid result): leave out the final FPS")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a ROCm device (no CPU fallback)"
    # TPU3_BENCH_BACKEND=gloo + TPU3_BENCH_ONE_DEVICE=1: functional check of the N > 1 code path on
    # a single-GPU box (all ranks share cuda:0, collectives through gloo); never a measurement
    backend = os.environ.get("TPU3_BENCH_BACKEND", "nccl")
    if os.environ.get("TPU3_BENCH_ONE_DEVICE"):
        local_rank = 0
    # (r6) first contact with an N-GPU node: say plainly what is wrong instead of failing inside set_device / RCCL
    ndev = torch.cuda.device_count()
    if ndev < local_rank + 1:
        raise SystemExit("bench.py rank %d: LOCAL_RANK=%d but this process sees %d ROCm device(s) "
                         "(HIP_VISIBLE_DEVICES=%r, ROCR_VISIBLE_DEVICES=%r): launch one rank per visible GPU "
                         "(--nproc-per-node <= %d), or set TPU3_BENCH_ONE_DEVICE=1 for a functional run on one device"
                         % (rank, local_rank, ndev, os.environ.get("HIP_VISIBLE_DEVICES"),
                            os.environ.get("ROCR_VISIBLE_DEVICES"), max(ndev, 1)))
    if world > 1 and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0" and rank == 0:
        print("bench.py: HSA_ENABLE_IPC_MODE_LEGACY is %r, not '0' -- RCCL's intra-node transport needs dmabuf IPC on "
              "this driver (hipIpcGetMemHandle: invalid argument otherwise)" % os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
              file=sys.stderr)
    # the device is bound BEFORE any allocation or communicator exists (RCCL binds its communicator to the current
    # device; device_id makes the binding explicit and lets init create the communicator eagerly)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # TPU3_BENCH_FORCE_DIST=1: initialise the process group and take every N > 1 code path (sharding, the all-gather on
    # the side stream, the `comm` block, the barrier in fence()) at WHATEVER world size, 1 included -- so that RCCL has
    # executed on a one-GPU box before the driver's 8-GPU node is the first to try (tests/test_rccl_world1.py)
    force_dist = os.environ.get("TPU3_BENCH_FORCE_DIST", "0") not in ("0", "")
    multi = world > 1 or force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if force_dist:
            os.environ["TPU3_FORCE_COLLECTIVES"] = "1"      # read by pipeline.py at import (below)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    ops, pipe, ups = pkg("network.operations"), pkg("pipeline"), pkg("network.upsampler")
    assert ops.BACKEND.name == "hip-gfx950"
    patch_mode = multi and args.shard == "patches"
    N, npnt, r = args.num_shape_point, args.num_point, args.up_ratio
    C = 1 if patch_mode else args.clouds
    torch.manual_seed(0)
    net = ups.Net(max_up_ratio=r, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev).eval()
    if patch_mode:
        clouds = poisson_sphere(0, N, dev, ops)                  # every rank holds the same single cloud
    else:
        clouds = torch.cat([poisson_sphere(rank * C + i, N, dev, ops) for i in range(C)], dim=0)

    timing = []
    # HIP events placed by the library immediately around the final-FPS kernel (fl_main_kernel) on ITS stream (the events
    # in `timing` bracket the whole final-FPS operator: Morton sort, bucket setup, kernel, write-back)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    tlib = pkg("_lib").lib()
    kernel_events = []

    def arm_kernel_events():
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipEventCreate(ctypes.byref(e0)) == 0 and hip.hipEventCreate(ctypes.byref(e1)) == 0
        tlib.tpu3_debug_fps_bucket_events(e0, e1)
        kernel_events.append((e0, e1))
    # side streams, used round-robin: the final FPS launches of consecutive steps occupy different
    # CUs (one per cloud) and overlap each other as well as the following steps' network stages
    sides = None if args.no_overlap else [torch.cuda.Stream(device=dev) for _ in range(args.fps_streams)]
    nets = [torch.cuda.Stream(device=dev) for _ in range(args.net_streams)] if args.net_streams > 1 else None
    counter = [0]

    split = sides is not None and nets is not None and args.fps_per_sub_batch and not patch_mode
    n_sub = -(-C // max(1, min(args.sub_batch, -(-C // max(1, args.net_streams))))) if split else 1

    def step():
        side = None if sides is None else sides[counter[0] % len(sides)]
        off = counter[0] * n_sub
        counter[0] += 1
        # the final FPS of this step (one CU per cloud, a pure latency chain) runs on a side stream
        # and overlaps with the network stages of the NEXT step; everything is inside the timed region
        if not args.diag_skip_final_fps:
            arm_kernel_events()
        if args.diag_skip_final_fps:
            return pipe.upsample(net, clouds, npnt, r, 3, final_fps=False, net_streams=nets, sub_batch=args.sub_batch,
                                 check_small=False, optimistic_graph=True)[:, :, :N * r].contiguous()
        if patch_mode:
            # outer patches split across ranks, ONE all-gather of the upsampled patches, final FPS replicated
            return pipe.upsample(net, clouds, npnt, r, 3, shard="patches", timing=timing, check_small=False,
                                 optimistic_graph=True)
        # optimistic_graph=True with check_small=False: this script owns the check (asserted after the timed region)
        out = pipe.upsample(net, clouds, npnt, r, 3, timing=timing, fps_stream=sides if split else side,
                            net_streams=nets, sub_batch=args.sub_batch, fps_offset=off, check_small=False,
                            optimistic_graph=True)                                                            # (C,3,N*r)
        # (cloud-sharded ranks: every rank upsamples ITS clouds with the unsharded call above, then ONE all-gather)
        if split:
            # the step's launches ran on sides[off .. off + n_sub): join them on the first of them
            side = sides[off % len(sides)]
            for i in range(1, n_sub):
                side.wait_stream(sides[(off + i) % len(sides)])
        if multi:                                                           # reassemble: ONE all-gather
            if side is not None:
                with torch.cuda.stream(side):
                    out = pipe._all_gather_cat(out)
            else:
                out = pipe._all_gather_cat(out)
        return out

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    fence()
    del timing[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_clouds = 1 if patch_mode else world * C
    assert tuple(out.shape) == (total_clouds, 3, N * r) and bool(torch.isfinite(out).all())
    assert int(net.small_cloud_events) == 0
    assert ops.BACKEND.graph_dup_events() == 0, "an optimistic kNN graph asked for the exact path: result not final"
    assert not ops.GENERIC_PATH_EVENTS, "generic (unfused) path taken in the measured run: %r" % dict(ops.GENERIC_PATH_EVENTS)
    assert ops.BACKEND.fps_cluster_faults() == 0, "a multi-workgroup FPS launch gave up: result not final"

    # all-gather bus bandwidth (N > 1): (P-1)/P * gathered bytes / time, 10 back-to-back gathers
    comm = None
    if multi:
        if patch_mode:
            P_ = pipe.num_outer_patches(N, npnt, 3)
            part = torch.empty((-(-P_ // world), npnt * r, 3), device=dev)
        else:
            part = torch.empty((C, 3, N * r), device=dev)
        for _ in range(2):
            pipe._all_gather_cat(part)
        fence()
        t1 = time.perf_counter()
        for _ in range(10):
            pipe._all_gather_cat(part)
        fence()
        dt = (time.perf_counter() - t1) / 10
        total_bytes = part.numel() * 4 * world
        # every rank's device as the runtime names it (uuid, PCI bus id, name): two ranks on ONE device -- a mis-set
        # HIP_VISIBLE_DEVICES -- would show here, not as a mysteriously halved number
        props = torch.cuda.get_device_properties(dev)
        mine = "rank %d: cuda:%d %s uuid=%s pci=%s" % (rank, local_rank, props.name, getattr(props, "uuid", "?"),
                                                      getattr(props, "pci_bus_id", "?"))
        names = [None] * dist.get_world_size()
        dist.all_gather_object(names, mine)
        comm = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                "allgather_bytes_total": total_bytes, "allgather_ms": dt * 1e3,
                "allgather_bus_GBps": (world - 1) / world * total_bytes / dt / 1e9,
                "forced_at_world_size_1": bool(force_dist and world == 1),
                "rank_devices": names,
                "distinct_devices": len({n.split("uuid=")[-1] for n in names}) if not os.environ.get("TPU3_BENCH_ONE_DEVICE") else 1,
                "expectation": ("weak scaling over clouds: N x the 1-GPU value at the same --clouds (one %.1f MB all-gather "
                                "per step, ~0.4 ms at 150 GB/s per xGMI link against a >200 ms step); DESIGN section 6"
                                % (total_bytes / 1e6)) if not patch_mode else
                               ("strong scaling over the 48 outer patches of ONE cloud: ~(network stages / N + replicated "
                                "final FPS) per cloud, < 2x at N = 8; DESIGN section 6")}
        if world > 1 and not os.environ.get("TPU3_BENCH_ONE_DEVICE"):
            assert comm["distinct_devices"] == world, "ranks share a device: %r" % (names,)

    op_ms = float(np.mean([a.elapsed_time(b) for a, b in timing])) if timing else None
    fps_ms = None
    if kernel_events:
        vals = []
        for e0, e1 in kernel_events[args.warmup:]:
            ms = ctypes.c_float()
            if hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1) == 0:
                vals.append(ms.value)
        fps_ms = float(np.mean(vals)) if vals else None

    if rank == 0:
        traffic = None
        for path in PROFILE_TRAFFIC:
            try:
                with open(path) as f:
                    traffic = json.load(f)
                traffic["_file"] = os.path.relpath(path, ROOT)
                break
            except (OSError, ValueError):
                pass
        P = pipe.num_outer_patches(N, npnt, 3)
        n_merged = P * npnt * r
        m_out = N * r
        CL = min(args.sub_batch, -(-C // max(1, args.net_streams))) if split else C      # clouds per launch
        # --- the final FPS (main.py:379-380): a LATENCY entry.  It runs on a side stream under the next step's network
        # stages and bounds nothing in this configuration; what describes it is the dependent chain per round.  Its HBM
        # figure (PMC traffic / launch time) and SURVEY 8d's streaming model (20 B per point per round, of which the
        # kernel skips > 99 %) are reported next to it, neither as a roofline fraction of the step.
        tr = None
        if traffic and traffic.get("clouds_per_launch") == CL and (N, npnt, r) == (5000, 312, 16):
            tr = traffic.get("traffic_bytes_per_launch")
        model_bytes = 20.0 * CL * n_merged * (m_out - 1)
        kind = ctypes.c_int(0)
        plan = tlib.tpu3_debug_fps_plan(CL, n_merged, m_out, ctypes.byref(kind))
        fps_entry = {
            "kernel": ("fc_main_kernel (tile-form FPS on %d workgroups per cloud)" % kind.value if plan == 6 else
                       "fl_main_kernel (tile-form FPS, one workgroup per cloud)")
            + ": final FPS %d->%d, %d cloud(s) per launch, on a side stream under the next step" % (n_merged, m_out, CL),
            "bound": "latency", "launch_ms": fps_ms, "operator_ms": op_ms,
            "us_per_sample": (fps_ms * 1e3 / (m_out - 1)) if fps_ms else None,
            "hbm": {"traffic_bytes_per_launch": tr, "traffic_file": (traffic or {}).get("_file"),
                    "traffic_provenance": (traffic or {}).get("provenance"),
                    "achieved_GBps": (tr / (fps_ms * 1e-3) / 1e9) if (tr and fps_ms) else None,
                    "frac_of_8TBps": (tr / (fps_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (tr and fps_ms) else None,
                    "compulsory_bytes_per_launch": float(CL) * (12.0 * n_merged + 4.0 * m_out),
                    "survey_streaming_model_bytes_per_launch": model_bytes},
            "floor_model": "a round (~39 exact samples of a cloud) is ONE dependent chain: tile prune -> bucket records of "
                           "the reached tiles (L2 trip) -> the reached buckets' points (L2 trip) -> candidate list -> the "
                           "ranked candidates' coordinates (L2 trip) -> clearance, with five workgroup barriers: three "
                           "trips of ~0.5 us + ~1 us of instruction issue = 2.5 us per round (a model of THIS algorithm on "
                           "one compute unit, not a hardware bound)",
            "us_per_round_floor": 2.5}
        total_points = total_clouds * N * r * args.steps
        line = {
            "metric": ("INVALID-DIAGNOSTIC " if args.diag_skip_final_fps else "")
            + "upsampled points/sec (16x, 312-pt patches, 5000->80000)",
            "value": total_points / elapsed, "unit": "points/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if patch_mode else "weak", "vs_baseline": None,
            # (r6, the default since the end-to-end gain passed 10 ms: the regressor's two matrix kernels take their fp32
            # operands as three bf16 terms each on the bf16 matrix pipe, fp32 accumulate; everything else is fp32
            # arithmetic; TPU3_SPLIT_BF16=0 restores fp32 matrix instructions -- profiles/r06_split_bf16_end_to_end.txt)
            "dtype": "f32 (3xbf16 split operands, fp32 accumulate)" if ops.BACKEND.split_bf16() else "f32",
            "data": "synthetic",
            "config": {"workload": "C2: %d cloud(s)/GPU x %d pts, num_point=%d, up_ratio=%d (4 levels), "
                                   "%d outer patches, knn=32, random-init weights, Poisson-sphere input"
                                   % (C, N, npnt, r, P) if not patch_mode else
                                   "C2 cloud, outer patches sharded: 1 cloud x %d pts over %d ranks, num_point=%d, "
                                   "up_ratio=%d, %d outer patches" % (N, world, npnt, r, P),
                       "clouds_per_gpu": C, "final_fps_overlap": sides is not None and not patch_mode,
                       "final_fps_launches_per_step": n_sub,
                       "parallelism": (
                           ("dp%d over outer patches: the %d outer patches of ONE cloud split across the ranks (strong "
                            "scaling), one all-gather of the upsampled patches (%.1f MB) per step, then the final FPS "
                            "REPLICATED on every rank -- the Amdahl term" % (world, P, P * npnt * r * 12 / 1e6))
                           if patch_mode else
                           ("dp%d over clouds: %d whole clouds per rank, per-GPU work fixed as N grows (weak scaling), "
                            "zero communication until ONE all-gather of the finished clouds (%.1f MB of fp32 xyz) per "
                            "step over RCCL / xGMI, issued on the final-FPS side stream"
                            % (world, C, world * C * 3 * N * r * 4 / 1e6))) if multi else "single GPU"},
        }
        if comm is not None:
            line["comm"] = comm
        if args.digest:
            import hashlib
            host = out.detach().cpu().contiguous().numpy()
            line["result_digest"] = [hashlib.sha256(host[i].tobytes()).hexdigest() for i in range(host.shape[0])]
        do_extras = not args.no_extras and world == 1 and not args.diag_skip_final_fps
        others = []
        if not args.no_extras and not args.diag_skip_final_fps and not patch_mode:
            # (rank 0 only, untimed: the other ranks wait at the final barrier)
            others = other_rooflines(ops, pipe, net, clouds, npnt, r, traffic)
            # rounds / samples of one final-FPS set (development counters of the tile form; untimed)
            try:
                merged = pipe.upsample(net, clouds[:CL], npnt, r, 3, final_fps=False, check_small=False,
                                       optimistic_graph=True).transpose(2, 1).contiguous()
                st = torch.zeros(32, dtype=torch.int64, device=dev)
                tlib.tpu3_debug_fps_tile_stats(ctypes.c_void_p(st.data_ptr()))
                ops.fps(merged, m_out)
                torch.cuda.synchronize()
                rounds, samples = int(st[0]), int(st[1])
                del merged
                if rounds and fps_ms:
                    fps_entry.update({"rounds": rounds, "samples_per_round": samples / rounds,
                                      "us_per_round": fps_ms * 1e3 / rounds,
                                      "x_over_floor": fps_ms * 1e3 / rounds / fps_entry["us_per_round_floor"]})
            except Exception as e:                                           # noqa: BLE001 (reported, not hidden)
                fps_entry["rounds"] = "failed: %s" % (str(e).splitlines()[0][:120])
        line["rooflines_other"] = others + [fps_entry]
        if not args.no_extras and not args.diag_skip_final_fps and not patch_mode:
            # (r6) the point-set kernels of the C ABI outside the inference step: nm-distance fwd / bwd, ball query, gather
            try:
                line["rooflines_other"] += point_kernel_rooflines(dev)
            except Exception as e:                                           # noqa: BLE001 (reported, not hidden)
                line["rooflines_other"].append({"kernel": "point kernels (nm-distance, ball query, gather)",
                                                "failed": str(e).splitlines()[0][:160]})
        # --- `roofline`: the kernel that bounds the step = the largest entry by ms_per_step that has a hardware bound
        bounded = [o for o in others if o.get("frac") is not None]
        if bounded:
            top = max(bounded, key=lambda o: o["ms_per_step"])
            import re
            mlaunch = re.search(r"(\d+) launches/step", top["kernel"])
            nl = int(mlaunch.group(1)) if mlaunch else None
            work = top.get("executed_flop_per_step", top.get("model_flop_per_step", top.get("algorithmic_bytes_per_step")))
            roof = {"kernel": top["kernel"], "bound": top["bound"], "achieved": top["achieved"], "peak": top["peak"],
                    "unit": top["unit"], "frac": top["frac"],
                    "traffic": (top["traffic"] / nl) if (top.get("traffic") and nl) else None,
                    "selected": "largest kernel of the step by ms_per_step among rooflines_other (median of 5 untimed "
                                "single-stream steps, HIP events on the launch stream)",
                    "basis": top["basis"], "launches_per_step": nl, "ms_per_step": top["ms_per_step"],
                    "avg_launch_ms": (top["ms_per_step"] / nl) if nl else None,
                    "algorithmic_work_per_launch": (work / nl) if (work and nl) else None,
                    "traffic_bytes_per_step": top.get("traffic"), "traffic_file": (traffic or {}).get("_file"),
                    "traffic_provenance": (traffic or {}).get("provenance")}
            if "useful_frac" in top:
                roof["useful_frac"] = top["useful_frac"]
            if top.get("survey_model_flop_per_step"):
                # SURVEY 8(d)'s ALGORITHMIC model (3168 FLOP per edge, un-hoisted) beside the executed figure: the kernel
                # legitimately executes about a third of it (per-point hoisting), so model_frac may exceed 1
                roof["model_flop_per_step"] = top["survey_model_flop_per_step"]
                roof["model_frac"] = top["survey_model_flop_per_step"] / (top["ms_per_step"] * 1e-3) / 1e12 / top["peak"]
                roof["executed_flop_per_step"] = top.get("executed_flop_per_step")
            line["roofline"] = roof
            # --- the whole step against the chip
            flop = sum(o.get("executed_flop_per_step", 0.0) for o in others)
            byts = [o.get("traffic") for o in others if o.get("traffic")]
            tsum = float(sum(byts)) + (tr or 0.0)
            ms_step = elapsed / args.steps * 1e3
            line["roofline_step"] = {
                "ms_per_step": ms_step, "executed_mfma_flop_per_step": flop,
                "mfma_TFLOPs": flop / (ms_step * 1e-3) / 1e12, "frac_of_fp32_mfma_peak": flop / (ms_step * 1e-3) / 1e12 / FP32_PEAK_TF,
                "counter_bytes_per_step": tsum if byts else None,
                "hbm_GBps": (tsum / (ms_step * 1e-3) / 1e9) if byts else None,
                "frac_of_hbm_peak": (tsum / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if byts else None,
                "kernels_ms_sum_single_stream": float(sum(o["ms_per_step"] for o in others)),
                "note": "executed matrix-core FLOP of the three MFMA kernels and the PMC traffic (FETCH_SIZE x 2 + "
                        "WRITE_SIZE, %s) of every profiled kernel incl. one final-FPS launch, over the TIMED ms_per_step: the "
                        "step is bound by VALU issue and dependent chains (kNN selection, FPS), not by either peak"
                        % ((traffic or {}).get("_file"),)}
        else:
            # (profiling runs with --no_extras: no per-kernel events -- the latency entry stands in, explicitly unbounded)
            line["roofline"] = {"kernel": fps_entry["kernel"], "bound": "latency", "achieved": None, "peak": None,
                                "unit": None, "frac": None, "traffic": tr, "launch_ms": fps_ms,
                                "note": "no per-kernel events in this run (--no_extras / sharded patches)"}
        if do_extras:
            line["extras"] = extras_block(args, ops, pipe, ups, net, clouds, dev, N, npnt, r, nets=nets, sides=sides)
            # first-class next to `value` (which is the --clouds batch): BASELINE's C2 read literally -- ONE cloud at a
            # time, as the reference's test() loop does -- and C4's per-rank share of 8 clouds per GPU per step
            lat = line["extras"].get("latency_ms_1cloud")
            if isinstance(lat, float):
                line["value_1cloud"] = N * r / (lat * 1e-3)
                line["ms_per_cloud_1cloud"] = lat
            ms8 = line["extras"].get("ms_per_step_8clouds")
            if isinstance(ms8, float) and clouds.shape[0] >= 8:
                line["value_8clouds"] = 8 * N * r / (ms8 * 1e-3)
            try:
                line["rooflines_train"] = train_rooflines(ops, ups, dev)
            except Exception as e:                                           # noqa: BLE001 (reported, not hidden)
                line["rooflines_train"] = "failed: %s" % (str(e).splitlines()[0][:160])
            assert ops.BACKEND.graph_dup_events() == 0 and int(net.small_cloud_events) == 0, \
                "an optimistic kNN graph / small-cloud event in the untimed extras: their numbers are not final"
            if ops.BACKEND.split_bf16() and not multi and not patch_mode:
                # the SAME step with the regressor's two matrix kernels on the fp32 matrix instructions (TPU3_SPLIT_BF16=0),
                # next to `value`: what the split-bf16 default buys, measured in this process on this box
                was = ops.BACKEND.split_bf16(False)
                try:
                    for _ in range(2):
                        step()
                    fence()
                    k_alt = max(1, min(args.steps, 10))
                    t_alt = time.perf_counter()
                    for _ in range(k_alt):
                        step()
                    fence()
                    ms_alt = (time.perf_counter() - t_alt) / k_alt * 1e3
                finally:
                    ops.BACKEND.split_bf16(was)
                line["value_fp32_matrix_instructions"] = C * N * r / (ms_alt * 1e-3)
                line["ms_per_step_fp32_matrix_instructions"] = ms_alt
        if not args.no_cpu_baseline and world == 1:        # rank 0 at N = 1 only (bench contract)
            from oracle import cpu_baseline
            base, cpu_out = cpu_baseline.measure_c1()
            line["cpu_baseline"] = base
            line["parity"] = parity_block(ops, pipe, ups, dev, cpu_out)
            try:
                # the metric's own configuration against the reference-driven fixture (tests/golden/c2_x16.npz: data)
                line["parity_c2"] = parity_c2_block(dev)
            except Exception as e:                                           # noqa: BLE001 (reported, not hidden)
                line["parity_c2"] = "failed: %s" % (str(e).splitlines()[0][:160])
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
