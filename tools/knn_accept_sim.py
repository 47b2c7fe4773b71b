"""What an accept/reject selection would cost the feature-space kNN graph kernel (k = 33 of 312, 24-d), on REAL features.

knn_graph_key_kernel keeps a lane per query and runs every chunk of 32 candidates through a fixed sorting network:
~20 two-source VALU instructions per (query, candidate) pair, the same for every lane.  The alternative the judge asked
to be built: an unsorted 33-slot buffer per lane with its running maximum tau; a candidate costs
one compare to reject, and an accepted one a replace-max + re-scan (~36 instructions).  A wave executes the accept path
whenever ANY of its 64 lanes accepts, so what matters is not the mean acceptance rate but, per candidate, whether any
lane of the wave accepts -- and, for a chunked variant that first filters 32 candidates against tau and then inserts the
survivors one by one, the MAXIMUM number of survivors over the 64 lanes.

This script measures both on the features the network's DenseEdgeConv blocks actually see (CPU, oracle backend): the
input rows of every feature graph of a 16x run on one outer patch of the C2 cloud, queries dealt to waves as the kernel
deals them (64 consecutive points), candidates in patch order.

usage: python tools/knn_accept_sim.py            (CPU only; ~1 min)
       python tools/knn_accept_sim.py --slab     (r5) the SLAB form: rows ordered along one direction, a wave's 64 queries
                                                 consecutive in that order, chunks visited outwards -- the share of (wave, chunk)
                                                 pairs closed by the projected-gap bound before any distance, and skipped after
                                                 the distances because no lane accepts (csrc/knn.hip, knn_graph_slab_kernel)"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.backend import OracleBackend          # noqa: E402  (measurement infrastructure, like the oracle itself)

ops = importlib.import_module("3pu_pytorch_amd.network.operations")
ups = importlib.import_module("3pu_pytorch_amd.network.upsampler")
layers = importlib.import_module("3pu_pytorch_amd.network.layers")

K = 33
COST_NET = 20.0            # sorting-network kernel: instructions per pair (csrc/knn.hip, measured 22 incl. overheads)
COST_REJECT = 2.0          # compare + mask bookkeeping per candidate
COST_ACCEPT = 36.0         # replace the maximum, re-scan 33 slots for the new one (33 v_max + index bookkeeping)


def graphs_inputs():
    g = np.load(os.path.join(ROOT, "tests", "golden", "c2_x16.npz"))
    state = np.load(os.path.join(ROOT, "tests", "golden", "net16_state.npz"))
    net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5)
    net.load_state_dict({k: torch.from_numpy(state[k]) for k in state.files if k != "meta"})
    net.eval()
    ops.BACKEND = OracleBackend()
    rows = []
    real = layers.DenseEdgeConv.get_local_graph_cl

    def spy(self, x, k, idx=None, layout=None):
        rows.append(x.detach().numpy().copy())
        return real(self, x, k, idx=idx, layout=layout)
    layers.DenseEdgeConv.get_local_graph_cl = spy
    try:
        cloud = torch.from_numpy(g["cloud"])
        pipe = importlib.import_module("3pu_pytorch_amd.pipeline")
        _, patches, _ = pipe.extract_outer_patches(cloud, 312, 3)
        with torch.no_grad():
            pipe.upsample_patches(net, patches.reshape(-1, 312, 3)[:1], 16)
    finally:
        layers.DenseEdgeConv.get_local_graph_cl = real
    return rows


def simulate(x):
    """x (P, 312, 24) -> per (patch, wave): instruction counts of the three schemes per pair."""
    P, n, _ = x.shape
    tot_pairs = 0
    c_accept_any = 0.0      # per-candidate accept/reject, wave executes accept if any lane accepts
    c_chunk_max = 0.0       # chunked: 32 compares per lane, then max-over-lanes survivors x insert
    acc_mean = []
    for p in range(min(P, 12)):
        d = ((x[p][:, None, :] - x[p][None, :, :]) ** 2).sum(-1)          # (n queries, n candidates)
        for w0 in range(0, n, 64):
            q = d[w0:w0 + 64]                                              # the wave's queries
            # running K-th smallest after each candidate (candidates in patch order)
            buf = np.full((q.shape[0], K), np.inf)
            accepted = np.zeros(q.shape, bool)
            for j in range(n):
                tau = buf.max(axis=1)
                a = q[:, j] < tau
                accepted[:, j] = a
                arg = buf.argmax(axis=1)
                rowsel = np.where(a)[0]
                buf[rowsel, arg[rowsel]] = q[rowsel, j]
            lanes = q.shape[0]
            tot_pairs += n * 64                       # (a partial last wave still occupies 64 lanes)
            any_acc = accepted.any(axis=0)
            c_accept_any += (COST_REJECT * n + COST_ACCEPT * any_acc.sum()) * 64
            for c0 in range(0, n, 32):
                surv = accepted[:, c0:c0 + 32].sum(axis=1)
                c_chunk_max += (COST_REJECT * 32 + COST_ACCEPT * surv.max()) * 64
            acc_mean.append(accepted.mean())
    return tot_pairs, c_accept_any, c_chunk_max, float(np.mean(acc_mean))


def slab_axis(x, kind):
    """t = projection of the rows on a unit direction chosen like the kernel's pre-pass (float64 here)."""
    n = len(x)
    if kind == "pca1":
        xc = x - x.mean(0)
        return xc @ np.linalg.svd(xc, full_matrices=False)[2][0]
    if kind == "ones-2":
        xc = x - x.mean(0)
        v = np.ones(x.shape[1])
        for _ in range(2):
            v = xc.T @ (xc @ v)
            v /= np.linalg.norm(v)
        return xc @ v
    v0 = (x[((x - x[0]) ** 2).sum(1).argmax()] if kind == "far-1" else x[n - 1]) - x[0]
    S = x.sum(0)
    v = x.T @ (x @ v0) - (S @ v0) / n * S
    return x @ (v / np.linalg.norm(v))


def slab_sim(x, perm, t=None):
    """(chunks in all, closed by the bound, skipped after the distances) for one patch; t None: no bound (an order
    that is not a projection of the current rows)."""
    n = x.shape[0]
    xp = x[perm]
    tp = None if t is None else t[perm]
    d = ((xp[:, None, :] - xp[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d, np.inf)
    L, nch = K - 1, (n + 31) // 32
    tot = lb = post = 0
    for wi, w0 in enumerate(range(0, n, 64)):
        q = d[w0:w0 + 64]
        lst = np.full((q.shape[0], L), np.inf)
        lo, hi, side, step = 2 * wi - 1, 2 * wi + 2, 0, 0
        while True:
            if step < 2:
                c, own = 2 * wi + step, True
                if c >= nch:
                    step += 1
                    continue
            else:
                own = False
                lopen, hopen = lo >= 0, hi < nch
                if not lopen and not hopen:
                    break
                left = lopen and (not hopen or side == 0)
                side ^= 1
                c = lo if left else hi
                if tp is not None:
                    tc = tp[c * 32:c * 32 + 32]
                    gap = np.maximum(tc.min() - tp[w0:w0 + 64], tp[w0:w0 + 64] - tc.max())
                    if ((gap > 0) & (gap * gap > lst.max(axis=1))).all():
                        closed = lo + 1 if left else nch - hi
                        tot += closed
                        lb += closed
                        if left:
                            lo = -1
                        else:
                            hi = nch
                        continue
                if left:
                    lo -= 1
                else:
                    hi += 1
            step += 1
            tot += 1
            nw = q[:, c * 32:c * 32 + 32]
            if not own and not (nw.min(axis=1) < lst.max(axis=1)).any():
                post += 1
                continue
            lst = np.partition(np.concatenate([lst, nw], axis=1), L - 1, axis=1)[:, :L]
    return tot, lb, post


def main_slab():
    feats = graphs_inputs()
    print("feature graphs seen: %d calls, shapes %s" % (len(feats), sorted({r.shape for r in feats})))
    res = {}
    for gi, x in enumerate(feats):
        for p in range(min(x.shape[0], 6)):
            xx = x[p].astype(np.float64)
            runs = {"patch order, chunks in sequence (shipped one-pass kernel)": (np.arange(len(xx)), None)}
            for kind, label in (("pca1", "first principal axis (exact)"), ("ones-2", "two power iterations from ones"),
                                ("far-1", "x_far - x_0, one power iteration (built)"),
                                ("last-1", "x_last - x_0, one power iteration")):
                t = slab_axis(xx, kind)
                runs[label] = (np.argsort(t, kind="stable"), t)
            for label, (perm, t) in runs.items():
                r = slab_sim(xx, perm, t)
                acc = res.setdefault(label, [0, 0, 0])
                for i in range(3):
                    acc[i] += r[i]
    print("share of (wave, chunk) pairs, over the feature rows of %d graphs (up to 6 patches each):" % len(feats))
    print("%-62s %-22s %s" % ("order of the patch's rows", "closed by the bound", "skipped after the distances"))
    for label, (tot, lb, post) in res.items():
        print("%-62s %-22.3f %.3f" % (label, lb / tot, post / tot))


def main():
    if "--slab" in sys.argv:
        return main_slab()
    rows = graphs_inputs()
    print("feature graphs seen: %d calls, shapes %s" % (len(rows), sorted({r.shape for r in rows})))
    tp = ca = cc = 0.0
    am = []
    for r in rows:
        t, a, c, m = simulate(r)
        tp += t; ca += a; cc += c; am.append(m)
    print("mean acceptance rate per (query, candidate): %.3f  (33 H(312/33) / 312 = %.3f expected for random order)"
          % (np.mean(am), (33 + 33 * np.log(312 / 33)) / 312))
    print("instructions per pair, wave level:")
    print("  sorting network (every lane, every chunk)           %.1f" % COST_NET)
    print("  accept / reject per candidate (any lane accepts)    %.1f" % (ca / tp))
    print("  filter 32, insert max-over-lanes survivors          %.1f" % (cc / tp))


if __name__ == "__main__":
    main()
