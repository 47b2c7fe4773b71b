"""Does the slab form's side-closing bound hold when the pre-pass order is only BINNED?  (r6, advisor finding on r5)

knn_slab_order_kernel sorts a patch's rows by t = <x, v> with a binned counting sort: inside a bin the order is the
atomics' arrival order.  knn_graph_slab_kernel closes a side of a wave's outward walk at chunk c when every lane's
projected gap to chunk c's t-range beats the lane's 32nd key -- and with the r5 table (each chunk's OWN range) that
silently assumed the ranges are monotone along the order.  A chunk that lies wholly inside one bin (a dense cluster
narrower than a bin) breaks the assumption: a later chunk of the same bin may hold a row with t closer to the query.

This script restates the order + bound logic in float64 numpy (the key test and the E1 / E2 margins left out, as in
the advisor's re-implementation) and counts rows whose neighbour set comes out wrong, for
    --table own        the r5 table: crange[c] = (min, max) of chunk c
    --table monotone   the r6 table: right side suffix-min of the chunk minima, left side prefix-max of the maxima
on `cluster_on_a_line` patches (tests/test_hip_kernels.py uses the same generator), with the in-bin order = original
row order (what one wave's LDS atomics produce) or random.

usage: python tools/knn_slab_bound_sim.py [--table own|monotone] [--patches 60] [--inbin index|random]"""
import argparse

import numpy as np

K, L, C = 33, 32, 24


def cluster_on_a_line(rng, n=312, c=C, ncl=150, width=0.2, perp=0.6, gap=1.0, adversarial=True):
    """n rows of c channels: n - ncl points on a line with unit spacing along a random direction u (with a little
    perpendicular noise) and ncl points in a cluster of t-extent `width` (narrower than a bin of the pre-pass) and
    perpendicular extent `perp`, sitting `gap` beyond one of the line points.  adversarial: the cluster's rows get the
    LOWEST original indices in the order (lowest third of t, highest third, middle third), so that an in-bin order by
    arrival = original index puts a chunk of far-in-t rows before a chunk of near-in-t rows."""
    u = rng.standard_normal(c)
    u /= np.linalg.norm(u)
    nl = n - ncl
    tl = np.arange(nl, dtype=np.float64)
    line = tl[:, None] * u[None, :] + 0.02 * rng.standard_normal((nl, c))
    at = float(rng.integers(nl // 4, 3 * nl // 4)) + gap * 0.5
    tc = at + width * (rng.random(ncl) - 0.5)
    pn = rng.standard_normal((ncl, c))
    pn -= (pn @ u)[:, None] * u[None, :]
    pn *= perp * rng.random((ncl, 1)) / np.linalg.norm(pn, axis=1, keepdims=True)
    clus = tc[:, None] * u[None, :] + pn
    if adversarial:
        o = np.argsort(tc)
        third = ncl // 3
        o = np.concatenate([o[:third], o[ncl - third:], o[third:ncl - third]])
        clus = clus[o]
    else:
        clus = clus[rng.permutation(ncl)]
    rest = line[rng.permutation(nl)]
    x = np.concatenate([clus, rest], 0)
    return np.ascontiguousarray(x.astype(np.float32))


def slab_order(x, inbin, rng):
    """positions of knn_slab_order_kernel: (t sorted, original row at each position)."""
    x = x.astype(np.float64)
    n = x.shape[0]
    nthreads = (n + 63) // 64 * 64
    d0 = ((x - x[0]) ** 2).sum(1)
    v0 = x[int(np.argmax(d0))] - x[0]
    s = x @ v0
    A, S, T = (s[:, None] * x).sum(0), x.sum(0), s.sum()
    v = A - T / n * S
    nrm2 = float(v @ v)
    sig = np.sqrt(np.sqrt(nrm2) / np.sqrt(float(v0 @ v0)) / n)
    v = v / np.sqrt(nrm2) * 0.9990234375
    t = x @ v
    mean = float(S @ v) / n
    f = np.clip((t - mean) * (0.2 * nthreads / sig) + 0.5 * nthreads, 0, nthreads - 1)
    bins = f.astype(np.int64)
    tie = np.arange(n) if inbin == "index" else rng.permutation(n)
    order = np.lexsort((tie, bins))
    return t[order], order, bins[order]


def slab_graph(x, t, order, table, margins=False):
    """neighbour sets (as sorted original indices) the slab walk produces without the key test; margins: with the
    kernel's E1 / E2 and the 32nd key taken at the top of its truncation bucket (the low 9 mantissa bits hold the
    position), i.e. the bound exactly as knn_graph_slab_kernel evaluates it."""
    xs = x.astype(np.float64)[order]
    n = xs.shape[0]
    M = float((xs ** 2).sum(1).max())
    E1, E2 = (3.0517578125e-05 * np.sqrt(M), 3.0517578125e-05 * M) if margins else (0.0, 0.0)
    nch = (n + L - 1) // L
    D = ((xs[:, None, :] - xs[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(D, np.inf)
    lo = np.array([t[c * L:(c + 1) * L].min() for c in range(nch)])
    hi = np.array([t[c * L:(c + 1) * L].max() for c in range(nch)])
    if table == "monotone":
        lo = np.minimum.accumulate(lo[::-1])[::-1]
        hi = np.maximum.accumulate(hi)
    out = np.empty((n, L), np.int64)
    for w in range((n + 63) // 64):
        q = np.arange(w * 64, min(n, w * 64 + 64))
        seen = np.zeros((len(q), n), bool)

        def visit(c):
            seen[:, c * L:min(n, (c + 1) * L)] = True

        def tau():
            d = np.where(seen, D[q], np.inf)
            return np.sort(d, 1)[:, L - 1]
        for c in (2 * w, 2 * w + 1):
            if c < nch:
                visit(c)
        lo_c, hi_c, side = 2 * w - 1, 2 * w + 2, 0
        while True:
            lopen, hopen = lo_c >= 0, hi_c < nch
            if not lopen and not hopen:
                break
            left = lopen and (not hopen or side == 0)
            side ^= 1
            c = lo_c if left else hi_c
            gap = np.maximum(lo[c] - t[q], t[q] - hi[c]) - E1
            tq = tau()
            if margins:
                tq = (tq.astype(np.float32).view(np.int32) | 0x1FF).view(np.float32).astype(np.float64)
            if np.all((gap > 0) & (gap * gap - E2 > tq)):
                if left:
                    lo_c = -1
                else:
                    hi_c = nch
                continue
            if left:
                lo_c -= 1
            else:
                hi_c += 1
            visit(c)
        d = np.where(seen, D[q], np.inf)
        out[q] = np.sort(order[np.argsort(d, 1, kind="stable")[:, :L]], 1)
    res = np.empty_like(out)
    res[order] = out
    return res


def exact_sets(x):
    x = x.astype(np.float64)
    D = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(D, np.inf)
    return np.sort(np.argsort(D, 1, kind="stable")[:, :L], 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--table", default="own", choices=["own", "monotone"])
    ap.add_argument("--patches", type=int, default=60)
    ap.add_argument("--inbin", default="index", choices=["index", "random"])
    ap.add_argument("--margins", action="store_true")
    ap.add_argument("--width", type=float, default=0.2)
    ap.add_argument("--perp", type=float, default=0.6)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    bad_patches, bad_rows, one_bin = 0, 0, 0
    for p in range(a.patches):
        x = cluster_on_a_line(rng, width=a.width, perp=a.perp)
        t, order, bins = slab_order(x, a.inbin, rng)
        one_bin += int(np.bincount(bins).max() >= 64)
        got = slab_graph(x, t, order, a.table, a.margins)
        wrong = int((got != exact_sets(x)).any(1).sum())
        bad_patches += wrong > 0
        bad_rows += wrong
    print("table=%s in-bin order=%s: %d of %d patches with a wrong neighbour set (%d rows); "
          "%d patches with >= 64 rows in one bin" % (a.table, a.inbin, bad_patches, a.patches, bad_rows, one_bin))


if __name__ == "__main__":
    main()
