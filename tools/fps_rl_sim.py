import numpy as np, sys
g = np.load("/root/repo/tests/golden/c2_x16.npz")
P = np.ascontiguousarray(g["pred_concat"][0].T).astype(np.float32)
# a level-4-like merged set: 5 consecutive patches of the final merged cloud (5 x 4992 = 24960 points)
P = P[:24960]
n = P.shape[0]
lo, hi = P.min(0), P.max(0)
q = np.clip(((P - lo) / (hi - lo) * 1023).astype(np.int64), 0, 1023)
def spread(v):
    v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v
code = spread(q[:,0]) | (spread(q[:,1]) << 1) | (spread(q[:,2]) << 2)
order = np.argsort(code, kind="stable")
X = P[order]
def sim(cell, m, wcap, nw, stale, cap):
    nc = (n + cell - 1) // cell
    pad = nc * cell - n
    dist = np.full(n, 1e10, np.float32)
    cur = [int(np.where(order == 0)[0][0])]
    r = 1; hist = []
    rstar_prev = np.float32(3e38)
    wave_of = np.arange(nc) // (nc // nw)
    while r < m:
        for s in cur:
            d = ((X - X[s]) ** 2).sum(1).astype(np.float32)
            np.minimum(dist, d, out=dist)
        dd = np.concatenate([dist, np.full(pad, -1, np.float32)]).reshape(nc, cell)
        am = dd.argmax(1); M = dd[np.arange(nc), am]
        dd2 = dd.copy(); dd2[np.arange(nc), am] = -2
        Rs = dd2.max(1).max()
        thr = rstar_prev if stale else Rs
        rstar_prev = Rs
        cand = np.where(M > thr)[0]
        drop = -1.0; keep = []
        for w in range(nw):
            cw = cand[wave_of[cand] == w]
            cw = cw[np.argsort(-M[cw], kind="stable")]
            keep.extend(cw[:wcap])
            if len(cw) > wcap: drop = max(drop, M[cw[wcap]])
        keep = np.array(keep, dtype=np.int64)
        if len(keep) < 1:
            c = int(M.argmax()); cur = [c * cell + int(am[c])]
        else:
            keep = keep[np.argsort(-M[keep], kind="stable")][:cap]
            k2 = keep[M[keep] > drop]
            if len(k2) == 0: k2 = keep[:1]
            pts = k2 * cell + am[k2]; Mj = M[k2]; J = len(pts); xs = X[pts]
            for j in range(1, len(pts)):
                if (((xs[:j] - xs[j]) ** 2).sum(1) < Mj[j]).any():
                    J = j; break
            cur = list(pts[:min(J, m - r)])
        r += len(cur); hist.append(len(cur))
    h = np.array(hist)
    print("cell=%d wcap=%d stale=%d cap=%d: rounds %d samples/round=%.2f (last third %.2f)" % (cell, wcap, stale, cap, len(h), h.mean(), h[len(h)*2//3:].mean()), flush=True)
for cell, wcap, stale, cap in ((25, 4, 1, 64), (25, 64, 0, 64), (25, 64, 0, 32), (25, 64, 1, 64)):
    sim(cell, 4992, wcap, 16, stale, cap)
