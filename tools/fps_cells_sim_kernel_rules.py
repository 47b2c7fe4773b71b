import numpy as np, sys, time
g = np.load("/root/repo/tests/golden/c2_x16.npz")
P = np.ascontiguousarray(g["pred_concat"][0].T).astype(np.float32)
n = P.shape[0]
lo, hi = P.min(0), P.max(0)
q = np.clip(((P - lo) / (hi - lo) * 1023).astype(np.int64), 0, 1023)
def spread(v):
    v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v
code = spread(q[:,0]) | (spread(q[:,1]) << 1) | (spread(q[:,2]) << 2)
order = np.argsort(code, kind="stable")
X = P[order]
def sim(cell, m, wcap, nw, stale, label, start=0):
    nc = (n + cell - 1) // cell
    pad = nc * cell - n
    dist = np.full(n, 1e10, np.float32)
    cur = [int(np.where(order == 0)[0][0])]
    r = 1; rounds = 0; hist = []
    rstar_prev = np.float32(3e38)
    wave_of = np.arange(nc) % nw
    while r < m:
        for s in cur:
            d = ((X - X[s]) ** 2).sum(1).astype(np.float32)
            np.minimum(dist, d, out=dist)
        dd = np.concatenate([dist, np.full(pad, -1, np.float32)]).reshape(nc, cell)
        am = dd.argmax(1)
        M = dd[np.arange(nc), am]
        dd2 = dd.copy(); dd2[np.arange(nc), am] = -2
        R = dd2.max(1)
        Rs = R.max()
        thr = rstar_prev if stale else Rs
        rstar_prev = Rs
        cand = np.where(M > thr)[0]
        drop = -1.0
        keep = []
        for w in range(nw):
            cw = cand[wave_of[cand] == w]
            cw = cw[np.argsort(-M[cw], kind="stable")]
            keep.extend(cw[:wcap])
            if len(cw) > wcap: drop = max(drop, M[cw[wcap]])
        keep = np.array(keep, dtype=np.int64)
        if len(keep) < 2:
            c = int(M.argmax()); cur = [c * cell + int(am[c])]
        else:
            keep = keep[np.argsort(-M[keep], kind="stable")]
            keep = keep[M[keep] > drop]
            if len(keep) == 0:
                c = int(M.argmax()); keep = np.array([c])
            pts = keep * cell + am[keep]
            Mj = M[keep]; J = len(pts); xs = X[pts]
            for j in range(1, len(pts)):
                if (((xs[:j] - xs[j]) ** 2).sum(1) < Mj[j]).any():
                    J = j; break
            cur = list(pts[:min(J, m - r)])
        r += len(cur); rounds += 1
        if r > start: hist.append(len(cur))
    h = np.array(hist)
    print("%s cell=%d wcap=%d nw=%d stale=%d: samples/round=%.2f (last third %.2f)" % (label, cell, wcap, nw, stale, h.mean(), h[len(h)*2//3:].mean()), flush=True)
m = int(sys.argv[1])
for cell, wcap, nw, stale in ((1024, 4, 8, 1), (1024, 8, 8, 1), (1024, 8, 8, 0), (1024, 64, 8, 0), (64, 8, 8, 1), (64, 16, 8, 1)):
    sim(cell, m, wcap, nw, stale, "merged")
