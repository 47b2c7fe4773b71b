import numpy as np, sys, time
g = np.load("/root/repo/tests/golden/c2_x16.npz")
P = np.ascontiguousarray(g["pred_concat"][0].T).astype(np.float32)   # (239616,3)
n = P.shape[0]
# morton order
lo, hi = P.min(0), P.max(0)
q = np.clip(((P - lo) / (hi - lo) * 1023).astype(np.int64), 0, 1023)
def spread(v):
    v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v
code = spread(q[:,0]) | (spread(q[:,1]) << 1) | (spread(q[:,2]) << 2)
order = np.argsort(code, kind="stable")
X = P[order]
def sim(cell, m, cap, label):
    nc = (n + cell - 1) // cell
    pad = nc * cell - n
    dist = np.full(n, 1e10, np.float32)
    first = int(np.where(order == 0)[0][0])
    cur = [first]
    r = 1; rounds = 0; hist = []
    ncand_hist = []
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
        cand = np.where(M > Rs)[0]
        ncand_hist.append(len(cand))
        if len(cand) == 0:
            c = int(M.argmax()); cur = [c * cell + int(am[c])]
        else:
            cand = cand[np.argsort(-M[cand], kind="stable")][:cap]
            pts = cand * cell + am[cand]
            Mj = M[cand]
            J = len(pts)
            xs = X[pts]
            for j in range(1, len(pts)):
                dj = ((xs[:j] - xs[j]) ** 2).sum(1)
                if (dj < Mj[j]).any():
                    J = j; break
            J = min(J, m - r)
            cur = list(pts[:J])
        r += len(cur); rounds += 1; hist.append(len(cur))
    h = np.array(hist); c = np.array(ncand_hist)
    print("%s cell=%d cap=%d: m=%d rounds=%d samples/round=%.2f (last third %.2f) cand/round=%.1f" %
          (label, cell, cap, m, rounds, h.mean(), h[len(h)*2//3:].mean(), c.mean()), flush=True)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
for cell, cap in ((1024, 32), (64, 32), (64, 64), (16, 64), (16, 128)):
    t0 = time.time(); sim(cell, m, cap, "merged239616"); print("  %.0f s" % (time.time() - t0), flush=True)
