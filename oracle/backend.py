"""oracle/backend.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Stand-in for network.operations.HipBackend, served by the CPU oracle.

It lets the `-m "not gpu"` suite run the product's HOST logic (module wiring, batched / ragged
control flow of Net, the pipeline, the sharding code) on CPU tensors and compare it with the
reference-generated fixtures.  The product never imports this; tests install it by assigning
`operations.BACKEND`."""
import numpy as np
import torch

from . import oracle as orc


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


class OracleBackend(object):
    name = "cpu-oracle (tests only)"

    def knn(self, k, query, points, unique, layout=None, want_dist=True, want_grouped=True):
        q, p = _np(query), _np(points)
        b, m, c = q.shape
        bp, n, _ = p.shape
        layout = layout or {}
        n_arr, m_arr = _np(layout.get("n_arr")), _np(layout.get("m_arr"))
        pts_of, grp = _np(layout.get("pts_of")), _np(layout.get("grp"))
        pts_of = np.arange(b) if pts_of is None else pts_of
        grp = np.zeros(b, np.int64) if grp is None else grp
        idx = np.zeros((b, m, k), np.int64)
        dist = np.zeros((b, m, k), np.float32)
        grouped = np.zeros((b, m, k, c), np.float32)
        # one reference call per group: all its query sets against their point sets
        for g in np.unique(grp):
            members = np.where(grp == g)[0]
            spans = []
            for i in members:
                pb = int(pts_of[i])
                nn = n if n_arr is None else int(n_arr[pb])
                mm = m if m_arr is None else int(m_arr[i])
                spans.append((i, pb, nn, mm))
            if len({(nn, mm) for (_, _, nn, mm) in spans}) == 1:
                # uniform sizes: this IS one reference call (batch = the group's query sets, point sets
                # expanded like previous_xyz.expand) -- the C oracle applies max(D) over the whole call
                _, _, nn, mm = spans[0]
                qq = np.ascontiguousarray(q[members][:, :mm])
                pp = np.ascontiguousarray(p[[pb for (_, pb, _, _) in spans]][:, :nn])
                ii, dd = orc.knn(k, qq, pp, unique)
                for t, (i, pb, _, _) in enumerate(spans):
                    idx[i, :mm], dist[i, :mm] = ii[t], dd[t]
                    grouped[i, :mm] = p[pb][ii[t]]
                continue
            if unique:
                # ragged group: emulate `D += max(D) * dup` over the whole group in two passes
                dmax, any_dup = None, False
                dups = {}
                for (i, pb, nn, mm) in spans:
                    if pb not in dups:
                        dups[pb] = orc.first_occurrence_dup(p[pb:pb + 1, :nn])[0]
                        any_dup |= bool(dups[pb].any())
                if any_dup:
                    for (i, pb, nn, mm) in spans:
                        _, d_all = orc.knn(nn, q[i:i + 1, :mm], p[pb:pb + 1, :nn], False)
                        mx = d_all.max()
                        dmax = mx if dmax is None else max(dmax, mx)
                for (i, pb, nn, mm) in spans:
                    if any_dup:
                        ii, dd = orc.knn(nn, q[i:i + 1, :mm], p[pb:pb + 1, :nn], False)
                        d_by_idx = np.empty((mm, nn), np.float32)
                        np.put_along_axis(d_by_idx, ii[0].astype(np.int64), dd[0], axis=1)
                        d_by_idx = d_by_idx + np.float32(dmax) * dups[pb].astype(np.float32)[None, :]
                        order = np.lexsort((np.broadcast_to(np.arange(nn), (mm, nn)), d_by_idx), axis=1)[:, :k]
                        idx[i, :mm] = order
                        dist[i, :mm] = np.take_along_axis(d_by_idx, order, axis=1)
                    else:
                        ii, dd = orc.knn(k, q[i:i + 1, :mm], p[pb:pb + 1, :nn], False)
                        idx[i, :mm], dist[i, :mm] = ii[0], dd[0]
            else:
                for (i, pb, nn, mm) in spans:
                    ii, dd = orc.knn(k, q[i:i + 1, :mm], p[pb:pb + 1, :nn], False)
                    idx[i, :mm], dist[i, :mm] = ii[0], dd[0]
            for (i, pb, nn, mm) in spans:
                grouped[i, :mm] = p[pb][idx[i, :mm]]
        dev = query.device
        return (torch.from_numpy(idx).to(dev), torch.from_numpy(dist).to(dev) if want_dist else None,
                torch.from_numpy(grouped).to(dev) if want_grouped else None)

    def fps(self, xyz, npoint, n_arr=None, m_arr=None):
        x = _np(xyz)
        b, n, _ = x.shape
        out = np.zeros((b, npoint), np.int32)
        for i in range(b):
            nn = n if n_arr is None else int(n_arr[i])
            mm = npoint if m_arr is None else int(m_arr[i])
            out[i, :mm] = orc.fps(x[i:i + 1, :nn], mm)[0][0]
        return torch.from_numpy(out).to(xyz.device)

    def gather_forward(self, features, idx):
        return torch.from_numpy(orc.gather_fwd(_np(features), _np(idx))).to(features.device)

    def gather_backward(self, grad_out, idx, c, n):
        return torch.from_numpy(orc.gather_bwd(_np(grad_out), _np(idx), n)).to(grad_out.device)

    def ball_query(self, query, xyz, radius, nsample):
        return torch.from_numpy(orc.ball_query(_np(query), _np(xyz), radius, nsample)).to(query.device)

    def normalize(self, pc, n_arr=None):
        assert n_arr is None
        o, c, r = orc.normalize_point_batch(_np(pc), NCHW=True)
        return torch.from_numpy(o), torch.from_numpy(c), torch.from_numpy(r)
