"""Test helper: the discrete choices of Net.forward (eval, 16x) for a few outer patches of the C2 cloud, REPLAYED from
the reference's own run or RECORDED from this build's -- tests/golden/c2_chain.npz, oracle/make_golden.py
`make_chain_golden` (reference network/upsampler.py:107-189, 59-86, 272-374, layers.py:33).

The product code is not touched: the choices enter and leave through the three seams every backend already has --
`operations.knn_query` (outlier filter k = 2, inner patches k = 312, inter-level neighbours k = 5, and on the CPU
stand-in the feature graphs k = 33), `operations.fps` (inner seeds, per-level resampling) and, on the device,
`BACKEND.knn_graph` (the feature graphs of the fused DenseEdgeConv).

Named choices, in the order a level takes them:
    l<l>_graph<b>   block b's 33-neighbour rows (the first is dropped, the rest is a SET)      all levels
    l<l>_mask       the outlier filter on the level's input cloud                               levels 2-4
    l<l>_seeds      FPS seeds of the inner patches
    l<l>_pidx       the inner patches' points (a set per patch: its order is the kNN's)
    l<l>_fm         inter-level neighbours (a set per point)
    l<l>_fps        the resampling of the merged level output
(the graphs of a level come after its mask / seeds / pidx and before its fm / fps)."""
import numpy as np
import torch

ORDER = ["l1_graph1", "l1_graph2", "l1_graph3", "l1_graph4"] + [
    "l%d_%s" % (l, n) for l in (2, 3, 4)
    for n in ("mask", "seeds", "pidx", "graph1", "graph2", "graph3", "graph4", "fm", "fps")]
N_IN = {2: 624, 3: 1248, 4: 2496}                  # points a level receives
P_MAX = {1: 1, 2: 10, 3: 20, 4: 40}                # inner patches per outer patch (padded count)


def reference_choices(g, q):
    """{name: array} of outer patch q as recorded from the reference."""
    return {n: g["p%d_%s" % (q, n)] for n in ORDER}


class Chain(object):
    """mode 'replay': every choice is answered from the reference's record (stacked over the outer patches `ids`);
    mode 'record': the real functions run and what they chose is kept in self.seen[name] (per outer patch)."""

    def __init__(self, ops, g, ids, dev, mode):
        self.ops, self.g, self.ids, self.dev, self.mode = ops, g, list(ids), dev, mode
        self.B = len(self.ids)
        self.graph_calls = 0            # 4 per level: the level of a call follows from how many graphs were built
        self.levels_closed = 0
        self.seen = {}
        self.real_knn_query, self.real_fps = ops.knn_query, ops.fps
        self.real_graph = getattr(ops.BACKEND, "knn_graph", None)

    # ---- plumbing ------------------------------------------------------------------------------------------------
    def __enter__(self):
        self.ops.knn_query, self.ops.fps = self.knn_query, self.fps
        if self.real_graph is not None:
            self.ops.BACKEND.knn_graph = self.knn_graph
        return self

    def __exit__(self, *exc):
        self.ops.knn_query, self.ops.fps = self.real_knn_query, self.real_fps
        if self.real_graph is not None:
            del self.ops.BACKEND.knn_graph
        return False

    def rec(self, name, q):
        return self.g["p%d_%s" % (q, name)]

    def live(self, l, q):
        """inner patches the reference cut for outer patch q at level l"""
        return 1 if l == 1 else self.rec("l%d_seeds" % l, q).shape[0]

    def per_patch(self, name, l, dtype):
        """the record of every outer patch padded to P_MAX[l] inner patches (a dead slot repeats the last live one,
        like the product's _repatch), stacked: (B * P, ...)."""
        rows = []
        for q in self.ids:
            a = np.asarray(self.rec(name, q)).astype(dtype)
            if l == 1:
                a = a.reshape((1,) + a.shape[-2:])
            pad = P_MAX[l] - a.shape[0]
            if pad:
                a = np.concatenate([a, np.repeat(a[-1:], pad, axis=0)], axis=0)
            rows.append(a)
        return torch.from_numpy(np.concatenate(rows, axis=0)).to(self.dev)

    def keep(self, name, value, per):
        """value (B * per, ...) tensor -> one array per outer patch"""
        v = value.detach().cpu().numpy()
        self.seen[name] = [v[i * per:(i + 1) * per] for i in range(self.B)]

    # ---- the three seams -----------------------------------------------------------------------------------------
    def knn_graph(self, k, x, layout=None, optimistic=None):
        assert k == 33
        l, b = 1 + self.graph_calls // 4, self.graph_calls % 4 + 1
        self.graph_calls += 1
        name = "l%d_graph%d" % (l, b)
        if self.mode == "replay":
            out = self.per_patch(name, l, np.int32)
            assert out.shape[0] == x.shape[0], (name, out.shape, x.shape)
            return out
        out = self.real_graph(k, x, layout)
        self.keep(name, out, P_MAX[l])
        return out

    def knn_query(self, k, query, points, unique=True, layout=None, want_dist=True, want_grouped=True, unique_cache=None):
        kw = {} if unique_cache is None else {"unique_cache": unique_cache}
        # (filter, seeds and inner patches of level l come BEFORE its four graphs, the inter-level search after)
        l = self.graph_calls // 4 if (k == 5 and unique) else 1 + self.graph_calls // 4
        if k == 33 and unique:                                   # the CPU stand-in's feature graphs
            b = self.graph_calls % 4 + 1
            self.graph_calls += 1
            name = "l%d_graph%d" % (l, b)
            if self.mode == "replay":
                idx = self.per_patch(name, l, np.int64)
                bb = torch.arange(points.size(0), device=points.device).view(-1, 1, 1)
                return idx, None, (points[bb, idx] if want_grouped else None)
            out = self.real_knn_query(k, query, points, unique, layout, want_dist, want_grouped, **kw)
            self.keep(name, out[0], P_MAX[l])
            return out
        if k == 2 and not unique:                                # the outlier filter
            name = "l%d_mask" % l
            if self.mode == "replay":
                # distances that make `d < 5 * mean(d)` the reference's mask: 1 for a kept point, 1e6 for a dropped one
                # (at most a fifth of the points can be dropped, so 5 * mean stays below 1e6 and above 1)
                m = torch.from_numpy(np.stack([np.asarray(self.rec(name, q)) for q in self.ids])).to(self.dev)
                d = torch.where(m, torch.ones((), device=self.dev), torch.full((), 1e6, device=self.dev))
                return None, torch.stack([torch.zeros_like(d), d], dim=-1), None
            out = self.real_knn_query(k, query, points, unique, layout, want_dist, want_grouped, **kw)
            d = out[1][:, :, 1]
            self.keep(name, d < 5 * torch.mean(d, dim=1, keepdim=True), 1)
            return out
        if k == 5 and unique:                                    # inter-level neighbours
            name = "l%d_fm" % l
            if self.mode == "replay":
                idx = self.per_patch(name, l, np.int64)
                grouped = None
                if want_grouped:
                    owner = layout["pts_of"].long() if layout and layout.get("pts_of") is not None else \
                        torch.arange(points.size(0), device=points.device)
                    grouped = points[owner.view(-1, 1, 1), idx]
                return idx, None, grouped
            out = self.real_knn_query(k, query, points, unique, layout, want_dist, want_grouped, **kw)
            self.keep(name, out[0], P_MAX[l])
            return out
        if not unique and k == 312:                              # the inner patches
            name = "l%d_pidx" % l
            if self.mode == "replay":
                idx = self.per_patch(name, l, np.int64).view(self.B, P_MAX[l], 312)
                bb = torch.arange(self.B, device=points.device).view(-1, 1, 1)
                return idx, None, points[bb, idx]
            out = self.real_knn_query(k, query, points, unique, layout, want_dist, want_grouped, **kw)
            self.keep(name, out[0].reshape(self.B * P_MAX[l], 312), P_MAX[l])
            return out
        raise AssertionError("unexpected knn_query: k=%d unique=%s at level %d" % (k, unique, l))

    def fps(self, xyz, npoint, n_arr=None, m_arr=None):
        if m_arr is not None:                                    # inner seeds (before the level's graphs)
            l = 1 + self.graph_calls // 4
            name = "l%d_seeds" % l
            if self.mode == "replay":
                return self.per_patch(name, l, np.int32).view(self.B, P_MAX[l])[:, :npoint].contiguous()
            out = self.real_fps(xyz, npoint, n_arr, m_arr)
            self.keep(name, out.reshape(self.B * npoint), npoint)
            return out
        l = self.graph_calls // 4                                # the resampling closes the level
        name = "l%d_fps" % l
        self.levels_closed += 1
        if self.mode == "replay":
            return torch.from_numpy(np.stack([np.asarray(self.rec(name, q)).astype(np.int32)
                                              for q in self.ids])).to(self.dev)
        out = self.real_fps(xyz, npoint, n_arr, m_arr)
        self.keep(name, out, 1)
        return out


def first_flip(chain, g, i, q):
    """Name of the first choice (in call order) in which this build's recorded run differs from the reference's for
    outer patch q (= position i of the batch), or None.  Sets are compared as sets, sequences position by position;
    dead inner-patch slots are ignored."""
    for name in ORDER:
        l = int(name[1])
        ref = np.asarray(g["p%d_%s" % (q, name)])
        mine = np.asarray(chain.seen[name][i])
        kind = name.split("_")[1]
        if kind == "mask":
            same = np.array_equal(mine.reshape(-1), ref.reshape(-1))
        elif kind in ("seeds", "fps"):
            same = np.array_equal(mine.reshape(-1)[:ref.size].astype(np.int64), ref.reshape(-1).astype(np.int64))
        else:
            live = 1 if l == 1 else ref.shape[0]
            a = mine.reshape((-1,) + ref.shape[1:])[:live].astype(np.int64)
            r = ref.astype(np.int64)
            if kind.startswith("graph"):
                a, r = a[..., 1:], r[..., 1:]
            same = np.array_equal(np.sort(a, axis=-1), np.sort(r, axis=-1))
            if same and kind == "pidx":
                same = np.array_equal(a, r)         # (the ORDER inside an inner patch numbers its rows for the graphs)
        if not same:
            return name
    return None


def run_chain(ops, net, g, ids, dev, mode):
    """-> (chain, per-level clouds [(B, n_l, 3)] in the outer patch's normalised frame, x16 (B,3,4992))"""
    x = torch.from_numpy(np.stack([g["p%d_in" % q] for q in ids])).to(dev)
    chain = Chain(ops, g, ids, dev, mode)
    saved, net.trace = net.trace, []
    try:
        with chain, torch.no_grad():
            out = net(x, ratio=16)
        levels = [rec["cloud"].detach().cpu().numpy() for rec in net.trace]
    finally:
        net.trace = saved
    return chain, levels, out.detach().cpu().numpy()
