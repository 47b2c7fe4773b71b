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


# ---- compact record of ALL outer patches (tests/golden/c2_chain_all*.npz, oracle/make_golden.py make_chain_all_golden) ---------
# The k = 33 feature graphs and the k = 5 inter-level sets are not stored in full (170 MB for 48 patches) but as a 16-bit
# hash of every row's set plus the explicit set of every TIGHT row -- a row whose last member and first non-member are
# closer than tight_tau * max |x|^2 in the reference's own distances, i.e. a row in which another correct fp32
# evaluation may legitimately choose differently.  Replay = the build's own choice, checked row by row against the
# reference's hash; a differing row takes the reference's set if it is a tight one and is reported as UNEXPLAINED
# otherwise (a flip at a clear margin is a finding, not rounding noise).
SET_HASH_MUL = 40503


def set_hash16(idx):
    """torch (..., k) index sets -> (...,) int64 in [0, 65536): the twin of oracle/make_golden.py set_hash16"""
    return ((idx.long() + 1) * SET_HASH_MUL).sum(-1) & 0xFFFF


class ChainAll(Chain):
    def __init__(self, ops, g, ids, dev, mode):
        super(ChainAll, self).__init__(ops, g, ids, dev, mode)
        self.unexplained = []           # (name, outer patch, inner patch, row): differing rows that are not tight
        self.forced = {}                # name -> rows that took the reference's set
        self.hashes = {}                # record mode: name -> [per outer patch (P, 312) hashes]
        self._tight = {}

    # ---- the record ----------------------------------------------------------------------------------------------
    def _offset(self, q, l, b, graphs):
        """start of (level l, block b)'s P_l * 312 hashes in p<q>_gh (graphs) / of level l's in p<q>_fh"""
        off = 0
        for ll in ((1, 2, 3, 4) if graphs else (2, 3, 4)):
            per = self.live(ll, q) * 312
            if ll == l:
                return off + ((b - 1) * per if graphs else 0)
            off += (4 if graphs else 1) * per
        raise KeyError(l)

    def ref_hashes(self, l, b=None):
        """(B * P_MAX[l], 312) int64: the reference's set hashes, dead inner-patch slots repeat the last live patch"""
        rows = []
        for q in self.ids:
            live = self.live(l, q)
            if b is None:
                flat = np.asarray(self.g["p%d_fh" % q])
                o = self._offset(q, l, 1, False)
            else:
                flat = np.asarray(self.g["p%d_gh" % q])
                o = self._offset(q, l, b, True)
            a = flat[o:o + live * 312].reshape(live, 312).astype(np.int64)
            pad = P_MAX[l] - live
            if pad:
                a = np.concatenate([a, np.repeat(a[-1:], pad, axis=0)], axis=0)
            rows.append(a)
        return torch.from_numpy(np.concatenate(rows, axis=0)).to(self.dev)

    def tight_sets(self, q, l, b=None):
        """{(inner patch, row): sorted member array} of the tight rows of (outer patch q, level l[, block b])"""
        key = (q, l, b)
        if key not in self._tight:
            if b is None:
                t, r = np.asarray(self.g["p%d_ft" % q]), np.asarray(self.g["p%d_fr" % q])
                sel = np.nonzero(t[:, 0] == l)[0]
                self._tight[key] = {(int(t[i, 1]), int(t[i, 2])): r[i].astype(np.int64) for i in sel}
            else:
                t, r = np.asarray(self.g["p%d_gt" % q]), np.asarray(self.g["p%d_gr" % q])
                sel = np.nonzero((t[:, 0] == l) & (t[:, 1] == b))[0]
                self._tight[key] = {(int(t[i, 2]), int(t[i, 3])): r[i].astype(np.int64) for i in sel}
        return self._tight[key]

    def _settle(self, name, l, b, sets):
        """sets (B * P_MAX[l], 312, k) the build's own member sets -> list of (batch row, point row, reference set) to
        force; record mode: keeps the hashes instead."""
        mine = set_hash16(sets)
        if self.mode != "replay":
            h = mine.cpu().numpy()
            self.hashes[name] = [h[i * P_MAX[l]:(i + 1) * P_MAX[l]] for i in range(self.B)]
            return []
        diff = torch.nonzero(mine != self.ref_hashes(l, b)).cpu().numpy()
        todo = []
        for r, row in diff:
            i, p = int(r) // P_MAX[l], int(r) % P_MAX[l]
            q = self.ids[i]
            p_live = min(p, self.live(l, q) - 1)
            ref = self.tight_sets(q, l, b).get((p_live, int(row)))
            if ref is None:
                self.unexplained.append((name, q, p_live, int(row)))
            else:
                todo.append((int(r), int(row), ref))
        self.forced[name] = self.forced.get(name, 0) + len(todo)
        return todo

    # ---- the seams ------------------------------------------------------------------------------------------------
    def knn_graph(self, k, x, layout=None, optimistic=None):
        assert k == 33
        l, b = 1 + self.graph_calls // 4, self.graph_calls % 4 + 1
        self.graph_calls += 1
        name = "l%d_graph%d" % (l, b)
        out = self.real_graph(k, x, layout)
        assert out.shape[0] == self.B * P_MAX[l], (name, out.shape)
        for r, row, ref in self._settle(name, l, b, out[:, :, 1:]):
            out[r, row, 1:] = torch.from_numpy(ref).to(out.device, out.dtype)
        return out

    def knn_query(self, k, query, points, unique=True, layout=None, want_dist=True, want_grouped=True, unique_cache=None):
        kw = {} if unique_cache is None else {"unique_cache": unique_cache}
        if k == 33 and unique:                                   # the CPU stand-in's feature graphs
            l, b = 1 + self.graph_calls // 4, self.graph_calls % 4 + 1
            self.graph_calls += 1
            name = "l%d_graph%d" % (l, b)
            idx, dist, grouped = self.real_knn_query(k, query, points, unique, layout, want_dist, want_grouped, **kw)
            for r, row, ref in self._settle(name, l, b, idx[:, :, 1:]):
                idx[r, row, 1:] = torch.from_numpy(ref).to(idx.device, idx.dtype)
                if grouped is not None:
                    grouped[r, row] = points[r, idx[r, row]]
            return idx, dist, grouped
        if k == 5 and unique:                                    # inter-level neighbours
            l = self.graph_calls // 4
            name = "l%d_fm" % l
            idx, dist, grouped = self.real_knn_query(k, query, points, unique, layout, want_dist, want_grouped, **kw)
            todo = self._settle(name, l, None, idx)
            if todo:
                owner = layout["pts_of"].long() if layout and layout.get("pts_of") is not None else \
                    torch.arange(points.size(0), device=points.device)
                for r, row, ref in todo:
                    idx[r, row] = torch.from_numpy(ref).to(idx.device, idx.dtype)
                    if grouped is not None:
                        grouped[r, row] = points[owner[r], idx[r, row].long()]
            return idx, dist, grouped
        return super(ChainAll, self).knn_query(k, query, points, unique, layout, want_dist, want_grouped, unique_cache)


def first_flip_all(chain, g, i, q):
    """first_flip for a ChainAll run in record mode: graphs and inter-level sets are compared by their row hashes"""
    for name in ORDER:
        l = int(name[1])
        kind = name.split("_")[1]
        if kind.startswith("graph") or kind == "fm":
            b = int(kind[5]) if kind.startswith("graph") else None
            one = ChainAll(chain.ops, g, [q], torch.device("cpu"), "replay")
            ref = one.ref_hashes(l, b).numpy()[:one.live(l, q)]
            same = np.array_equal(np.asarray(chain.hashes[name][i])[:ref.shape[0]], ref)
        else:
            ref = np.asarray(g["p%d_%s" % (q, name)])
            mine = np.asarray(chain.seen[name][i])
            if kind == "mask":
                same = np.array_equal(mine.reshape(-1), ref.reshape(-1))
            elif kind in ("seeds", "fps"):
                same = np.array_equal(mine.reshape(-1)[:ref.size].astype(np.int64), ref.reshape(-1).astype(np.int64))
            else:                                                # pidx: the order inside an inner patch matters
                a = mine.reshape((-1,) + ref.shape[1:])[:ref.shape[0]].astype(np.int64)
                same = np.array_equal(a, ref.astype(np.int64))
        if not same:
            return name
    return None


def run_chain_all(ops, net, g, ids, dev, mode):
    """run_chain on the compact record: -> (chain, per-level clouds, x16)"""
    x = torch.from_numpy(np.stack([g["p%d_in" % q] for q in ids])).to(dev)
    chain = ChainAll(ops, g, ids, dev, mode)
    saved, net.trace = net.trace, []
    try:
        with chain, torch.no_grad():
            out = net(x, ratio=16)
        levels = [rec["cloud"].detach().cpu().numpy() for rec in net.trace]
    finally:
        net.trace = saved
    return chain, levels, out.detach().cpu().numpy()
