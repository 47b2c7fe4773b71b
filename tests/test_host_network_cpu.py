"""not-gpu: the product's HOST logic (module wiring, batched/ragged Net control flow) runs on CPU
with the kernels served by the oracle stand-in (oracle/backend.py) and is compared with
fixtures the reference's own Python produced (oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import golden, pkg, sphere
from oracle.backend import OracleBackend


@pytest.fixture()
def net_modules(orc, monkeypatch):
    ops = pkg("network.operations")
    ups = pkg("network.upsampler")
    monkeypatch.setattr(ops, "BACKEND", OracleBackend())
    return ops, ups


def _net(ups):
    net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5)
    state = golden("net16_state.npz")
    sd = {k: torch.from_numpy(state[k]) for k in state.files if k != "meta"}
    missing = net.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return net.eval()


def test_state_dict_keys_and_shapes_match_reference(net_modules):
    _, ups = net_modules
    net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5)
    state = golden("net16_state.npz")
    ours = net.state_dict()
    ref_keys = sorted(k for k in state.files if k != "meta")
    assert sorted(ours.keys()) == ref_keys
    for k in ref_keys:
        assert tuple(ours[k].shape) == state[k].shape, k
    assert sum(v.numel() for v in ours.values()) == 304108          # SURVEY 3.2 [probe]


def test_random_init_is_the_references(net_modules):
    """Same construction + init order as the reference => same weights under the same seed."""
    _, ups = net_modules
    torch.manual_seed(0)
    net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5)
    state = golden("net16_state.npz")
    for k, v in net.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), state[k], err_msg=k)


def test_level_forward_matches_reference(net_modules):
    _, ups = net_modules
    net = _net(ups)
    g = golden("level_forward.npz")
    with torch.no_grad():
        patch = torch.from_numpy(g["patch"])
        x1, f1 = net.levels["level_1"](patch, patch, previous_level4=None)
        np.testing.assert_allclose(x1.numpy(), g["l1_xyz"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(f1.numpy(), g["l1_feat"], rtol=1e-4, atol=1e-4)
        x2, f2 = net.levels["level_2"](torch.from_numpy(g["l2_in"]), torch.from_numpy(g["l2_in_norm"]),
                                       previous_level4=(patch, torch.from_numpy(g["l1_feat"])))
        np.testing.assert_allclose(x2.numpy(), g["l2_xyz"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(f2.numpy(), g["l2_feat"], rtol=1e-4, atol=1e-4)


def _chamfer(orc, a, b):
    return float(orc.chamfer_loss(a, b))


def _set_close_fraction(orc, y, ref, tol=1e-5):
    """fraction of points of y (1,3,n) with a point of ref within tol, and vice versa"""
    d1, _, d2, _ = orc.nmdistance_fwd(np.ascontiguousarray(y.transpose(0, 2, 1)),
                                      np.ascontiguousarray(ref.transpose(0, 2, 1)))
    return min((np.sqrt(d1) <= tol).mean(), (np.sqrt(d2) <= tol).mean())


@pytest.mark.parametrize("ratio", [2, 4, 16])
def test_net_eval_matches_reference(orc, net_modules, ratio):
    """Up to 4x every output coordinate is within 1e-5 of the reference's, position by position.
    At 16x the reference's own expanded-form distances (|a|^2 - 2ab + |b|^2 through a BLAS matmul,
    operations.py:158-161) carry ~1e-7 absolute noise against 5th/6th-neighbour gaps of ~1e-5 in the
    level-4 inter-level search, so a fraction of a percent of neighbour choices differ between ANY
    two evaluation orders (the reference on another BLAS included); a flipped choice moves a few
    points and re-orders the FPS sequence after it.  The bar there: >= 95 % of the output points
    coincide (1e-5) with a reference point as a SET, and the clouds agree under Chamfer."""
    _, ups = net_modules
    net = _net(ups)
    g = golden("net_eval.npz")
    with torch.no_grad():
        y = net(torch.from_numpy(g["patch"]), ratio=ratio).numpy()
    ref = g["x%d" % ratio]
    assert y.shape == ref.shape == (1, 3, 312 * ratio)
    if ratio <= 4:
        np.testing.assert_allclose(y, ref, rtol=0, atol=1e-5)
    else:
        # measured on this path: 0.9998 of the 4992 points coincide as a set, Chamfer 3.9e-10
        assert _set_close_fraction(orc, y, ref) >= 0.999
    # squared-distance Chamfer; the median squared point spacing of the 4992-point output is 6.8e-3
    assert _chamfer(orc, y, ref) < (1e-10 if ratio <= 4 else 1e-9)


@pytest.mark.parametrize("ratio", [2, 4, 8])
def test_net_eval_input_smaller_than_a_patch(orc, net_modules, ratio):
    """A 300-point input (< max_num_point = 312): levels re-patch with k = min(num_point, 312) = 300
    (reference upsampler.py:120-128).  Fixture from the reference's own Net."""
    _, ups = net_modules
    net = _net(ups)
    g = golden("net_small.npz")
    with torch.no_grad():
        y = net(torch.from_numpy(g["patch"]), ratio=ratio).numpy()
    ref = g["x%d" % ratio]
    assert y.shape == ref.shape == (1, 3, 300 * ratio)
    assert int(net.small_cloud_events) == 0
    if ratio <= 4:
        np.testing.assert_allclose(y, ref, rtol=0, atol=1e-5)
    else:
        assert _set_close_fraction(orc, y, ref) >= 0.99
        assert _chamfer(orc, y, ref) < 1e-8


def test_outlier_filter_keeps_at_least_four_fifths():
    """Why the reference's `k = min(k, N')` (upsampler.py:75-78) cannot trigger: the filter keeps d < 5 mean(d); at
    most N/5 non-negative values reach five times their mean (Markov), a level's input holds >= 2k points, so
    N' >= 1.6 k.  Checked on adversarial distance vectors incl. heavy tails; the only way below is non-finite
    input or all-zero distances (every point duplicated), where nothing passes `0 < 0` -- and the reference
    itself then fails on an empty FPS."""
    rng = np.random.default_rng(0)
    for trial in range(200):
        n = int(rng.integers(2, 2000))
        kind = trial % 4
        if kind == 0:
            d = rng.pareto(0.3, n)
        elif kind == 1:
            d = np.where(rng.random(n) < 0.2, 1.0, 1e-9)           # the extremal case: a fifth at the threshold
        elif kind == 2:
            d = np.exp(rng.normal(0, 6, n))
        else:
            d = np.where(rng.random(n) < 0.5, 0.0, rng.random(n))
        d = torch.from_numpy(d.astype(np.float32)).view(1, n)
        if not torch.isfinite(d).all() or float(d.sum()) == 0 or not np.isfinite(float(d.mean())):
            continue
        mask = d < 5 * torch.mean(d, dim=1, keepdim=True)
        assert int(mask.sum()) >= int(np.floor(0.8 * n)) - 1, (trial, n, int(mask.sum()))
    z = torch.zeros(1, 100)
    assert int((z < 5 * z.mean(dim=1, keepdim=True)).sum()) == 0    # all duplicated: everything is dropped


def test_net_eval_levels_teacher_checked(net_modules):
    """Level by level against what the reference's Level.forward saw and produced in its 16x run:
    levels 1-2 (inputs and outputs) within 1e-5 everywhere; levels 3-4 inputs within 1e-5 and
    >= 98 % of their output points within 1e-5 (see test_net_eval_matches_reference)."""
    _, ups = net_modules
    net = _net(ups)
    g = golden("net_eval.npz")
    lv = golden("net_levels_x16.npz")
    net.trace = []
    with torch.no_grad():
        net(torch.from_numpy(g["patch"]), ratio=16)
    assert len(net.trace) == 4
    for l, t in enumerate(net.trace, 1):
        ref_in, ref_out = lv["l%d_patch_xyz" % l], lv["l%d_out_norm" % l]
        P = ref_in.shape[0]
        if l > 1:
            assert int(t["patch_num"][0]) == P
        mine_in = t["patch_xyz"][:P].numpy()
        mine_out = t["out_norm"][:P].numpy()
        if l < 4:
            np.testing.assert_allclose(mine_in, ref_in, rtol=0, atol=1e-5, err_msg="level %d input" % l)
        else:   # downstream of the level-3 flips
            assert (np.abs(mine_in - ref_in).max(axis=1) <= 1e-5).mean() >= 0.98
        err = np.abs(mine_out - ref_out).max(axis=1)
        if l < 3:
            assert err.max() <= 1e-5, (l, err.max())
        else:   # first neighbour flips of the noisy inter-level search (3120 / 6240 candidates)
            assert (err <= 1e-5).mean() >= 0.98, (l, (err <= 1e-5).mean())


def test_net_eval_batched_equals_one_patch_at_a_time(orc, net_modules):
    """The batched ragged path must give, per patch, what the reference's one-patch-at-a-time
    loop gives (main.py:237-244)."""
    _, ups = net_modules
    net = _net(ups)
    from conftest import sphere
    ops = pkg("network.operations")
    patches = torch.from_numpy(np.ascontiguousarray(sphere(21, 312, 3).transpose(0, 2, 1)))
    patches, _, _ = ops.normalize_point_batch(patches)
    with torch.no_grad():
        together = net(patches, ratio=4).numpy()
        single = np.concatenate([net(patches[i:i + 1], ratio=4).numpy() for i in range(3)])
    np.testing.assert_array_equal(together, single)


@pytest.mark.parametrize("ratio", [2, 4, 16])
def test_net_train_forward_matches_reference(net_modules, ratio):
    _, ups = net_modules
    net = _net(ups).train()
    g = golden("net_train.npz")
    seeds = [torch.from_numpy(s) for s in g["seeds_x%d" % ratio]]
    real = torch.randint
    calls = []

    def replay(*a, **kw):
        calls.append(1)
        return seeds[len(calls) - 1].clone()
    torch.randint = replay
    try:
        with torch.no_grad():
            pred, gt = net(torch.from_numpy(g["input"]), ratio=ratio, gt=torch.from_numpy(g["gt_x%d" % ratio]))
    finally:
        torch.randint = real
    assert len(calls) == len(seeds)
    assert tuple(pred.shape) == (4, 3, 624) and tuple(gt.shape) == g["gtout_x%d" % ratio].shape
    np.testing.assert_array_equal(gt.numpy(), g["gtout_x%d" % ratio])
    np.testing.assert_allclose(pred.numpy(), g["pred_x%d" % ratio], rtol=0, atol=1e-5)


def _train_grads(ups, device):
    """The net_train_grad.npz case on `device`: (pred, gt, input gradient, {parameter: gradient})."""
    g = golden("net_train_grad.npz")
    net = _net(ups).train().to(device)
    seeds = [torch.from_numpy(s) for s in g["seeds"]]
    real = torch.randint
    calls = []

    def replay(*a, **kw):
        calls.append(1)
        return seeds[len(calls) - 1].clone().to(kw.get("device", "cpu"))
    inp = torch.from_numpy(g["input"]).to(device).requires_grad_(True)
    torch.randint = replay
    try:
        pred, gt = net(inp, ratio=8, gt=torch.from_numpy(g["gt"]).to(device))
    finally:
        torch.randint = real
    assert len(calls) == len(seeds)
    (pred * torch.from_numpy(g["w"]).to(device)).sum().backward()
    grads = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    return g, pred, gt, inp.grad, grads


def check_train_grads(g, pred, gt, ginp, grads, tol, tol_median=None):
    """Every parameter gradient against the fixture, error relative to the tensor's largest gradient: the worst
    tensor below `tol`, the median tensor below `tol_median` (default: tol)."""
    np.testing.assert_array_equal(gt.cpu().numpy(), g["gtout"])
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g["pred"], rtol=0, atol=1e-5)
    ref = {k[5:]: g[k] for k in g.files if k.startswith("grad_level")}
    assert sorted(ref) == sorted(grads)
    errs = []
    for name, r in ref.items():
        mine = grads[name].cpu().numpy()
        scale = max(1e-6, float(np.abs(r).max()))
        errs.append(float(np.abs(mine - r).max()) / scale)
    assert max(errs) < tol, max(errs)
    assert float(np.median(errs)) < (tol if tol_median is None else tol_median), float(np.median(errs))
    gi = g["grad_input"]
    assert float(np.abs(ginp.cpu().numpy() - gi).max()) / float(np.abs(gi).max()) < tol


def test_net_train_backward_matches_reference(net_modules):
    """Gradients of the reference's training forward (ratio 8: level 3's previous cloud is a network output, so
    the inter-level skip's gathered neighbour COORDINATES carry a gradient too, reference operations.py:209-211):
    every parameter of levels 1-3 and the input, max error relative to the tensor's largest gradient."""
    _, ups = net_modules
    check_train_grads(*_train_grads(ups, "cpu"), tol=2e-5)          # measured 1.9e-6


def test_net_train_backward_reaches_every_level(net_modules):
    _, ups = net_modules
    net = _net(ups).train()
    from conftest import sphere
    inp = torch.from_numpy(np.ascontiguousarray(sphere(1, 312, 2).transpose(0, 2, 1)))
    gt = torch.from_numpy(np.ascontiguousarray(sphere(2, 1248, 2).transpose(0, 2, 1)))
    torch.manual_seed(0)
    pred, gto = net(inp, ratio=4, gt=gt)
    pred.square().mean().backward()
    for name, p in net.named_parameters():
        lvl = int(name.split("level_")[1][0])
        if lvl <= 2:
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
            assert p.grad.abs().sum() > 0 or name.endswith("bias"), name
        else:
            assert p.grad is None


# ---- the unused variants of the reference ---------------------------------------
def _load_prefixed(module, g, prefix):
    sd = {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}
    res = module.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return module.eval()


@pytest.mark.parametrize("nsample", [48, 1])
def test_sampled_dense_edge_conv_matches_reference(net_modules, nsample):
    """layers.py:67-112: FPS (or closest-to-centroid) subset, kNN of the subset in the full feature
    set, dense layers, max over k."""
    layers = pkg("network.layers")
    g = golden("adaptive_level.npz")
    conv = _load_prefixed(layers.SampledDenseEdgeConv(24, growth_rate=12, n=3, k=8), g, "sdec_state_")
    with torch.no_grad():
        y, sxyz, sidx = conv(torch.from_numpy(g["sdec_x"]), nsample, torch.from_numpy(g["sdec_xyz"]))
    np.testing.assert_array_equal(sidx.numpy().astype(np.int32), g["sdec_sidx_%d" % nsample])
    np.testing.assert_array_equal(sxyz.numpy(), g["sdec_sxyz_%d" % nsample])
    np.testing.assert_allclose(y.numpy(), g["sdec_y_%d" % nsample], rtol=0, atol=1e-5)


def test_adaptive_level_matches_reference(net_modules):
    """upsampler.py:377-512 with knn=8: same parameter names / shapes, same output."""
    _, ups = net_modules
    g = golden("adaptive_level.npz")
    lvl = ups.AdaptiveLevel(dense_n=3, growth_rate=12, knn=8, fm_knn=5)
    ref_keys = sorted(k[len("alevel_state_"):] for k in g.files if k.startswith("alevel_state_"))
    assert sorted(lvl.state_dict().keys()) == ref_keys
    _load_prefixed(lvl, g, "alevel_state_")
    with torch.no_grad():
        x, glob = lvl(torch.from_numpy(g["alevel_in"]), int(g["alevel_target"]))
    assert x.shape == g["alevel_xyz"].shape and glob.shape == g["alevel_global"].shape
    np.testing.assert_allclose(glob.numpy(), g["alevel_global"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(x.numpy(), g["alevel_xyz"], rtol=0, atol=1e-5)


def test_adaptive_level_refuses_knn_above_15(net_modules):
    """its last sampled layer asks for knn+1 of 16 points: the reference's topk raises, so does this"""
    _, ups = net_modules
    lvl = ups.AdaptiveLevel(knn=16).eval()
    with torch.no_grad(), pytest.raises((AssertionError, RuntimeError)):
        lvl(torch.from_numpy(np.ascontiguousarray(sphere(3, 312).transpose(0, 2, 1))), 100)


def test_group_ball_against_oracle(net_modules, orc):
    """operations.group_ball: consumer of the ball_query export (sampling_cuda.cu:269-317)."""
    ops, _ = net_modules
    pts = sphere(31, 700, 2)
    q = np.ascontiguousarray(pts[:, ::7][:, :60])
    grouped, idx = ops.group_ball(0.25, 16, torch.from_numpy(q), torch.from_numpy(pts), NCHW=False)
    ref = orc.ball_query(q, pts, 0.25, 16)
    np.testing.assert_array_equal(idx.numpy(), ref)
    np.testing.assert_array_equal(grouped.numpy(), np.stack([pts[b][ref[b]] for b in range(2)]))
    g2, idx2 = ops.group_ball(0.25, 16, torch.from_numpy(q).transpose(2, 1), torch.from_numpy(pts).transpose(2, 1))
    assert tuple(g2.shape) == (2, 3, 60, 16)
    np.testing.assert_array_equal(g2.permute(0, 2, 3, 1).numpy(), grouped.numpy())
