"""-m gpu: the network, Chamfer loss and the pipeline on the MI355X through the HIP kernels,
against the reference-generated fixtures and the oracle; and, at BASELINE.json's full sizes,
through size-independent properties."""
import os

import numpy as np
import pytest
import torch

from conftest import golden, pkg, sphere

pytestmark = pytest.mark.gpu


# measured on MI355X in round 3 (GPU run log in profiles/r03_parity.txt): thresholds = measured with <= 2x slack
# net eval 16x of one patch vs the reference fixture, measured on MI355X with three arithmetically equivalent kernel
# sets of this round (each a different fp32 summation order somewhere): set_close 0.9984 / 0.9998 / 0.9938 (8, 1 and
# 31 of 4992 points without a partner within 1e-5), Chamfer 2.3e-10 / 3.9e-10 / 1.3e-6; the oracle-driven CPU path
# scores 0.9998 / 3.9e-10.  Which near-tie of a feature-space kNN flips is summation-order noise;
# the thresholds are the worst of the three with 2x slack on the miss count / the Chamfer distance.
TH_NET16_SET = 0.9876
TH_NET16_CHAMFER = 2.7e-6


def _net(dev):
    ups = pkg("network.upsampler")
    net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5)
    state = golden("net16_state.npz")
    net.load_state_dict({k: torch.from_numpy(state[k]) for k in state.files if k != "meta"})
    return net.to(dev).eval()


def _set_close(orc, y, ref, tol=1e-5):
    d1, _, d2, _ = orc.nmdistance_fwd(np.ascontiguousarray(y.transpose(0, 2, 1)),
                                      np.ascontiguousarray(ref.transpose(0, 2, 1)))
    return min((np.sqrt(d1) <= tol).mean(), (np.sqrt(d2) <= tol).mean())


def test_backend_is_the_hip_library(dev):
    ops = pkg("network.operations")
    assert ops.BACKEND.name == "hip-gfx950"
    assert pkg("_lib").lib().tpu3_version().decode().endswith("gfx950")


def test_level_forward_on_device(dev):
    net = _net(dev)
    g = golden("level_forward.npz")
    with torch.no_grad():
        patch = torch.from_numpy(g["patch"]).to(dev)
        x1, f1 = net.levels["level_1"](patch, patch, previous_level4=None)
        np.testing.assert_allclose(x1.cpu().numpy(), g["l1_xyz"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(f1.cpu().numpy(), g["l1_feat"], rtol=1e-4, atol=1e-4)
        x2, f2 = net.levels["level_2"](torch.from_numpy(g["l2_in"]).to(dev),
                                       torch.from_numpy(g["l2_in_norm"]).to(dev),
                                       previous_level4=(patch, torch.from_numpy(g["l1_feat"]).to(dev)))
        np.testing.assert_allclose(x2.cpu().numpy(), g["l2_xyz"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(f2.cpu().numpy(), g["l2_feat"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("ratio", [2, 4, 16])
def test_net_eval_on_device(orc, dev, ratio):
    """Same bar as tests/test_host_network_cpu.py::test_net_eval_matches_reference."""
    net = _net(dev)
    g = golden("net_eval.npz")
    with torch.no_grad():
        y = net(torch.from_numpy(g["patch"]).to(dev), ratio=ratio).cpu().numpy()
    ref = g["x%d" % ratio]
    assert y.shape == ref.shape
    if ratio == 2:
        np.testing.assert_allclose(y, ref, rtol=0, atol=1e-5)
    elif ratio == 4:
        # rocBLAS sums in another order than the CPU BLAS the fixture came from: 1e-6-level
        # differences can re-order the level-2 FPS sequence, so compare as point SETS
        frac, cd = _set_close(orc, y, ref), float(orc.chamfer_loss(y, ref))
        print("net eval 4x on device vs reference fixture: set_close_1e-5 = %.4f, chamfer = %.3e" % (frac, cd))
        assert frac >= 0.97 and cd < 1e-4
    else:
        # 16x: four levels of discrete choices (feature kNN, FPS resampling, patch seeds).  Every level is pinned
        # on its own (test_level_teacher_forced_on_device: 1-5 one-ulp feature-graph flips per 12 480 queries);
        # end to end the cloud is compared as a point SET and in Chamfer distance, with the measured numbers:
        frac = _set_close(orc, y, ref)
        cd = float(orc.chamfer_loss(y, ref))
        print("net eval 16x on device vs reference fixture: set_close_1e-5 = %.4f, chamfer = %.3e" % (frac, cd))
        assert frac >= TH_NET16_SET and cd < TH_NET16_CHAMFER
    if ratio == 2:
        assert float(orc.chamfer_loss(y, ref)) < 1e-10


@pytest.mark.parametrize("ratio", [2, 4, 8])
def test_net_eval_input_smaller_than_a_patch_on_device(orc, dev, ratio):
    """300-point input: k = min(num_point, 312) = 300 in every re-patching (reference upsampler.py:120-128)."""
    net = _net(dev)
    g = golden("net_small.npz")
    with torch.no_grad():
        y = net(torch.from_numpy(g["patch"]).to(dev), ratio=ratio).cpu().numpy()
    ref = g["x%d" % ratio]
    assert y.shape == ref.shape
    assert int(net.small_cloud_events) == 0
    if ratio == 2:
        np.testing.assert_allclose(y, ref, rtol=0, atol=1e-5)
    else:
        assert _set_close(orc, y, ref) >= 0.97
        assert float(orc.chamfer_loss(y, ref)) < 1e-6


def _teacher(level):
    from test_teacher_levels_cpu import teacher_inputs
    g = golden("net_teacher_x16.npz")
    return g, teacher_inputs(g, level)


def _set_flips(mine, ref):
    a = np.sort(np.asarray(mine, np.int64), axis=-1)
    b = np.sort(np.asarray(ref, np.int64), axis=-1)
    return int((a != b).any(axis=-1).sum()), int(a[..., 0].size)


@pytest.mark.parametrize("level", [3, 4])
def test_level_teacher_forced_on_device(dev, level, monkeypatch):
    """Levels 3 / 4 of the 16x run on the HIP path, fed with what the reference's Level.forward
    received (3120 / 6240 merged previous points + their 264-channel features).  Every discrete
    choice is compared with the reference's: the inter-level sets (fm_knn = 5) must not flip at all;
    the feature graphs (k = 33 in 24-d) flip for a handful of the 12 480 queries per block -- the
    reference's BLAS-evaluated expanded-form distances against 33rd/34th-neighbour gaps of one ulp,
    tests/test_teacher_levels_cpu.py -- and a flip perturbs ITS patch; every patch without a flip must
    agree within 1e-5 everywhere."""
    ops = pkg("network.operations")
    net = _net(dev)
    g, (xyz, xyzn, prev_xyz, prev_feat) = _teacher(level)
    seen = []
    real = ops.BACKEND.knn_graph

    def spy(k, x, layout=None):
        out = real(k, x, layout)
        assert out is not None
        seen.append(out)
        return out
    monkeypatch.setattr(ops.BACKEND, "knn_graph", spy, raising=False)
    with torch.no_grad():
        out, feat = net.levels["level_%d" % level](xyz.to(dev), xyzn.to(dev),
                                                   previous_level4=(prev_xyz.to(dev), prev_feat.to(dev)))
    monkeypatch.undo()
    assert len(seen) == 4
    err = np.abs(out.cpu().numpy() - g["l%d_out" % level]).max(axis=1)
    frac = float((err <= 1e-5).mean())
    bad_patches = int((err > 1e-5).any(axis=1).sum())
    P = xyz.shape[0]
    idx, _, _ = ops.knn_query(5, xyz.to(dev).transpose(2, 1).contiguous(),
                              prev_xyz.to(dev).transpose(2, 1).contiguous(), unique=True,
                              layout=dict(pts_of=torch.zeros(P, dtype=torch.int32, device=dev)),
                              want_dist=False, want_grouped=False)
    flips, total = _set_flips(idx.cpu().numpy(), g["l%d_knn_idx" % level])
    gflips = [_set_flips(seen[b].cpu().numpy()[..., 1:], g["l%d_graph%d" % (level, b + 1)][..., 1:])[0]
              for b in range(4)]
    print("level %d teacher-forced on device: %.4f of %d output points within 1e-5 (%d of %d patches touched); "
          "inter-level sets flipped: %d of %d; feature graphs flipped per block: %s of %d"
          % (level, frac, err.size, bad_patches, P, flips, total, gflips, total))
    assert flips == 0, (flips, total)
    # measured (MI355X, round 3, lane-per-point DenseEdgeConv: plain ascending-k fma chains): level 3 -- one flip in
    # block 4 (none with round 2's 16x16x4 kernel, whose chains ran in another order: the flips ARE summation-order
    # noise), 0.9998 of the points within 1e-5; level 4 -- 1 / 1 / 1 / 5 of 12 480 queries per block, 2 of 40 patches
    # touched, 0.9923 of the points within 1e-5
    measured = {3: ([0, 0, 0, 1], 1, 0.9998), 4: ([1, 1, 1, 5], 2, 0.9923)}[level]      # (level 3: 0 or 1 flip by kernel set)
    assert all(f <= m + 1 for f, m in zip(gflips, measured[0])), gflips            # exact flip counts + 1
    assert bad_patches <= min(sum(gflips), measured[1] + 1), (bad_patches, gflips)  # a patch without a flip is exact
    assert frac >= (0.999 if level == 3 else 0.99), frac


@pytest.mark.parametrize("level", [3, 4])
def test_level_with_reference_graphs_is_exact_on_device(dev, level, monkeypatch):
    """The same call with the reference's four feature graphs replayed into the fused DenseEdgeConv
    kernels: nothing discrete is left to differ, so EVERY output coordinate of the HIP path must be
    within 1e-5 of the reference's (north_star: "upsampled xyz within 1e-5 fp32")."""
    ops = pkg("network.operations")
    net = _net(dev)
    g, (xyz, xyzn, prev_xyz, prev_feat) = _teacher(level)
    graphs = [torch.from_numpy(g["l%d_graph%d" % (level, b + 1)].astype(np.int32)).to(dev) for b in range(4)]
    calls = []

    def replay(k, x, layout=None):
        calls.append(k)
        return graphs[len(calls) - 1]
    monkeypatch.setattr(ops.BACKEND, "knn_graph", replay, raising=False)
    with torch.no_grad():
        out, feat = net.levels["level_%d" % level](xyz.to(dev), xyzn.to(dev),
                                                   previous_level4=(prev_xyz.to(dev), prev_feat.to(dev)))
    monkeypatch.undo()
    assert calls == [33] * 4
    np.testing.assert_allclose(out.cpu().numpy(), g["l%d_out" % level], rtol=0, atol=1e-5)
    if level == 3:
        np.testing.assert_allclose(feat.cpu().numpy(), g["l3_feat"], rtol=1e-4, atol=1e-4)


def test_net_eval_batched_equals_single_on_device(dev):
    net = _net(dev)
    ops = pkg("network.operations")
    patches = torch.from_numpy(np.ascontiguousarray(sphere(21, 312, 5).transpose(0, 2, 1))).to(dev)
    patches, _, _ = ops.normalize_point_batch(patches)
    with torch.no_grad():
        together = net(patches, ratio=8)
        single = torch.cat([net(patches[i:i + 1], ratio=8) for i in range(5)])
    # identical kernels and identical inputs per patch; the GEMMs see different batch sizes, so
    # allow rounding-level differences in the values and demand the same point sets
    assert together.shape == single.shape == (5, 3, 2496)
    # (a flipped near-tie in a kNN choice perturbs the rest of THAT patch: most patches must agree to
    # rounding, and none may differ by more than such a perturbation)
    diff = (together - single).abs().amax(dim=1)                    # (5, 2496)
    agree = (diff <= 1e-5).float().mean(dim=1)
    assert agree.median() > 0.99 and (agree > 0.99).sum() >= 3
    # a perturbed patch may resample (FPS) a different subset in a different order: compare as sets
    nn = torch.cdist(together.transpose(1, 2), single.transpose(1, 2)).amin(dim=2)     # (5, 2496)
    assert nn.mean(dim=1).max() < 0.02       # point spacing on the unit sphere is ~0.07


@pytest.mark.parametrize("ratio", [2, 4, 16])
def test_net_train_forward_on_device(dev, ratio):
    net = _net(dev).train()
    g = golden("net_train.npz")
    seeds = [torch.from_numpy(s).to(dev) for s in g["seeds_x%d" % ratio]]
    real = torch.randint
    calls = []

    def replay(*a, **kw):
        calls.append(1)
        return seeds[len(calls) - 1].clone()
    torch.randint = replay
    try:
        with torch.no_grad():
            pred, gt = net(torch.from_numpy(g["input"]).to(dev), ratio=ratio,
                           gt=torch.from_numpy(g["gt_x%d" % ratio]).to(dev))
    finally:
        torch.randint = real
    np.testing.assert_array_equal(gt.cpu().numpy(), g["gtout_x%d" % ratio])
    np.testing.assert_allclose(pred.cpu().numpy(), g["pred_x%d" % ratio], rtol=0, atol=1e-5)


def test_chamfer_loss_forward_backward_on_device(orc, dev):
    ml = pkg("network.model_loss")
    g = golden("chamfer.npz")
    a = torch.from_numpy(g["a"]).to(dev).requires_grad_()
    b = torch.from_numpy(g["b"]).to(dev).requires_grad_()
    assert abs(float(ml.ChamferLoss()(a, b)) - float(g["cd"])) < 1e-6
    assert abs(float(ml.ChamferLoss(threshold=2.0)(a, b)) - float(g["cd_thr"])) < 1e-6
    loss = ml.ChamferLoss(threshold=2.0, forward_weight=50.0)(a.transpose(2, 1).contiguous(), b)
    assert abs(float(loss) - float(g["cd_thr_w"])) < 1e-5
    # backward = the analytic gradient of the kernel definition (nmdistance_cuda.cu:154-173)
    cd = ml.ChamferLoss()(a, b)
    cd.backward()
    d1, i1, d2, i2 = orc.nmdistance_fwd(g["a"], g["b"])
    B, n, m = 3, g["a"].shape[1], g["b"].shape[1]
    g1 = np.full((B, n), 1.0 / (n * B), np.float32)
    g2 = np.full((B, m), 1.0 / (m * B), np.float32)
    r1, r2 = orc.nmdistance_bwd(g["a"], g["b"], g1, g2, i1, i2)
    np.testing.assert_allclose(a.grad.cpu().numpy(), r1, rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(b.grad.cpu().numpy(), r2, rtol=1e-4, atol=1e-8)


def test_skip_train_node_matches_autograd(dev):
    """network/upsampler.py _SkipTrain (fused forward that leaves the weights + scatter backward) against the
    autograd formulation of the same skip connection (reference :317-347) on the same neighbour indices."""
    ups, ops = pkg("network.upsampler"), pkg("network.operations")
    g = torch.Generator(device="cpu").manual_seed(5)
    B, N, M, C, K = 3, 312, 312, 264, 5
    xyz = torch.randn(B, N, 3, generator=g).to(dev)
    prev_xyz = (xyz + 0.05 * torch.randn(B, M, 3, generator=g).to(dev)).contiguous()
    x0 = torch.randn(B, N, C, generator=g).to(dev)
    pf0 = (x0 + 0.3 * torch.randn(B, M, C, generator=g).to(dev)).contiguous()
    gout = torch.randn(B, N, C, generator=g).to(dev)
    idx, _, _ = ops.knn_query(K, xyz, prev_xyz, unique=True, want_dist=False, want_grouped=False)
    res = []
    for fused in (True, False):
        x = x0.clone().requires_grad_(True)
        pf = pf0.clone().requires_grad_(True)
        if fused:
            y = ups._SkipTrain.apply(x * 1.0, pf, xyz, prev_xyz, None, idx)
        else:
            bsel = torch.arange(B, device=dev).view(-1, 1, 1)
            kf = pf[bsel, idx.long()]
            w = ups.Level.exponential_distance_cl(xyz, prev_xyz[bsel, idx.long()]) * ups.Level.exponential_distance_cl(x, kf)
            w = w / torch.sum(w + 1e-5, dim=-1, keepdim=True)
            y = 0.2 * torch.sum(w.unsqueeze(-1) * kf, dim=2) + x
        (y * gout).sum().backward()
        res.append((y.detach(), x.grad, pf.grad))
    assert (res[0][0] - res[1][0]).abs().max() < 1e-5
    assert (res[0][1] - res[1][1]).abs().max() < 1e-6
    assert (res[0][2] - res[1][2]).abs().max() < 1e-5 * max(1.0, float(res[1][2].abs().max()))


def test_net_train_backward_matches_reference_on_device(dev):
    """The reference's training backward (tests/golden/net_train_grad.npz: ratio 8, three levels, every parameter's
    gradient and the input's) through the device path: fused DenseEdgeConv, per-point layer and skip-connection
    nodes.  On CPU with the oracle backend the same check holds at 2e-6 (tests/test_host_network_cpu.py).  On the
    device a feature-space kNN graph picks the other of two near-tied neighbours here and there (the forward agrees
    to 1e-5, inputs to 1e-7) and the gradient follows it -- tools/train_grad_probe.py traces it block by block:
    measured worst tensor 7.3e-3 of its largest gradient, median tensor 2.9e-4.  A missing gradient path shows
    as O(1)."""
    import test_host_network_cpu as host
    ups = pkg("network.upsampler")
    host.check_train_grads(*host._train_grads(ups, dev), tol=2e-2, tol_median=1e-3)


def test_net_train_backward_with_reference_graphs_on_device(dev, monkeypatch):
    """The same fixture with the reference's twelve feature graphs (3 levels x 4 DenseEdgeConv blocks, recorded by
    oracle/make_golden.py::make_train_grad_golden) replayed into dec_train_fwd / dec_train_bwd: nothing discrete is
    left to differ, so every parameter gradient must agree with the reference's autograd to fp32 accuracy -- 1e-4 of
    each tensor's largest gradient, against the 2e-2 the free-running test above needs for its flipped near-ties
    (VERDICT round 3, item 1 vi).  Proves the attribution: the looseness above IS the graph flips."""
    import test_host_network_cpu as host
    ops, ups = pkg("network.operations"), pkg("network.upsampler")
    g = golden("net_train_grad.npz")
    graphs = [torch.from_numpy(g["graph%02d" % i].astype(np.int32)).to(dev) for i in range(12)]
    calls = []

    def replay(k, x, layout=None, optimistic=None):
        calls.append((k, tuple(x.shape)))
        return graphs[len(calls) - 1]
    monkeypatch.setattr(ops.BACKEND, "knn_graph", replay, raising=False)
    res = host._train_grads(ups, dev)
    monkeypatch.undo()
    assert calls == [(33, (2, 312, 24))] * 12
    host.check_train_grads(*res, tol=1e-4)


def test_training_step_runs_and_updates(dev):
    """config C3 shape: batch 32 patches, Chamfer fwd+bwd at n = m = 624, clip, Adam."""
    model_mod = pkg("model")
    ups = pkg("network.upsampler")
    torch.manual_seed(0)
    net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev)

    class Opt(object):
        lr_init = 0.0005
        ckpt = None
    model = model_mod.Model(net, "train", Opt())
    inp = torch.from_numpy(np.ascontiguousarray(sphere(1, 312, 32).transpose(0, 2, 1))).to(dev)
    lab = torch.from_numpy(np.ascontiguousarray(sphere(2, 312 * 16, 32).transpose(0, 2, 1))).to(dev)
    before = [p.detach().clone() for p in net.parameters()]
    model.set_input(inp, 4, label_pc=lab[:, :, :312 * 4].contiguous())
    model.optimize()
    assert model.step == 1
    assert tuple(model.predicted.shape) == (32, 3, 624) and tuple(model.gt.shape) == (32, 3, 624)
    changed = sum(int((a != b).any()) for a, b in zip(before, net.parameters()))
    assert changed > 20
    assert "cd_loss_x4" in model.error_log and np.isfinite(model.error_log["cd_loss_x4"])


class _OracleChamfer(torch.nn.Module):
    """CPU checker of the loss inside the training-step comparison: nm-distance forward / backward
    from the C oracle (oracle/ref_kernels.c), reduction in torch (the reference's formula)."""

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a, b):
            from oracle import oracle as orc
            d1, i1, d2, i2 = orc.nmdistance_fwd(a.detach().numpy(), b.detach().numpy())
            ctx.save_for_backward(a, b)
            ctx.idx = (i1, i2)
            return torch.from_numpy(d1), torch.from_numpy(d2)

        @staticmethod
        def backward(ctx, g1, g2):
            from oracle import oracle as orc
            a, b = ctx.saved_tensors
            ga, gb = orc.nmdistance_bwd(a.detach().numpy(), b.detach().numpy(), g1.contiguous().numpy(),
                                        g2.contiguous().numpy(), *ctx.idx)
            return torch.from_numpy(ga), torch.from_numpy(gb)

    def __init__(self):
        super().__init__()
        self.threshold = None

    def forward(self, pred, gt):
        d1, d2 = self.Fn.apply(pred.contiguous(), gt.contiguous())
        return torch.mean(torch.mean(d1, dim=1) + torch.mean(d2, dim=1))


@pytest.mark.parametrize("ratio", [4, 16])
def test_optimize_step_matches_cpu_path(orc, dev, ratio, monkeypatch):
    """Model.optimize (a16) at config C3's shape (B = 32 patches of 312 points): one step on the HIP
    path against the SAME modules on CPU with the oracle stand-in backend and the oracle's Chamfer --
    loss, every gradient and the parameters after Adam.  The random patch seeds of
    extract_xyz_feature_patch are pinned to the same values on both sides.
    ratio 16 = the metric's ratio: the reference's loss weight log2(16/16) is 0 there (model.py:72), so
    the step must leave every parameter exactly where it was -- asserted, not avoided.
    ratio 4: a real update.  Adam's first step moves every weight by lr * sign(g) (+- lr 1e-3), so the
    parameters agree within 1e-5 wherever the gradient is above rounding noise; both are reported."""
    from oracle.backend import OracleBackend
    model_mod, ups, ops = pkg("model"), pkg("network.upsampler"), pkg("network.operations")

    class Opt(object):
        lr_init = 0.001
        ckpt = None
    inp = torch.from_numpy(np.ascontiguousarray(sphere(1, 312, 32).transpose(0, 2, 1)))
    lab = torch.from_numpy(np.ascontiguousarray(sphere(2, 312 * ratio, 32).transpose(0, 2, 1)))
    seeds = []
    real_randint = torch.randint

    def run(device, backend, criteria):
        torch.manual_seed(0)
        net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(device)
        model = model_mod.Model(net, "train", Opt())
        if criteria is not None:
            model.chamfer_criteria = criteria
        calls = [0]

        def pinned(*a, **kw):
            dv = kw.get("device")
            kw["device"] = "cpu"
            if calls[0] == len(seeds):
                seeds.append(real_randint(*a, **kw))
            t = seeds[calls[0]]
            calls[0] += 1
            return t.to(dv)
        monkeypatch.setattr(torch, "randint", pinned)
        if backend is not None:
            monkeypatch.setattr(ops, "BACKEND", backend)
        before = {n: p.detach().clone().cpu() for n, p in net.named_parameters()}
        model.set_input(inp.to(device), ratio, label_pc=lab.to(device))
        model.optimize()
        monkeypatch.undo()
        # (levels above the trained ratio take no part in the step: their .grad stays None)
        grads = {n: (torch.zeros_like(p) if p.grad is None else p.grad.detach()).cpu()
                 for n, p in net.named_parameters()}
        after = {n: p.detach().cpu() for n, p in net.named_parameters()}
        return float(model.error_log["cd_loss_x%d" % ratio]), grads, before, after

    loss_g, grad_g, before_g, after_g = run(dev, None, None)
    loss_c, grad_c, before_c, after_c = run(torch.device("cpu"), OracleBackend(), _OracleChamfer())
    for n in before_g:
        assert torch.equal(before_g[n], before_c[n])
    if ratio == 16:
        assert loss_g == 0.0 and loss_c == 0.0
        for n in after_g:
            assert torch.equal(after_g[n], before_g[n]), n          # weight 0 => no update at all
            assert float(grad_g[n].abs().max()) == 0.0
        return
    assert abs(loss_g - loss_c) <= 1e-5 * max(1.0, abs(loss_c)), (loss_g, loss_c)
    worst, n_par, n_close, n_sig, n_sig_close = 0.0, 0, 0, 0, 0
    for n in grad_g:
        scale = float(grad_c[n].abs().max()) + 1e-12
        worst = max(worst, float((grad_g[n] - grad_c[n]).abs().max()) / scale)
        close = (after_g[n] - after_c[n]).abs() <= 1e-5
        sig = grad_c[n].abs() > 1e-4 * scale
        n_par += close.numel()
        n_close += int(close.sum())
        n_sig += int(sig.sum())
        n_sig_close += int((close & sig).sum())
    print("optimize step x%d: loss %.8f (HIP) vs %.8f (CPU oracle path); worst per-tensor gradient error %.2e "
          "of the tensor's max; parameters within 1e-5 after Adam: %d of %d, %d of %d with a gradient above noise"
          % (ratio, loss_g, loss_c, worst, n_close, n_par, n_sig_close, n_sig))
    assert worst <= 2e-2
    assert n_sig_close >= 0.99 * n_sig
    assert n_close >= 0.97 * n_par


def test_optimize_step_as_hipgraph_matches_eager(dev):
    """opt.graph_steps: the captured step must do what the eager step does (same seeds through the device
    generator, same update), and replay must keep counting steps and logging the loss."""
    model_mod, ups = pkg("model"), pkg("network.upsampler")

    def make(graph):
        class Opt(object):
            lr_init = 0.001
            ckpt = None
            graph_steps = graph
        torch.manual_seed(0)
        net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev)
        return model_mod.Model(net, "train", Opt())
    inp = torch.from_numpy(np.ascontiguousarray(sphere(1, 312, 32).transpose(0, 2, 1))).to(dev)
    lab = torch.from_numpy(np.ascontiguousarray(sphere(2, 312 * 2, 32).transpose(0, 2, 1))).to(dev)
    eager, graphed = make(False), make(True)
    for m in (eager, graphed):
        for step in range(3):
            m.set_input(inp, 2, label_pc=lab)
            m.optimize()
        assert m.step == 3
    # ratio 2 has one level and no random patch extraction: the two runs see identical inputs
    le, lg = eager.error_log["cd_loss_x2"], graphed.error_log["cd_loss_x2"]
    assert np.isfinite(le) and abs(le - lg) <= 1e-4 * abs(le), (le, lg)
    close = total = 0
    for a, b in zip(eager.net.parameters(), graphed.net.parameters()):
        close += int(((a - b).abs() <= 1e-4).sum())
        total += a.numel()
    assert close >= 0.97 * total, (close, total)


def test_captured_step_follows_the_chamfer_threshold_toggle(dev):
    """main.py's curriculum switches the Chamfer threshold on and off mid-training (set_threshold /
    unset_threshold).  Threshold and forward weight are scalar launch arguments, i.e. frozen into a captured step: they
    are part of the capture key, so a toggle captures a second graph and the replayed loss is the eager one (advisor,
    round 2: the toggle was silently ignored for shapes already captured)."""
    model_mod, ups = pkg("model"), pkg("network.upsampler")

    def make(graph):
        class Opt(object):
            lr_init = 0.001
            ckpt = None
            graph_steps = graph
        torch.manual_seed(0)
        net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev)
        return model_mod.Model(net, "train", Opt())
    inp = torch.from_numpy(np.ascontiguousarray(sphere(3, 312, 8).transpose(0, 2, 1))).to(dev)
    lab = torch.from_numpy(np.ascontiguousarray(sphere(4, 312 * 2, 8).transpose(0, 2, 1))).to(dev)
    lab[:, :, :6] *= 2.5                                    # strong outliers: the threshold changes the loss
    eager, graphed = make(False), make(True)
    for m in (eager, graphed):
        for step in range(4):
            if step == 2:
                m.chamfer_criteria.set_threshold(1.5)
            m.set_input(inp, 2, label_pc=lab)
            m.optimize()
    le, lg = eager.error_log["cd_loss_x2"], graphed.error_log["cd_loss_x2"]
    assert np.isfinite(le) and abs(le - lg) <= 1e-4 * abs(le), (le, lg)
    assert len(graphed._captured) == 2                      # one graph per (shapes, threshold, weight)
    keys = sorted(graphed._captured, key=lambda k: (k[3] is not None, k[3] or 0))
    assert keys[0][3] is None and keys[1][3] == 1.5
    # the parameters after two plain + two thresholded steps agree between eager and captured execution
    close = total = 0
    for a, b in zip(eager.net.parameters(), graphed.net.parameters()):
        close += int(((a - b).abs() <= 1e-4).sum())
        total += a.numel()
    assert close >= 0.97 * total, (close, total)
    # and they differ from a run that ignores the toggle
    ignored = make(False)
    for step in range(4):
        ignored.set_input(inp, 2, label_pc=lab)
        ignored.optimize()
    diff = sum(float((a - b).abs().max()) for a, b in zip(eager.net.parameters(), ignored.net.parameters()))
    assert diff > 1e-4


def test_pipeline_on_device_against_reference_driver(orc, dev):
    pipe = pkg("pipeline")
    g = golden("pc_prediction.npz")
    net = _net(dev)
    cloud = torch.from_numpy(g["cloud"]).to(dev)
    seed_idx, patches, pidx = pipe.extract_outer_patches(cloud, 312, 3)
    np.testing.assert_array_equal(seed_idx.cpu().numpy(), g["seed_idx"])
    ref_pidx = g["patch_idx"].astype(np.int64)
    assert (np.sort(pidx.cpu().numpy(), -1) == np.sort(ref_pidx, -1)).all()
    final = pipe.upsample(net, cloud, 312, 4, 3).cpu().numpy()
    assert final.shape == (1, 3, 4000)
    assert _set_close(orc, final, g["final"]) >= 0.99


def test_dense_edge_conv_training_idx_can_be_fed_back(dev):
    """DenseEdgeConv under autograd on the device (the fused training kernels, csrc/dec_train.hip) returns its
    neighbour indices as int64 like the reference (network/layers.py:6-42: torch.topk indices) and like the sibling
    training paths; handing them back through `idx=` must reproduce the block's output (advisor, round 4)."""
    layers = pkg("network.layers")
    torch.manual_seed(3)
    blk = layers.DenseEdgeConv(24, growth_rate=12, n=3, k=32).to(dev).train()
    x = torch.randn(4, 312, 24, device=dev, requires_grad=True)
    y, idx = blk.forward_cl(x)
    assert idx.dtype == torch.int64 and tuple(idx.shape) == (4, 312, 32)
    y2, idx2 = blk.forward_cl(x, idx=idx)
    assert torch.equal(y, y2) and torch.equal(idx, idx2)
    blk.fused_train = False                               # the hoisted autograd formulation takes int64 too
    try:
        y3, _ = blk.forward_cl(x, idx=idx)
    finally:
        del blk.fused_train
    torch.testing.assert_close(y3, y, rtol=1e-4, atol=1e-5)
    gathered = torch.gather(x.detach(), 1, idx[:, :, :1].expand(-1, -1, 24))     # torch.gather needs int64
    assert gathered.shape == x.shape


def test_graphed_upsample_equals_eager(dev):
    """pipeline.GraphedUpsample: one cloud's whole 16x pass (seeds, outer patches, four levels with every inner FPS /
    kNN, merge, final FPS on 16 workgroups) captured once as a hipGraph and replayed -- the result must be the eager
    call's bit for bit, for the captured cloud and for a different cloud replayed through the same graph, and a cloud
    with duplicated points (the optimistic feature graphs raise their event) must come back recomputed exactly."""
    pipe = pkg("pipeline")
    net = _net(dev)
    clouds = [torch.from_numpy(np.ascontiguousarray(sphere(60 + i, 5000).transpose(0, 2, 1))).to(dev) for i in range(2)]
    fast = pipe.GraphedUpsample(net, (1, 3, 5000), 312, 16, 3)
    for rep in range(2):
        for x in clouds:
            ref = pipe.upsample(net, x, 312, 16, 3)
            out = fast(x, clone=True)
            assert torch.equal(out, ref)
    dup = clouds[0].clone()
    dup[:, :, 4000:] = dup[:, :, :1000]
    ref = pipe.upsample(net, dup, 312, 16, 3)
    assert torch.equal(fast(dup, clone=True), ref)
    assert torch.equal(fast(clouds[1], clone=True), pipe.upsample(net, clouds[1], 312, 16, 3))
    with pytest.raises(ValueError):
        fast(torch.zeros(1, 3, 4000, device=dev))


def test_graphed_upsample_sees_weight_updates(dev):
    """(r6, advisor finding on r5) The fused DenseEdgeConv kernels read a PACKED operand blob and the folded prep
    weights, both built on the host side of a call; a hipGraph replay would keep using the blobs of the capture.
    GraphedUpsample compares the parameters' version key on every call and captures again: after an in-place update of
    EVERY parameter (what optimizer.step() / load_state_dict do) the replay must equal the eager call with the new
    weights, and must differ from the result under the old ones; a stale fault count left behind by an earlier call
    must not send a replay to the eager path."""
    pipe, ops = pkg("pipeline"), pkg("network.operations")
    net = _net(dev)
    x = torch.from_numpy(np.ascontiguousarray(sphere(71, 5000).transpose(0, 2, 1))).to(dev)
    fast = pipe.GraphedUpsample(net, (1, 3, 5000), 312, 16, 3)
    before = fast(x, clone=True)
    assert fast.captures == 1
    assert torch.equal(fast(x, clone=True), before) and fast.captures == 1
    saved = [p.detach().clone() for p in net.parameters()]
    try:
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(1.01)
        after = fast(x, clone=True)
        assert fast.captures == 2
        assert torch.equal(after, pipe.upsample(net, x, 312, 16, 3))
        assert not torch.equal(after, before)
        # an edit the version counters cannot see: the documented hooks
        with torch.no_grad():
            for p, q in zip(net.parameters(), saved):
                p.data.copy_(q)
        net.invalidate_weight_caches()
        fast.invalidate()
        assert torch.equal(fast(x, clone=True), before) and fast.captures == 3
    finally:
        with torch.no_grad():
            for p, q in zip(net.parameters(), saved):
                p.copy_(q)
        net.invalidate_weight_caches()


def test_pipeline_recomputes_when_a_cluster_fps_launch_faults(dev):
    """The final FPS of one cloud runs on 16 workgroups that spin on each other (csrc/fps_cluster.hip).  With a member
    made absent (tpu3_debug_fps_cluster_absent: it leaves at once, as if it had never become resident) the others give
    up together after a bounded number of polls, the fault is counted and the samples not taken are index 0 (in
    range).  pipeline.upsample must notice at its synchronisation point and recompute on the single-workgroup
    kernels: the result equals a run that never used the cluster form."""
    pipe, ops, lib = pkg("pipeline"), pkg("network.operations"), pkg("_lib").lib()
    net = _net(dev)
    x = torch.from_numpy(np.ascontiguousarray(sphere(5, 5000).transpose(0, 2, 1))).to(dev)
    be = ops.BACKEND
    saved = be.fps_cluster(0)
    try:
        ref = pipe.upsample(net, x, 312, 16, 3)
    finally:
        be.fps_cluster(saved)
    assert be.fps_cluster_faults(reset=True) == 0
    # the kernel level first: a faulted launch counts, and leaves nothing out of range behind
    pts = torch.from_numpy(sphere(9, 30000)).to(dev)
    assert lib.tpu3_debug_fps_cluster_absent(1) == 0
    try:
        idx = be.fps(pts, 3000)
        torch.cuda.synchronize()
        assert be.fps_cluster_faults(reset=True) > 0
        assert int(idx.min()) >= 0 and int(idx.max()) < 30000
        out = pipe.upsample(net, x, 312, 16, 3)          # faults inside, recomputed without the cluster form
    finally:
        assert lib.tpu3_debug_fps_cluster_absent(0) == 0
    assert be.fps_cluster_faults(reset=True) == 0            # the recomputation consumed and cleared the count
    assert torch.equal(out, ref)
    assert be.fps_cluster(-1) == -1                          # the default policy is back


def test_pipeline_recomputes_when_an_optimistic_graph_reports_duplicates(orc, dev):
    """The inference path launches the feature-space kNN graphs optimistically (no gated fallback launches).
    A cloud with duplicated points makes duplicated feature rows: the event must be seen by pipeline.upsample,
    which recomputes with the exact form -- same result as running the exact form from the start."""
    pipe, ops = pkg("pipeline"), pkg("network.operations")
    net = _net(dev)
    base = sphere(77, 800)
    cloud = np.concatenate([base, base[:, :200]], axis=1)                    # 200 duplicated points
    x = torch.from_numpy(np.ascontiguousarray(cloud.transpose(0, 2, 1))).to(dev)
    be = ops.BACKEND
    calls = []
    real = be.knn_graph

    def spy(k, xx, layout=None, optimistic=None):
        calls.append(be.optimistic_graph if optimistic is None else optimistic)
        return real(k, xx, layout, optimistic)
    be.knn_graph = spy
    try:
        out = pipe.upsample(net, x, 312, 2, 3)
    finally:
        del be.knn_graph
    assert True in calls and False in calls                  # optimistic first, exact on the recomputation
    assert be.optimistic_graph is False                      # the backend's default is restored: exact form
    ref = pipe.upsample(net, x, 312, 2, 3, optimistic_graph=False)
    assert torch.equal(out, ref)
    # direct callers (no synchronisation point of their own) get the exact form without asking
    calls.clear()
    be.knn_graph = spy
    try:
        pipe.pc_prediction(net, x, 312, 2, 3)
    finally:
        del be.knn_graph
    assert calls and not any(calls)


def test_pipeline_concurrent_sub_batches_and_side_stream(orc, dev):
    """upsample(net_streams=..., fps_stream=...): clouds split over concurrent streams, final FPS on
    a side stream.  Every cloud is independent, so the result must be the single-stream one up to
    the rounding of differently shaped GEMMs (compared as point sets, like the other network tests)."""
    pipe = pkg("pipeline")
    net = _net(dev)
    clouds = torch.cat([torch.from_numpy(np.ascontiguousarray(sphere(40 + i, 1000).transpose(0, 2, 1)))
                        for i in range(3)]).to(dev)
    ref = pipe.upsample(net, clouds, 312, 4, 3)
    nets = [torch.cuda.Stream(device=dev) for _ in range(2)]
    side = torch.cuda.Stream(device=dev)
    out = pipe.upsample(net, clouds, 312, 4, 3, net_streams=nets, fps_stream=side)
    side.synchronize()
    assert out.shape == ref.shape == (3, 3, 4000)
    for i in range(3):
        assert _set_close(orc, out[i:i + 1].cpu().numpy(), ref[i:i + 1].cpu().numpy()) >= 0.99
    # (r6) the stagger of the sub-batches (on by default) only orders launches: the same bits with and without, with the
    # join on the side stream or on the caller's, one cloud per sub-batch on three streams, and called twice in a row
    torch.cuda.synchronize()
    plain = pipe.upsample(net, clouds, 312, 4, 3, net_streams=nets, fps_stream=side, stagger=False)
    side.synchronize()
    assert torch.equal(plain, out)
    nets3 = [torch.cuda.Stream(device=dev) for _ in range(3)]
    a = pipe.upsample(net, clouds, 312, 4, 3, net_streams=nets3, fps_stream=side, sub_batch=1, stagger=True)
    b = pipe.upsample(net, clouds, 312, 4, 3, net_streams=nets3, fps_stream=side, sub_batch=1, stagger=True)
    c = pipe.upsample(net, clouds, 312, 4, 3, net_streams=nets3, sub_batch=1, stagger=True)      # join on the caller's stream
    torch.cuda.synchronize()
    one = torch.cat([pipe.upsample(net, clouds[i:i + 1], 312, 4, 3) for i in range(3)])
    assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, one)
    assert pkg("network.operations").STAGE_HOOK is None


def test_full_size_config_c2_properties(orc, dev):
    """BASELINE config C2 at full size (5000 -> 80000, 16x, 48 patches): properties that do not
    need a 300 s CPU run -- shape, finiteness, the FPS prefix property (the first m' picks of an
    m-point FPS are the m'-point FPS), no point picked twice, the first 300 picks bit-exact against
    the oracle, and the FPS covering-radius property on the 80 000 output points."""
    pipe, ops = pkg("pipeline"), pkg("network.operations")
    net = _net(dev)
    cand = sphere(0, 40000)
    sel, _ = orc.fps(cand, 5000)
    cloud = torch.from_numpy(np.ascontiguousarray(cand[:, sel[0]].transpose(0, 2, 1))).to(dev)
    merged = pipe.upsample(net, cloud, 312, 16, 3, final_fps=False)
    assert tuple(merged.shape) == (1, 3, 48 * 4992)
    assert torch.isfinite(merged).all()
    assert int(net.small_cloud_events) == 0
    mcl = merged.transpose(2, 1).contiguous()
    idx_full = ops.fps(mcl, 80000)
    idx_half = ops.fps(mcl, 20000)
    assert torch.equal(idx_full[:, :20000], idx_half)                       # prefix property
    assert idx_full.unique().numel() == 80000                               # no point picked twice
    out = torch.gather(mcl, 1, idx_full.long().unsqueeze(-1).expand(-1, -1, 3))
    # cross-check a slice of the big FPS against the oracle (first 300 picks: 72 M point-rounds)
    ref_idx, _ = orc.fps(mcl.cpu().numpy(), 300)
    np.testing.assert_array_equal(idx_full[:, :300].cpu().numpy(), ref_idx)
    # FPS 2-approximation property: the covering radius of the sample (largest distance from any
    # merged point to its nearest sample) never exceeds the smallest distance between two samples
    ml = pkg("network.model_loss")
    d_cover, _, _, _ = ml.nndistance(mcl, out)
    _, d_self, _ = ops.knn_query(2, out, out, unique=False, want_grouped=False)
    # (d_self comes from the expanded-form kNN distances: ~1e-7 absolute noise on O(1) coordinates)
    assert float(d_self[:, :, 1].min()) >= float(d_cover.max()) - 1e-6


def test_config_c5_shapes_large_patches(orc, dev):
    """BASELINE config C5 in miniature (num_point = 1024 outer patches, 312-point inner patches):
    the first level runs on 1024-point patches (multi-tile kNN graph, 1024-point DenseEdgeConv,
    k = 1024 patch extraction through the sort kernel), later levels re-patch to 312.  Properties:
    shapes, finiteness, run-to-run determinism, the final FPS bit-exact against the oracle on a
    prefix, and the level-1 output against the unfused torch formulation of the same weights."""
    pipe, ops, ups = pkg("pipeline"), pkg("network.operations"), pkg("network.upsampler")
    net = _net(dev)
    cand = sphere(5, 20000)
    cloud = torch.from_numpy(np.ascontiguousarray(cand.transpose(0, 2, 1))).to(dev)
    P = pipe.num_outer_patches(20000, 1024, 3)
    assert P == 58
    merged = pipe.upsample(net, cloud, 1024, 4, 3, final_fps=False)
    assert tuple(merged.shape) == (1, 3, P * 4096) and torch.isfinite(merged).all()
    again = pipe.upsample(net, cloud, 1024, 4, 3, final_fps=False)
    assert torch.equal(merged, again)
    out = pipe.upsample(net, cloud, 1024, 4, 3)
    assert tuple(out.shape) == (1, 3, 80000)
    mcl = merged.transpose(2, 1).contiguous()
    ref_idx, _ = orc.fps(mcl.cpu().numpy(), 200)
    np.testing.assert_array_equal(ops.fps(mcl, 200).cpu().numpy(), ref_idx)
    # level 1 on 1024-point patches: fused kernels vs the plain torch path (grad-enabled branch)
    _, patches, _ = pipe.extract_outer_patches(cloud, 1024, 3)
    pn, _, _ = ops.normalize_point_batch(patches.reshape(P, 1024, 3)[:6].transpose(2, 1).contiguous())
    with torch.no_grad():
        fused, _ = net.levels["level_1"](pn, pn, None)
    with torch.enable_grad():
        plain, _ = net.levels["level_1"](pn, pn, None)
    assert ((fused - plain.detach()).abs().amax(dim=1) <= 1e-5).float().mean() > 0.99


def _dec_block(layers, dev, k, seed):
    torch.manual_seed(seed)
    blk = layers.DenseEdgeConv(24, growth_rate=12, n=3, k=k).to(dev)
    for mconv in blk.mlps:
        torch.nn.init.xavier_uniform_(mconv.weight)
        torch.nn.init.uniform_(mconv.bias, -0.5, 0.5)
    return blk


# ---- derived error bounds of the fp16-operand kernels (mlp_precision "f16", BASELINE config C5) --------------------
# An MFMA of that flavour rounds BOTH operands to fp16 (unit roundoff u = 2^-11 in the normal range, absolute 2^-25
# below 2^-14) and accumulates exact products in fp32.  For one layer y = W x evaluated on an input that already
# carries an error e (|x~ - x| <= e elementwise):
#     |y~ - y| <= |W| (e + u (|x| + e)) (1 + u)  +  u |W| |x|  +  (cin + 2) 2^-24 |W| |x|  +  2^-24 sum|W|
# (input rounding on top of the inherited error, weight rounding, fp32 accumulation, subnormal operands).  ReLU and
# max are 1-Lipschitz: they pass the bound through.  The tests evaluate this bound in fp64 next to the fp64 result
# and require the kernel's error to stay below it EVERYWHERE -- a per-element statement, not a blanket tolerance.
_U16 = 2.0 ** -11


def _f16_layer_bound(abs_in, err_in, w):
    """abs_in, err_in (..., cin) fp64 >= 0, w (cout, cin) fp64 -> bound (..., cout) on |W x~ evaluated in f16/f32 - W x|."""
    aw = w.abs().t()
    cin = w.shape[1]
    carried = (err_in + _U16 * (abs_in + err_in)) * (1 + _U16)
    return carried @ aw + (_U16 + (cin + 2) * 2.0 ** -24) * (abs_in @ aw) + 2.0 ** -24 * aw.sum(0)


# 312 / 1024: the path's patch sizes (C2 / C5); 2048: `--num_point 2048` of the CLI help (table still in LDS);
# 2731 / 5000: beyond the LDS table -- global z table, patch split over several workgroups ("pWhole" mode)
@pytest.mark.parametrize("P,N,k", [(3, 312, 32), (5, 100, 16), (2, 312, 48), (1, 17, 16), (2, 1024, 32), (1, 200, 64),
                                   (2, 2048, 32), (1, 2731, 16), (2, 5000, 32)])
def test_dense_edge_conv_fused_matches_unfused(dev, P, N, k, monkeypatch):
    """The MFMA kernel against the plain torch formulation of the same block (fp32 reference of the
    same op, same neighbour indices): 1e-5 absolute on O(1) activations.  Also checks writing into a
    channel slice of a wider buffer (how the Level uses it)."""
    layers, ops = pkg("network.layers"), pkg("network.operations")
    blk = _dec_block(layers, dev, k, P * 100 + N)
    x = torch.randn(P, N, 24, device=dev)
    with torch.no_grad():
        assert blk.fused_reason(x) is None
        wide = torch.full((P, N, 84), 7.0, device=dev)
        y_f, idx_f = blk.forward_cl(x, out=wide[..., 12:72])      # 16-byte aligned channel slice
        assert y_f.data_ptr() == wide[..., 12:72].data_ptr()
        assert (wide[..., :12] == 7.0).all() and (wide[..., 72:] == 7.0).all()
        # the plain torch formulation on the SAME neighbour rows (a caller-supplied idx takes the generic path)
        y_u, idx_u = blk.forward_cl(x, idx=idx_f.long())
    assert torch.equal(idx_f.long(), idx_u)
    np.testing.assert_allclose(y_f.cpu().numpy(), y_u.cpu().numpy(), rtol=0, atol=1e-5)
    assert torch.equal(y_f[..., 36:], x)                      # the x_i pass-through channels
    # and the neighbour SET is the exact kNN's (the fused path returns index order, the kNN sorted by distance)
    if N <= 2048:
        idx_k, _, _ = ops.knn_query(k + 1, x, x, unique=True, want_dist=False, want_grouped=False)
        assert torch.equal(idx_f.long().sort(-1)[0], idx_k[:, :, 1:].sort(-1)[0])


# 312: five steps on four waves, the fifth split four ways; 330 / 624: two left over, split two ways; 448: three left
# over, dealt whole; k = 16 / 48: 4 / 12 slots per part
@pytest.mark.parametrize("P,N,k", [(7, 312, 32), (3, 330, 32), (2, 624, 16), (2, 312, 48), (2, 448, 32)])
@pytest.mark.parametrize("fold", [False, True])
def test_dense_edge_conv_split_steps_give_the_same_bits(dev, P, N, k, fold):
    """The left-over 64-point steps of a patch split by neighbour slots over the four waves (LDS maximum) against the
    same kernel with one wave per step: bit-identical rows, with and without the folded prep convolutions."""
    layers, ops = pkg("network.layers"), pkg("network.operations")
    lib = pkg("_lib").lib()
    blk = _dec_block(layers, dev, k, 7 * P + N)
    g = torch.Generator(device=dev).manual_seed(N + k)
    x = torch.randn(P, N, 24, device=dev, generator=g)
    idx = torch.randint(0, N, (P, N, k + 1), device=dev, dtype=torch.int32, generator=g)
    fw = torch.randn(72, 60, device=dev, generator=g) * 0.1
    fb = torch.randn(72, device=dev, generator=g)

    def run():
        out = torch.zeros((P, N, 60), device=dev)
        if not fold:
            ops.BACKEND.dense_edge_conv(x, idx, 1, k, blk.mlps, out)
            return (out,)
        acc, xnext = torch.zeros((P, N, 48), device=dev), torch.zeros((P, N, 24), device=dev)
        assert ops.BACKEND.dense_edge_conv_fold(x, idx, 1, k, blk.mlps, out, fw, fb, acc, 0, 0, xnext)
        return out, acc, xnext

    old = lib.tpu3_debug_dec_split(1)
    try:
        with torch.no_grad():
            a = run()
            lib.tpu3_debug_dec_split(0)
            b = run()
    finally:
        lib.tpu3_debug_dec_split(old)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    assert float(a[0].abs().max()) > 0


@pytest.mark.parametrize("P,N,k", [(7, 312, 32), (3, 330, 16), (2, 64, 32), (1, 1000, 48)])
@pytest.mark.parametrize("fold_n", [0, 24, 72])
def test_dense_edge_conv_packed_operands_give_the_same_bits(dev, P, N, k, fold_n):
    """tpu3_dense_edge_conv_pack_f32 + the *_pk_* launches (operand tables written once per set of weights, copied by
    every workgroup) against the launches that build the tables from the weights: bit-identical rows."""
    layers, ops = pkg("network.layers"), pkg("network.operations")
    blk = _dec_block(layers, dev, k, 5 * P + N)
    g = torch.Generator(device=dev).manual_seed(N + k + fold_n)
    x = torch.randn(P, N, 24, device=dev, generator=g)
    idx = torch.randint(0, N, (P, N, k + 1), device=dev, dtype=torch.int32, generator=g)
    fw = torch.randn(max(fold_n, 24), 60, device=dev, generator=g) * 0.1
    fb = torch.randn(max(fold_n, 24), device=dev, generator=g)

    def run(pack):
        out = torch.zeros((P, N, 60), device=dev)
        if not fold_n:
            ops.BACKEND.dense_edge_conv(x, idx, 1, k, blk.mlps, out, pack=pack)
            return (out,)
        acc, xnext = torch.zeros((P, N, 48), device=dev), torch.zeros((P, N, 24), device=dev)
        assert ops.BACKEND.dense_edge_conv_fold(x, idx, 1, k, blk.mlps, out, fw, fb, acc, 0, 0, xnext, pack=pack)
        return out, acc, xnext

    with torch.no_grad():
        pack = ops.BACKEND.dense_edge_conv_pack(blk.mlps, fw if fold_n else None)
        a, b = run(None), run(pack)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    assert float(a[0].abs().max()) > 0


def test_dense_edge_conv_module_repacks_when_a_weight_changes(dev):
    """DenseEdgeConv keeps the packed operands of its current weights: an in-place weight update must show in the next
    forward (version counters), and the cached blob must be reused while nothing changes."""
    layers = pkg("network.layers")
    blk = _dec_block(layers, dev, 16, 3)
    x = torch.randn(2, 312, 24, device=dev)
    with torch.no_grad():
        y0, _ = blk.forward_cl(x)
        p0 = blk._pack_cache[0][1]
        y1, _ = blk.forward_cl(x)
        assert blk._pack_cache[0][1] is p0 and torch.equal(y0, y1)
        blk.mlps[2].bias.add_(1.0)
        y2, _ = blk.forward_cl(x)
        assert blk._pack_cache[0][1] is not p0
    assert torch.allclose(y2[..., :12], y0[..., :12] + 1.0, atol=1e-5)
    assert torch.equal(y2[..., 12:], y0[..., 12:])


def test_modules_with_cached_operands_can_be_copied_and_pickled(dev):
    """The packed-operand blobs and fold plans carry stream events; they live outside the modules, so a block that has
    run can still be deep-copied and pickled, and the copy computes the same rows from its own cache."""
    import copy
    import io
    layers = pkg("network.layers")
    blk = _dec_block(layers, dev, 16, 5)
    x = torch.randn(2, 312, 24, device=dev)
    with torch.no_grad():
        y0, _ = blk.forward_cl(x)
        twin = copy.deepcopy(blk)
        buf = io.BytesIO()
        torch.save(blk, buf)
        y1, _ = twin.forward_cl(x)
    assert torch.equal(y0, y1)
    assert twin._pack_cache[0][1] is not blk._pack_cache[0][1]


@pytest.mark.parametrize("P,N,k", [(4, 312, 32), (3, 1024, 32), (1, 3000, 16)])
def test_dense_edge_conv_fp16_mfma_within_derived_bound(dev, P, N, k):
    """mlp_precision = "f16" (config C5: fp16 operands on the matrix cores, fp32 accumulate) on the same neighbour
    rows as the fp32 flavour.  The error against an fp64 evaluation of the block stays below the bound derived from
    the operand roundings layer by layer (see _f16_layer_bound) for every output element; the pass-through channels
    are bit-identical.  No silent fallback: a shape the fused kernel does not cover raises."""
    layers = pkg("network.layers")
    blk = _dec_block(layers, dev, k, 7 * N + k)
    x = torch.randn(P, N, 24, device=dev)
    with torch.no_grad():
        y32, idx = blk.forward_cl(x)
        blk.mlp_precision = "f16"
        y16, idx16 = blk.forward_cl(x)
        assert torch.equal(idx, idx16)                            # the kNN graph stays fp32
        assert torch.equal(y16[..., 36:], x)
        assert float((y16 - y32)[..., :36].abs().max()) > 0       # it really is another arithmetic
        # fp64 reference and error bound on the kernel's hoisted formulation:
        #   h0 = relu((W0a - W0b) x_i + W0b x_j + b0), h1 = relu(W1a h0 + W1b x_i + b1), h2 = W2a h1 + W2b h0 + W2c x_i + b2
        d = lambda t: t.detach().double().cpu()
        w0, w1, w2 = (d(c.weight).reshape(c.weight.size(0), -1) for c in blk.mlps)
        b0, b1, b2 = (d(c.bias) for c in blk.mlps)
        xd = d(x)
        nb = d(idx).long()[:, :, -k:] if idx.size(2) > k else d(idx).long()
        xj = torch.gather(xd.unsqueeze(1).expand(-1, N, -1, -1), 2, nb.unsqueeze(-1).expand(-1, -1, -1, 24))   # (P,N,k,24)
        xi = xd.unsqueeze(2)
        wa, wb = w0[:, :24], w0[:, 24:]
        zero = torch.zeros_like
        c0, e_c0 = xi @ (wa - wb).t() + b0, _f16_layer_bound(xi.abs(), zero(xi), wa - wb)
        z, e_z = xj @ wb.t(), _f16_layer_bound(xj.abs(), zero(xj), wb)
        h0, e0 = torch.relu(c0 + z), e_c0 + e_z + 2.0 ** -23 * (c0.abs() + z.abs())
        c1, e_c1 = xi @ w1[:, 12:].t() + b1, _f16_layer_bound(xi.abs(), zero(xi), w1[:, 12:])
        h1 = torch.relu(h0 @ w1[:, :12].t() + c1)
        e1 = _f16_layer_bound(h0, e0, w1[:, :12]) + e_c1
        c2, e_c2 = xi @ w2[:, 24:].t() + b2, _f16_layer_bound(xi.abs(), zero(xi), w2[:, 24:])
        h2 = h1 @ w2[:, :12].t() + h0 @ w2[:, 12:24].t() + c2
        e2 = _f16_layer_bound(h1, e1, w2[:, :12]) + _f16_layer_bound(h0, e0, w2[:, 12:24]) + e_c2
        ref = torch.cat([h2, h1, h0], dim=-1).amax(dim=2)                                  # (P,N,36)
        bound = torch.cat([e2, e1, e0], dim=-1).amax(dim=2) + 1e-6                         # max is 1-Lipschitz
        err = (d(y16)[..., :36] - ref).abs()
        ratio = float((err / bound).max())
        print("DenseEdgeConv f16 (%d,%d,k=%d): max |err| %.2e, max err/bound %.3f, mean bound %.2e"
              % (P, N, k, float(err.max()), ratio, float(bound.mean())))
        assert ratio <= 1.0, ratio
        assert float((d(y32)[..., :36] - ref).abs().max()) < 2e-5                          # (the fp32 flavour on the same scale)
        odd = layers.DenseEdgeConv(24, growth_rate=12, n=3, k=20).to(dev)
        odd.mlp_precision = "f16"
        with pytest.raises(RuntimeError, match="does not cover"):
            odd.forward_cl(x)


@pytest.mark.parametrize("P,N", [(7, 312), (2, 1024), (3, 100)])
def test_prep_convolutions_folded_into_dense_edge_conv(dev, P, N, monkeypatch):
    """Inference folds layer{2,3,4}_prep (reference upsampler.py:298-311) into the write-out of the DenseEdgeConv block
    before each (tpu3_dense_edge_conv_fold_f32): the level's 264-channel features and regressed coordinates against
    the unfolded path -- prep convolutions as their own kernel reading the concatenated buffer -- on the same
    weights.  Same arithmetic up to the order of the partial sums: 1e-5 on O(1) features wherever the feature-space
    neighbour sets agree (a differently rounded prep output can flip a near-tie of the next block's kNN graph, which
    then differs for that patch: counted, not hidden)."""
    ops, ups = pkg("network.operations"), pkg("network.upsampler")
    net = _net(dev)
    lvl = net.levels["level_1"]
    x = torch.from_numpy(np.ascontiguousarray(sphere(300 + N, N, P))).to(dev)
    calls = []
    real = ops.BACKEND.dense_edge_conv_fold

    def spy(*a, **kw):
        calls.append(a[6].shape[0])
        return real(*a, **kw)
    monkeypatch.setattr(ops.BACKEND, "dense_edge_conv_fold", spy, raising=False)
    with torch.no_grad():
        y1, f1 = lvl.forward_cl(x, x)
        assert calls == [72, 48, 24]                               # three blocks folded, the fourth has nothing after it
        monkeypatch.setattr(ups.Level, "fold_preps", False)
        y0, f0 = lvl.forward_cl(x, x)
        assert calls == [72, 48, 24]
    ok = ((f1 - f0).abs().amax(dim=(1, 2)) <= 2e-5)               # per patch
    print("prep fold %dx%d: %d of %d patches agree to 2e-5 on all 264 channels; max |diff| there %.2e"
          % (P, N, int(ok.sum()), P, float((f1 - f0)[ok].abs().max())))
    assert int(ok.sum()) >= P - 1
    assert float((y1 - y0)[ok].abs().max()) <= 1e-5
    # the first block's rows do not depend on the fold at all
    assert torch.equal(f1[..., 180:], f0[..., 180:])


def test_generic_path_is_reported_not_silent(dev):
    """A DenseEdgeConv shape outside the fused kernel (k = 20) runs through the generic PyTorch
    formulation -- with a RuntimeWarning and a counted event, never silently."""
    layers, ops = pkg("network.layers"), pkg("network.operations")
    blk = _dec_block(layers, dev, 20, 5)
    x = torch.randn(2, 100, 24, device=dev)
    ops.GENERIC_PATH_EVENTS.clear()
    with torch.no_grad(), pytest.warns(RuntimeWarning, match="generic"):
        y, idx = blk.forward_cl(x)
    assert tuple(y.shape) == (2, 100, 60) and sum(ops.GENERIC_PATH_EVENTS.values()) == 1


def test_interlevel_skip_fused_matches_unfused(dev, monkeypatch):
    """Level with a previous cloud shared by several patches (pts_of), ragged previous counts and
    duplicated previous points: the fused skip kernel against the plain torch formulation of
    network/upsampler.py:317-347 on the device.  1e-5 on the regressed coordinates, 1e-4 on the
    264-channel features (O(1..10) values)."""
    ops, ups = pkg("network.operations"), pkg("network.upsampler")
    net = _net(dev)
    lvl = net.levels["level_3"]
    torch.manual_seed(3)
    B, Bp, M = 12, 3, 936
    prev_xyz = torch.randn(Bp, M, 3, device=dev)
    prev_xyz = prev_xyz / prev_xyz.norm(dim=2, keepdim=True)
    prev_xyz[:, 624:] = prev_xyz[:, :312]                        # overlapping merged patches
    prev_feat = torch.randn(Bp, M, 264, device=dev)
    prev_feat[:, 624:] = prev_feat[:, :312]
    prev_count = torch.tensor([936, 900, 936], dtype=torch.int32, device=dev)
    owner = torch.arange(Bp, dtype=torch.int32, device=dev).repeat_interleave(B // Bp)
    xyz = torch.randn(B, 312, 3, device=dev)
    xyz = 0.3 * xyz / xyz.norm(dim=2, keepdim=True) + prev_xyz[owner.long(), :1]
    norm, _, _ = ops.normalize_point_batch(xyz.transpose(2, 1).contiguous())
    norm = norm.transpose(2, 1).contiguous()
    with torch.no_grad():
        y_f, f_f = lvl._forward_cl(xyz, norm, (prev_xyz, prev_feat, prev_count), owner, Bp)
        monkeypatch.delattr(ops.HipBackend, "interlevel_skip")
        y_u, f_u = lvl._forward_cl(xyz, norm, (prev_xyz, prev_feat, prev_count), owner, Bp)
    np.testing.assert_allclose(f_f.cpu().numpy(), f_u.cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(y_f.cpu().numpy(), y_u.cpu().numpy(), rtol=0, atol=1e-5)


@pytest.mark.parametrize("C,K,idx_dtype", [(264, 5, torch.int64), (256, 5, torch.int32), (64, 3, torch.int64),
                                           (288, 8, torch.int32), (272, 1, torch.int64), (300, 5, torch.int64),
                                           (6, 2, torch.int32)])
def test_interlevel_skip_kernel_against_formula(dev, C, K, idx_dtype):
    """tpu3_interlevel_skip_f32 alone against network/upsampler.py:317-347 written out in float64: every row
    layout of the kernel (one float4 per lane; the packed tail of 257..288 channels; the scalar path)."""
    ops = pkg("network.operations")
    g = torch.Generator(device="cpu").manual_seed(C * 10 + K)
    B, Bp, N, M = 7, 3, 100, 150
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    feat = torch.randn(B, N, C, generator=g).to(dev)
    pxyz = torch.rand(Bp, M, 3, generator=g).to(dev)
    pfeat = torch.randn(Bp, M, C, generator=g).to(dev)
    owner = torch.tensor([0, 0, 1, 1, 1, 2, 2], dtype=torch.int32, device=dev)
    idx = torch.randint(0, M, (B, N, K), generator=g).to(dev).to(idx_dtype)
    got = ops.BACKEND.interlevel_skip(xyz, feat.clone(), pxyz, pfeat, owner, idx)
    o = owner.long().view(-1, 1, 1)
    kf = pfeat.double()[o, idx.long()]                                     # (B,N,K,C)
    kp = pxyz.double()[o, idx.long()]
    ds = ((xyz.double().unsqueeze(2) - kp) ** 2).sum(-1)                   # (B,N,K)
    df = ((feat.double().unsqueeze(2) - kf) ** 2).sum(-1)
    hs = ds.min(-1, keepdim=True)[0].mean(-2, keepdim=True)
    hf = df.min(-1, keepdim=True)[0].mean(-2, keepdim=True)
    w = torch.exp(-ds / (hs / 2)) * torch.exp(-df / (hf / 2))
    w = w / (w + 1e-5).sum(-1, keepdim=True)
    ref = feat.double() + 0.2 * (w.unsqueeze(-1) * kf).sum(2)
    assert (got.double() - ref).abs().max() < 2e-5


@pytest.mark.parametrize("C,K,N,per,rows", [(264, 5, 312, 4, torch.float32), (264, 5, 312, 0, torch.float32),
                                            (256, 5, 100, 0, torch.float32), (288, 8, 330, 0, torch.float32),
                                            (64, 3, 17, 0, torch.float32), (264, 5, 1024, 2, torch.float32),
                                            (264, 5, 1024, 2, torch.float16), (256, 5, 312, 0, torch.float16)])
def test_interlevel_skip_one_launch_gives_the_two_kernels_bits(dev, C, K, N, per, rows):
    """The inference skip as ONE launch (a 16-wave workgroup per patch, distances and minima in LDS) against the two
    kernels with a global scratch (tpu3_debug_skip_fused): the same operations in the same order -- bit-identical
    rows, with and without the XCD-aware block mapping (`per_cloud`), with and without the packed tail."""
    ops = pkg("network.operations")
    lib = pkg("_lib").lib()
    g = torch.Generator(device="cpu").manual_seed(C + K + N)
    Bp, M = 8, 500
    B = Bp * (per if per else 3)
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    feat = torch.randn(B, N, C, generator=g).to(dev).to(rows)           # (fp16: activation storage mode "f16")
    pxyz = torch.rand(Bp, M, 3, generator=g).to(dev)
    pfeat = torch.randn(Bp, M, C, generator=g).to(dev).to(rows)
    owner = torch.repeat_interleave(torch.arange(Bp, dtype=torch.int32), B // Bp).to(dev)
    idx = torch.randint(0, M, (B, N, K), generator=g).to(dev)
    old = lib.tpu3_debug_skip_fused(1)
    try:
        one = ops.BACKEND.interlevel_skip(xyz, feat.clone(), pxyz, pfeat, owner, idx, per_cloud=per)
        lib.tpu3_debug_skip_fused(0)
        two = ops.BACKEND.interlevel_skip(xyz, feat.clone(), pxyz, pfeat, owner, idx, per_cloud=per)
    finally:
        lib.tpu3_debug_skip_fused(old)
    assert torch.equal(one, two)
    assert not torch.equal(one, feat)


# (5, 101): 505 points, not a multiple of the 8 points of a pass, one pass per workgroup; (21, 312): 6552 points = 819
# passes on 512 workgroups -- some walk two passes, some one (idle halves must add nothing to the weight gradients that the
# backward kernel accumulates across its passes)
@pytest.mark.parametrize("P,N", [(5, 101), (21, 312)])
@pytest.mark.parametrize("given_idx", [False, True])
def test_dense_edge_conv_fused_training_matches_autograd(dev, given_idx, P, N):
    """csrc/dec_train.hip (one launch per direction) against the autograd formulation of the same block
    (reference layers.py:44-64): output, input gradient and all six parameter gradients, on several patches
    with a non-multiple-of-8 point count."""
    layers = pkg("network.layers")
    torch.manual_seed(11)
    blk = layers.DenseEdgeConv(24, growth_rate=12, n=3, k=32).to(dev)
    x0 = torch.randn(P, N, 24, device=dev)
    idx = torch.randint(0, N, (P, N, 32), device=dev) if given_idx else None
    w = torch.randn(P, N, 60, device=dev)
    res = []
    for fused in (True, False):
        blk.fused_train, blk.hoist_train = fused, False
        blk.zero_grad()
        x = x0.clone().requires_grad_(True)
        y, used = blk.forward_cl(x, idx)
        (y * w).sum().backward()
        res.append((y.detach(), x.grad.detach(), [p.grad.detach().clone() for p in blk.parameters()], used))
    (ya, gxa, gpa, ia), (yb, gxb, gpb, ib) = res
    assert torch.equal(ia, ib)
    assert (ya - yb).abs().max() < 1e-4
    # A maximum over the 32 edges that two edges attain to within an ulp can go to either edge in the two formulations
    # (different summation order): the gradient of that channel then lands on another edge -- the point's row and its
    # neighbours' rows move.  Rare (seen: 6 of 6552 rows for one seed, none for others), so: (almost) every row equal.
    scale = max(1.0, float(gxb.abs().max()))
    off = int(((gxa - gxb).abs().max(-1)[0] > 1e-3 * scale).sum())
    assert off <= max(0, int(0.002 * P * N)), off
    tol = 1e-3 if off == 0 else 5e-3
    for a, b in zip(gpa, gpb):
        assert a.shape == b.shape and (a - b).abs().max() < tol * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("idx_dtype", [torch.int64, torch.int32])
def test_gather_neighbours_forward_and_backward(dev, idx_dtype):
    """The differentiable neighbour gather of the training path (tpu3_gather_rows_f32 / tpu3_scatter_add_rows_f32)
    against torch's advanced indexing and its autograd: rows bit-equal, gradients to 1e-5 (atomic summation
    order), repeated indices included."""
    layers = pkg("network.layers")
    g = torch.Generator(device="cpu").manual_seed(5)
    B, N, k, C = 3, 200, 32, 24
    x = torch.randn(B, N, C, generator=g).to(dev).requires_grad_(True)
    idx = torch.randint(0, N, (B, N, k), generator=g).to(dev).to(idx_dtype)
    idx[0, :, :4] = 7                                             # many contributions to one row
    w = torch.randn(B, N, k, C, generator=g).to(dev)
    y = layers.gather_neighbours(x, idx)
    (y * w).sum().backward()
    gx = x.grad.clone()
    x.grad = None
    bsel = torch.arange(B, device=dev).view(-1, 1, 1)
    y2 = x[bsel, idx.long()]
    (y2 * w).sum().backward()
    assert torch.equal(y, y2)
    assert (gx - x.grad).abs().max() < 1e-4 * max(1.0, float(x.grad.abs().max()))


@pytest.fixture(params=["fp32", "split_bf16"])
def regressor_form(request):
    """The two arithmetic forms of the regressor's matrix layers (tpu3_split_bf16): the test body runs under each, the
    setting of the process is restored afterwards."""
    be = pkg("network.operations").BACKEND
    old = be.split_bf16(request.param == "split_bf16")
    yield request.param
    be.split_bf16(old)


@pytest.mark.parametrize("m,cin", [(5000, 264), (17, 264), (1, 260), (4099, 272), (128 * 700 + 3, 264)])
def test_linear_wide_matches_torch(dev, m, cin, regressor_form):
    """tpu3_linear_wide_f32 / tpu3_linear_wide_sb_f32 (per-point half of up_layer1, 264 -> 128) against torch in fp64;
    the weight is a column slice of the (128, 265) convolution weight like at its call site.  (17, 264) after
    (5000, 264): a new weight tensor at a recycled address must not find the previous one's split image.)"""
    ops = pkg("network.operations")
    g = torch.Generator(device="cpu").manual_seed(m + cin)
    x = torch.randn(m, cin, generator=g).to(dev)
    wfull = (torch.randn(128, cin + 1, generator=g) / cin ** 0.5).to(dev)
    b = torch.randn(128, generator=g).to(dev)
    y = ops.BACKEND.linear_wide(x, wfull[:, :cin], b)
    ref = torch.nn.functional.linear(x.double(), wfull[:, :cin].double(), b.double())
    assert y is not None and (y.double() - ref).abs().max() < 1e-5
    assert ops.BACKEND.linear_wide(torch.randn(8, 128, device=dev), torch.randn(128, 128, device=dev), None) is None


@pytest.mark.parametrize("m,cin,cout,relu", [(1000, 3, 24, False), (77, 3, 24, True), (5, 8, 64, False), (300, 1, 4, True)])
def test_linear_lift_matches_torch(dev, m, cin, cout, relu):
    """tpu3_linear_lift_f32 (the 3 -> 24 lift of a Level) against torch in fp64, with and without the second
    copy of the rows into a channel slice of a wider buffer."""
    ops = pkg("network.operations")
    g = torch.Generator(device="cpu").manual_seed(m + cin)
    x = torch.randn(2, m, cin, generator=g).to(dev)
    w = torch.randn(cout, cin, generator=g).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    ref = torch.relu(ref) if relu else ref
    y = ops.BACKEND.linear_lift(x, w, b, relu)
    assert y is not None and (y.double() - ref).abs().max() < 1e-5
    wide = torch.full((2, m, cout + 40), -7.0, device=dev)
    y2 = ops.BACKEND.linear_lift(x, w, b, relu, also=wide[..., 40:])
    assert torch.equal(y2, y) and torch.equal(wide[..., 40:], y) and bool((wide[..., :40] == -7.0).all())
    assert ops.BACKEND.linear_lift(torch.randn(4, 9, device=dev), torch.randn(4, 9, device=dev), None, False) is None


@pytest.mark.parametrize("m,cin,cout,relu,off", [(1000, 84, 24, True, 180), (777, 204, 24, True, 60),
                                                (50, 144, 24, False, 120), (33, 16, 32, True, 0),
                                                (5, 8, 4, False, 0)])
def test_linear_small_matches_torch(dev, m, cin, cout, relu, off):
    """tpu3_linear_small_f32 (prep convolutions) against torch on the same rows (fp64 reference),
    reading a channel slice of a wider buffer in place."""
    ops = pkg("network.operations")
    g = torch.Generator(device="cpu").manual_seed(m + cin)
    buf = torch.randn(m, off + cin, generator=g).to(dev)
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    x = buf[:, off:]
    y = ops.BACKEND.linear_small(x, w, b, relu)
    assert y is not None and y.shape == (m, cout)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    ref = torch.relu(ref) if relu else ref
    assert (y.double() - ref).abs().max() < 1e-5
    # rows that are not 16-byte aligned are declined, not mis-computed
    wide = torch.randn(m, cin + 4, generator=g).to(dev)
    assert ops.BACKEND.linear_small(wide[:, 1:1 + cin], w, b, relu) is None


@pytest.mark.parametrize("m,cin,cout,relu", [(9984, 3, 24, False), (9984, 204, 24, True), (19968, 265, 128, True),
                                             (19968, 128, 64, True), (19968, 64, 3, False), (1111, 63, 17, True)])
def test_linear_wgrad_bias_and_pointwise_layer(dev, m, cin, cout, relu):
    """tpu3_linear_wgrad_bias_f32 (dW and db of any per-point layer of a Level, strided rows) against fp64; the
    autograd node built on it (network/layers.py _PointwiseLayer) against plain autograd of the same layer."""
    ops, layers = pkg("network.operations"), pkg("network.layers")
    g = torch.Generator(device="cpu").manual_seed(m + cin)
    x = torch.randn(m, cin + 4, generator=g).to(dev)[:, :cin]
    dy = torch.randn(m, cout + 1, generator=g).to(dev)[:, :cout]
    dw, db = ops.BACKEND.linear_wgrad_bias(x, dy)
    ref = dy.double().t() @ x.double()
    assert dw.shape == (cout, cin) and db.shape == (cout,)
    assert ((dw.double() - ref).abs() / (ref.abs() + m ** 0.5)).max() < 1e-5
    assert ((db.double() - dy.double().sum(0)).abs() / m ** 0.5).max() < 1e-5
    dw2, db2 = ops.BACKEND.linear_wgrad_bias(x, dy)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)                    # deterministic
    layer = layers.Conv1d(cin, cout, 1, activation="relu" if relu else None).to(dev)
    xa = x.contiguous().requires_grad_(True)
    xb = x.contiguous().requires_grad_(True)
    ya = layers.pointwise_train(layer, xa)
    assert ya is not None
    (ya * dy).sum().backward()
    got = (xa.grad, layer.conv.weight.grad.clone(), layer.conv.bias.grad.clone())
    layer.zero_grad()
    yb = torch.nn.functional.linear(xb, layer.conv.weight.view(cout, cin), layer.conv.bias)
    yb = torch.relu(yb) if relu else yb
    (yb * dy).sum().backward()
    assert (ya - yb).abs().max() < 1e-5
    assert (got[0] - xb.grad).abs().max() < 1e-4
    assert ((got[1].view(cout, cin) - layer.conv.weight.grad.view(cout, cin)).abs() / m ** 0.5).max() < 1e-4
    assert ((got[2] - layer.conv.bias.grad).abs() / m ** 0.5).max() < 1e-4


@pytest.mark.parametrize("m,cin,cout", [(319488, 48, 12), (20000, 36, 12), (70001, 64, 16), (100, 5, 3)])
def test_linear_wgrad_matches_torch(dev, m, cin, cout):
    """tpu3_linear_wgrad_f32 (dW = dy^T x of a skinny layer over many rows) against fp64, reading
    strided rows; and the autograd Function built on it against plain F.linear."""
    ops, layers = pkg("network.operations"), pkg("network.layers")
    g = torch.Generator(device="cpu").manual_seed(m)
    x = torch.randn(m, cin + 4, generator=g).to(dev)[:, :cin]
    dy = torch.randn(m, cout, generator=g).to(dev)
    dw = ops.BACKEND.linear_wgrad(x, dy)
    ref = dy.double().t() @ x.double()
    assert dw.shape == (cout, cin)
    assert ((dw.double() - ref).abs() / (ref.abs() + m ** 0.5)).max() < 1e-5
    assert torch.equal(dw, ops.BACKEND.linear_wgrad(x, dy))                 # deterministic
    if m >= 16384:
        conv = torch.nn.Conv2d(cin, cout, 1).to(dev)
        xc = x.contiguous().requires_grad_(True)
        y = layers.linear_1x1(conv, xc)
        (y * dy).sum().backward()
        gw, gb, gx = conv.weight.grad.clone(), conv.bias.grad.clone(), xc.grad.clone()
        conv.zero_grad()
        xr = x.contiguous().requires_grad_(True)
        yr = torch.nn.functional.linear(xr, conv.weight.view(cout, cin), conv.bias)
        (yr * dy).sum().backward()
        assert torch.allclose(y, yr) and torch.allclose(gx, xr.grad)
        assert ((gw.view(cout, cin) - conv.weight.grad.view(cout, cin)).abs().max()
                <= 1e-5 * (1 + conv.weight.grad.abs().max()))
        assert torch.allclose(gb, conv.bias.grad, rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("m,r", [(312, 2), (1000, 2), (17, 4), (4096 * 3 + 5, 1), (40000, 3)])
def test_regress_tail_matches_torch(dev, m, r, regressor_form):
    """tpu3_regress_tail_f32 (both arithmetic forms) against the unfused torch formulation (fp64 reference)."""
    ops = pkg("network.operations")
    g = torch.Generator(device="cpu").manual_seed(m)
    a = torch.randn(m, 128, generator=g).to(dev)
    c = torch.randn(r, 128, generator=g).to(dev)
    w2 = (torch.randn(128, 128, generator=g) / 128 ** 0.5).to(dev)
    w3 = (torch.randn(64, 128, generator=g) / 128 ** 0.5).to(dev)
    w4 = (torch.randn(3, 64, generator=g) / 8).to(dev)
    b2, b3, b4 = (torch.randn(n, generator=g).to(dev) for n in (128, 64, 3))
    res = torch.randn(m, 3, generator=g).to(dev)
    out = ops.BACKEND.regress_tail(a, c, w2, b2, w3, b3, w4, b4, res)
    d = lambda t: t.double()
    h = torch.relu(d(a).unsqueeze(1) + d(c).unsqueeze(0))                       # (m,r,128)
    h = torch.relu(h @ d(w2).t() + d(b2))
    h = torch.relu(h @ d(w3).t() + d(b3))
    ref = (h @ d(w4).t() + d(b4) + d(res).unsqueeze(1)).reshape(m * r, 3)
    assert out.shape == (m * r, 3)
    assert (out.double() - ref).abs().max() < 2e-5


def test_split_bf16_forms_are_as_accurate_as_fp32(dev):
    """The split-bf16 form of the two regressor kernels against their fp32 form on inputs with a wide dynamic range
    (rows scaled over four decades, zero channels, weights of mixed magnitude): each within 1.5x the fp32 kernel's own
    error against fp64 (+ 1e-6), the two within 1e-5 of each other relative to the row's magnitude -- and not
    bit-identical (the switch really selects another kernel)."""
    ops = pkg("network.operations")
    be = ops.BACKEND
    g = torch.Generator(device="cpu").manual_seed(11)
    m = 20000
    scale = 10.0 ** (torch.rand(m, 1, generator=g) * 4 - 2)
    x = (torch.randn(m, 264, generator=g) * scale).to(dev)
    x[:, 40:60] = 0
    w = (torch.randn(128, 265, generator=g) * 10.0 ** (torch.rand(128, 1, generator=g) * 2 - 1.5) / 16).to(dev)
    b = torch.randn(128, generator=g).to(dev)
    a_in = torch.randn(m, 128, generator=g).to(dev) * 3
    c = torch.randn(2, 128, generator=g).to(dev)
    w2 = (torch.randn(128, 128, generator=g) / 11).to(dev)
    w3 = (torch.randn(64, 128, generator=g) / 11).to(dev)
    w4 = (torch.randn(3, 64, generator=g) / 8).to(dev)
    b2, b3, b4 = (torch.randn(n, generator=g).to(dev) for n in (128, 64, 3))
    res = torch.randn(m, 3, generator=g).to(dev)
    d = lambda t: t.double()
    ref_wide = d(x) @ d(w[:, :264]).t() + d(b)
    h = torch.relu(d(a_in).unsqueeze(1) + d(c).unsqueeze(0))
    h = torch.relu(h @ d(w2).t() + d(b2))
    h = torch.relu(h @ d(w3).t() + d(b3))
    ref_tail = (h @ d(w4).t() + d(b4) + d(res).unsqueeze(1)).reshape(m * 2, 3)
    old = be.split_bf16()
    try:
        out = {}
        for form in (False, True):
            be.split_bf16(form)
            out[form] = (be.linear_wide(x, w[:, :264], b), be.regress_tail(a_in, c, w2, b2, w3, b3, w4, b4, res))
    finally:
        be.split_bf16(old)
    for k, ref, name in ((0, ref_wide, "linear_wide"), (1, ref_tail, "regress_tail")):
        e32 = float((d(out[False][k]) - ref).abs().max())
        esb = float((d(out[True][k]) - ref).abs().max())
        rel = float(((out[True][k] - out[False][k]).abs().max(1)[0] / (ref.abs().max(1)[0] + 1e-3)).max())
        print("%s: max |err| vs fp64: fp32 form %.3e, split-bf16 form %.3e; forms differ by %.2e of the row's largest output"
              % (name, e32, esb, rel))
        assert esb <= 1.5 * e32 + 1e-6, (name, esb, e32)
        assert rel < 1e-5, (name, rel)
        assert not torch.equal(out[True][k], out[False][k]), name


@pytest.mark.parametrize("m,cin,cout,relu", [(5000, 84, 24, True), (777, 204, 24, True), (4099, 144, 24, False),
                                             (100, 16, 8, True),
                                             # the per-point half of up_layer1 (264 -> 128): linear_wide_f16_kernel
                                             (4099, 264, 128, False), (312, 264, 128, True), (50, 260, 64, False)])
def test_linear_fp16_mfma_within_derived_bound(dev, m, cin, cout, relu):
    """mfma = F16 flavour of the per-point layers (prep convolutions, and -- cout > 32 -- up_layer1's per-point
    half, a library fp16 GEMM until round 2): every output within the bound derived from the operand roundings."""
    ops, L = pkg("network.operations"), pkg("_lib")
    g = torch.Generator(device="cpu").manual_seed(m + cin)
    x = torch.randn(m, cin, generator=g).to(dev)
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    y = ops.BACKEND.linear_small(x, w, b, relu, mfma=L.MFMA_F16)
    assert y is not None and tuple(y.shape) == (m, cout)
    xd, wd = x.double().cpu(), w.double().cpu()
    ref = xd @ wd.t() + b.double().cpu()
    ref = torch.relu(ref) if relu else ref
    bound = _f16_layer_bound(xd.abs(), torch.zeros_like(xd), wd) + 2.0 ** -23 * ref.abs() + 1e-7
    err = (y.double().cpu() - ref).abs()
    ratio = float((err / bound).max())
    print("linear f16 %d x %d -> %d: max |err| %.2e, max err/bound %.3f" % (m, cin, cout, float(err.max()), ratio))
    assert ratio <= 1.0, ratio
    if cout <= 32:
        y32 = ops.BACKEND.linear_small(x, w, b, relu)
        assert float((y - y32).abs().max()) > 0            # it is another arithmetic than the fp32 flavour
    else:
        assert ops.BACKEND.linear_small(x, w, b, relu) is None      # fp32 wide layers: tpu3_linear_wide_f32


@pytest.mark.parametrize("m,r", [(312, 2), (4096 * 3 + 5, 2), (17, 4)])
def test_regress_tail_fp16_mfma_within_derived_bound(dev, m, r):
    """mfma = F16 flavour of the regressor tail against fp64: relu(a_i + c_j) -> 128 -> 64 -> 3 + residual, three
    fp16-operand layers; the bound is propagated layer by layer."""
    ops, L = pkg("network.operations"), pkg("_lib")
    g = torch.Generator(device="cpu").manual_seed(m)
    a = torch.randn(m, 128, generator=g).to(dev)
    c = torch.randn(r, 128, generator=g).to(dev)
    w2 = (torch.randn(128, 128, generator=g) / 128 ** 0.5).to(dev)
    w3 = (torch.randn(64, 128, generator=g) / 128 ** 0.5).to(dev)
    w4 = (torch.randn(3, 64, generator=g) / 8).to(dev)
    b2, b3, b4 = (torch.randn(n, generator=g).to(dev) for n in (128, 64, 3))
    res = torch.randn(m, 3, generator=g).to(dev)
    out = ops.BACKEND.regress_tail(a, c, w2, b2, w3, b3, w4, b4, res, mfma=L.MFMA_F16)
    d = lambda t: t.double().cpu()
    h = torch.relu(d(a).unsqueeze(1) + d(c).unsqueeze(0))
    e = 2.0 ** -23 * h
    for w, b, last in ((w2, b2, False), (w3, b3, False), (w4, b4, True)):
        e = _f16_layer_bound(h, e, d(w))
        h = h @ d(w).t() + d(b)
        e = e + 2.0 ** -23 * h.abs()
        if not last:
            h = torch.relu(h)
    ref = (h + d(res).unsqueeze(1)).reshape(m * r, 3)
    bound = (e + 2.0 ** -23 * ref.reshape(m, r, 3).abs()).reshape(m * r, 3) + 1e-6
    err = (out.double().cpu() - ref).abs()
    ratio = float((err / bound).max())
    print("regress tail f16 m=%d r=%d: max |err| %.2e, max err/bound %.3f" % (m, r, float(err.max()), ratio))
    assert ratio <= 1.0, ratio


def test_fp16_feature_buffer_kernels(dev):
    """Activation storage "f16" (Net.set_mlp_precision("f16", activations="f16")), kernel by kernel:
    * the fp16-operand per-point layers reading fp16 rows give BIT FOR BIT what they give on the same values held
      in fp32 rows (the number that enters the matrix instruction is the stored one) -- prep convolutions and the
      264 -> 128 half of up_layer1, reading a channel slice of a (m,264) buffer;
    * DenseEdgeConv writing into an fp16 buffer = its fp32 rows rounded to fp16;
    * the skip connection on fp16 rows = the fp32 kernel on the widened rows, rounded on the way back (its
      arithmetic is the same fp32 code)."""
    ops, L, layers = pkg("network.operations"), pkg("_lib"), pkg("network.layers")
    g = torch.Generator(device="cpu").manual_seed(3)
    m = 3001
    buf16 = torch.randn(m, 264, generator=g).to(dev).half()
    buf32 = buf16.float()
    for lo, cin, cout in ((180, 84, 24), (60, 204, 24), (0, 264, 128)):
        w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(dev)
        b = torch.randn(cout, generator=g).to(dev)
        y16 = ops.BACKEND.linear_small(buf16[:, lo:lo + cin], w, b, True, mfma=L.MFMA_F16)
        y32 = ops.BACKEND.linear_small(buf32[:, lo:lo + cin], w, b, True, mfma=L.MFMA_F16)
        assert y16 is not None and y16.dtype == torch.float32 and torch.equal(y16, y32)
    assert ops.BACKEND.linear_small(buf16[:, :84], w[:24, :84].contiguous(), b[:24], True) is None     # fp32 flavour: no
    # DenseEdgeConv into a slice of an fp16 buffer
    P, N = 5, 312
    x = torch.randn(P, N, 24, generator=g).to(dev)
    conv = layers.DenseEdgeConv(24, growth_rate=12, n=3, k=32).to(dev)
    conv.mlp_precision = "f16"
    with torch.no_grad():
        o32 = torch.zeros(P, N, 264, device=dev)
        o16 = torch.zeros(P, N, 264, device=dev, dtype=torch.float16)
        _, idx = conv.forward_cl(x, out=o32[..., 144:204])
        conv.forward_cl(x, out=o16[..., 144:204])
    assert torch.equal(o16[..., 144:204], o32[..., 144:204].half())
    assert float(o16[..., :144].abs().max()) == 0 and float(o16[..., 204:].abs().max()) == 0
    # skip connection on fp16 rows
    B, M, C, K = 3, 400, 264, 5
    xyz = torch.randn(B, N, 3, generator=g).to(dev)
    pxyz = torch.randn(B, M, 3, generator=g).to(dev)
    f16 = torch.randn(B, N, C, generator=g).to(dev).half()
    pf16 = (0.5 * torch.randn(B, M, C, generator=g)).to(dev).half()
    idx5, _, _ = ops.knn_query(K, xyz, pxyz, unique=True, want_dist=False, want_grouped=False)
    want = ops.BACKEND.interlevel_skip(xyz, f16.float().contiguous(), pxyz, pf16.float().contiguous(), None, idx5).half()
    got = ops.BACKEND.interlevel_skip(xyz, f16.clone(), pxyz, pf16, None, idx5)
    assert got.dtype == torch.float16 and torch.equal(got, want)


def test_level_with_fp16_feature_buffers(dev):
    """A Level with fp16 operands, feature buffers stored as fp16 against the same Level with fp32 buffers: the
    matrix kernels see identical operands; what differs is (i) the rows the skip connection computes on and (ii)
    the rounding of the row it writes back -- relative 2^-11 each on values of order 1, through a regressor whose
    layers have gain of order 1: the level's output coordinates agree to 4e-4 of the patch radius (measured, printed
    below), inside the fp16-operand error itself."""
    ups, ops = pkg("network.upsampler"), pkg("network.operations")
    net = _net(dev)
    g = golden("level_forward.npz")
    lvl1, lvl2 = net.levels["level_1"], net.levels["level_2"]
    patch = torch.from_numpy(g["patch"]).to(dev).transpose(2, 1).contiguous()
    p3 = torch.from_numpy(g["l2_in"]).to(dev).transpose(2, 1).contiguous()
    p3n = torch.from_numpy(g["l2_in_norm"]).to(dev).transpose(2, 1).contiguous()
    outs = {}
    with torch.no_grad():
        for act in ("f32", "f16"):
            net.set_mlp_precision("f16", activations=act)
            x1, f1 = lvl1.forward_cl(patch, patch, None)
            assert f1.dtype == (torch.float16 if act == "f16" else torch.float32)
            x2, f2 = lvl2.forward_cl(p3, p3n, (patch, f1, None), owner=torch.zeros(3, dtype=torch.int32, device=dev),
                                     groups=1)
            outs[act] = (x1.float(), f1.float(), x2.float(), f2.float())
    net.set_mlp_precision("f32")
    a, b = outs["f32"], outs["f16"]
    d1, df1 = float((a[0] - b[0]).abs().max()), float((a[1] - b[1]).abs().max() / a[1].abs().max())
    d2, df2 = float((a[2] - b[2]).abs().max()), float((a[3] - b[3]).abs().max() / a[3].abs().max())
    print("fp16 buffers vs fp32 buffers: level 1 xyz %.2e, features %.2e rel; level 2 xyz %.2e, features %.2e rel"
          % (d1, df1, d2, df2))
    assert df1 <= 2.0 ** -11 and d1 == 0.0            # level 1: no skip -- the rows are only rounded once, at the store
    assert d2 < 1e-3 and df2 < 1.5e-3                 # measured 4.2e-4 / 6.9e-4


def test_config_c5_full_size_fp16_mlps(orc, dev):
    """BASELINE config C5 at FULL size: one 80 000-point cloud, num_point = 1024 (234 outer patches of 1024
    points, inner patches of 312), up_ratio 16 -> 1.28 M points, feature MLPs on fp16-operand MFMA
    (Net.set_mlp_precision("f16"); FPS / kNN / Chamfer stay fp32).  Size-independent properties:
    shape, finiteness, no generic (unfused) path, no point sampled twice, the FPS covering-radius property on
    the 3.83 M -> 1.28 M resampling, and the fp16 cloud against the fp32 cloud of the same weights under
    Chamfer (both are samplings of the same surface: far below the output's own point spacing)."""
    pipe, ops, ups = pkg("pipeline"), pkg("network.operations"), pkg("network.upsampler")
    ml = pkg("network.model_loss")
    net = _net(dev)
    g = torch.Generator().manual_seed(0)
    cand = torch.randn(1, 80000, 3, generator=g)
    cloud = (cand / cand.norm(dim=2, keepdim=True)).transpose(2, 1).contiguous().to(dev)
    P = pipe.num_outer_patches(80000, 1024, 3)
    assert P == 234
    ops.GENERIC_PATH_EVENTS.clear()
    merged32 = pipe.upsample(net, cloud, 1024, 16, 3, final_fps=False)
    net.set_mlp_precision("f16", activations="f16")        # fp16 operands, feature buffers stored as fp16
    merged16 = pipe.upsample(net, cloud, 1024, 16, 3, final_fps=False)
    out16 = pipe.upsample(net, cloud, 1024, 16, 3)
    net.set_mlp_precision("f32")
    assert not ops.GENERIC_PATH_EVENTS, dict(ops.GENERIC_PATH_EVENTS)
    assert tuple(merged16.shape) == (1, 3, P * 1024 * 16) and tuple(out16.shape) == (1, 3, 1280000)
    assert torch.isfinite(merged16).all() and torch.isfinite(out16).all()
    assert float((merged16 - merged32).abs().max()) > 0            # another arithmetic ...
    o = out16.transpose(2, 1).contiguous()
    m16, m32 = merged16.transpose(2, 1).contiguous(), merged32.transpose(2, 1).contiguous()
    # ... describing the same surface: Chamfer(fp16 cloud, fp32 cloud) against the fp32 cloud's own spacing
    sub16, sub32 = m16[:, ::16].contiguous(), m32[:, ::16].contiguous()
    d1, _, d2, _ = ml.nndistance(sub16, sub32)
    _, dself, _ = ops.knn_query(2, sub32[:, :20000].contiguous(), sub32[:, :20000].contiguous(), unique=False,
                                want_grouped=False)
    spacing2 = float(dself[:, :, 1].clamp_min(0).median())
    print("C5 fp16 vs fp32 merged clouds: mean sq NN distance %.3e / %.3e, own squared spacing %.3e"
          % (float(d1.mean()), float(d2.mean()), spacing2))
    assert float(d1.mean()) < 4 * spacing2 and float(d2.mean()) < 4 * spacing2
    # FPS properties of the 3.83 M -> 1.28 M resampling
    idx = ops.fps(m16, 1280000)
    assert idx.unique().numel() == 1280000
    assert torch.equal(torch.gather(m16, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)), o)
    ref_idx, _ = orc.fps(m16.cpu().numpy(), 40)
    np.testing.assert_array_equal(idx[:, :40].cpu().numpy(), ref_idx)
    d_cover, _, _, _ = ml.nndistance(m16, o)
    part = o[:, :200000].contiguous()
    _, d_self, _ = ops.knn_query(2, part, o, unique=False, want_grouped=False)
    assert float(d_self[:, :, 1].min()) >= float(d_cover.max()) - 1e-6


def test_cli_test_and_train_phases(dev, tmp_path, monkeypatch):
    """The drop-in CLI on the device: --phase test on .xyz files (checkpoint in the reference's
    format) writes <name>.ply with N*up_ratio points; --phase train runs optimiser steps."""
    main, pu, ups, pcu = pkg("main"), pkg("utils.pytorch_utils"), pkg("network.upsampler"), pkg("utils.pc_utils")
    torch.manual_seed(0)
    net = ups.Net(max_up_ratio=4, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5)
    ckpt = pu.save_network(net, str(tmp_path), "model", epoch_label="0", step="0")
    os.makedirs(os.path.join(str(tmp_path), "data"))
    for i in range(2):
        np.savetxt(os.path.join(str(tmp_path), "data", "cloud%d.xyz" % i), sphere(i, 900)[0] * 2.0 + 0.5)
    out = os.path.join(str(tmp_path), "out")
    main.main(["--phase", "test", "--ckpt", ckpt, "--num_point", "312", "--num_shape_point", "1000",
               "--up_ratio", "4", "--test_data", os.path.join(str(tmp_path), "data", "*.xyz"),
               "--result_dir", out])
    for i in range(2):
        up = pcu.load(os.path.join(out, "data", "cloud%d.ply" % i))
        src = pcu.load(os.path.join(out, "data", "cloud%d_input.ply" % i))
        assert up.shape == (4000, 3) and src.shape == (1000, 3) and np.isfinite(up).all()
        assert abs(up.mean(0) - src.mean(0)).max() < 1.0          # same frame (de-normalised; radius 2)
    monkeypatch.setenv("TPU3_STEPS_PER_EPOCH", "2")
    main.main(["--phase", "train", "--h5_data", "synthetic", "--num_point", "312", "--up_ratio", "4",
               "--batch_size", "4", "--max_epoch", "2", "--stage_steps", "1", "--log_dir", str(tmp_path)])
    # the data set path: a small file in the reference's layout, resident on the device
    h5 = pkg("data").write_synthetic(str(tmp_path), num_shapes=2, points=(1000, 2000, 4000))
    main.main(["--phase", "train", "--h5_data", h5, "--num_point", "312", "--num_shape_point", "1000",
               "--up_ratio", "4", "--batch_size", "4", "--max_epoch", "2", "--stage_steps", "1",
               "--log_dir", str(tmp_path), "--id", "h5run"])


def test_data_path_on_device_matches_reference_items(dev):
    """data.H5Dataset on the device (patches through tpu3_knn_f32, incl. the sort kernel for
    k = 256) against the items the reference's data.py produced (tests/golden/data_path.npz)."""
    data = pkg("data")
    g = golden("data_path.npz")
    store = {k[len("store_"):]: g[k] for k in g.files if k.startswith("store_")}
    ds = data.H5Dataset(str(g["file_name"]), num_shape_point=312, num_patch_point=64, up_ratio=4, step_ratio=2,
                        batch_size=4, store=store, device=dev)
    assert ds.input_array.is_cuda
    for i in range(6):
        if i == 4:
            ds.unset_combined()
            ds.set_max_ratio(2)
        a, b, r = ds[i]
        assert a.is_cuda and r == int(g["item%d_ratio" % i])
        np.testing.assert_allclose(a.cpu().numpy(), g["item%d_input" % i], atol=1e-6, rtol=0)
        np.testing.assert_allclose(b.cpu().numpy(), g["item%d_label" % i], atol=1e-6, rtol=0)


# ---- the unused variants of the reference on the device -------------------------
def _load_prefixed(module, g, prefix, dev):
    module.load_state_dict({k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)},
                           strict=True)
    return module.to(dev).eval()


@pytest.mark.parametrize("nsample", [48, 1])
def test_sampled_dense_edge_conv_on_device(dev, nsample):
    layers = pkg("network.layers")
    g = golden("adaptive_level.npz")
    conv = _load_prefixed(layers.SampledDenseEdgeConv(24, growth_rate=12, n=3, k=8), g, "sdec_state_", dev)
    with torch.no_grad():
        y, sxyz, sidx = conv(torch.from_numpy(g["sdec_x"]).to(dev), nsample, torch.from_numpy(g["sdec_xyz"]).to(dev))
    np.testing.assert_array_equal(sidx.cpu().numpy().astype(np.int32), g["sdec_sidx_%d" % nsample])
    np.testing.assert_array_equal(sxyz.cpu().numpy(), g["sdec_sxyz_%d" % nsample])
    np.testing.assert_allclose(y.cpu().numpy(), g["sdec_y_%d" % nsample], rtol=0, atol=1e-5)


def test_adaptive_level_on_device(dev):
    ups = pkg("network.upsampler")
    g = golden("adaptive_level.npz")
    lvl = _load_prefixed(ups.AdaptiveLevel(dense_n=3, growth_rate=12, knn=8, fm_knn=5), g, "alevel_state_", dev)
    with torch.no_grad():
        x, glob = lvl(torch.from_numpy(g["alevel_in"]).to(dev), int(g["alevel_target"]))
    np.testing.assert_allclose(glob.cpu().numpy(), g["alevel_global"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(x.cpu().numpy(), g["alevel_xyz"], rtol=0, atol=1e-5)
    with torch.no_grad(), pytest.raises((AssertionError, RuntimeError)):
        ups.AdaptiveLevel(knn=16).to(dev).eval()(torch.from_numpy(g["alevel_in"]).to(dev), 100)


def test_group_ball_on_device(orc, dev):
    ops = pkg("network.operations")
    pts = sphere(31, 5000, 4)
    q = np.ascontiguousarray(pts[:, ::16][:, :312])
    grouped, idx = ops.group_ball(0.1, 32, torch.from_numpy(q).to(dev), torch.from_numpy(pts).to(dev), NCHW=False)
    ref = orc.ball_query(q, pts, 0.1, 32)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    np.testing.assert_array_equal(grouped.cpu().numpy(), np.stack([pts[b][ref[b]] for b in range(4)]))
    # differentiable with respect to the points: every slot passes its gradient to the point it holds
    p = torch.from_numpy(pts).to(dev).requires_grad_(True)
    gr, ix = ops.group_ball(0.1, 32, torch.from_numpy(q).to(dev), p, NCHW=False)
    gr.sum().backward()
    counts = np.zeros((4, 5000), np.float32)
    for b in range(4):
        np.add.at(counts[b], ref[b].reshape(-1), 1.0)
    np.testing.assert_array_equal(p.grad.cpu().numpy(), np.repeat(counts[:, :, None], 3, axis=2))
