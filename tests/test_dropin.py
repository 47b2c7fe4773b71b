"""The drop-in boundary used the way the reference uses it: bare `import sampling`, `import losses`
(network/operations.py:6, network/model_loss.py:2) resolved by install_dropin() to the compiled
pybind11 extension modules, then the reference's exact call sequences replayed through them."""
import sys

import numpy as np
import pytest
import torch

from conftest import pkg, sphere

BARE = ("sampling", "losses", "network", "network.operations", "network.layers", "network.upsampler",
        "network.model_loss")


@pytest.fixture()
def dropin():
    saved = {k: sys.modules.get(k) for k in BARE}
    pkg("build").build_dropin()
    mods = pkg().install_dropin()
    yield mods
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def test_install_dropin_registers_compiled_modules(dropin):
    import losses
    import sampling
    from network import operations, model_loss                   # noqa: F401
    assert sampling.__file__.endswith(".so") and losses.__file__.endswith(".so")       # compiled, not the mirrors
    for fn in ("furthest_sampling", "gather_forward", "gather_backward", "ball_query"):      # sampling.cpp:85-88
        assert callable(getattr(sampling, fn))
    for fn in ("nmdistance_forward", "nmdistance_backward"):                                   # nmdistance.cpp:25-26
        assert callable(getattr(losses, fn))
    assert operations is pkg("network.operations")
    # the reference's checks: CPU / non-contiguous tensors raise RuntimeError (sampling.cpp:20-24)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        sampling.furthest_sampling(1, 4, 2, torch.zeros(1, 4, 3), torch.zeros(1, 4), torch.zeros(1, 2, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        sampling.ball_query(torch.zeros(1, 2, 3), torch.zeros(1, 4, 3), 0.1, 4)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        losses.nmdistance_forward(torch.zeros(1, 2, 3), torch.zeros(1, 2, 3), torch.zeros(1, 2), torch.zeros(1, 2),
                                  torch.zeros(1, 2, dtype=torch.int32), torch.zeros(1, 2, dtype=torch.int32))
    # positional signatures only, like the pybind11 modules of the reference
    with pytest.raises(TypeError):
        sampling.furthest_sampling(1, 4, 2)


def test_mirrors_are_used_when_not_compiled():
    saved = {k: sys.modules.get(k) for k in BARE}
    try:
        mods = pkg().install_dropin(compiled=False)
        assert mods["sampling"] is pkg("sampling") and mods["losses"] is pkg("losses")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


@pytest.mark.gpu
def test_reference_call_sequences_through_bare_imports(orc, dev, dropin):
    """operations.py:288-295 (FurthestPointSampling... `sampling.furthest_sampling`), :241-245 and
    :257-261 (GatherFunction forward / backward), model_loss.py:8-18,25-27 (NmDistanceFunction), written
    as the reference writes them: caller-allocated outputs, positional arguments, bare module names."""
    import losses
    import sampling
    # --- operations.py:288-295
    xyz = torch.from_numpy(sphere(3, 5000)).to(dev)                     # (B,N,3) contiguous
    B, N, _ = xyz.size()
    npoint = 48
    idx = torch.empty([B, npoint], dtype=torch.int32, device=xyz.device)
    temp = torch.full([B, N], 1e10, dtype=torch.float32, device=xyz.device)
    ret = sampling.furthest_sampling(B, N, npoint, xyz, temp, idx)
    assert ret.data_ptr() == idx.data_ptr()                              # returns an alias of idx
    ref_idx, ref_temp = orc.fps(xyz.cpu().numpy(), npoint)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref_idx)
    np.testing.assert_array_equal(temp.cpu().numpy(), ref_temp)
    # --- operations.py:241-245
    features = xyz.transpose(2, 1).contiguous()                          # (B,C,N)
    _, C, _ = features.size()
    output = torch.empty([B, C, npoint], dtype=features.dtype, device=features.device)
    output = sampling.gather_forward(B, C, N, npoint, features, idx, output)
    np.testing.assert_array_equal(output.cpu().numpy(), orc.gather_fwd(features.cpu().numpy(), ref_idx))
    # --- operations.py:257-261
    grad_out = torch.ones_like(output)
    grad_features = torch.zeros(B, C, N, dtype=grad_out.dtype, device=grad_out.device)
    grad_features = sampling.gather_backward(B, C, N, npoint, grad_out.contiguous(), idx, grad_features)
    np.testing.assert_array_equal(grad_features.cpu().numpy(),
                                  orc.gather_bwd(grad_out.cpu().numpy(), ref_idx, N))
    # --- sampling.cpp:59-81 (no caller in the reference)
    q = xyz[:, :100].contiguous()
    bq = sampling.ball_query(q, xyz, 0.1, 16)
    np.testing.assert_array_equal(bq.cpu().numpy(), orc.ball_query(q.cpu().numpy(), xyz.cpu().numpy(), 0.1, 16))
    # --- model_loss.py:8-18
    xyz1 = torch.from_numpy(sphere(4, 624, 2)).to(dev)
    xyz2 = torch.from_numpy(sphere(5, 700, 2) * np.float32(1.02)).to(dev)
    Bn, Nn, _ = xyz1.size()
    Bn, Mn, _ = xyz2.size()
    result = torch.empty(Bn, Nn, dtype=xyz1.dtype, device=xyz1.device)
    result_i = torch.empty(Bn, Nn, dtype=torch.int32, device=xyz1.device)
    result2 = torch.empty(Bn, Mn, dtype=xyz2.dtype, device=xyz2.device)
    result2_i = torch.empty(Bn, Mn, dtype=torch.int32, device=xyz1.device)
    assert losses.nmdistance_forward(xyz1, xyz2, result, result2, result_i, result2_i) == 1
    d1, i1, d2, i2 = orc.nmdistance_fwd(xyz1.cpu().numpy(), xyz2.cpu().numpy())
    np.testing.assert_array_equal(result.cpu().numpy(), d1)
    np.testing.assert_array_equal(result_i.cpu().numpy(), i1)
    np.testing.assert_array_equal(result2.cpu().numpy(), d2)
    np.testing.assert_array_equal(result2_i.cpu().numpy(), i2)
    # --- model_loss.py:25-27
    gradxyz1 = torch.zeros_like(xyz1)
    gradxyz2 = torch.zeros_like(xyz2)
    g1 = torch.ones_like(result)
    g2 = torch.ones_like(result2)
    assert losses.nmdistance_backward(xyz1, xyz2, gradxyz1, gradxyz2, g1, g2, result_i, result2_i) == 1
    r1, r2 = orc.nmdistance_bwd(xyz1.cpu().numpy(), xyz2.cpu().numpy(), g1.cpu().numpy(), g2.cpu().numpy(), i1, i2)
    np.testing.assert_allclose(gradxyz1.cpu().numpy(), r1, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gradxyz2.cpu().numpy(), r2, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_product_path_runs_on_the_compiled_modules(orc, dev, dropin, monkeypatch):
    """network.operations / model_loss call `sampling.*` / `losses.*` through their module globals:
    pointing those at the compiled extension modules must give the same bits as the ctypes mirrors."""
    ops, ml = pkg("network.operations"), pkg("network.model_loss")
    x = torch.from_numpy(sphere(7, 2000, 2)).to(dev)
    idx_a = ops.fps(x, 100)
    d_a = ml.nndistance(x, x.flip(1).contiguous())
    monkeypatch.setattr(ops, "sampling", dropin["sampling"])
    monkeypatch.setattr(ml, "losses", dropin["losses"])
    idx_b = ops.fps(x, 100)
    d_b = ml.nndistance(x, x.flip(1).contiguous())
    assert torch.equal(idx_a, idx_b)
    for a, b in zip(d_a, d_b):
        assert torch.equal(a, b)
