"""not-gpu: CHAINED replay of the reference's discrete choices through all four levels.

tests/golden/c2_chain.npz holds, for four outer patches of the C2 cloud, every discrete choice the reference's
Net.forward took from the 312-point patch to its 4992 output points, and the cloud it held after every level
(oracle/make_golden.py `make_chain_golden`).  Here the product's Net runs on the CPU stand-in backend:
  * with the choices REPLAYED (tests/chain_replay.py) nothing discrete is left to differ, so the final 4992 points of
    every patch -- and the cloud after every level -- must be within 1e-5 of the reference's: end-to-end "upsampled
    xyz within 1e-5" wherever the discrete choices agree;
  * on its own, the first choice in which the build departs from the reference is NAMED per patch, and every level
    before that choice must still be within 1e-5: no patch may drift without a named flip.
The -m gpu twin (tests/test_c2_parity.py) does the same on the HIP path."""
import numpy as np
import pytest
import torch

from chain_replay import ORDER, first_flip, run_chain
from conftest import golden, pkg
from oracle.backend import OracleBackend

TOL = 1e-5


@pytest.fixture()
def modules(orc, monkeypatch):
    ops, ups = pkg("network.operations"), pkg("network.upsampler")
    monkeypatch.setattr(ops, "BACKEND", OracleBackend())
    net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5)
    state = golden("net16_state.npz")
    net.load_state_dict({k: torch.from_numpy(state[k]) for k in state.files if k != "meta"}, strict=True)
    return ops, net.eval()


def level_errors(g, ids, levels, x16):
    """max |difference| per outer patch and level (the last level is the net's output itself)"""
    err = np.zeros((len(ids), 4))
    for i, q in enumerate(ids):
        for l in (1, 2, 3, 4):
            ref = g["p%d_l%d_out" % (q, l)]                                   # (3, n_l)
            mine = levels[l - 1][i].T if l < 4 else x16[i]
            err[i, l - 1] = np.abs(mine - ref).max()
    return err


def test_chain_fixture_is_the_c2_run():
    """The chain fixture IS the c2_x16.npz run: same outer patches, and its level clouds / final points, put back
    into the cloud's frame (main.py:242), are that fixture's lv1..lv3 / pred_concat."""
    g, c2 = golden("c2_chain.npz"), golden("c2_x16.npz")
    ids = [int(q) for q in g["patch_ids"]]
    assert ids == [0, 6, 16, 23]
    np.testing.assert_array_equal(g["outer_patch_idx"], c2["patch_idx"][0][ids])
    cloud = c2["cloud"][0]                                                    # (3, 5000)
    for i, q in enumerate(ids):
        pts = cloud[:, g["outer_patch_idx"][i].astype(np.int64)]             # (3, 312)
        centroid = pts.mean(axis=1, keepdims=True)
        radius = np.sqrt(((pts - centroid) ** 2).sum(axis=0)).max()
        np.testing.assert_allclose((pts - centroid) / radius, g["p%d_in" % q], rtol=0, atol=2e-6)
        for l, key in ((1, "lv1"), (2, "lv2"), (3, "lv3")):
            np.testing.assert_allclose(g["p%d_l%d_out" % (q, l)] * radius + centroid, c2[key][q], rtol=0, atol=5e-6)
        n4 = 4992
        np.testing.assert_allclose(g["p%d_x16" % q] * radius + centroid, c2["pred_concat"][0][:, q * n4:(q + 1) * n4],
                                   rtol=0, atol=5e-6)
    # patch 16 loses a point to the outlier filter at level 3: 19 inner patches instead of 20 -- the ragged case
    assert int(g["p16_l3_mask"].sum()) == 1247 and g["p16_l3_seeds"].shape[0] == 19


def test_chain_replayed_is_within_1e5_end_to_end(modules):
    ops, net = modules
    g = golden("c2_chain.npz")
    ids = [int(q) for q in g["patch_ids"]]
    chain, levels, x16 = run_chain(ops, net, g, ids, torch.device("cpu"), "replay")
    assert chain.graph_calls == 16 and chain.levels_closed == 3
    err = level_errors(g, ids, levels, x16)
    print("chained replay (CPU stand-in): max |dx| per outer patch %s and level 1..4:\n%s" % (ids, err))
    assert err.max() <= TOL, err


def test_chain_on_its_own_departs_only_at_named_flips(modules):
    ops, net = modules
    g = golden("c2_chain.npz")
    ids = [int(q) for q in g["patch_ids"]]
    chain, levels, x16 = run_chain(ops, net, g, ids, torch.device("cpu"), "record")
    assert sorted(chain.seen) == sorted(ORDER)
    err = level_errors(g, ids, levels, x16)
    for i, q in enumerate(ids):
        flip = first_flip(chain, g, i, q)
        upto = 4 if flip is None else int(flip[1]) - 1                      # levels before the flip's level
        print("outer patch %2d: first choice that differs from the reference's: %-10s max |dx| per level %s"
              % (q, flip, " ".join("%.1e" % e for e in err[i])))
        assert (err[i, :upto] <= TOL).all(), (q, flip, err[i])
        if flip is None:
            assert err[i, 3] <= TOL
