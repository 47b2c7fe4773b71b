"""not-gpu: the compact chained-replay record of ALL 48 outer patches of the C2 cloud (and 16 of a second cloud),
tests/golden/c2_chain_all.npz / c2_chain_all_seed1.npz (oracle/make_golden.py `make_chain_all_golden`).

The record holds every discrete choice of the reference's Net.forward per outer patch; the k = 33 feature graphs and
the k = 5 inter-level sets as a 16-bit hash per row plus the explicit set of every TIGHT row (tests/chain_replay.py,
ChainAll).  On the CPU stand-in backend, for a sample of the patches (the device twin in tests/test_c2_parity.py runs
all of them):
  * replay: the build chooses, rows whose hash differs from the reference's take the reference's set -- every such row
    must be a tight one (no UNEXPLAINED flip) -- and every level's cloud and the final 4992 points are within 1e-5;
  * on its own choices the first differing choice is named per patch and every level before it is within 1e-5."""
import numpy as np
import pytest
import torch

from chain_replay import first_flip_all, run_chain_all
from conftest import golden, pkg
from oracle.backend import OracleBackend

TOL = 1e-5
# (a sample per record keeps the CPU suite short; tests/test_c2_parity.py runs every recorded patch on the device)
SAMPLES = {"c2_chain_all.npz": [6, 16, 47], "c2_chain_all_seed1.npz": [0, 11],
           "c2_chain_all_trained.npz": [13, 40]}             # (trained weights: tests/golden/net16_trained.npz)


@pytest.fixture()
def modules(orc, monkeypatch):
    ops, ups = pkg("network.operations"), pkg("network.upsampler")
    monkeypatch.setattr(ops, "BACKEND", OracleBackend())

    def net_for(g):
        net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5)
        state = golden(str(g["weights"]) if "weights" in g.files else "net16_state.npz")
        net.load_state_dict({k: torch.from_numpy(state[k]) for k in state.files if k != "meta"}, strict=True)
        return net.eval()
    return ops, net_for


def level_errors(g, ids, levels, x16):
    err = np.zeros((len(ids), 4))
    for i, q in enumerate(ids):
        for l in (1, 2, 3, 4):
            ref = g["p%d_l%d_out" % (q, l)]
            mine = levels[l - 1][i].T if l < 4 else x16[i]
            err[i, l - 1] = np.abs(mine - ref).max()
    return err


def test_the_records_cover_the_clouds():
    g0, g1, c2 = golden("c2_chain_all.npz"), golden("c2_chain_all_seed1.npz"), golden("c2_x16.npz")
    gt = golden("c2_chain_all_trained.npz")
    assert str(gt["weights"]) == "net16_trained.npz" and [int(q) for q in gt["patch_ids"]] == list(range(48))
    np.testing.assert_array_equal(gt["cloud"], g0["cloud"])          # same cloud, other weights: other outputs
    assert not np.allclose(gt["p0_l4_out"], g0["p0_l4_out"], atol=1e-3)
    st, sr = golden("net16_trained.npz"), golden("net16_state.npz")
    moved = [k for k in sr.files if k != "meta" and np.abs(st[k] - sr[k]).max() > 1e-3]
    assert len(moved) >= 100, len(moved)                              # levels 1-3 trained (120 of 160 tensors moved)
    assert [int(q) for q in g0["patch_ids"]] == list(range(48))
    assert [int(q) for q in g1["patch_ids"]] == list(range(16))
    # seed 0 IS the c2_x16.npz run: same cloud, same outer patches, same final points per patch
    np.testing.assert_array_equal(g0["cloud"], c2["cloud"])
    np.testing.assert_array_equal(g0["outer_patch_idx"], c2["patch_idx"][0])
    assert int(g1["cloud_seed"]) == 1 and not np.array_equal(g1["cloud"], g0["cloud"])
    cloud = c2["cloud"][0]
    for q in (0, 17, 47):
        pts = cloud[:, g0["outer_patch_idx"][q].astype(np.int64)]
        centroid = pts.mean(axis=1, keepdims=True)
        radius = np.sqrt(((pts - centroid) ** 2).sum(axis=0)).max()
        np.testing.assert_allclose(g0["p%d_l4_out" % q] * radius + centroid,
                                   c2["pred_concat"][0][:, q * 4992:(q + 1) * 4992], rtol=0, atol=5e-6)
    # the share of tight rows is what makes the record small: a few per cent
    rows = sum(g0["p%d_gh" % q].size for q in range(48))
    tight = sum(g0["p%d_gt" % q].shape[0] for q in range(48))
    assert 0.005 < tight / rows < 0.05, tight / rows


@pytest.mark.parametrize("name", list(SAMPLES))
def test_chain_all_replayed_is_within_1e5_with_no_unexplained_flip(modules, name):
    ops, net_for = modules
    g, ids = golden(name), SAMPLES[name]
    chain, levels, x16 = run_chain_all(ops, net_for(g), g, ids, torch.device("cpu"), "replay")
    assert chain.graph_calls == 16 and chain.levels_closed == 3
    err = level_errors(g, ids, levels, x16)
    print("%s, CPU stand-in, replay: rows forced to the reference's set %s; max |dx| per outer patch %s and level:\n%s"
          % (name, {k: v for k, v in chain.forced.items() if v}, ids, err))
    assert chain.unexplained == [], chain.unexplained[:10]
    assert err.max() <= TOL, err


@pytest.mark.parametrize("name", list(SAMPLES))
def test_chain_all_on_its_own_departs_only_at_named_flips(modules, name):
    ops, net_for = modules
    g, ids = golden(name), SAMPLES[name]
    chain, levels, x16 = run_chain_all(ops, net_for(g), g, ids, torch.device("cpu"), "record")
    err = level_errors(g, ids, levels, x16)
    for i, q in enumerate(ids):
        flip = first_flip_all(chain, g, i, q)
        upto = 4 if flip is None else int(flip[1]) - 1
        print("%s outer patch %2d: first choice that differs from the reference's: %-10s max |dx| per level %s"
              % (name, q, flip, " ".join("%.1e" % e for e in err[i])))
        assert (err[i, :upto] <= TOL).all(), (q, flip, err[i])
