"""-m gpu: parity at the METRIC's own configuration (BASELINE config C2: one 5000-point Poisson-sphere cloud,
16x, 312-point patches, 48 outer patches, 4 progressive levels, 239 616 -> 80 000 final FPS).

tests/golden/c2_x16.npz was produced in the build container by the REFERENCE's own operations / Net driven exactly
as its main.py:214-246 (pc_prediction) + :375-380 (concat, final FPS) drive them, one outer patch at a time at
batch 1 (oracle/make_golden.py `make_c2_golden`; FPS / gather inside served by the C oracle).  Here the HIP path
(pipeline: every patch of every level in batched launches) runs the same cloud with the same weights.

What can be asked of two fp32 implementations of this pipeline is MEASURED, not argued: c2_x16_alt{,2,3}.npz are the
reference's own code evaluated with equal arithmetic in another summation order (tests/test_c2_controls_cpu.py).
(1) Everything up to the first network level is bit-exact (seeds; the outer patch indices up to torch.topk's
unspecified order among exactly tied distances); (2) the final FPS is bit-exact GIVEN the reference's merged cloud;
(3) (r6) per outer patch, ALL 48 of them (and 16 of a second cloud): with the reference's discrete choices replayed
every level's cloud and the final 4992 points are within 1e-5 (measured 1.6e-6), every row that had to be forced is a
TIGHT one, and on the build's own choices every patch has a NAMED first flip with every level before it within 1e-5
(test_c2_chain_all_*; profiles/r06_c2_first_flips.txt).  That per-patch rule replaces the bar of rounds 4-5 ("every
aggregate number within 1.25x of the loosest reference-vs-reference control"), whose numbers are still computed and
printed next to the controls (bench.py `parity_c2`)."""
import numpy as np
import pytest
import torch

from conftest import golden, pkg

pytestmark = pytest.mark.gpu


def _net(dev, weights="net16_state.npz"):
    ups = pkg("network.upsampler")
    net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5)
    state = golden(weights)
    net.load_state_dict({k: torch.from_numpy(state[k]) for k in state.files if k != "meta"})
    return net.to(dev).eval()


def _fixtures():
    return {k: golden("c2_x16%s.npz" % ("" if k == "ref" else "_" + k)) for k in ("ref", "alt", "alt2", "alt3")}


def test_c2_cloud_is_the_bench_workload(dev):
    """The fixture's input cloud IS bench.py's C2 workload (poisson_sphere(seed 0)): regenerated on the device
    through the HIP FPS, bit for bit."""
    g = golden("c2_x16.npz")
    cloud = pkg("utils.workloads").poisson_sphere(0, 5000, dev)
    np.testing.assert_array_equal(cloud.cpu().numpy(), g["cloud"])
    # and the bench's random-init net under torch.manual_seed(0) IS net16_state.npz
    torch.manual_seed(0)
    ups = pkg("network.upsampler")
    net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5)
    state = golden("net16_state.npz")
    for k, v in net.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), state[k])


def test_c2_final_fps_on_the_reference_merged_cloud_is_bit_exact(dev):
    """main.py:379-380 at the metric's size on the reference's own merged cloud: all 80 000 indices."""
    ops = pkg("network.operations")
    g = golden("c2_x16.npz")
    merged = torch.from_numpy(np.ascontiguousarray(g["pred_concat"].transpose(0, 2, 1))).to(dev)     # (1,239616,3)
    idx = ops.fps(merged, 80000)
    np.testing.assert_array_equal(idx.cpu().numpy(), g["final_idx"])
    out = torch.gather(merged, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).transpose(2, 1)
    np.testing.assert_array_equal(out.cpu().numpy(), g["final"])


def test_c2_outer_patches_are_the_references(dev):
    """main.py:228-234 on the device: the 48 seeds bit for bit; the 48 x 312 patch indices equal as SETS for every
    patch and position by position except where two candidates are at exactly the same computed distance
    (torch.topk's order among equals is unspecified; this build takes the lower index)."""
    pipe = pkg("pipeline")
    g = golden("c2_x16.npz")
    cloud = torch.from_numpy(g["cloud"]).to(dev)
    seed_idx, _, pidx = pipe.extract_outer_patches(cloud, 312, 3)
    np.testing.assert_array_equal(seed_idx.cpu().numpy(), g["seed_idx"])
    mine, ref = pidx[0].cpu().numpy(), g["patch_idx"][0].astype(np.int64)
    np.testing.assert_array_equal(np.sort(mine, axis=1), np.sort(ref, axis=1))
    assert (mine != ref).sum() <= 4


def test_c2_end_to_end_against_the_reference_driver(dev):
    par = pkg("utils.parity")
    r = par.c2_parity(dev, _net(dev), _fixtures())
    print("C2 parity: %r" % (r,))
    assert r["final_shape"] == [1, 3, 80000]
    assert r["outer_seeds_bit_exact"] and r["outer_patch_idx_mismatches"] <= 4
    # (iv) (r6) the aggregate distances are REPORTED beside the reference-vs-reference controls (merged Chamfer 3.0 ...
    # 3.7e-5 / set 0.75 ... 0.78, final 3.5e-4 / 0.37 ... 0.39 against the controls' 3.73e-5 / 0.747 and 3.57e-4 / 0.361);
    # the pass / fail rule for "is every difference a named discrete flip" is per patch, on all 48 patches:
    # test_c2_chain_all_on_device_every_patch_is_exact_or_has_a_named_first_flip.  What stays asserted here is the
    # sanity of the whole: the output is a cloud of the reference's density around the reference's surface
    print("aggregate numbers outside 1.25x of the loosest control (informational since r6): %r" % (r["outside_1.25x_floor"],))
    assert r["final_chamfer_vs_ref"] < 1.0 * r["ref_output_spacing_sq_median"]
    # (v) how many of the 48 outer patches are position-wise within 1e-5 THROUGH level k: level 1 has no discrete
    # choice upstream of it except the outer kNN's exact ties, so (nearly) every patch must hold there -- a count that
    # cannot hide a real bug; deeper levels are held to the controls' counts
    lv = r["patches_exact_through_level"]
    assert lv[0] >= 44, lv


def test_c1_hip_path_against_the_oracle_driven_path(dev):
    """BASELINE config C1 (5000 points, 2x, one level, 48 patches -> 29 952 -> FPS 10 000): the HIP path against the
    oracle-driven CPU path on the same cloud and weights -- the comparison bench.py prints as `parity`."""
    from oracle import cpu_baseline
    ups = pkg("network.upsampler")
    _, cpu_out = cpu_baseline.measure_c1(repeats=1)
    r = pkg("utils.parity").c1_parity(dev, cpu_baseline.c1_net(ups).to(dev), cpu_baseline.c1_cloud(0, 5000), cpu_out)
    print("C1 parity: %r" % (r,))
    assert r["chamfer_vs_oracle"] < 1e-10
    assert r["set_close_1e-5"] >= 0.998
    assert r["position_wise_close_1e-5"] >= 0.98


def _chain_errors(g, ids, levels, x16):
    err = np.zeros((len(ids), 4))
    for i, q in enumerate(ids):
        for l in (1, 2, 3, 4):
            ref = g["p%d_l%d_out" % (q, l)]
            mine = levels[l - 1][i].T if l < 4 else x16[i]
            err[i, l - 1] = np.abs(mine - ref).max()
    return err


def test_c2_chain_replayed_on_device_is_within_1e5_end_to_end(dev):
    """Four outer patches of the C2 cloud through ALL FOUR levels on the HIP path with every discrete choice of the
    reference's run replayed (tests/golden/c2_chain.npz, tests/chain_replay.py: outlier masks, inner seeds, inner
    patches, the 16 feature graphs, the inter-level neighbour sets, the per-level FPS picks -- patch 16 with the
    ragged 19-of-20 inner patches at level 3): the final 4992 points of every patch, and the cloud after every level,
    within 1e-5 of the reference's.  north_star's "upsampled xyz within 1e-5" holds end to end wherever the discrete
    choices agree (the CPU twin: tests/test_c2_chain_cpu.py)."""
    from chain_replay import run_chain
    ops = pkg("network.operations")
    g = golden("c2_chain.npz")
    ids = [int(q) for q in g["patch_ids"]]
    chain, levels, x16 = run_chain(ops, _net(dev), g, ids, dev, "replay")
    assert chain.graph_calls == 16 and chain.levels_closed == 3
    err = _chain_errors(g, ids, levels, x16)
    print("chained replay on the device: max |dx| per outer patch %s (rows) and level 1..4 (columns):\n%s" % (ids, err))
    assert err.max() <= 1e-5, err


def test_c2_chain_on_device_departs_only_at_named_flips(dev):
    """The same four patches on the HIP path's OWN choices: the first choice that differs from the reference's is
    named per patch, and every level before it is within 1e-5 -- no patch drifts without a named flip (patches 16 and
    23 flip a feature-graph row already at level 1 on this path, tools/c2_level_flips.py)."""
    from chain_replay import ORDER, first_flip, run_chain
    ops = pkg("network.operations")
    g = golden("c2_chain.npz")
    ids = [int(q) for q in g["patch_ids"]]
    chain, levels, x16 = run_chain(ops, _net(dev), g, ids, dev, "record")
    assert sorted(chain.seen) == sorted(ORDER)
    err = _chain_errors(g, ids, levels, x16)
    for i, q in enumerate(ids):
        flip = first_flip(chain, g, i, q)
        upto = 4 if flip is None else int(flip[1]) - 1
        print("outer patch %2d on the device: first choice that differs from the reference's: %-10s max |dx| per level %s"
              % (q, flip, " ".join("%.1e" % e for e in err[i])))
        assert (err[i, :upto] <= 1e-5).all(), (q, flip, err[i])


# ---- (r6) ALL 48 outer patches of the C2 cloud + 16 of a second cloud, compact record ---------------
# ... and all 48 again under TRAINED weights (tests/golden/net16_trained.npz = the product's Net after
# 600 C3 training steps on the device, tools/train_weights.py; the record = the REFERENCE's Python under those weights)
CHAIN_ALL = ["c2_chain_all.npz", "c2_chain_all_seed1.npz", "c2_chain_all_trained.npz"]


def _weights_of(g):
    return str(g["weights"]) if "weights" in g.files else "net16_state.npz"


@pytest.mark.parametrize("name", CHAIN_ALL)
def test_c2_chain_all_replayed_on_device(dev, name):
    """Every outer patch of the C2 cloud (and 16 of the seed-1 cloud) through all four levels on the HIP path, every
    discrete choice of the reference's run replayed from the compact record (tests/chain_replay.py, ChainAll: the
    build chooses its feature graphs / inter-level sets itself, rows whose hash differs from the reference's take the
    reference's set).  (i) every such row is a TIGHT one -- no flip at a clear margin anywhere in ~4.3 M graph rows;
    (ii) the cloud after every level and the final 4992 points of every patch are within 1e-5 of the reference's."""
    from chain_replay import run_chain_all
    ops = pkg("network.operations")
    g = golden(name)
    ids = [int(q) for q in g["patch_ids"]]
    net = _net(dev, _weights_of(g))
    worst, forced = 0.0, {}
    for s in range(0, len(ids), 16):
        part = ids[s:s + 16]
        chain, levels, x16 = run_chain_all(ops, net, g, part, dev, "replay")
        assert chain.graph_calls == 16 and chain.levels_closed == 3
        assert chain.unexplained == [], chain.unexplained[:10]
        err = _chain_errors(g, part, levels, x16)
        assert err.max() <= 1e-5, (part, err)
        worst = max(worst, float(err.max()))
        for k, v in chain.forced.items():
            forced[k] = forced.get(k, 0) + v
    print("%s replayed on the device: %d outer patches, max |dx| over all levels %.2e; rows forced to the reference's "
          "(tight) set: %s" % (name, len(ids), worst, {k: v for k, v in sorted(forced.items()) if v}))


@pytest.mark.parametrize("name", CHAIN_ALL)
def test_c2_chain_all_on_device_every_patch_is_exact_or_has_a_named_first_flip(dev, name):
    """The HIP path's OWN choices on every recorded outer patch: the first choice that differs from the reference's is
    named, every level before it is within 1e-5, and a patch without a flip is within 1e-5 through level 4 -- no patch
    drifts without a named flip.  This replaces the 1.25x-of-the-loosest-control bar of rounds 4-5 for the per-patch
    comparison (tools/c2_first_flips.py prints the 48 lines: profiles/r06_c2_first_flips.txt)."""
    from chain_replay import first_flip_all, run_chain_all
    ops = pkg("network.operations")
    g = golden(name)
    ids = [int(q) for q in g["patch_ids"]]
    net = _net(dev, _weights_of(g))
    clean = 0
    for s in range(0, len(ids), 16):
        part = ids[s:s + 16]
        chain, levels, x16 = run_chain_all(ops, net, g, part, dev, "record")
        err = _chain_errors(g, part, levels, x16)
        for i, q in enumerate(part):
            flip = first_flip_all(chain, g, i, q)
            upto = 4 if flip is None else int(flip[1]) - 1
            assert (err[i, :upto] <= 1e-5).all(), (q, flip, err[i])
            clean += flip is None
    print("%s on the device's own choices: %d of %d outer patches without any flip through level 4" % (name, clean, len(ids)))
