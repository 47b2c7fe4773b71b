"""-m gpu: parity at the METRIC's own configuration (BASELINE config C2: one 5000-point Poisson-sphere cloud,
16x, 312-point patches, 48 outer patches, 4 progressive levels, 239 616 -> 80 000 final FPS).

tests/golden/c2_x16.npz was produced in the build container by the REFERENCE's own operations / Net driven exactly
as its main.py:214-246 (pc_prediction) + :375-380 (concat, final FPS) drive them, one outer patch at a time at
batch 1 (oracle/make_golden.py `make_c2_golden`; FPS / gather inside served by the C oracle).  Here the HIP path
(pipeline.upsample: every patch of every level in batched launches) runs the same cloud with the same weights.

What can be asked of two fp32 implementations of this pipeline: the discrete choices -- FPS
picks, kNN sets -- are bit-exact wherever their inputs agree, the feature-space kNN (k = 33 in 24-d) flips ~1e-4
of its choices on one-ulp distance noise (the reference evaluates D through a BLAS matmul), and one flip
re-orders every FPS downstream of it.  So: (1) everything up to the first network level is bit-exact; (2) the
final FPS is bit-exact GIVEN the reference's merged cloud; (3) end to end the clouds agree as point sets /
in Chamfer distance at a small fraction of the point spacing -- with the measured numbers pinned below."""
import numpy as np
import pytest
import torch

from conftest import golden, pkg

pytestmark = pytest.mark.gpu


def _net(dev):
    ups = pkg("network.upsampler")
    net = ups.Net(max_up_ratio=16, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5)
    state = golden("net16_state.npz")
    net.load_state_dict({k: torch.from_numpy(state[k]) for k in state.files if k != "meta"})
    return net.to(dev).eval()


def set_stats(ml, mine_cl, ref_cl, tol=1e-5):
    """mine_cl, ref_cl (1,n,3) device tensors -> (Chamfer = mean squared NN distance both ways as
    model_loss.py:50-85 defines it, fraction of points with a partner within tol in the other set -- the smaller of
    the two directions)."""
    d1, _, d2, _ = ml.nndistance(mine_cl.contiguous(), ref_cl.contiguous())
    chamfer = float(d1.mean() + d2.mean())
    close = min(float((d1.sqrt() <= tol).float().mean()), float((d2.sqrt() <= tol).float().mean()))
    return chamfer, close


def test_c2_cloud_is_the_bench_workload(dev):
    """The fixture's input cloud IS bench.py's C2 workload (poisson_sphere(seed 0)): regenerated on the device
    through the HIP FPS, bit for bit."""
    import bench
    ops = pkg("network.operations")
    g = golden("c2_x16.npz")
    cloud = bench.poisson_sphere(0, 5000, dev, ops)
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


def c2_parity(dev, net=None):
    """The numbers bench.py reports as `parity_c2` and the test below pins."""
    pipe, ml = pkg("pipeline"), pkg("network.model_loss")
    g = golden("c2_x16.npz")
    net = net if net is not None else _net(dev)
    cloud = torch.from_numpy(g["cloud"]).to(dev)
    seed_idx, _, _ = pipe.extract_outer_patches(cloud, 312, 3)
    seeds_equal = bool((seed_idx.cpu().numpy() == g["seed_idx"]).all())
    merged = pipe.upsample(net, cloud, 312, 16, 3, final_fps=False)                  # (1,3,239616)
    final = pipe.upsample(net, cloud, 312, 16, 3)                                    # (1,3,80000)
    ref_merged = torch.from_numpy(g["pred_concat"]).to(dev)
    ref_final = torch.from_numpy(g["final"]).to(dev)
    cd_m, close_m = set_stats(ml, merged.transpose(2, 1), ref_merged.transpose(2, 1))
    cd_f, close_f = set_stats(ml, final.transpose(2, 1), ref_final.transpose(2, 1))
    # the scale the Chamfer numbers are to be read against: squared nearest-neighbour spacing of the reference's
    # 80 000 output points
    ops = pkg("network.operations")
    rf = ref_final.transpose(2, 1).contiguous()
    _, d_self, _ = ops.knn_query(2, rf, rf, unique=False, want_grouped=False)
    spacing2 = float(d_self[:, :, 1].clamp_min(0).median())
    pos_m = float(((merged - ref_merged).abs().amax(dim=1) <= 1e-5).float().mean())
    return {"config": "C2: 1 cloud x 5000 pts (poisson_sphere seed 0), num_point=312, up_ratio=16, 48 outer patches, "
                      "239616 -> FPS 80000; reference = its own Python driven per patch (tests/golden/c2_x16.npz)",
            "outer_seeds_bit_exact": seeds_equal,
            "merged_chamfer_vs_ref": cd_m, "merged_set_close_1e-5": close_m, "merged_position_wise_close_1e-5": pos_m,
            "final_chamfer_vs_ref": cd_f, "final_set_close_1e-5": close_f,
            "ref_output_spacing_sq_median": spacing2,
            "final_shape": list(final.shape)}


def test_c2_end_to_end_against_the_reference_driver(dev):
    r = c2_parity(dev)
    print("C2 parity: %r" % (r,))
    assert r["final_shape"] == [1, 3, 80000]
    assert r["outer_seeds_bit_exact"]
    # Measured on MI355X (round 3, profiles/r03_parity.txt): merged (239 616 points before the final FPS) Chamfer
    # 3.19e-5, 77.5 % of the points coincide with a reference point within 1e-5; final 80 000 points Chamfer 3.53e-4 =
    # 0.51 x the squared point spacing (6.9e-4), 37.5 % coincide -- the final FPS picks one third of a cloud whose
    # other points differ in a quarter of the positions, so its choices decorrelate.  The oracle-driven CPU path of
    # the same host logic scores 2.60e-5 / 81.3 % / 3.44e-4 / 39.8 % against the same fixture: the distance is
    # between ANY two fp32 evaluations of this pipeline, not between HIP and CPU.  Thresholds: measured, <= 2x slack.
    assert r["merged_chamfer_vs_ref"] < 6.4e-5
    assert r["merged_set_close_1e-5"] > 0.60
    assert r["final_chamfer_vs_ref"] < 7.0e-4
    assert r["final_chamfer_vs_ref"] < 1.0 * r["ref_output_spacing_sq_median"]
    assert r["final_set_close_1e-5"] > 0.30


def test_c1_hip_path_against_the_oracle_driven_path(dev):
    """BASELINE config C1 (5000 points, 2x, one level, 48 patches -> 29 952 -> FPS 10 000): the HIP path against the
    oracle-driven CPU path on the same cloud and weights -- the comparison bench.py prints as `parity`."""
    import bench
    from oracle import cpu_baseline
    ops, pipe, ups = pkg("network.operations"), pkg("pipeline"), pkg("network.upsampler")
    _, cpu_out = cpu_baseline.measure_c1(repeats=1)
    r = bench.parity_block(ops, pipe, ups, dev, cpu_out)
    print("C1 parity: %r" % (r,))
    assert r["chamfer_vs_oracle"] < 1e-10
    assert r["set_close_1e-5"] >= 0.998
    assert r["position_wise_close_1e-5"] >= 0.98
