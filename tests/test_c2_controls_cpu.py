"""The control the C2 parity claim rests on: how far apart are two fp32 evaluations of the
REFERENCE ITSELF at the metric's own configuration?

tests/golden/c2_x16.npz is the reference's Python driven as its main.py:214-246 + :375-380 drive it (oracle/
make_golden.py::make_c2_golden).  Three more runs of the SAME reference code, weights and cloud, each with equal
arithmetic in another summation order:

  c2_x16_alt.npz   every group_knn with the channel axis of query and points reversed (operations.py:151-162: |q|^2,
                   q.p, |p|^2 are the same sums in the opposite order -- distances move by an ulp, nothing else)
  c2_x16_alt2.npz  torch's oneDNN convolution path switched off (ATen native kernels: another blocking of the same
                   products; the small batches of levels 1-2 happen to come out bit-identical)
  c2_x16_alt3.npz  every nn.Conv1d / nn.Conv2d with the input-channel axis of activation and weight reversed (what any
                   other implementation of the MLPs -- another BLAS, MIOpen, an MFMA kernel -- does to the last bits)

The reference is deterministic run to run (re-generating c2_x16.npz reproduces it bit for bit), and chaotic across
summation orders: one flipped 33rd-neighbour tie or one swapped pair of FPS picks re-orders everything downstream.
These numbers are the floor tests/test_c2_parity.py holds the HIP path to (within 1.25x, number by number).
"""
import json
import os

import numpy as np
import pytest

from conftest import ROOT, golden

# measured with oracle/parity_np.py on the committed fixtures (scipy cKDTree, float64 distances)
PINNED = {
    "alt": dict(merged_chamfer=1.0668e-05, merged_set=0.91614, merged_pos=0.78589, final_chamfer=2.3820e-04,
                final_set=0.59396, levels=[48, 38, 31, 12]),
    "alt2": dict(merged_chamfer=7.6538e-06, merged_set=0.94212, merged_pos=0.77496, final_chamfer=2.4188e-04,
                 final_set=0.60246, levels=[48, 48, 4, 0]),
    "alt3": dict(merged_chamfer=3.7269e-05, merged_set=0.74683, merged_pos=0.37967, final_chamfer=3.5713e-04,
                 final_set=0.36098, levels=[45, 4, 0, 0]),
}


@pytest.fixture(scope="module")
def runs():
    return {k: golden("c2_x16%s.npz" % ("" if k == "ref" else "_" + k)) for k in ("ref", "alt", "alt2", "alt3")}


def test_controls_share_everything_before_the_first_network_level(runs):
    for k in ("alt", "alt2", "alt3"):
        for name in ("cloud", "seed_idx", "patch_idx"):
            np.testing.assert_array_equal(runs[k][name], runs["ref"][name])


@pytest.mark.parametrize("name", ["alt", "alt2", "alt3"])
def test_reference_vs_reference_is_pinned(runs, name):
    from oracle import parity_np as pn
    r = pn.compare_runs(runs["ref"], runs[name])
    print("ref vs %s: %r" % (name, r))
    p = PINNED[name]
    assert r["merged_chamfer"] == pytest.approx(p["merged_chamfer"], rel=2e-3)
    assert r["merged_set_close_1e-5"] == pytest.approx(p["merged_set"], abs=2e-4)
    assert r["merged_position_wise_close_1e-5"] == pytest.approx(p["merged_pos"], abs=2e-4)
    assert r["final_chamfer"] == pytest.approx(p["final_chamfer"], rel=2e-3)
    assert r["final_set_close_1e-5"] == pytest.approx(p["final_set"], abs=2e-4)
    assert r["patches_exact_through_level"] == p["levels"]


def test_the_reference_is_not_within_1e_5_of_itself(runs):
    """The statement DESIGN section 2 makes: under a re-ordered but equal evaluation of its convolutions the
    reference's own merged cloud keeps a quarter of its points outside the 1e-5 band of the other run, and its final
    80 000 two thirds -- 'upsampled xyz within 1e-5' is attainable level by level (teacher-forced tests), not end to
    end, for ANY second implementation."""
    from oracle import parity_np as pn
    r = pn.compare_runs(runs["ref"], runs["alt3"])
    assert r["merged_set_close_1e-5"] < 0.80 and r["final_set_close_1e-5"] < 0.40
    assert r["patches_exact_through_level"][3] == 0
    # ... while level 1 (no discrete choice upstream but the outer kNN) holds for nearly every patch
    assert r["patches_exact_through_level"][0] >= 45


def test_committed_cpu_path_numbers_match_their_generator():
    """profiles/r04_c2_cpu_vs_ref.json is `python -m oracle.cpu_baseline --c2` (the oracle-driven CPU path of the
    product's host logic against the same fixtures, 2.5 min): its reference-vs-reference rows must be the pinned ones
    (the CPU-path rows are re-measured by that command, not here)."""
    path = os.path.join(ROOT, "profiles", "r04_c2_cpu_vs_ref.json")
    with open(path) as f:
        j = json.load(f)
    for name, p in PINNED.items():
        row = j["ref_vs_" + name]
        assert row["merged_chamfer"] == pytest.approx(p["merged_chamfer"], rel=2e-3)
        assert row["patches_exact_through_level"] == p["levels"]
    cpu = j["cpu_path_vs_ref"]
    floor = j["ref_vs_alt3"]
    assert cpu["merged_chamfer"] <= 1.25 * floor["merged_chamfer"]
    assert 1 - cpu["merged_set_close_1e-5"] <= 1.25 * (1 - floor["merged_set_close_1e-5"])
    assert cpu["final_chamfer"] <= 1.25 * floor["final_chamfer"]
    assert 1 - cpu["final_set_close_1e-5"] <= 1.25 * (1 - floor["final_set_close_1e-5"])
