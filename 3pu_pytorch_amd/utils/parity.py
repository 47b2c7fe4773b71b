"""Distances between two evaluations of the upsampling pipeline, on the device (HIP nm-distance kernel).

Used by bench.py (`parity_c2`) and tests/test_c2_parity.py; oracle/parity_np.py is the same definition on CPU
arrays (scipy), which the `-m "not gpu"` suite pins against the reference-vs-reference control fixtures.

A "run" of config C2 is a dict of arrays / tensors: lv1 (48,3,624), lv2 (48,3,1248), lv3 (48,3,2496) -- the cloud
every outer patch holds after level 1..3, de-normalised --, pred_concat (1,3,239616) = level 4 of all patches in patch
order, final (1,3,80000).  tests/golden/c2_x16.npz is the reference driver's run, c2_x16_alt*.npz are the reference's
OWN code evaluated with equal arithmetic in another summation order (oracle/make_golden.py): the distance between two
of those is the floor any fp32 implementation is measured against.
"""
import numpy as np
import torch

from .. import pipeline
from ..network import model_loss, operations

TOL = 1e-5            # north_star: "upsampled xyz within 1e-5 fp32"
KEYS = ("lv1", "lv2", "lv3", "pred_concat", "final")


def _t(x, dev):
    return (x if torch.is_tensor(x) else torch.from_numpy(np.asarray(x))).to(dev)


def set_stats(a_cl, b_cl, tol=TOL):
    """a_cl, b_cl (1,n,3) device tensors -> (Chamfer = mean squared NN distance both ways as model_loss.py:50-85
    defines it, share of points with a partner within tol in the other set -- the smaller of the two directions)."""
    d1, _, d2, _ = model_loss.nndistance(a_cl.contiguous(), b_cl.contiguous())
    chamfer = float(d1.mean() + d2.mean())
    close = min(float((d1.sqrt() <= tol).float().mean()), float((d2.sqrt() <= tol).float().mean()))
    return chamfer, close


def level_clouds(run, dev):
    p = _t(run["pred_concat"], dev)
    n4 = p.shape[2] // 48
    return [_t(run["lv1"], dev), _t(run["lv2"], dev), _t(run["lv3"], dev),
            p.reshape(3, 48, n4).permute(1, 0, 2).contiguous()]


def patches_exact_through(a, b, dev, tol=TOL):
    """How many of the 48 outer patches agree position by position within tol THROUGH level k, k = 1..4."""
    alive = torch.ones(48, dtype=torch.bool, device=dev)
    out = []
    for x, y in zip(level_clouds(a, dev), level_clouds(b, dev)):
        alive &= (x - y).abs().reshape(48, -1).amax(dim=1) <= tol
        out.append(int(alive.sum()))
    return out


def compare_runs(a, b, dev, tol=TOL):
    pa, pb = _t(a["pred_concat"], dev), _t(b["pred_concat"], dev)
    cd_m, close_m = set_stats(pa.transpose(2, 1), pb.transpose(2, 1), tol)
    cd_f, close_f = set_stats(_t(a["final"], dev).transpose(2, 1), _t(b["final"], dev).transpose(2, 1), tol)
    return {"merged_chamfer": cd_m, "merged_set_close_1e-5": close_m,
            "merged_position_wise_close_1e-5": float(((pa - pb).abs().amax(dim=1) <= tol).float().mean()),
            "final_chamfer": cd_f, "final_set_close_1e-5": close_f,
            "patches_exact_through_level": patches_exact_through(a, b, dev, tol)}


@torch.no_grad()
def run_c2(net, cloud, num_point=312, up_ratio=16, patch_num_ratio=3):
    """Config C2 through the product path with the per-level clouds recorded: cloud (1,3,5000) on the device ->
    a run (device tensors) + the outer seeds / patch indices."""
    seed_idx, patches, pidx = pipeline.extract_outer_patches(cloud, num_point, patch_num_ratio)
    P = patches.size(1)
    levels = []
    up, _ = pipeline.upsample_patches(net, patches.reshape(P, num_point, 3), up_ratio, levels_out=levels)
    merged = up.reshape(1, P * up.size(1), 3)
    idx = operations.fps(merged, cloud.shape[2] * up_ratio)
    final = torch.gather(merged, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3))
    return {"seed_idx": seed_idx, "patch_idx": pidx[0], "lv1": levels[0], "lv2": levels[1], "lv3": levels[2],
            "pred_concat": merged.transpose(2, 1).contiguous(), "final": final.transpose(2, 1).contiguous()}


def control_floor(pairs):
    """The loosest of several reference-vs-reference comparisons, number by number."""
    return {"merged_chamfer": max(p["merged_chamfer"] for p in pairs),
            "merged_set_close_1e-5": min(p["merged_set_close_1e-5"] for p in pairs),
            "merged_position_wise_close_1e-5": min(p["merged_position_wise_close_1e-5"] for p in pairs),
            "final_chamfer": max(p["final_chamfer"] for p in pairs),
            "final_set_close_1e-5": min(p["final_set_close_1e-5"] for p in pairs),
            "patches_exact_through_level": [min(p["patches_exact_through_level"][k] for p in pairs) for k in range(4)]}


def within_floor(mine, floor, slack=1.25):
    """HIP-vs-reference is acceptable when every number is within `slack` of the reference-vs-reference floor:
    distances at most slack x the floor's, shares of NON-coinciding points at most slack x the floor's."""
    bad = []
    for k in ("merged_chamfer", "final_chamfer"):
        if not mine[k] <= slack * floor[k]:
            bad.append((k, mine[k], floor[k]))
    for k in ("merged_set_close_1e-5", "final_set_close_1e-5", "merged_position_wise_close_1e-5"):
        if not (1.0 - mine[k]) <= slack * (1.0 - floor[k]):
            bad.append((k, mine[k], floor[k]))
    return bad


def c2_parity(dev, net, fixtures, spacing=True):
    """bench.py's `parity_c2` / the numbers tests/test_c2_parity.py pins.  fixtures: {"ref": run, "alt": run, ...}
    (the arrays of tests/golden/c2_x16*.npz); net: the seed-0 Net on `dev` (tests/golden/net16_state.npz)."""
    ref = fixtures["ref"]
    mine = run_c2(net, _t(ref["cloud"], dev))
    out = {"config": "C2: 1 cloud x 5000 pts (poisson_sphere seed 0), num_point=312, up_ratio=16, 48 outer patches, "
                     "239616 -> FPS 80000; reference = its own Python driven per patch (tests/golden/c2_x16.npz)",
           "outer_seeds_bit_exact": bool((mine["seed_idx"].cpu().numpy() == np.asarray(ref["seed_idx"])).all()),
           "outer_patch_idx_mismatches": int((mine["patch_idx"].cpu().numpy() != np.asarray(ref["patch_idx"])[0]).sum()),
           "final_shape": list(mine["final"].shape)}
    vs = compare_runs(mine, ref, dev)
    out.update({"merged_chamfer_vs_ref": vs["merged_chamfer"], "merged_set_close_1e-5": vs["merged_set_close_1e-5"],
                "merged_position_wise_close_1e-5": vs["merged_position_wise_close_1e-5"],
                "final_chamfer_vs_ref": vs["final_chamfer"], "final_set_close_1e-5": vs["final_set_close_1e-5"],
                "patches_exact_through_level": vs["patches_exact_through_level"]})
    controls = {k: v for k, v in fixtures.items() if k != "ref"}
    if controls:
        pairs = {k: compare_runs(ref, v, dev) for k, v in controls.items()}
        out["ref_vs_ref"] = pairs
        out["ref_vs_ref_floor"] = control_floor(list(pairs.values()))
        out["outside_1.25x_floor"] = ["%s: %.4g vs floor %.4g" % t for t in within_floor(vs, out["ref_vs_ref_floor"])]
        out["hip_vs_controls"] = {k: compare_runs(mine, v, dev) for k, v in controls.items()}
    if spacing:
        rf = _t(ref["final"], dev).transpose(2, 1).contiguous()
        _, d_self, _ = operations.knn_query(2, rf, rf, unique=False, want_grouped=False)
        out["ref_output_spacing_sq_median"] = float(d_self[:, :, 1].clamp_min(0).median())
    return out


@torch.no_grad()
def c1_parity(dev, net, cloud, other_cl, num_point=312):
    """Config C1 (5000 points, 2x, one level) on the device against another evaluation of the same cloud and weights
    (bench.py: the oracle-driven CPU output, (1, 2N, 3) channel-last)."""
    N = cloud.shape[2]
    out = pipeline.upsample(net, cloud.to(dev), num_point, 2, 3)                              # (1,3,2N)
    mine = out.transpose(2, 1).contiguous()
    ref = other_cl.to(dev)
    cd, close = set_stats(mine, ref)
    same = float(((mine - ref).abs().amax(dim=2) <= TOL).float().mean())      # identical positions: same FPS order too
    return {"config": "C1: 1 cloud x %d pts, num_point=%d, up_ratio=2, one level, 48 patches -> FPS %d"
                      % (N, num_point, 2 * N),
            "chamfer_vs_oracle": cd, "set_close_1e-5": close, "position_wise_close_1e-5": same,
            "note": "HIP path vs the oracle-driven CPU path (oracle/cpu_baseline.py) on the same cloud and weights; "
                    "chamfer = mean squared NN distance both ways (model_loss.py:50-85)"}
