"""oracle/parity_np.py -- TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.

Distances between two evaluations of config C2 (BASELINE: 5000 -> 80 000 points, 16x) on CPU arrays: what
tests/test_c2_controls_cpu.py asserts about the reference-vs-reference controls, what
`python -m oracle.cpu_baseline --c2` writes into profiles/, and the definition the device-side
3pu_pytorch_amd/utils/parity.py repeats with the HIP nm-distance kernel.

A "run" is a dict with the arrays of a tests/golden/c2_x16*.npz fixture: lv1 (48,3,624), lv2 (48,3,1248),
lv3 (48,3,2496) = the cloud every outer patch holds after level 1..3 (de-normalised), pred_concat (1,3,239616) =
level 4 of all patches in patch order, final (1,3,80000).
"""
import numpy as np
from scipy.spatial import cKDTree

TOL = 1e-5            # north_star: "upsampled xyz within 1e-5 fp32"


def set_stats(x, y, tol=TOL):
    """x, y (3,n): (Chamfer = mean squared NN distance both ways, model_loss.py:50-85; share of points with a
    partner within tol in the other set -- the smaller of the two directions)."""
    X, Y = np.ascontiguousarray(x.T, np.float64), np.ascontiguousarray(y.T, np.float64)
    d1, _ = cKDTree(Y).query(X)
    d2, _ = cKDTree(X).query(Y)
    return float((d1 ** 2).mean() + (d2 ** 2).mean()), float(min((d1 <= tol).mean(), (d2 <= tol).mean()))


def level_clouds(run):
    """[(48,3,n_l) for l = 1..4]"""
    p = np.asarray(run["pred_concat"])
    n4 = p.shape[2] // 48
    return [np.asarray(run["lv1"]), np.asarray(run["lv2"]), np.asarray(run["lv3"]),
            np.ascontiguousarray(p.reshape(3, 48, n4).transpose(1, 0, 2))]


def patches_exact_through(a, b, tol=TOL):
    """How many of the 48 outer patches agree position by position within tol THROUGH level k, k = 1..4 (a patch
    that left the band at level k stays out: everything downstream of a flipped discrete choice is re-ordered)."""
    la, lb = level_clouds(a), level_clouds(b)
    alive = np.ones(48, bool)
    out = []
    for x, y in zip(la, lb):
        alive &= np.abs(x - y).reshape(48, -1).max(axis=1) <= tol
        out.append(int(alive.sum()))
    return out


def compare_runs(a, b, tol=TOL):
    """The numbers every C2 comparison reports."""
    pa, pb = np.asarray(a["pred_concat"]), np.asarray(b["pred_concat"])
    cd_m, close_m = set_stats(pa[0], pb[0], tol)
    cd_f, close_f = set_stats(np.asarray(a["final"])[0], np.asarray(b["final"])[0], tol)
    return {"merged_chamfer": cd_m, "merged_set_close_1e-5": close_m,
            "merged_position_wise_close_1e-5": float((np.abs(pa - pb).max(axis=1) <= tol).mean()),
            "final_chamfer": cd_f, "final_set_close_1e-5": close_f,
            "patches_exact_through_level": patches_exact_through(a, b, tol)}
