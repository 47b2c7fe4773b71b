"""Input generator of the slab-form regression tests (csrc/knn.hip, knn_graph_slab_kernel; advisor finding on r5).

`sparse_line_with_one_bin_cluster` builds a 24-d patch along ONE line (t = the coordinate along it, unit = the spacing
of the sparse part):

    64 sparse rows  at t = 0, -1, ..., -63          (original row 0 = the one at -63: it fixes the sign of the
                                                      pre-pass direction, v points from row 0 to the farthest row)
    nh "high" rows  at t in [32.25, 32.45]           } one cluster, narrower in t than a bin of the pre-pass, spread by
    nl "low"  rows  at t in [31.0, 31.5]             } 1 .. 4 units PERPENDICULAR to the line (so that the cluster's
                                                      mutual distances are far above the rounding of the expanded-form
                                                      distance and no truncated keys collide: no exact-path event);
                                                      original rows 1 .. nh + nl: one wave of the order kernel
    the rest        spread over [far_lo, 400]        (a long tail: wide bins)

The sparse rows sort into positions 0 .. 63 = the first wave of the graph kernel; the cluster follows at position 64.
When the cluster falls into a single bin and the high rows arrive first, chunk 2 holds only high rows: for the query at
t = 0 (32nd neighbour: the sparse row at -32, squared distance 1024) the projected gap to chunk 2 is 32.25 > 32, likewise
for every other lane, and a table of per-chunk ranges closes the right side -- although chunk 3 holds the low rows, at
squared distance <= 31.5^2 + 4^2 = 1008 from that query, which belong in its list.  With the suffix-min table chunk 2's
bound is ~31 and the side stays open.  `far_lo` moves the mean (hence the bin grid) so that a sweep has cases with the
cluster inside one bin; `highs_first` = False is the mirrored arrival order."""
import numpy as np


def sparse_line_with_one_bin_cluster(rng, n=312, c=24, far_lo=42.0, far_hi=400.0, nh=32, nl=31, highs_first=True,
                                     rho=4.0):
    u = rng.standard_normal(c)
    u /= np.linalg.norm(u)
    sparse = -np.arange(64, dtype=np.float64)
    sparse[1:32] += 0.01 * (rng.random(31) - 0.5)   # (no exactly equidistant pairs; rows 0 and -32 .. stay put)
    highs = np.linspace(32.25, 32.45, nh)
    lows = np.linspace(31.0, 31.5, nl)
    nfar = n - 64 - nh - nl
    far = np.linspace(far_lo, far_hi, nfar) + 0.2 * (rng.random(nfar) - 0.5)
    clus = np.concatenate([highs, lows]) if highs_first else np.concatenate([lows, highs])
    t = np.concatenate([[sparse[63]], clus, sparse[:63][rng.permutation(63)], far[rng.permutation(nfar)]])
    t = t - (t.max() + t.min()) / 2                 # centred: the kernel's margins scale with max |x|^2
    x = t[:, None] * u[None, :]
    pn = rng.standard_normal((nh + nl, c))
    pn -= (pn @ u)[:, None] * u[None, :]
    pn *= (rho * (0.25 + 0.75 * rng.random((nh + nl, 1)))) / np.linalg.norm(pn, axis=1, keepdims=True)
    x[1:1 + nh + nl] += pn
    return np.ascontiguousarray(x.astype(np.float32))
