"""CPU twin of the slab-form bound (csrc/knn.hip, knn_graph_slab_kernel): tools/knn_slab_bound_sim.py restates the
pre-pass order and the side-closing walk in numpy.  The r5 table (each chunk's OWN t-range) returns a wrong neighbour
set on a cluster that sits inside one bin of the binned order; the r6 table (suffix-min / prefix-max) is exact whatever
the order inside a bin is.  The device test of the same inputs is test_hip_kernels.py::
test_knn_graph_slab_form_cluster_inside_one_bin."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import knn_slab_bound_sim as sim                      # noqa: E402
from slab_cases import sparse_line_with_one_bin_cluster   # noqa: E402


@pytest.mark.parametrize("inbin", ["index", "random"])
def test_monotone_table_is_exact_and_the_per_chunk_table_is_not(inbin):
    rng = np.random.default_rng(3)
    wrong_own = wrong_mono = one_bin = 0
    for far_lo in np.arange(40.0, 72.0, 2.0):
        x = sparse_line_with_one_bin_cluster(rng, far_lo=float(far_lo))
        t, order, bins = sim.slab_order(x, inbin, rng)
        one_bin += int(len(np.unique(bins[np.isin(order, np.arange(1, 64))])) == 1)
        exact = sim.exact_sets(x)
        wrong_own += int((sim.slab_graph(x, t, order, "own", margins=True) != exact).any(1).sum())
        wrong_mono += int((sim.slab_graph(x, t, order, "monotone", margins=True) != exact).any(1).sum())
    assert one_bin >= 4                 # the sweep does put the cluster into a single bin
    assert wrong_mono == 0
    if inbin == "index":
        assert wrong_own >= 4           # the defect the r6 table removes (arrival order = row order)
