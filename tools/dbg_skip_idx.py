import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from conftest import pkg, sphere
ops = pkg("network.operations")
dev = torch.device("cuda", 0)
q = torch.from_numpy(sphere(3, 312, 2)).to(dev)
p = torch.from_numpy(sphere(4, 312, 2)).to(dev)
a = ops.knn_query(5, q, p, unique=True, want_dist=False, want_grouped=True)
b = ops.knn_query(5, q, p, unique=True, want_dist=False, want_grouped=False)
print("idx equal:", bool(torch.equal(a[0], b[0])), a[0].dtype, b[0].dtype, a[0].shape, b[0].shape, a[0].is_contiguous(), b[0].is_contiguous(), b[0].stride())
