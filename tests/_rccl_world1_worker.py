"""Worker of tests/test_rccl_world1.py: pipeline.upsample's sharded forms under a REAL RCCL process group at world
size 1 (TPU3_FORCE_COLLECTIVES=1 is set by the test before this module is imported)."""
import importlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]),
                            device_id=dev)
    pipe = importlib.import_module("3pu_pytorch_amd.pipeline")
    ups = importlib.import_module("3pu_pytorch_amd.network.upsampler")
    assert pipe.FORCE_COLLECTIVES and pipe._distributed(1)
    torch.manual_seed(0)
    net = ups.Net(max_up_ratio=4, step_ratio=2, knn=32, growth_rate=12, dense_n=3, fm_knn=5).to(dev).eval()
    rng = np.random.default_rng(3)
    c = rng.standard_normal((3, 700, 3)).astype(np.float32)
    c /= np.linalg.norm(c, axis=2, keepdims=True)
    clouds = torch.from_numpy(np.ascontiguousarray(c.transpose(0, 2, 1))).to(dev)
    calls = {"all_gather": 0, "all_reduce": 0}
    real_ag, real_ar = dist.all_gather_into_tensor, dist.all_reduce

    def ag(*a, **kw):
        calls["all_gather"] += 1
        return real_ag(*a, **kw)

    def ar(*a, **kw):
        calls["all_reduce"] += 1
        return real_ar(*a, **kw)
    dist.all_gather_into_tensor, dist.all_reduce = ag, ar
    by_cloud = pipe.upsample(net, clouds, 312, 4, shard="clouds")
    by_patch = pipe.upsample(net, clouds[:1], 312, 4, shard="patches")
    dist.all_gather_into_tensor, dist.all_reduce = real_ag, real_ar
    pipe.FORCE_COLLECTIVES = False
    ref = pipe.upsample(net, clouds, 312, 4, shard=None)
    torch.cuda.synchronize()
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "calls": calls,
           "clouds_equal": bool(torch.equal(by_cloud, ref)), "patches_equal": bool(torch.equal(by_patch, ref[:1])),
           "shape": list(by_cloud.shape)}
    dist.barrier()
    dist.destroy_process_group()
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
