"""-m gpu: the N > 1 branch of bench.py executed on a device.

The GPU boxes the tests run on have ONE GPU, so the ranks share cuda:0 and the collectives go through gloo
(TPU3_BENCH_BACKEND=gloo TPU3_BENCH_ONE_DEVICE=1: bench.py's functional mode, never a measurement).  What is
checked is everything but the transport: the launch form the driver uses (torch.distributed.run, one rank per
"GPU"), sharding of clouds / of one cloud's outer patches, the single all-gather, rank-0 reporting with the `comm`
block -- and that the gathered result is bit for bit what one rank computes on the same clouds (BASELINE config
C4 in miniature; reference main.py:237-244,375-380)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _bench(nproc, extra, timeout=900):
    env = dict(os.environ)
    env.update(TPU3_BENCH_BACKEND="gloo", TPU3_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    common = ["--steps", "1", "--warmup", "0", "--no_cpu_baseline", "--no_extras", "--digest",
              "--net_streams", "2", "--fps_streams", "1"]
    if nproc == 1:
        cmd = [sys.executable, "bench.py", "--gpus", "1"] + common + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
               "bench.py", "--gpus", str(nproc)] + common + extra
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=timeout)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]            # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_two_ranks_clouds_sharded_equals_one_rank():
    two = _bench(2, ["--clouds", "2"])
    assert two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["comm"]["world_size"] == 2 and two["comm"]["backend"] == "gloo"
    assert two["comm"]["allgather_bytes_total"] == 2 * 2 * 3 * 80000 * 4
    assert two["value"] > 0 and two["config"]["clouds_per_gpu"] == 2
    one = _bench(1, ["--clouds", "4"])                      # the same four clouds (seeds 0..3) on one rank
    assert len(two["result_digest"]) == 4
    assert two["result_digest"] == one["result_digest"]


def test_bench_two_ranks_patches_sharded_equals_one_rank():
    two = _bench(2, ["--shard", "patches"])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["comm"]["world_size"] == 2
    one = _bench(1, ["--clouds", "1"])
    assert len(two["result_digest"]) == 1
    assert two["result_digest"] == one["result_digest"]


def test_bench_eight_ranks_dry_run_equals_one_rank():
    """The driver's scaling run launches `bench.py --gpus 8` as eight ranks.  No 8-GPU node is available to the
    tests, so the eight ranks share cuda:0 over gloo -- every line of the N = 8 path but the transport: eight
    processes initialise, shard_range deals clouds 0..7 one per rank, the single all-gather reassembles them in
    rank order, rank 0 alone prints the line, every rank leaves through the final barrier.  The gathered clouds
    are bit for bit what one rank computes for the same eight seeds (config C4 at one cloud per rank)."""
    eight = _bench(8, ["--clouds", "1"], timeout=1500)
    assert eight["n_gpus"] == 8 and eight["scaling"] == "weak"
    assert eight["comm"]["world_size"] == 8 and eight["comm"]["backend"] == "gloo"
    # (r6) first-contact fields: every rank names its device, and the line states what scaling to expect
    assert len(eight["comm"]["rank_devices"]) == 8 and all("uuid=" in d for d in eight["comm"]["rank_devices"])
    assert [d.split(":")[0] for d in eight["comm"]["rank_devices"]] == ["rank %d" % i for i in range(8)]
    assert "weak scaling" in eight["comm"]["expectation"]
    assert eight["comm"]["allgather_bytes_total"] == 8 * 1 * 3 * 80000 * 4
    assert eight["config"]["clouds_per_gpu"] == 1 and eight["value"] > 0
    one = _bench(1, ["--clouds", "8"])
    assert len(eight["result_digest"]) == 8
    assert eight["result_digest"] == one["result_digest"]
