"""-m gpu: RCCL executes before the driver's 8-GPU node is the first to try.

The GPU boxes the tests run on have ONE GPU, so the only RCCL communicator that can exist here has world size 1.
That is enough to run every call of the N-rank path through the real backend: `init_process_group("nccl",
device_id=...)`, `all_gather_into_tensor` on the final-FPS side stream, the MAX all-reduce of the event flags and of
the elapsed time, `dist.barrier()` inside the timed region's fence -- with results that must equal the plain run bit
for bit.  (The 2-rank functional check over gloo is tests/test_bench_multirank.py; reference concat site:
main.py:375.)"""
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


def _run(cmd, env_extra, timeout=900):
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(env_extra)
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=timeout)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    return res.stdout


def _torchrun(script_and_args):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
            "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_and_args


COMMON = ["--steps", "2", "--warmup", "1", "--no_cpu_baseline", "--no_extras", "--digest", "--clouds", "2",
          "--net_streams", "2", "--fps_streams", "2"]


def _line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_bench_world1_over_rccl_equals_plain_run():
    forced = _line(_run(_torchrun(["bench.py", "--gpus", "1"] + COMMON), {"TPU3_BENCH_FORCE_DIST": "1"}))
    assert forced["comm"]["backend"] == "nccl" and forced["comm"]["world_size"] == 1
    assert forced["comm"]["forced_at_world_size_1"] is True
    assert forced["comm"]["allgather_bytes_total"] == 2 * 3 * 80000 * 4 and forced["comm"]["allgather_ms"] > 0
    assert "dp1 over clouds" in forced["config"]["parallelism"] and forced["scaling"] == "weak"
    plain = _line(_run([sys.executable, "bench.py", "--gpus", "1"] + COMMON, {}))
    assert "comm" not in plain and plain["config"]["parallelism"] == "single GPU"
    assert len(forced["result_digest"]) == 2 and forced["result_digest"] == plain["result_digest"]


def test_bench_world1_patches_sharded_over_rccl_equals_plain_run():
    forced = _line(_run(_torchrun(["bench.py", "--gpus", "1", "--shard", "patches"] + COMMON),
                        {"TPU3_BENCH_FORCE_DIST": "1"}))
    assert forced["comm"]["backend"] == "nccl" and forced["scaling"] == "strong"
    assert "dp1 over outer patches" in forced["config"]["parallelism"]
    one = _line(_run([sys.executable, "bench.py", "--gpus", "1"] + COMMON[:-6] + ["--clouds", "1", "--net_streams", "2",
                                                                                  "--fps_streams", "2"], {}))
    assert forced["result_digest"] == one["result_digest"]


def test_pipeline_sharded_forms_over_rccl_world1():
    out = _run(_torchrun([os.path.join("tests", "_rccl_world1_worker.py")]), {"TPU3_FORCE_COLLECTIVES": "1"})
    res = [l for l in out.splitlines() if l.startswith("RESULT ")]
    assert len(res) == 1, out[-2000:]
    r = json.loads(res[0][7:])
    assert r["backend"] == "nccl" and r["world"] == 1
    assert r["clouds_equal"] and r["patches_equal"] and r["shape"] == [3, 3, 2800]
    # one all-gather per sharded call; the event flags' MAX all-reduce ran as well (check_small path)
    assert r["calls"]["all_gather"] == 2 and r["calls"]["all_reduce"] >= 2
