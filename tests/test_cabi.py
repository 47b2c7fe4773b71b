"""not-gpu: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/tpu3.h declares (no compute calls here: there is no GPU in this container); argument
validation that happens before any launch is exercised through the ctypes binding."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, pkg


def _declared():
    text = open(os.path.join(ROOT, "include", "tpu3.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tpu3_[a-z0-9_]+)\s*\(", text)))


def test_header_functions_are_exported_and_bound():
    L = pkg("_lib")
    pkg("build").build()
    lib = ctypes.CDLL(L.LIB_PATH)
    names = _declared()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), "lib3pu_hip.so does not export %s" % n
    assert sorted(L.SIGNATURES.keys()) == names            # the Python binding covers the header


def test_every_exported_symbol_is_declared():
    """The converse: nothing with C linkage leaves the library that include/tpu3.h does not declare."""
    import subprocess
    L = pkg("_lib")
    pkg("build").build()
    out = subprocess.check_output(["nm", "-D", "--defined-only", L.LIB_PATH]).decode()
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if re.search(r" T tpu3_[a-z0-9_]+$", ln)})
    assert exported, "no tpu3_* symbols found"
    assert set(exported) <= set(_declared()), sorted(set(exported) - set(_declared()))


def test_version_and_strerror():
    L = pkg("_lib")
    assert L.lib().tpu3_version().decode().startswith("3pu-hip")
    assert b"invalid" in L.lib().tpu3_strerror(-1)
    assert L.lib().tpu3_strerror(0) == b"ok"


def test_argument_validation_without_gpu():
    lib = pkg("_lib").lib()
    assert lib.tpu3_fps_f32(None, -1, 4, 2, None, None, None) == -1
    assert lib.tpu3_fps_f32(None, 1, 4, 2, None, None, None) == -1         # NULL pointers
    assert lib.tpu3_fps_f32(None, 0, 4, 2, None, None, None) == 0          # empty batch: no-op
    assert lib.tpu3_gather_fwd(None, 1, 1, 4, 2, 3, 8, 8, 8) == -1         # element size 3
    assert lib.tpu3_knn_f32(None, 1, 4, 8, 3, 9, 8, 8, None, None, None, 8, 8, None, None) == -1  # k > n
    assert lib.tpu3_knn_f32(None, 1, 4, 8, 3, 2, 8, 8, None, None, None, 8, 2, None, None) == -1  # idx size
    # the entry points added for the network stages
    assert lib.tpu3_knn_graph_self_f32(None, 1, 8, 3, 17, 8, None, 8, 8, 8, None, 0) == -1      # k > n
    assert lib.tpu3_knn_graph_self_f32(None, 1, 64, 3, 5, 8, None, 8, 8, 8, None, 0) == -2      # k not 17 / 33
    assert lib.tpu3_knn_unique_compact_i32(None, 2, 8, None, None, None, None, None) == -1      # NULL pointers
    assert lib.tpu3_knn_unique_compact_i32(None, 0, 8, None, None, None, None, None) == 0
    assert lib.tpu3_linear_small_f32(None, 4, 84, 40, 8, 84, 8, None, 1, 8, 40, 0) == -2        # 40 outputs
    assert lib.tpu3_linear_small_f32(None, 4, 82, 24, 8, 84, 8, None, 1, 8, 24, 0) == -2        # cin % 4
    assert lib.tpu3_linear_small_f32(None, 4, 84, 24, 8, 80, 8, None, 1, 8, 24, 0) == -1        # stride < cin
    assert lib.tpu3_linear_small_f32(None, 0, 84, 24, None, 84, None, None, 1, None, 24, 1) == 0
    assert lib.tpu3_linear_small_f32(None, 0, 84, 24, None, 84, None, None, 1, None, 24, 7) == -1   # unknown mfma
    assert lib.tpu3_dense_edge_conv_f32(None, 1, 312, 32, 8, 8, 4, 33, 1, 8, 8, 8, 8, 8, 8, 8, 60, 5) == -1
    # (r5) packed DenseEdgeConv operands
    assert lib.tpu3_dense_edge_conv_pack_floats(0) == 2080 and lib.tpu3_dense_edge_conv_pack_floats(72) == 2080 + 72 * 60
    assert lib.tpu3_dense_edge_conv_pack_floats(36) == 0                                         # fold_n not 0/24/48/72
    assert lib.tpu3_dense_edge_conv_pack_f32(None, 16, 16, 16, 16, 16, 16, 36, 16, 16) == -1
    assert lib.tpu3_dense_edge_conv_pack_f32(None, 16, 16, 16, 16, 16, 16, 24, None, 16) == -1   # fold_n without fold_w
    assert lib.tpu3_dense_edge_conv_pack_f32(None, 16, 16, 16, 16, 16, 16, 0, None, 8) == -1     # blob not 16-byte aligned
    assert lib.tpu3_dense_edge_conv_pk_f32(None, 1, 312, 32, 16, 16, 4, 33, 1, None, 16, 60) == -1   # no blob
    assert lib.tpu3_dense_edge_conv_pk_f32(None, 0, 312, 32, None, None, 4, 33, 1, None, None, 60) == 0
    assert lib.tpu3_dense_edge_conv_pk_f32(None, 1, 4000, 32, 16, 16, 4, 33, 1, 16, 16, 60) == -2    # patch beyond LDS
    assert lib.tpu3_dense_edge_conv_fold_pk_f32(None, 1, 312, 32, 16, 16, 4, 33, 1, 16, 16, 60, 36, 16, 16, 48, 0, 0, 16) == -1
    assert lib.tpu3_debug_skip_fused(1) in (0, 1)
    assert lib.tpu3_linear_wgrad_f32(None, 100, 80, 12, 8, 80, 8, 12, 8, 8, 1 << 20) == -2       # cin > 64
    assert lib.tpu3_linear_wgrad_f32(None, 100, 48, 12, 8, 48, 8, 12, 8, None, 0) == -1          # no workspace
    assert lib.tpu3_linear_wgrad_workspace_bytes(319488) == 1024 * 16 * 64 * 4
    assert lib.tpu3_regress_tail_f32(None, 4, 5, *([8] * 10), 0) == -1                           # r > 4
    assert lib.tpu3_regress_tail_f32(None, 0, 2, *([None] * 10), 1) == 0
    assert lib.tpu3_regress_tail_f32(None, 0, 2, *([None] * 10), 3) == -1                        # unknown mfma
    assert lib.tpu3_interlevel_skip_workspace_bytes(3, 312, 5) == 3 * 312 * 12 * 4
    assert lib.tpu3_interlevel_skip_f32(None, 1, 312, 9, 264, 8, 8, 264, 8, 8, 10, None, 8, 4, 0.2, 0, None, 0) == -1
    assert lib.tpu3_fps_workspace_bytes(4, 1000) == 0 and lib.tpu3_fps_workspace_bytes(4, 30000) > 0
    # (r6) the split-bf16 form of up_layer1's per-point half
    assert lib.tpu3_linear_wide_split_bytes(264) == 9 * 3 * 128 * 32 * 2 and lib.tpu3_linear_wide_split_bytes(0) == 0
    assert lib.tpu3_linear_wide_split_bf16(None, 264, 128, None, 265, 32) == -1                  # no weights
    assert lib.tpu3_linear_wide_split_bf16(None, 264, 64, 32, 265, 32) == -2                     # 64 outputs
    assert lib.tpu3_linear_wide_split_bf16(None, 264, 128, 32, 200, 32) == -1                    # stride < cin
    assert lib.tpu3_linear_wide_sb_f32(None, 0, 264, 128, None, 264, None, None, None, 128) == 0
    assert lib.tpu3_linear_wide_sb_f32(None, 4, 264, 128, 32, 264, None, None, 32, 128) == -1    # no split image
    assert lib.tpu3_linear_wide_sb_f32(None, 4, 260, 128, 32, 264, 32, None, 32, 128) == -2      # cin % 8
    assert lib.tpu3_linear_wide_sb_f32(None, 4, 264, 128, 48, 264, 32, None, 32, 128) == -2      # rows not 32-byte aligned
    assert lib.tpu3_linear_wide_sb_f32(None, 4, 320, 128, 32, 320, 32, None, 32, 128) == -2      # beyond nine slabs
    assert lib.tpu3_linear_wide_sb_f32(None, -1, 264, 128, 32, 264, 32, None, 32, 128) == -1


def test_split_bf16_switch_and_environment():
    """tpu3_split_bf16: query / set / previous value; the initial setting comes from TPU3_SPLIT_BF16 (unset: on -- the
    regressor's default arithmetic since round 6), read once per process."""
    import subprocess
    import sys
    lib = pkg("_lib").lib()
    was = lib.tpu3_split_bf16(-1)
    assert was in (0, 1)
    assert lib.tpu3_split_bf16(0) == was and lib.tpu3_split_bf16(-1) == 0
    assert lib.tpu3_split_bf16(1) == 0 and lib.tpu3_split_bf16(-1) == 1
    lib.tpu3_split_bf16(was)
    code = ("import importlib, sys; sys.path.insert(0, %r); L = importlib.import_module('3pu_pytorch_amd._lib'); "
            "print(L.lib().tpu3_split_bf16(-1))" % ROOT)
    for env_val, expect in ((None, "1"), ("0", "0"), ("1", "1")):
        env = {k: v for k, v in os.environ.items() if k != "TPU3_SPLIT_BF16"}
        if env_val is not None:
            env["TPU3_SPLIT_BF16"] = env_val
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-400:]
        assert out.stdout.strip().splitlines()[-1] == expect, (env_val, out.stdout)


def test_missing_library_fails_loudly(monkeypatch):
    L = pkg("_lib")
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", os.path.join(ROOT, "no_such_dir", "lib3pu_hip.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.lib()


def test_cpu_tensors_are_rejected():
    import torch
    sampling, losses = pkg("sampling"), pkg("losses")
    ops = pkg("network.operations")
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        sampling.gather_forward(1, 1, 4, 2, torch.zeros(1, 1, 4), torch.zeros(1, 2, dtype=torch.int32),
                                torch.zeros(1, 1, 2))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        sampling.ball_query(torch.zeros(1, 2, 3), torch.zeros(1, 4, 3), 0.1, 2)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        losses.nmdistance_forward(torch.zeros(1, 2, 3), torch.zeros(1, 2, 3), torch.zeros(1, 2),
                                  torch.zeros(1, 2), torch.zeros(1, 2, dtype=torch.int32),
                                  torch.zeros(1, 2, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ops.knn_query(2, torch.zeros(1, 4, 3), torch.zeros(1, 8, 3))
    # group_knn is the one operator the reference also calls on host tensors (data.py:135-139): they are staged to the
    # device and searched by the HIP kernel -- there is no CPU implementation behind it, so without a device it fails
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="ROCm device"):
            ops.group_knn(2, torch.zeros(1, 3, 4), torch.zeros(1, 3, 8))


def test_fps_dispatch_table():
    """Which kernel a (b, n, m) call takes -- DESIGN section 4's table, as the dispatcher reports it.  A wrong size
    threshold would silently select a slower kernel: the regimes are pinned here (no launch: runs without a GPU
    through the same plan function the launcher uses)."""
    lib = pkg("_lib").lib()
    lib.tpu3_debug_fps_cluster(-1)
    cl = ctypes.c_int(0)

    def plan(b, n, m):
        k = lib.tpu3_debug_fps_plan(b, n, m, ctypes.byref(cl))
        return k, cl.value
    # per-level resampling of the network (one set per outer patch): a lane per bucket, several samples per round
    assert plan(1536, 6240, 1248)[0] == 2 and plan(1536, 12480, 2496)[0] == 2 and plan(1536, 24960, 4992)[0] == 2
    # up to 25 600 points with few samples (outer / inner patch seeds): the plain register-resident kernel
    assert plan(32, 5000, 48)[0] == 0 and plan(4, 7000, 100)[0] == 0 and plan(2, 20000, 100)[0] == 0
    assert plan(1536, 2496, 40)[0] == 0
    # exactly 4096 points with many samples: rows in registers, one sample per round
    assert plan(48, 4096, 300)[0] == 1
    # the metric's final FPS: one cloud (latency) on 16 members, a sub-batch of 8 on 8, the bench's 32-cloud launch on
    # one workgroup per cloud (two-level tile form)
    assert plan(1, 239616, 80000) == (6, 16) and plan(4, 239616, 80000) == (6, 16)
    assert plan(8, 239616, 80000) == (6, 8) and plan(32, 239616, 80000) == (4, 0)
    # config C5's 3.83 M -> 1.28 M: two levels need 16 members (one or two sets: 32); a batch that cannot have them
    # takes three levels
    assert plan(1, 3833856, 1280000) == (6, 32) and plan(4, 3833856, 1280000) == (6, 16)
    assert plan(8, 3833856, 1280000) == (5, 0)
    # just above the register-resident limit: too few tiles for 16 members
    assert plan(1, 25601, 3000) == (6, 4) and plan(1, 70000, 3000) == (6, 16)
    # beyond every plan
    assert plan(1, 5000000, 1000)[0] == -1
