"""not-gpu: static checks on the gfx950 assembly hipcc produces (cross-compiles without a GPU)."""
import importlib.util
import os
import shutil

import pytest

from conftest import ROOT


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not installed")
def test_hand_issued_mfmas_have_their_wait_states():
    """The 4x4x1 MFMAs of knn.hip are inline asm, outside hipcc's hazard recognizer: no VALU write of an
    A/B operand right before its use, no dependent MFMA straight after its producer (a missing s_nop
    there produced an accumulator that had skipped one product -- see DESIGN.md section 4)."""
    chk = _tool("check_mfma_hazards")
    assert chk.main([os.path.join(ROOT, "3pu_pytorch_amd", "csrc", "knn.hip")]) == 0


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not installed")
@pytest.mark.parametrize("src", ["dense_edge_conv.hip", "mlp.hip", "dec_train.hip"])
def test_no_inline_asm_reads_a_fresh_mfma_result(src):
    """A VALU instruction inside an inline-asm statement must not read a register a (builtin) MFMA wrote a few
    issue slots earlier: hipcc does not insert the wait states for it (tools/check_mfma_hazards.py)."""
    chk = _tool("check_mfma_hazards")
    assert chk.main([os.path.join(ROOT, "3pu_pytorch_amd", "csrc", src)]) == 0
