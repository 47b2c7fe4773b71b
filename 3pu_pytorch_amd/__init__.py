"""3pu_pytorch_amd -- the MI355X (gfx950) native patch-upsampling hot path of 3PU.

The directory name starts with a digit, so import it with
``importlib.import_module("3pu_pytorch_amd")`` (or put this directory on ``sys.path`` /
call :func:`install_dropin` to get the reference's bare module names ``sampling``, ``losses``
and ``network`` -- see INTEGRATION.md).

Layout (only what the hot path needs):
  csrc/            hand-written HIP kernels + the C ABI of include/tpu3.h  -> lib3pu_hip.so
  sampling.py      mirror of the reference's `sampling` extension module
  losses.py        mirror of the reference's `losses` extension module
  network/         operations / layers / upsampler / model_loss (reference call sites)
  pipeline.py      batched patch pipeline (pc_prediction + final FPS), multi-GPU sharding
"""
import importlib
import sys

__version__ = "0.1.0"

_SUBMODULES = ("sampling", "losses", "network", "network.operations", "network.layers",
               "network.upsampler", "network.model_loss")


def install_dropin():
    """Register this package's modules under the reference's bare import names
    (`import sampling`, `import losses`, `from network import operations`, ...)."""
    for name in _SUBMODULES:
        sys.modules[name] = importlib.import_module(__name__ + "." + name)
    return {name: sys.modules[name] for name in _SUBMODULES}
