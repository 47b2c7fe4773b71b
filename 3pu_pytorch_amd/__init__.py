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

_SUBMODULES = ("network", "network.operations", "network.layers", "network.upsampler", "network.model_loss")


def _compiled_dropin(name):
    """The pybind11 extension module `name` from dropin/ (built by build.build_dropin), or None."""
    import importlib.util
    import os
    import sysconfig
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin",
                        name + sysconfig.get_config_var("EXT_SUFFIX"))
    if not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def install_dropin(compiled=True):
    """Register this package's modules under the reference's bare import names, so that the
    reference's own files run unchanged on top of them (`import sampling`, operations.py:6;
    `import losses`, model_loss.py:2; `from network import operations`, ...).
    `sampling` / `losses` are the compiled pybind11 extension modules (csrc/ext/, the counterpart of
    what sampling/setup.py and losses/setup.py build) when they have been built and `compiled` is
    true, else the ctypes mirrors sampling.py / losses.py -- both are thin bindings of the same C ABI
    of lib3pu_hip.so.  Returns {name: module}."""
    out = {}
    for name in ("sampling", "losses"):
        mod = _compiled_dropin(name) if compiled else None
        if mod is None:
            mod = importlib.import_module(__name__ + "." + name)
        sys.modules[name] = out[name] = mod
    for name in _SUBMODULES:
        sys.modules[name] = out[name] = importlib.import_module(__name__ + "." + name)
    return out
