"""Build recipe of lib3pu_hip.so: hipcc, gfx950 only, in-tree (the .so travels to the GPU box
with the repo snapshot; it is git-ignored, not gpurun-ignored).  hipcc cross-compiles without a
GPU, so this is also the "does it build" check of __graft_entry__.build()."""
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib3pu_hip.so")

# -ffp-contract=off: every fused multiply-add in the kernels is an explicit fmaf(), so the
#   arithmetic is the oracle's operation for operation (bit-exact indices).
# -munsafe-fp-atomics: fp32 atomicAdd lowers to the hardware global_atomic_add_f32.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [
        os.path.join(HERE, "..", "include", "tpu3.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def build(force=False, verbose=False):
    """Compile csrc/*.hip -> lib3pu_hip.so if any source is newer than the library."""
    if not force and not _stale():
        return LIB
    cmd = [hipcc_path()] + HIPCC_FLAGS + ["-o", LIB + ".tmp"] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
