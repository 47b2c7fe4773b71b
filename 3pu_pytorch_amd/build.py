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
OBJ = os.path.join(HERE, "objs")       # per-file objects (git-ignored)

# -ffp-contract=off: every fused multiply-add in the kernels is an explicit fmaf(), so the
#   arithmetic is the oracle's operation for operation (bit-exact indices).
# -munsafe-fp-atomics: fp32 atomicAdd lowers to the hardware global_atomic_add_f32.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "tpu3.h"),
                                                    os.path.abspath(__file__)]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _compile_one(args):
    src, obj, verbose = args
    cmd = [hipcc_path()] + [f for f in HIPCC_FLAGS if f != "-shared"] + ["-c", src, "-o", obj + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(obj + ".tmp", obj)
    return obj


def build(force=False, verbose=False):
    """Compile csrc/*.hip -> objs/*.o (only the translation units whose source or any header is
    newer than their object, in parallel) and link lib3pu_hip.so."""
    os.makedirs(OBJ, exist_ok=True)
    hdr_time = max(os.path.getmtime(h) for h in _headers())
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            todo.append((src, obj, verbose))
    if todo:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as ex:
            list(ex.map(_compile_one, todo))
    stale = set(os.path.basename(o) for o in objs)
    for f in os.listdir(OBJ):                       # objects of deleted sources must not be linked
        if f.endswith(".o") and f not in stale:
            os.remove(os.path.join(OBJ, f))
    if todo or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
        os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
