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
    hdr_time = max(os.path.getmtime(h) for h in _headers())
    if not force and os.path.exists(LIB) and \
            os.path.getmtime(LIB) >= max([hdr_time] + [os.path.getmtime(f) for f in sources()]):
        return LIB                           # up to date (the objects need not exist, e.g. on the GPU box)
    os.makedirs(OBJ, exist_ok=True)
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


# ---- compiled drop-in extension modules (`import sampling`, `import losses`) ---------------------------
EXT_SRC = os.path.join(CSRC, "ext")
DROPIN = os.path.join(HERE, "dropin")       # put THIS directory on sys.path to get the bare module names


def _dropin_path(name):
    import sysconfig
    return os.path.join(DROPIN, name + sysconfig.get_config_var("EXT_SUFFIX"))


def build_dropin(force=False, verbose=False):
    """g++ -> dropin/sampling.*.so and dropin/losses.*.so: pybind11 modules over torch tensors (host
    code only) that call the C ABI of lib3pu_hip.so; what the reference builds with
    sampling/setup.py and losses/setup.py."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension
    build()
    os.makedirs(DROPIN, exist_ok=True)
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    incs = cpp_extension.include_paths() + ["/opt/rocm/include", sysconfig.get_paths()["include"]]
    deps = [os.path.join(EXT_SRC, "ext_common.h"), os.path.join(HERE, "..", "include", "tpu3.h"),
            os.path.abspath(__file__)]
    outs = []
    jobs = []
    for name in ("sampling", "losses"):
        src = os.path.join(EXT_SRC, name + "_module.cpp")
        out = _dropin_path(name)
        outs.append(out)
        newest = max(os.path.getmtime(f) for f in [src] + deps)
        if not force and os.path.exists(out) and os.path.getmtime(out) >= newest:
            continue
        cmd = (["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
                "-DTORCH_EXTENSION_NAME=" + name, "-DTORCH_API_INCLUDE_EXTENSION_H",
                "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-Wno-deprecated-declarations"]
               + ["-I" + i for i in incs] + [src, "-o", out + ".tmp", "-L" + tlib, "-L" + HERE,
                                              "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_python",
                                              "-l:lib3pu_hip.so", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + tlib])
        jobs.append((cmd, out))
    if jobs:
        from concurrent.futures import ThreadPoolExecutor

        def run(job):
            cmd, out = job
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(out + ".tmp", out)
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(run, jobs))
    return outs


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_dropin(force="--force" in sys.argv, verbose=True))
