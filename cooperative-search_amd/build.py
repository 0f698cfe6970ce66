"""Builds csrc/libcoopsearch_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import contextlib
import fcntl
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB_PATH = os.environ.get("COOPSEARCH_LIB") or os.path.join(CSRC, "libcoopsearch_hip.so")  # override: experiments only
TORCH_LIB_PATH = os.path.join(CSRC, "coopsearch_torch.so")   # torch.ops.coopsearch.*: the op layer over the C ABI
SOURCES = ["coopsearch.hip", "policy.hip", "episodes.hip", "policy_dev.h", "trig_table.inc"]
HEADERS = [os.path.join(ROOT, "include", "coopsearch.h")]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def is_stale():
    if os.environ.get("COOPSEARCH_LIB"):
        return False
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


@contextlib.contextmanager
def _build_lock():
    """One builder at a time (bench.py under torchrun starts N ranks at once); the others wait, then re-check."""
    path = os.path.join(CSRC, ".build.lock")
    with open(path, "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def build_extension(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> csrc/libcoopsearch_hip.so.  -ffp-contract=off is part of the numerics
    contract, not a tuning knob."""
    if not force and not is_stale():
        return LIB_PATH
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build libcoopsearch_hip.so")
    with _build_lock():
        if not force and not is_stale():  # another process built it while we waited
            return LIB_PATH
        tmp = f"{LIB_PATH}.tmp.{os.getpid()}"
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-I", os.path.join(ROOT, "include"), os.path.join(CSRC, "coopsearch.hip"), os.path.join(CSRC, "policy.hip"), os.path.join(CSRC, "episodes.hip"), "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
        os.replace(tmp, LIB_PATH)
    return LIB_PATH


def torch_ops_stale():
    if not os.path.exists(TORCH_LIB_PATH):
        return True
    t = os.path.getmtime(TORCH_LIB_PATH)
    deps = [os.path.join(CSRC, "torch_ops.cpp")] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build_torch_ops(force=False, verbose=False):
    """g++ -> csrc/coopsearch_torch.so: the thin PyTorch-ROCm op layer (csrc/torch_ops.cpp; host code only, it links
    libcoopsearch_hip.so through an $ORIGIN rpath and torch's libraries)."""
    if not force and not torch_ops_stale():
        return TORCH_LIB_PATH
    build_extension()
    import torch
    from torch.utils import cpp_extension as ce
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("g++ not found: cannot build coopsearch_torch.so")
    tlib = ce.library_paths()[0]
    with _build_lock():
        if not force and not torch_ops_stale():
            return TORCH_LIB_PATH
        tmp = f"{TORCH_LIB_PATH}.tmp.{os.getpid()}"
        cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
               f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-w",
               "-I", os.path.join(ROOT, "include")]
        for inc in ce.include_paths() + ["/opt/rocm/include"]:
            cmd += ["-I", inc]
        cmd += [os.path.join(CSRC, "torch_ops.cpp"), "-o", tmp, "-L", tlib, "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu",
                "-ltorch_hip", "-L", CSRC, "-l:" + os.path.basename(LIB_PATH), "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
        os.replace(tmp, TORCH_LIB_PATH)
    return TORCH_LIB_PATH


if __name__ == "__main__":
    print(build_extension(force=True, verbose=True))
    print(build_torch_ops(force=True, verbose=True))
