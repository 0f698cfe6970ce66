"""Builds csrc/libcoopsearch_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import contextlib
import fcntl
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB_PATH = os.environ.get("COOPSEARCH_LIB") or os.path.join(CSRC, "libcoopsearch_hip.so")  # override: experiments only
# torch.ops.coopsearch.*: the op layer over the C ABI (override: the sanitizer build of tests/test_sanitizers_cpu.py)
TORCH_LIB_PATH = os.environ.get("COOPSEARCH_TORCH_LIB") or os.path.join(CSRC, "coopsearch_torch.so")
SOURCES = ["coopsearch.hip", "rollout_policy.h", "rollout_lane.h", "rollout_oct.h", "rollout_od.h", "rollout_lanev.h",
           "flight_map.h", "policy.hip", "episodes.hip", "policy_dev.h", "trig_table.inc"]
HEADERS = [os.path.join(ROOT, "include", "coopsearch.h")]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def _hash_files(paths):
    h = hashlib.sha256()
    for d in paths:
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def source_hash():
    """Hash of everything libcoopsearch_hip.so is compiled from.  The build embeds it (cs_source_hash()) and writes it
    next to the library (<lib>.srchash): staleness is decided by CONTENT, never by mtime -- the .so files are git-ignored
    and travel by copy (rsync, gpurun snapshot, checkout), so their mtimes relative to the sources mean nothing."""
    return _hash_files([os.path.join(CSRC, s) for s in SOURCES] + HEADERS)


def torch_ops_source_hash():
    return _hash_files([os.path.join(CSRC, "torch_ops.cpp")] + HEADERS)


def _recorded_hash(lib_path):
    try:
        with open(lib_path + ".srchash") as f:
            return f.read().strip()
    except OSError:
        return None


def is_stale():
    """True when the in-tree library is missing or was built from other sources than the ones present (recorded hash
    differs or is absent)."""
    if os.environ.get("COOPSEARCH_LIB"):
        return False
    if not os.path.exists(LIB_PATH):
        return True
    return _recorded_hash(LIB_PATH) != source_hash()


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
        digest = source_hash()
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               f'-DCS_SOURCE_HASH="{digest}"', "-I", os.path.join(ROOT, "include"), os.path.join(CSRC, "coopsearch.hip"), os.path.join(CSRC, "policy.hip"), os.path.join(CSRC, "episodes.hip"), "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
        os.replace(tmp, LIB_PATH)
        with open(LIB_PATH + ".srchash", "w") as f:
            f.write(digest + "\n")
    return LIB_PATH


# ---- experimental / test builds of the library for ONE team size (build/var/<name>.so, used through COOPSEARCH_LIB) ----------
VAR_DIR = os.path.join(ROOT, "build", "var")


def variant_path(name):
    return os.path.join(VAR_DIR, name + ".so")


def variant_hash(flags, only_n):
    """What a variant build is compiled from: the sources' hash AND its flags.  Embedded in the variant (cs_source_hash()) and
    written next to it; a variant is current when both equal this -- never decided by mtime (the .so files travel by copy)."""
    h = hashlib.sha256()
    h.update(source_hash().encode() + b"\0" + " ".join(list(flags) + [f"-DCS_ONLY_N={only_n}"]).encode())
    return h.hexdigest()[:16]


def embedded_hash(lib_path):
    """cs_source_hash() of a built library, read in a child process (dlopen of several HIP libraries in one process registers
    their code objects side by side; a child keeps the caller clean)."""
    code = ("import ctypes, sys; L = ctypes.CDLL(sys.argv[1]); L.cs_source_hash.restype = ctypes.c_char_p; "
            "print(L.cs_source_hash().decode())")
    r = subprocess.run([shutil.which("python3") or "python3", "-c", code, lib_path], capture_output=True, text=True)
    return r.stdout.strip() if r.returncode == 0 else None


def build_variant(name, flags, only_n=3, force=False):
    """hipcc <flags> -DCS_ONLY_N=<n> -DCS_SOURCE_HASH=<variant_hash> -> build/var/<name>.so (+ .srchash).  Rebuilt when the
    recorded hash is not variant_hash(flags, only_n).  Without hipcc: the existing file (the caller compares its embedded
    hash) or None."""
    lib = variant_path(name)
    want = variant_hash(flags, only_n)
    if not force and os.path.exists(lib) and _recorded_hash(lib) == want:
        return lib
    hipcc = hipcc_path()
    if hipcc is None:
        return lib if os.path.exists(lib) else None
    os.makedirs(VAR_DIR, exist_ok=True)
    tmp = f"{lib}.tmp.{os.getpid()}"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"] + list(flags) +
                          [f"-DCS_ONLY_N={only_n}", f'-DCS_SOURCE_HASH="{want}"', "-I", os.path.join(ROOT, "include"),
                           "coopsearch.hip", "policy.hip", "episodes.hip", "-o", tmp], cwd=CSRC)
    os.replace(tmp, lib)
    with open(lib + ".srchash", "w") as f:
        f.write(want + "\n")
    return lib


def torch_ops_stale():
    if os.environ.get("COOPSEARCH_TORCH_LIB"):
        return False
    if not os.path.exists(TORCH_LIB_PATH):
        return True
    return _recorded_hash(TORCH_LIB_PATH) != torch_ops_source_hash()


def cxx_path():
    return shutil.which("g++") or shutil.which("c++")


def build_torch_ops(force=False, verbose=False):
    """g++ -> csrc/coopsearch_torch.so: the thin PyTorch-ROCm op layer (csrc/torch_ops.cpp; host code only, it links
    libcoopsearch_hip.so through an $ORIGIN rpath and torch's libraries)."""
    if not force and not torch_ops_stale():
        return TORCH_LIB_PATH
    build_extension()
    import torch
    from torch.utils import cpp_extension as ce
    cxx = cxx_path()
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
        with open(TORCH_LIB_PATH + ".srchash", "w") as f:
            f.write(torch_ops_source_hash() + "\n")
    return TORCH_LIB_PATH


if __name__ == "__main__":
    print(build_extension(force=True, verbose=True))
    print(build_torch_ops(force=True, verbose=True))
