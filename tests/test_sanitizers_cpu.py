"""Sanitizer runs on the CPU box (never on the GPU pool, which refuses GPU ASan):

* the C oracle built with `make -C oracle asan` (gcc -fsanitize=address,undefined) replays golden traces of the reference
  -- the checker itself has no out-of-bounds access, no signed overflow, no misaligned access on those inputs;
* the HOST half of the product -- the host section of csrc/coopsearch.hip / policy.hip / episodes.hip (config checks,
  layout, parameter blocks, weight packing: everything that runs before a launch) and csrc/torch_ops.cpp -- built with
  clang's ASan + UBSan (`hipcc -fsanitize=address,undefined -fno-gpu-sanitize`: device code untouched) and driven by the
  host-only tests of tests/test_capi_cpu.py.

Each run is a child process with the sanitizer runtime preloaded (python itself is not instrumented); gcc's and clang's
runtimes cannot share a process, hence two children.
"""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAD = ("ERROR: AddressSanitizer", "runtime error:", "SUMMARY: UndefinedBehaviorSanitizer", "SUMMARY: AddressSanitizer")


def _run(cmd, env_extra, timeout=1500):
    env = dict(os.environ)
    env.update(env_extra)
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:halt_on_error=1"   # CPython "leaks" by design
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    return p, p.stdout + p.stderr


def test_oracle_replays_goldens_under_asan_ubsan():
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "oracle", "liboracle_flight_asan.so")
    rt = subprocess.check_output([gcc, "-print-file-name=libasan.so"], text=True).strip()
    assert os.path.exists(lib) and os.path.isabs(rt) and os.path.exists(rt)
    # every golden trace of both variants (walls, pokes, past-done, three-episode streams, maps), the MT19937 / randn
    # stream tests and the batch entry points bench.py's cpu_baseline uses
    p, out = _run([sys.executable, "-m", "pytest", "tests/test_oracle_golden.py", "tests/test_properties_cpu.py", "-q", "-x",
                   "-p", "no:cacheprovider"], {"LD_PRELOAD": rt, "ORACLE_LIB": lib})
    assert p.returncode == 0 and " passed" in out, out[-3000:]
    assert not any(b in out for b in BAD), out[-3000:]


def _clang_asan_runtime(hipcc):
    llvm = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm")
    for base in (llvm, "/opt/rocm/lib/llvm"):
        hits = sorted(glob.glob(os.path.join(base, "lib", "clang", "*", "lib", "linux", "libclang_rt.asan-x86_64.so")))
        if hits:
            return hits[-1], os.path.join(base, "bin", "clang++")
    return None, None


def test_host_half_of_the_product_under_asan_ubsan(tmp_path):
    from cooperative_search_amd import build
    hipcc = build.hipcc_path()
    if hipcc is None:
        pytest.skip("hipcc not available")
    rt, clangxx = _clang_asan_runtime(hipcc)
    if rt is None or not os.path.exists(clangxx):
        pytest.skip("clang ASan runtime not found")
    import torch
    from torch.utils import cpp_extension as ce
    out_dir = os.path.join(ROOT, "build", "asan")
    os.makedirs(out_dir, exist_ok=True)
    hip_lib = os.path.join(out_dir, "libcoopsearch_hip.so")
    csrc = build.CSRC
    # one team size (-DCS_ONLY_N) keeps the device compile short: the host code is the same for every size
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-gpu-sanitize",
                           "-DCS_ONLY_N=3", "-I", os.path.join(ROOT, "include"), os.path.join(csrc, "coopsearch.hip"),
                           os.path.join(csrc, "policy.hip"), os.path.join(csrc, "episodes.hip"), "-o", hip_lib], cwd=out_dir)
    torch_lib = os.path.join(out_dir, "coopsearch_torch.so")
    tlib = ce.library_paths()[0]
    cmd = [clangxx, "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address,undefined", "-shared-libsan",
           "-fno-sanitize-recover=undefined",
           "-mllvm", "-asan-globals=0",   # libstdc++ string literals shared with the uninstrumented torch trip the ODR check
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           "-w", "-I", os.path.join(ROOT, "include")]
    for inc in ce.include_paths() + ["/opt/rocm/include"]:
        cmd += ["-I", inc]
    cmd += [os.path.join(csrc, "torch_ops.cpp"), "-o", torch_lib, "-L", tlib, "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu",
            "-ltorch_hip", "-L", out_dir, "-l:libcoopsearch_hip.so", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    subprocess.check_call(cmd, cwd=out_dir)
    # the host-only tests of the boundary against the instrumented libraries (the staleness / binding-choice / example tests
    # are about the in-tree build, not about host code paths)
    p, out = _run([sys.executable, "-m", "pytest", "tests/test_capi_cpu.py", "-q", "-x", "-p", "no:cacheprovider", "-k",
                   "not staleness and not binding_falls and not example"],
                  {"LD_PRELOAD": rt, "COOPSEARCH_LIB": hip_lib, "COOPSEARCH_TORCH_LIB": torch_lib})
    assert p.returncode == 0 and " passed" in out, out[-3000:]
    assert not any(b in out for b in BAD), out[-3000:]
