"""The octet pair kernel's K / D / E hand-shakes under perturbed timing: a -DCS_JITTER build inserts pseudo-random
pauses of up to two thirds of a step into every role at every counter read / post, so the wavefronts meet in interleavings the
natural timing never produces; the results must stay bit-identical to the step kernel's."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JITTER_LIB = os.path.join(ROOT, "build", "var", "jitter_n3.so")


def build_jitter_lib():
    """hipcc -DCS_JITTER -DCS_ONLY_N=3 -> build/var/jitter_n3.so (also built by __graft_entry__.build(), so that it travels to
    the GPU box); rebuilt when older than its sources."""
    from cooperative_search_amd import build as b
    srcs = [os.path.join(b.CSRC, s) for s in b.SOURCES] + b.HEADERS
    if os.path.exists(JITTER_LIB) and all(os.path.getmtime(JITTER_LIB) >= os.path.getmtime(s) for s in srcs):
        return JITTER_LIB
    hipcc = b.hipcc_path()
    if hipcc is None:
        return JITTER_LIB if os.path.exists(JITTER_LIB) else None
    os.makedirs(os.path.dirname(JITTER_LIB), exist_ok=True)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DCS_JITTER",
                           "-DCS_ONLY_N=3", "-I", os.path.join(ROOT, "include"), "coopsearch.hip", "policy.hip", "episodes.hip",
                           "-o", JITTER_LIB], cwd=b.CSRC)
    return JITTER_LIB


def test_pair_kernel_handshakes_survive_timing_jitter():
    lib = build_jitter_lib()
    if lib is None:
        pytest.skip("no jitter build and no hipcc")
    env = dict(os.environ, COOPSEARCH_LIB=lib)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "jitter_child.py")], env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert p.stdout.count("bit-identical") == 2, p.stdout
