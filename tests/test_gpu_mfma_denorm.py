"""The split-fp16 matrix path hands SUBNORMAL halves to v_mfma_f32_16x16x32_f16 (csrc/policy_dev.h, split_f16: hi = fp16(v) whatever
v's size).  This is the hardware statement it rests on: a 16 x 16 x 32 product whose A operand is all subnormal halves and whose
B operand is a third subnormal, against the exact sum in fp64 -- the error must be the fp32 accumulation's, and nothing may be
flushed to zero.  (tools/mfma_f16_denorm.hip, compiled here with hipcc.)"""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_matrix_pipe_multiplies_subnormal_halves_exactly(tmp_path):
    from cooperative_search_amd import build
    hipcc = build.hipcc_path()
    if hipcc is None:
        pytest.skip("hipcc not available")
    exe = tmp_path / "mfma_f16_denorm"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-w", os.path.join(ROOT, "tools", "mfma_f16_denorm.hip"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1000:]
    m = re.search(r"max relative error vs exact ([0-9.e+-]+), zero outputs (\d+) of 256", out.stdout)
    assert m, out.stdout
    assert float(m.group(1)) < 1e-6 and int(m.group(2)) == 0, out.stdout
