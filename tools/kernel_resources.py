#!/usr/bin/env python3
"""Per-kernel VGPR / SGPR / scratch / LDS of the gfx950 code objects inside libcoopsearch_hip.so (reads the AMDGPU
metadata notes; no GPU needed).  usage: python tools/kernel_resources.py [pattern]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
so = os.environ.get("SO") or os.path.join(ROOT, "cooperative-search_amd", "csrc", "libcoopsearch_hip.so")
pat = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as d:
    fat = os.path.join(d, "fat.bin")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    data = open(fat, "rb").read()
    # one clang offload bundle per translation unit, each starting with the magic string
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
    for k, st in enumerate(starts):
        part = os.path.join(d, f"b{k}.bin")
        open(part, "wb").write(data[st:starts[k + 1] if k + 1 < len(starts) else len(data)])
        co = os.path.join(d, f"b{k}.co")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                            f"--output={co}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
        if r.returncode or not os.path.exists(co):
            continue
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0].replace("void ", "")
            if pat and pat not in dem:
                continue
            g = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))
            agpr = int(re.match(r":\s+(\d+)", blk).group(1))
            print(f"{dem:34s} vgpr {g('vgpr_count'):4d} agpr {agpr:3d} sgpr {g('sgpr_count'):4d} "
                  f"sgpr_spill {g('sgpr_spill_count'):4d} vgpr_spill {g('vgpr_spill_count'):4d} "
                  f"scratch {g('private_segment_fixed_size'):5d} lds {g('group_segment_fixed_size'):6d}")
