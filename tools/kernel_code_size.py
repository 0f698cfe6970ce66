#!/usr/bin/env python3
"""Code size (bytes of machine code) of every kernel inside a libcoopsearch_hip.so: the symbol sizes of the gfx950 code objects.
usage: [SO=path] python tools/kernel_code_size.py [pattern]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
so = os.environ.get("SO") or os.path.join(ROOT, "cooperative-search_amd", "csrc", "libcoopsearch_hip.so")
pat = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as d:
    fat = os.path.join(d, "fat.bin")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
    for k, st in enumerate(starts):
        part = os.path.join(d, f"b{k}.bin")
        open(part, "wb").write(data[st:starts[k + 1] if k + 1 < len(starts) else len(data)])
        co = os.path.join(d, f"b{k}.co")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", f"--output={co}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
        if r.returncode or not os.path.exists(co):
            continue
        out = subprocess.run([f"{LLVM}/llvm-readelf", "-sW", "--demangle", co], capture_output=True, text=True).stdout
        for ln in out.splitlines():
            m = re.match(r"\s*\d+:\s+[0-9a-f]+\s+(\d+)\s+FUNC\s+\S+\s+\S+\s+\S+\s+(.*)", ln)
            if m and pat in m.group(2):
                name = re.sub(r"\(anonymous namespace\)::", "", m.group(2)).split("(")[0].replace("void ", "")
                print(f"{name:40s} {int(m.group(1)):8d} B")
