#!/usr/bin/env python3
"""sha1 of every kernel's disassembled instruction stream (mnemonics + operands; addresses stripped, symbolic branch targets
dropped) in the gfx950 code objects of a library: `python tools/isa_hashes.py lib.so > a.txt`, then diff two outputs to see which
kernels a source change touched (a refactoring must touch none)."""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from spill_report import LLVM, code_objects  # noqa: E402


def main():
    so = sys.argv[1]
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(so, tmp):
            out = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--demangle", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            cur, h, n, res = None, None, 0, {}
            for ln in out.splitlines():
                m = re.match(r"^([0-9a-f]{16}) <(.*)>:$", ln)
                if m:
                    if cur:
                        res[cur] = (h.hexdigest()[:12], n)
                    cur, h, n = m.group(2), hashlib.sha1(), 0
                    continue
                if cur and ln.strip() and not ln.startswith("Disassembly"):
                    body = ln.split("//")[0].strip()
                    body = re.sub(r"<.*?>", "", body)
                    h.update(body.encode() + b"\n")
                    n += 1
            if cur:
                res[cur] = (h.hexdigest()[:12], n)
            for k in sorted(res):
                print(f"{res[k][0]} {res[k][1]:7d} {k}")


if __name__ == "__main__":
    main()
