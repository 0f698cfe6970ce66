#!/usr/bin/env python3
"""Instruction mix of the big loops of one disassembled kernel (llvm-objdump -d --no-show-raw-insn text of ONE kernel):
    python tools/loop_mix.py kernel.s [min_bytes]
For every outermost loop longer than min_bytes (default 4096): bytes, instruction counts by unit, and the VALU mnemonics by count --
STATIC counts (what the loop holds, cold blocks included), to see what a role's step is made of."""
import collections
import re
import sys


def parse(path):
    ins = []
    for ln in open(path).read().splitlines()[1:]:
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]{12}):", ln)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return ins


def loops_of(ins):
    out = []
    for a, mn, ops in ins:
        if mn.startswith("s_cbranch") or mn == "s_branch":
            try:
                off = int(ops.split()[-1])
            except ValueError:
                continue
            if off >= 32768:
                off -= 65536
            t = a + 4 + 4 * off
            if t <= a:
                out.append((t, a))
    return out


def unit(mn):
    if mn.startswith("v_"):
        return "valu"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "vmem"
    if mn.startswith("s_load") or mn.startswith("s_buffer"):
        return "smem"
    return "salu"


def main():
    ins = parse(sys.argv[1])
    minb = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    lp = loops_of(ins)
    big = [l for l in lp if l[1] - l[0] >= minb]
    outer = [l for l in big if not any(o != l and o[0] <= l[0] and l[1] <= o[1] for o in big)]
    for t, a in sorted(set(outer)):
        body = [(x, mn, ops) for x, mn, ops in ins if t <= x <= a]
        inner = [l for l in lp if t <= l[0] and l[1] <= a and l != (t, a)]
        def in_inner(x):
            return any(i0 <= x <= i1 for i0, i1 in inner)
        cnt = collections.Counter(unit(mn) for _, mn, _ in body)
        cnt_s = collections.Counter(unit(mn) for x, mn, _ in body if not in_inner(x))
        print(f"loop {t - ins[0][0]:#x}..{a - ins[0][0]:#x}  {a - t} B  {len(body)} instructions {dict(cnt)}; outside inner loops: {dict(cnt_s)}")
        top = collections.Counter(mn for x, mn, _ in body if unit(mn) == "valu" and not in_inner(x))
        print("   VALU outside inner loops:", ", ".join(f"{k} {v}" for k, v in top.most_common(40)))


if __name__ == "__main__":
    main()
