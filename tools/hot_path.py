#!/usr/bin/env python3
"""The fall-through path of a loop of one disassembled kernel: from the loop head, conditional forward branches are NOT taken
(the compiler lays the expected path out as fall-through: the kernels mark their rare paths with __builtin_expect), unconditional
branches are followed, the walk ends at the backward branch that closes the loop.
    python tools/hot_path.py kernel.s <head offset hex> <tail offset hex> [--list]
Prints the instruction mix of that path (an ESTIMATE of the steady-state step: cold blocks skipped, wave-uniform skips that are
usually taken count as not taken)."""
import collections
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from loop_mix import parse, unit  # noqa: E402


def main():
    ins = parse(sys.argv[1])
    base = ins[0][0]
    head, tail = base + int(sys.argv[2], 16), base + int(sys.argv[3], 16)
    idx = {a: i for i, (a, _, _) in enumerate(ins)}
    i, path, seen = idx[head], [], set()
    while True:
        a, mn, ops = ins[i]
        if a in seen:
            break
        seen.add(a)
        path.append(ins[i])
        if a == tail:
            break
        if mn == "s_branch":
            off = int(ops.split()[-1])
            off -= 65536 if off >= 32768 else 0
            t = a + 4 + 4 * off
            if t in idx and t > a:
                i = idx[t]
                continue
        i += 1
    cnt = collections.Counter(unit(mn) for _, mn, _ in path)
    print(f"{len(path)} instructions on the fall-through path: {dict(cnt)}")
    top = collections.Counter(mn for _, mn, _ in path if unit(mn) == "valu")
    print("VALU:", ", ".join(f"{k} {v}" for k, v in top.most_common(60)))
    if "--list" in sys.argv:
        for a, mn, ops in path:
            print(f"{a - base:#07x}  {mn} {ops}")


if __name__ == "__main__":
    main()
