#!/usr/bin/env python3
"""Where the register spills of a kernel sit: every scratch_load / scratch_store (VGPR spills) and every v_readlane / v_writelane
(SGPR spills held in VGPR lanes) of the gfx950 code objects inside libcoopsearch_hip.so, classified by the LOOP STRUCTURE of the
machine code -- no GPU needed.

    python tools/spill_report.py [pattern ...]        (default: the kernels cs_rollout / cs_step dispatch to by default)
    SO=build/var/x.so python tools/spill_report.py k_rollout_od

Method: the kernel's instructions are disassembled (llvm-objdump), every backward branch defines a loop [target, branch]; the
"step loop" of a rollout kernel is the largest loop (the K / D / E roles of k_rollout_od: the largest loop of each role, i.e.
every loop longer than STEP_LOOP_MIN bytes that is not nested in another).  A spill instruction is then one of
    step-loop  straight  inside a step loop and in no inner loop that is itself guarded ... (executed once per step)
    step-loop  inner     inside an inner loop of a step loop (reset rounds, top-up loops, waits: executed 0..k times per step)
    outside              prologue / epilogue: executed once per launch
and for the step-loop ones the report says whether the enclosing basic block is reached from the loop head by straight-line
fall-through (hot: no forward branch jumps over it) or sits behind a forward branch that can skip it (conditional: resets,
top-ups, fix-ups, the fp64 fallback of a sensor test ...).  Static: it says where the instructions ARE, not how often they run;
the conditional ones are the cold paths by construction of the kernels.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
STEP_LOOP_MIN = 4096   # bytes: anything shorter is an inner loop (waits, attempt rounds, twist batches)
DEFAULT = ["k_rollout_od<3, true, true, true>", "k_rollout_od<5, true, true, true>", "k_rollout_od<5, true, true, false>",
           "k_rollout_od<3, true, true, false>", "k_rollout_oct<3, true, true>", "k_rollout_oct<5, true, true>",
           "k_rollout_lanev<3, true, 2>", "k_rollout_lanev<3, true, 3>", "k_rollout_lanev<5, true, 2>", "k_rollout_lane<3, true>", "k_rollout_lane<5, true>",
           "k_step<3, 0>", "k_step<3, 1>", "k_flight_pipe<3>", "k_rollout_policy<3>"]


def code_objects(so, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
    for k, st in enumerate(starts):
        part = os.path.join(tmp, f"b{k}.bin")
        open(part, "wb").write(data[st:starts[k + 1] if k + 1 < len(starts) else len(data)])
        co = os.path.join(tmp, f"b{k}.co")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", f"--output={co}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(co):
            yield co


def kernels(co):
    """{demangled short name: [(addr, mnemonic, branch target or None)]}"""
    out = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--demangle", co], capture_output=True, text=True).stdout
    res, cur, start = {}, None, 0
    for ln in out.splitlines():
        m = re.match(r"^([0-9a-f]{16}) <(.*)>:$", ln)
        if m:
            name = re.sub(r"\(anonymous namespace\)::", "", m.group(2))
            name = name.replace("void ", "", 1).split("(")[0]
            cur, start = res.setdefault(name, []), int(m.group(1), 16)
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]{12}):", ln)
        if not m:
            continue
        mnem, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
        tgt = None
        if mnem.startswith("s_cbranch") or mnem == "s_branch":
            t = re.search(r"<.*\+0x([0-9a-f]+)>\s*$", ln)
            tgt = start + int(t.group(1), 16) if t else (start if re.search(r"<[^+]*>\s*$", ln) else None)
        cur.append((addr, mnem, tgt))
    return res


def analyse(name, ins):
    addrs = [a for a, _, _ in ins]
    end = addrs[-1]
    loops = sorted({(t, a) for a, m, t in ins if t is not None and t <= a}, key=lambda l: (l[0], -l[1]))
    big = [l for l in loops if l[1] - l[0] >= STEP_LOOP_MIN]
    step_loops = [l for l in big if not any(o != l and o[0] <= l[0] and l[1] <= o[1] for o in big)]
    fwd = [(a, t) for a, m, t in ins if t is not None and t > a]   # forward branches: [a, t) can be skipped
    rows = []
    for a, m, _ in ins:
        kind = "vgpr" if m.startswith("scratch_") else ("sgpr" if m in ("v_readlane_b32", "v_writelane_b32") else None)
        if not kind:
            continue
        sl = next((l for l in step_loops if l[0] <= a <= l[1]), None)
        if sl is None:
            where = "outside the step loop (once per launch)"
        else:
            inner = [l for l in loops if l != sl and sl[0] <= l[0] and l[1] <= sl[1] and l[0] <= a <= l[1]]
            skipped = [f for f in fwd if sl[0] <= f[0] < a < f[1] <= sl[1] + 8]
            if inner:
                where = "step loop, inside an inner loop (rounds / waits)"
            elif skipped:
                where = "step loop, behind a forward branch (conditional block)"
            else:
                where = "step loop, straight line (every step)"
        rows.append((kind, m, where))
    return step_loops, rows, end - addrs[0]


def main():
    so = os.environ.get("SO") or os.path.join(ROOT, "cooperative-search_amd", "csrc", "libcoopsearch_hip.so")
    pats = sys.argv[1:] or DEFAULT
    with tempfile.TemporaryDirectory() as tmp:
        allk = {}
        for co in code_objects(so, tmp):
            allk.update(kernels(co))
    print(f"# spill report of {os.path.relpath(so, ROOT)} (static: where the instructions are)")
    for pat in pats:
        for name in sorted(k for k in allk if pat in k):
            step_loops, rows, size = analyse(name, allk[name])
            print(f"\n{name}: {size} B of code, step loop(s): " + (", ".join(f"{b - a + 4} B" for a, b in step_loops) or "none"))
            for kind, label in (("vgpr", "VGPR spills (scratch_load / scratch_store)"), ("sgpr", "SGPR spills (v_readlane / v_writelane)")):
                sel = [r for r in rows if r[0] == kind]
                if not sel:
                    print(f"  {label}: none")
                    continue
                print(f"  {label}: {len(sel)} instructions")
                for where in sorted({r[2] for r in sel}):
                    ms = [r[1] for r in sel if r[2] == where]
                    ld = sum(1 for m in ms if "load" in m or "readlane" in m)
                    print(f"    {len(ms):5d}  {where}  ({ld} reloads, {len(ms) - ld} saves)")


if __name__ == "__main__":
    main()
