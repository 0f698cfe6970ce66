#!/usr/bin/env python3
"""Executed SGPR-spill instructions (v_readlane / v_writelane) of k_rollout_lanev<N> per wavefront-step, as a share of its VALU
instructions.

    python tools/spill_exec.py build N          (CPU) the two one-team-size builds this needs, into build/var/
    python tools/spill_exec.py static N         (CPU) where the spill instructions of the step loop are, by REGION of the source
    python tools/spill_exec.py run N [B]        (GPU) how often each region runs per wavefront-step, and the product

Method.  `lines_nN.so` is the shipped kernel compiled with -gline-tables-only: every instruction of the step loop carries a source
line; a spill instruction belongs to the region of rollout_lanev.h its line falls in (instructions of inlined helpers out of
coopsearch.hip and its other headers take the region of the nearest preceding kernel-body line), and is either in the region's straight line or inside an
inner loop of it (machine-code loop structure, as tools/spill_report.py).  `count_nN.so` is the same kernel with -DCS_REGION_COUNTS:
one atomic counter per region entry and per inner-loop iteration (LV_COUNT in rollout_lanev.h), read back after a run of the bench's
lane workload.  executed = sum over regions of static count x measured entries (or iterations) per wavefront-step.  The VALU
instructions per wavefront-step come from the committed SQ_INSTS_VALU pass (profiles/r04_lanev{3,5}_pmc.json)."""
import ctypes, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import spill_report as sr
CSRC = os.path.join(ROOT, "cooperative-search_amd", "csrc")
VAR = os.path.join(ROOT, "build", "var")
# region -> (entry counter, inner-loop iteration counter) of LV_COUNT
REGIONS = {"step": (0, 0), "reset": (1, 2), "refresh": (5, 5), "advance_now": (6, 7), "miss_walk": (8, 9), "fp64_fallback": (10, 10),
           "repulsion": (11, 11), "never": (None, None)}


def build(n):
    os.makedirs(VAR, exist_ok=True)
    for name, flags in ((f"lines_n{n}", ["-gline-tables-only"]), (f"count_n{n}", ["-DCS_REGION_COUNTS"])):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", f"-DCS_ONLY_N={n}"] + flags +
                              ["-I", os.path.join(ROOT, "include"), "coopsearch.hip", "policy.hip", "episodes.hip", "-o", os.path.join(VAR, name + ".so")], cwd=CSRC)
        print("built", name)


def line_regions():
    """[(first line, last line, region)] of rollout_lanev.h, found by the code that opens / closes each region."""
    src = open(os.path.join(CSRC, "rollout_lanev.h")).read().splitlines()
    def find(s, start=0):
        return next(i + 1 for i in range(start, len(src)) if s in src[i])
    k0 = find("void k_rollout_lanev(DevParams p, StepIO io)")
    out = []
    a = find("const unsigned long long need = __ballot(done && auto_reset);", k0); b = find("LANE_STAMP(1);", a)
    out.append((a, b - 1, "reset"))
    a = find("if (__builtin_expect(__ballot(any != 0u) != 0ull, 0)) {", k0); b = find("const unsigned long long hl = lo & ~mlo", a)
    out.append((a, b - 1, "miss_walk"))
    a = find("while (fz) {", k0); b = find("if (i < 4) lo |=", a)
    out.append((a, b - 1, "fp64_fallback"))
    a = find("if (cand >= 0) {", k0); out.append((a, a + 3, "refresh"))
    a = find("while (pend) {   // flight_env_easy.py:293-301"); b = find("const double x = (x0 + p.velocity * c1) + fx;", a)
    out.append((a, b - 1, "repulsion"))
    a = find("__device__ __forceinline__ void lv_advance_finish("); b = find("__device__ __forceinline__ void lv_advance_now(", a)
    out.append((a - 1, b - 2, "refresh"))
    a = b; b = find("template <int N, bool VEC, int WV", a)
    out.append((a - 1, b - 1, "advance_now"))
    return k0, out


def static(n):
    so = os.path.join(VAR, f"lines_n{n}.so")
    k0, regs = line_regions()
    with tempfile.TemporaryDirectory() as tmp:
        text = None
        for co in sr.code_objects(so, tmp):
            out = subprocess.run([f"{sr.LLVM}/llvm-objdump", "-d", "-l", "--demangle", co], capture_output=True, text=True).stdout
            m = re.search(r"^[0-9a-f]{16} <void \(anonymous namespace\)::k_rollout_lanev<%d, true(?:, 2)?>.*?(?=^[0-9a-f]{16} <)" % n, out, re.S | re.M)
            if m:
                text = m.group(0)
    ins, cur = [], None   # (addr, mnemonic, branch target, (file, line))
    start = int(text[:16], 16)
    for ln in text.splitlines()[1:]:
        m = re.match(r"^; (\S+):(\d+)", ln)
        if m:
            cur = (os.path.basename(m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]{12}):(.*)$", ln)
        if not m:
            continue
        mn, addr, rest = m.group(1), int(m.group(3), 16), m.group(4)
        tgt = None
        if mn.startswith("s_cbranch") or mn == "s_branch":
            t = re.search(r"\+0x([0-9a-f]+)>", rest)
            tgt = start + int(t.group(1), 16) if t else None
        ins.append((addr, mn, tgt, cur))
    loops = sorted({(t, a) for a, m, t, _ in ins if t is not None and t <= a}, key=lambda l: (l[0], -l[1]))
    step = max(loops, key=lambda l: l[1] - l[0])
    # blocks of the inlined helpers that no reachable state enters: the off-grid series of trig_heading / trig_heading_pair (headings
    # leave the pi/18 grid only if the raw state is edited), the plain divisions behind div2_same_denominator's guard
    # (round 6: the helpers live in coopsearch.hip and in the headers it includes -- find each block in whichever file holds it)
    files = {f: open(os.path.join(CSRC, f)).read().splitlines() for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))}
    def hfind(s_, fname=None, start=0):
        for f in ([fname] if fname else sorted(files)):
            for i in range(start, len(files[f])):
                if s_ in files[f][i]:
                    return f, i + 1
        raise KeyError(s_)
    never = []   # (file, first line, last line)
    f, a = hfind("if (__builtin_expect((fabs(dh[0]) > 1e-6) | (fabs(dh[1]) > 1e-6), 0)) {"); never.append((f, a, hfind("sa = s[0];", f, a)[1] - 1))
    f, a = hfind("void trig_heading(const double *T, double yaw, double &s, double &c) {"); b = hfind("void load_trig_to_lds", f, a)[1]
    a2 = next(i + 1 for i in range(a, b) if "fabs(dh) > 1e-6" in files[f][i]); never.append((f, a2, b - 2))
    f, a = hfind("    qx = nx / den;"); never.append((f, a, a + 1))
    def region_of(loc):
        if loc and any(loc[0] == f_ and x <= loc[1] <= y for f_, x, y in never):
            return "never"
        if loc and loc[0] == "rollout_lanev.h":
            for a, b, r in regs:
                if a <= loc[1] <= b:
                    return r
            return "step" if loc[1] >= k0 else None
        return None
    res, last = {}, "step"
    n_valu = 0
    for a, mn, _, loc in ins:
        if not (step[0] <= a <= step[1]):
            continue
        r = region_of(loc)
        if r == "never":
            if mn in ("v_readlane_b32", "v_writelane_b32"):
                res[("never", "cold block", "sgpr")] = res.get(("never", "cold block", "sgpr"), 0) + 1
            continue
        if r:
            last = r
        if mn.startswith("v_"):
            n_valu += 1
        if mn not in ("v_readlane_b32", "v_writelane_b32") and not mn.startswith("scratch_"):
            continue
        inner = any(l != step and step[0] <= l[0] and l[1] <= step[1] and l[0] <= a <= l[1] for l in loops)
        if last in ("step", "repulsion", "fp64_fallback"):
            inner = False   # (the step body's loops are unrolled; what the structure shows as loops there are the regions named above)
        key = (last, "inner loop" if inner else "once per entry", "vgpr" if mn.startswith("scratch_") else "sgpr")
        res[key] = res.get(key, 0) + 1
    return res, n_valu, step[1] - step[0]


def run(n, B):
    import numpy as np, torch
    os.environ["COOPSEARCH_LIB"] = os.path.join(VAR, f"count_n{n}.so")
    import cooperative_search_amd as cs
    L = cs.lib.load()
    T, launches = 100, 4
    env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True, kernel="lanev")
    acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
    out = env.rollout(acts)      # first launch: every env starts an episode at once -- not counted
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    L.cs_debug_region_counts.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.cs_debug_region_counts(buf, 1)
    for _ in range(launches):
        env.rollout(acts, out=out, update_views=False)
    torch.cuda.synchronize()
    L.cs_debug_region_counts(buf, 0)
    c = [int(v) for v in buf]
    steps = c[0]
    assert steps == launches * T * (B // 64), (steps, launches * T * (B // 64))
    return {k: (c[e] / steps, c[i] / steps) for k, (e, i) in REGIONS.items() if e is not None}, c


def main():
    what, n = sys.argv[1], int(sys.argv[2])
    if what == "build":
        return build(n)
    res, n_valu, size = static(n)
    print(f"k_rollout_lanev<{n}, true>: step loop {size} B, {n_valu} VALU instructions in it (static)")
    for k in sorted(res):
        print(f"  {res[k]:5d} {k[2]} spill instructions   region {k[0]:14s} {k[1]}")
    if what == "static":
        return
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
    freq, raw = run(n, B)
    pmc = {3: 1776.0, 5: 2917.0}.get(n)   # SQ_INSTS_VALU per wavefront-step (profiles/r04_lanev{3,5}_pmc.json, B = 262144)
    print(f"region frequencies per wavefront-step at B = {B} (entries, inner-loop iterations): " +
          ", ".join(f"{k} {v[0]:.3f} / {v[1]:.3f}" for k, v in freq.items()))
    tot = 0.0
    for (reg, kind, cls), cnt in sorted(res.items()):
        if cls != "sgpr" or reg == "never":
            continue
        f = freq[reg][1 if kind == "inner loop" else 0]
        tot += cnt * f
        print(f"  {reg:14s} {kind:15s} {cnt:4d} x {f:.3f} = {cnt * f:7.2f} executed per wavefront-step")
    print(f"executed v_readlane / v_writelane per wavefront-step: {tot:.1f}" + (f" = {100 * tot / pmc:.2f} % of the {pmc:.0f} VALU instructions "
          f"per wavefront-step (SQ_INSTS_VALU)" if pmc else ""))
    print("raw counters:", raw[:12])


if __name__ == "__main__":
    main()
