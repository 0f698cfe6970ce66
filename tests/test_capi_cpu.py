"""CPU tests of the boundary: the C-ABI library builds, loads and exports every symbol include/*.h declares,
the host-only entry points behave (layout, config errors), and the host-side restatements (target loader, env
constants) match the fixtures.  No compute call is made here (no GPU in the build container)."""
import ctypes as C
import os
import re
import types

import numpy as np
import pytest

import cooperative_search_amd as cs
from cooperative_search_amd import _lib
from golden_util import target_table

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            txt = open(os.path.join(ROOT, "include", fn)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names += re.findall(r"\b(cs_[a-z_0-9]+)\s*\(", txt)
    return sorted(set(names))


def test_library_builds_and_exports_every_declared_symbol():
    L = _lib.load()
    assert os.path.exists(_lib.library_path())
    syms = declared_symbols()
    assert set(syms) == set(_lib.EXPORTS), (syms, _lib.EXPORTS)
    for s in syms:
        assert hasattr(L, s), s
    assert L.cs_abi_version() == _lib.ABI_VERSION == 7


def test_staleness_is_decided_by_source_content_not_mtime(tmp_path, monkeypatch):
    """the .so files travel by copy, so their mtimes mean nothing.  The build records a hash of its sources
    (embedded: cs_source_hash(); beside the library: <lib>.srchash); a newer mtime on a source changes nothing, a changed
    byte does; and where hipcc is absent a library built from other sources is still loaded (with a warning) as long
    as its ABI version matches."""
    from cooperative_search_amd import build
    L = _lib.load()
    build_hash = L.cs_source_hash().decode()
    assert build_hash == build.source_hash() == build._recorded_hash(build.LIB_PATH)
    assert not build.is_stale() and not build.torch_ops_stale()
    src = os.path.join(build.CSRC, "episodes.hip")
    st = os.stat(src)
    try:
        os.utime(src, (st.st_atime, st.st_mtime + 10 ** 6))   # "newer than the .so"
        assert not build.is_stale()
    finally:
        os.utime(src, (st.st_atime, st.st_mtime))
    # other sources than the ones the library was built from
    monkeypatch.setattr(build, "source_hash", lambda: "0" * 16)
    assert build.is_stale()
    monkeypatch.setattr(build, "hipcc_path", lambda: None)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.delenv("COOPSEARCH_STRICT", raising=False)
    with pytest.warns(RuntimeWarning) as rec:
        L2 = _lib.load()
    assert L2.cs_abi_version() == _lib.ABI_VERSION
    msgs = [str(w.message) for w in rec]
    assert any("not built from the sources" in m for m in msgs)            # the sidecar's verdict, before dlopen
    assert any("was built from sources with hash" in m and build_hash in m and "0" * 16 in m for m in msgs)   # the embedded hash's
    # the EMBEDDED hash is the truth (a copied .so can carry any sidecar), and under COOPSEARCH_STRICT=1 -- what
    # tests/conftest.py sets -- a library of other sources is an error: kernels change behaviour without touching the ABI version
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(build, "_recorded_hash", lambda path: "0" * 16)    # a sidecar that agrees with the (fake) sources
    assert not build.is_stale()
    monkeypatch.setenv("COOPSEARCH_STRICT", "1")
    with pytest.raises(_lib.CoopSearchError, match="was built from sources with hash"):
        _lib.load()
    monkeypatch.delenv("COOPSEARCH_STRICT", raising=False)
    # ... and a missing library with no compiler is an error, not a fallback
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(build, "LIB_PATH", str(tmp_path / "libmissing.so"))
    with pytest.raises(_lib.CoopSearchError, match="no CPU fallback"):
        _lib.load()


def test_binding_falls_back_to_ctypes_only_when_torch_was_not_asked_for(monkeypatch):
    """without g++ / torch headers the default binding degrades to ctypes (same library, same kernels) with a
    warning; an explicit binding='torch' raises."""
    assert _lib.pick_binding("ctypes") == ("ctypes", None)
    name, ops = _lib.pick_binding(None)
    assert name == "torch" and ops is _lib.torch_ops()

    def broken():
        raise _lib.CoopSearchError("coopsearch_torch.so is missing and there is no g++ to build it")
    monkeypatch.setattr(_lib, "torch_ops", broken)
    with pytest.warns(RuntimeWarning, match="using the ctypes binding"):
        assert _lib.pick_binding(None) == ("ctypes", None)
    with pytest.raises(_lib.CoopSearchError):
        _lib.pick_binding("torch")
    with pytest.raises(ValueError):
        _lib.pick_binding("pybind")


def _cfg(**kw):
    args = cs.make_env_args(**{k: v for k, v in kw.items() if k in ("env", "n_agents", "agent_mode", "target_mode")})
    from cooperative_search_amd.env import _cfg_from_args
    return _cfg_from_args(args, cs.default_circle_dict(), kw.get("batch", 4096), 1 if kw.get("env") == "flight" else 0)


def test_state_layout_host_only():
    L = _lib.load()
    lay = _lib.CsLayout()
    cfg = _cfg(n_agents=3, batch=4096)
    assert L.cs_state_layout(C.byref(cfg), C.byref(lay)) == 0
    B = 4096
    assert lay.tgt_off == 0 and lay.agent_off == B * 256 and lay.hdr_off == lay.agent_off + B * 256
    assert lay.mt_off == lay.hdr_off + B * 64 and lay.ahead_off == lay.mt_off + B * _lib.MT_STRIDE * 4
    assert lay.tape_off == lay.ahead_off + B * 4 and lay.prob_off == lay.tape_off + B * 64
    assert lay.total_bytes == lay.prob_off == lay.job_off   # flight_easy: no map, no map-update job records
    cfg = _cfg(env="flight", n_agents=3, batch=8192)
    assert L.cs_state_layout(C.byref(cfg), C.byref(lay)) == 0
    assert lay.job_off == lay.prob_off + 8192 * 2500 * 4 and lay.total_bytes == lay.job_off + 2 * 8192 * 256
    for off in (lay.tgt_off, lay.agent_off, lay.hdr_off, lay.mt_off, lay.ahead_off, lay.tape_off, lay.prob_off, lay.job_off):
        assert off % 256 == 0


@pytest.mark.parametrize("field,value,msg", [("n_agents", 9, "n_agents"), ("n_targets", 17, "n_targets"),
                                             ("agent_mode", 4, "No such agent mode"),
                                             ("target_mode", 2, "No such target mode"), ("batch", 0, "batch")])
def test_config_errors_are_reported(field, value, msg):
    L = _lib.load()
    cfg = _cfg(n_agents=3)
    setattr(cfg, field, value)
    lay = _lib.CsLayout()
    assert L.cs_state_layout(C.byref(cfg), C.byref(lay)) == -1
    assert msg in L.cs_last_error().decode()


def test_load_targets_matches_reference_parse():
    t = target_table()
    d = cs.load_targets()
    assert d == t
    assert cs.default_circle_dict() == t


def test_env_constants_match_reference_setters():
    a = cs.make_env_args("flight_easy", n_agents=5)
    assert (a.agent_velocity, a.time_limit, a.safe_dist, a.detect_prob, a.force_dist, a.search_env, a.conv) == \
        (1, 200, 1, 0.9, 3, True, False)
    assert (a.map_size, a.target_num, a.view_range, a.n_agents) == (50, 15, 7, 5)
    f = cs.make_env_args("flight")
    assert f.conv is True and f.conv_out_dim == 16 and f.wrong_alarm_prob == 0.1


def test_env_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cs.BatchedFlightEnv(cs.make_env_args(), batch=4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "cooperative-search_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in txt.replace("the oracle keeps its own copy", "").replace('"oracle", "trig_table.inc"', ""), \
                    f"{fn} mentions the oracle"


@pytest.mark.parametrize("example", ["c_api_demo", "closed_loop_demo"])
def test_c_example_compiles_and_links_against_the_abi(tmp_path, example):
    """examples/*.cpp are the non-Python consumers of the boundary: they must compile against include/coopsearch.h
    and link against the built library (they are executed on the GPU box by the GPU suite)."""
    import shutil
    import subprocess
    from cooperative_search_amd import build
    hipcc = build.hipcc_path()
    if hipcc is None:
        pytest.skip("hipcc not available")
    _lib.load()
    out = tmp_path / example
    csrc = os.path.dirname(_lib.library_path())
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", example + ".cpp"), "-L", csrc, "-lcoopsearch_hip",
                           f"-Wl,-rpath,{csrc}", "-o", str(out)])
    assert out.exists()


def test_policy_pack_layout_host_only():
    """cs_policy_pack is pure host code.  Split-fp16 layout (csrc/policy_dev.h): every weight w travels as hi = fp16(w) (subnormal
    below fp16's normal range) and lo = fp16((w - hi) * 2048); fragment (column tile nt, k-step ks of 32) is the hi plane then the lo
    plane, each [64 lanes][8 halves] with lane l, j holding W[16 nt + (l & 15)][32 ks + 8 (l >> 4) + j] -- the B operand of one
    16x16x32 MFMA --, zero padded; fp32 biases follow.  hi + lo / 2048 reproduces the weight to 22 bits."""
    import ctypes as C
    L = _lib.load()
    rng = np.random.default_rng(3)
    in_dim, nA = 10, 3
    ws = [rng.standard_normal(s).astype(np.float32) for s in
          ((64, in_dim), (64,), (192, 64), (192,), (192, 64), (192,), (64, 64), (64,), (nA, 64), (nA,))]
    ws[2][0, :4] = [1e-6, -3e-5, 6.0e-5, 6.3e-5]   # around fp16's smallest normal number
    n = L.cs_policy_packed_floats()
    packed = np.full(n, np.nan, dtype=np.float32)
    rc = L.cs_policy_pack(*[C.c_void_p(w.ctypes.data) for w in ws], in_dim, nA, C.c_void_p(packed.ctypes.data))
    assert rc == 0
    halves = packed.view(np.float16)
    lanes = np.arange(64)

    def split(w):
        hi = w.astype(np.float16)   # subnormal halves included: the matrix pipe takes them exactly (tests/test_gpu_mfma_denorm.py)
        lo = ((w - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
        return hi, lo

    def want_frag(w, nt, k0, kblocks):
        rows = 16 * nt + (lanes & 15)[:, None]
        ks = k0 + 8 * (lanes >> 4)[:, None] + np.arange(8)[None, :]
        ok = (rows < w.shape[0]) & (ks < w.shape[1]) & ((lanes >> 4)[:, None] < kblocks)
        v = np.where(ok, w[np.minimum(rows, w.shape[0] - 1), np.minimum(ks, w.shape[1] - 1)], np.float32(0)).astype(np.float32)
        return split(v)

    off = 0   # in dwords
    frags = ([(ws[0], nt, 0, 4) for nt in range(4)] +
             [(ws[2], nt, 32 * ks, 4) for nt in range(12) for ks in range(2)] +
             [(ws[4], nt, 32 * ks, 4) for nt in range(12) for ks in range(2)] +
             [(ws[6], nt, 32 * ks, 4) for nt in range(4) for ks in range(2)] +
             [(ws[8], 0, 32 * ks, 4) for ks in range(2)])
    for w, nt, k0, kb in frags:
        hi, lo = want_frag(w, nt, k0, kb)
        got_hi = halves[2 * off: 2 * off + 512].reshape(64, 8)
        got_lo = halves[2 * off + 512: 2 * off + 1024].reshape(64, 8)
        assert np.array_equal(got_hi.view(np.uint16), hi.view(np.uint16)) and np.array_equal(got_lo.view(np.uint16), lo.view(np.uint16))
        back = got_hi.astype(np.float64) + got_lo.astype(np.float64) / 2048.0
        ref = want_frag(w, nt, k0, kb)
        full = ref[0].astype(np.float64) + ref[1].astype(np.float64) / 2048.0
        assert np.array_equal(back, full)
        off += 512
    # the split keeps 22 bits of every weight (11 bits, i.e. an absolute error below 3.1e-8, of one below fp16's normal range)
    hi, lo = split(ws[2])
    err = np.abs(hi.astype(np.float64) + lo.astype(np.float64) / 2048.0 - ws[2].astype(np.float64))
    assert (err <= np.maximum(np.abs(ws[2]) * 2.0 ** -21, 3.1e-8)).all()
    for b, width in ((ws[1], 64), (ws[3], 192), (ws[5], 192), (ws[7], 64), (ws[9], 16)):
        assert np.array_equal(packed[off:off + len(b)], b) and not packed[off + len(b):off + width].any()
        off += width
    assert off == n
    assert L.cs_policy_pack(*[C.c_void_p(w.ctypes.data) for w in ws], 33, nA, C.c_void_p(packed.ctypes.data)) != 0
    assert b"in_dim" in L.cs_policy_last_error()
    # a weight above fp16's range would travel as hi = inf, lo = NaN: refused, by name; the largest half itself passes
    for bad in (70000.0, -1e9, float("inf"), float("nan")):
        w2 = [w.copy() for w in ws]
        w2[4][17, 5] = bad
        assert L.cs_policy_pack(*[C.c_void_p(w.ctypes.data) for w in w2], in_dim, nA, C.c_void_p(packed.ctypes.data)) != 0
        assert b"rnn.weight_hh" in L.cs_policy_last_error() and b"65504" in L.cs_policy_last_error()
    w2 = [w.copy() for w in ws]
    w2[0][3, 1] = -65504.0
    assert L.cs_policy_pack(*[C.c_void_p(w.ctypes.data) for w in w2], in_dim, nA, C.c_void_p(packed.ctypes.data)) == 0


def test_torch_op_library_builds_loads_and_registers_every_op():
    """The PyTorch-ROCm op layer over the C ABI (csrc/torch_ops.cpp) builds in-tree and registers its ops; host-only
    entry points work without a GPU."""
    ops = _lib.torch_ops()
    for name in ("abi_version", "state_bytes", "env_init", "env_seed", "env_reset", "env_step", "env_rollout", "env_emit",
                 "env_metrics", "mt_advance", "mt_canonical", "policy_packed_floats", "policy_forward", "policy_conv_features",
                 "rollout_policy", "rollout_policy_flight", "store_episodes", "epsilon_step"):
        assert hasattr(ops, name), name
    assert int(ops.abi_version()) == _lib.ABI_VERSION
    import torch
    cfg = _cfg(n_agents=3, batch=4096)
    t = torch.frombuffer(bytearray(bytes(cfg)), dtype=torch.uint8)
    lay = _lib.CsLayout()
    assert _lib.load().cs_state_layout(C.byref(cfg), C.byref(lay)) == 0
    assert int(ops.state_bytes(t)) == lay.total_bytes
    with pytest.raises(RuntimeError, match="sizeof"):
        ops.state_bytes(t[:-4].clone())
    with pytest.raises(RuntimeError, match="GPU"):   # no device here: a CPU state tensor must be refused, not dereferenced
        ops.env_init(t, torch.zeros(lay.total_bytes, dtype=torch.uint8))


def test_committed_kernel_resources_describe_the_shipped_binary():
    """profiles/r03_kernel_resources.txt had gone stale (a last commit changed a kernel after the measurement
    set).  profiles/kernel_resources.txt is the CURRENT table -- VGPRs, spills, scratch, LDS of every kernel, read from the
    code objects of the in-tree library -- and this test fails whenever the library was rebuilt from sources that change it:
    regenerate with `python tools/kernel_resources.py > profiles/kernel_resources.txt`."""
    import subprocess
    import sys
    _lib.load()   # (builds the library if the sources changed)
    want = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")], capture_output=True, text=True,
                          check=True).stdout
    path = os.path.join(ROOT, "profiles", "kernel_resources.txt")
    assert os.path.exists(path), "python tools/kernel_resources.py > profiles/kernel_resources.txt"
    have = open(path).read()
    assert sorted(have.split("\n")) == sorted(want.split("\n")), \
        "profiles/kernel_resources.txt is not of this binary: python tools/kernel_resources.py > profiles/kernel_resources.txt"


def test_dispatch_table_matches_the_code():
    """the kernel-dispatch thresholds disagreed between README, DESIGN, INTEGRATION, the header and the code.  ONE
    table now (between the dispatch markers); this test reads its macros and values and compares them with the
    defaults in csrc/coopsearch.hip and with bench.py's kernel_label at every boundary; the other documents must not carry
    thresholds of their own that disagree (the old 131072 / 32768 rollout switches)."""
    import importlib.util
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    table = design[design.index("<!-- dispatch:begin -->"):design.index("<!-- dispatch:end -->")]
    doc = {m.group(1): int(m.group(2)) for m in re.finditer(r"`(CS_[A-Z_]+)` = (\d+)", table)}
    assert set(doc) == {"CS_ODE_UPTO", "CS_OD_UPTO", "CS_OCT_FROM", "CS_LANEV_FROM", "CS_LV_W_FROM", "CS_LANE_FROM_LARGE_TEAMS"}
    csrc = os.path.join(ROOT, "cooperative-search_amd", "csrc")   # coopsearch.hip and the kernel headers it includes
    src = "\n".join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h")))
    for name, value in doc.items():
        m = re.search(r"#ifndef " + name + r"\s*\n#define " + name + r"\s+(\d+)", src)
        assert m and int(m.group(1)) == value, (name, value, m and m.group(1))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    lab = lambda n, B: bench.kernel_label("flight_easy", n, B, "rollout", "auto")
    assert lab(3, doc["CS_ODE_UPTO"]) == "k_rollout_od<3,E>" and lab(3, doc["CS_ODE_UPTO"] + 1) == "k_rollout_od<3>"
    assert lab(5, doc["CS_OD_UPTO"]) == "k_rollout_od<5>" and lab(5, doc["CS_OD_UPTO"] + 1) == "k_rollout_oct<5>"
    assert lab(5, doc["CS_LANEV_FROM"] - 1) == "k_rollout_oct<5>" and lab(5, doc["CS_LANEV_FROM"]) == "k_rollout_lanev<5>"
    assert lab(6, doc["CS_LANEV_FROM"]) == "k_rollout_oct<6>" and lab(6, doc["CS_LANE_FROM_LARGE_TEAMS"]) == "k_rollout_lane<6>"
    for fn in ("README.md", "INTEGRATION.md", os.path.join("include", "coopsearch.h")):
        txt = open(os.path.join(ROOT, fn)).read()
        assert "131072" not in txt, fn + " still names the old lane-kernel threshold; refer to DESIGN.md section 4 instead"


def test_variant_builds_present_carry_the_hash_of_the_present_sources_and_their_flags():
    """the timing-jitter / drained-wait / no-async builds of tests/test_gpu_jitter.py travel to the GPU box as files;
    one compiled from older sources would still equal its own step kernel.  Every variant that is present must carry (embedded
    and in its sidecar) build.variant_hash(flags): regenerate with `python -c "import __graft_entry__ as g; g.build()"`."""
    import test_gpu_jitter as tj
    from cooperative_search_amd import build as b
    present = [n for n in tj.VARIANTS if os.path.exists(b.variant_path(n))]
    if not present:
        pytest.skip("no variant builds in this tree")
    for n in present:
        want = tj.expected_hash(n)
        assert b._recorded_hash(b.variant_path(n)) == want, f"build/var/{n}.so: stale sidecar"
        assert b.embedded_hash(b.variant_path(n)) == want, f"build/var/{n}.so: compiled from other sources or flags"
