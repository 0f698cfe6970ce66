"""ctypes binding of libcoopsearch_hip.so (include/coopsearch.h).  No fallback: a missing library raises."""
import ctypes as C
import os
import warnings

from . import build as _build

ABI_VERSION = 7   # CS_ABI_VERSION of include/coopsearch.h; bumped whenever an export or a struct changes
MT_STRIDE = 672    # CS_MT_STRIDE
MAX_AGENTS = 8
MAX_TARGETS = 16
H_WORDS = 16
# header word indices (enum CS_H_* in coopsearch.h)
H_FOUND, H_NEWLY, H_TARGET_FIND, H_FLAGS, H_TIME_STEP, H_TOTAL_REWARD, H_MT_POS, H_EPISODES = range(8)
H_WORDS_LO, H_WORDS_HI, H_CURR_REWARD, H_NEWLY_RESET = 8, 9, 10, 11
SELECT_SOFTMAX, SELECT_SAMPLE = 1, 2
FREEZE_DONE, AUTO_RESET, ACTIONS_I64, KERNEL_GROUP, KERNEL_LANE, KERNEL_SOLO, KERNEL_DUO, KERNEL_OCT, KERNEL_OD, KERNEL_ODE, KERNEL_LANEV, CHECK_ACTIONS = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048

EXPORTS = ["cs_abi_version", "cs_source_hash", "cs_has_legacy_kernels", "cs_last_error", "cs_state_layout", "cs_init", "cs_seed", "cs_reset", "cs_step",
           "cs_rollout", "cs_rollout_policy", "cs_rollout_policy_flight", "cs_emit", "cs_metrics", "cs_mt_canonical", "cs_mt_advance", "cs_policy_packed_floats", "cs_policy_pack", "cs_policy_forward",
           "cs_policy_conv_features", "cs_policy_last_error", "cs_store_episodes", "cs_episodes_last_error", "cs_epsilon_step"]


class CsConfig(C.Structure):
    _fields_ = [
        ("variant", C.c_int32), ("n_agents", C.c_int32), ("n_targets", C.c_int32), ("map_size", C.c_int32),
        ("view_range", C.c_int32), ("time_limit", C.c_int32), ("agent_mode", C.c_int32), ("target_mode", C.c_int32),
        ("velocity", C.c_double), ("safe_dist", C.c_double), ("detect_prob", C.c_double),
        ("force_dist", C.c_double), ("force_factor", C.c_double),
        ("cx", C.c_double * MAX_TARGETS), ("cy", C.c_double * MAX_TARGETS),
        ("dx", C.c_double * MAX_TARGETS), ("dy", C.c_double * MAX_TARGETS),
        ("deter", C.c_int32 * MAX_TARGETS),
        ("batch", C.c_int64),
    ]


class CsLayout(C.Structure):
    _fields_ = [("total_bytes", C.c_size_t), ("tgt_off", C.c_size_t), ("agent_off", C.c_size_t),
                ("hdr_off", C.c_size_t), ("mt_off", C.c_size_t), ("ahead_off", C.c_size_t), ("tape_off", C.c_size_t), ("prob_off", C.c_size_t),
                ("job_off", C.c_size_t)]


class CsEpisodeOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("o", "u", "s", "r", "o_next", "s_next", "avail_u", "avail_u_next", "u_onehot",
                                          "padded", "terminated")]


class CsEpsilon(C.Structure):
    """cs_epsilon of include/coopsearch.h: the exploration schedule of common/rollout.py:35-41,75-76,133-135."""
    _fields_ = [("epsilon", C.c_double), ("anneal", C.c_double), ("min_epsilon", C.c_double), ("per_step", C.c_int32),
                ("reserved", C.c_int32), ("eps_dev", C.c_void_p), ("trace_dev", C.c_void_p)]


class CoopSearchError(RuntimeError):
    pass


_lib = None


def library_path():
    return _build.LIB_PATH


def load():
    """Load (building first if the in-tree .so is missing or was built from other sources).  Staleness is decided by the
    source hash recorded at build time (build.source_hash), not by mtimes; where hipcc is absent a library whose recorded
    hash differs is still loaded -- with a warning -- as long as its ABI version matches (checked below)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if _build.is_stale():
        if _build.hipcc_path() is None:
            if not os.path.exists(path):
                raise CoopSearchError(
                    f"{path} is missing and hipcc is not available: the HIP extension is required (no CPU fallback)")
            warnings.warn(f"{path} was not built from the sources present here (recorded source hash "
                          f"{_build._recorded_hash(path)!r}, sources {_build.source_hash()!r}) and hipcc is not available to "
                          "rebuild it: loading it as it is (the ABI version is checked)", RuntimeWarning, stacklevel=2)
        else:
            _build.build_extension()
    L = C.CDLL(path)
    vp = C.c_void_p
    L.cs_abi_version.restype = C.c_int
    L.cs_last_error.restype = C.c_char_p
    if hasattr(L, "cs_source_hash"):
        L.cs_source_hash.restype = C.c_char_p
    L.cs_state_layout.argtypes = [C.POINTER(CsConfig), C.POINTER(CsLayout)]
    L.cs_init.argtypes = [C.POINTER(CsConfig), vp, vp]
    L.cs_seed.argtypes = [C.POINTER(CsConfig), vp, vp, vp]
    L.cs_reset.argtypes = [C.POINTER(CsConfig), vp, vp, C.c_int, vp, vp, vp]
    L.cs_step.argtypes = [C.POINTER(CsConfig), vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.cs_rollout.argtypes = [C.POINTER(CsConfig), vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    L.cs_rollout_policy.argtypes = [C.POINTER(CsConfig), vp, vp, vp, vp, C.c_int, C.c_int, C.POINTER(CsEpsilon), C.c_uint64, C.c_uint32,
                                    C.c_uint64, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    L.cs_rollout_policy_flight.argtypes = [C.POINTER(CsConfig)] + [vp] * 11 + [C.c_int, C.c_int, C.POINTER(CsEpsilon), C.c_uint64, C.c_uint32,
                                           C.c_uint64, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    L.cs_epsilon_step.argtypes = [C.POINTER(CsConfig), vp, C.c_int, vp, C.c_double, C.c_double, vp, vp]
    L.cs_emit.argtypes = [C.POINTER(CsConfig), vp, vp, vp, vp]
    L.cs_metrics.argtypes = [C.POINTER(CsConfig), vp, vp, vp]
    L.cs_mt_canonical.argtypes = [C.POINTER(CsConfig), vp, vp, vp]
    L.cs_mt_advance.argtypes = [C.POINTER(CsConfig), vp, C.c_int, vp]
    L.cs_policy_packed_floats.restype = C.c_size_t
    L.cs_policy_last_error.restype = C.c_char_p
    L.cs_episodes_last_error.restype = C.c_char_p
    L.cs_store_episodes.argtypes = [C.c_int] * 6 + [vp] * 6 + [C.POINTER(CsEpisodeOut), vp]
    L.cs_policy_pack.argtypes = [vp] * 10 + [C.c_int, C.c_int, vp]
    L.cs_policy_forward.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int,
                                    C.c_float, vp, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, vp]
    L.cs_policy_conv_features.argtypes = [vp] * 7 + [C.c_int64, C.c_int, vp, vp]
    for name in EXPORTS:
        fn = getattr(L, name)
        if name not in ("cs_abi_version", "cs_source_hash", "cs_has_legacy_kernels", "cs_last_error", "cs_policy_packed_floats", "cs_policy_last_error",
                        "cs_episodes_last_error"):
            fn.restype = C.c_int
    if L.cs_abi_version() != ABI_VERSION:
        raise CoopSearchError(f"{path}: ABI version {L.cs_abi_version()}, this package binds version {ABI_VERSION} "
                              "(stale library: rebuild with `python -m cooperative_search_amd.build`)")
    # The hash COMPILED INTO the library is the truth about what it was built from (the .srchash sidecar above is only the
    # shortcut that decides about a rebuild before anything is dlopen'ed; a copied .so can carry a wrong or no sidecar).  The ABI
    # version guards exports and structs only -- kernels change behaviour without touching either -- so a library of other sources
    # is an error under COOPSEARCH_STRICT=1 (tests, CI) and a warning naming both hashes otherwise.  COOPSEARCH_LIB (experimental
    # one-team-size builds) opts out.
    if os.environ.get("COOPSEARCH_LIB"):
        # a variant build says what it should have been compiled from (build.variant_hash: sources AND flags) through
        # COOPSEARCH_LIB_HASH; without that variable the opt-out stands (ad-hoc experiments)
        want = os.environ.get("COOPSEARCH_LIB_HASH")
        built = L.cs_source_hash().decode() if hasattr(L, "cs_source_hash") else ""
        if want and built != want:
            raise CoopSearchError(f"{path} carries source hash {built!r}, expected {want!r}: a stale variant build "
                                  "(rebuild it: cooperative_search_amd.build.build_variant)")
    else:
        built = L.cs_source_hash().decode() if hasattr(L, "cs_source_hash") else ""
        want = _build.source_hash()
        if built != want:
            msg = (f"{path} was built from sources with hash {built!r}, the sources present here hash to {want!r}: "
                   "kernels may behave differently from what the tests restate (rebuild with `python -m cooperative_search_amd.build`)")
            if os.environ.get("COOPSEARCH_STRICT") == "1":
                raise CoopSearchError(msg)
            warnings.warn(msg, RuntimeWarning, stacklevel=2)
    _lib = L
    return L


_ops = None


def torch_ops():
    """torch.ops.coopsearch (csrc/torch_ops.cpp): tensor-level ops over the same C ABI -- device / dtype / contiguity
    checks in C++ (TORCH_CHECK), stream = torch's current HIP stream, the tensors' device made current.  Builds
    coopsearch_torch.so (g++: host code only) first when it is missing or was built from other sources; raises when it can
    neither be built nor loaded -- `pick_binding` turns that into the ctypes route for callers that did not ask for torch."""
    global _ops
    if _ops is not None:
        return _ops
    import torch
    load()   # libcoopsearch_hip.so first: coopsearch_torch.so links it
    if _build.torch_ops_stale():
        if _build.cxx_path() is None:
            if not os.path.exists(_build.TORCH_LIB_PATH):
                raise CoopSearchError(f"{_build.TORCH_LIB_PATH} is missing and there is no g++ to build it")
            warnings.warn(f"{_build.TORCH_LIB_PATH} was not built from the torch_ops.cpp present here and there is no g++ "
                          "to rebuild it: loading it as it is (the ABI version is checked)", RuntimeWarning, stacklevel=2)
        else:
            _build.build_torch_ops()
    torch.ops.load_library(_build.TORCH_LIB_PATH)
    if int(torch.ops.coopsearch.abi_version()) != ABI_VERSION:
        raise CoopSearchError(f"{_build.TORCH_LIB_PATH}: ABI version mismatch (stale library)")
    _ops = torch.ops.coopsearch
    return _ops


def pick_binding(binding=None):
    """'torch' | 'ctypes' | None -> (name, torch.ops.coopsearch or None).  None = the torch op layer when it can be built /
    loaded, else the ctypes route with a warning (same library, same kernels); an experimental library (COOPSEARCH_LIB) is
    only reachable through ctypes.  An explicit 'torch' raises when the op library is unavailable."""
    if binding not in (None, "torch", "ctypes"):
        raise ValueError("binding must be 'torch' (torch.ops.coopsearch, csrc/torch_ops.cpp) or 'ctypes'")
    if binding == "ctypes" or (binding is None and os.environ.get("COOPSEARCH_LIB")):
        return "ctypes", None
    try:
        return "torch", torch_ops()
    except Exception as exc:   # noqa: BLE001 -- compiler missing, torch headers missing, dlopen failure, ABI mismatch
        if binding == "torch":
            raise
        warnings.warn(f"torch.ops.coopsearch is unavailable ({type(exc).__name__}: {exc}); using the ctypes binding of the "
                      "same library", RuntimeWarning, stacklevel=3)
        return "ctypes", None


def has_legacy_kernels():
    """cs_has_legacy_kernels() of the loaded library: False since round 6 (the 16-lane rollout kernels of rounds 1-2 were removed; the
    entry point stays for ABI 7)."""
    L = load()
    L.cs_has_legacy_kernels.restype = C.c_int
    return bool(L.cs_has_legacy_kernels())


OP_NO_CHECK_ACTIONS = 1 << 30   # torch op layer only (csrc/torch_ops.cpp): 'the caller decided: no check'; never reaches the C ABI


def check(rc):
    if rc != 0:
        msg = load().cs_last_error().decode()
        if msg.startswith("list index out of range"):   # CS_CHECK_ACTIONS: the reference's IndexError (dyaw[act], flight_env_easy.py:262)
            raise IndexError(msg)
        raise CoopSearchError(f"coopsearch error {rc}: {msg}")


def check_actions_default(batch):
    """CS_CHECK_ACTIONS is on by default for batches of up to 64 envs; COOPSEARCH_CHECK_ACTIONS=0 / 1 forces it off / on
    (same rule as csrc/torch_ops.cpp:check_actions_default)."""
    e = os.environ.get("COOPSEARCH_CHECK_ACTIONS", "")
    return batch <= 64 if e == "" else e[0] != "0"
