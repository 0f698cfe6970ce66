"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (cooperative-search_amd/) never does.

`OracleEnv` mirrors the reference env protocol (reset/step/get_obs/get_state,
`/root/reference/env/flight_env_easy.py:14-346`, `env/flight_env.py:14-400`) for
ONE environment; `OracleBatch` drives B independent envs (OpenMP over envs).
"""
import ctypes as C
import fcntl
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORACLE_LIB: tests/test_sanitizers_cpu.py points this at the ASan / UBSan build (`make -C oracle asan`)
_LIB_PATH = os.environ.get("ORACLE_LIB") or os.path.join(_HERE, "liboracle_flight.so")

MAX_AGENTS = 8
MAX_TARGETS = 16


class OrcConfig(C.Structure):
    _fields_ = [
        ("variant", C.c_int32), ("n_agents", C.c_int32), ("n_targets", C.c_int32), ("map_size", C.c_int32),
        ("view_range", C.c_int32), ("time_limit", C.c_int32), ("agent_mode", C.c_int32), ("target_mode", C.c_int32),
        ("velocity", C.c_double), ("safe_dist", C.c_double), ("detect_prob", C.c_double),
        ("force_dist", C.c_double), ("force_factor", C.c_double),
        ("cx", C.c_double * MAX_TARGETS), ("cy", C.c_double * MAX_TARGETS),
        ("dx", C.c_double * MAX_TARGETS), ("dy", C.c_double * MAX_TARGETS),
        ("deter", C.c_int32 * MAX_TARGETS),
    ]


def build(force=False):
    """Compile oracle/liboracle_flight.so with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "flight_oracle.c")
    if os.environ.get("ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "flight_oracle.h")),
            os.path.getmtime(os.path.join(_HERE, "trig_table.inc"))):
        with open(os.path.join(_HERE, ".build.lock"), "w") as lk:  # one builder at a time
            fcntl.flock(lk, fcntl.LOCK_EX)
            try:
                subprocess.check_call(["make", "-C", _HERE, "liboracle_flight.so"], stdout=subprocess.DEVNULL)
            finally:
                fcntl.flock(lk, fcntl.LOCK_UN)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, i32p, f64p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)
        L.orc_create.restype = vp
        L.orc_create.argtypes = [C.POINTER(OrcConfig)]
        L.orc_destroy.argtypes = [vp]
        L.orc_seed.argtypes = [vp, C.c_uint32]
        L.orc_reset.argtypes = [vp, C.c_int]
        L.orc_step.restype = C.c_int
        L.orc_step.argtypes = [vp, i32p, i32p, i32p, i32p]
        L.orc_get_obs.argtypes = [vp, f64p]
        L.orc_get_state.argtypes = [vp, f64p]
        L.orc_get_agents.argtypes = [vp, f64p, f64p, i32p]
        L.orc_set_agents.argtypes = [vp, f64p, f64p]
        L.orc_set_exact_pow.argtypes = [C.c_int]
        L.orc_set_trig_mode.argtypes = [C.c_int]
        L.orc_get_targets.argtypes = [vp, f64p, i32p]
        L.orc_set_targets.argtypes = [vp, f64p, i32p]
        L.orc_get_counters.argtypes = [vp, i32p]
        L.orc_get_prob_map.argtypes = [vp, f64p]
        L.orc_set_prob_map.argtypes = [vp, f64p]
        L.orc_words_consumed.restype = C.c_uint64
        L.orc_words_consumed.argtypes = [vp]
        L.orc_set_draw_log.argtypes = [vp, f64p, C.c_int64]
        L.orc_draw_log_count.restype = C.c_int64
        L.orc_draw_log_count.argtypes = [vp]
        L.orc_clear_draw_log.argtypes = [vp]
        L.orc_rng_u32.restype = C.c_uint32
        L.orc_rng_u32.argtypes = [vp]
        L.orc_rng_rand.restype = C.c_double
        L.orc_rng_rand.argtypes = [vp]
        L.orc_rng_randn.restype = C.c_double
        L.orc_rng_randn.argtypes = [vp]
        L.orc_batch_create.restype = vp
        L.orc_batch_create.argtypes = [C.POINTER(OrcConfig), C.c_int64, C.POINTER(C.c_uint32)]
        L.orc_batch_destroy.argtypes = [vp]
        L.orc_batch_reset.argtypes = [vp, C.c_int, C.POINTER(C.c_uint8), C.c_int]
        L.orc_batch_step.argtypes = [vp, i32p, C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8),
                                     C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int]
        L.orc_batch_rollout.argtypes = [vp, i32p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint8),
                                        C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int,
                                        C.c_int, C.c_int]
        L.orc_batch_rollout_rep.argtypes = [vp, i32p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint8),
                                            C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int,
                                            C.c_int, C.c_int]
        L.orc_batch_env.restype = vp
        L.orc_batch_env.argtypes = [vp, C.c_int64]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


# flight_targets.txt as parsed by the reference's load_targets (main.py:19-32); data, pinned by
# tests/test_oracle_golden.py::test_target_table_matches_fixture
DEFAULT_CIRCLE = dict(
    x=[5, 2, 7.5, 2.8, 6.9, 5.5, 5.3, 1.8, 3, 4.5, 6.3, 8, 0.9, 9.4, 4.2],
    y=[9.1, 7.5, 7, 8, 8.5, 8, 6.6, 6.8, 5.7, 5, 5.7, 6.7, 8.7, 9, 9.3],
    deter=list("ftfftfttfftfftf"),
    priority=[3, 3, 3, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1],
    dx=[0.2, 0.3, 0.3, 0.27, 0.25, 0.25, 0.1, 0.28, 0.18, 0.23, 0.31, 0.29, 0.15, 0.21, 0.34],
    dy=[0.2, 0.3, 0.26, 0.27, 0.25, 0.25, 0.12, 0.28, 0.18, 0.25, 0.30, 0.28, 0.16, 0.21, 0.33],
)


def make_config(variant="flight_easy", n_agents=3, n_targets=15, agent_mode=0, target_mode=0, map_size=50,
                view_range=7, time_limit=200, velocity=1.0, safe_dist=1.0, detect_prob=0.9, force_dist=3.0,
                force_factor=0.8, circle=None):
    circle = circle or DEFAULT_CIRCLE
    cfg = OrcConfig()
    cfg.variant = {"flight_easy": 0, "easy": 0, "flight": 1}[variant]
    cfg.n_agents, cfg.n_targets, cfg.map_size, cfg.view_range = n_agents, n_targets, map_size, view_range
    cfg.time_limit, cfg.agent_mode, cfg.target_mode = time_limit, agent_mode, target_mode
    cfg.velocity, cfg.safe_dist, cfg.detect_prob = velocity, safe_dist, detect_prob
    cfg.force_dist, cfg.force_factor = force_dist, force_factor
    if target_mode == 0:
        for j in range(n_targets):
            cfg.cx[j], cfg.cy[j] = circle["x"][j], circle["y"][j]
            cfg.dx[j], cfg.dy[j] = circle["dx"][j], circle["dy"][j]
            cfg.deter[j] = 1 if circle["deter"][j] == "t" else 0
    return cfg


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def set_trig_mode(mode):
    """0 (default): libm sin/cos, bit-pinned to the reference; 1: correctly rounded table = the HIP kernel's trig."""
    lib().orc_set_trig_mode(int(mode))


class hip_equivalent_arithmetic:
    """Context manager: oracle in the configuration the HIP path must match bit for bit (CR trig, x*x)."""

    def __enter__(self):
        set_trig_mode(1)
        set_exact_pow(False)
        return self

    def __exit__(self, *exc):
        set_trig_mode(0)
        set_exact_pow(True)


def set_exact_pow(on):
    """True (default): square via libm pow(x, 2.0) exactly like the reference's `**2`; False: x*x (faster)."""
    lib().orc_set_exact_pow(1 if on else 0)


class OracleEnv:
    """One environment, reference-shaped API, fp64."""

    def __init__(self, cfg, seed=None):
        self.cfg = cfg
        self.n, self.m, self.L = cfg.n_agents, cfg.n_targets, cfg.map_size
        self.flight = cfg.variant == 1
        self._h = lib().orc_create(C.byref(cfg))
        if not self._h:
            raise ValueError("oracle: bad config")
        self._log = None
        if seed is not None:
            self.seed(seed)

    def __del__(self):
        try:
            if self._h:
                lib().orc_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def seed(self, s):
        lib().orc_seed(self._h, C.c_uint32(int(s)))

    def reset(self, init=False):
        lib().orc_reset(self._h, 1 if init else 0)

    def step(self, actions):
        a = np.ascontiguousarray(np.asarray([int(x) for x in actions], dtype=np.int32))
        if a.shape[0] != self.n:
            raise Exception("Act num mismatch agent")
        r, t, w = C.c_int32(), C.c_int32(), C.c_int32()
        rc = lib().orc_step(self._h, _p(a, C.c_int32), C.byref(r), C.byref(t), C.byref(w))
        if rc:
            raise IndexError("action out of range")
        return int(r.value), bool(t.value), bool(w.value)

    def get_obs(self):
        w = self.L * self.L + 4 if self.flight else 4
        out = np.empty((self.n, w), dtype=np.float64)
        lib().orc_get_obs(self._h, _p(out, C.c_double))
        return out

    def get_state(self):
        out = np.empty(4 * self.n + 3 * self.m, dtype=np.float64)
        lib().orc_get_state(self._h, _p(out, C.c_double))
        return out

    def agents(self):
        """(pos[n,2] fp64, yaw[n] fp64 radians, out_flag[n])"""
        pos = np.empty((self.n, 2), dtype=np.float64)
        yaw = np.empty(self.n, dtype=np.float64)
        out = np.empty(self.n, dtype=np.int32)
        lib().orc_get_agents(self._h, _p(pos, C.c_double), _p(yaw, C.c_double), _p(out, C.c_int32))
        return pos, yaw, out

    def set_agents(self, pos, yaw):
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        yaw = np.ascontiguousarray(yaw, dtype=np.float64)
        lib().orc_set_agents(self._h, _p(pos, C.c_double), _p(yaw, C.c_double))

    def targets(self):
        pos = np.empty((self.m, 2), dtype=np.float64)
        found = np.empty(self.m, dtype=np.int32)
        lib().orc_get_targets(self._h, _p(pos, C.c_double), _p(found, C.c_int32))
        return pos, found

    def set_targets(self, pos, found=None):
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        fp = None
        if found is not None:
            found = np.ascontiguousarray(found, dtype=np.int32)
            fp = _p(found, C.c_int32)
        lib().orc_set_targets(self._h, _p(pos, C.c_double), fp)

    def counters(self):
        o = np.empty(6, dtype=np.int32)
        lib().orc_get_counters(self._h, _p(o, C.c_int32))
        return dict(target_find=int(o[0]), win=int(o[1]), time_step=int(o[2]), total_reward=int(o[3]),
                    curr_reward=int(o[4]), newly_mask=int(o[5]))

    @property
    def target_find(self):
        return self.counters()["target_find"]

    def prob_map(self):
        out = np.empty((self.L, self.L), dtype=np.float64)
        lib().orc_get_prob_map(self._h, _p(out, C.c_double))
        return out

    def set_prob_map(self, m):
        m = np.ascontiguousarray(m, dtype=np.float64)
        lib().orc_set_prob_map(self._h, _p(m, C.c_double))

    def words_consumed(self):
        return int(lib().orc_words_consumed(self._h))

    def start_draw_log(self, cap=1 << 16):
        self._log = np.zeros(cap, dtype=np.float64)
        lib().orc_set_draw_log(self._h, _p(self._log, C.c_double), cap)

    def take_draws(self):
        n = int(lib().orc_draw_log_count(self._h))
        out = self._log[:n].copy()
        lib().orc_clear_draw_log(self._h)
        return out

    # raw generator (known-answer tests against numpy)
    def rng_u32(self):
        return int(lib().orc_rng_u32(self._h))

    def rng_rand(self):
        return float(lib().orc_rng_rand(self._h))

    def rng_randn(self):
        return float(lib().orc_rng_randn(self._h))


class OracleBatch:
    """B independent oracle envs stepped with OpenMP (cpu_baseline leg; batched parity checks)."""

    def __init__(self, cfg, batch, seeds):
        self.cfg, self.B = cfg, int(batch)
        self.n, self.m, self.L = cfg.n_agents, cfg.n_targets, cfg.map_size
        self.flight = cfg.variant == 1
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        assert seeds.shape == (self.B,)
        self._h = lib().orc_batch_create(C.byref(cfg), self.B, _p(seeds, C.c_uint32))
        if not self._h:
            raise ValueError("oracle: bad config")
        self.obs_w = self.L * self.L + 4 if self.flight else 4
        self.reward = np.zeros(self.B, dtype=np.float32)
        self.terminated = np.zeros(self.B, dtype=np.uint8)
        self.win = np.zeros(self.B, dtype=np.uint8)
        self.obs = np.zeros((self.B, self.n, self.obs_w), dtype=np.float32)
        self.state = np.zeros((self.B, 4 * self.n + 3 * self.m), dtype=np.float32)

    def __del__(self):
        try:
            if self._h:
                lib().orc_batch_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @staticmethod
    def max_threads():
        return int(lib().orc_max_threads())

    def reset(self, init=False, mask=None, threads=1):
        mp = None
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            mp = _p(mask, C.c_uint8)
        lib().orc_batch_reset(self._h, 1 if init else 0, mp, threads)

    def step(self, actions, auto_reset=False, freeze_done=True, threads=1, emit=True):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        assert a.shape == (self.B, self.n)
        lib().orc_batch_step(self._h, _p(a, C.c_int32), _p(self.reward, C.c_float), _p(self.terminated, C.c_uint8),
                             _p(self.win, C.c_uint8), _p(self.obs, C.c_float) if emit else None,
                             _p(self.state, C.c_float) if emit else None,
                             1 if auto_reset else 0, 1 if freeze_done else 0, threads)
        return self.reward, self.terminated, self.win

    def rollout(self, actions, auto_reset=False, freeze_done=True, threads=1, out=None, repeat=1):
        """T steps in ONE OpenMP region (env-major): actions [T, B, n] -> dict of [T, B, ...] arrays (reused when
        `out` is a dict from a previous call).  repeat > 1 walks the action table that many times inside the region
        (outputs overwritten): the timed baseline's way of paying the fork/join once per repeat * T steps."""
        a = np.ascontiguousarray(actions, dtype=np.int32)
        T = a.shape[0]
        assert a.shape == (T, self.B, self.n)
        if out is None or out["reward"].shape[0] != T:
            out = dict(reward=np.zeros((T, self.B), np.float32), terminated=np.zeros((T, self.B), np.uint8),
                       win=np.zeros((T, self.B), np.uint8), obs=np.zeros((T, self.B, self.n, self.obs_w), np.float32),
                       state=np.zeros((T, self.B, 4 * self.n + 3 * self.m), np.float32))
        lib().orc_batch_rollout_rep(self._h, _p(a, C.c_int32), T, int(repeat), _p(out["reward"], C.c_float),
                                    _p(out["terminated"], C.c_uint8), _p(out["win"], C.c_uint8), _p(out["obs"], C.c_float),
                                    _p(out["state"], C.c_float), 1 if auto_reset else 0, 1 if freeze_done else 0, threads)
        self.reward, self.terminated, self.win = out["reward"][-1], out["terminated"][-1], out["win"][-1]
        self.obs, self.state = out["obs"][-1], out["state"][-1]
        return out

    def env(self, i):
        """Borrowed single-env view (do not outlive the batch)."""
        e = OracleEnv.__new__(OracleEnv)
        e.cfg, e.n, e.m, e.L, e.flight = self.cfg, self.n, self.m, self.L, self.flight
        e._h = lib().orc_batch_env(self._h, i)
        e._log = None
        e.__class__ = _BorrowedEnv
        return e


class _BorrowedEnv(OracleEnv):
    def __del__(self):
        pass
