#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched flight_easy / flight environment path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5] [--mode step|rollout]

One "step" = one env.step() of EVERY environment of the batch, including the obs + state emission that
common/rollout.py:45-63 needs each step.  Inputs (pre-drawn uniform actions) are resident in HBM before the
timed region.  Done environments are auto-reset (CS_AUTO_RESET), so every environment does a full step every
time: value = batch x K / time.

Workloads (BASELINE.json configs):
    c2 (default, N = 1 headline)  flight_easy, 3 agents, 15 targets, 4096 envs per GPU
    c3                            flight_easy, 5 agents, 15 targets, 16384 envs per GPU
    c4                            flight (probability map), 3 agents, 15 targets, 8192 envs per GPU
    c5                            flight_easy, 5 agents, 15 targets, 8192 envs per GPU (65536 over 8 GPUs)
Modes:
    step     one cs_step launch per step (the closed-loop path a policy drives), replayed from a hipGraph
    rollout  cs_rollout over an open-loop action table, 100 steps per call (default).  flight_easy: up to 100 steps per
             launch with the env resident in registers (kernel by batch: k_rollout_od up to 16384 envs, k_rollout_oct
             below 131072, k_rollout_lane from there); flight: one launch per step in which the map sweep of step t and
             the kinematics / detection of step t + 1 run side by side (k_flight_pipe)

Timing protocol (one clock): after W warm-up steps, the K-step region -- bracketed by a barrier and
torch.cuda.synchronize() on both sides -- is repeated until it has accumulated >= MIN_GPU_S of GPU time (at least
5 times); each repeat is timed with HIP events recorded on the stream the kernels are launched on (torch's current
stream).  `value`, `ms_per_step` and `roofline.achieved` all derive from the MEDIAN event interval of the K-step
region (max over ranks); the host wall clock of the same repeats is reported next to it as `wall_ms_per_step`.

Multi-GPU: `--gpus N` with no torchrun environment starts N fresh rank processes itself (python -m
torch.distributed.run, one rank per GPU, backend nccl = RCCL) BEFORE anything touches the GPU, relays rank 0's JSON
line and exits with the children's return code; under an external torchrun (RANK / WORLD_SIZE set) it is a rank.
The batch is sharded by global env index with no data-path collective ("scaling": "weak": per-GPU batch fixed); the
only collective is the all-gather of the evaluation-metric partials after the timed region (runner.py:86-96), whose
`eval.envs` = N x batch proves that RCCL saw N ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "c2": dict(env="flight_easy", n_agents=3, batch=4096),
    "c3": dict(env="flight_easy", n_agents=5, batch=16384),
    "c4": dict(env="flight", n_agents=3, batch=8192),
    "c5": dict(env="flight_easy", n_agents=5, batch=8192),
}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E
C5_GLOBAL_BATCH = 65536  # BASELINE.json config 5
MIN_GPU_S = 1.0          # headline: repeat the K-step region until this much GPU time has been measured
MIN_GPU_S_ALSO = 0.25    # secondary workloads
MIN_REPEATS, MAX_REPEATS = 5, 20000
PG_WATCHDOG_S = 120      # --pg auto at N = 1: how long the optional RCCL communicator may take to come up


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default=None, choices=["step", "rollout"])
    ap.add_argument("--batch", type=int, default=None, help="override envs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads reported under 'also'")
    ap.add_argument("--also", action="store_true",
                    help="N > 1: run the whole list of secondary workloads on every rank too (default at N > 1: the headline and "
                         "the two c5 lines only, so that a scaling run is not 8x longer than it has to be)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "group", "od", "ode", "oct", "lane", "lanev"],
                    help="flight_easy kernel: 16 lanes per env (group: a rollout is T launches of the step kernel), 8 lanes per env "
                         "(od: kinematics + detection wavefront pair, oct: one wavefront), one lane per env, or everything by "
                         "batch size (auto: ode up to 8192 envs, od up to 16384, oct below 65536 -- teams of 6 to 8: below 2^20 --, lanev from 65536 "
                         "for teams of up to 5, lane from 2^20 for larger ones; DESIGN.md section 4)")
    ap.add_argument("--min-gpu-s", type=float, default=MIN_GPU_S)
    ap.add_argument("--pg", default="auto", choices=["auto", "on", "off"],
                    help="process group at N = 1: 'on' = init_process_group('nccl') even for one rank and fail if RCCL does "
                         "not come up; 'auto' (default) = try it, report the failure in the line and carry on without "
                         "(the collective is outside the timed region); 'off' = none.  N > 1 always has one.")
    ap.add_argument("--force-pg", dest="pg", action="store_const", const="on", help="same as --pg on")
    ap.add_argument("--cpu-baseline-child", default=None, help=argparse.SUPPRESS)   # "env,n,batch,budget": see cpu_baseline()
    ap.add_argument("--no-numa-bind", action="store_true",
                    help="do not bind the rank to the CPUs of its GPU's NUMA node (bound by default when sysfs tells)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU-only check of the N-rank control flow (launcher, gloo process group, metric all-gather); "
                         "no kernels run and the line says so")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------- the one stdout line
MAX_LINE = 6000          # hard ceiling of the stdout line in characters (a consumer that keeps the last 8 KB still parses it)
CPU_SAMPLE_MAX = 300     # cpu_baseline.sample in the line; the long form is in the detail file
DETAIL_NAME = "bench_detail.json"


def sig(x, digits=6):
    """Floats of the side entries rounded to `digits` significant digits (the headline keeps every digit)."""
    if isinstance(x, float) and math.isfinite(x) and x != 0.0:
        return float(f"{x:.{digits}g}")
    if isinstance(x, (list, tuple)):
        return [sig(v, digits) for v in x]
    if isinstance(x, dict):
        return {k: sig(v, digits) for k, v in x.items()}
    return x


def compact_line(full, detail_path=None):
    """The line that goes to stdout, built from the full record: the contract's headline fields untouched, `config`, `timing`,
    `roofline`, `eval`, `per_rank_value` and `cpu_baseline` cut down to their numbers, and every secondary workload as one
    `also_summary` pair {key: [env-steps/s, roofline fraction]}.  The full record (every `also` entry with its own roofline
    object, the CPU baseline's thread scaling and pinning) goes to `detail_path`.  Never longer than MAX_LINE characters:
    optional blocks are dropped, least important first, until it fits (r05's 21.8 KB line was cut by its consumer)."""
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                 "scaling", "vs_baseline", "dtype", "data") if k in full}
    if full.get("dry_run"):
        line["dry_run"] = True
    cfg = dict(full.get("config") or {})
    if isinstance(cfg.get("cpu_binding"), dict):
        cfg["cpu_binding"] = {k: cfg["cpu_binding"][k] for k in ("numa_node", "cpus") if k in cfg["cpu_binding"]}
    if "backend_error" in cfg:
        cfg["backend_error"] = str(cfg["backend_error"])[:160]
    line["config"] = cfg
    if "timing" in full:
        t = full["timing"]
        line["timing"] = {"clock": "HIP events, launch stream; median of repeats, max over ranks", "repeats": t["repeats"],
                          "timed_gpu_s": sig(t["timed_gpu_s"]), "region_ms_min_median_max": sig(t["region_ms_min_median_max"]),
                          "wall_ms_per_step": sig(t["wall_ms_per_step"])}
    if "roofline" in full:
        r = full["roofline"]
        line["roofline"] = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel",
                                             "algorithmic_bytes_per_env_step", "algorithmic_bytes_per_launch", "avg_launch_us",
                                             "context")
                            if k in r}
        src = r.get("traffic_source")
        line["roofline"]["traffic_source"] = src.split(":")[0] + " (committed --pmc passes, scaled)" if src else None
    if "eval" in full:
        e = dict(full["eval"])
        if isinstance(e.get("allgather_us"), dict):
            e["allgather_us"] = sig({k: e["allgather_us"][k] for k in ("median", "min", "max", "reps", "floats", "error")
                                     if k in e["allgather_us"]})
        e["reduced_by"] = "all_gather" if str(e.get("reduced_by", "")).startswith("all_gather") else e.get("reduced_by")
        line["eval"] = sig(e)
    if full.get("per_rank_value"):
        line["per_rank_value"] = sig({k: v for k, v in full["per_rank_value"].items() if k != "unit"}, 7)
    for key in ("also_at_this_n", "c5_strong_total", "c5_weak_total"):
        if key in full:
            line[key] = sig(full[key]) if isinstance(full[key], dict) else full[key]
    if full.get("also"):
        line["also_summary"] = {"_": "key: [env-steps/s, HBM roofline fraction (null: closed loop)]"}
        for ent in full["also"]:
            frac = (ent.get("roofline") or {}).get("frac")
            line["also_summary"][ent.get("key") or str(ent.get("workload"))[:24]] = [sig(ent.get("value"), 4), sig(frac, 3)]
    if full.get("cpu_baseline"):
        c = full["cpu_baseline"]
        line["cpu_baseline"] = {"value": sig(c.get("value"), 7), "unit": c.get("unit"), "cores": c.get("cores"),
                                "kind": c.get("kind"), "single_thread_value": sig(c.get("single_thread_value"), 7),
                                "sample": str(c.get("sample_short") or c.get("sample"))[:CPU_SAMPLE_MAX]}
    if detail_path:
        line["detail"] = detail_path
    for victim in ("also_summary", "per_rank_value", "timing", "c5_weak_total", "c5_strong_total", "eval"):
        if len(json.dumps(line)) <= MAX_LINE:
            break
        line.pop(victim, None)
        line["dropped_to_fit"] = line.get("dropped_to_fit", []) + [victim]
    if len(json.dumps(line)) > MAX_LINE:
        raise RuntimeError(f"bench.py: the stdout line is {len(json.dumps(line))} characters (> {MAX_LINE})")
    return line


def _fits(line):
    if len(json.dumps(line)) > MAX_LINE:
        raise RuntimeError(f"bench.py: the stdout line is {len(json.dumps(line))} characters (> {MAX_LINE})")
    return line


def write_detail(full):
    """The full record next to the run: gpurun_out/bench_detail.json under the repo root (merged back by gpurun), the system's
    temporary directory when that is not writable.  Returns the path as the line names it (relative to the repo root when
    inside it), or None."""
    import tempfile
    for d in (os.environ.get("BENCH_DETAIL_DIR"), os.path.join(ROOT, "gpurun_out"), tempfile.gettempdir()):
        if not d:
            continue
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, DETAIL_NAME)
            with open(path, "w") as fh:
                json.dump(full, fh, indent=1)
            return os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT + os.sep) else path
        except OSError:
            continue
    return None



# ------------------------------------------------------------------------------------------------- N-rank launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(a, argv):
    """`--gpus N` outside torchrun: start N fresh rank processes (this process never touches the GPU), relay rank 0's
    JSON line on stdout and everything else on stderr, return the children's exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True)
    line = None
    for out in proc.stdout:
        s = out.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the rank processes printed no JSON line\n")
        rc = 1
    if line is not None:
        print(line, flush=True)
    return rc


# -------------------------------------------------------------------------------------------------- byte models
def algorithmic_bytes_per_env_step(env, n, m, mode):
    """SURVEY.md section 8(d).  flight_easy, one launch per step: 61n + 22m + 22 (535 B at 3a15t, 657 B at 5a15t);
    fused T-step rollout (state resident in registers): 36n + 12m + 6 (294 B / 366 B);
    flight: obs n*(2500+4)*4 + state 4*(4n+3m) + ONE read of the 10 000-byte map + the flight_easy remainder
    (40 535 B at 3a15t).  This is the model that matches k_map: it writes the map back only where a cell changed
    (data dependent, not counted); SURVEY's full-map streaming model (+10 000 B write-back) would be 50 535 B."""
    if env == "flight":
        return n * 2504 * 4 + 4 * (4 * n + 3 * m) + 10000 + (61 * n + 22 * m + 22 - 16 * n - 4 * (4 * n + 3 * m))
    if mode == "rollout":
        return 36 * n + 12 * m + 6
    return 61 * n + 22 * m + 22


def largest_divisor_leq(k, cap):
    for d in range(min(cap, k), 0, -1):
        if k % d == 0:
            return d
    return 1


def kernel_label(env_name, n, B, mode, kernel):
    """The kernel cs_step / cs_rollout dispatches to (csrc/coopsearch.hip: use_lane_kernel, use_od_kernel, use_oct_kernel)."""
    lane_from = (65536 if n <= 5 else 1048576) if mode == "rollout" else 32768   # lane_from() (rollout); single steps: 32768
    lane = env_name == "flight_easy" and (kernel in ("lane", "lanev") or (kernel == "auto" and B >= lane_from))
    if env_name == "flight":   # rollout call: step t + 1 rides inside the map sweep of step t, one launch per step
        return f"k_flight_pipe<{n}>" if mode == "rollout" else f"k_step<{n},1> + k_map<{n}>"
    if lane:   # teams of up to 5: the second-generation kernel (targets in registers) unless the first is asked for
        return f"k_rollout_lanev<{n}>" if (n <= 5 and kernel != "lane") else f"k_rollout_lane<{n}>"
    if mode == "step":
        return f"k_step<{n},0>"
    if kernel == "ode" or (kernel == "auto" and B <= 8192):        # CS_ODE_UPTO: K + D + emitting wavefront per 8 envs
        return f"k_rollout_od<{n},E>"
    if kernel == "od" or (kernel == "auto" and B <= 16384):      # CS_OD_UPTO
        return f"k_rollout_od<{n}>"
    if kernel == "oct" or kernel == "auto":                         # CS_OCT_FROM < B < CS_LANE_FROM
        return f"k_rollout_oct<{n}>"
    return f"k_step<{n},0>"   # kernel == "group": T launches of the 16-lane step kernel


def pmc_traffic(label, B, steps_per_launch):
    """(HBM bytes per launch, source) from the COMMITTED PMC passes (profiles/traffic.json: FETCH_SIZE / WRITE_SIZE in
    separate rocprofv3 --pmc passes, per env-step), scaled to THIS run's batch and steps per launch.  Not measured in this
    run -- counters need their own profiler passes -- so the line names the entry it came from.  (None, None) when no pass
    of this kernel AT THIS NUMBER OF STEPS PER LAUNCH is committed: prologue traffic per env-step depends on the launch
    length, so a pass taken at another length is not scaled across."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None, None
    kernels = json.load(open(path)).get("kernels") or {}
    key = f"{label}@{steps_per_launch}"
    ent = kernels.get(key)
    if ent and int(ent.get("steps_per_launch", -1)) == steps_per_launch:
        return int(ent["hbm_bytes_per_env_step"] * B * steps_per_launch), f"profiles/traffic.json[{key!r}]"
    return None, None


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_list_str(cpus):
    """{0,1,2,3,8,9} -> '0-3,8-9' (the kernel's cpulist notation)."""
    cs_, out, i = sorted(cpus), [], 0
    while i < len(cs_):
        j = i
        while j + 1 < len(cs_) and cs_[j + 1] == cs_[j] + 1:
            j += 1
        out.append(str(cs_[i]) if i == j else f"{cs_[i]}-{cs_[j]}")
        i = j + 1
    return ",".join(out)


def physical_cores():
    """(number of physical cores this process may run on, threads per core): sibling lists of the CPUs in the affinity mask."""
    cpus = sorted(os.sched_getaffinity(0))
    cores = set()
    for c in cpus:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        cores.add(sib)
    return max(1, len(cores)), max(1, round(len(cpus) / max(1, len(cores))))


STEADY_SPAN = 1.25   # p90 / p10 of the per-region rates below which a thread count counts as a repeatable baseline
QUOTA_HEADROOM = 2   # CPUs of a cgroup quota the baseline's team leaves unused (see cpu_baseline_inproc)


def cpu_quota():
    """CPUs' worth of time the container's cgroup grants (cpu.max of cgroup v2, cfs_quota_us / cfs_period_us of v1), or None
    when unlimited / unreadable.  A gpurun box shows 256 logical CPUs and a quota of 16: a team of more threads than that runs
    until the quota of the 100 ms period is spent and is then stopped for the rest of it -- regions 4x slower than the median."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if quota <= 0 else quota / period
    except (OSError, ValueError):
        return None


def cpu_baseline(env_name, n, batch, budget_s=10.0):
    """The CPU baseline in a CHILD process with a pinned OpenMP team: OMP_PROC_BIND=close and OMP_PLACES=cores have to be in
    the environment before libgomp initialises, and the parent has long loaded its OpenMP runtimes (torch's, numpy's) and
    carries the GPU runtime's helper threads.  The child never touches the GPU: it imports numpy and the oracle only.
    Unpinned, the same measurement was bimodal (p10 / p90 of the per-region rates 8x apart: threads migrating between
    cores and SMT siblings in the middle of a region)."""
    env = dict(os.environ)
    env.update({"OMP_PROC_BIND": "close", "OMP_PLACES": "cores", "OMP_DYNAMIC": "false"})
    env.pop("OMP_NUM_THREADS", None)
    # the rank was bound to the CPUs of its GPU's NUMA node (bind_rank_to_gpu_node) and a child inherits that mask: the
    # "host's cores" would be one node's.  The child restores the mask the process had before the binding.
    if ORIGINAL_AFFINITY is not None:
        env["BENCH_CPU_BASELINE_AFFINITY"] = ",".join(str(c) for c in sorted(ORIGINAL_AFFINITY))
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", f"{env_name},{n},{batch},{budget_s}"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or not lines:
        raise RuntimeError(f"cpu baseline child failed (rc {out.returncode}): {out.stderr[-500:]}")
    return json.loads(lines[-1])


def cpu_baseline_inproc(env_name, n, batch, budget_s=10.0):
    """The C oracle (a port of the reference's algorithm, parity-pinned by tests/) timed on this host's cores on a
    bounded sample of the same workload: same batch, auto-reset, obs+state emission, x*x squares (its fast mode).
    flight_easy: orc_batch_rollout_rep -- ONE OpenMP region per 400 steps (the 100-step action table walked four times),
    env-major, so the fork/join cost is paid once per 400 steps and an env stays in its core's cache.  The team is pinned
    (see cpu_baseline): one thread per PHYSICAL core at most -- SMT siblings are never used -- and never more threads than the
    cgroup's CPU quota less QUOTA_HEADROOM (cpu_quota); thread counts from 1 to that limit are calibrated first (median of
    three regions each); the three best share the time budget and `value` is the MEDIAN per-region rate of the fastest one
    whose regions span less than STEADY_SPAN between p10 and p90 (`cores` = that thread count); every measured point is listed
    with its p10 / median / p90.  The envs of a region are handed out dynamically, 64 at a time (whole cache lines of every
    output), so one descheduled thread costs its current chunk and not the whole region."""
    import numpy as np
    mask0 = set(os.sched_getaffinity(0))   # before libgomp binds the master thread to its place (OMP_PROC_BIND)
    from oracle import oracle as orc
    n_cores, smt = physical_cores()
    quota = cpu_quota()
    # two CPUs' worth of the quota stay free for the parent (its GPU runtime and RCCL helper threads keep polling while it waits
    # for this child) and the box's own daemons: with a team of exactly `quota` threads the group still ran into the throttle
    # now and then (16 threads of a 16-CPU quota: region rates spanning 1.02x in a bare child, 1.24-1.28x under bench.py)
    cap = max(1, int(quota) - QUOTA_HEADROOM) if quota and quota >= 1 else n_cores
    max_threads = max(1, min(orc.OracleBatch.max_threads(), n_cores, cap))
    flight = env_name == "flight"
    B = batch if not flight else min(batch, 256)
    T, R = (100, 4) if not flight else (10, 1)
    cfg = orc.make_config(variant=env_name, n_agents=n)
    seeds = (20240000 + np.arange(B)).astype(np.uint32)
    orc.set_exact_pow(False)
    try:
        ob = orc.OracleBatch(cfg, B, seeds)
        warm = min(max_threads, 16)   # warm-up on a small team: the full one would leave its idle threads behind in the pool
        ob.reset(init=True, threads=warm)
        acts = np.random.RandomState(1).randint(0, 3, size=(T, B, n)).astype(np.int32)
        out = ob.rollout(acts, auto_reset=True, freeze_done=False, threads=warm)   # warm-up + buffers

        def regions(th, reps, rep=R):
            """`reps` OpenMP regions of T * rep steps on `th` threads, each timed on its own: env-steps/s per region."""
            rates = []
            for _ in range(reps):
                t0 = time.perf_counter()
                ob.rollout(acts, auto_reset=True, freeze_done=False, threads=th, out=out, repeat=rep)
                rates.append(B * T * rep / (time.perf_counter() - t0))
            return rates

        cands = sorted({1, 2, 4, 8, 16, 32, 64, 128, max_threads // 2, max_threads} & set(range(1, max_threads + 1)))
        scaling = {}
        for th in cands:   # calibration: the MEDIAN of three regions per thread count; the slow single thread runs quarter regions
            scaling[str(th)] = statistics.median(regions(th, 3, rep=1 if th == 1 else R))
        # the three best calibration points share the budget.  `value` is the MEDIAN region rate of the fastest point whose
        # regions hold together (p90 / p10 < 1.25); a faster point that does not is listed under `measured_points` with its
        # spread and is not the baseline: GPU boxes are slices of a shared 8-GPU host, and above ~16-32 threads the regions
        # of the same pinned team were 4x apart while the per-thread rate up to 16 threads repeats to 2 % box after box
        top = sorted(cands, key=lambda th: scaling[str(th)], reverse=True)[:3]
        points, steady, attempts = {}, [], 0
        # a pass in which the largest team does not hold together (a neighbour's burst on the shared host) is repeated, twice at
        # most, before a smaller team's figure is taken: the smaller teams are steadier but a different number
        while top[0] not in steady and attempts < 3:
            attempts += 1
            for th in top:
                per = B * T * R / scaling[str(th)]
                n_rep = int(max(10, min(5000, budget_s / len(top) / max(per, 1e-6))))
                t0 = time.perf_counter()
                rates = regions(th, n_rep)
                d = time.perf_counter() - t0
                q = statistics.quantiles(rates, n=10)
                v = statistics.median(rates)
                scaling[str(th)] = v
                points[th] = {"p10": q[0], "median": v, "p90": q[-1], "span": q[-1] / q[0] if q[0] > 0 else float("inf"),
                              "regions": n_rep, "wall_s": d}
            steady = [th for th in top if points[th]["span"] < STEADY_SPAN]
        # none steady even then: the FASTEST point, flagged -- the one that moves least from run to run
        best = max(steady, key=lambda th: points[th]["median"]) if steady else max(top, key=lambda th: points[th]["median"])
    finally:
        orc.set_exact_pow(True)
    pt = points[best]
    value, single = pt["median"], scaling["1"]
    faster = {str(th): [points[th]["p10"], points[th]["median"], points[th]["p90"]] for th in top
              if th != best and points[th]["median"] > value}
    return {"value": value, "unit": "env-steps/s", "cores": best, "kind": "port",
            "sample_short": f"C oracle (oracle/flight_oracle.c, OpenMP, env-major), {B} envs x {T * R * pt['regions']} steps on {best} "
                            f"threads, auto-reset, obs+state emitted, {pt['wall_s']:.1f} s wall, {cpu_model()}"
                            + (f", cgroup quota {quota:g} CPUs" if quota else ""),
            "sample": f"C oracle (oracle/flight_oracle.c orc_batch_rollout_rep: one OpenMP region per {T * R} steps, "
                      f"env-major, dynamic schedule), {B} envs x {T * R * pt['regions']} steps on {best} threads (fastest of "
                      f"{top} whose regions span < {STEADY_SPAN}x; calibrated over {cands}), auto-reset, "
                      f"obs+state emitted, {pt['wall_s']:.1f} s wall on {cpu_model()} ({os.cpu_count()} logical CPUs, {n_cores} physical cores)",
            "region_rate_p10_median_p90": [pt["p10"], pt["median"], pt["p90"]], "statistic": "median over the timed regions",
            "p90_over_p10": pt["span"], "steady": bool(steady), "passes": attempts,
            "measured_points": {str(th): [points[th]["p10"], points[th]["median"], points[th]["p90"]] for th in top},
            "faster_but_unsteady": faster or None,
            "pinning": {"OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "OMP_PLACES": os.environ.get("OMP_PLACES"),
                        "physical_cores": n_cores, "threads_per_core": smt, "cgroup_cpu_quota": quota,
                        "affinity_cpus": len(mask0), "affinity": cpu_list_str(mask0),
                        "affinity_source": "parent's mask before its NUMA binding" if os.environ.get("BENCH_CPU_BASELINE_AFFINITY")
                        else "inherited",
                        "note": "own process, one thread per physical core at most (no SMT siblings), never more threads than "
                                "the cgroup's CPU quota - 2"},
            "single_thread_value": single, "speedup_vs_single_thread": value / single,
            "thread_scaling_env_steps_per_s": scaling}


# ------------------------------------------------------------------------------------------------ rank -> CPU binding
def parse_cpulist(text):
    """'0-63,128-191' -> {0..63, 128..191}"""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(local_rank, sysfs="/sys"):
    """(numa node, CPUs) next to the GPU this rank will use, from sysfs alone -- NO GPU call: the binding has to be in place
    before the HIP runtime starts its helper threads.  GPUs = the KFD topology nodes this process may read that have SIMDs,
    in node order (the order HIP enumerates them in); the node's drm_render_minor leads to the PCI device's local_cpulist.
    None when sysfs does not tell (no KFD, fewer GPUs than local_rank, a device filter that is not a list of indices)."""
    base = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
    try:
        nodes = sorted((int(d) for d in os.listdir(base) if d.isdigit()))
    except OSError:
        return None
    gpus, cpu_nodes = [], set()
    for nd in nodes:
        try:
            props = dict(ln.split()[:2] for ln in open(os.path.join(base, str(nd), "properties")) if len(ln.split()) >= 2)
        except OSError:
            continue   # another container's GPU: not ours to read, not ours to enumerate
        if int(props.get("simd_count", "0")) > 0:
            gpus.append((nd, props))
        elif int(props.get("cpu_cores_count", "0")) > 0:
            cpu_nodes.add(nd)
    # device filters in the environment renumber what the runtime shows: ROCR_VISIBLE_DEVICES picks from the topology's GPUs,
    # HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (and GPU_DEVICE_ORDINAL) from what is left.  (A gpurun box sets the first two to "0".)
    order = list(range(len(gpus)))
    for names in (("ROCR_VISIBLE_DEVICES",), ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"), ("GPU_DEVICE_ORDINAL",)):
        val = next((os.environ[k] for k in names if os.environ.get(k)), None)
        if val is None:
            continue
        try:
            pick = [int(x) for x in val.split(",") if x.strip() != ""]
            order = [order[i] for i in pick]
        except (ValueError, IndexError):
            return None   # UUIDs or indices past what this process may read: no guess
    if local_rank >= len(order):
        return None
    nd, props = gpus[order[local_rank]]

    def read_dev(dev):
        try:
            cpus = parse_cpulist(open(os.path.join(dev, "local_cpulist")).read())
            node = int(open(os.path.join(dev, "numa_node")).read().strip() or -1)
        except (OSError, ValueError):
            return None
        return (node, cpus) if cpus else None

    # 1. the render node's PCI device; 2. the same device by its PCI address (a container's /sys/class/drm may not list the
    #    minor it was handed); 3. the KFD io_link from the GPU to a CPU node, whose number is the NUMA node's
    minor = int(props.get("drm_render_minor", "-1"))
    found = read_dev(os.path.join(sysfs, f"class/drm/renderD{minor}/device")) if minor >= 0 else None
    if not found and "location_id" in props:
        loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
        bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}"
        found = read_dev(os.path.join(sysfs, "bus/pci/devices", bdf))
    if not found:
        links = os.path.join(base, str(nd), "io_links")
        try:
            names = sorted(os.listdir(links))
        except OSError:
            names = []
        for ln in names:
            try:
                lp = dict(x.split()[:2] for x in open(os.path.join(links, ln, "properties")) if len(x.split()) >= 2)
                to = int(lp.get("node_to", "-1"))
                if to in cpu_nodes:
                    cpus = parse_cpulist(open(os.path.join(sysfs, f"devices/system/node/node{to}/cpulist")).read())
                    if cpus:
                        found = (to, cpus)
                        break
            except (OSError, ValueError):
                continue
    return found


ORIGINAL_AFFINITY = None   # this process's CPU mask BEFORE bind_rank_to_gpu_node narrowed it: what the cpu_baseline child runs on


def bind_rank_to_gpu_node(local_rank, sysfs="/sys"):
    """Restricts this process (and every thread it starts later: HIP's, RCCL's) to the CPUs of its GPU's NUMA node.  The
    path shards with no data-path collective, so the host cost of a launch (7 us per call against a 40 us region) is the one
    thing that can bend a weak-scaling line: a rank launching across the socket interconnect pays it on every call.
    Returns a description for the JSON line, or None when nothing was bound."""
    found = gpu_local_cpus(local_rank, sysfs)
    if not found:
        return None
    node, cpus = found
    global ORIGINAL_AFFINITY
    before = os.sched_getaffinity(0)
    allowed = cpus & before
    if not allowed:
        return None
    if ORIGINAL_AFFINITY is None:
        ORIGINAL_AFFINITY = set(before)   # children inherit the narrowed mask: cpu_baseline() hands this one to its child
    os.sched_setaffinity(0, allowed)
    return {"numa_node": node, "cpus": len(allowed), "source": "KFD topology + drm local_cpulist, before any GPU call"}


# ------------------------------------------------------------------------------------------------ timed region
class Comm:
    """Barrier / max-over-ranks for 1..N ranks (gloo on host tensors in the shared-GPU test mode, RCCL otherwise)."""

    def __init__(self, world, dev, share, pg=None):
        import torch
        self.world, self.torch = world, torch
        self.cdev = torch.device("cpu") if share else dev
        self.pg = world > 1 if pg is None else pg    # a one-rank RCCL group runs the same barrier / all-reduce calls
        if self.pg:
            import torch.distributed as dist
            self.dist = dist

    def barrier(self):
        if self.pg:
            self.dist.barrier()

    def max(self, x):
        if not self.pg:
            return float(x)
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, x):
        """x of every rank, in rank order (every rank gets the list)."""
        if not self.pg:
            return [float(x)]
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.cdev)
        got = [self.torch.zeros_like(t) for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(got, t)
        return [float(g.item()) for g in got]


METRIC_PARTIALS = 204   # 4 sums (reward, wins, targets found, envs) + the 200-step found-fraction curve (runner.py:86-96, :139-171)


def time_metric_allgather(dist, cdev, reps=100, torch=None, sync=None):
    """The path's ONE collective, timed on its own: `reps` all-gathers of the [204] float32 metric partials (SURVEY.md
    section 8e: latency-bound, over xGMI at N > 1), each bracketed by a device synchronisation; median / min / max in
    microseconds.  Every rank must call it."""
    part = torch.zeros(METRIC_PARTIALS, dtype=torch.float32, device=cdev)
    got = [torch.zeros_like(part) for _ in range(dist.get_world_size())]
    sync = sync or (lambda: None)
    for _ in range(5):
        dist.all_gather(got, part)
    sync()
    us = []
    for _ in range(reps):
        t0 = time.perf_counter()
        dist.all_gather(got, part)
        sync()
        us.append((time.perf_counter() - t0) * 1e6)
    return {"median": statistics.median(us), "min": min(us), "max": max(us), "reps": reps, "floats": METRIC_PARTIALS,
            "clock": "host wall time around all_gather + device synchronisation"}


def timed_region(region, dev, comm, min_gpu_s):
    """Repeats `region()` (the K-step region) bracketed by barrier + synchronize on both sides; returns the per-repeat
    HIP-event seconds and host wall seconds.  Every rank runs the same number of repeats."""
    import torch

    def once():
        torch.cuda.synchronize(dev)
        comm.barrier()
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        region()
        ev1.record()
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        comm.barrier()
        torch.cuda.synchronize(dev)
        return ev0.elapsed_time(ev1) / 1e3, wall

    first = once()
    reps = int(min(MAX_REPEATS, max(MIN_REPEATS, math.ceil(min_gpu_s / max(comm.max(first[0]), 1e-7)))))
    ev, wall = [first[0]], [first[1]]
    for _ in range(reps - 1):
        e, w = once()
        ev.append(e)
        wall.append(w)
    return ev, wall


def run_workload(cs, dev, comm, env_name, n, B, mode, K, W, kernel, rank=0, no_graph=False, min_gpu_s=MIN_GPU_S_ALSO):
    """Measures K steps of one workload on every rank; returns (result dict, env).  Times are max over ranks."""
    import torch
    m = 15
    S = largest_divisor_leq(K, 100)  # steps per graph replay / per rollout launch
    env = cs.BatchedFlightEnv(cs.make_env_args(env_name, n_agents=n), batch=B, device=dev, env_offset=rank * B,
                              freeze_done=False, auto_reset=True, kernel=kernel)
    g = torch.Generator(device=dev).manual_seed(1 + rank)
    acts = torch.randint(0, 3, (S, B, n), dtype=torch.int32, device=dev, generator=g)
    if mode == "rollout":
        out = dict(
            reward=torch.empty(S, B, dtype=torch.float32, device=dev),
            terminated=torch.empty(S, B, dtype=torch.uint8, device=dev),
            win=torch.empty(S, B, dtype=torch.uint8, device=dev),
            obs=torch.empty(S, B, n, env.obs_width, dtype=torch.float32, device=dev),
            state=torch.empty(S, B, env.state_shape, dtype=torch.float32, device=dev))

        def chunk_eager():
            env.rollout(acts, out=out, update_views=False)
    else:
        def chunk_eager():
            for s in range(S):
                env.step(acts[s])
    if no_graph or mode == "rollout":
        # rollout mode: one cs_rollout call per chunk, launched directly.  (Replaying it from a hipGraph was measured: the
        # graph launch costs MORE than the plain launch it replaces -- 58.7 against 52.2 us per 20-step region at c2.)
        chunk = chunk_eager
    else:
        # step mode: the S cs_step (+ cs_mt_advance) launches of a chunk captured once and replayed
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            chunk_eager()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            chunk_eager()

        def chunk():
            graph.replay()
    for _ in range(max(1, math.ceil(W / S))):
        chunk()

    def region():
        for _ in range(K // S):
            chunk()

    ev, wall = timed_region(region, dev, comm, min_gpu_s)
    t_ev = comm.max(statistics.median(ev))
    t_wall = comm.max(statistics.median(wall))
    # every rank's own rate (its median region), not only the slowest one's: a straggler shows as min << median
    per_rank = [B * K / t for t in comm.gather(statistics.median(ev))]
    steps_per_launch = S if mode == "rollout" and env_name != "flight" else 1   # flight: one launch per step either way
    alg = algorithmic_bytes_per_env_step(env_name, n, m, mode)
    label = kernel_label(env_name, n, B, mode, kernel)
    achieved = alg * B * K / t_ev / 1e9
    traffic, traffic_source = pmc_traffic(label, B, steps_per_launch)
    res = {
        "value": B * comm.world * K / t_ev, "unit": "env-steps/s", "ms_per_step": t_ev * 1e3 / K,
        "wall_ms_per_step": t_wall * 1e3 / K, "repeats": len(ev), "timed_gpu_s": sum(ev),
        "region_ms_min_median_max": [min(ev) * 1e3, statistics.median(ev) * 1e3, max(ev) * 1e3],
        "steps_per_launch": steps_per_launch,
        "per_rank_value": {"min": min(per_rank), "median": statistics.median(per_rank), "max": max(per_rank),
                           "slowest_rank": per_rank.index(min(per_rank)), "unit": "env-steps/s per GPU"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "traffic_source": (traffic_source + ": committed rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), per "
                                        "env-step, scaled by this run's batch x steps per launch") if traffic_source else None,
                     "kernel": label,
                     "algorithmic_bytes_per_env_step": alg,
                     "algorithmic_bytes_per_launch": alg * B * steps_per_launch,
                     "avg_launch_us": t_ev / (K / steps_per_launch) * 1e6,
                     "timing": "median HIP-event interval of the K-step region on the launch stream / launches",
                     # (peak is the contract's 8 TB/s; what the part was measured to sustain, for scale)
                     "context": "peak = HBM3E spec; measured on MI355X (tools/probe_write_bw.hip, profiles/r06_write_bw.log): read 6.2-6.4, "
                                "copy 4.6-5.4, pure write 4.1-5.4 TB/s; this path writes >= 90 % of its bytes"},
    }
    return res, env


def side_measurement(cs, dev, comm, key, label, env_name, n, B, mode, K, W, kernel, rank=0):
    """Entry for the `also` list (detail file; `key` names it in the line's also_summary): another workload measured in the
    same run with the same protocol."""
    import torch
    res, env = run_workload(cs, dev, comm, env_name, n, B, mode, K, W, kernel, rank=rank)
    del env
    torch.cuda.empty_cache()
    return {"key": key, "workload": label, "mode": mode, "kernel": kernel, "steps": K, "warmup": W, **res}


def closed_loop_measurement(cs, dev, key, n, B, K, W, env_name="flight_easy"):
    """`also` entry: the reference's recurrent agent network (agents.FusedAgents -> csrc/policy.hip) picks every action
    from the live obs, then env.step, nothing leaves the device."""
    import torch
    args = cs.make_env_args(env_name, n_agents=n)
    env = cs.BatchedFlightEnv(args, batch=B, device=dev, freeze_done=False, auto_reset=True)
    cs.apply_env_info(args, env)
    torch.manual_seed(0)
    agents = cs.FusedAgents(args, B, device=dev)
    # one call per 100 closed-loop steps: k_rollout_policy (flight_easy: one launch) / cs_rollout_policy_flight (flight:
    # conv on the map in place + network + step + map update per step, no observation copies of the map)
    chunk = 100
    emit = env_name == "flight_easy"
    out = env.rollout_policy(agents, chunk, emit=emit)

    def advance(steps):
        for _ in range(steps // chunk):
            env.rollout_policy(agents, chunk, emit=emit, out=out, update_views=False)

    K, W = (K // chunk) * chunk, max(chunk, (W // chunk) * chunk)
    advance(W)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    advance(K)
    e1.record()
    torch.cuda.synchronize(dev)
    dt = e0.elapsed_time(e1) / 1e3
    obs = env.get_obs()
    e0.record()
    for _ in range(200):
        agents.choose_action(obs)
    e1.record()
    torch.cuda.synchronize(dev)
    pol_us = e0.elapsed_time(e1) * 1e3 / 200
    del env, agents
    torch.cuda.empty_cache()
    out = {"key": key, "workload": f"closed loop: {env_name} {n}a15t B={B}, recurrent policy picks every action"
                       + (" (k_rollout_policy: network + env step fused, 100 steps per launch)" if env_name == "flight_easy"
                          else " (cs_rollout_policy_flight: conv on the map in place, network, step, map update per step; "
                               "the n observation copies of the map are not written)"),
           "mode": "closed-loop", "value": B * K / dt, "unit": "env-steps/s", "ms_per_step": dt * 1e3 / K,
           "policy_kernels_us": pol_us}
    if env_name == "flight_easy":   # one kernel, GEMM-shaped: price it against the matrix pipe it runs on
        in_dim = 4 + 3 + n          # useful FLOPs only (the kernel pads fc1's input to 32 and fc2's output to 16)
        flops = 2.0 * B * n * (in_dim * 64 + 2 * 192 * 64 + 64 * 64 + 64 * 3)
        # csrc/policy_dev.h: every fp32 product runs as THREE v_mfma_f32_16x16x32_f16 (split-fp16 operands, 22 bits), so the
        # useful-FLOP peak of this path is a third of the dense 16-bit matrix peak (2.5 PFLOP/s, MI355X_MICROARCH.md)
        peak = 2500.0 / 3.0
        out["policy_roofline"] = {"bound": "mfma", "achieved": flops / pol_us / 1e6, "peak": peak, "unit": "TFLOP/s",
                                  "frac": flops / pol_us / 1e6 / peak, "dtype": "f16x2 (fp32 operands split into two fp16, fp32 accumulate)",
                                  "note": "useful fp32-equivalent FLOPs; the fp32 matrix pipe's peak is 157.3 TFLOP/s"}
    return out


def dry_run(a, rank, world):
    """No GPU: every rank fabricates the metric partials of its shard and the ranks all-gather them over gloo, so the
    launcher, the rendezvous and the reduction can be exercised in a CPU-only container.  Not a measurement."""
    import torch
    import torch.distributed as dist
    B = a.batch or WORKLOADS[a.workload]["batch"]
    if world > 1:
        dist.init_process_group("gloo")
    part = torch.tensor([-100.0 * B, 0.0, 3.0 * B, float(B)], dtype=torch.float64)
    allgather_us, per_rank = None, None
    if world > 1:
        gathered = [torch.zeros_like(part) for _ in range(world)]
        dist.all_gather(gathered, part)
        part = torch.stack(gathered).sum(0)
        # the fields of the real line that only exist at N > 1, through the same code (gloo instead of RCCL; fabricated rates)
        allgather_us = time_metric_allgather(dist, torch.device("cpu"), reps=20, torch=torch)
        comm = Comm(world, torch.device("cpu"), True, True)
        rates = comm.gather(1.0e9 + rank)
        per_rank = {"min": min(rates), "median": statistics.median(rates), "max": max(rates), "slowest_rank": rates.index(min(rates)),
                    "unit": "env-steps/s per GPU (fabricated: 1e9 + rank)"}
    if rank == 0:
        print(json.dumps(_fits({"metric": "env-steps/sec", "value": None, "unit": "env-steps/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "dry_run": True,
                          "eval": {"envs": int(part[3].item()), "world_size": world, "allgather_us": allgather_us},
                          "per_rank_value": per_rank,
                          "also_at_this_n": "all" if (world == 1 or a.also) else "c5 weak + c5 strong only",
                          "c5_strong_total": {"value": None, "n_gpus": world, "envs_total": C5_GLOBAL_BATCH,
                                              "envs_per_gpu": C5_GLOBAL_BATCH // world if C5_GLOBAL_BATCH % world == 0 else None},
                          "c5_weak_total": {"value": None, "n_gpus": world, "envs_total": 8192 * world, "envs_per_gpu": 8192}})),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    a = parse_args(argv)
    if a.cpu_baseline_child:   # the pinned child of cpu_baseline(): numpy + the oracle only, never the GPU
        mask = os.environ.get("BENCH_CPU_BASELINE_AFFINITY")
        if mask:   # the parent's mask from before its NUMA binding (must happen before libgomp places its team)
            try:
                os.sched_setaffinity(0, {int(c) for c in mask.split(",")})
            except (OSError, ValueError) as exc:
                sys.stderr.write(f"bench.py: cpu baseline child keeps the inherited CPU mask ({exc})\n")
        env_name, n, batch, budget = a.cpu_baseline_child.split(",")
        print(json.dumps(cpu_baseline_inproc(env_name, int(n), int(batch), float(budget))), flush=True)
        return
    # the host driver supports dmabuf IPC only: has to be in the environment before ANYTHING initialises HSA
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    in_torchrun = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and not in_torchrun:
        sys.exit(launch_ranks(a, argv))     # nothing above this line has touched the GPU

    import numpy as np  # noqa: F401  (oracle / env import it; fail early if missing)
    import torch


    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    if a.dry_run:
        return dry_run(a, rank, world)
    cpu_binding = None
    if not a.no_numa_bind and os.environ.get("BENCH_SHARE_GPU") != "1":
        try:
            cpu_binding = bind_rank_to_gpu_node(local_rank)   # before the first GPU call of this process
        except OSError as exc:
            sys.stderr.write(f"bench.py: rank {rank}: no CPU binding ({exc})\n")
    # Exactly ONE line goes to stdout: native libraries write there too (RCCL prints its version banner through C stdio
    # when the first communicator comes up, and the buffer is flushed at exit -- AFTER the JSON line).  From here on file
    # descriptor 1 is stderr; the JSON line is written to the saved descriptor at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit_line(obj):
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:   # noqa: BLE001
            pass
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    # BENCH_SHARE_GPU=1 (tests only): all ranks use cuda:0 and the gloo backend, to exercise the N > 1 control flow on
    # a one-GPU box.  The real multi-GPU run is one process per GPU over RCCL (backend "nccl").
    share = os.environ.get("BENCH_SHARE_GPU") == "1"
    if not share and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank} but only {torch.cuda.device_count()} are visible")
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    pg, pg_error, pg_hung = False, None, False
    if world > 1 or in_torchrun or a.pg != "off":
        # one rank outside torchrun: its own rendezvous on the loopback
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port())
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("LOCAL_RANK", "0")
        optional = not (world > 1 or in_torchrun or a.pg == "on")   # N = 1, --pg auto: the line says so; everything else fails loudly

        def bring_up():
            if share:
                dist.init_process_group("gloo")
            else:
                torch.cuda.set_device(dev_index)
                dist.init_process_group("nccl", device_id=dev)

        if not optional:
            bring_up()
            pg = True
        else:
            # a broken RCCL stack usually HANGS in the eager communicator init instead of raising: the optional one-rank group
            # comes up on a watchdog thread; if it has not within PG_WATCHDOG_S the bench carries on without it (the collective
            # is outside the timed region) and leaves through os._exit so that the stuck thread cannot hold the process
            import threading
            box = {}

            def guarded():
                try:
                    bring_up()
                    box["ok"] = True
                except Exception as exc:   # noqa: BLE001
                    box["err"] = f"{type(exc).__name__}: {exc}"[:300]

            th = threading.Thread(target=guarded, daemon=True, name="rccl-init")
            th.start()
            th.join(PG_WATCHDOG_S)
            if box.get("ok"):
                pg = True
            else:
                pg_error = box.get("err") or f"init_process_group('nccl') did not return within {PG_WATCHDOG_S} s"
                pg_hung = th.is_alive()
                sys.stderr.write(f"bench.py: RCCL process group at world_size 1 failed ({pg_error}); continuing without\n")
    comm = Comm(world, dev, share, pg)

    import cooperative_search_amd as cs

    wl = dict(WORKLOADS[a.workload])
    if a.batch:
        wl["batch"] = a.batch
    env_name, n, B, m = wl["env"], wl["n_agents"], wl["batch"], 15
    mode = a.mode or "rollout"
    K, W = a.steps, a.warmup

    res, env = run_workload(cs, dev, comm, env_name, n, B, mode, K, W, a.kernel, rank=rank, no_graph=a.no_graph,
                            min_gpu_s=a.min_gpu_s)

    # evaluation-metric reduction (runner.py:86-96): the path's only collective, outside the timed region
    part = env.metric_partials().clone().to(comm.cdev)
    if pg:
        try:
            gathered = [torch.zeros_like(part) for _ in range(world)]
            dist.all_gather(gathered, part)
            part = torch.stack(gathered).sum(0)
            torch.cuda.synchronize(dev)
        except Exception as exc:   # noqa: BLE001
            if world > 1 or in_torchrun or a.pg == "on":
                raise
            pg, pg_error = False, f"all_gather: {type(exc).__name__}: {exc}"[:300]
            comm.pg = False
    part = part.cpu().numpy()
    allgather_us = None
    if pg:
        try:
            allgather_us = time_metric_allgather(dist, comm.cdev, torch=torch, sync=lambda: torch.cuda.synchronize(dev))
        except Exception as exc:   # noqa: BLE001
            if world > 1 or in_torchrun or a.pg == "on":
                raise
            allgather_us = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    del env
    torch.cuda.empty_cache()

    line = None
    if rank == 0:
        line = {
            "metric": "env-steps/sec", "value": res["value"], "unit": "env-steps/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{env_name}, {n} agents, {m} targets, batch={B} envs per GPU, agent_mode=0, "
                                   f"target_mode=0 ({a.workload})", "mode": mode,
                       "steps_per_launch": res["steps_per_launch"], "auto_reset": True, "emits": "obs+state every step",
                       "kernel": a.kernel, "hip_graph": mode == "step" and not a.no_graph,
                       "backend": ("gloo (BENCH_SHARE_GPU test mode)" if share else "nccl (RCCL)") if pg else None,
                       "cpu_binding": cpu_binding,
                       **({"backend_error": pg_error} if pg_error else {})},
            "timing": {"clock": "HIP events on the launch stream, median over repeats of the K-step region, max over ranks",
                       "repeats": res["repeats"], "timed_gpu_s": res["timed_gpu_s"],
                       "region_ms_min_median_max": res["region_ms_min_median_max"],
                       "wall_ms_per_step": res["wall_ms_per_step"]},
            "roofline": res["roofline"],
            "eval": {"mean_episode_reward_so_far": part[0] / part[3], "win_rate_now": part[1] / part[3],
                     "mean_targets_found_now": part[2] / part[3], "envs": int(part[3]), "world_size": world,
                     "reduced_by": "all_gather over the process group" if pg else "local (no process group)",
                     "allgather_us": allgather_us},
            "per_rank_value": res["per_rank_value"],
        }
    also = []
    if not a.no_also and a.workload == "c2" and not a.batch:
        # c5 (BASELINE config 5: flight_easy 5a15t, 65536 envs over the node) at every N: the weak point (8192 per GPU,
        # = the 8-GPU configuration's per-GPU share) and the strong point (65536 / N per GPU)
        also.append(side_measurement(cs, dev, comm, "c5w", "c5 weak: flight_easy 5a15t, 8192 envs per GPU", "flight_easy", 5,
                                     8192, "rollout", 1000, 100, "auto", rank=rank))
        if C5_GLOBAL_BATCH % world == 0:
            also.append(side_measurement(cs, dev, comm, "c5s", f"c5 strong: flight_easy 5a15t, 65536 envs over {world} GPU(s)",
                                         "flight_easy", 5, C5_GLOBAL_BATCH // world, "rollout", 400, 100, "auto", rank=rank))
        if world == 1 or a.also:
            also += [
                side_measurement(cs, dev, comm, "c2_step", "c2 flight_easy 3a15t B=4096, one launch per step (hipGraph)",
                                 "flight_easy", 3, 4096, "step", 2000, 200, "auto"),
                side_measurement(cs, dev, comm, "c3", "c3 flight_easy 5a15t B=16384", "flight_easy", 5, 16384, "rollout",
                                 1000, 100, "auto"),
                side_measurement(cs, dev, comm, "fe3_16384", "flight_easy 3a15t B=16384 (the 8192..65536 valley of round 2)", "flight_easy", 3,
                                 16384, "rollout", 1000, 100, "auto"),
                side_measurement(cs, dev, comm, "fe3_32768", "flight_easy 3a15t B=32768 (one-wavefront octet kernel)", "flight_easy", 3,
                                 32768, "rollout", 400, 100, "auto"),
                side_measurement(cs, dev, comm, "fe5_32768", "flight_easy 5a15t B=32768 (one-wavefront octet kernel)", "flight_easy", 5,
                                 32768, "rollout", 400, 100, "auto"),
                side_measurement(cs, dev, comm, "c4_step", "c4 flight 3a15t B=8192, cs_step per step (hipGraph): k_step then k_map",
                                 "flight", 3, 8192, "step", 400, 100, "auto"),
                side_measurement(cs, dev, comm, "c4", "c4 flight 3a15t B=8192 (cs_rollout: sweep of step t beside step t + 1)",
                                 "flight", 3, 8192, "rollout", 400, 100, "auto"),
                side_measurement(cs, dev, comm, "c4_32768", "flight 3a15t B=32768 (cs_rollout; 328 MB of maps: past the 256 MiB Infinity Cache)",
                                 "flight", 3, 32768, "rollout", 40, 20, "auto"),
                side_measurement(cs, dev, comm, "fe3_2^18", "flight_easy 3a15t B=262144 (lane-per-env kernel: the HBM-regime kernel)",
                                 "flight_easy", 3, 262144, "rollout", 200, 100, "auto"),
                side_measurement(cs, dev, comm, "fe3_2^20", "flight_easy 3a15t B=1048576 (lane-per-env kernel; batch sweep asymptote)",
                                 "flight_easy", 3, 1048576, "rollout", 100, 100, "auto"),
                side_measurement(cs, dev, comm, "fe5_2^18", "flight_easy 5a15t B=262144 (lane-per-env kernel)",
                                 "flight_easy", 5, 262144, "rollout", 400, 100, "auto"),
                side_measurement(cs, dev, comm, "fe5_2^20", "flight_easy 5a15t B=1048576 (lane-per-env kernel)",
                                 "flight_easy", 5, 1048576, "rollout", 200, 100, "auto"),
                closed_loop_measurement(cs, dev, "loop3_4096", 3, 4096, 2000, 200),
                closed_loop_measurement(cs, dev, "loop3_65536", 3, 65536, 400, 100),
                closed_loop_measurement(cs, dev, "loop_flight_8192", 3, 8192, 400, 100, "flight"),
            ]
    if rank == 0:
        if also:
            line["also"] = also
            # BASELINE config 5 (flight_easy 5a15t, 65536 envs over the node) as fields of their own, whole-job totals over
            # all N ranks: the strong point (65536 / N envs per GPU) and the weak point (8192 envs per GPU)
            for key, tag in (("c5_strong_total", "c5 strong"), ("c5_weak_total", "c5 weak")):
                ent = next((x for x in also if str(x.get("workload", "")).startswith(tag)), None)
                if ent:
                    line[key] = {"value": ent["value"], "unit": "env-steps/s", "n_gpus": world,
                                 "envs_total": (C5_GLOBAL_BATCH if tag == "c5 strong" else 8192 * world),
                                 "envs_per_gpu": (C5_GLOBAL_BATCH // world if tag == "c5 strong" else 8192),
                                 "roofline_frac_per_gpu": ent["roofline"]["frac"]}
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(env_name, n, B)
        # stdout carries the compact line (<= MAX_LINE characters); the full record goes to gpurun_out/bench_detail.json and,
        # one entry per row, to stderr BEFORE the line (so the line stays the last thing this process prints)
        detail = write_detail(line)
        for ent in also:
            fr = (ent.get("roofline") or {}).get("frac")
            sys.stderr.write(f"bench.py: also {ent.get('key'):>16}  {ent['value']:.4g} env-steps/s  "
                             f"{'frac %.3f' % fr if fr is not None else 'closed loop'}  {ent.get('workload')}\n")
        sys.stderr.flush()
        emit_line(compact_line(line, detail))
    if pg:
        dist.barrier()
        dist.destroy_process_group()
    if pg_hung:   # the watchdog gave up on a communicator init that is still stuck: do not let it hold the exit
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
