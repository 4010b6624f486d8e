#!/usr/bin/env python3
"""Benchmark of the hot path: env-steps/s of the batched Inversus step at 16 384 envs per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N child ranks (fresh
processes, started before this process makes any GPU call; never an exec of a process that touched the GPU); under an
external torchrun the ranks are used as given.  Either way it is one process per GPU over RCCL (backend "nccl").

One "step" = one vector step of the shard's 16 384 envs: P1 actions from a synthetic stream (device
Philox, the stand-in for the policy in BASELINE.json configs[1] "HIP step + obs kernels only";
generated before the timed region), device dummy opponent, core tick, reward/done/info, auto-reset, 12-channel observation render.
Inputs (env state) are resident in HBM before the timed region.  Envs are sharded over ranks
by global index with no data-path collective (weak scaling: 16 384 envs per GPU).

The timed loop of `value` writes every step's outputs into one of 32 DISTINCT buffers (3.8 GB of observations: what a collector
that keeps a rollout does, and a footprint the 256 MiB Infinity Cache cannot hold), so `roofline.achieved / frac` are HBM figures.
The same launches re-writing ONE 118 MB buffer (rounds 1-3's headline loop; memory-side-cache writes) are timed beside it and
reported as `value_cache_resident` / `roofline.*_cache_resident`.

Rank 0 prints ONE JSON line -- except when the C4 extra runs (N > 1, or --force-dist): then the complete headline line is printed
and flushed BEFORE the extra (the first multi-rank RCCL all-reduce of the PPO path cannot lose it), and the same line with the
extra's result -- or "failed: ..." -- is printed again after it: the LAST line is the superset, both carry the same `value`, and
the exit code is 0 either way (c4_guarded).  `roofline` is for the dominant kernel (k_step4) from HIP events on
the launch stream (`frac` = single launches, `frac_multi_step` = inv_rollout, T steps per launch: SURVEY 8d's two figures); `cpu_baseline` times the CPU oracle (test infrastructure, used here only as
the yardstick) on a bounded sample of the same workload, in a child process of its own (OpenMP binding set before the runtime starts).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NUM_ENVS = 16384            # per GPU (BASELINE.json metric: "env-steps/sec at 16384 envs")
W, H, K = 15, 10, 16
# HIP-event groups inside a timed region: 0 = ONE pair around the region's K launches (the default: an event record is a queue marker
# that costs the stream ~1 us, and with a pair every 10 launches -- rounds 1-4 -- the markers were 0.3-0.5 us per step of `value`'s wall
# clock); N > 0 = a pair around every N launches (diagnostics: per-group times).
EV_GROUP = int(os.environ.get("INVERSUS_BENCH_EV_GROUP", "0"))
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured achievable)


def algorithmic_bytes_per_env_step(w=W, h=H, k=K, selfplay=False):
    """obs + extra + reward + done + actions + 2 x packed state."""
    s = (w * h + 7) // 8 + 10 + 1 + 2 * k + 2 + 1 + 8
    obs = 12 * w * h * 4 + 16
    return (2 if selfplay else 1) * obs + 4 + 1 + 2 + 2 * s


def traffic_from_profiles(difficulty, suffix=""):
    """HBM bytes per k_step launch from the committed rocprofv3 --pmc passes (profiles/traffic.json);
    PMC collection needs its own profiler run, so bench.py reports the recorded value or null."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            return json.load(fh)[f"vs_dummy_{difficulty}_{W}x{H}{suffix}"]["traffic_bytes"]
    except Exception:
        return None


def host_threads():
    """Threads this job may actually use: the scheduler affinity, further limited by a cgroup CPU quota if one is set."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()
            if q != "max":
                quota = max(1, int(float(q) / float(per)))
    except (OSError, ValueError):
        pass
    return aff, quota


def gpu_identity(torch):
    """Which GPU of the pool this was: the boards differ in sustained HBM write rate (5.7 vs 6.8 TB/s measured on two of them with the
    same binary, profiles/r03_v2_c5_launch_series_*), which moves every figure whose footprint exceeds the Infinity Cache."""
    try:
        p = torch.cuda.get_device_properties(torch.cuda.current_device())
        out = {"name": p.name, "arch": getattr(p, "gcnArchName", None), "cus": p.multi_processor_count}
        u = getattr(p, "uuid", None)
        if u is not None:
            out["uuid"] = str(u)
        return out
    except Exception as exc:                                              # pragma: no cover
        return {"error": str(exc)}


def thread_placement():
    """Where the OpenMP threads of the CPU baseline may run and where their memory comes from: the binding policy in force
    (none unless the environment sets one: threads float over the affinity mask under a CPU-time quota), the allowed CPUs /
    NUMA nodes, and the node of the calling thread (buffers are first-touched by the threads that write them)."""
    out = {"OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "OMP_PLACES": os.environ.get("OMP_PLACES"),
           "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS"), "first_touch": "inside the OpenMP region (each thread faults in the pages it writes)"}
    try:
        with open("/proc/self/status") as f:
            for ln in f:
                if ln.startswith(("Cpus_allowed_list", "Mems_allowed_list")):
                    k, v = ln.split(":", 1)
                    out[k] = v.strip()
    except OSError:
        pass
    try:
        import ctypes
        cpu = ctypes.CDLL(None).sched_getcpu()                            # os.sched_getcpu arrives with Python 3.13
        out["calling_thread_cpu"] = cpu
        if cpu is not None:
            import glob
            nodes = [d for d in glob.glob(f"/sys/devices/system/cpu/cpu{cpu}/node*")]
            out["calling_thread_numa_node"] = int(os.path.basename(nodes[0])[4:]) if nodes else None
    except (OSError, AttributeError):
        pass
    return out


def cpu_baseline_child(difficulty):
    """cpu_baseline() in fresh processes whose OpenMP runtime starts with the binding under test (the variables are read once, when
    the runtime initialises -- too late in a process that has imported torch).  The full set of figures under OMP_PROC_BIND=spread /
    OMP_PLACES=cores (the job's threads on as many core complexes and memory channels as the affinity mask gives: the CPU's best
    case here), and the all-threads figure alone under `close` (all threads on neighbouring cores of ONE complex) and unbound."""
    import subprocess

    def child(bind, quick):
        env = dict(os.environ)
        if bind is None:
            env.pop("OMP_PROC_BIND", None); env.pop("OMP_PLACES", None)
        else:
            env["OMP_PROC_BIND"], env["OMP_PLACES"] = bind, "cores"
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--difficulty", difficulty] + (["--cpu-baseline-quick"] if quick else [])
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": f"cpu baseline child failed (rc {r.returncode}): {r.stderr[-300:]}"}
        return json.loads(line[-1])

    res = child("spread", False)
    if "error" in res:
        return res
    by = {"spread": round(res["value"])}
    for name, bind in (("close", "close"), ("unbound", None)):
        q = child(bind, True)
        by[name] = round(q["value"]) if "value" in q else q
    res["all_threads_env_steps_per_s_by_binding"] = by
    best = max((v for v in by.values() if isinstance(v, int)), default=res["value"])
    res["value_binding"] = "spread"
    res["best_binding_env_steps_per_s"] = best
    return res


def cpu_baseline(difficulty, budget_s=8.0, quick=False):
    """The C oracle (bit-checked against the reference) on the host cores: same seeds, same action stream, same
    config; bounded sample.  Threads = what the job may use (affinity / cgroup quota, both reported), the output
    buffers are allocated once and first touched inside the OpenMP region (each thread faults in the pages it writes)."""
    import numpy as np
    from oracle import oracle as orc
    aff, quota = host_threads()
    cores = min(aff, quota) if quota else aff
    L = orc.lib()
    hard = difficulty == "hard"

    def run(n, threads, budget):
        ob = orc.OracleBatch(n, W, H)
        ob.seed_reset(np.arange(n, dtype=np.int64))
        acts = [np.array([L.orc_uniform_action(1234, i, t) for i in range(n)], np.uint8) for t in range(4)]
        bufs = ob.alloc_out()
        ob.step(acts[0], None, hard=hard, opp_seed=1, nthreads=threads, out=bufs)          # warm-up + first touch
        steps, t0 = 0, time.perf_counter()
        while True:
            ob.step(acts[steps % 4], None, hard=hard, opp_seed=1, nthreads=threads, out=bufs)
            steps += 1
            dt = time.perf_counter() - t0
            if dt > budget or steps >= 20000:
                break
        return n * steps / dt, steps

    def python_loop(n, budget):
        """The reference's structure (env_wrappers.py:494-528): a Python loop over env OBJECTS, one C call per env for the
        dummy, the step and the observation, np.stack at the end; single thread."""
        envs = [orc.OracleEnv(W, H, seed=i) for i in range(n)]
        steps, t0 = 0, time.perf_counter()
        while True:
            obs = []
            for i, e in enumerate(envs):
                a1 = L.orc_uniform_action(1234, i, steps)
                _, done, _ = e.step(a1, e.dummy_device(hard, 1, i))
                if done:
                    e.reset()
                obs.append(e.build_observation(0))
            np.stack([o[0] for o in obs]); np.stack([o[1] for o in obs])
            steps += 1
            dt = time.perf_counter() - t0
            if dt > budget or steps >= 200000:
                break
        return n * steps / dt

    if quick:                                            # the all-threads figure alone (binding comparison)
        v, steps = run(NUM_ENVS, cores, 2.5)
        return {"value": v, "unit": "env-steps/s", "cores": cores, "placement": thread_placement()}
    v, steps = run(NUM_ENVS, cores, budget_s)
    v1, _ = run(NUM_ENVS, 1, 3.0)
    # one thread on a per-thread SLICE of the job (16 384 envs / 16 threads = 1 024 envs: 7.4 MB of observations instead of
    # 118 MB): the per-core yardstick that all-threads / threads should be compared with -- one thread streaming the whole
    # 118 MB batch is bound by one core's share of the memory system, which is why all-threads looked super-linear against it
    v1s, _ = run(max(4, NUM_ENVS // max(1, cores)), 1, 2.0)
    scaling = {"1": round(v1), str(cores): round(v)}
    if cores > 16:
        scaling["16"] = round(run(NUM_ENVS, 16, 3.0)[0])
    v4, _ = run(4, 1, 1.0)
    vpy = python_loop(4, 1.0)
    vpy_big = python_loop(NUM_ENVS, 0.0)                  # ONE vector step of 16 384 env objects (N = 4 and N = 16 384)
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"value": v, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{steps} vector steps x {NUM_ENVS} envs, vs_dummy_{difficulty}, C oracle (OpenMP over envs, static schedule, "
                      "buffers reused and first-touched in the parallel region), obs rendered",
            "env_steps_per_s_by_threads": scaling, "speedup_all_threads_vs_one": round(v / v1, 2),
            "one_core_env_steps_per_s": round(v1),
            "one_core_on_a_per_thread_slice_env_steps_per_s": round(v1s), "per_thread_slice_envs": max(4, NUM_ENVS // max(1, cores)),
            "speedup_all_threads_vs_one_on_its_slice": round(v / v1s, 2),
            "n4_one_core_env_steps_per_s": round(v4),
            "n4_python_object_loop_env_steps_per_s": round(vpy),
            "n16384_python_object_loop_env_steps_per_s": round(vpy_big),
            "python_object_loop": "the reference's MultiEnvRunner structure (env_wrappers.py:494-528) over the C port: a Python loop over env "
                                  "objects, one C call per env for dummy / step / observation, np.stack; one thread; N = 16 384: one vector step",
            # the Python reference itself cannot travel to the GPU box: its rates measured in the survey container
            "reference_python_container": {"n4_env_steps_per_s": 11300, "n16384_env_steps_per_s": 8100, "eight_processes_x_2048_envs_env_steps_per_s": 89400,
                                           "what": "MultiEnvRunner.step + training-loop-style reset, vs_dummy easy, no policy (inversus_rl/env_wrappers.py:485-528)",
                                           "host": "8 vCPU Intel Xeon @ 2.10 GHz (the survey container), NOT this box"},
            "host": {"cpu_model": model, "nproc": os.cpu_count(), "sched_affinity": aff, "cgroup_cpu_quota": quota,
                     "placement": thread_placement()}}


def side_measurements(torch, np, n):
    """Secondary numbers of the same build for the record (never `value`): other configurations of the
    step kernel, the GAE scan, the fused policy trunk and the policy-in-the-loop self-play collector.
    Every entry is best-effort: a failure is reported as a string and never breaks the JSON line."""
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch, gae

    def timed(fn, reps, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3          # us

    out = {}

    def guard(name, fn):
        try:
            out[name] = fn()
        except Exception as exc:                          # pragma: no cover
            out[name] = f"failed: {type(exc).__name__}: {exc}"

    def step_cfg(W_, H_, opponent, difficulty, obs_p2, selfplay_bytes, bits=False):
        def run():
            b = DeviceEnvBatch(n, W_, H_, K, opponent=opponent, difficulty=difficulty, opp_seed=1)
            b.seed(np.arange(n, dtype=np.int64)); b.reset()
            o = b._alloc_out((), True, obs_p2, False, None, bits)
            acts = torch.stack([b.uniform_actions(5, t) for t in range(64)])
            a2 = torch.stack([b.uniform_actions(6, t) for t in range(64)]) if opponent == "selfplay" else None
            k = [0]
            steppers = [b.make_stepper(o, None if a2 is None else a2[j], auto_reset=True) for j in range(64)] if a2 is not None \
                else [b.make_stepper(o, None, auto_reset=True)] * 64

            def f():
                j = k[0] % 64
                k[0] += 1
                steppers[j](acts[j].data_ptr())
            us = timed(f, 128, warm=64)
            by = algorithmic_bytes_per_env_step(W_, H_, K, selfplay=selfplay_bytes) * n
            if bits:                                      # packed observations: OW words instead of 12*W*H floats per perspective
                by -= (2 if obs_p2 else 1) * (12 * W_ * H_ * 4 - 4 * ((12 * W_ * H_ + 31) // 32)) * n
            return {"us_per_vector_step": round(us, 2), "env_steps_per_s": round(n / us * 1e6), "GBps": round(by / us / 1e3, 1)}
        return run

    guard("step vs_dummy_hard 15x10", step_cfg(15, 10, "dummy", "hard", False, False))
    guard("step selfplay 15x10 (P1+P2 obs)", step_cfg(15, 10, "selfplay", "easy", True, True))
    guard("step vs_dummy_hard 32x32 K=16 (C5)", step_cfg(32, 32, "dummy", "hard", False, False))
    guard("step selfplay 15x10, PACKED P1+P2 obs (INV_STEP_OBS_BITS)", step_cfg(15, 10, "selfplay", "easy", True, True, bits=True))

    def rollout_cfg(T=32):
        """The persistent multi-step variant: T vector steps in ONE launch (inv_rollout) against T single launches.
        The launch writes T DISTINCT output slots (3.8 GB: an HBM stream); `single launches` here re-write ONE slot (Infinity-Cache
        resident, like the headline loop) -- the like-for-like single-launch figure is the next entry (distinct buffers)."""
        b = DeviceEnvBatch(n, 15, 10, K, opponent="dummy", difficulty="easy", opp_seed=1)
        b.seed(np.arange(n, dtype=np.int64)); b.reset()
        a1 = torch.stack([b.uniform_actions(7, t) for t in range(T)])
        o = b._alloc_out((T,), True, False, False, None)
        us = timed(lambda: b.rollout(a1, out=o), 6, warm=2) / T
        o1 = b._alloc_out((), True, False, False, None)
        stp = b.make_stepper(o1, None, auto_reset=True)
        k = [0]

        def single():
            stp(a1[k[0] % T].data_ptr()); k[0] += 1
        us1 = timed(single, 6 * T, warm=T)
        by = algorithmic_bytes_per_env_step() * n
        return {"T": T, "us_per_vector_step_in_one_launch": round(us, 2), "GBps": round(by / us / 1e3, 1),
                "us_per_vector_step_single_launches_one_buffer": round(us1, 2), "GBps_single_launches_one_buffer": round(by / us1 / 1e3, 1)}
    guard("inv_rollout vs_dummy_easy 15x10 (T steps per launch)", rollout_cfg)

    def streaming_cfg(T=32):
        """The headline loop re-writes ONE 118 MB observation buffer every step, which the 256 MiB Infinity Cache can keep
        resident; a collector that keeps every step's observations (or a rollout launch) writes T distinct buffers (3.8 GB at
        T = 32), a true HBM stream.  Same kernel, same launches, output slot t % T."""
        from inversus_reinforcement_learning_amd.device_env import StepResult
        b = DeviceEnvBatch(n, 15, 10, K, opponent="dummy", difficulty="easy", opp_seed=1)
        b.seed(np.arange(n, dtype=np.int64)); b.reset()
        a1 = torch.stack([b.uniform_actions(7, t) for t in range(T)])
        o = b._alloc_out((T,), True, False, False, None)
        outs = [StepResult(**{k2: (getattr(o, k2)[j] if getattr(o, k2) is not None else None) for k2 in StepResult.__slots__}) for j in range(T)]
        st = [b.make_stepper(outs[j], None, auto_reset=True) for j in range(T)]
        k = [0]

        def f():
            j = k[0] % T; k[0] += 1; st[j](a1[j].data_ptr())
        us = timed(f, 8 * T, warm=T)
        by = algorithmic_bytes_per_env_step() * n
        return {"T": T, "us_per_vector_step": round(us, 2), "GBps": round(by / us / 1e3, 1), "env_steps_per_s": round(n / us * 1e6)}
    guard("step vs_dummy_easy 15x10 into 32 DISTINCT output buffers (3.8 GB footprint: streams to HBM)", streaming_cfg)

    def two_shards_cfg():
        """The GPU's 16 384 envs as TWO independent shards of 8 192 (global env indices [0, 8192) and [8192, 16384): the same envs,
        the same results) stepped on two HIP streams -- the way the shards of different GPUs are independent.  One shard's launch
        boundary and the ~2 us in front of its first store then run under the other shard's observation stream.  us per FULL vector
        step (both shards) from the wall clock over 512 steps, median of 5; cache-resident (one output buffer per shard) and
        streaming (32 distinct buffers per shard); the one-batch figures of the same process beside them."""
        from inversus_reinforcement_learning_amd.device_env import StepResult

        from inversus_reinforcement_learning_amd.device_env import EnvShards

        def make(m, off, T, stream):
            sh = EnvShards(m, 1, seed=0, env_offset=off, streams=[stream], width=15, height=10, max_bullets=K, opponent="dummy",
                           difficulty="easy", opp_seed=1).shards[0]
            with torch.cuda.stream(stream):
                b = sh.batch
                acts = torch.stack([b.uniform_actions(5, t) for t in range(64)])
                o = b._alloc_out((T,), True, False, False, None)
                outs = [StepResult(**{k2: (getattr(o, k2)[j] if getattr(o, k2) is not None else None) for k2 in StepResult.__slots__}) for j in range(T)]
                st = [b.make_stepper(outs[j % T], None, auto_reset=True) for j in range(64)]          # bound to `stream`
            stream.synchronize()
            return b, acts, o, st

        def run(parts, reps=512):
            for t in range(64):
                for _, acts, _, st in parts:
                    st[t % 64](acts[t % 64].data_ptr())
            torch.cuda.synchronize()
            out_ = []
            for _ in range(5):
                t0 = time.perf_counter()
                for t in range(reps):
                    for _, acts, _, st in parts:
                        st[t % 64](acts[t % 64].data_ptr())
                torch.cuda.synchronize()
                out_.append((time.perf_counter() - t0) / reps * 1e6)
            for b_, *_ in parts:
                b_.check()
            return sorted(out_)[2]

        by = algorithmic_bytes_per_env_step() * n
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        res = {}
        for T, tag in ((1, "cache_resident"), (32, "streaming")):
            one = [make(n, 0, T, s1)]
            us1 = run(one)
            del one
            torch.cuda.empty_cache()
            two = [make(n // 2, 0, T, s1), make(n // 2, n // 2, T, s2)]
            us2 = run(two)
            del two
            torch.cuda.empty_cache()
            res[tag] = {"us_per_full_vector_step_one_batch_one_stream": round(us1, 2), "us_per_full_vector_step_two_shards_two_streams": round(us2, 2),
                        "env_steps_per_s_two_shards": round(n / us2 * 1e6), "GBps_two_shards": round(by / us2 / 1e3, 1),
                        "frac_of_8000_GBps_two_shards": round(by / us2 / 1e3 / HBM_PEAK_GBS, 3)}
        return res
    guard("two independent env shards of 8192 on two HIP streams (vs_dummy_easy 15x10 f32, wall clock per full 16384-env step)", two_shards_cfg)

    def gae_cfg():
        T = 128
        r = torch.randn(T, n, device="cuda"); v = torch.randn(T, n, device="cuda")
        d = (torch.rand(T, n, device="cuda") < 0.03).to(torch.uint8)
        us = timed(lambda: gae(r, v, d, layout="per_env"), 20)
        # the kernel alone: one pre-marshalled ctypes call per launch (the wrapper's allocations cost more host time than the
        # kernel runs, so the figure above is host-bound)
        import ctypes as C
        from inversus_reinforcement_learning_amd import _lib as L
        lib = L.load()
        adv, ret = torch.empty_like(r), torch.empty_like(r)
        args = (C.c_void_p(r.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(d.data_ptr()), T, n, float(np.float32(0.99)),
                float(np.float32(0.99 * 0.95)), None, L.INV_GAE_PER_ENV, C.c_void_p(adv.data_ptr()), C.c_void_p(ret.data_ptr()),
                C.c_void_p(torch.cuda.current_stream().cuda_stream))
        usk = timed(lambda: lib.inv_gae(*args), 200, warm=20)
        return {"T": T, "us_incl_wrapper": round(us, 2), "us_kernel_back_to_back": round(usk, 2), "GBps": round(17 * T * n / usk / 1e3, 1),
                "frac_of_8000_GBps": round(17 * T * n / usk / 1e3 / HBM_PEAK_GBS, 3)}
    guard("gae per_env T=128", gae_cfg)

    def policy_cfg():
        from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
        from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
        from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
        from inversus_reinforcement_learning_amd.training import RolloutCollector
        torch.manual_seed(0)
        pol = InversusCNNPolicy(12, H, W, 4).cuda().eval()
        fused = FusedPolicy(pol)
        g = (torch.rand(n, 12, H, W, device="cuda") < 0.2).float()
        x = torch.rand(n, 4, device="cuda")
        # 40 launches per figure: with 5 the clock ramp after the idle gap reads 1.34-1.37 ms for a kernel that runs 1.13-1.16 ms
        # in steady state, whatever the observations hold (tools/trunk_data_dep.py)
        res = {"fused_trunk_ms": round(timed(lambda: fused.features(g), 40, warm=5) / 1e3, 3),
               "fused_forward_ms": round(timed(lambda: fused(g, x), 40, warm=5) / 1e3, 3)}
        res["fused_trunk_TFLOPs"] = round(72.9e6 * n / (res["fused_trunk_ms"] * 1e-3) / 1e12, 1)
        res["fused_trunk_frac_of_2500_TFLOPs_dense_bf16_peak"] = round(res["fused_trunk_TFLOPs"] / 2500.0, 3)
        gbits = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, 57), dtype=torch.int32, device="cuda") & \
            torch.randint(-2 ** 31, 2 ** 31 - 1, (n, 57), dtype=torch.int32, device="cuda")        # ~25 % ones
        res["fused_trunk_packed_obs_ms"] = round(timed(lambda: fused.features(gbits), 40, warm=5) / 1e3, 3)
        with torch.no_grad():
            res["torch_f32_forward_ms"] = round(timed(lambda: pol(g, x), 3, warm=1) / 1e3, 3)
        b = DeviceEnvBatch(n, opponent="selfplay")
        b.seed(np.arange(n, dtype=np.int64)); b.reset()
        agent = PPOAgent(pol, fused_act=True)
        opp = FusedPolicy(InversusCNNPolicy(12, H, W, 4).cuda().eval(), actor_only=True)       # as training.py builds it
        for packed, key in ((True, "selfplay_collector_env_steps_per_s"), (False, "selfplay_collector_f32_obs_env_steps_per_s")):
            col = RolloutCollector(b, agent, opp, packed_obs=packed)
            # warm-up with the SAME number of steps: the rollout buffer keeps every step's observations alive, so a shorter
            # warm-up leaves hipMalloc calls of 118 MB blocks inside the timed region (the f32 variant read 1.3 M instead of 4.8 M)
            col.collect(8); agent.reset_buffers(); torch.cuda.synchronize()
            rates = []
            for _ in range(3):                                   # median of 3: the first timed collect after a (re)build reads 5-7 % low
                t0 = time.perf_counter(); col.collect(8); torch.cuda.synchronize()
                rates.append(round(8 * n / (time.perf_counter() - t0)))
                agent.reset_buffers()
            res[key] = sorted(rates)[1]
        return res
    guard("policy in the loop (selfplay, fused bf16 trunk, 2 forwards/step)", policy_cfg)

    def policy32_cfg():
        """BASELINE config C5 with the policy in the loop: the banded fused trunk on the 32 x 32 board and a self-play collector."""
        from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
        from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
        from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
        from inversus_reinforcement_learning_amd.training import RolloutCollector
        torch.manual_seed(0)
        b = DeviceEnvBatch(n, 32, 32, K, opponent="selfplay")
        b.seed(np.arange(n, dtype=np.int64)); b.reset()
        pol = InversusCNNPolicy(12, 32, 32, 4).cuda().eval()
        agent = PPOAgent(pol, fused_act=True)
        bits, _ = b.render(0, bits=True)
        ms = timed(lambda: agent.fused.features(bits), 3, warm=1) / 1e3
        col = RolloutCollector(b, agent, FusedPolicy(InversusCNNPolicy(12, 32, 32, 4).cuda().eval(), actor_only=True))
        col.collect(3); agent.reset_buffers(); torch.cuda.synchronize()
        t0 = time.perf_counter(); col.collect(3); torch.cuda.synchronize()
        rate = round(3 * n / (time.perf_counter() - t0))
        agent.reset_buffers()
        return {"banded_fused_trunk_ms": round(ms, 2), "TFLOPs": round(497.7e6 * n / (ms * 1e-3) / 1e12, 1),
                "selfplay_collector_env_steps_per_s": rate}
    guard("C5 with the policy: 32x32 banded fused trunk + self-play collector (packed obs, 2 forwards/step)", policy32_cfg)

    def update_cfg():
        """PPO update over a packed self-play rollout, trunk forward/backward in the fused kernels (fused_update=True)."""
        from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
        from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
        from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
        from inversus_reinforcement_learning_amd.training import RolloutCollector
        torch.manual_seed(0)
        b = DeviceEnvBatch(n, opponent="selfplay")
        b.seed(np.arange(n, dtype=np.int64)); b.reset()
        agent = PPOAgent(InversusCNNPolicy(12, H, W, 4), fused_act=True, fused_update=True, batch_size=n, epochs=1)
        col = RolloutCollector(b, agent, FusedPolicy(InversusCNNPolicy(12, H, W, 4).cuda().eval(), actor_only=True))
        T = 8
        col.collect(T); agent.update()                                     # warm-up (allocator, GEMM heuristics)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        col.collect(T); torch.cuda.synchronize(); t1 = time.perf_counter()
        agent.update(); torch.cuda.synchronize(); t2 = time.perf_counter()
        res = {"rollout_steps": T, "minibatch": n, "collect_ms": round((t1 - t0) * 1e3, 1), "update_ms": round((t2 - t1) * 1e3, 1),
               "update_ms_per_minibatch": round((t2 - t1) * 1e3 / T, 2), "update_samples_per_s": round(T * n / (t2 - t1)),
               "train_env_steps_per_s_1_epoch": round(T * n / (t2 - t0))}
        agent.epochs = 4                                                   # the reference's epochs (ppo_agent.py:22): 4 passes over the rollout
        col.collect(T); torch.cuda.synchronize(); t3 = time.perf_counter()
        agent.update(); torch.cuda.synchronize(); t4 = time.perf_counter()
        res["update_ms_4_epochs"] = round((t4 - t3) * 1e3, 1)
        res["train_env_steps_per_s_4_epochs"] = round(T * n / ((t1 - t0) + (t4 - t3)))
        return res
    guard("PPO iteration (selfplay, packed obs, fused act + fused update, minibatch = 16384; 1 epoch and the reference's 4 epochs)", update_cfg)

    def c5_update_cfg():
        """BASELINE config C5 trained through the fused path (round 3): self-play collection on the 32 x 32 board and the PPO update
        through the banded training kernels, minibatch = 16 384; medians of 3 iterations.  The reference-shaped alternative --
        torch autograd over the f32 module -- costs ~390 ms per such minibatch (tools/c5_update_speed.py, profiles/)."""
        from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
        from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
        from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
        from inversus_reinforcement_learning_amd.training import RolloutCollector
        torch.manual_seed(0)
        b = DeviceEnvBatch(n, 32, 32, K, opponent="selfplay")
        b.seed(np.arange(n, dtype=np.int64)); b.reset()
        agent = PPOAgent(InversusCNNPolicy(12, 32, 32, 4), fused_act=True, fused_update=True, batch_size=n, epochs=1)
        col = RolloutCollector(b, agent, FusedPolicy(InversusCNNPolicy(12, 32, 32, 4).cuda().eval(), actor_only=True))
        T = 2
        col.collect(T); agent.update(); torch.cuda.synchronize()
        cs, us = [], []
        for _ in range(3):
            t0 = time.perf_counter(); col.collect(T); torch.cuda.synchronize(); t1 = time.perf_counter()
            agent.update(); torch.cuda.synchronize(); t2 = time.perf_counter()
            cs.append(t1 - t0); us.append(t2 - t1)
        c, u = sorted(cs)[1], sorted(us)[1]
        return {"rollout_steps": T, "minibatch": n, "collect_ms_per_vector_step": round(c * 1e3 / T, 2),
                "selfplay_collector_env_steps_per_s": round(n * T / c), "fused_update_ms_per_minibatch": round(u * 1e3 / T, 2),
                "train_env_steps_per_s_1_epoch": round(n * T / (c + u)), "peak_memory_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}
    guard("C5 PPO iteration: 32x32 self-play collection + fused (banded) update, minibatch = 16384", c5_update_cfg)
    return out


def c4_iteration(torch, np, dist, n, rank, world, off, backend, T=8):
    """BASELINE config C4's real workload on this rank's env shard (runs on EVERY rank; collectives inside): a self-play
    collection of T vector steps (two fused policy forwards per step, no collective), then one PPO epoch over it = T
    minibatches of n samples, each an optimizer step with the flat-bucket gradient all-reduce (parallel.FlatGradReducer:
    the heads' 39.3 MB launched before the trunk's backward, the rest after it).  The all-reduce is priced three ways:
    alone (the bucket's two pieces back to back with nothing else running), EXPOSED (update with the collectives minus the
    same update with them switched off) and overlapped (alone - exposed).  Times are the MAX over ranks."""
    from inversus_reinforcement_learning_amd import parallel
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    from inversus_reinforcement_learning_amd.fused_policy import FusedPolicy
    from inversus_reinforcement_learning_amd.ppo_agent import PPOAgent
    from inversus_reinforcement_learning_amd.training import RolloutCollector
    coll_dev = "cuda" if backend == "nccl" else "cpu"

    def mx(x):
        t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def fence():
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()

    torch.manual_seed(0)                                                   # the same initial replica on every rank
    parallel.seed_device_generator(0, rank)                                # ... and a sampling stream of its own per shard
    b = DeviceEnvBatch(n, opponent="selfplay", env_offset=off)
    b.seed(off + np.arange(n, dtype=np.int64)); b.reset()
    agent = PPOAgent(InversusCNNPolicy(12, H, W, 4), fused_act=True, fused_update=True, batch_size=n, epochs=1)
    if agent.reducer is None:
        raise RuntimeError("c4_iteration needs an initialised process group (the agent built no gradient reducer)")
    parallel.broadcast_parameters(agent.policy)
    agent.fused.refresh()
    col = RolloutCollector(b, agent, FusedPolicy(InversusCNNPolicy(12, H, W, 4).cuda().eval(), actor_only=True))
    col.collect(T); agent.update()                                         # warm-up (allocator, GEMM heuristics, communicator)
    fence(); t0 = time.perf_counter()
    col.collect(T); torch.cuda.synchronize(); t1 = time.perf_counter()
    dist.barrier(); t1b = time.perf_counter()
    agent.update(); torch.cuda.synchronize(); t2 = time.perf_counter()
    digest_ok = True
    try:
        parallel.assert_replicas_in_sync(agent.policy)
    except RuntimeError:
        digest_ok = False
    col.collect(T); agent.reducer.comm = False
    fence(); t3 = time.perf_counter()
    agent.update(); torch.cuda.synchronize(); t4 = time.perf_counter()
    agent.reducer.comm = True
    flat, ne = agent.reducer.flat, agent.reducer.n_early
    reps = 10
    for _ in range(2):
        dist.all_reduce(flat[:ne]); dist.all_reduce(flat[ne:])
    fence(); t5 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(flat[:ne]); dist.all_reduce(flat[ne:])
    torch.cuda.synchronize(); t6 = time.perf_counter()
    collect_s, upd_s, upd_nc_s, ar_s = mx(t1 - t0), mx(t2 - t1b), mx(t4 - t3), mx(t6 - t5)
    exposed = (upd_s - upd_nc_s) / T * 1e3
    alone = ar_s / reps * 1e3
    b.check()
    return {"world": world, "backend": backend, "envs_per_rank": n, "rollout_steps": T, "minibatch_per_rank": n,
            "selfplay_collector_env_steps_per_s": round(world * n * T / collect_s),
            "update_ms_per_minibatch": round(upd_s / T * 1e3, 3),
            "update_ms_per_minibatch_collectives_off": round(upd_nc_s / T * 1e3, 3),
            "allreduce_exposed_ms_per_optimizer_step": round(exposed, 3),
            "allreduce_alone_ms_per_optimizer_step": round(alone, 3),
            "allreduce_overlapped_ms_per_optimizer_step": round(max(0.0, alone - max(0.0, exposed)), 3),
            "gradient_bucket_bytes": int(flat.numel()) * 4, "early_piece_bytes": int(ne) * 4,
            "train_env_steps_per_s_1_epoch": round(world * n * T / (collect_s + upd_s)),
            "replicas_identical_after_update": digest_ok}


MULTI_T = 32


def multi_step_kernel_ms(torch, b, difficulty, T=MULTI_T, launches=8):
    """SURVEY 8(d)'s second figure: the multi-step persistent launch.  ONE inv_rollout launch does T vector steps of the headline's
    env batch into a [T]-deep rollout buffer (T distinct 118 MB observation slots: an HBM stream, like the headline loop);
    HIP events on the launch stream around `launches` such launches; -> kernel ms per VECTOR step, or None on failure
    (best-effort: it must never cost the headline line)."""
    try:
        a1 = torch.stack([b.uniform_actions(4321, t) for t in range(T)])
        o = b._alloc_out((T,), True, False, False, None)
        for _ in range(6):                                                  # untimed: first touch of the 3.8 GB buffer, clock ramp after a short headline loop
            b.rollout(a1, out=o)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            b.rollout(a1, out=o)
        e1.record()
        torch.cuda.synchronize()
        b.check()
        ms = e0.elapsed_time(e1) / (launches * T)
        del o, a1
        torch.cuda.empty_cache()
        return ms
    except Exception as exc:                                                # pragma: no cover
        print(f"multi_step_kernel_ms failed: {type(exc).__name__}: {exc}", file=sys.stderr, flush=True)
        return None


C4_DEADLINE_S = float(os.environ.get("INVERSUS_BENCH_C4_DEADLINE_S", "420"))


def c4_guarded(torch, np, dist, n, rank, world, off, backend, res):
    """Run c4_iteration so that it can neither lose the (already printed) headline line nor strand a rank:
      * an exception on this rank becomes the string "failed: ..." and is announced to the other ranks through the process group's
        key-value store (a TCP store, NOT a collective -- the other ranks may be inside one that will never complete);
      * a watcher thread on every rank polls that key (and a hard deadline, INVERSUS_BENCH_C4_DEADLINE_S, default 420 s, well inside
        the process group's own timeout): when another rank has failed, or the deadline passes while this rank still sits in the
        extra, rank 0 prints the line with the failure recorded and the process leaves with exit code 0 via os._exit (the main
        thread may be blocked inside RCCL / gloo and cannot be unwound).
    -> (c4 dict | "failed: ..." string, clean: every rank completed).
    INVERSUS_BENCH_FAIL_C4 injects faults for tests/test_bench_launch.py: "1" = every rank raises, "rank<k>" = rank k raises,
    "hang<k>" = rank k never enters the extra (sleeps)."""
    import threading
    try:
        store = dist.distributed_c10d._get_default_store()
    except Exception:                                                       # pragma: no cover
        store = None
    key = "inversus_bench_c4_failed"
    done = threading.Event()
    print_lock = threading.Lock()

    def bail(reason):
        with print_lock:
            if done.is_set():
                return
            if rank == 0 and res is not None:
                res.setdefault("extras", {})["C4 selfplay PPO iteration"] = f"failed: {reason}"
                print(json.dumps(res), flush=True)
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(0)

    def watch():
        t_end = time.time() + C4_DEADLINE_S
        while not done.wait(1.0):
            if time.time() > t_end:
                bail(f"watchdog: rank {rank} still inside the C4 extra after {C4_DEADLINE_S:.0f} s")
            if store is not None:
                try:
                    if store.check([key]):
                        bail("another rank failed: " + store.get(key).decode(errors="replace"))
                except Exception as exc:                                    # the store lives in rank 0: gone = rank 0 has left
                    bail(f"rank {rank}: the process group's store went away ({type(exc).__name__})")

    threading.Thread(target=watch, daemon=True).start()
    fault = os.environ.get("INVERSUS_BENCH_FAIL_C4", "")
    try:
        torch.cuda.empty_cache()
        if fault == "1" or fault == f"rank{rank}":
            raise RuntimeError(f"injected failure on rank {rank} (INVERSUS_BENCH_FAIL_C4={fault})")
        if fault == f"hang{rank}":
            time.sleep(10 * C4_DEADLINE_S)
        c4 = c4_iteration(torch, np, dist, n, rank, world, off, backend)
        torch.cuda.empty_cache()
        # every rank is through: agree on it without a collective that a failed peer could block (store counter)
        if store is not None and world > 1:
            store.add("inversus_bench_c4_ok", 1)
            while int(store.add("inversus_bench_c4_ok", 0)) < world:
                if store.check([key]):
                    raise RuntimeError("another rank failed: " + store.get(key).decode(errors="replace"))
                time.sleep(0.05)
        with print_lock:
            done.set()
        return c4, True
    except Exception as exc:
        msg = f"{type(exc).__name__}: {exc}"[:2000]
        with print_lock:
            done.set()
        if store is not None:
            try:
                store.set(key, f"rank {rank}: {msg}")
            except Exception:                                               # pragma: no cover
                pass
        print(f"rank {rank}: C4 extra failed: {msg}", file=sys.stderr, flush=True)
        if rank == 0 and world > 1:
            time.sleep(3.0)                                                 # the store lives here: let the other ranks read the key first
        return f"failed: {msg}", False


def spawn_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher: start N fresh child ranks of this script (one per GPU) and wait.
    The parent has not imported torch or touched the GPU, and it never execs: the children are new processes that
    inherit stdout (rank 0 prints the JSON line).  Returns the exit code (non-zero if any rank failed)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")                # dmabuf IPC (RCCL across processes on this driver)
        env.setdefault("OMP_NUM_THREADS", "4")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    deadline = time.time() + 1500
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in procs:                                        # a dead rank leaves the others in a collective
                        q.terminate()
            if time.time() > deadline:
                rc = rc or 124
                for q in procs:
                    q.terminate()
                deadline = float("inf")
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of --steps vector steps each; the line reports the MEDIAN region (SURVEY 8d: median of 5)")
    ap.add_argument("--difficulty", default="easy", choices=["easy", "hard"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements")
    ap.add_argument("--no-cache-resident", action="store_true",
                    help="skip the second timed loop (the same launches re-writing ONE output buffer); for profiling the headline loop alone, "
                         "so that a rocprofv3 average of k_step4 is the streaming launches only")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) print the cpu_baseline object and exit; no GPU is touched")
    ap.add_argument("--cpu-baseline-quick", action="store_true", help="(internal) with --cpu-baseline-only: the all-threads figure alone")
    ap.add_argument("--no-c4", action="store_true",
                    help="N > 1 only: skip the self-play PPO iteration (collector + fused update + gradient all-reduce) extra")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N>1 (nccl = RCCL; gloo only to rehearse the control flow)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: put every rank on GPU 0 (a 1-GPU box cannot give each rank its own GPU)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group even for one rank (runs the RCCL code path on a 1-GPU box)")
    args = ap.parse_args()

    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.difficulty, quick=args.cpu_baseline_quick)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:                 # no launcher: be the launcher (before any GPU call)
        raise SystemExit(spawn_ranks(args.gpus))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")              # dmabuf IPC (RCCL across processes on this driver); before HIP starts
    import torch
    import torch.distributed as dist
    from inversus_reinforcement_learning_amd.device_env import DeviceEnvBatch
    import numpy as np

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE={world})")
    dev = 0 if args.share_device else local_rank
    if dev >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible); "
                         "--share-device rehearses N ranks on one GPU")
    torch.cuda.set_device(dev)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import datetime
        PG_TIMEOUT = datetime.timedelta(seconds=900)                       # bounded: a collective that cannot complete ends the job, never hangs it
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:                               # only a forced one-rank group gets here without one
            from inversus_reinforcement_learning_amd.parallel import free_port
            os.environ["MASTER_PORT"] = str(free_port())
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev), timeout=PG_TIMEOUT)
        else:
            dist.init_process_group("gloo", timeout=PG_TIMEOUT)
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"

    n = NUM_ENVS
    off = rank * n                                   # contiguous global env range per rank
    b = DeviceEnvBatch(n, W, H, K, opponent="dummy", difficulty=args.difficulty, opp_seed=1, env_offset=off)
    b.seed(off + np.arange(n, dtype=np.int64))       # seed = base 0 + global env index
    b.reset()
    # the synthetic P1 action stream of every step is generated BEFORE the timed region (inputs resident in HBM);
    # every repeat replays the same K action rows on the env state the previous region left behind
    total = args.warmup + args.steps
    a1_all = torch.empty((total, n), dtype=torch.uint8, device="cuda")
    for t in range(total):
        b.uniform_actions(1234, t, out=a1_all[t])
    base = a1_all.data_ptr()
    from inversus_reinforcement_learning_amd.device_env import StepResult

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def over_ranks(x, op):
        if not use_dist:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=op)
        return float(tt.item())

    def gather(x):
        """x of every rank, in rank order, on every rank."""
        if not use_dist:
            return [float(x)]
        v = torch.zeros(world, dtype=torch.float64, device=coll_dev)
        v[rank] = float(x)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        return [float(t) for t in v.tolist()]

    # N > 1: every rank's GPU, gathered -- the line proves by itself that N ranks sat on N distinct devices
    me = gpu_identity(torch)
    me["rank"], me["local_rank"], me["device_index"] = rank, local_rank, dev
    if use_dist:
        gpus = [None] * world
        dist.all_gather_object(gpus, me)
        ranks_seen = dist.get_world_size()
        ids = [g.get("uuid") or f"no-uuid-{g.get('rank')}" for g in gpus]
        if len(set(ids)) != len(ids) and not args.share_device:
            raise SystemExit(f"rank {rank}: {world} ranks but the GPUs are not distinct: {ids} (--share-device rehearses ranks on one GPU)")
    else:
        gpus, ranks_seen = [me], 1

    # HIP events on the launch stream around the region's k_step launches (one pair per region, or per EV_GROUP launches);
    # kernel_ms = event time / launches, so it includes the ~1.5 us dependent-launch boundary between steps of one env batch.
    evg = EV_GROUP if EV_GROUP > 0 else max(1, args.steps)
    groups = [(k, min(k + evg, args.steps)) for k in range(0, args.steps, evg)]

    def timed_loop(slots_n, repeats):
        """`repeats` regions of exactly args.steps vector steps into `slots_n` output buffers (slot = step % slots_n; no allocator in
        the loop).  -> [(wall seconds, max over ranks; kernel ms per step of this rank)]"""
        so = b._alloc_out((slots_n,), True, False, False, None)
        slots = [StepResult(**{k2: (getattr(so, k2)[j] if getattr(so, k2) is not None else None) for k2 in StepResult.__slots__})
                 for j in range(slots_n)]
        stp = [b.make_stepper(slots[j], None, auto_reset=True) for j in range(slots_n)]      # one pre-marshalled ctypes call per vector step
        for t in range(max(args.warmup, slots_n)):
            stp[t % slots_n](base + (t % total) * n)
        regs = []
        for _ in range(max(1, repeats)):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in groups]
            fence()
            t0 = time.perf_counter()
            for (lo, hi), (e0, e1) in zip(groups, ev):
                e0.record()
                for k in range(lo, hi):
                    stp[k % slots_n](base + (args.warmup + k) * n)
                e1.record()
            torch.cuda.synchronize()
            dt_ = time.perf_counter() - t0                # THIS rank's K steps, complete on its GPU
            # the closing barrier + synchronize follow at once, but the barrier's OWN latency (a collective of tens of us -- ms under
            # gloo -- against a 0.5-ms region at the driver's 20 steps) is not part of the K steps: the region's time is the MAX over
            # ranks of the per-rank times (ranks leave the opening barrier together), i.e. when the slowest shard finished
            fence()
            regs.append((over_ranks(dt_, dist.ReduceOp.MAX), sum(s_.elapsed_time(e_) for s_, e_ in ev) / args.steps))
        b.check()
        del so, slots, stp
        return regs

    def median_region(regs):
        order = sorted(range(len(regs)), key=lambda i: regs[i][0])
        return regs[order[len(order) // 2]]          # the median region by wall time (max over ranks)

    # The headline loop: 32 distinct output buffers (3.8 GB of observations: every byte goes to HBM) ...
    STREAM_SLOTS = 32
    regions = timed_loop(STREAM_SLOTS, args.repeats)
    dt, kern_ms = median_region(regions)
    kern_ms_ranks = gather(kern_ms)                  # the same region's k_step time on every rank: a slow rank is visible
    # ... and the same launches, same actions, same protocol re-writing ONE 118 MB buffer, which the 256 MiB Infinity Cache keeps resident
    # (rounds 1-3's headline loop): memory-side-cache writes, not HBM
    cregions = None if args.no_cache_resident else timed_loop(1, args.repeats)
    if cregions is not None:
        cdt, ckern_ms = median_region(cregions)
        ckern_ms_ranks = gather(ckern_ms)

    # ---- the headline line is complete HERE, before the first multi-rank collective of the PPO path runs (round-5 order: the C4 extra
    # used to sit between the measured value and its only print) ----
    res = None
    if rank == 0:
        bytes_per_launch = algorithmic_bytes_per_env_step() * n
        kern_ms_roof = sorted(kern_ms_ranks)[len(kern_ms_ranks) // 2]
        achieved = bytes_per_launch / (kern_ms_roof * 1e-3) / 1e9
        res = {
            "metric": "env-steps/sec at 16384 envs per MI355X",
            "value": world * n * args.steps / dt,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"vs_dummy_{args.difficulty}: {n} envs/GPU, 15x10 grid, K=16, step+obs kernels only "
                                   "(Philox a1 stream, device dummy a2, reward/done, auto-reset, 12-ch obs), every step's outputs kept "
                                   f"({STREAM_SLOTS} distinct output buffers = 3.8 GB of observations: HBM writes)",
                       "envs_per_gpu": n, "sharding": f"env-shards x{world}, no data-path collective",
                       "backend": (args.backend if use_dist else "none"), "gpu": gpus[0], "gpus": gpus, "ranks_seen": ranks_seen,
                       "distinct_gpus": len({g.get("uuid") or f"no-uuid-{g.get('rank')}" for g in gpus})},
            "repeats": len(regions), "protocol": "median of the repeats' wall time (each = exactly `steps` vector steps between barrier + synchronize on both sides; a rank's clock stops at its own synchronize, the region is the MAX over ranks)",
            "ms_per_step_repeats": [round(r[0] / args.steps * 1e3, 6) for r in regions],
            # `value`, `achieved`, `frac`, `kernel_ms` all describe ONE loop: the launches that write 32 distinct buffers, i.e. HBM traffic.
            # The *_cache_resident fields are the same launches re-writing one 118 MB buffer (the Infinity Cache absorbs it): rounds 1-3
            # reported THAT loop as `value` / `frac` (0.81 of "8 TB/s" -- a memory-side-cache number, not an HBM one).
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_from_profiles(args.difficulty, "_streaming"),
                         "regime": "streaming: every launch writes its 118 MB of observations to a buffer of its own (32 slots, 3.8 GB)",
                         "kernel": "k_step4 (four envs per wavefront; csrc/inv_step4.h)" if b.step_kernel != "wave" else "k_step",
                         "kernel_ms": kern_ms_roof, "kernel_ms_events": ("one HIP-event pair around each timed region" if EV_GROUP <= 0 else f"a HIP-event pair per {EV_GROUP} launches"),
                         "kernel_ms_per_rank": [round(x, 6) for x in kern_ms_ranks],
                         "algorithmic_bytes_per_env_step": algorithmic_bytes_per_env_step()},
        }
        if cregions is not None:
            ckern_ms_roof = sorted(ckern_ms_ranks)[len(ckern_ms_ranks) // 2]
            cach = bytes_per_launch / (ckern_ms_roof * 1e-3) / 1e9
            res["value_cache_resident"] = world * n * args.steps / cdt
            res["ms_per_step_cache_resident"] = cdt / args.steps * 1e3
            res["roofline"].update({
                "achieved_cache_resident": cach, "frac_cache_resident": cach / HBM_PEAK_GBS,
                "kernel_ms_cache_resident": ckern_ms_roof, "kernel_ms_cache_resident_per_rank": [round(x, 6) for x in ckern_ms_ranks],
                "traffic_cache_resident": traffic_from_profiles(args.difficulty),
                "regime_cache_resident": "one 118 MB output buffer re-written per step: memory-side-cache (Infinity Cache) writes -- its "
                                         "'fraction of 8 TB/s' is NOT an HBM fraction"})
    if world == 1:
        # SURVEY 8(d)'s SECOND number -- the multi-step persistent launch (inv_rollout: T steps per launch) -- inside `roofline`, so that it
        # survives a parser that keeps the contract's keys only
        ms_multi = multi_step_kernel_ms(torch, b, args.difficulty)
        if rank == 0 and ms_multi is not None:
            am = algorithmic_bytes_per_env_step() * n / (ms_multi * 1e-3) / 1e9
            res["roofline"].update({"kernel_ms_multi_step": ms_multi, "achieved_multi_step": am, "frac_multi_step": am / HBM_PEAK_GBS,
                                    "regime_multi_step": f"inv_rollout: {MULTI_T} steps per launch into a [T] rollout buffer of f32 observations "
                                                         "(HIP events around whole launches / T)"})
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:       # reported at N = 1 only (the other ranks would sit in the barrier)
            res["cpu_baseline"] = cpu_baseline_child(args.difficulty)
        if not args.no_extras and world == 1:
            del b, a1_all
            torch.cuda.empty_cache()
            res["extras"] = side_measurements(torch, np, n)
            two = res["extras"].get("two independent env shards of 8192 on two HIP streams (vs_dummy_easy 15x10 f32, wall clock per full 16384-env step)")
            if isinstance(two, dict):                                     # also inside `roofline`, so that it survives a parser that keeps the contract's keys
                res["roofline"]["two_shards_on_two_streams"] = two

    # N > 1 (or a forced one-rank group): C4's self-play PPO iteration with the gradient all-reduce, on every rank -- AFTER rank 0 has
    # printed and flushed the complete headline line.  A successful extra prints the superset line again (the last line wins); a failed
    # or hung one prints the line with "failed: ..." in its place and every rank leaves with exit code 0 (never a bare raise that
    # strands the other ranks in a collective: c4_guarded).
    if use_dist and not args.no_c4:
        if rank == 0:
            print(json.dumps(res), flush=True)
        c4, clean = c4_guarded(torch, np, dist, n, rank, world, off, args.backend, res)
        if rank == 0:
            res.setdefault("extras", {})["C4 selfplay PPO iteration"] = c4
            print(json.dumps(res), flush=True)
        if not clean:                                                       # a rank failed: the group's collectives cannot be trusted to return
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(0)
    elif rank == 0:
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
