"""bench.py's N > 1 entry points: `python bench.py --gpus N` must start its own fresh child ranks when no
launcher did, keep working under an external launcher, run the RCCL ("nccl") branch, and print ONE JSON line from rank 0
with `n_gpus = N` and a per-rank k_step time list."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCHER_VARS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "INVERSUS_FORCE_DIST")


def run_bench(*flags, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_VARS}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, cwd=ROOT, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def parse_line(proc, with_c4=False):
    """ONE JSON line from rank 0 -- or, when the C4 extra runs, the headline line first (printed and flushed BEFORE the extra) and the
    superset line last: same headline fields in both, the last one is the result."""
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [json.loads(ln) for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == (2 if with_c4 else 1), proc.stdout[-2000:]
    if with_c4:
        first, last = lines
        assert "C4 selfplay PPO iteration" not in first.get("extras", {})
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "roofline", "config"):
            assert first[k] == last[k], k
    return lines[-1]


def check_streaming_fields(res):
    """Round 4: `value` / `roofline.frac` describe the HBM-streaming loop (32 distinct output buffers); the loop that re-writes ONE buffer
    (the Infinity Cache absorbs it: rounds 1-3's headline) rides beside it as *_cache_resident."""
    r = res["roofline"]
    assert r["bound"] == "hbm" and r["regime"].startswith("streaming")
    assert r["kernel_ms"] > 0 and len(r["kernel_ms_per_rank"]) == res["n_gpus"]
    assert r["achieved"] == pytest.approx(7369 * 16384 / (r["kernel_ms"] * 1e-3) / 1e9, rel=1e-6)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    assert r["kernel_ms_cache_resident"] > 0 and len(r["kernel_ms_cache_resident_per_rank"]) == res["n_gpus"]
    assert r["frac_cache_resident"] == pytest.approx(7369 * 16384 / (r["kernel_ms_cache_resident"] * 1e-3) / 1e9 / r["peak"], rel=1e-6)
    assert res["value_cache_resident"] == pytest.approx(res["n_gpus"] * 16384 / (res["ms_per_step_cache_resident"] * 1e-3), rel=1e-6)
    assert "32 distinct output buffers" in res["config"]["workload"]


def check_gpu_identities(res, world, shared):
    """Round 4: the N > 1 line proves by itself which devices the ranks sat on: one identity per rank (rank order), the world size the
    process group saw, and the number of distinct devices (1 when the ranks of a rehearsal share the box's one GPU)."""
    c = res["config"]
    assert c["ranks_seen"] == world and len(c["gpus"]) == world
    assert [g["rank"] for g in c["gpus"]] == list(range(world))
    assert all("name" in g and "device_index" in g for g in c["gpus"])
    assert c["gpu"] == c["gpus"][0]
    assert c["distinct_gpus"] == (1 if shared else world)


def check_c4_entry(res, world, backend):
    """Round 3: for N > 1 (and a forced one-rank group) rank 0's line prices C4's real workload: self-play collection, the fused
    update, and the gradient all-reduce per optimizer step (alone / exposed / overlapped)."""
    c4 = res["extras"]["C4 selfplay PPO iteration"]
    assert isinstance(c4, dict), c4
    assert c4["world"] == world and c4["backend"] == backend and c4["envs_per_rank"] == 16384
    assert c4["gradient_bucket_bytes"] == 10249582 * 4 and 0 < c4["early_piece_bytes"] < c4["gradient_bucket_bytes"]
    assert c4["selfplay_collector_env_steps_per_s"] > 0 and c4["update_ms_per_minibatch"] > 0
    assert c4["update_ms_per_minibatch_collectives_off"] > 0 and c4["allreduce_alone_ms_per_optimizer_step"] > 0
    assert c4["allreduce_overlapped_ms_per_optimizer_step"] >= 0
    assert c4["replicas_identical_after_update"] is True


def test_self_launch_fails_cleanly_without_gpus():
    """CPU: the parent spawns 2 child ranks (no GPU call in the parent); without GPUs each child exits with a message, the
    parent reaps both and returns non-zero instead of hanging in a rendezvous."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without GPUs")
    p = run_bench("--gpus", "2", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-extras", timeout=180)
    assert p.returncode != 0 and "has no GPU" in p.stderr and p.stdout.strip() == ""


@pytest.mark.gpu
def test_single_gpu_line_carries_both_rooflines_of_survey_8d():
    """N = 1: ONE line; `roofline` holds the single-launch figure (frac, kernel_ms) AND the multi-step persistent one
    (inv_rollout, T steps per launch: frac_multi_step, kernel_ms_multi_step) -- SURVEY 8(d) asks for both."""
    res = parse_line(run_bench("--steps", "60", "--warmup", "5", "--repeats", "2", "--no-cpu-baseline", "--no-extras"))
    r = res["roofline"]
    check_streaming_fields(res)
    assert r["kernel_ms_multi_step"] > 0
    assert r["achieved_multi_step"] == pytest.approx(7369 * 16384 / (r["kernel_ms_multi_step"] * 1e-3) / 1e9, rel=1e-6)
    assert r["frac_multi_step"] == pytest.approx(r["achieved_multi_step"] / r["peak"], rel=1e-9)
    assert 0.2 < r["frac_multi_step"] < 1.0 and "inv_rollout" in r["regime_multi_step"]


@pytest.mark.gpu
def test_self_launched_two_ranks_share_the_gpu_over_gloo():
    """`python bench.py --gpus 2` WITHOUT torchrun: two fresh child ranks (sharing the one GPU of the box over gloo)."""
    res = parse_line(run_bench("--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "50", "--warmup", "5",
                               "--repeats", "3", "--no-cpu-baseline", "--no-extras"), with_c4=True)
    assert res["n_gpus"] == 2 and res["steps"] == 50 and res["scaling"] == "weak" and res["repeats"] == 3
    assert len(res["roofline"]["kernel_ms_per_rank"]) == 2 and all(x > 0 for x in res["roofline"]["kernel_ms_per_rank"])
    assert abs(res["value"] - 2 * 16384 * 50 / (res["ms_per_step"] * 50e-3)) < 1e-3 * res["value"]
    assert len(res["ms_per_step_repeats"]) == 3 and sorted(res["ms_per_step_repeats"])[1] == pytest.approx(res["ms_per_step"], rel=1e-4)
    check_streaming_fields(res)
    check_gpu_identities(res, 2, shared=True)
    check_c4_entry(res, 2, "gloo")


@pytest.mark.gpu
@pytest.mark.parametrize("fault", ["1", "rank1", "rank0", "hang1"])
def test_a_failing_c4_extra_cannot_lose_the_headline_line(fault):
    """Round 5 (verdict 1): the headline line is printed and flushed BEFORE the C4 extra -- the first multi-rank all-reduce of the PPO
    path.  Injected faults in the 2-rank gloo rehearsal: every rank raises / only rank 1 raises while rank 0 sits in the extra's
    collectives / only rank 0 raises / rank 1 never arrives (watchdog deadline).  Every time: exit code 0, a complete headline line
    first, and the same line with `failed: ...` in the extra's place last."""
    p = run_bench("--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "20", "--warmup", "5", "--repeats", "1",
                  "--no-cpu-baseline", "--no-extras",
                  env_extra={"INVERSUS_BENCH_FAIL_C4": fault, "INVERSUS_BENCH_C4_DEADLINE_S": "45" if fault.startswith("hang") else "300"})
    res = parse_line(p, with_c4=True)
    c4 = res["extras"]["C4 selfplay PPO iteration"]
    assert isinstance(c4, str) and c4.startswith("failed: "), c4
    assert res["n_gpus"] == 2 and res["value"] > 0 and len(res["roofline"]["kernel_ms_per_rank"]) == 2
    check_streaming_fields(res)
    check_gpu_identities(res, 2, shared=True)


@pytest.mark.gpu
def test_self_launched_four_ranks_share_the_gpu_over_gloo():
    """Four ranks (env shards 0..3 of 16 384 envs each) on the one GPU: the rank arithmetic beyond two ranks."""
    res = parse_line(run_bench("--gpus", "4", "--backend", "gloo", "--share-device", "--steps", "30", "--warmup", "5",
                               "--repeats", "2", "--no-cpu-baseline", "--no-extras", "--no-c4"))
    assert "extras" not in res
    assert res["n_gpus"] == 4 and len(res["roofline"]["kernel_ms_per_rank"]) == 4
    assert res["config"]["sharding"].startswith("env-shards x4")
    check_gpu_identities(res, 4, shared=True)
    assert abs(res["value"] - 4 * 16384 * 30 / (res["ms_per_step"] * 30e-3)) < 1e-3 * res["value"]


@pytest.mark.gpu
def test_external_launcher_is_respected():
    """Under torch.distributed.run (the driver's N > 1 form) bench.py must NOT spawn again."""
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_VARS}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29577", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device",
                        "--steps", "30", "--warmup", "5", "--repeats", "1", "--no-cpu-baseline", "--no-extras", "--no-c4"],
                       env=env, cwd=ROOT, timeout=600, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    res = parse_line(p)
    assert res["n_gpus"] == 2 and len(res["roofline"]["kernel_ms_per_rank"]) == 2


@pytest.mark.gpu
def test_rccl_branch_runs_with_one_rank():
    """The nccl (= RCCL) branch -- communicator creation on the GPU, barrier, MAX / SUM all-reduces of the timing -- executed
    for real on the 1-GPU box with a one-rank group (--force-dist)."""
    res = parse_line(run_bench("--gpus", "1", "--force-dist", "--backend", "nccl", "--steps", "50", "--warmup", "5", "--repeats", "2",
                               "--no-cpu-baseline", "--no-extras"), with_c4=True)
    assert res["n_gpus"] == 1 and res["config"]["backend"] == "nccl" and len(res["roofline"]["kernel_ms_per_rank"]) == 1
    check_streaming_fields(res)
    check_gpu_identities(res, 1, shared=False)
    check_c4_entry(res, 1, "nccl")                                    # the flat-bucket all-reduce through RCCL (one rank)


@pytest.mark.gpu
def test_rccl_two_ranks_when_two_gpus_are_visible():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU node)")
    res = parse_line(run_bench("--gpus", "2", "--steps", "50", "--warmup", "5", "--repeats", "2", "--no-cpu-baseline", "--no-extras"), with_c4=True)
    assert res["n_gpus"] == 2 and res["config"]["backend"] == "nccl"
    check_gpu_identities(res, 2, shared=False)                         # two ranks on two DISTINCT devices, or bench.py exits


@pytest.mark.gpu
def test_two_ranks_on_one_device_are_refused_without_share_device():
    """Distinct devices are asserted: two ranks that land on the same GPU (here: forced by CUDA_VISIBLE_DEVICES-free LOCAL_RANK = 0 for
    both) make bench.py exit instead of printing a line that looks like a 2-GPU measurement."""
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_VARS}
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "5", "--warmup", "1",
                               "--repeats", "1", "--no-cpu-baseline", "--no-extras", "--no-c4"],
                              env=dict(env, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)),
                              cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode != 0 for p in procs)
    assert any("GPUs are not distinct" in e for _, e in outs)
    assert not any(ln.startswith("{") for o, _ in outs for ln in o.splitlines())      # no JSON line that could pass for a 2-GPU measurement


@pytest.mark.gpu
def test_training_loop_over_rccl_with_one_rank():
    """training.py's collectives (parameter broadcast, flat gradient all-reduce launched from the backward hooks, advantage
    statistics, episode-stat all-gather, parameter-digest check) through RCCL on a one-rank group."""
    code = ("import os; os.environ['INVERSUS_FORCE_DIST']='1'; os.environ['MASTER_PORT']='29579'\n"
            "from inversus_reinforcement_learning_amd import training, parallel\n"
            "import torch.distributed as dist\n"
            "r = training.train_selfplay(num_envs=64, total_steps=64*128, log_dir=None, seed=2, verbose=False, agent_kwargs=dict(batch_size=1024, epochs=1))\n"
            "assert dist.is_initialized() and dist.get_backend() == 'nccl', dist.get_backend()\n"
            "print('RCCL_TRAIN_OK', r['steps'], r['episodes'], r['param_abs_sum'])\n"
            "dist.destroy_process_group()\n")
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_VARS}
    env["PYTHONPATH"] = ROOT
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, timeout=600, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0 and "RCCL_TRAIN_OK 8192" in p.stdout, p.stdout[-3000:]


@pytest.mark.gpu
def test_fused_training_with_two_ranks_keeps_the_replicas_identical():
    """The fused update on two ranks (gloo, both ranks on the one GPU): the gradient path that bypasses the autograd engine
    (FusedTrainPolicy.loss_and_grads) starts the heads' all-reduce explicitly (FlatGradReducer.start_early), reduces the rest after
    the trunk's backward and steps the flat Adam kernel with 1 / world_size; after three updates both ranks hold the same parameters
    (training.py additionally checks the digests across ranks while it runs)."""
    code = ("import os\n"
            "from inversus_reinforcement_learning_amd import training, parallel\n"
            "import torch.distributed as dist\n"
            "r = training.train_selfplay(num_envs=256, total_steps=256*128*3, log_dir=None, seed=5, verbose=False,\n"
            "                            agent_kwargs=dict(fused_act=True, fused_update=True, batch_size=4096, epochs=1))\n"
            "assert dist.is_initialized() and dist.get_world_size() == 2\n"
            "print('FUSED2_OK', os.environ['RANK'], r['steps'], repr(r['param_abs_sum']))\n"
            "dist.destroy_process_group()\n")
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_VARS}
    env.update(PYTHONPATH=ROOT, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29583", INVERSUS_DIST_BACKEND="gloo",
               INVERSUS_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=900)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    lines = []
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "FUSED2_OK" in o, o[-3000:]
        lines.append([ln for ln in o.splitlines() if ln.startswith("FUSED2_OK")][0].split())
    assert {ln[1] for ln in lines} == {"0", "1"} and lines[0][2] == lines[1][2]
    assert lines[0][3] == lines[1][3], lines                       # identical parameter digests on both ranks
