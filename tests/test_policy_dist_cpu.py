"""CPU tests: the policy module against the reference's graph (G5), and the N>1 path
(env sharding rule, flat gradient all-reduce, global advantage statistics, PPO update) on
world_size-2 gloo."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_policy_matches_reference_graph():
    import torch
    from inversus_reinforcement_learning_amd.policies import InversusCNNPolicy
    with np.load(os.path.join(GOLDEN, "g5_policy.npz")) as z:
        g = {k: z[k] for k in z.files}
    torch.manual_seed(0)
    pol = InversusCNNPolicy(12, 10, 15, 4).eval()
    sd = pol.state_dict()
    assert list(sd.keys()) == list(g["keys"])                           # checkpoint compatibility (training.py:193-200)
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["shapes"])
    assert sum(p.numel() for p in pol.parameters()) == int(g["n_params"]) == 10249582
    # same construction order => same initial weights under the same seed
    np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], g["sums"], rtol=0, atol=1e-6)
    np.testing.assert_allclose([float(v.double().abs().sum()) for v in sd.values()], g["abs_sums"], rtol=1e-9)
    with torch.no_grad():
        logits, value = pol(torch.from_numpy(g["grid"]), torch.from_numpy(g["extra"]))
    # f32 forward on CPU; tolerance 1e-5 absolute (conv / GEMM summation order may differ between hosts)
    np.testing.assert_allclose(logits.numpy(), g["logits"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(value.numpy(), g["value"], atol=1e-5, rtol=0)
    assert logits.shape == (8, 13) and value.shape == (8, 1)          # tests/test_rl_env_wrapper.py:132-153


def test_shard_range_partitions_the_global_env_range():
    from inversus_reinforcement_learning_amd.parallel import shard_range
    for total, w in [(131072, 8), (16384, 1), (10, 4), (7, 8)]:
        parts = [shard_range(total, r, w) for r in range(w)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == total
        for (o0, c0), (o1, _) in zip(parts, parts[1:]):
            assert o0 + c0 == o1
    assert shard_range(131072, 3, 8) == (3 * 16384, 16384)              # BASELINE config 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_update_equals_single_process():
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", PYTHONPATH=ROOT)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out}"
        assert "DIST_OK" in out, out


def test_forced_single_rank_groups_pick_a_free_port(monkeypatch):
    """a forced one-rank group used a fixed MASTER_PORT (29531 / 29533), so two such runs on one host collided.
    init_from_env now binds port 0 to find a free one; with WORLD_SIZE > 1 a missing MASTER_PORT is an error, not a guess."""
    import torch.distributed as dist
    from inversus_reinforcement_learning_amd import parallel
    a, b = parallel.free_port(), parallel.free_port()
    assert 1024 < a < 65536 and 1024 < b < 65536
    if dist.is_initialized():
        return
    for k in ("MASTER_PORT", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(RuntimeError, match="MASTER_PORT"):
        parallel.init_from_env(backend="gloo")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("INVERSUS_FORCE_DIST", "1")
    assert parallel.init_from_env(backend="gloo") == (0, 1)
    try:
        assert parallel.initialized() and int(os.environ["MASTER_PORT"]) > 1024
    finally:
        dist.destroy_process_group()
