"""N > 1 on hardware without an 8-GPU node: `bench.py --gpus 2` under torch.distributed.run, both ranks mapped onto the one
GPU of the box (GLGYM_BENCH_SHARE_GPU=1: gloo for the metric gather instead of RCCL; everything else is the multi-GPU code
path -- env sharding by rank, per-rank seeds, barrier + max-over-ranks timing, the single end-of-run gather, rank 0's
aggregated JSON line).  The launchers are started by conftest.pytest_collection_finish as fresh child processes."""
import json

import pytest

import conftest

pytestmark = pytest.mark.gpu


def test_two_rank_bench_on_one_gpu():
    proc, log = conftest.TWO_RANK["proc"], conftest.TWO_RANK["log"]
    if proc is None:
        pytest.skip("launcher not started (no GPU at session start, or GLGYM_SKIP_TWO_RANK=1)")
    rc = proc.wait(timeout=600)
    text = open(log).read()
    assert rc == 0, text[-3000:]
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, text[-3000:]                       # rank 0 prints exactly one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["batch_per_gpu"] == 4096 and d["config"]["global_batch"] == 8192
    ranks = d["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1]
    assert ranks[0]["seed"] != ranks[1]["seed"]                 # disjoint RNG streams (actions, episode starts)
    assert all(r["env_steps"] == 4096 * 6 for r in ranks)       # each rank stepped its own shard
    assert ranks[0]["sum_reward"] != ranks[1]["sum_reward"]     # ... on different data
    # whole-job value = all env-steps / the slowest rank's time
    t_max = max(r["elapsed_s"] for r in ranks)
    assert abs(d["value"] - 8192 * 6 / t_max) < 1e-6 * d["value"]
    assert d["ode_failures"] == 0
    # the N > 1 line carries the collective's world size, per-rank kernel times and the metric's accuracy half (every rank runs the
    # 10-day fixture after its timed region; the job's figure is the worst rank's)
    assert d["collective"]["world"] == 2 and d["collective"]["ranks_gathered"] == 2
    assert all(r["kernel_ms"] > 0 and 0 < r["max_scaled_err"] < 1e-4 for r in ranks)
    assert d["parity"]["max_scaled_err_10day"] == max(r["max_scaled_err"] for r in ranks) and d["parity"]["failed"] == 0


def test_rccl_at_world_size_one():
    """backend = "nccl" (RCCL on ROCm) on hardware: process-group init with a device id, barrier, and the end-of-run all_gather
    of a DEVICE tensor (gl_gym_amd.dist.gather_metrics(force_collective=True)), one rank under torch.distributed.run with
    GLGYM_FORCE_DIST=1 -- the code path `bench.py --gpus N` takes on an N-GPU node, which this pool cannot offer."""
    proc, log = conftest.RCCL_WS1["proc"], conftest.RCCL_WS1["log"]
    if proc is None:
        pytest.skip("launcher not started (no GPU at collection time, or GLGYM_SKIP_TWO_RANK=1)")
    rc = proc.wait(timeout=600)
    text = open(log).read()
    assert rc == 0, text[-3000:]
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, text[-3000:]
    d = json.loads(lines[0])
    assert d["collective"] == {"backend": "nccl", "world": 1, "ranks_gathered": 1, "note": d["collective"]["note"]}
    assert d["n_gpus"] == 1 and d["ranks"][0]["env_steps"] == 4096 * 6 and d["ode_failures"] == 0


def test_gpus_flag_launches_the_ranks_itself_or_fails_loudly():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts torch.distributed.run itself (fresh child processes,
    before anything touched the GPU) -- on a node with fewer GPUs than asked for it must refuse with exit code 2 and say why,
    NOT run one rank and print n_gpus = 1 (round-2 review, missing item 2); with GLGYM_BENCH_SHARE_GPU=1 it runs both ranks on
    the one GPU."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("node has >= 2 GPUs: the refusal path cannot be provoked")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GLGYM_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, str(conftest.ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "--gpus 2" in r.stderr and not r.stdout.strip(), (r.returncode, r.stdout[-500:], r.stderr[-500:])
    r = subprocess.run([sys.executable, str(conftest.ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch",
                        "2048", "--no-cpu-baseline", "--no-alt-scheme"], capture_output=True, text=True,
                       env=dict(env, GLGYM_BENCH_SHARE_GPU="1"), timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4096 and [x["rank"] for x in d["ranks"]] == [0, 1]
