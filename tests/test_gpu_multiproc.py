"""N > 1 on hardware without an 8-GPU node: `bench.py --gpus 2` under torch.distributed.run, both ranks mapped onto the one
GPU of the box (GLGYM_BENCH_SHARE_GPU=1: gloo for the metric gather instead of RCCL; everything else is the multi-GPU code
path -- env sharding by rank, per-rank seeds, barrier + max-over-ranks timing, the single end-of-run gather, rank 0's
aggregated JSON line).  The launcher is started by conftest.pytest_sessionstart before this process initialises the GPU."""
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
