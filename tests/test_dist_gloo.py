"""CPU test of the N > 1 path: two processes, gloo backend, 127.0.0.1 rendezvous (the GPU path uses RCCL for the
same single all_gather; there is no data-path collective to test)."""
import os
import socket

import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "greenlight-gym2_amd"))
    import torch.distributed as dist
    from gl_gym_amd.dist import shard_range, gather_metrics, aggregate
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(131073, rank, world)
    rows = gather_metrics([1.0 + rank, float(hi - lo) * 5, 10.0 * rank, 0.0, 1.0, 2.0 + rank])
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, lo, hi, aggregate(rows)))


def test_shard_and_gather_two_ranks():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, a0), (r1, lo1, hi1, a1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 65537, 65537, 131073)          # contiguous, disjoint, complete
    assert a0 == a1                                                    # every rank sees the same aggregate
    assert a0["env_steps"] == 131073 * 5 and a0["t_max"] == 2.0 and a0["value"] == 131073 * 5 / 2.0
    assert a0["kernel_ms_max"] == 3.0 and a0["sum_reward"] == 10.0


def test_shard_range_properties():
    from gl_gym_amd.dist import shard_range
    for B in (1, 7, 64, 65536, 524288):
        for w in (1, 2, 3, 8):
            rs = [shard_range(B, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == B
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1


def _worker_config4(rank, world, port, q):
    """One rank of BASELINE configs[3]: batch 524 288 over 8 ranks, 10-day horizon (961 env-steps) -- the bookkeeping only."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "greenlight-gym2_amd"))
    import torch.distributed as dist
    from gl_gym_amd.dist import shard_range, gather_metrics, aggregate
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(524288, rank, world)
    n_steps = 961
    # bench.py's 19-entry vector: elapsed, env-steps, sum reward, failed, episodes, kernel ms, retries, refined, rank, seed, four
    # first-attempt flag counts, max scaled error of the rank's parity leg, failed integrations in it, and (round 6) the `sustained`
    # continuation: its elapsed time, env-steps and kernel time
    mine = [10.0 + 0.25 * rank, float((hi - lo) * n_steps), 1.5 * rank, float(rank == 3), float(hi - lo), 0.9 + 0.01 * rank,
            2.0, 100.0 * rank, float(rank), float(666 + rank), 1.0, 0.0, 0.0, float(rank == 5), 2.0e-5 + 1.0e-6 * rank, 0.0,
            1.0 + 0.01 * rank, float((hi - lo) * (1500 + rank)), 0.91 + 0.01 * rank]
    rows = gather_metrics(mine)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, lo, hi, aggregate(rows)))


def test_config4_bookkeeping_eight_ranks():
    """BASELINE configs[3] as the driver would launch it (8 ranks, one all_gather of 19 doubles at the end): contiguous 65 536-env
    shards, whole-job env-steps / slowest rank, the worst rank's max scaled error, per-rank kernel times -- on gloo, no GPU."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 8
    procs = [ctx.Process(target=_worker_config4, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [(lo, hi) for _, lo, hi, _ in res] == [(65536 * r, 65536 * (r + 1)) for r in range(world)]
    a = res[0][3]
    assert all(r[3] == a for r in res)                                 # every rank sees the same aggregate
    assert a["env_steps"] == 524288 * 961 and a["t_max"] == 11.75 and a["value"] == 524288 * 961 / 11.75
    assert abs(a["max_scaled_err"] - 2.7e-5) < 1e-12 and a["parity_failed"] == 0          # the worst rank's
    assert a["ode_failures"] == 1 and a["guard_retries"] == 16 and a["refined_substeps"] == 2800
    assert a["first_attempt_flags"] == {"error_estimate": 8.0, "branch_invariant": 0.0, "cap_or_nonfinite": 0.0, "heavy": 1.0}
    assert len(a["ranks"]) == 8 and [r["rank"] for r in a["ranks"]] == list(range(8))
    assert [round(r["kernel_ms"], 2) for r in a["ranks"]] == [round(0.9 + 0.01 * r, 2) for r in range(8)] and a["kernel_ms_max"] == 0.97
    assert a["ranks"][7]["seed"] == 673 and abs(a["ranks"][7]["max_scaled_err"] - 2.7e-5) < 1e-12
    # the `sustained` block: every rank's continuation env-steps / the slowest rank's continuation time
    su = a["sustained"]
    assert su["t_max"] == 1.07 and su["env_steps"] == 65536 * sum(1500 + r for r in range(8)) and abs(su["kernel_ms_max"] - 0.98) < 1e-12
    assert su["value"] == su["env_steps"] / 1.07
    # a rank that ran no continuation (an older 16-entry vector): the job reports none
    from gl_gym_amd.dist import aggregate
    assert aggregate([[1.0, 10.0, 0.0, 0.0, 0.0, 0.5] + [0.0] * 10, [1.0, 10.0, 0.0, 0.0, 0.0, 0.5] + [0.0] * 13])["sustained"] is None
