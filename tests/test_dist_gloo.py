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
