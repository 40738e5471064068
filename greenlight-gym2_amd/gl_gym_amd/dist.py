"""Multi-GPU plumbing for the env-sharded path: one process per GPU, no data-path collective.

Environments are independent (no cross-env term anywhere in the model), so rank r simply owns a contiguous block of
envs.  The only communication is the end-of-run metric gather: a single small all_gather (RCCL when the backend is
"nccl" on ROCm; gloo in the CPU tests)."""
from __future__ import annotations

from typing import Dict, List, Sequence


def shard_range(global_batch: int, rank: int, world: int):
    """Contiguous env-index range [lo, hi) of `rank`; sizes differ by at most one."""
    base, rem = divmod(int(global_batch), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_metrics(local: Sequence[float], device=None, force_collective: bool = False) -> List[List[float]]:
    """all_gather a short float64 vector from every rank (identity when torch.distributed is not initialised).
    force_collective: run the collective even at world size 1 (so that RCCL has executed on a 1-GPU box)."""
    import torch
    import torch.distributed as dist
    mine = torch.tensor(list(local), dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective):
        return [mine.cpu().tolist()]
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [t.cpu().tolist() for t in out]


def aggregate(rows: List[List[float]]) -> Dict[str, float]:
    """rows[r] = [elapsed_s, env_steps, sum_reward, failed integrations, episodes_done, kernel_ms,
    (optional:) guard retries, refined sub-steps, rank, seed, first-attempt flags: error estimate, branch invariant, cap /
    non-finite, heavy, (round 4:) max scaled state error of the rank's parity leg (10-day fixture; < 0 = not run), failed
    integrations in that leg, (round 6:) the `sustained` continuation: elapsed_s, env_steps, kernel_ms].
    Whole-job throughput = all env-steps / the slowest rank's wall time; the job's parity figure = the worst rank's."""
    t_max = max(r[0] for r in rows)
    steps = sum(r[1] for r in rows)
    col = lambda i: [r[i] if len(r) > i else 0.0 for r in rows]  # noqa: E731
    perr = [r[14] for r in rows if len(r) > 14 and r[14] >= 0.0]
    sus = None
    if all(len(r) > 18 for r in rows):      # every rank ran the continuation: same rule as the main region
        st = max(r[16] for r in rows)
        sus = {"value": sum(r[17] for r in rows) / st, "t_max": st, "env_steps": sum(r[17] for r in rows),
               "kernel_ms_max": max(r[18] for r in rows)}
    return {"sustained": sus,"value": steps / t_max, "t_max": t_max, "env_steps": steps, "sum_reward": sum(r[2] for r in rows),
            "ode_failures": sum(r[3] for r in rows), "episodes_finished": sum(r[4] for r in rows),
            "kernel_ms_max": max(r[5] for r in rows), "guard_retries": sum(col(6)), "refined_substeps": sum(col(7)),
            "first_attempt_flags": {"error_estimate": sum(col(10)), "branch_invariant": sum(col(11)),
                                    "cap_or_nonfinite": sum(col(12)), "heavy": sum(col(13))},
            "max_scaled_err": max(perr) if perr else None, "parity_failed": sum(col(15)),
            "ranks": [{"rank": int(r[8]) if len(r) > 8 else i, "seed": int(r[9]) if len(r) > 9 else None,
                       "elapsed_s": r[0], "env_steps": r[1], "sum_reward": r[2], "kernel_ms": r[5],
                       "max_scaled_err": (r[14] if r[14] >= 0.0 else None) if len(r) > 14 else None}
                      for i, r in enumerate(rows)]}
