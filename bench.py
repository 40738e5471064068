#!/usr/bin/env python3
"""bench.py -- TomatoEnv env-steps/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--scheme rk4|rk2] [--n-sub S] [--dtype f32|f64]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one batched TomatoEnv.step(): crop-noise-free control update, fused ODE step kernel
(RK4, n_sub sub-steps, reward / violation / info epilogue) and the observation-assembly kernel, for B
independent environments resident in HBM.  Workload = BASELINE.json configs[2]: batch 65 536, fp32, one
synthetic weather year (the Amsterdam KNMI files are not in the reference mount), random actions.
Deviation from the config text: "RK4 with 4 sub-steps" diverges (stiff ODE, lambda_max ~ 0.67 1/s needs
>= 224 sub-steps, tests/test_gpu_parity.py::test_n_sub_4_is_unstable_and_flagged); n_sub = 320 is run -- the count at
which the stability guard stays idle under sustained random actions.  The defaults time 2 000
steps so that `value` is the sustained rate, not the first milliseconds after a reset.  A second, informational leg
times the library's explicit-midpoint sub-stepper on the same workload (`other_scheme`).

Prints ONE JSON line on rank 0.  `value` = all env-steps of all ranks / max-over-ranks wall time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "greenlight-gym2_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# Algorithmic work per RHS evaluation: reference expression graph, no CSE, integer powers
# strength-reduced.  RK4 does 4 evaluations per sub-step.
F_RHS = 1502          # add/mul flops
S_RHS = 219           # quarter-rate special-function ops (rcp, exp, log, sqrt, ...)
PEAK_VALU_TFLOPS = 157.3 / 2      # fp32 vector peak counts FMA = 2; the path's flops are mostly un-fused -> 78.65
PEAK_SPECIAL_TOPS = 157.3 / 2 / 4  # quarter rate
PEAK_HBM_GBPS = 8000.0
BYTES_PER_ENV_STEP = 351          # fp32 algorithmic minimum, without the obs block
# HBM bytes per step_kernel launch from rocprofv3 PMC passes on the DEFAULT workload (B = 65 536, fp32):
# FETCH_SIZE 5 820 KB (x2: gfx950 reports half of the fetched bytes, MI355X_MICROARCH.md "HBM") + WRITE_SIZE 12 384 KB;
# profiles/r01_v12_rk4_pmc_summary.csv.  bench.py cannot collect counters itself, so this is a recorded measurement.
PMC_TRAFFIC_DEFAULT = (2 * 5822.73 + 12384.0) * 1024


DEFAULT_SCHEME = "rk4"
STAGES = {"rk4": 4, "rk2": 2}
N_SUB = {"rk4": 320, "rk2": 360}


def cpu_baseline(n_sub: int, budget_s: float = 10.0):
    """Oracle (plain-C fp64 port of the same scheme, ODE step only) on a bounded sample: ONE host core (the contract's
    `cpu_baseline`) and, beside it, all host cores (threads over env slices; ctypes releases the GIL)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    from oracle import gl_oracle as O
    from gl_gym_amd.parameters import init_default_params
    from gl_gym_amd.utils import synthetic_weather, init_state
    p = init_default_params().astype(np.float64)
    w = synthetic_weather(n_rows=4000)
    rng = np.random.default_rng(1234)
    n = 256
    rows = rng.integers(0, 3000, n)
    X = np.array([init_state(w[r]) * (1 + 1e-3 * rng.standard_normal(28)) for r in rows])
    U = rng.uniform(0, 1, (n, 6))
    D = w[rows]
    O.rk4_batch(X[:8], U[:8], D[:8], p, 900.0, n_sub)       # warm-up
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        O.rk4_batch(X, U, D, p, 900.0, n_sub)
        done += n
    el = time.perf_counter() - t0
    one = {"value": done / el, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": f"{done} env-steps (fp64 C oracle, RK4 n_sub={n_sub}, ODE step only) in {el:.1f} s"}
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:        # a container may be granted fewer CPUs than it can see (cgroup v2 quota)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    per = 64

    def work(_):
        O.rk4_batch(X[:per], U[:per], D[:per], p, 900.0, n_sub)
        return per
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(work, range(cores)))                    # warm-up: library loaded in every thread
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s / 2:
            done += sum(ex.map(work, range(2 * cores)))
        el = time.perf_counter() - t0
    allc = {"value": done / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{done} env-steps on {cores} threads (= CPUs granted to this container) in {el:.1f} s"}
    return one, allc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # Defaults measure SUSTAINED throughput: under continuous load the MI355X settles at a lower clock within ~0.1 s, so a
    # 20-step (20 ms) timed region after an idle period reports the boost-clock burst (about 25 % higher; tools/
    # sustained_rate.py).  2 000 steps = 2.3 s timed after 0.25 s of warm-up; the whole default run takes about 40 s.
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=65536, help="environments per GPU")
    ap.add_argument("--scheme", default=DEFAULT_SCHEME, choices=["rk4", "rk2"],
                    help="sub-stepper: classical RK4 (n_sub 320) or explicit midpoint (n_sub 360); include/glgym.h")
    ap.add_argument("--n-sub", type=int, default=None, help="sub-steps per 900 s env-step (default: 320 rk4 / 360 rk2)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-obs", action="store_true", help="skip the observation-assembly kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step sequence from a captured HIP graph (one launch per step instead of five); the "
                         "step_kernel time for the roofline leg is then taken from W eager warm-up steps")
    ap.add_argument("--no-alt-scheme", action="store_true", help="skip the informational leg with the other sub-stepper")
    ap.add_argument("--uncertainty", type=float, default=0.0, help="crop-parameter noise scale (config 5: 0.2)")
    ap.add_argument("--vecnorm", action="store_true", help="also run the on-device VecNormalize (obs + reward) each step")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or (os.environ.get("GLGYM_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ)
    # Test hook for 1-GPU boxes: GLGYM_BENCH_SHARE_GPU=1 maps every rank to cuda:0 and uses gloo for the (tiny) metric
    # gather, so the N > 1 control flow can be exercised end to end.  Never set by the driver.
    share_gpu = os.environ.get("GLGYM_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local = 0
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local}"))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")

    from gl_gym_amd.tomato_env import TomatoVecEnv
    from gl_gym_amd.utils import synthetic_weather

    B, K, W = args.batch, args.steps, args.warmup
    weather = synthetic_weather(n_rows=35040, dt=900.0, seed=2024)          # one year, shared by all envs
    starts = np.arange(0, 35040 - 5760 - 60, 96)                           # any midnight that leaves a 60-day season
    if args.n_sub is None:
        args.n_sub = N_SUB[args.scheme]
    env = TomatoVecEnv(B, weather=weather, dtype="float64" if args.dtype == "f64" else "float32", n_sub=args.n_sub,
                       scheme=args.scheme,
                       season_length=60, pred_horizon=0.5, device=f"cuda:{local}", seed=666 + rank,
                       start_rows=starts, uncertainty_scale=args.uncertainty, auto_reset=True)
    vn = None
    if args.vecnorm:
        from gl_gym_amd.vec_normalize import VecNormalizeGPU
        vn = VecNormalizeGPU(env, clip_obs=10.0, gamma=0.9631)
    env.reset_tensor()
    env.x_T.mul_(1 + 1e-3 * torch.randn(env.x_T.shape, device=dev,
                                        generator=torch.Generator(device=dev).manual_seed(1234 + rank)).to(env.tdtype))
    gen = torch.Generator(device=dev).manual_seed(666 + rank)
    acts = [torch.rand(B, 6, generator=gen, device=dev) * 2 - 1 for _ in range(min(K + W, 32))]

    def one_step(i, ev=None):
        # the full SB3-semantics step: control update + fused ODE step kernel + observation block + auto-reset of
        # finished envs (new episode start drawn in-kernel, terminal observation kept, their obs rows recomputed)
        env.action_t.copy_(acts[i % len(acts)])
        if ev is not None:
            ev[0].record()
        env._launch_step(raw_control=False)
        if ev is not None:
            ev[1].record()
        if not args.no_obs:
            env._launch_obs(env.obs_t)
        env._launch_reset(env.done_t)
        if not args.no_obs:
            env._launch_obs(env.obs_t, env.done_t, env.term_obs_t)
        if vn is not None:
            vn._call(env.obs_t, env.reward_t, env.done_t)

    warm_events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(W)]
    for i in range(W):
        one_step(i, warm_events[i])
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    replay = None
    if args.graph:
        if vn is not None or args.uncertainty > 0:
            raise SystemExit("--graph: not with --vecnorm / --uncertainty (host-side per-step state)")
        replay = env.capture_step_graph(want_obs=not args.no_obs)
    if env.metrics_t is not None:
        env.metrics_t.zero_()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        if replay is not None:
            replay(acts[(W + i) % len(acts)])
        else:
            one_step(W + i, events[i])
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in (warm_events if replay is not None else events)]))
    m = env.metrics()

    # Informational second leg (never `value`): the same K steps with the other sub-stepper the library offers
    # (include/glgym.h glgym_scheme), timed the same way right after the main leg.
    alt = None
    if not args.no_alt_scheme:
        other = "rk2" if args.scheme == "rk4" else "rk4"
        env.set_scheme(other)
        for i in range(min(W, 2)):
            one_step(i)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for i in range(K):
            one_step(W + i)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        alt = (other, env.n_sub, time.perf_counter() - ta)
        env.set_scheme(args.scheme, args.n_sub)
    # final metric gather: the only collective on this path (RCCL all_gather of 6 doubles per rank)
    from gl_gym_amd.dist import gather_metrics, aggregate
    rows = gather_metrics([elapsed, float(B * K), m.get("sum_reward", 0.0), m.get("n_ode_fail", 0.0),
                           m.get("n_done", 0.0), kern_ms], device=None if share_gpu else dev)
    if rank == 0:
        agg = aggregate(rows)
        t_max, value, kern_ms_max = agg["t_max"], agg["value"], agg["kernel_ms_max"]
        per_gpu_kernel_rate = B / (kern_ms_max * 1e-3)
        flops = STAGES[args.scheme] * args.n_sub * F_RHS
        specials = STAGES[args.scheme] * args.n_sub * S_RHS
        ach_tflops = per_gpu_kernel_rate * flops / 1e12
        ach_tops = per_gpu_kernel_rate * specials / 1e12
        frac = ach_tflops / PEAK_VALU_TFLOPS + ach_tops / PEAK_SPECIAL_TOPS
        obs_bytes = 0 if args.no_obs else 4 * env.obs_dim
        hbm_gbps = per_gpu_kernel_rate * BYTES_PER_ENV_STEP * (2 if args.dtype == "f64" else 1) / 1e9
        out = {
            "metric": "TomatoEnv env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": 1e3 * t_max / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: TomatoEnv batch 65536/GPU, RK4 sub-stepped, synthetic "
                                   "weather year (KNMI Amsterdam files absent), random actions U(-1,1)",
                       "batch_per_gpu": B, "global_batch": B * world, "integrator": args.scheme, "n_sub": args.n_sub,
                       "dt_s": 900,
                       "obs_kernel": not args.no_obs, "obs_dim": env.obs_dim, "auto_reset": True, "hip_graph": bool(args.graph),
                       "vecnormalize": bool(args.vecnorm),
                       "uncertainty_scale": args.uncertainty, "parallelism": f"env-shard x{world} (no data-path collective)",
                       "deviation": "config text says 4 RK4 sub-steps; that is unstable for this stiff ODE (floor 224); the default "
                                    "n_sub keeps the stability guard idle under sustained random actions (DESIGN.md 2)",
                       "scheme": "classical RK4, Strang-split exact harvest flow, slow sub-expressions (LAI optics, crop "
                                 "block, soil chain) evaluated once per sub-step at the predicted midpoint (DESIGN.md 2)"},
            "roofline": {"bound": "valu", "kernel": "step_kernel", "achieved": ach_tflops, "peak": PEAK_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": frac,
                         "traffic": PMC_TRAFFIC_DEFAULT if (B == 65536 and args.dtype == "f32" and not args.uncertainty
                                                            and args.scheme == "rk4")
                         else None,
                         "traffic_note": "HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + "
                                         "WRITE_SIZE, recorded in profiles/r01_v12_rk4_pmc_summary.csv (default workload only)",
                         "note": "path is VALU/transcendental-bound, not HBM/MFMA (SURVEY 8d): achieved = algorithmic "
                                 "add/mul flops (stages*n_sub*1502 per env-step, stages = 4 rk4 / 2 rk2) / mean "
                                 "step_kernel time; frac adds the quarter-rate special-op term (stages*n_sub*219)",
                         "executed_note": "frac > 1 because the kernel executes far less than the reference expression "
                                          "graph (hoisting, CSE, slow sub-expressions once per sub-step): PMC on the "
                                          "default workload (profiles/r01_v12_rk4_pmc_summary.csv) counts 4.90e8 VALU wave-"
                                          "instructions per launch (1 496 per RK4 sub-step per wave, 169 of them "
                                          "transcendental) and SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = 0.883 VALU-busy "
                                          "at one wave per SIMD -- the issue roof that actually binds",
                         "special_ops_achieved_Tops": ach_tops, "special_ops_peak_Tops": PEAK_SPECIAL_TOPS,
                         "kernel_ms": kern_ms_max, "kernel_env_steps_per_s": per_gpu_kernel_rate,
                         "hbm": {"achieved": hbm_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                 "frac": hbm_gbps / PEAK_HBM_GBPS, "algorithmic_bytes_per_env_step": BYTES_PER_ENV_STEP,
                                 "obs_bytes_per_env_step": obs_bytes}},
            "other_scheme": None if alt is None else {
                "integrator": alt[0], "n_sub": alt[1], "value": B * world * K / alt[2], "unit": "env-steps/s",
                "ms_per_step": 1e3 * alt[2] / K,
                "note": "informational: same workload and timing protocol with the library's other sub-stepper (rank-0 "
                        "clock); accuracy of both vs the tight fixtures in DESIGN.md section 2"},
            "sum_reward": agg["sum_reward"], "ode_failures": agg["ode_failures"],
            "episodes_finished": agg["episodes_finished"],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["cpu_baseline_all_cores"] = cpu_baseline(N_SUB["rk4"])      # the RK4 C port at the default sub-step count
        print(json.dumps(out), flush=True)
    env.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
