#!/usr/bin/env python3
"""bench.py -- TomatoEnv env-steps/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--scheme ls5|rk4|rk3|rk2] [--n-sub S] [--window W] [--dtype f32|f64]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one batched TomatoEnv.step(): crop-noise-free control update, fused ODE step kernel
(fourth-order Runge-Kutta sub-steps, reward / violation / info epilogue) and the observation-assembly kernel, for B
independent environments resident in HBM.  Workload = BASELINE.json configs[2]: batch 65 536, fp32, one
synthetic weather year (the Amsterdam KNMI files are not in the reference mount), random actions.
Deviation from the config text: "RK4 with 4 sub-steps" diverges (stiff ODE: the cover pair's conduction alone, 0.65 1/s, needs
>= 224 classical sub-steps, tests/test_gpu_parity.py::test_n_sub_4_is_refined_to_what_the_ode_needs_or_flagged).  Since round 4 that
one linear mode is integrated exactly and classical RK4 runs at n_sub = 240 -- the NOMINAL count of the stability-controlled
sub-stepper: environments whose local rate bound needs more take more, smaller sub-steps.  Since round 5 `value`
is the library's default scheme "ls5": a five-stage FOURTH-order Runge-Kutta scheme in 2N-storage form whose stability interval per
right-hand side is 1.4x classical RK4's (n_sub 128, 640 right-hand sides per env-step against 960 at the same accuracy on every
fixture); classical RK4 at 240 is timed in the informational second leg (`other_scheme`) of the same line, and
the PARITY configuration (n_sub 192, one sub-step per window: inside the band of the reference solver's tolerances) in a third
(`parity_config`).  The defaults time 2 000 steps so that `value` is the sustained rate, not the first milliseconds after a reset.

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
# fp32 vector peaks of MI355X (MI355X_MICROARCH.md: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz): 157.3 TFLOP/s is the plain (non-packed)
# FMA peak -- a wave64 v_fma_f32 issues every 2 cycles, and v_pk_fma_f32 takes 4, so packing buys no rate; add / mul code tops out
# at half of it, transcendentals (quarter rate: 8 cycles per wave64 op) at 19.7 Tops.
PEAKS_TFLOPS = {"fma": 157.3, "non_fma": 78.65, "transcendental_Tops": 19.7}
PEAK_HBM_GBPS = 8000.0
BYTES_PER_ENV_STEP = 351          # fp32 algorithmic minimum, without the obs block
N_SIMD = 1024
# Issue cost of one wave64 vector instruction on a SIMD, in cycles.  Two rulers:
#   guide    (MI355X_MICROARCH.md: SIMD-32, wave64 op 2 cycles, transcendental 8, 2.4 GHz)            -> roofline.frac
#   measured (tools/microbench.hip with verified wave placement, profiles/r02_microbench_issue_rates.txt: with >= 2 co-resident
#            waves a SIMD issues a plain fp32 op every 2.3-2.7 cycles and a transcendental every 7.7; clock = the one recorded
#            under the profiler)                                                                       -> roofline.frac_measured_ruler
# A LONE wave -- all this kernel can have at B = 65 536 = 1 024 waves on 1 024 SIMDs -- gets one issued only every 5.0 / 8.4.
CYC_PLAIN_GUIDE, CYC_TRANS_GUIDE, CLOCK_GUIDE_HZ = 2.0, 8.0, 2.4e9
CYC_PLAIN, CYC_TRANS = 2.3, 7.7
# Recorded rocprofv3 PMC measurements of step_kernel per launch at B = 65 536, one entry per shipped variant (bench.py cannot
# collect counters itself).  Written by tools/pmc_summary.py from the separate --pmc passes of tools/profile_round.sh /
# tools/profile_variants.sh; the summaries they come from are committed next to it.
PMC_FILES = [ROOT / "profiles" / "r06_pmc_constants.json", ROOT / "profiles" / "r05_pmc_constants.json", ROOT / "profiles" / "r04_pmc_constants.json", ROOT / "profiles" / "r03_pmc_constants.json"]


def load_pmc(variant: str):
    """-> the recorded counters of `variant` ("f32_rk4", "f32_rk3", "f32_rk2", "f64_rk4", "f32_rk4_config5"), or None.
    r03 file: {variant: {...}}; the r02 file holds the default variant only."""
    for f in PMC_FILES:
        try:
            d = json.loads(f.read_text())
        except (OSError, ValueError):
            continue
        if variant in d:
            return d[variant]
        if "SQ_INSTS_VALU" in d and variant == "f32_rk4":
            return d
    return None


def workload_label(args, B, world):
    """Which BASELINE.json config this run is (index into `configs`), or that it is a variant of the headline."""
    if args.scheme in ("ls5", "rk4") and not args.vecnorm and not args.window:        # both fourth-order Runge-Kutta
        if args.uncertainty:
            return "BASELINE configs[4]" if (B == 65536 and args.dtype == "f32") else "variant of BASELINE configs[4]"
        if args.dtype == "f64":
            return "BASELINE configs[1]" if B == 4096 else "variant of BASELINE configs[1]"
        if B == 65536:
            return "BASELINE configs[2]" if world == 1 else "BASELINE configs[2] per GPU (configs[3] at 8 GPUs)"
    return "variant of BASELINE configs[2]"


DEFAULT_SCHEME = "ls5"
STAGES = {"rk4": 4, "rk2": 2, "rk3": 3, "ls5": 5}
N_SUB = {"rk4": 240, "rk2": 336, "rk3": 270, "ls5": 128}
PARITY_CFG = {"ls5": (192, 1), "rk4": (640, 0), "rk3": (720, 0), "rk2": (896, 0)}      # (n_sub, window): gl_gym_amd/_lib.py PRESETS["parity"]


def cpu_baseline(n_sub: int, budget_s: float = 8.0, order: int = 4, window: int = 4):
    """The CPU path beside the GPU number, on a bounded sample of the bench workload's env-steps (ODE step only), timed on THIS
    box's host cores.  Four figures:
      cpu_baseline                 the reference-like integrator: variable-order BDF + modified Newton + reused finite-difference
                                   Jacobian at rtol = atol = 1e-6 (oracle/gl_oracle.c gl_oracle_bdf -- the algorithm family and
                                   tolerances of greenlight_model.cpp:46-63; CasADi / CVODES themselves are absent), ONE core --
                                   how the reference runs an env;
      cpu_baseline_all_cores       the same on every core granted to this container (threads over env slices);
      cpu_baseline_same_scheme     the kernels' own scheme as a plain-C fp64 port (the checker's rk_sc_guarded at the bench's
                                   scheme / n_sub / window), one core;
      cpu_baseline_same_scheme_all_cores."""
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
    U = rng.uniform(0, 1, (n, 6))
    D = w[rows]
    # spun-up states (one env-step from the reset state under the same inputs): at the reset state every exchange law sits on
    # its kink, which is not what a running env looks like
    X = O.rk4_batch(np.array([init_state(w[r]) * (1 + 1e-3 * rng.standard_normal(28)) for r in rows]), U, D, p, 900.0, 256)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:        # a container may be granted fewer CPUs than it can see (cgroup v2 quota)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass

    def bdf_slice(lo, hi):
        return O.bdf_batch(X[lo:hi], U[lo:hi], D[lo:hi], p, 900.0, 1e-6, 1e-6)[1]

    def timed(one_call, per_call, workers, budget):
        """one_call(k) does per_call env-steps and returns its RHS evaluations (or None)"""
        with ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(one_call, range(workers)))                # warm-up: library loaded in every thread
            done, evals, t0 = 0, 0, time.perf_counter()
            while time.perf_counter() - t0 < budget:
                r = list(ex.map(one_call, range(2 * workers if workers > 1 else 1)))
                done += per_call * len(r)
                evals += sum(v for v in r if v is not None)
            return done, evals, time.perf_counter() - t0

    per = 32
    done, evals, el = timed(lambda k: bdf_slice((k * per) % n, (k * per) % n + per), per, 1, budget_s)
    bdf_one = {"value": done / el, "unit": "env-steps/s", "cores": 1, "kind": "port",
               "algorithm": "variable-order BDF, rtol = atol = 1e-6 (the reference integrator's family and tolerances)",
               "rhs_evaluations_per_env_step": evals / done,
               "sample": f"{done} env-steps (fp64 C restatement: variable-order BDF, rtol = atol = 1e-6, ODE step only) in {el:.1f} s"}
    done, evals, el = timed(lambda k: bdf_slice((k * per) % n, (k * per) % n + per), per, cores, budget_s / 2)
    bdf_all = {"value": done / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
               "algorithm": "variable-order BDF, rtol = atol = 1e-6",
               "rhs_evaluations_per_env_step": evals / done,
               "sample": f"{done} env-steps on {cores} threads (= CPUs granted to this container) in {el:.1f} s"}

    def rk(k):      # the checker's restatement of the kernels' own integration (stability control, guard ladder), tuple by tuple
        for i in range(16):
            O.rk_sc_guarded(X[i], U[i], D[i], p, 900.0, n_sub, order, window)
    stages = {4: 4, 5: 5, 3: 3, 2: 2}[order]
    alg = "the kernels' own scheme (order code %d, n_sub %d, window %d: stability-controlled sub-stepper + guard)" % (order, n_sub, window)
    done, _, el = timed(rk, 16, 1, budget_s / 2)
    rk_one = {"value": done / el, "unit": "env-steps/s", "cores": 1, "kind": "port", "algorithm": alg, "rhs_evaluations_per_env_step": stages * n_sub,
              "sample": f"{done} env-steps (fp64 C oracle, n_sub={n_sub}, ODE step only) in {el:.1f} s"}
    done, _, el = timed(rk, 16, cores, budget_s / 2)
    rk_all = {"value": done / el, "unit": "env-steps/s", "cores": cores, "kind": "port", "rhs_evaluations_per_env_step": stages * n_sub,
              "sample": f"{done} env-steps on {cores} threads in {el:.1f} s"}
    return {"cpu_baseline": bdf_one, "cpu_baseline_all_cores": bdf_all, "cpu_baseline_same_scheme": rk_one,
            "cpu_baseline_same_scheme_all_cores": rk_all}


def parity_leg(args, dev, layout, n_sub=None, window=None, occupancy=0, fixture="rollout_10day.npz", want_abs=False):
    """-> (max scaled state error over the 10-day fixture rollout, failed integrations, note).  64 identical environments (one
    wavefront of the one-lane kernel / 4 of the quad kernel); the fixture travels with the repository.  occupancy = 2: the timed
    batch ran the two-waves-per-SIMD build of the one-lane kernel (B >= 131 072 or GLGYM_OCC=2) -- the accuracy half is then
    measured on THAT build (glgym_set_occupancy), not on the one-wave build a 64-environment handle would pick by itself."""
    import numpy as np
    import torch
    from gl_gym_amd.tomato_env import TomatoVecEnv
    g = np.load(ROOT / "tests" / "golden" / fixture)
    acts, w, XR = g["actions"], g["weather"], g["X"]
    env = TomatoVecEnv(64, weather=w, dtype="float64" if args.dtype == "f64" else "float32", n_sub=args.n_sub if n_sub is None else n_sub,
                       window=args.window if window is None else window, scheme=args.scheme,
                       season_length=(len(acts) - 1) // 96, pred_horizon=0.5, device=str(dev), auto_reset=False)
    if layout:
        env.set_layout(layout)               # the layout the timed batch ran (handle state, glgym_set_layout)
    if occupancy:
        env.set_occupancy(occupancy)         # ... and the register build (glgym_set_occupancy)
    env.reset_tensor()
    a_all = torch.as_tensor(acts, device=dev)
    X = [env.x[0].double().cpu().numpy()]
    for k in range(len(acts)):
        env.step_tensor(a_all[k][None].expand(64, 6).contiguous(), want_obs=False)
        X.append(env.x[0].double().cpu().numpy())
    failed = env.metrics().get("n_ode_fail", 0.0)
    env.close()
    X = np.array(X)
    scale = np.maximum(np.abs(XR), 1e-3 * np.abs(XR).max(axis=0, keepdims=True))
    scale[scale == 0] = 1.0
    if want_abs:        # + the largest TEMPERATURE error in kelvin and the metric restricted to the states that are not temperatures
        temps = list(range(2, 15)) + list(range(17, 22)); others = [0, 1, 15, 16, 22, 23, 24, 25, 26]
        return (float(np.max(np.abs(X - XR) / scale)), float(failed), float(np.abs(X - XR)[:, temps].max()),
                float((np.abs(X - XR) / scale)[:, others].max()))
    return float(np.max(np.abs(X - XR) / scale)), float(failed), "%s, %d steps" % (fixture, len(acts))


def main():
    ap = argparse.ArgumentParser()
    # Defaults measure SUSTAINED throughput: 2 000 steps = 1.3 s timed after 0.13 s of warm-up; the whole default run takes about
    # 40 s (most of it the CPU-baseline legs).  A short region reads a little high: the driver's `--steps 20 --warmup 5` (12.6 ms
    # timed) gave 1.042e8 in round 5 against 1.02e8 over 2 000 steps, +2 % (not the +25 % this comment claimed in rounds 1-5, which
    # was the round-1 kernel's boost-clock burst).  Whatever K is, the line also carries a `sustained` block: the same loop
    # continued until >= 1 s has been timed (below), so a short driver run records the sustained figure itself.
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=65536, help="environments per GPU")
    ap.add_argument("--scheme", default=DEFAULT_SCHEME, choices=["ls5", "rk4", "rk2", "rk3"],
                    help="sub-stepper: the five-stage fourth-order 2N scheme (n_sub 128), classical RK4 (240), the three-stage third-order "
                         "scheme (270) or the midpoint rule (336); include/glgym.h")
    ap.add_argument("--n-sub", type=int, default=None, help="sub-steps per 900 s env-step (default: 128 ls5 / 240 rk4 / 270 rk3 / 336 rk2)")
    ap.add_argument("--window", type=int, default=0, help="nominal sub-steps per tier-2b window (0 = the scheme's own: ls5 2, rk4 4, rk3 3, rk2 4)")
    ap.add_argument("--no-parity-config", action="store_true", help="skip the leg that times the PARITY configuration (ls5: n_sub 192, window 1)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-obs", action="store_true", help="skip the observation-assembly kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the accuracy leg (10-day fixture rollout after the timed region)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step sequence from a captured HIP graph (one launch per step instead of five); the "
                         "step_kernel time for the roofline leg is then taken from W eager warm-up steps")
    ap.add_argument("--no-alt-scheme", action="store_true", help="skip the informational leg with the other sub-stepper")
    ap.add_argument("--no-sustained", action="store_true", help="skip the `sustained` continuation (>= 1 s timed) after the K steps")
    ap.add_argument("--uncertainty", type=float, default=0.0, help="crop-parameter noise scale (config 5: 0.2)")
    ap.add_argument("--vecnorm", action="store_true", help="also run the on-device VecNormalize (obs + reward) each step")
    ap.add_argument("--gpus", type=int, default=1, help="GPUs of this node: one rank per GPU; without a launcher bench.py starts "
                                                        "torch.distributed.run itself")
    args = ap.parse_args()

    # ---- N > 1 without a launcher: become the launcher.  Nothing in THIS process has touched the GPU yet (no HIP call, no
    # torch.cuda.is_available(); device_count() does not initialise the runtime on this image), the ranks are fresh child
    # processes, their one JSON line and exit code are forwarded.  Never an exec from a process that initialised the GPU.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        import torch
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus and os.environ.get("GLGYM_BENCH_SHARE_GPU") != "1":
            print(f"bench.py: --gpus {args.gpus} but this node has {n_dev} GPU(s) (GLGYM_BENCH_SHARE_GPU=1 maps every rank to "
                  "cuda:0 for control-flow tests)", file=sys.stderr)
            sys.exit(2)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus):
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        sys.exit(2)
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
            # "nccl" IS RCCL on ROCm.  Also taken at world size 1 under GLGYM_FORCE_DIST=1 (tests/test_gpu_multiproc.py), so that
            # RCCL init, barrier and the all_gather of a device tensor have run on hardware even where only one GPU exists.
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
                       scheme=args.scheme, window=args.window,
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

    def one_step(i, ev=None):
        # the full SB3-semantics step: control update + fused ODE step kernel + observation block + auto-reset of
        # finished envs (new episode start drawn in-kernel, terminal observation kept, their obs rows recomputed).
        # The random policy draws a FRESH U(-1, 1) action block every step, inside the timed region.  (Rounds 1-3 cycled through 32 pre-drawn blocks: with period-32 increments every control drifts
        # to one of its bounds and stays there -- half of the environments with the vents fully open is not what a random
        # policy does.)  One device kernel: uniform_ draws U(-1, 1) in place.
        env.action_t.uniform_(-1.0, 1.0, generator=gen)
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
            replay(torch.rand(B, 6, generator=gen, device=dev) * 2 - 1)
        else:
            one_step(W + i, events[i])
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in (warm_events if replay is not None else events)]))
    m = env.metrics()

    # ---- `sustained`: the SAME loop continued (same env, same action stream, no reset) until at least SUSTAIN_S seconds have been
    # timed in one region; `steps` / `value` / `ms_per_step` above keep their meaning (exactly K steps).  When the K steps already
    # lasted that long the block restates them.  Bracketed like the main region; every rank takes its own step count from its own
    # clock, the job's figure = all continuation env-steps / the slowest rank's time (gathered below).
    SUSTAIN_S, SUSTAIN_MAX_STEPS = 1.0, 50000
    if elapsed >= SUSTAIN_S or args.no_sustained or replay is not None:
        sus = (elapsed, float(B * K), kern_ms, K, False) if not args.no_sustained else None
    else:
        n_more = int(min(SUSTAIN_MAX_STEPS, max(K, np.ceil(1.1 * SUSTAIN_S / (elapsed / K)))))
        sev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(n_more, 2000))]
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for i in range(n_more):
            one_step(W + K + i, sev[i] if i < len(sev) else None)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        ts1 = time.perf_counter() - ts0
        sus = (ts1, float(B * n_more), float(np.mean([a.elapsed_time(b) for a, b in sev])), n_more, True)
        m = env.metrics()                      # the integrator-event counters then cover both regions

    # Informational second leg (never `value`): the same K steps with the other sub-stepper the library offers
    # (include/glgym.h glgym_scheme), timed the same way right after the main leg.
    alt = None
    if not args.no_alt_scheme:
        other = "rk4" if args.scheme != "rk4" else "ls5"
        env.set_scheme(other, N_SUB[other], 0)
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
        env.set_scheme(args.scheme, args.n_sub, args.window)
    # ---- the metric's second half (BASELINE.json: "max |delta state| vs CasADi ref"; SURVEY 8d / 8e: max_scaled_err in the gathered
    # vector): after timing, the same library, dtype, scheme and n_sub run the 961 steps of the 10-day fixture
    # (tests/golden/rollout_10day.npz: Bleiswijk weather, delta-u-bounded random actions, truth = Radau rtol = atol = 1e-11 of the
    # reference-text right-hand side -- the reference's CVODES itself is not available) through the SAME kernel
    # layout as the timed batch, on every rank; the state error is max |x - x_truth| / max(|x_truth|, 1e-3 max_t |x_truth|).
    parity = (-1.0, 0.0, None)
    lay = "one" if (B > 16384 and args.dtype != "f64") else None                                # (fp64 has one layout)
    # the register build the timed batch took (glgym.hip launch_step: default parameters, shared crop block, one lane per environment,
    # two or more wavefronts per SIMD or GLGYM_OCC=2)
    occ_env0 = os.environ.get("GLGYM_OCC", "")
    occ_timed = 2 if (lay == "one" and not args.uncertainty and (occ_env0 == "2" or (occ_env0 != "1" and B >= 131072))) else 0
    holdout = None
    if not args.no_parity:
        parity = parity_leg(args, dev, lay, occupancy=occ_timed)
        if rank == 0 and (ROOT / "tests" / "golden" / "holdout_gl2010_random.npz").exists():
            # the same leg on a fixture generated AFTER the sub-stepper's constants were frozen (round 6; tests/test_gpu_holdout.py)
            holdout = parity_leg(args, dev, lay, occupancy=occ_timed, fixture="holdout_gl2010_random.npz", want_abs=True)
    # ---- the accuracy-speed trade on record: the same workload at the PARITY configuration (inside the band the reference solver's
    # tolerances keep from the tight solution; include/glgym.h), timed the same way over min(K, 200) steps, and its own 10-day error
    pcfg = None
    if not args.no_parity_config:
        pn, pw = PARITY_CFG[args.scheme]
        env.set_scheme(args.scheme, pn, pw)
        Kp = min(K, 200)
        for i in range(2):
            one_step(i)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for i in range(Kp):
            one_step(W + i)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        tp = time.perf_counter() - tp
        env.set_scheme(args.scheme, args.n_sub, args.window)
        perr = parity_leg(args, dev, lay, pn, pw, occupancy=occ_timed) if not args.no_parity else (-1.0, 0.0, None)
        pcfg = {"integrator": args.scheme, "n_sub": pn, "window": pw if pw else "scheme default", "steps": Kp,
                "value": B * world * Kp / tp, "unit": "env-steps/s", "ms_per_step": 1e3 * tp / Kp,
                "max_scaled_err_10day": None if perr[0] < 0 else perr[0], "failed": perr[1],
                "note": "informational (rank-0 clock; rank 0's fixture error): the configuration that sits inside the 1.3e-5 band a BDF solve at "
                        "the reference's tolerances (greenlight_model.cpp:51-52) keeps from the tight solution on the one-step tuples "
                        "(ls5 192 / window 1: 1.0e-5; rk4 640: 8.7e-6; tests/test_gpu_parity.py) -- the iso-accuracy throughput "
                        "beside the headline's, whose one-step error is 6.1e-5 (bar 1e-4)"}
    # final metric gather: the only collective on this path (RCCL all_gather of 19 doubles per rank: 16 + the `sustained` block's three)
    from gl_gym_amd.dist import gather_metrics, aggregate
    rows = gather_metrics([elapsed, float(B * K), m.get("sum_reward", 0.0), m.get("n_ode_fail", 0.0),
                           m.get("n_done", 0.0), kern_ms, m.get("n_guard_retries", 0.0), m.get("n_refined_substeps", 0.0),
                           float(rank), float(666 + rank), m.get("n_flag_err", 0.0), m.get("n_flag_branch", 0.0),
                           m.get("n_flag_cap", 0.0), m.get("n_flag_heavy", 0.0), parity[0], parity[1]]
                          + ([sus[0], sus[1], sus[2]] if sus is not None else []),
                          device=None if share_gpu else dev, force_collective=use_dist)
    if rank == 0:
        agg = aggregate(rows)
        t_max, value, kern_ms_max = agg["t_max"], agg["value"], agg["kernel_ms_max"]
        per_gpu_kernel_rate = B / (kern_ms_max * 1e-3)
        flops = STAGES[args.scheme] * args.n_sub * F_RHS
        specials = STAGES[args.scheme] * args.n_sub * S_RHS
        alg_tflops = per_gpu_kernel_rate * flops / 1e12
        alg_tops = per_gpu_kernel_rate * specials / 1e12
        algorithmic_ratio = alg_tflops / PEAKS_TFLOPS["non_fma"] + alg_tops / PEAKS_TFLOPS["transcendental_Tops"]
        obs_bytes = 0 if args.no_obs else 4 * env.obs_dim
        hbm_gbps = per_gpu_kernel_rate * BYTES_PER_ENV_STEP * (2 if args.dtype == "f64" else 1) / 1e9
        # executed work: the recorded PMC instruction counts of the default workload, scaled to this run's batch / n_sub,
        # over the kernel time measured live with HIP events on the launch stream
        variant = f"{args.dtype}_{args.scheme}" + ("_config5" if args.uncertainty else "")
        # batches up to 16 384 -- and every fp64 batch -- run the four-lanes-per-environment kernel (glgym.hip launch_step;
        # GLGYM_LAYOUT overrides): another kernel, other counters
        layout = os.environ.get("GLGYM_LAYOUT", "")
        quad = args.dtype == "f64" or (not args.uncertainty and (layout == "quad" or (layout == "" and B <= 16384)))
        if quad:
            variant += "_quad"
        pmc = load_pmc(variant + "_b65536") if (quad and B > 16384) else None      # recorded at this size (four rounds of waves)
        # batches of two or more wavefronts per SIMD take the two-waves-per-SIMD build of the one-lane kernel (glgym.hip launch_step;
        # GLGYM_OCC forces either build): another kernel, other counters (recorded at B = 262 144)
        occ_env = os.environ.get("GLGYM_OCC", "")
        occ2 = (not quad) and args.dtype == "f32" and not args.uncertainty and (occ_env == "2" or (occ_env != "1" and B >= 131072))
        pmc_key = variant + "_b65536" if pmc is not None else variant
        if occ2 and load_pmc(variant + "_occ2") is not None:
            pmc, pmc_key = load_pmc(variant + "_occ2"), variant + "_occ2"
        pmc = pmc or load_pmc(variant)
        waves = (4 if quad else 1) * ((B + 63) // 64)
        sus_agg = agg.get("sustained")
        roof = {"bound": "valu", "kernel": "step_kernel_quad" if quad else "step_kernel", "achieved": None, "peak": PEAKS_TFLOPS["fma"],
                "unit": "TFLOP/s", "frac": None, "traffic": None,
                # live kernel time (HIP events on the launch stream, mean over the timed launches, slowest rank) and the sustained
                # figures: kept among the FIRST keys, where a record that keeps only the head of this object still has them
                "kernel_ms": kern_ms_max,
                "sustained_value": None if sus_agg is None else sus_agg["value"],
                "sustained_kernel_ms": None if sus_agg is None else sus_agg["kernel_ms_max"],
                "waves_per_launch": waves, "waves_per_simd": waves / N_SIMD}
        if occ2:
            roof["kernel"] = "step_kernel, two-waves-per-SIMD build (256 registers, window state in LDS: glgym.hip launch_step)"
            if pmc_key != variant + "_occ2":       # no counters recorded for this scheme's two-wave build: no fraction from another kernel's
                pmc = None
                roof["pmc_variant"] = "none recorded for the two-waves-per-SIMD build of %s (frac = null rather than the one-wave build's counters)" % variant
        if pmc is not None:
            pmc_batch = float(pmc.get("batch", 65536))
            scale = (B / pmc_batch) * (args.n_sub / float(pmc.get("n_sub", N_SUB[args.scheme])))
            # the clock recorded under the profiler (GRBM_GUI_ACTIVE / kernel time of that pass); a value above the part's
            # 2.4 GHz is a counter artefact (seen on the 18 ms fp64 launches) -- the guide's clock then
            clk = pmc.get("clock_ghz") or 0.0
            clk_note = "the clock recorded under the profiler (%.3f GHz)" % clk
            if not (0.5 <= clk <= 2.45):
                clk_note = "the guide's 2.4 GHz (the profiler pass recorded an implausible %.2f GHz)" % clk
                clk = CLOCK_GUIDE_HZ / 1e9
            tkey = "SQ_INSTS_VALU_TRANS_F32" if args.dtype == "f32" else "SQ_INSTS_VALU_TRANS_F64"
            valu, trans = pmc["SQ_INSTS_VALU"] * scale, pmc.get(tkey, 0.0) * scale
            sfx = "_F32" if args.dtype == "f32" else "_F64"
            fma, mul, add = (pmc.get(k + sfx, 0.0) * scale for k in ("SQ_INSTS_VALU_FMA", "SQ_INSTS_VALU_MUL", "SQ_INSTS_VALU_ADD"))
            # fp64 vector ops issue at half the fp32 rate on this part (2x the cycles); the software transcendentals of the fp64
            # kernels are ordinary FMA chains and are counted as such
            wide = 2.0 if args.dtype == "f64" else 1.0
            t_s = kern_ms_max * 1e-3
            # ONE clock source per ruler, never the kernel time that is also the denominator: the guide's 2.4 GHz,
            # and the clock recorded under the profiler (GRBM_GUI_ACTIVE / kernel time of that pass).  Both fractions are
            # proportional to 1 / live kernel time.
            need_guide = wide * ((valu - trans) * CYC_PLAIN_GUIDE + trans * CYC_TRANS_GUIDE)
            need_meas = wide * ((valu - trans) * CYC_PLAIN + trans * CYC_TRANS)
            # a v_pk_*_f32 op is two lane-operations: it holds the SIMD for 4 cycles (same flop rate as two plain ops), but the PMC
            # counts it once; with the packed share of the shipped ISA's sub-step loop the slots the instruction stream really needs
            pk = valu * float(pmc.get("pk_share_static") or 0.0)
            need_pk = wide * ((valu - trans - pk) * CYC_PLAIN_GUIDE + pk * 2 * CYC_PLAIN_GUIDE + trans * CYC_TRANS_GUIDE)
            roof.update({
                # executed flops (64 lanes; FMA = 2; packed ops are counted once by the PMC, so this is a lower bound)
                "achieved": 64 * (2 * fma + mul + add) / t_s / 1e12,
                "frac": need_guide / (N_SIMD * t_s * CLOCK_GUIDE_HZ),
                "frac_note": "executed issue slots / available on the GUIDE's ruler: ((INSTS_VALU - TRANS) x 2 + TRANS x 8 cycles"
                             + (", x 2 for fp64" if wide > 1 else "") + ") / (1024 SIMDs x live kernel time x 2.4 GHz), instruction "
                             "counts from the recorded PMC passes of this variant; <= 1 by construction",
                "frac_packed_weighted": need_pk / (N_SIMD * t_s * CLOCK_GUIDE_HZ),
                "frac_packed_weighted_note": "the guide's ruler with packed fp32 ops at their real cost of 4 cycles (%.0f %% of this "
                                             "variant's vector instructions, tools/pk_share.py on the shipped ISA): the share of the "
                                             "issue capacity this instruction stream could occupy at ANY occupancy; `frac` counts a "
                                             "packed op as one 2-cycle slot and is the conservative figure"
                                             % (100 * float(pmc.get("pk_share_static") or 0.0)),
                "frac_measured_ruler": need_meas / (N_SIMD * t_s * clk * 1e9),
                "frac_measured_ruler_note": "same with the issue costs measured on this chip with >= 2 co-resident waves (2.3 / 7.7 "
                                            "cycles, profiles/r02_microbench_issue_rates.txt) and " + clk_note,
                "valu_busy_per_wave": pmc.get("valu_busy"),
                "valu_busy_note": "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES of the recorded profile of this variant (waves per SIMD "
                                  "above).  With ONE wave per SIMD -- all a one-lane-per-environment kernel can have at B = 65 536 -- "
                                  "a lone wave is issued a vector instruction only every ~5 cycles (transcendental 8.4), so this -- "
                                  "not `frac` -- is how close the kernel is to the ceiling its launch geometry allows",
                "traffic": pmc["traffic_bytes"] * (B / pmc_batch),
                "valu_insts_per_launch": valu, "trans_insts_per_launch": trans,
                "pmc_source": pmc.get("source"), "pmc_variant": pmc_key,
            })
        roof.update({
            "peaks_TFLOPs": PEAKS_TFLOPS,
            "algorithmic_ratio": algorithmic_ratio,
            "algorithmic_note": "SURVEY 8d figure: reference expression graph without CSE (stages x n_sub x (1502 flops + 219 "
                                "special ops) per env-step) / kernel time, against the non-FMA peak (78.65) and the transcendental rate (19.7); > 1 "
                                "because the kernel executes far less than that graph (hoisting, CSE, slow sub-expressions "
                                "once per window)",
            "algorithmic_TFLOPs": alg_tflops, "algorithmic_special_Tops": alg_tops,
            "kernel_env_steps_per_s": per_gpu_kernel_rate,
            "note": "path is VALU / transcendental bound, not HBM or MFMA (SURVEY 8d)",
            "traffic_note": "HBM bytes per launch: rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE",
            "hbm": {"achieved": hbm_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": hbm_gbps / PEAK_HBM_GBPS,
                    "algorithmic_bytes_per_env_step": BYTES_PER_ENV_STEP, "obs_bytes_per_env_step": obs_bytes}})
        out = {
            "collective": None if not use_dist else {"backend": dist.get_backend(), "world": world,
                                                     "ranks_gathered": len(rows), "note": "one all_gather of 19 doubles per rank at the end of the run"},
            "metric": "TomatoEnv env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": 1e3 * t_max / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload_label(args, B, world) + ": TomatoEnv batch %d/GPU, %s %s sub-stepped, synthetic "
                                   "weather year (KNMI Amsterdam files absent), random actions U(-1,1)"
                                   % (B, "fp32" if args.dtype == "f32" else "fp64", args.scheme.upper()),
                       "batch_per_gpu": B, "global_batch": B * world, "integrator": args.scheme, "n_sub": args.n_sub,
                       "window": args.window if args.window else "scheme default",
                       "dt_s": 900,
                       "obs_kernel": not args.no_obs, "obs_dim": env.obs_dim, "auto_reset": True, "hip_graph": bool(args.graph),
                       "vecnormalize": bool(args.vecnorm),
                       "uncertainty_scale": args.uncertainty, "parallelism": f"env-shard x{world} (no data-path collective)",
                       "deviation": "config text says 4 RK4 sub-steps; that is unstable for this stiff ODE (classical floor 224); n_sub is "
                                    "the nominal count of the stability-controlled sub-stepper, and the default scheme is a five-stage "
                                    "fourth-order Runge-Kutta method rather than the classical four-stage one (same order, same accuracy "
                                    "on every fixture, 640 instead of 960 right-hand sides; classical RK4 timed in other_scheme) (DESIGN.md 2)",
                       "scheme": ("five-stage FOURTH-order explicit Runge-Kutta scheme in 2N-storage form (stability interval 5.009 = 1.00 per "
                                  "right-hand side; classical RK4: 0.70)" if args.scheme == "ls5" else args.scheme.upper()) +
                                 " with the cover pair's conduction integrated exactly, "
                                 "stability-controlled per environment (rate bound at every window -> that window's own length, "
                                 "more, smaller sub-steps per window where needed; embedded error estimate as safety net), "
                                 "Strang-split exact harvest flow, slow sub-expressions once per window at the predicted midpoint "
                                 "(DESIGN.md 2)"},
            "parity": None if agg["max_scaled_err"] is None else {
                "max_scaled_err_10day": agg["max_scaled_err"], "bar": 1e-4, "failed": agg["parity_failed"],
                "fixture": "tests/golden/rollout_10day.npz: 961 env-steps (10 days, Bleiswijk weather, delta-u-bounded random "
                           "actions), truth = Radau rtol = atol = 1e-11 of the reference-text right-hand side",
                "holdout": None if holdout is None else {
                    "max_scaled_err": holdout[0], "failed": holdout[1], "max_temperature_err_K": holdout[2], "max_scaled_err_non_temperature": holdout[3],
                    "fixture": "tests/golden/holdout_gl2010_random.npz: 961 env-steps of the reference's second weather file (GL2010 from day 20: "
                               "frost), generated after the sub-stepper's constants were frozen; same truth.  Next to 0 C the metric divides a Celsius "
                               "temperature by itself: the kelvin figure and the non-temperature figure say the same without that (DESIGN.md 2.8)"},
                "kernel_build": ("one lane per environment, two-waves-per-SIMD build (as timed)" if occ_timed else
                                 ("one lane per environment, one-wave build (as timed)" if lay == "one" else "four lanes per environment (as timed)")),
                "note": "the metric's second half (max |delta state| vs the reference solution), run after the timed region with the "
                        "same dtype / scheme / n_sub / kernel layout AND register build on every rank; worst rank reported.  CVODES itself is not "
                        "available here or on the GPU box: the truth is a tight stiff solve, the reference-tolerance band "
                        "(BDF 1e-6) sits 4e-6 ... 1.3e-5 from it (DESIGN.md section 3)"},
            "roofline": roof,
            "integrator_events": {"failed_integrations": agg["ode_failures"], "guard_retries": agg["guard_retries"],
                                  "refined_substeps": agg["refined_substeps"], "first_attempt_flags": agg["first_attempt_flags"],
                                  "note": "failed integrations = env-steps reported like a failed CVODES call (done = 1, "
                                          "state unchanged: no two attempts of the n_sub, 2x, 4x, 8x ladder agreed); guard retries "
                                          "= extra attempts of that ladder; refined sub-steps = sub-steps beyond n_sub inserted by "
                                          "the stability control; first_attempt_flags = why first attempts were not accepted as "
                                          "they stood (error estimate / branch invariant / cap or non-finite / >= 3x the nominal "
                                          "sub-steps).  The action path runs guarded but unverified (delta_u_max = 0.1; "
                                          "include/glgym.h glgym_verify)"},
            "ranks": agg["ranks"],
            "other_scheme": None if alt is None else {
                "integrator": alt[0], "n_sub": alt[1], "value": B * world * K / alt[2], "unit": "env-steps/s",
                "ms_per_step": 1e3 * alt[2] / K,
                "note": "informational: same workload and timing protocol with the other fourth-order scheme the library offers "
                        "(classical RK4 with the cover conduction exact at n_sub 240 -- `value` of rounds 1-4 -- when the main leg runs ls5; "
                        "rank-0 clock); accuracy of all schemes vs the tight fixtures in DESIGN.md section 2"},
            "parity_config": pcfg,
            "sum_reward": agg["sum_reward"], "ode_failures": agg["ode_failures"],
            "episodes_finished": agg["episodes_finished"],
        }
        if world == 1 and not args.no_cpu_baseline:
            out.update(cpu_baseline(args.n_sub, order={"rk4": 4, "ls5": 5, "rk3": 3, "rk2": 2}[args.scheme],
                                    window=args.window or {"rk4": 4, "ls5": 2, "rk3": 3, "rk2": 4}[args.scheme]))
        # LAST key of the line (a record that keeps only the tail of stdout still has it)
        out["sustained"] = None if sus_agg is None else {
            "value": sus_agg["value"], "unit": "env-steps/s", "steps": sus[3], "timed_s": sus_agg["t_max"],
            "ms_per_step": 1e3 * sus_agg["t_max"] / sus[3], "kernel_ms": sus_agg["kernel_ms_max"], "continued": sus[4],
            "note": ("the same loop continued after the K timed steps (same env, action stream and bracketing) until >= 1 s was timed: "
                     "whole-job env-steps of the continuation / slowest rank's time; `steps` is rank 0's count" if sus[4] else
                     "the K timed steps themselves (they lasted >= 1 s, or the continuation was switched off / --graph)")}
        print(json.dumps(out), flush=True)
    env.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
