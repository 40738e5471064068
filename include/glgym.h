/*
 * glgym.h -- C ABI of libglgym.so: the MI355X (gfx950) implementation of GreenLight-Gym's
 * per-timestep ODE integration, i.e. the path TomatoEnv.step() drives.
 *
 * Plain C, pointers and sizes only (no torch / pybind types).  One handle = one HIP device + one
 * parameter block + one dtype.  The library never allocates per call on the hot path and never
 * synchronises the stream in glgym_step / glgym_obs / glgym_reset (safe to capture in a hipGraph);
 * the host-pointer convenience entry points (glgym_evalF, glgym_rhs) do synchronise.
 * Every function returns a glgym_status; nothing throws.  There is NO CPU fallback: without a HIP
 * device glgym_create() returns GLGYM_ENODEV.
 *
 * Reference interface each entry point replaces (paths relative to the GreenLight-Gym2 repo):
 *   glgym_create   <- GreenLight::GreenLight(nx,nu,nd,np,dt)   gl_gym/environments/models/greenlight_model.cpp:31-94
 *                     + env.p = init_default_params(np)         gl_gym/environments/tomato_env.py:62
 *   glgym_evalF    <- GreenLight::evalF(x,u,d,p) -> x_next      gl_gym/environments/models/greenlight_model.cpp:96-120
 *                     (pybind binding                            gl_gym/environments/models/greenlight_model.cpp:130-136)
 *   glgym_step     <- TomatoEnv.step / step_raw_control         gl_gym/environments/tomato_env.py:115-173
 *                     (action_to_control :109-113, evalF call :120, terminal test :131-132,
 *                      reward rewards.py:218-231, info tomato_env.py:208-222), batched over B envs
 *   glgym_obs      <- TomatoEnv._get_obs + 6 observation modules gl_gym/environments/observations.py:59-182
 *   glgym_reset    <- TomatoEnv.reset (state part)               gl_gym/environments/tomato_env.py:262-266,
 *                     init_state                                  gl_gym/environments/utils.py:13-46
 *   glgym_crop_noise <- parametric_crop_uncertainty               gl_gym/environments/noise.py:3-23
 *   glgym_rule_based <- RuleBasedController.predict              gl_gym/environments/baseline.py:68-227
 *                     (constants gl_gym/configs/agents/rule_based.yml; caller experiments/evaluate_baseline.py:22)
 *   glgym_weather  <- load_weather_data (array part)              gl_gym/environments/utils.py:48-125
 *   glgym_rhs      <- ODE(x,u,d,p) (test hook; no reference binding) gl_gym/environments/models/ode.hpp:6-124
 *
 * Layouts.  "SoA [n][ld]" = n planes of ld elements, element (i, b) at base[i*ld + b]; lane b of a
 * wavefront touches consecutive addresses.  Element type T is float (GLGYM_F32) or double (GLGYM_F64).
 */
#ifndef GLGYM_H
#define GLGYM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLGYM_NX 28
#define GLGYM_NU 6
#define GLGYM_ND 10        /* columns the model reads; glgym_create accepts nd = 10..16 = row stride of weather / d */
#define GLGYM_NP 208
#define GLGYM_NCROP 34      /* p[128..161], the block noise.py perturbs */
#define GLGYM_NINFO 11      /* EPI, revenue, variable_costs, fixed_costs, co2_cost, heat_cost, elec_cost,
                               temp_violation, co2_violation, rh_violation, lamp_violation (tomato_env.py:208-222) */
#define GLGYM_NMETRIC 14    /* sum reward, sum EPI, n done, n failed integrations, sum co2/temp/rh violation, n env-steps,
                               n guard retries (extra attempts of the n_sub, 2x, 4x, 8x ladder: unverified or -- verified
                               mode -- every env-step's second attempt), n refined sub-steps (sub-steps beyond n_sub that the
                               stability control inserted: storms, wet screens pinned to the air temperature), then why
                               first attempts were unverified: n error-estimate flags, n branch-invariant flags, n cap /
                               non-finite flags, n heavy (>= 3x the nominal sub-steps) */
#define GLGYM_METRIC_REPLICAS 64   /* accumulator blocks, one 128-byte line each (atomics onto a single line serialise) */
#define GLGYM_METRIC_STRIDE 32     /* floats per replica */

typedef struct glgym_handle_s* glgym_handle;

typedef enum { GLGYM_F32 = 0, GLGYM_F64 = 1 } glgym_dtype;

/* Right-hand side variants of gl_gym/environments/models/ode.hpp.  GLGYM_ODE = ODE (:6-124), what the reference's
 * compiled module integrates.  GLGYM_ODE_PIPE = ODE_pipe (:126-263): columns 10..13 of each weather / d row are
 * (tPipe, tGroPipe, pipeSwitchOff, groPipeSwitchOff); dxdt(9) = d10 - x9 unless d10 < 1 or d12 > 0, dxdt(19) = 0.
 * Its tracking term has rate 1 1/s, which the stability control accounts for (smaller sub-steps while tracking when
 * dt / n_sub > 2.5 s; none needed at dt = 300, n_sub = 256, experiments/gl_predefined_controls.py's setting). */
typedef enum { GLGYM_ODE = 0, GLGYM_ODE_PIPE = 1 } glgym_variant;

/* Sub-stepping scheme of glgym_step / glgym_evalF (greenlight_model.cpp:46-63 uses CVODES BDF, error-controlled and
 * implicit; any scheme that meets the accuracy bar against it is admissible).  n_sub is the NOMINAL (= minimum) number
 * of sub-steps per env-step.  Both schemes are stability-controlled per environment: a bound on the fastest local
 * relaxation rate (top-compartment exchange 0.2-0.7 1/s, up to 1.1 1/s in storms; a wet screen pinned to
 * the air temperature 3 ... 15 1/s) is evaluated at the start of every window of 1-4 nominal sub-steps; a window whose bound asks
 * for shorter sub-steps is itself shortened (round 5: 6 % over the limit costs 6 % more stages; a pinned surface's burst is looked at
 * again after 1-2 s), and beyond that the environment takes as many smaller sub-steps in the window as its scheme's stability
 * interval asks for; an embedded error estimate is the safety net.  An attempt that is flagged (error estimate, non-finite, rate beyond 64x the nominal count for more than 120 s,
 * a wet surface that changed sides inside its bistable regime in a capped window) or that took 3x the nominal number of sub-steps is UNVERIFIED:
 * the env-step is redone with 2x, 4x, 8x n_sub until an attempt is clean or two consecutive attempts agree on the fast states
 * (step doubling; counted in GLGYM_NMETRIC).  An environment for which no two attempts agree is reported like a failed CVODES
 * call in the reference (tomato_env.py:119-123): done = 1, state unchanged (glgym_step) / GLGYM_EODE (glgym_evalF).
 *   GLGYM_SCHEME_RK4: RK4 (stability interval 2.785) with the one fast LINEAR mode of the model taken out of it: the conduction
 *     between the two faces of the cover glass (hCovInCovE, aux_states.hpp:918 / ode.hpp:37-42; 0.65 1/s, state-independent) is
 *     integrated exactly -- Cox-Matthews' exponential RK4 on w = tCovIn - tCovE, classical RK4 on every other state.  The
 *     nominal sub-step is then set by the top compartment's air exchange: use n_sub 240 (nominal environments cover rates up to
 *     0.68 1/s).  The slow sub-expressions and the
 *     harvest flow are evaluated once per window of four nominal sub-steps in both precisions (n_sub is rounded up to a multiple of 4).
 *     That window is what sets the accuracy at a given n_sub (max scaled error on the tight one-step tuples, fp64: 6.2e-5 at 240,
 *     1.5e-5 at 480, 8.7e-6 at 640, 6.9e-6 at 720; 10-day rollout 1.5e-5 at 240): for PARITY runs against the reference's solver set
 *     n_sub 640, which sits inside the 1.3e-5 band of a BDF solve at the reference's tolerances (tests/test_gpu_parity.py).
 *   GLGYM_SCHEME_RK2: the midpoint rule of the same family (ETD2RK on the cover conduction, explicit midpoint elsewhere; stability
 *     interval 2.0): use n_sub 336.  30 % fewer right-hand sides than RK4; the slow sub-expressions and the harvest flow are shared
 *     by four nominal sub-steps (n_sub is rounded up to a multiple of 4).  Second order: the least accurate of the three (tight
 *     one-step tuples 3.2e-5, storm fixture 2.9e-5 in fp64).
 *   GLGYM_SCHEME_RK3: the three-stage, third-order member of the same exponential family (Cox-Matthews ETD3RK on the cover
 *     conduction, Kutta's RK3 -- stages at 0, h/2, h -- on every other state; stability interval 2.513): use n_sub 270.  19 % fewer
 *     right-hand sides than RK4 at RK4-like accuracy (the error of both is set by the slow tier's window, not by the order); the
 *     slow sub-expressions and the harvest flow are shared by three nominal sub-steps (n_sub is rounded up to a multiple of 3).
 *     (Rounds 2-3 shipped Bogacki-Shampine 3(2) with the conduction in its right-hand side at n_sub 354 under this name.)
 *   GLGYM_SCHEME_LS5 (round 5; the Python layer's default): a FIVE-stage FOURTH-order explicit Runge-Kutta scheme in Williamson's 2N-storage
 *     form (two registers per state: dy <- A_i dy + h f(y), y <- y + B_i dy).  The 2N five-stage fourth-order family has one free
 *     parameter, the z^5 coefficient of its stability polynomial; Carpenter-Kennedy's published member (1/200) has a real-axis stability
 *     interval of 4.657, the member used here (0.0047) 5.009 with |R| <= 0.28 at its working point: 1.00 per right-hand side against
 *     RK4's 0.70.  (Members with a longer interval settle on spurious quasi-steady states of the strongly ventilated top compartment
 *     beyond h lambda ~ 4.1; this one shows none up to its limit.)  The cover conduction is integrated exactly here too (Lawson's
 *     transformation applied to the deviation of the forcing from a linear predictor: exact for a linearly varying forcing, no stage
 *     history).  Use n_sub 128 (7.03 s sub-steps: rates up to 0.655 1/s); the slow sub-expressions and the harvest flow are shared by TWO
 *     nominal sub-steps (14 s; RK4-240: 15 s): 640 right-hand sides + 64 windows per env-step where RK4 takes 960 + 60, at RK4's accuracy
 *     or better on every fixture (tight one-step tuples 5.4e-5, 10-day rollout 1.3e-5 in fp64; RK4 at 240: 6.1e-5 / 1.5e-5) and on the wide
 *     random stress (43 of 5 954 one-step maps above 1e-4 against RK4-240's 49).  Its stability control differs from the other schemes' in
 *     two measured points: the movement limiter's allowance grows with the head-room the window's rate bound leaves below the stability
 *     limit, and a window the limiter does bind re-partitions its remainder sub-step by sub-step (the initial layer of an env-step decays
 *     within seconds).  PARITY preset: n_sub 192 with glgym_set_window(h, 1): 9.1e-6 on the tight one-step tuples, inside the 1.3e-5 band
 *     of a BDF solve at the reference's tolerances, at 960 right-hand sides + 192 windows (RK4 needs n_sub 640: 2 560 + 160).
 *     oracle/studies/lsrk_study.py, stress_ls5.py; DESIGN.md section 2.7. */
typedef enum { GLGYM_SCHEME_RK4 = 0, GLGYM_SCHEME_RK2 = 1, GLGYM_SCHEME_RK3 = 2, GLGYM_SCHEME_LS5 = 3 } glgym_scheme;

/* Step-doubling VERIFIED integration: no attempt is accepted on its own -- the result is the finer of two agreeing attempts (at
 * least n_sub and 2 n_sub: 3x the work, 4e-7 median error instead of 2e-6) -- with ONE exception, in this mode and the unverified one
 * alike: when no two attempts of the ladder agree, the finest one (8x n_sub) is taken as it stands if nothing flagged it (env-steps
 * that start on a kink or pass a bifurcation are sensitive at the 1e-4 level for any solver).  glgym_step reports that per env-step
 * (step_flags: GLGYM_SF_ACCEPT_LAST_ALONE), as it reports a result accepted by agreement on attempts that carried a flag.  The reference's solver is error-controlled
 * (greenlight_model.cpp:46-63) and its raw-control entry points apply any u with no delta-u clip (tomato_env.py:148-173; the
 * rule-based controller bang-bangs, baseline.py:68-227): after an all-actuator jump an explicit scheme near its stability limit
 * can put a wet cover on the wrong branch with every in-step check green.
 *   GLGYM_VERIFY_AUTO (default): verified wherever the control can jump -- glgym_evalF, glgym_step(control = ...), and
 *     glgym_step(action = ...) when delta_u_max > 0.1; the reference's action path (delta_u_max = 0.1) runs unverified + guard.
 *   GLGYM_VERIFY_ALWAYS / GLGYM_VERIFY_NEVER: every / no entry point. */
typedef enum { GLGYM_VERIFY_AUTO = 0, GLGYM_VERIFY_ALWAYS = 1, GLGYM_VERIFY_NEVER = 2 } glgym_verify;

typedef enum {
    GLGYM_OK = 0,
    GLGYM_EINVAL = -1,   /* bad argument (sizes other than 28/6/10/208, null pointer, n_sub < 1 ...) */
    GLGYM_ENODEV = -2,   /* no usable HIP device */
    GLGYM_EHIP = -3,     /* a HIP runtime call failed; see glgym_last_error() */
    GLGYM_ENOMEM = -4,
    GLGYM_EODE = -5      /* glgym_evalF: the integration failed for at least one row (the reference's evalF raises on a
                            CVODES failure, greenlight_model.cpp:110); failed rows of x_next are NaN, the others valid */
} glgym_status;

/* Reward constants (gl_gym/configs/envs/TomatoEnv.yml:38-67; rewards.py:47-124). */
typedef struct {
    double elec_price, heating_price, co2_price, fruit_price, dmfm;
    double fixed_greenhouse_cost, fixed_co2_cost, fixed_lamp_cost, fixed_screen_cost;
    double pen_lamp;
    double co2_min, co2_max, temp_min, temp_max, rh_min, rh_max;
} glgym_reward_cfg;

/* ABI version of this header; glgym_abi_version() returns the library's.  5: glgym_step_args starts with struct_size (round 5) */
#define GLGYM_ABI_VERSION 6

/* Device-pointer arguments of one batched env-step.  Exactly one of `action` / `control` is non-null. */
typedef struct {
    int32_t struct_size;       /* sizeof(glgym_step_args) as the CALLER compiled it: glgym_step refuses (GLGYM_EINVAL) a struct of another
                                  size instead of reading pointers that are not there (the struct grew by step_flags in round 4) */
    int32_t B;                 /* environments in this shard */
    int32_t ld;                /* leading dimension of every SoA array (>= B) */
    void* x;                   /* SoA [28][ld] T, in/out: state */
    void* u;                   /* SoA [6][ld]  T, in/out: previous control in, applied control out */
    const float* action;       /* [B][6] row-major f32 in [-1,1]: u <- clip(u + 0.1f*a, 0, 1)   (step)            */
    const void* control;       /* SoA [6][ld] T: u <- control                                    (step_raw_control) */
    const void* weather;       /* [weather_rows][nd] row-major T (nd of glgym_create), shared by all envs */
    int32_t weather_rows;
    const int32_t* w_off;      /* [B] first weather row of each env's episode                    */
    int32_t* timestep;         /* [B] in/out; row integrated over = w_off[b] + timestep[b]; then ++ */
    const void* crop_p;        /* SoA [34][ld] T per-env p[128..161] (config 5) or NULL = shared  */
    int32_t N;                 /* terminal test `timestep >= N` before the increment (episode = N+1 steps) */
    void* reward;              /* [ld] T out */
    void* info;                /* SoA [11][ld] T out, order of GLGYM_NINFO */
    uint8_t* done;             /* [B] out: terminated (season end, or ODE failure -> state left unchanged) */
    float* metrics;            /* [GLGYM_METRIC_REPLICAS][GLGYM_METRIC_STRIDE] f32 accumulators or NULL: wavefront w adds
                                  its sums (GLGYM_NMETRIC order) to replica w % GLGYM_METRIC_REPLICAS; the reader sums the
                                  replicas */
    int32_t* step_flags;       /* [B] out or NULL: how this env-step's integration went (the reference has no counterpart:
                                  CVODES either returns or throws, greenlight_model.cpp:110) -- GLGYM_SF_* bits below */
} glgym_step_args;

/* step_flags[b]: bits 0..4 = why the FIRST attempt was not accepted as it stood (0 = it was): 1 rate bound beyond 64x the nominal
 * sub-step count for more than 120 s, 2 non-finite, 4 error estimate above tolerance, 8 a wet surface changed sides inside its
 * bistable regime in a capped window, 16 it took >= 3x the nominal number of sub-steps.  Bits 8..10 = extra attempts used
 * (2x, 4x, 8x n_sub).  Bits 16..30 = sub-steps taken beyond the nominal n_sub, all attempts together (saturating at 32 767; bit 31 is
 * never set, so the word is non-negative as an int32).  How the returned state was accepted: */
#define GLGYM_SF_ACCEPT_AGREE_FLAGGED 32   /* two consecutive attempts agreed, but the accepted (finer) one carried a flag itself */
#define GLGYM_SF_ACCEPT_LAST_ALONE 64      /* the finest attempt (8x n_sub), unflagged, taken as it stood although it did not agree
                                              with the attempt before it -- in verified mode as well */
#define GLGYM_SF_FAILED 128                /* failed integration: done = 1, state unchanged */

/* Device-pointer arguments of observation assembly (row-major output, what SB3 / Gymnasium consume). */
typedef struct {
    int32_t B, ld;
    const void* x;             /* SoA [28][ld] T */
    const void* u;             /* SoA [6][ld]  T */
    const void* weather;       /* [weather_rows][10] T */
    int32_t weather_rows;
    const int32_t* w_off;      /* [B] */
    const int32_t* timestep;   /* [B] value AFTER the step's increment; 0 = freshly reset env */
    const float* start_day;    /* [B] day of year at reset */
    int32_t Np;                /* forecast rows (int(pred_horizon*86400/dt)); obs_dim = 23 + 5*Np with the default modules */
    float* obs;                /* [B][glgym_obs_dim(h, Np)] row-major f32 out */
    const uint8_t* mask;       /* NULL: every row.  Else only rows with mask[b] != 0 are recomputed ...            */
    float* term_obs;           /* ... after their CURRENT content was saved here ([B][dim], SB3 "terminal_observation");
                                  may be NULL */
} glgym_obs_args;

/* Device-pointer arguments of a masked reset. */
typedef struct {
    int32_t B, ld;
    const uint8_t* mask;       /* [B] reset env b iff mask[b] != 0; NULL = all */
    void* x;                   /* SoA [28][ld] T out: init_state(weather[w_off[b]]) */
    void* u;                   /* SoA [6][ld]  T out: zeros */
    int32_t* timestep;         /* [B] out: 0 */
    const void* weather;
    int32_t weather_rows;
    int32_t* w_off;            /* [B] in/out: first weather row of the episode.  With a start table (below) the kernel
                                  draws the new episode's start itself, otherwise it uses the value found here */
    /* optional start table (TomatoEnv.reset picks (growth_year, start_day) at random, tomato_env.py:236-244):
       entry j = Philox4x32-10(key = seed, counter = (env, episode[b])) mod n_starts; episode[b] is then incremented */
    const int32_t* start_rows; /* [n_starts] or NULL */
    const float* start_days;   /* [n_starts] day of year of each start row */
    int32_t n_starts;
    float* start_day;          /* [B] out: day of year of the drawn start */
    int32_t* episode;          /* [B] in/out: episodes started so far per env */
    uint64_t seed;
} glgym_reset_args;

const char* glgym_version(void);
int glgym_abi_version(void);                                 /* GLGYM_ABI_VERSION the library was built with */
const char* glgym_last_error(void);

/* p: host, np doubles (the reference promotes its float32 parameter block to double before evalF). */
int glgym_create(int nx, int nu, int nd, int np, double dt, const double* p, int dtype, int n_sub, int device,
                 glgym_handle* out);
int glgym_destroy(glgym_handle h);
int glgym_set_params(glgym_handle h, const double* p);
/* The reference's `env.p = new_p` on an env that already exists (experiments/run_time.py:40-41, gl_predefined_controls.py:70-77): the model, the
 * crop block and the per-step cost coefficients of the reward follow new_p (rewards.py:164-166 read env.p at every step), but the reward's SCALE
 * -- max_profit / min_profit -- stays what it was at construction (rewards.py:82-83 computes them once, in __init__).  glgym_set_params
 * recomputes the scale too, i.e. it is "construct the env with new_p".  (ABI 6) */
int glgym_set_params_keep_reward_scale(glgym_handle h, const double* p);
int glgym_set_n_sub(glgym_handle h, int n_sub);
int glgym_set_scheme(glgym_handle h, int scheme);            /* GLGYM_SCHEME_RK4 (default) | GLGYM_SCHEME_RK2 | GLGYM_SCHEME_RK3 | GLGYM_SCHEME_LS5 */
/* Nominal sub-steps per tier-2b / harvest window.  0 (default) = the scheme's own (RK4 4, RK2 4, RK3 3, LS5 2); 1..8 overrides it at run
 * time (n_sub is rounded up to a multiple).  The window's length in seconds is what sets the accuracy at a given n_sub. */
int glgym_set_window(glgym_handle h, int window);
/* Kernel layout of glgym_step for GLGYM_F32 handles (GLGYM_F64 has one layout).  Initial value: environment variable GLGYM_LAYOUT =
 * one | quad read ONCE at glgym_create (A/B tools), else AUTO. */
typedef enum { GLGYM_LAYOUT_AUTO = 0, GLGYM_LAYOUT_ONE = 1, GLGYM_LAYOUT_QUAD = 2 } glgym_layout;
int glgym_set_layout(glgym_handle h, int layout);
/* Waves per SIMD the one-lane fp32 step kernel is built for: 1 (up to 512 registers, no scratch), 2 (256 registers, the windows' state
 * in LDS: two wavefronts share a SIMD) or 0 (default): 2 for batches of at least two wavefronts per SIMD (131 072 environments on
 * MI355X: 1.06-1.11x), 1 below.  Results of the two builds agree to rounding (tests/test_gpu_parity.py).  Initial value: GLGYM_OCC =
 * 1 | 2 read once at glgym_create. */
int glgym_set_occupancy(glgym_handle h, int waves_per_simd);
/* Verified glgym_evalF calls (GLGYM_VERIFY_AUTO / _ALWAYS) on batches that leave lanes free -- rows * 8 lanes <= one wavefront per SIMD of the
 * device: 8 192 rows on MI355X -- run the step-doubling ladder two rungs at a time on two lane groups per row (n_sub and 2 n_sub side by
 * side, then 4 n_sub and 8 n_sub if needed): the accepted attempt and the returned state are those of the sequential ladder bit for bit,
 * the elapsed time is the 2 n_sub attempt's instead of n_sub + 2 n_sub (profiles/r05_evalf_latency.txt).  Round 6: verified glgym_step
 * launches (control = ..., i.e. step_raw_control and the rule-based controller; four-lanes-per-environment kernels, shared crop block) on
 * batches of up to 8 192 environments do the same on two lane groups per environment, step_flags included.  1 (default) = where it
 * applies, 0 = always the sequential ladder (what per-row / per-env parameter blocks and larger batches run anyway). */
int glgym_set_ladder_parallel(glgym_handle h, int on);
int glgym_set_verify(glgym_handle h, int mode);              /* GLGYM_VERIFY_AUTO (default) | _ALWAYS | _NEVER */
/* action_to_control (tomato_env.py:109-113): u = clip(u_prev + action * delta_u_max, u_min, u_max), held in float32 like
 * base_env.py:72-74.  Default: the yml's [0, 1] bounds and 0.1 (configs/envs/TomatoEnv.yml:12-14).  The `control` input
 * of glgym_step (step_raw_control) is applied unclipped, as the reference does (tomato_env.py:148-149). */
int glgym_set_control_limits(glgym_handle h, const double* u_min /*[6]*/, const double* u_max /*[6]*/, double delta_u_max);
int glgym_set_model_variant(glgym_handle h, int variant);   /* GLGYM_ODE_PIPE needs a handle created with nd >= 14 */
int glgym_set_reward(glgym_handle h, const glgym_reward_cfg* cfg);
int glgym_get_reward_scale(glgym_handle h, double* max_profit, double* min_profit, double* fixed_costs);

/* Host pointers, row-major, double.  p_rows = 1 (one block for all rows) or B.  Synchronous. */
int glgym_evalF(glgym_handle h, const double* x, const double* u, const double* d, const double* p, int p_rows,
                int B, double* x_next);
int glgym_rhs(glgym_handle h, const double* x, const double* u, const double* d, int B, double* dx);

/* Observation modules (gl_gym/environments/observations.py:59-182), concatenated per row in the order of
 * TomatoEnv's `observation_modules` list (tomato_env.py:77-81, 193-198).  Default: all six in configs/envs/TomatoEnv.yml
 * order.  glgym_set_obs_modules takes 1..6 distinct ids; glgym_obs_dim returns the resulting row width (< 0 on error).
 * The reference's seventh entry, StateObservations (:35-57), cannot be constructed by TomatoEnv (its __init__ takes no env)
 * and returns random numbers; it is not offered. */
enum { GLGYM_OBS_INDOOR = 0,    /* IndoorClimateObservations  co2_air [ppm], temp_air, rh_air, pipe_temp          (4) */
       GLGYM_OBS_CROP = 1,      /* BasicCropObservations      24CanTemp (x21), cFruit (x25), tSum (x26)            (3) */
       GLGYM_OBS_CONTROL = 2,   /* ControlObservations        uBoil, uCo2, uThScr, uVent, uLamp, uBlScr            (6) */
       GLGYM_OBS_WEATHER = 3,   /* WeatherObservations        glob_rad, temp_out, rh_out, co2_out [ppm], wind      (5) */
       GLGYM_OBS_TIME = 4,      /* TimeObservations           timestep, sin/cos day of year, sin/cos hour of day   (5) */
       GLGYM_OBS_FORECAST = 5   /* WeatherForecastObservations raw weather columns 0..4 of the next Np rows    (5 Np) */ };
int glgym_set_obs_modules(glgym_handle h, const int32_t* modules, int n);
int glgym_obs_dim(glgym_handle h, int Np);

/* Device pointers; asynchronous on `stream` (a hipStream_t, NULL = default stream).
 * glgym_step picks its kernel layout per launch.  GLGYM_F64: four lanes per environment (csrc/gl_model_quad.hpp) in every scheme,
 * ODE variant and batch size -- the only fp64 integrator on the device.  GLGYM_F32: four lanes per environment for B <= 16 384 with
 * shared crop parameters and the default ODE (every scheme), one lane per environment otherwise; glgym_set_layout overrides that choice
 * (fp32 only).  Same scheme decision for decision in both layouts; results equal to fp32 rounding accumulated over the sub-steps
 * (measured on the storm / jump fixtures: <= 2.5e-4 scaled between the two fp32 layouts, tests/test_gpu_storm.py). */
int glgym_step(glgym_handle h, const glgym_step_args* a, void* stream);
int glgym_obs(glgym_handle h, const glgym_obs_args* a, void* stream);
int glgym_reset(glgym_handle h, const glgym_reset_args* a, void* stream);
/* crop_p[i][b] = fl(p[128+i] * (1 + U(-scale/2, scale/2))), then p144 = p141/p142; Philox4x32-10 keyed by
 * (seed, stream_id), counter (env index, draw_index).  crop_p: SoA [34][ld] T. */
int glgym_crop_noise(glgym_handle h, void* crop_p, int B, int ld, double scale, uint64_t seed, uint64_t draw_index,
                     void* stream);

/* ---- rule-based controller (BASELINE config 1 "fixed rule-based actions") ------------------------------
 * u[6] = RuleBasedController.predict(x, weather[w_off + timestep], env clocks) for every env of the shard, written in the
 * SoA layout glgym_step consumes as `control` (step_raw_control).  Constants: the constructor arguments of
 * baseline.py:22-66 in declaration order.  Computed in fp64 for either handle dtype (12 exp per env-step: free). */
typedef struct {
    double lamps_on, lamps_off, lamps_day_start, lamps_day_stop, lamps_off_sun, lamp_rad_sum_limit;
    double temp_setpoint_day, temp_setpoint_night, heat_correction, heat_deadzone, co2_day;
    double vent_heat_Pband, rh_max, mech_dehumid_Pband, vent_rh_Pband, t_vent_off, vent_cold_Pband;
    double thScrSpDay, thScrSpNight, thScrPband, thScrDeadZone, thScrRh, thScrRhPband, lampExtraHeat;
    double blScrExtraRh, rhMax, tHeatBand, co2Band, useBlScr;
} glgym_rule_cfg;

typedef struct {
    int32_t B, ld;
    const void* x;             /* SoA [28][ld] T */
    const void* weather;       /* [weather_rows][10] T; row used = w_off[b] + timestep[b] (the row about to be integrated) */
    int32_t weather_rows;
    const int32_t* w_off;      /* [B] */
    const int32_t* timestep;   /* [B] */
    const float* start_day;    /* [B] day of year at reset: day_of_year = start_day + timestep*((dt/86400) mod 365),
                                  hour_of_day = (timestep*dt/3600) mod 24   (tomato_env.py:126-128) */
    const double* hour;        /* optional [B] explicit clocks; NULL = derive as above.  The reference keeps hour_of_day as a RUNNING SUM
                                  (hour += dt/3600; hour %= 24 per step): exact for dt = 900 s, but for increments that are not multiples of
                                  1/8 h (dt = 300 s, experiments/run_time.py) the sum reads 17.999999999999996 where the product reads 18 and
                                  the rules compare the clock with whole hours -- pass the running sum here to switch the lamps on the
                                  reference's step (gl_gym_amd.TomatoVecEnv does; tests/test_gpu_holdout.py rule_based_controller_kernel) */
    const double* doy;
    void* control;             /* SoA [6][ld] T out */
} glgym_rule_args;

int glgym_rule_based(glgym_handle h, const glgym_rule_cfg* cfg, const glgym_rule_args* a, void* stream);

/* ---- on-device VecNormalize ---------------------------------------------------------------------
 * Replaces stable_baselines3.common.vec_env.VecNormalize (3rd party, pinned ==2.6.0 in the reference's requirements.txt)
 * as the reference configures it: norm_obs, norm_reward, clip_obs = 10, gamma (gl_gym/RL/experiment_manager.py:142-147,
 * gl_gym/RL/utils.py:62-66).  Algorithm: batch mean/var -> RunningMeanStd.update_from_moments -> clip((x-mean)/sqrt(var+eps));
 * rewards: returns = returns*gamma + r; ret_rms.update(returns); clip(r/sqrt(ret_var+eps)); returns[done] = 0.
 * All statistics are caller-owned device doubles so they can be saved / restored like VecNormalize.save(). */
typedef struct {
    int32_t B, dim;
    const float* obs;          /* [B][dim] raw observations (glgym_obs output) */
    float* obs_out;            /* [B][dim] normalised + clipped (may alias obs) */
    const void* reward;        /* [B] T raw rewards, or NULL (reset: observations only) */
    float* reward_out;         /* [B] f32 normalised + clipped */
    const uint8_t* done;       /* [B] or NULL */
    double* obs_mean;          /* [dim] running mean   (init 0) */
    double* obs_var;           /* [dim] running var    (init 1) */
    double* obs_count;         /* [1]   running count  (init 1e-4) */
    double* ret_stats;         /* [3]   mean, var, count of the discounted returns (init 0, 1, 1e-4) */
    double* returns;           /* [B]   discounted return accumulators (init 0) */
    double* workspace;         /* [32*dim + 2] scratch, zeroed by the call */
    double gamma, epsilon;     /* 0.99, 1e-8 in SB3 */
    float clip_obs, clip_reward;
    int32_t training, norm_obs, norm_reward;
} glgym_vecnorm_args;

int glgym_vecnorm(glgym_handle h, const glgym_vecnorm_args* a, void* stream);

/* ---- weather pipeline on the device -----------------------------------------------------------------------
 * Replaces the array part of load_weather_data (gl_gym/environments/utils.py:48-125; called per reset at
 * tomato_env.py:249-259): raw samples (the CSV columns, already sliced to the season window) -> vpOut / co2Out / tSoOut
 * unit conversions (:281-444, :262-279), daily light sum (:216-250), daylight flags with their one-hour ramps (:177-214),
 * PCHIP resample (scipy PchipInterpolator, Fritsch-Carlson slopes) onto linspace(time[0], time[n_raw-1], n_out), and the
 * iGlob < 1e-10 -> 0 clean-up.  fp64 arithmetic, output in the handle's dtype, row-major [n_out][nd] -- the table
 * glgym_step / glgym_obs / glgym_reset read.  Parsing the CSV stays with the caller. */
typedef struct {
    int32_t n_raw;             /* raw samples (>= 3), uniformly spaced in time */
    const double* time;        /* [n_raw] device: seconds */
    const double* i_glob;      /* [n_raw] device: global radiation [W m-2]      ("global radiation") */
    const double* t_out;       /* [n_raw] device: outdoor temperature [C]       ("air temperature") */
    const double* rh;          /* [n_raw] device: relative humidity [%]         ("RH") */
    const double* wind;        /* [n_raw] device: wind speed [m s-1]            ("wind speed") */
    const double* t_sky;       /* [n_raw] device: sky temperature [C]           ("sky temperature") */
    double co2_ppm;            /* outdoor CO2 (the reference uses 400) */
    int32_t n_out;             /* rows of the resampled table: int(dt_raw / dt_env * n_raw) */
    int32_t nd;                /* row stride of out (>= 10; columns 10.. are zeroed) */
    void* out;                 /* [n_out][nd] T device */
    double* workspace;         /* [20 * n_raw] device scratch */
} glgym_weather_args;

int glgym_weather(glgym_handle h, const glgym_weather_args* a, void* stream);

/* Kernel timing on the stream the kernels are launched on (bench.py's roofline leg). */
int glgym_timer_start(glgym_handle h, void* stream);
int glgym_timer_stop(glgym_handle h, void* stream, float* elapsed_ms);   /* synchronises the stop event */

#ifdef __cplusplus
}
#endif
#endif /* GLGYM_H */
