// sc_policy.hpp -- the POLICY of the stability-controlled sub-stepper, stated once for both kernel layouts (round 6).
//
// rk_delta (gl_model.hpp: one lane per environment) and rk_delta_quad (gl_model_quad.hpp: four lanes per environment) differ in how
// they hold the state and gather a maximum over it; what they DECIDE -- how long the next window is, how many sub-steps it gets, when
// the movement limiter or the refinement cap acts, when an attempt is flagged, when the guard's ladder accepts -- is scalar logic per
// environment, and until round 5 it was written out twice (plus the CPU checker's independent restatement in C, which
// stays independent: it is the thing these are tested against).  Everything here takes scalars and returns scalars; the operations and
// their order are exactly those of the round-5 code, so both layouts produce the bits they produced before (tests/test_gpu_fuzz.py,
// the step-by-step tests of tests/test_gpu_parity.py, quad-vs-one-lane guard words in tests/test_gpu_storm.py).
//
// Every tunable of the scheme is a macro below and a member of ScTunables with its provenance; tools/tunable_sensitivity.py perturbs
// them one at a time (profiles/r06_tunable_sensitivity.txt).  None of them was touched in
// round 6; the hold-out fixtures (tests/test_gpu_holdout.py) were generated after they were frozen.
//
// Included by gl_model.hpp right after Math<T> / ceil_pos (Ls5<T>, the five-stage scheme's coefficients, is only named here).
#pragma once

// (plain #defines: tools/tunable_sensitivity.py rebuilds the host instantiation from a copy of this file with ONE value changed)
#define SC_SAFETY 0.92
#define SC_MAX_REFINE 64
#define SC_GRACE_S 60.0
#define SC_GRACE_MUL 64.0
#define SC_CAP_S 120.0
#define SC_MOVE 8.0
#define SC_MOVE_HMAX 4.0
#define SC_PRE_MARGIN 1.02
#define SC_PRE_MAX 2.0
#define SC_BURST_STEPS 8.0
#define SC_BURST_DIV 8.0
#define SC_KEEP 0.97
#define SC_HEAVY 3
#define SC_AGREE 1e-2
#define SC_ATTEMPTS 4
#define SC_TOL_T 0.125
#define SC_TOL_P 12.5
#define SC_TOL_LAMP 0.5
#define SC_HARM_K 2.154e-3
#define SC_HARM_RELAX 0.3
#define SC_LOOK 4.0
#define SC_GROW 2.0001

namespace glm {

// Provenance of every tunable (value, what it does, where its value came from).  "fixtures" = step_tight / storm / jump / the two
// rollouts / the bench-workload tuples: the set the hold-out fixtures of round 6 are NOT part of.
struct ScTunables {
    static constexpr double safety = SC_SAFETY;            // fraction of the scheme's real-axis stability interval a sub-step may use; round 2: rate bound = 0.95-1.25 x lambda_max on 576 storm states
    static constexpr int max_refine = SC_MAX_REFINE;       // finest sub-step = nominal / 64; round 3 (16 in round 2): a pinned cover's 36 1/s needs it, jump tuples A / B
    static constexpr double grace_s = SC_GRACE_S;          // estimate tolerance is looser during the first 60 s of an env-step; round 2: a 0 -> 1 actuator jump reads 0.16 K and decays 5x per window
    static constexpr double grace_mul = SC_GRACE_MUL;      // ... by this factor (64 x 0.125 K = 8 K: only an instability reaches it)
    static constexpr double cap_s = SC_CAP_S;              // seconds at the refinement cap before an attempt is flagged SC_FLAG_CAP; round 2, a cold wet surface crossing the air node takes < 60 s
    static constexpr double move = SC_MOVE;                // movement limiter: no fast state moves more than move x tolerance (1 K / 100 Pa / 100 mg m-3) per sub-step; round 4 (4x tighter: RK4 rang on the initial layer at 3.75 s)
    static constexpr double move_hmax = SC_MOVE_HMAX;      // ls5: the allowance grows with the head-room below the stability limit, at most 4x; round 5 (limiter acted in 20 % of bench env-steps without it)
    static constexpr double pre_margin = SC_PRE_MARGIN;    // a window is shortened when 1.02 x its rate bound exceeds the nominal limit; round 4 (a lane 5 % over pays 5 %, not 50 %)
    static constexpr double pre_max = SC_PRE_MAX;          // up to 2x shorter windows follow the bound exactly; beyond: a burst
    static constexpr double burst_steps = SC_BURST_STEPS;  // a burst window holds 8 sub-steps at the bound ...
    static constexpr double burst_div = SC_BURST_DIV;      // ... and is at least 1/8 of the nominal window; round 5 (one env per launch took 128 sub-steps of 0.11 s through a 14 s window)
    static constexpr double keep = SC_KEEP;                // hysteresis: a window keeps its length while at most 3 % shorter than allowed; round 5 (every new length = five exponentials for the whole wavefront)
    static constexpr int heavy = SC_HEAVY;                 // an attempt with >= 3x the nominal sub-steps is not accepted alone; round 3
    static constexpr double agree = SC_AGREE;              // two attempts agree when the fast states differ by <= 1e-2 x the tolerances (1.25e-3 K); round 3, Richardson: the finer is then good to 1/15 of that
    static constexpr int attempts = SC_ATTEMPTS;           // ladder n, 2n, 4n, 8n; round 3
    static constexpr double tol_t = SC_TOL_T;              // error-estimate tolerance of a temperature [K]; round 2: accurate steps stay below 0.07 x on the storm fixture
    static constexpr double tol_p = SC_TOL_P;              // ... of a vapour pressure [Pa] / CO2 concentration [mg m-3]
    static constexpr double tol_lamp = SC_TOL_LAMP;        // ... of the lamp [K] (linear, jumps by tens of kelvin legitimately)
    static constexpr double harm_k = SC_HARM_K;            // a wet surface's singular slope is ignored while an unresolved step misplaces it by < 1e-4 max(|T|, 2 K): (1e-4)^(2/3), the accuracy bar itself; round 2
    static constexpr double harm_relax = SC_HARM_RELAX;    // ... or while its pinned equilibrium relaxes slower than ~0.1 1/s ((kap G)^3 > 0.3 rfree^2); round 2
    static constexpr double look = SC_LOOK;                // pinned analysis looks 4 nominal sub-steps ahead for the nearest point the surface can reach; round 2
    static constexpr double grow = SC_GROW;                // an adaptive window's sub-step at most doubles from one to the next (2.0001: the doubling meeting the equal partition exactly); round 5
};

constexpr int SC_FLAG_CAP = 1, SC_FLAG_NONFINITE = 2, SC_FLAG_ERR = 4, SC_FLAG_BRANCH = 8;
template <class T> struct Ls5;      // gl_model.hpp: coefficients and stability interval of the five-stage 2N scheme

// stability interval the sub-steps may use, and the factor of the embedded estimate, per scheme (ORDER 5 = ls5, 4 = RK4, 3, 2)
template <class T, int ORDER> struct ScScheme {
    static GL_HD T S() { return T(SC_SAFETY * (ORDER == 5 ? Ls5<T>::S : ORDER == 4 ? 2.785 : ORDER == 3 ? 2.5127 : 2.0)); }
    static GL_HD T est_fac() { return T(ORDER == 5 ? Ls5<T>::B(4) : 1.0 / 6.0); }
};

// the nominal windows of one attempt
template <class T> struct ScGrid {
    int winr, n_win;
    T hw_nom, hnom_nom, hmin, t_grace;
};
template <class T> GL_HD ScGrid<T> sc_grid(T dt, int n_sub, int winr)
{
    ScGrid<T> g;
    g.winr = winr;
    g.n_win = (n_sub + winr - 1) / winr;
    g.hw_nom = dt / T(g.n_win);
    g.hnom_nom = g.hw_nom / T(winr);
    g.hmin = g.hnom_nom * T(1.0 / SC_MAX_REFINE);
    g.t_grace = T((int)::ceil(SC_GRACE_S / (double)g.hw_nom)) * g.hw_nom + T(0.01) * g.hw_nom;
    return g;
}

// This window's length from its rate bound (round 5: the window LENGTH follows the bound).  In: hw = the window just taken (any value
// when first), t_left.  Out: hw, hnom = hw / winr, n_left = the equal windows the rest of the env-step is divided into.
//   sc = SC_PRE_MARGIN lam hnom_nom / S   <= 1: the nominal window;   <= SC_PRE_MAX: hw_nom / sc (the window's WIN sub-steps sit at the bound);
//   beyond: a burst -- SC_BURST_STEPS sub-steps at the bound, between hw_nom / SC_BURST_DIV and hw_nom / SC_PRE_MAX;
//   SC_KEEP hysteresis: the previous length stays while still allowed and at most 3 % shorter than what the bound now allows.
template <class T> GL_HD void sc_window_length(const ScGrid<T>& g, T S, T lam, T t_left, bool first, T& hw, T& hnom, int& n_left)
{
    using M = Math<T>;
    const T sc = T(SC_PRE_MARGIN) * lam * g.hnom_nom * M::rcp(S);
    T hw_t = g.hw_nom;                                      // (a NaN rate leaves the nominal window)
    if (sc > T(1) && sc <= T(SC_PRE_MAX)) hw_t = g.hw_nom * M::rcp(sc);
    else if (sc > T(SC_PRE_MAX))
        hw_t = M::min(g.hw_nom * T(1.0 / SC_PRE_MAX), M::max(g.hw_nom * T(1.0 / SC_BURST_DIV), T(SC_BURST_STEPS / SC_PRE_MARGIN) * S * M::rcp(lam)));
    const bool keep = !first && !(hw > hw_t * T(1.0 + 1e-6)) && hw >= T(SC_KEEP) * hw_t;
    hw_t = keep ? hw : hw_t;
    const T nl = M::max(T(1), ceil_pos(t_left * M::rcp(hw_t) - T(1e-3)));
    n_left = (int)nl;
    hw = (nl <= T(1)) ? t_left : (keep ? hw : t_left * M::rcp(nl));
    hnom = hw / T(g.winr);
}

// The window's sub-steps: as many equal ones as stability asks for (never fewer than WIN), shortened by the movement limiter
// (mv = the largest |rate| x 1 / tolerance over the limited fast states, gathered by the layout), held at the refinement cap.
template <class T> struct ScPlan {
    T h, n_rem;             // sub-step length, sub-steps in the window
    T hs_stab, move_allow;  // what stability alone allows; the limiter's allowance (both reused by sc_replan)
    bool limited, capped, adaptive;
};
template <class T, int ORDER> GL_HD T sc_move_allowance(T S, T lam, T hnom)
{
    using M = Math<T>;
    // ls5: what the limiter guards against is the rate bound going stale inside the window, so its allowance grows with the head-room
    // H = S / (lam hnom) in [1, SC_MOVE_HMAX] the bound leaves below the stability limit
    if (ORDER == 5) return T(SC_MOVE) * M::min(M::max(S * M::rcp(lam * hnom), T(1)), T(SC_MOVE_HMAX));
    return T(SC_MOVE);
}
template <class T, int ORDER> GL_HD ScPlan<T> sc_plan(const ScGrid<T>& g, T S, T lam, T hw, T hnom, T mv)
{
    using M = Math<T>;
    ScPlan<T> p;
    T hs = M::min(S * M::rcp(lam), hnom);
    p.hs_stab = hs;
    p.move_allow = sc_move_allowance<T, ORDER>(S, lam, hnom);
    p.limited = mv * hs > p.move_allow;
    hs = p.limited ? p.move_allow * M::rcp(mv) : hs;
    p.capped = !(hs >= g.hmin);                             // also true for a NaN rate
    hs = p.capped ? g.hmin : hs;
    p.n_rem = M::max(T(1), ceil_pos(hw * M::rcp(hs) - T(1e-3)));
    p.h = hw * M::rcp(p.n_rem);
    // ls5: a window whose sub-step was set by the limiter re-plans the REST of the window with the first stage of every sub-step
    p.adaptive = (ORDER == 5) && p.limited && !p.capped;
    return p;
}
// ls5, inside an adaptive window: the limiter again with this sub-step's first stage (mvj); the rest of the window (t_rem) re-partitioned,
// at most doubling from one sub-step to the next.  Returns through h / n_rem (unchanged unless `adaptive`).
template <class T> GL_HD void sc_replan(const ScGrid<T>& g, const ScPlan<T>& p, bool adaptive, T t_rem, T mvj, T& h, T& n_rem)
{
    using M = Math<T>;
    T hsj = (mvj * p.hs_stab > p.move_allow) ? p.move_allow * M::rcp(mvj) : p.hs_stab;
    hsj = !(hsj >= g.hmin) ? g.hmin : hsj;
    T nn = M::max(T(1), ceil_pos(t_rem * M::rcp(hsj) - T(1e-3)));
    T hj = t_rem * M::rcp(nn);
    const bool grow = hj > T(SC_GROW) * h;       // (2.0001: t_rem / nn IS 2 h in exact arithmetic when the doubling meets the equal partition)
    hj = grow ? T(2) * h : hj;
    nn = (grow && nn < T(2)) ? T(2) : nn;
    h = adaptive ? hj : h;
    n_rem = adaptive ? nn : n_rem;
}

// flags
GL_HD int sc_branch_flag(int side_prev, int side, bool capped_prev)
{
    // a wet surface that was below its air node at the last look (bits 3..5 of side_prev) and now sits above it inside the bistable
    // regime with positive drive (bits 0..2 of side) has jumped branches -- acted on only where the window just taken was capped
    return ((((side_prev >> 3) & side & 7) != 0) && capped_prev) ? SC_FLAG_BRANCH : 0;
}
template <class T> GL_HD int sc_estimate_flag(const ScGrid<T>& g, T worst, T h_last, T est_fac, T t_now)
{
    const T tolmul = (t_now <= g.t_grace) ? T(SC_GRACE_MUL) : T(1);
    return (worst * h_last * est_fac <= tolmul) ? 0 : SC_FLAG_ERR;    // NaN -> flagged
}
// the exact harvest flow stays half the window just taken ahead of the windows: -> how much further to apply it now
template <class T> GL_HD T sc_harvest_advance(T dt, T t_now, T hw, T& t_harv)
{
    using M = Math<T>;
    const T target = M::min(dt, t_now + T(0.5) * hw);
    const T hh = M::max(T(0), target - t_harv);
    t_harv = M::max(t_harv, target);
    return hh;
}

// Wet surface j (0 inner cover face, 1 thermal screen, 2 blackout screen) at a window start: its side bits for the branch invariant and
// whether the singular part of its slope can do harm (-> the pinned-rate analysis).  d(dT)/dt = rfree - kap |dT|^(1/3) (dT + G):
//   bit 3 + j: the surface is below its air node (dT > 0);
//   bit j (only evaluated when want_far): above it INSIDE THE BISTABLE REGIME with positive drive, 0 < rfree^3 < 27/256 kap^3 G^4;
//   harm: an unresolved step would misplace the surface by more than SC_HARM_BAR max(|T|, 2 K)  [(kap G h)^(3/2) > bar T  <=>
//         kap G h > bar^(2/3) T^(2/3), T^(2/3) by its chord over 2 ... 40 C]  AND a pinned equilibrium exists (dT > 0, rfree > 0)  AND it
//         could relax faster than ~0.1 1/s ((kap G)^3 > SC_HARM_RELAX rfree^2).
template <class T> GL_HD bool sc_wet_surface(bool on, int j, T iCap, T hcoef, T hec, T g, T tSurf, T dT, T ddT, T LK, T h_nominal, bool want_far, int& sbits)
{
    using M = Math<T>;
    const T tc = M::min(M::max(M::abs(tSurf), T(2)), T(40));
    const T kap = iCap * M::abs(hcoef), G = LK * M::max(g, T(0));
    const T kG = kap * G;
    const T rfree = ddT + iCap * hec * (dT + LK * g);
    const T kG3 = kG * kG * kG;
    sbits |= (on && dT > T(0)) ? (8 << j) : 0;
    if (want_far) sbits |= (on && (dT < T(0)) && (rfree > T(0)) && (rfree * rfree * rfree < T(27.0 / 256.0) * kG3 * G)) ? (1 << j) : 0;
    // SC_HARM_K = bar^(2/3) at the bar 1e-4; chord of T^(2/3) over [2, 40] C: 1.5874 + 0.26603 (T - 2)
    return on && (kG * h_nominal > T(SC_HARM_K) * (T(1.5874) + T(0.26603) * (tc - T(2)))) && (dT > T(0)) && (rfree > T(0)) &&
           (kG3 > T(SC_HARM_RELAX) * rfree * rfree);
}

// The guard's ladder (attempts with n, 2n, 4n, 8n nominal sub-steps): what to do with the attempt just finished.
//   finite: every increment of the attempt is finite;  worst: its distance from the previous attempt on the nine fast states in units of
//   the estimate tolerances (gathered by the layout; only used when a previous complete attempt exists).
// An attempt is CLEAN when it is complete (finite, not capped), carries no flag and took fewer than SC_HEAVY x the nominal sub-steps.
// Accepted: a clean attempt (unless verify), or a complete attempt that agrees with the previous complete one to SC_AGREE x the
// tolerances (step doubling; by Richardson the finer one is then good to a fifteenth of that), or the finest attempt when nothing
// flagged it.  first_flags (include/glgym.h GLGYM_SF_*): the first attempt's flags | 16 = heavy; | 32 accepted by agreement although
// flagged itself; | 64 the finest attempt alone.  The two-rungs-at-a-time ladder of gl_model_quad.hpp replays exactly these decisions
// with integer selects (its own form: a lane-mask miscompile of hipcc 7.2, tools/README.md "pair ladder").
struct ScLadder {
    int n, extra, total;
    bool done, ok, have_prev;
};
GL_HD ScLadder sc_ladder_start(int n_sub) { return ScLadder{n_sub, 0, 0, false, false, false}; }
template <class T> GL_HD void sc_ladder_judge(ScLadder& L, int attempt, int st_flags, int st_n_steps, int winr, bool finite, T worst, bool verify,
                                              int* first_flags)
{
    L.total += st_n_steps;
    const int n_nom = ((L.n + winr - 1) / winr) * winr;
    // diagnostics: why the FIRST attempt was not accepted as it stood (SC_FLAG_* | 16 = SC_HEAVY sub-steps)
    if (first_flags && attempt == 0) *first_flags = st_flags | ((st_n_steps >= SC_HEAVY * n_nom) ? 16 : 0);
    const bool complete = finite && !(st_flags & (SC_FLAG_CAP | SC_FLAG_NONFINITE));
    const bool clean = complete && st_flags == 0 && st_n_steps < SC_HEAVY * n_nom;
    const bool by_clean = clean && !verify, by_agree = complete && L.have_prev && worst <= T(SC_AGREE);
    L.ok = by_clean || by_agree || (attempt == SC_ATTEMPTS - 1 && complete && st_flags == 0);
    if (first_flags && L.ok && !by_clean) *first_flags |= by_agree ? ((st_flags != 0) ? 32 : 0) : 64;
    L.done = L.ok || attempt == SC_ATTEMPTS - 1;
    L.have_prev = complete;
    L.extra += L.done ? 0 : 1;
    L.n *= 2;
}
GL_HD int sc_ladder_extra_steps(const ScLadder& L, int n_sub, int winr)
{
    const int ex = L.total - ((n_sub + winr - 1) / winr) * winr;
    return ex > 0 ? ex : 0;
}

}  // namespace glm
