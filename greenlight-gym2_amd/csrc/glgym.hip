// glgym.hip -- gfx950 kernels + the C ABI declared in include/glgym.h.
//
// Kernels (one lane = one environment; a 64-lane wavefront = one workgroup = 64 environments):
//   step_kernel   fused TomatoEnv.step(): action->control, weather-row gather, tier-2 precompute,
//                 n_sub x RK4 of the GreenLight ODE (gl_model.hpp), failure check, reward / violation /
//                 info epilogue, terminal test, wave-level metric reduction.   [VALU/transcendental bound]
//   obs_kernel    row-major observation assembly incl. the weather-forecast gather.      [HBM-write bound]
//   reset_kernel  masked init_state().        crop_noise_kernel  Philox4x32-10 parameter noise.
//   evalf_kernel / rhs_kernel                 row-major double I/O for the reference-compatible evalF and tests.
// No MFMA anywhere: the path is elementwise + transcendental, not a contraction.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "glgym.h"
#include "gl_model.hpp"
#include "gl_model_quad.hpp"

using namespace glm;

namespace {

thread_local std::string g_err;

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                                 \
            return GLGYM_EHIP;                                                                         \
        }                                                                                              \
    } while (0)

constexpr int WAVE = 64;

// ---------------------------------------------------------------------------------------------------
// constants of the DEFAULT parameter block, visible to the compiler (see gen_default_const.cpp)
// ---------------------------------------------------------------------------------------------------
#include "gl_default_const.inc"
static_assert(sizeof(ModelConst<float>) == sizeof(kDefaultConstF32_image), "regenerate gl_default_const.inc");
static_assert(sizeof(ModelConst<double>) == sizeof(kDefaultConstF64_image), "regenerate gl_default_const.inc");
template <class T> struct DefaultConst;
template <> struct DefaultConst<float> {
    static constexpr ModelConst<float> value = __builtin_bit_cast(ModelConst<float>, kDefaultConstF32_bits);
};
template <> struct DefaultConst<double> {
    static constexpr ModelConst<double> value = __builtin_bit_cast(ModelConst<double>, kDefaultConstF64_bits);
};
// Device-side objects with the same constant initialisers: the kernels bind references to THESE (a host constexpr has
// no device address; only its folded values exist there).  Being const with a constant initialiser, every load from
// them that the optimiser can see still folds to a literal.
__constant__ const ModelConst<float> g_default_f32 = __builtin_bit_cast(ModelConst<float>, kDefaultConstF32_bits);
__constant__ const ModelConst<double> g_default_f64 = __builtin_bit_cast(ModelConst<double>, kDefaultConstF64_bits);
template <class T> __device__ __forceinline__ const ModelConst<T>& device_default();
template <> __device__ __forceinline__ const ModelConst<float>& device_default<float>() { return g_default_f32; }
template <> __device__ __forceinline__ const ModelConst<double>& device_default<double>() { return g_default_f64; }

// ---------------------------------------------------------------------------------------------------
// reward constants (rewards.py:96-124,156-231; TomatoEnv.yml:38-67)
// ---------------------------------------------------------------------------------------------------
template <class T> struct RewardConst {
    T heatK, elecK, co2K;       // cost per unit of u0 / u4 / u1 per env-step
    T gainK;                    // EUR per mg m-2 of fruit dry matter
    T minProfit, invRange;      // scale_reward(profit, min, max)
    T fixedCosts;
    T lo[3], hi[3], invMaxViol[3];
    T kPpm;
};

template <class T> void make_reward_const(const double* p, double dt, const glgym_reward_cfg& c, RewardConst<T>& r,
                                          double* max_profit, double* min_profit, double* fixed_costs)
{
    const double heat = p[108] / p[46] * dt / 3600 * 1e-3 * c.heating_price;
    const double elec = p[172] * dt / 3600 * 1e-3 * c.elec_price;
    const double co2 = p[109] / p[46] * dt * 1e-6 * c.co2_price;
    const double maxP = p[154] * dt * 1e-6 / c.dmfm * c.fruit_price;
    const double minP = -(heat + elec + co2);
    const double yearly = c.fixed_greenhouse_cost + c.fixed_co2_cost + c.fixed_lamp_cost * 116 + c.fixed_screen_cost;
    const double fixed = yearly / 365 / (double)(86400 / (long)dt);       // rewards.py:155 uses floor division
    r.heatK = T(heat); r.elecK = T(elec); r.co2K = T(co2);
    r.gainK = T(1e-6 / c.dmfm * c.fruit_price);
    r.minProfit = T(minP); r.invRange = T(1.0 / (maxP - minP)); r.fixedCosts = T(fixed);
    r.lo[0] = T(c.co2_min); r.lo[1] = T(c.temp_min); r.lo[2] = T(c.rh_min);
    r.hi[0] = T(c.co2_max); r.hi[1] = T(c.temp_max); r.hi[2] = T(c.rh_max);
    r.invMaxViol[0] = T(1.0 / 2500.0); r.invMaxViol[1] = T(1.0 / 15.0); r.invMaxViol[2] = T(1.0 / 15.0);   // :89-93
    r.kPpm = T(8.3144598 / (101325.0 * 44.01e-3));
    if (max_profit) *max_profit = maxP;
    if (min_profit) *min_profit = minP;
    if (fixed_costs) *fixed_costs = fixed;
}

template <class T> struct StepArgsT {
    int B, ld;
    T* x; T* u;
    const float* action; const T* control;
    const T* weather; int weather_rows;
    const int* w_off; int* timestep;
    const T* crop_p;
    int N;
    T* reward; T* info; unsigned char* done; float* metrics; int* step_flags;
    T dt; int n_sub;
    T gasR, tCanMin;
    int nd;                     // weather row stride (10, or 14 with the measured-pipe columns of ODE_pipe)
    float du, u_min[NU], u_max[NU];      // action_to_control: clip(u + action * delta_u_max, u_min, u_max)
    int verify;                 // 1: step-doubling verified integration (rk4_delta_guarded), see glgym_set_verify
    int pipe;                   // 1: the handle's ODE variant is GLGYM_ODE_PIPE (kernels that select the variant at run time)
    int window;                 // > 0: nominal sub-steps per tier-2b window, overriding the scheme's own (glgym_set_window)
};

template <class T> __device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

template <class T> __device__ __forceinline__ T sat_vp_exact(T t)
{
    return T(610.78) * Math<T>::exp(T(17.2694) * t / (t + T(238.3)));
}

// ---------------------------------------------------------------------------------------------------
// fused env-step
// ---------------------------------------------------------------------------------------------------
// DEFAULT_P = true: the handle's parameter block is bit-identical to the default one, so every tier-1 constant is a
// compile-time literal (no SGPRs, constant products folded, zero-coefficient exchange terms removed).
#ifndef GL_STEP_WAVES_PER_SIMD
#define GL_STEP_WAVES_PER_SIMD 1
#endif
// Tier-2b / harvest window of the RK4 scheme: FOUR sub-steps (15 s at the default n_sub 240) in both precisions (round 4; two in
// round 3, three mid-round).  The per-window work -- tier 2b, the rate-bound stage, estimate, limiter: 820 instructions -- is then
// shared by 16 stages.  Measured on the GPU, window 2 / 3 / 4 at n_sub 240: 6.82e7 / 7.7e7 / 8.15e7 env-steps/s; 10-day fixture
// fp32 2.41e-5 / 2.43e-5 / 2.93e-5, fp64 8.3e-6 / 8.5e-6 / 1.5e-5; storm fixture 4e-6 -> 8e-6, jump fixture 2.5e-5 -> 4.5e-5 (0 of
// 576 above 1e-4, 0 failed); tight one-step tuples 2.4e-5 / 3.5e-5 / 6.2e-5 in fp64 -- all of the growth one artificial tuple that
// starts with an EMPTY carbohydrate buffer (every other tuple 2.4e-5 at any window; the CPU studies); 0 flagged
// first attempts in 1.3e8 env-steps of the bench workload.  Window 5 would put that tuple at 9.6e-5: not taken.
// (Midpoint scheme: 4 sub-steps, three-stage scheme: 3.)
#ifndef GL_RK4_WIN_F32
#define GL_RK4_WIN_F32 4
#endif
#ifndef GL_RK4_WIN_F64
#define GL_RK4_WIN_F64 4
#endif
template <class T> struct RK4_WINDOW { static constexpr int value = sizeof(T) == 4 ? GL_RK4_WIN_F32 : GL_RK4_WIN_F64; };
// SCH (template argument of the integrating kernels) = GLGYM_SCHEME_*: 0 RK4, 1 the midpoint member (four sub-steps per tier-2b
// window), 2 the three-stage member (three) of the exponential family (gl_model.hpp rk_delta)
// 3 (GLGYM_SCHEME_LS5): the five-stage fourth-order 2N scheme, two sub-steps per window (gl_model.hpp rk_delta ORDER 5)
constexpr int gl_order(int sch) { return sch == 0 ? 4 : sch == 1 ? 2 : sch == 2 ? 3 : 5; }
template <class T, int SCH> struct SchemeWin { static constexpr int value = SCH == 0 ? RK4_WINDOW<T>::value : SCH == 1 ? 4 : SCH == 2 ? 3 : 2; };

// PIPE = true: the reference's ODE_pipe variant (ode.hpp:126-263; weather rows carry tPipe / pipeSwitchOff in columns 10 / 12).
// SCH = GLGYM_SCHEME_RK2 / _RK3: midpoint / three-stage sub-steps, tier 2b and the harvest flow shared by four / three of them,
// instead of RK4 sub-steps (GLGYM_SCHEME_RK4) -- rk_delta<T, PIPE, ORDER, WIN> in gl_model.hpp.  One lane per environment: fp32 only.
// OCC = waves per SIMD the kernel is compiled for.  1: up to 512 registers per lane, no scratch -- the right choice
// when the batch gives every SIMD one wave (B <= 65 536).  2: 256 registers per lane so that two waves share a SIMD -- a lone
// wave issues a vector instruction only every ~5 cycles, two co-resident waves one every ~2.7 (tools/microbench.hip with
// verified placement, profiles/r02_microbench_issue_rates.txt).  Rounds 2-4 measured 0.70x for that build: its 120 spilled
// registers were re-read from scratch at every window, 2 x 63 MB of scratch per launch did not fit the L2 (1.4 GB of HBM reads per
// launch, 41 % of the wave cycles waiting: tools/pmc_occ2.sh).  Round 5: what the windows read once each (z0, del, the slow
// slots' differences; x0 is re-read after the integrator) lives in LDS, 73 floats per lane = eight wavefronts per CU
// (rk_delta<WBUF>); 208 B of scratch are left and stay in the L2: 1.06x at B = 131 072, 1.08x at 262 144, 1.11x from 524 288 --
// the build batches of two or more wavefronts per SIMD take (launch_step).
template <class T, bool PER_ENV_CROP, bool DEFAULT_P, bool PIPE = false, int SCH = 0, int OCC = GL_STEP_WAVES_PER_SIMD>
__global__ __launch_bounds__(WAVE, OCC) void step_kernel(StepArgsT<T> a, ModelConst<T> m_arg, RewardConst<T> rw)
{
    const ModelConst<T>& m = DEFAULT_P ? device_default<T>() : m_arg;
    // OCC = 2 (round 5): what the windows read once each -- z0, the increments del, the slow slots' window differences -- lives in LDS,
    // WSTRIDE floats per lane (odd: conflict-free), 19.2 KB per wavefront = eight wavefronts per CU; the action tile of the prologue
    // shares the storage (it is consumed before the integrator starts).  gl_model.hpp rk_delta<WBUF>.
    constexpr bool WBUF = OCC == 2;
    constexpr int WSTRIDE = (2 * NX + GL_N_SLOW) | 1;
    __shared__ float sh_w[WBUF ? WAVE * WSTRIDE : WAVE * NU];
    float* sh_act = sh_w;
    const int lane = threadIdx.x;
    const int b0 = blockIdx.x * WAVE;
    const int b = b0 + lane;
    const bool live = b < a.B;
    const int bb = live ? b : a.B - 1;        // out-of-range lanes shadow the last env, stores are masked

    // ---- controls: coalesced tile load of the row-major action block through LDS
    T u[NU];
    if (a.action) {
        const int n_tile = min(WAVE, a.B - b0) * NU;
        for (int i = lane; i < n_tile; i += WAVE) sh_act[i] = a.action[(size_t)b0 * NU + i];
        __syncthreads();
        const int l = live ? lane : (a.B - 1 - b0);
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const float inc = sh_act[l * NU + j] * a.du;                     // f32 product, as tomato_env.py:113
            const T v = a.u[(size_t)j * a.ld + bb] + T(inc);
            u[j] = Math<T>::min(Math<T>::max(v, T(a.u_min[j])), T(a.u_max[j]));
        }
    } else {
#pragma unroll
        for (int j = 0; j < NU; ++j) u[j] = a.control[(size_t)j * a.ld + bb];
    }

    // ---- state + the weather row of this step (zero-order hold, tomato_env.py:120)
    T x0[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) x0[i] = a.x[(size_t)i * a.ld + bb];
    const int ts = a.timestep[bb];
    int row = a.w_off[bb] + ts;
    row = row < 0 ? 0 : (row >= a.weather_rows ? a.weather_rows - 1 : row);
    T d[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) d[j] = a.weather[(size_t)row * a.nd + j];

    // ---- tier 2, then the sub-stepped RK4
    CropConst<T> crLocal;
    if (PER_ENV_CROP) {
        T pc[NCROP];
#pragma unroll
        for (int i = 0; i < NCROP; ++i) pc[i] = a.crop_p[(size_t)i * a.ld + bb];
        make_crop_const<T, T>(pc, a.gasR, a.tCanMin, crLocal);
    }
    const CropConst<T>& cr = PER_ENV_CROP ? crLocal : m.crop;
    // (fp64: keeping this block in LDS instead -- 79 doubles x 64 lanes = 40 KB per wave -- was built and measured: the
    // generic kernel still needs 494 registers + 2 512 B of scratch, against 512 + 3 016 B, because the six 28-entry stage
    // arrays alone are 336 registers in fp64; not kept)
    StepCoef<T> s_reg;
    precompute(u, d, m, cr, s_reg);
    if (PIPE) {                                                           // ode.hpp:184-189
        const T tPipe = a.weather[(size_t)row * a.nd + 10], swOff = a.weather[(size_t)row * a.nd + 12];
        s_reg.pipeTrack = ((tPipe < T(1)) || (swOff > T(0))) ? T(0) : T(1);
        s_reg.tPipeSet = tPipe;
    }
    const StepCoef<T>& s = s_reg;
    // the applied control is final here (self.u is set before evalF and survives a failed integration, tomato_env.py:117-123):
    // store it now and re-read the three entries the reward needs afterwards, so that no u[] stays live across the integrator
    if (live) {
#pragma unroll
        for (int j = 0; j < NU; ++j) a.u[(size_t)j * a.ld + b] = u[j];
    }
    T del_reg[WBUF ? 1 : NX];
    T* wb = reinterpret_cast<T*>(sh_w) + (size_t)lane * WSTRIDE;
    T* del = WBUF ? wb + NX + GL_N_SLOW : del_reg;
    if (WBUF) {
        __syncthreads();                                   // every lane has taken its actions out of the shared storage
#pragma unroll
        for (int i = 0; i < NX; ++i) wb[i] = x0[i];
        wb[5] = x0[3] - x0[5]; wb[7] = x0[2] - x0[7]; wb[20] = x0[2] - x0[20]; wb[6] = x0[5] - x0[6];      // rk_delta's coordinates
    }
    bool bad;
    int extra_steps, first_flags = 0;
    const int retries = rk4_delta_guarded<T, PIPE, gl_order(SCH), SchemeWin<T, SCH>::value, WBUF>(x0, s, m, cr, a.dt, a.n_sub, del, &bad, &extra_steps, a.verify != 0, &first_flags, a.window, wb);
    const T uBoil = a.u[(size_t)0 * a.ld + bb], uCo2 = a.u[(size_t)1 * a.ld + bb], uLamp = a.u[(size_t)4 * a.ld + bb];
    if (WBUF) {                                            // the state again (not kept across the integrator in this build)
#pragma unroll
        for (int i = 0; i < NX; ++i) x0[i] = a.x[(size_t)i * a.ld + bb];
    }

    // ---- failure (tomato_env.py:119-123: on an integrator error the state is left unchanged and the env terminates)
    T x1[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) x1[i] = bad ? x0[i] : x0[i] + del[i];
    {   // x27 = time [days since reset]: exact from the step counter when the episode started at 0 (it always does in
        // the reference), so that fp32 storage does not random-walk over a 5 761-step season
        const double per_step = (double)a.dt / 86400.0;
        double t_start = (double)x0[NX - 1] - (double)ts * per_step;
        if (fabs(t_start) < 5e-4) t_start = 0.0;
        if (!bad) x1[NX - 1] = T(t_start + ((double)ts + 1.0) * per_step);
    }

    // ---- reward epilogue (rewards.py:156-231); indoor obs conversions (observations.py:70-77)
    const T co2ppm = rw.kPpm * (x1[2] + T(273.15)) * x1[0];
    const T rh = Math<T>::min(Math<T>::max(T(100) * x1[15] / sat_vp_exact(x1[2]), T(0)), T(100));
    const T o3[3] = {co2ppm, x1[2], rh};
    T viol[3], pen = T(0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        viol[i] = Math<T>::max(rw.lo[i] - o3[i], T(0)) + Math<T>::max(o3[i] - rw.hi[i], T(0));
        pen += viol[i] * rw.invMaxViol[i];
    }
    const T heat = uBoil * rw.heatK, elec = uLamp * rw.elecK, co2c = uCo2 * rw.co2K;
    const T varc = heat + co2c + elec;
    const T gains = (bad ? T(0) : del[25]) * rw.gainK;
    const T profit = gains - varc;
    const T reward = (profit - rw.minProfit) * rw.invRange - pen;        // lamp penalty is identically 0 (:203-212)
    const bool term = bad || (ts >= a.N);

    if (live) {
#pragma unroll
        for (int i = 0; i < NX; ++i) a.x[(size_t)i * a.ld + b] = x1[i];
        a.timestep[b] = ts + 1;
        a.reward[b] = reward;
        a.done[b] = term ? 1 : 0;
        if (a.step_flags) a.step_flags[b] = first_flags | (retries << 8) | (bad ? GLGYM_SF_FAILED : 0) | (min(extra_steps, 32767) << 16);
        if (a.info) {
            const T inf[GLGYM_NINFO] = {profit, gains, varc, rw.fixedCosts, co2c, heat, elec, viol[1], viol[0], viol[2], T(0)};
#pragma unroll
            for (int i = 0; i < GLGYM_NINFO; ++i) a.info[(size_t)i * a.ld + b] = inf[i];
        }
    }
    if (a.metrics) {     // wave-level reductions, one atomic per wave per metric
        const float w = live ? 1.f : 0.f;
        float mv[GLGYM_NMETRIC] = {w * (float)reward, w * (float)profit, (live && term) ? 1.f : 0.f,
                                   (live && bad) ? 1.f : 0.f, w * (float)viol[0], w * (float)viol[1],
                                   w * (float)viol[2], w, w * (float)retries, w * (float)extra_steps,
                                   (live && (first_flags & SC_FLAG_ERR)) ? 1.f : 0.f, (live && (first_flags & SC_FLAG_BRANCH)) ? 1.f : 0.f,
                                   (live && (first_flags & (SC_FLAG_CAP | SC_FLAG_NONFINITE))) ? 1.f : 0.f,
                                   (live && (first_flags & 16)) ? 1.f : 0.f};
        // one atomic per wave per metric, onto the wave's replica of the accumulator block (its own 128-byte line): 1 024 waves
        // finishing together onto ONE line serialised in the L2 for 52 us per launch
        float* mrep = a.metrics + (size_t)(blockIdx.x % GLGYM_METRIC_REPLICAS) * GLGYM_METRIC_STRIDE;
#pragma unroll
        for (int i = 0; i < GLGYM_NMETRIC; ++i) {
            const float sum = wave_sum(mv[i]);
            if (lane == 0) atomicAdd(mrep + i, sum);
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// fused env-step, FOUR LANES PER ENVIRONMENT (gl_model_quad.hpp): 16 environments per wavefront; for batches that leave SIMDs
// idle (B <= 16 384: profiles/r03_lanes_stage_proto.txt).  Same prologue / epilogue as step_kernel; lane 0 of a quad writes the
// per-env outputs, every lane the states it owns.
// ---------------------------------------------------------------------------------------------------
// SCH / CROP as in step_kernel: every scheme of the family, per-env crop blocks (config 5).  PIPE: ODE_pipe compiled in, selected at run
// time by a.pipe (gq_stage) -- in fp64 this kernel is the only step kernel (round 4: the one-lane fp64 kernels with their LDS mailbox
// and 2-3 KB of scratch are gone) and is always built with it.
// PAIR (round 6): TWO quads per environment run the verified ladder two rungs at a time (gl_model_quad.hpp rk4_delta_guarded_quad_pair, what
// glgym_evalF has done since round 5): for glgym_step(control = ...) -- step_raw_control, the rule-based controller: config 1 -- on batches
// that leave lanes free; the quad that holds the accepted attempt writes.  Bit-identical to the sequential ladder incl. step_flags.
template <class T, bool DEFAULT_P, int SCH = 0, bool PIPE = false, bool CROP = false, bool PAIR = false>
__global__ __launch_bounds__(WAVE) void step_kernel_quad(StepArgsT<T> a, ModelConst<T> m_arg, RewardConst<T> rw)
{
    static_assert(!(PAIR && CROP), "the two-rungs ladder is instantiated for shared crop constants only");
    // fp64: the handle's parameter block is staged in LDS as well (one uniform record per wavefront, broadcast reads behind the stage
    // fence): as a kernel argument its ~180 doubles live in SGPRs, of which there are 100 -- the compiler parks the rest in VGPR
    // lanes and pays two v_readlane per use (2 000 of the kernel's 20 000 static instructions)
    constexpr bool LDSM = sizeof(T) == 8 && !DEFAULT_P;
    __shared__ ModelConst<T> sh_m[1];
    if (LDSM) {
        static_assert(sizeof(ModelConst<T>) % 4 == 0, "word copy");
        const unsigned* src = reinterpret_cast<const unsigned*>(&m_arg);
        unsigned* dst = reinterpret_cast<unsigned*>(&sh_m[0]);
        for (int i = threadIdx.x; i < (int)(sizeof(ModelConst<T>) / 4); i += WAVE) dst[i] = src[i];
        __syncthreads();                       // one wavefront per block
    }
    const ModelConst<T>& m = LDSM ? sh_m[0] : DEFAULT_P ? device_default<T>() : m_arg;
    const int gl = blockIdx.x * WAVE + threadIdx.x, role = gl & 3;
    const int b = PAIR ? gl >> 3 : gl >> 2, half = PAIR ? (gl >> 2) & 1 : 0;
    const bool live = b < a.B;
    const int bb = live ? b : a.B - 1;        // out-of-range quads shadow the last env, stores are masked
    T u[NU];
    if (a.action) {
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const float inc = a.action[(size_t)bb * NU + j] * a.du;               // f32 product, as tomato_env.py:113
            const T v = a.u[(size_t)j * a.ld + bb] + T(inc);
            u[j] = Math<T>::min(Math<T>::max(v, T(a.u_min[j])), T(a.u_max[j]));
        }
    } else {
#pragma unroll
        for (int j = 0; j < NU; ++j) u[j] = a.control[(size_t)j * a.ld + bb];
    }
    const int ts = a.timestep[bb];
    int row = a.w_off[bb] + ts;
    row = row < 0 ? 0 : (row >= a.weather_rows ? a.weather_rows - 1 : row);
    T d[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) d[j] = a.weather[(size_t)row * a.nd + j];
    // fp64: the coefficient blocks live in LDS (rk_delta_quad<LDSQ>): StepCoef (and the env's crop constants) once per quad, LaneK per
    // lane, records padded to an odd number of 8-byte words (conflict-free ds_read_b64 across the lanes)
    constexpr bool LDSQ = sizeof(T) == 8;
    struct SRec { StepCoef<T> s; T pad[(sizeof(StepCoef<T>) / sizeof(T)) % 2 == 0 ? 1 : 2]; };
    struct KRec { LaneK<T> k; T pad[(sizeof(LaneK<T>) / sizeof(T)) % 2 == 0 ? 1 : 2]; };
    struct CRec { CropConst<T> c; T pad[(sizeof(CropConst<T>) / sizeof(T)) % 2 == 0 ? 1 : 2]; };
    __shared__ SRec sh_s[LDSQ ? WAVE / 4 : 1];
    __shared__ KRec sh_k[LDSQ ? WAVE : 1];
    __shared__ CRec sh_c[(LDSQ && CROP) ? WAVE / 4 : 1];
    StepCoef<T> s_reg;
    LaneK<T> k_reg;
    CropConst<T> cr_reg;
    if (CROP && (!LDSQ || role == 0)) {
        T pc[NCROP];
#pragma unroll
        for (int i = 0; i < NCROP; ++i) pc[i] = a.crop_p[(size_t)i * a.ld + bb];
        make_crop_const<T, T>(pc, a.gasR, a.tCanMin, cr_reg);
    }
    if (!LDSQ || role == 0) {
        precompute(u, d, m, CROP ? cr_reg : m.crop, s_reg);
        if (PIPE && a.pipe) {                                                 // ode.hpp:184-189
            const T tPipe = a.weather[(size_t)row * a.nd + 10], swOff = a.weather[(size_t)row * a.nd + 12];
            s_reg.pipeTrack = ((tPipe < T(1)) || (swOff > T(0))) ? T(0) : T(1);
            s_reg.tPipeSet = tPipe;
            s_reg.pipeOde = T(1);
        }
    }
    if (LDSQ) {
        if (role == 0) { sh_s[threadIdx.x >> 2].s = s_reg; if (CROP) sh_c[threadIdx.x >> 2].c = cr_reg; }
        __syncthreads();                       // one wavefront per block: orders the LDS writes before the quad's reads
    }
    const CropConst<T>& cr = !CROP ? m.crop : LDSQ ? sh_c[threadIdx.x >> 2].c : cr_reg;
    const StepCoef<T>& s = LDSQ ? sh_s[threadIdx.x >> 2].s : s_reg;
    LaneK<T>& K = LDSQ ? sh_k[threadIdx.x].k : k_reg;
    // the lane's states, in the integrator's coordinates (screens / inner cover face as differences to their air node)
    auto X = [&](int i) { return a.x[(size_t)i * a.ld + bb]; };
    QVec<T> x0, z0, del;
#pragma unroll
    for (int i = 0; i < 6; ++i) x0.sh[i] = X(role == 0 ? gq_sh_ix(i) : role == 1 ? gq_sh_ix(i) : role == 2 ? gq_sh_ix(i) : gq_sh_ix(i));
    x0.p = gq_mk<T>(X(role == 0 ? 4 : role == 1 ? 8 : role == 2 ? 7 : 5), X(role == 0 ? 9 : role == 1 ? 17 : role == 2 ? 20 : 6));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ix = role == 0 ? gq_other_ix(0, j) : role == 1 ? gq_other_ix(1, j) : role == 2 ? gq_other_ix(2, j) : gq_other_ix(3, j < 2 ? j : 1);
        x0.o[j] = X(ix);
    }
    z0 = x0;
    z0.p = gq_mk<T>(role == 2 ? x0.sh[2] - x0.p.x : role == 3 ? x0.sh[3] - x0.p.x : x0.p.x,
                    role == 2 ? x0.sh[2] - x0.p.y : role == 3 ? x0.p.x - x0.p.y : x0.p.y);      // cover lane: w = tCovIn - tCovE
    if (live && role == 0 && half == 0) {
#pragma unroll
        for (int j = 0; j < NU; ++j) a.u[(size_t)j * a.ld + b] = u[j];
    }
    bool bad;
    int extra_steps = 0, first_flags = 0, retries = 0, mine = 1;
    constexpr bool HVU = sizeof(T) == 8 || DEFAULT_P;         // harvest_flow's wave-uniform exits: not where the parameters live in SGPRs (fp32, handle parameters)
    if (PAIR) {
        int bad_i;
        rk4_delta_guarded_quad_pair<T, gl_order(SCH), SchemeWin<T, SCH>::value, LDSQ, PIPE, false, true, HVU>(role, half, z0, s, K, m, cr, a.dt, a.n_sub, del, &bad_i, &mine,
                                                                                                         a.window, &retries, &extra_steps, &first_flags);
        asm volatile("" : "+v"(bad_i), "+v"(mine));
        bad = bad_i != 0;
    } else {
        retries = rk4_delta_guarded_quad<T, gl_order(SCH), SchemeWin<T, SCH>::value, LDSQ, PIPE, LDSQ && CROP, HVU>(role, z0, s, K, m, cr, a.dt, a.n_sub, del, &bad,
                                                                                                    &extra_steps, a.verify != 0, &first_flags, a.window);
    }
    // ---- new state: physical increments of what the lane owns.  Nothing but the integrator's own state is kept live across the
    // integrator (the fp64 build is at its register limit there): the old state and the applied control are read again
    P2<T> dP;
    gq_phys_pair<T>(role, del, dP);
    asm volatile("" ::: "memory");
    // (opaque copies of the env indices: the address arithmetic of the epilogue is redone here instead of being carried -- as spilled
    // 64-bit pointers -- across the integrator)
    int gl2 = blockIdx.x * WAVE + threadIdx.x;
    asm volatile("" : "+v"(gl2));
    const int role2 = gl2 & 3, b2 = PAIR ? gl2 >> 3 : gl2 >> 2;
    const bool live2 = b2 < a.B && mine != 0;             // (PAIR: the quad that holds the accepted attempt writes)
    const int bb2 = (b2 < a.B) ? b2 : a.B - 1;
    const int ts2 = a.timestep[bb2];
    auto X2 = [&](int i) { return a.x[(size_t)i * a.ld + bb2]; };
#pragma unroll
    for (int i = 0; i < 6; ++i) x0.sh[i] = X2(gq_sh_ix(i));
    x0.p = gq_mk<T>(X2(role2 == 0 ? 4 : role2 == 1 ? 8 : role2 == 2 ? 7 : 5), X2(role2 == 0 ? 9 : role2 == 1 ? 17 : role2 == 2 ? 20 : 6));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ix = role2 == 0 ? gq_other_ix(0, j) : role2 == 1 ? gq_other_ix(1, j) : role2 == 2 ? gq_other_ix(2, j) : gq_other_ix(3, j < 2 ? j : 1);
        x0.o[j] = X2(ix);
    }
#pragma unroll
    for (int j = 0; j < NU; ++j) u[j] = a.u[(size_t)j * a.ld + bb2];       // (written above by lane 0 of the quad)
    QVec<T> x1;
    x1.p = gq_mk<T>(bad ? x0.p.x : x0.p.x + dP.x, bad ? x0.p.y : x0.p.y + dP.y);
#pragma unroll
    for (int i = 0; i < 6; ++i) x1.sh[i] = bad ? x0.sh[i] : x0.sh[i] + del.sh[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) x1.o[j] = bad ? x0.o[j] : x0.o[j] + del.o[j];
    if (role2 == 3) {    // x27 = time [days since reset]: exact from the step counter (step_kernel)
        const double per_step = (double)a.dt / 86400.0;
        double t_start = (double)x0.o[1] - (double)ts2 * per_step;
        if (fabs(t_start) < 5e-4) t_start = 0.0;
        if (!bad) x1.o[1] = T(t_start + ((double)ts2 + 1.0) * per_step);
    }
    const T dFruit = gq_bcast<2>(del.o[3]);                 // cFruit lives on lane 2
    // ---- reward epilogue (step_kernel): every lane has what it needs, lane 0 writes
    const T co2ppm = rw.kPpm * (x1.sh[2] + T(273.15)) * x1.sh[0];
    const T rh = Math<T>::min(Math<T>::max(T(100) * x1.sh[4] / sat_vp_exact(x1.sh[2]), T(0)), T(100));
    const T o3[3] = {co2ppm, x1.sh[2], rh};
    T viol[3], pen = T(0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        viol[i] = Math<T>::max(rw.lo[i] - o3[i], T(0)) + Math<T>::max(o3[i] - rw.hi[i], T(0));
        pen += viol[i] * rw.invMaxViol[i];
    }
    const T heat = u[0] * rw.heatK, elec = u[4] * rw.elecK, co2c = u[1] * rw.co2K;
    const T varc = heat + co2c + elec;
    const T gains = (bad ? T(0) : dFruit) * rw.gainK;
    const T profit = gains - varc;
    const T reward = (profit - rw.minProfit) * rw.invRange - pen;
    const bool term = bad || (ts2 >= a.N);
    if (live2) {
        auto W = [&](int i, T v) { a.x[(size_t)i * a.ld + b2] = v; };
        W(role2 == 0 ? 4 : role2 == 1 ? 8 : role2 == 2 ? 7 : 5, x1.p.x);
        W(role2 == 0 ? 9 : role2 == 1 ? 17 : role2 == 2 ? 20 : 6, x1.p.y);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (role2 != 3 || j < 2) W(role2 == 0 ? gq_other_ix(0, j) : role2 == 1 ? gq_other_ix(1, j) : role2 == 2 ? gq_other_ix(2, j) : gq_other_ix(3, j < 2 ? j : 1), x1.o[j]);
        if (role2 == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) W(gq_sh_ix(i), x1.sh[i]);
            a.timestep[b2] = ts2 + 1;
            a.reward[b2] = reward;
            a.done[b2] = term ? 1 : 0;
            if (a.step_flags) a.step_flags[b2] = first_flags | (retries << 8) | (bad ? GLGYM_SF_FAILED : 0) | (min(extra_steps, 32767) << 16);
            if (a.info) {
                const T inf[GLGYM_NINFO] = {profit, gains, varc, rw.fixedCosts, co2c, heat, elec, viol[1], viol[0], viol[2], T(0)};
#pragma unroll
                for (int i = 0; i < GLGYM_NINFO; ++i) a.info[(size_t)i * a.ld + b2] = inf[i];
            }
        }
    }
    if (a.metrics) {
        const bool cnt = live2 && role2 == 0;
        const float w = cnt ? 1.f : 0.f;
        float mv[GLGYM_NMETRIC] = {w * (float)reward, w * (float)profit, (cnt && term) ? 1.f : 0.f,
                                   (cnt && bad) ? 1.f : 0.f, w * (float)viol[0], w * (float)viol[1],
                                   w * (float)viol[2], w, w * (float)retries, w * (float)extra_steps,
                                   (cnt && (first_flags & SC_FLAG_ERR)) ? 1.f : 0.f, (cnt && (first_flags & SC_FLAG_BRANCH)) ? 1.f : 0.f,
                                   (cnt && (first_flags & (SC_FLAG_CAP | SC_FLAG_NONFINITE))) ? 1.f : 0.f,
                                   (cnt && (first_flags & 16)) ? 1.f : 0.f};
        float* mrep = a.metrics + (size_t)(blockIdx.x % GLGYM_METRIC_REPLICAS) * GLGYM_METRIC_STRIDE;
#pragma unroll
        for (int i = 0; i < GLGYM_NMETRIC; ++i) {
            const float sum = wave_sum(mv[i]);
            if (threadIdx.x == 0) atomicAdd(mrep + i, sum);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// reference-compatible step map, four lanes per row (what fp64 handles run: row-major double I/O, per-row crop blocks)
// ---------------------------------------------------------------------------------------------------
// PAIR (round 5): two quads per row, the verified ladder two rungs at a time (gl_model_quad.hpp rk4_delta_guarded_quad_pair) -- what
// glgym_evalF launches in verified mode while the batch leaves lanes free; bit-identical results, two thirds of the latency
template <class T, int SCH, bool PIPE, bool CROP, bool PAIR = false>
__global__ __launch_bounds__(WAVE) void evalf_kernel_quad(const double* x, const double* u, const double* d, const double* crop, int B, T dt,
                                                          int n_sub, T gasR, T tCanMin, ModelConst<T> m_arg, double* x_next, int nd,
                                                          int* n_failed, int verify, int pipe, int window)
{
    constexpr bool LDSM = sizeof(T) == 8;      // the parameter block in LDS (step_kernel_quad)
    __shared__ ModelConst<T> sh_m[1];
    if (LDSM) {
        const unsigned* src = reinterpret_cast<const unsigned*>(&m_arg);
        unsigned* dst = reinterpret_cast<unsigned*>(&sh_m[0]);
        for (int i = threadIdx.x; i < (int)(sizeof(ModelConst<T>) / 4); i += WAVE) dst[i] = src[i];
        __syncthreads();
    }
    const ModelConst<T>& m = LDSM ? sh_m[0] : m_arg;
    const int gl = blockIdx.x * WAVE + threadIdx.x, role = gl & 3, b = PAIR ? gl >> 3 : gl >> 2, half = PAIR ? (gl >> 2) & 1 : 0;
    const bool live = b < B;
    const int bb = live ? b : B - 1;
    T uu[NU], dd[7];
#pragma unroll
    for (int i = 0; i < NU; ++i) uu[i] = T(u[(size_t)bb * NU + i]);
#pragma unroll
    for (int i = 0; i < 7; ++i) dd[i] = T(d[(size_t)bb * nd + i]);
    constexpr bool LDSQ = sizeof(T) == 8;
    struct SRec { StepCoef<T> s; T pad[(sizeof(StepCoef<T>) / sizeof(T)) % 2 == 0 ? 1 : 2]; };
    struct KRec { LaneK<T> k; T pad[(sizeof(LaneK<T>) / sizeof(T)) % 2 == 0 ? 1 : 2]; };
    struct CRec { CropConst<T> c; T pad[(sizeof(CropConst<T>) / sizeof(T)) % 2 == 0 ? 1 : 2]; };
    __shared__ SRec sh_s[LDSQ ? WAVE / 4 : 1];
    __shared__ KRec sh_k[LDSQ ? WAVE : 1];
    __shared__ CRec sh_c[(LDSQ && CROP) ? WAVE / 4 : 1];
    StepCoef<T> s_reg;
    LaneK<T> k_reg;
    CropConst<T> cr_reg;
    if (CROP && (!LDSQ || role == 0)) {
        T pc[NCROP];
#pragma unroll
        for (int i = 0; i < NCROP; ++i) pc[i] = T(crop[(size_t)bb * NCROP + i]);
        make_crop_const<T, T>(pc, gasR, tCanMin, cr_reg);
    }
    if (!LDSQ || role == 0) {
        precompute(uu, dd, m, CROP ? cr_reg : m.crop, s_reg);
        if (PIPE && pipe) {
            const T tPipe = T(d[(size_t)bb * nd + 10]), swOff = T(d[(size_t)bb * nd + 12]);
            s_reg.pipeTrack = ((tPipe < T(1)) || (swOff > T(0))) ? T(0) : T(1);
            s_reg.tPipeSet = tPipe;
            s_reg.pipeOde = T(1);
        }
    }
    if (LDSQ) {
        if (role == 0) { sh_s[threadIdx.x >> 2].s = s_reg; if (CROP) sh_c[threadIdx.x >> 2].c = cr_reg; }
        __syncthreads();
    }
    const CropConst<T>& cr = !CROP ? m.crop : LDSQ ? sh_c[threadIdx.x >> 2].c : cr_reg;
    const StepCoef<T>& s = LDSQ ? sh_s[threadIdx.x >> 2].s : s_reg;
    LaneK<T>& K = LDSQ ? sh_k[threadIdx.x].k : k_reg;
    auto X = [&](int i) { return T(x[(size_t)bb * NX + i]); };
    QVec<T> x0, z0, del;
#pragma unroll
    for (int i = 0; i < 6; ++i) x0.sh[i] = X(gq_sh_ix(i));
    x0.p = gq_mk<T>(X(role == 0 ? 4 : role == 1 ? 8 : role == 2 ? 7 : 5), X(role == 0 ? 9 : role == 1 ? 17 : role == 2 ? 20 : 6));
#pragma unroll
    for (int j = 0; j < 4; ++j)
        x0.o[j] = X(role == 0 ? gq_other_ix(0, j) : role == 1 ? gq_other_ix(1, j) : role == 2 ? gq_other_ix(2, j) : gq_other_ix(3, j < 2 ? j : 1));
    z0 = x0;
    z0.p = gq_mk<T>(role == 2 ? x0.sh[2] - x0.p.x : role == 3 ? x0.sh[3] - x0.p.x : x0.p.x,
                    role == 2 ? x0.sh[2] - x0.p.y : role == 3 ? x0.p.x - x0.p.y : x0.p.y);
    bool bad;
    constexpr bool HVU = sizeof(T) == 8;       // fp32 rows carry their parameters in SGPRs: harvest_flow per lane (gl_model.hpp)
    int mine = 1;                        // (an integer on purpose: rk4_delta_guarded_quad_pair)
    if (PAIR) {
        int bad_i;
        rk4_delta_guarded_quad_pair<T, gl_order(SCH), SchemeWin<T, SCH>::value, LDSQ, PIPE, LDSQ && CROP, false, HVU>(role, half, z0, s, K, m, cr, dt, n_sub, del, &bad_i, &mine, window);
        asm volatile("" : "+v"(bad_i), "+v"(mine));
        bad = bad_i != 0;
    } else {
        int extra_steps, first_flags = 0;
        rk4_delta_guarded_quad<T, gl_order(SCH), SchemeWin<T, SCH>::value, LDSQ, PIPE, LDSQ && CROP, HVU>(role, z0, s, K, m, cr, dt, n_sub, del, &bad, &extra_steps, verify != 0,
                                                                                      &first_flags, window);
    }
    // a failed integration (the reference's evalF raises): the row is NaN and the call returns GLGYM_EODE
    P2<T> dP;
    gq_phys_pair<T>(role, del, dP);
    asm volatile("" ::: "memory");
    int gl2 = blockIdx.x * WAVE + threadIdx.x;
    asm volatile("" : "+v"(gl2));
    const int role2 = gl2 & 3, b2 = PAIR ? gl2 >> 3 : gl2 >> 2;
    if (b2 < B && mine != 0) {
        const double nan = __builtin_nan("");
        auto X2 = [&](int i) { return x[(size_t)b2 * NX + i]; };
        auto W = [&](int i, T dv) { x_next[(size_t)b2 * NX + i] = bad ? nan : X2(i) + (double)dv; };
        W(role2 == 0 ? 4 : role2 == 1 ? 8 : role2 == 2 ? 7 : 5, dP.x);
        W(role2 == 0 ? 9 : role2 == 1 ? 17 : role2 == 2 ? 20 : 6, dP.y);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (role2 != 3 || j < 2)
                W(role2 == 0 ? gq_other_ix(0, j) : role2 == 1 ? gq_other_ix(1, j) : role2 == 2 ? gq_other_ix(2, j) : gq_other_ix(3, j < 2 ? j : 1), del.o[j]);
        if (role2 == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) W(gq_sh_ix(i), del.sh[i]);
            if (bad) atomicAdd(n_failed, 1);
        }
    }
}

// the reference's right-hand side at one state per row (test hook behind glgym_rhs)
template <class T, bool PER_ENV_CROP, bool PIPE>
__global__ __launch_bounds__(WAVE) void rhs_kernel(const double* x, const double* u, const double* d, const double* crop, int B, T gasR,
                                                   T tCanMin, ModelConst<T> m, double* dx, int nd)
{
    const int b = blockIdx.x * WAVE + threadIdx.x;
    if (b >= B) return;
    T x0[NX], uu[NU], dd[7];
    for (int i = 0; i < NX; ++i) x0[i] = T(x[(size_t)b * NX + i]);
    for (int i = 0; i < NU; ++i) uu[i] = T(u[(size_t)b * NU + i]);
    for (int i = 0; i < 7; ++i) dd[i] = T(d[(size_t)b * nd + i]);
    CropConst<T> crLocal;
    if (PER_ENV_CROP) {
        T pc[NCROP];
        for (int i = 0; i < NCROP; ++i) pc[i] = T(crop[(size_t)b * NCROP + i]);
        make_crop_const<T, T>(pc, gasR, tCanMin, crLocal);
    }
    const CropConst<T>& cr = PER_ENV_CROP ? crLocal : m.crop;
    StepCoef<T> s;
    precompute(uu, dd, m, cr, s);
    if (PIPE) {
        const T tPipe = T(d[(size_t)b * nd + 10]), swOff = T(d[(size_t)b * nd + 12]);
        s.pipeTrack = ((tPipe < T(1)) || (swOff > T(0))) ? T(0) : T(1);
        s.tPipeSet = tPipe;
    }
    T k[NX];
    rhs<T, true, PIPE>(x0, s, m, cr, k);
    for (int i = 0; i < NX; ++i) dx[(size_t)b * NX + i] = (double)k[i];
}

// ---------------------------------------------------------------------------------------------------
// reference-compatible step map / RHS with row-major double I/O (B small; B = 1 for the drop-in evalF)
// ---------------------------------------------------------------------------------------------------
template <class T, bool PER_ENV_CROP, bool PIPE = false, int SCH = 0>
__global__ __launch_bounds__(WAVE) void evalf_kernel(const double* x, const double* u, const double* d,
                                                     const double* crop, int B, T dt, int n_sub, T gasR, T tCanMin,
                                                     ModelConst<T> m, double* x_next, int rhs_only, int nd,
                                                     int* n_failed, int verify, int window)
{
    const int b = blockIdx.x * WAVE + threadIdx.x;
    if (b >= B) return;
    T x0[NX], uu[NU], dd[7];
    for (int i = 0; i < NX; ++i) x0[i] = T(x[(size_t)b * NX + i]);
    for (int i = 0; i < NU; ++i) uu[i] = T(u[(size_t)b * NU + i]);
    for (int i = 0; i < 7; ++i) dd[i] = T(d[(size_t)b * nd + i]);
    CropConst<T> crLocal;
    if (PER_ENV_CROP) {
        T pc[NCROP];
        for (int i = 0; i < NCROP; ++i) pc[i] = T(crop[(size_t)b * NCROP + i]);
        make_crop_const<T, T>(pc, gasR, tCanMin, crLocal);
    }
    const CropConst<T>& cr = PER_ENV_CROP ? crLocal : m.crop;
    StepCoef<T> s;
    precompute(uu, dd, m, cr, s);
    if (PIPE) {
        const T tPipe = T(d[(size_t)b * nd + 10]), swOff = T(d[(size_t)b * nd + 12]);
        s.pipeTrack = ((tPipe < T(1)) || (swOff > T(0))) ? T(0) : T(1);
        s.tPipeSet = tPipe;
    }
    if (rhs_only) {
        T k[NX];
        rhs<T, true, PIPE>(x0, s, m, cr, k);
        for (int i = 0; i < NX; ++i) x_next[(size_t)b * NX + i] = (double)k[i];
        return;
    }
    T del[NX];
    bool failed;
    rk4_delta_guarded<T, PIPE, gl_order(SCH), SchemeWin<T, SCH>::value>(x0, s, m, cr, dt, n_sub, del, &failed, nullptr, verify != 0, nullptr, window);
    // a failed integration (the reference's evalF raises): the row is NaN and the call returns GLGYM_EODE
    for (int i = 0; i < NX; ++i)
        x_next[(size_t)b * NX + i] = failed ? __builtin_nan("") : (double)x0[i] + (double)del[i];
    if (failed) atomicAdd(n_failed, 1);
}

// ---------------------------------------------------------------------------------------------------
// observations: [B][23 + 5*Np] f32 row-major (observations.py:59-182).  One wavefront per env row so the
// 1 KB row is written with consecutive lanes on consecutive addresses.
// ---------------------------------------------------------------------------------------------------
template <class T> struct ObsArgsT {
    int B, ld;
    const T* x; const T* u; const T* weather; int weather_rows;
    const int* w_off; const int* timestep; const float* start_day;
    int Np; float* obs; double doy_inc, hod_inc;     // (dt/86400) mod 365 [days], dt/3600 [h] per env-step
    const unsigned char* mask; float* term_obs;
    int nd;                                          // weather row stride
    int moff[6], dim;                                // first column of each observation module (-1 = absent), row width
};

// OBS_ROWS (16) consecutive env rows = one contiguous, 32-byte aligned span of the row-major output.  The span is
// assembled in LDS and streamed out with 16-byte stores, lane t writing float4 t, t+256, ... (fully coalesced).
// The kernel is latency-, not bandwidth-bound (a block's chain is: scalar loads of the rows' clocks -> gathers -> LDS ->
// barrier -> stores), so everything is arranged to keep many independent loads in flight per lane: the 16 window bases
// are loaded up front, the 23 state / current-weather / clock features are straight-line code (pointer selects instead
// of a branch per feature: 3 loads in flight), and each lane gathers one forecast element for all 16 rows at once from
// the L2-resident weather table.  Measured at B = 65 536, Np = 48: 44 us (first LDS version) -> 33 (independent gathers)
// -> 26 (branch-free features) -> 24 us (16 rows per block) = 2.9 TB/s written; a plain fill of the buffer takes 11.6 us.
// Masked mode (auto-reset) copies only the finished rows, after saving their previous content as SB3's
// terminal_observation.
#ifndef GL_OBS_ROWS
#define GL_OBS_ROWS 16
#endif
constexpr int OBS_ROWS = GL_OBS_ROWS, OBS_NCORE = 23, OBS_MAX_NP = 128;     // LDS span = rows * (23 + 5 Np) floats <= 42 KB
template <class T> __global__ __launch_bounds__(256) void obs_kernel(ObsArgsT<T> a)
{
    constexpr int ROWS = OBS_ROWS, NCORE = OBS_NCORE;
    extern __shared__ float4 span4[];               // ROWS * dim floats (dynamic: 8.4 KB at Np = 48), 16-byte aligned
    float* span = reinterpret_cast<float*>(span4);
    const int tid = threadIdx.x;
    const int dim = a.dim;
    const float kPpm = (float)(8.3144598 / (101325.0 * 44.01e-3));
    for (int rb = blockIdx.x * ROWS; rb < a.B; rb += gridDim.x * ROWS) {
        const int nrows = min(ROWS, a.B - rb);
        if (a.mask) {                                 // masked mode: skip strips without a finished env
            int any = 0;
            for (int r = 0; r < nrows; ++r) any |= a.mask[rb + r];
            if (!any) continue;                       // block-uniform
        }
        // first weather row of each env's window (the row / "timestep" the reference shows is the pre-increment one):
        // block-uniform addresses -> scalar loads, all eight issued before anything depends on them
        int base_r[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int b = rb + (r < nrows ? r : 0);
            const int ts = a.timestep[b];
            base_r[r] = a.w_off[b] + (ts > 0 ? ts - 1 : 0);
        }
        for (int e = tid; e < ROWS * NCORE; e += 256) {
            // straight-line code for all 23 features: two pointer selects, three loads in flight, arithmetic selects --
            // a branch per feature would serialise one memory round trip per branch inside the wave
            const int r = e / NCORE, j = e - r * NCORE;
            const int b = rb + (r < nrows ? r : 0);
            const int ts = a.timestep[b];
            const int k = ts > 0 ? ts - 1 : 0;
            int base = 0;
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) base = (rr == r) ? base_r[rr] : base;
            base = base >= a.weather_rows ? a.weather_rows - 1 : (base < 0 ? 0 : base);
            const T* wrow = a.weather + (size_t)base * a.nd;
            // j: 0 co2_ppm(x0,x2) 1 x2 2 RH(x15,x2) 3 x9 | 4 x21 5 x25 6 x26 | 7..12 u | 13 d0 14 d1 15 RH(d2,d1)
            //    16 co2_ppm(d3,d1) 17 d4 | 18 timestep 19..22 sin/cos clocks      (observations.py:70-161)
            const int xi = j == 0 ? 0 : j == 1 ? 2 : j == 2 ? 15 : j == 3 ? 9 : j == 4 ? 21 : j == 5 ? 25 : 26;
            const T* p1 = j < 7 ? a.x + (size_t)xi * a.ld + b
                                : (j < 13 ? a.u + (size_t)(j - 7) * a.ld + b : wrow + (j < 18 ? j - 13 : 0));
            const T* p2 = j < 13 ? a.x + (size_t)2 * a.ld + b : wrow + 1;        // tAir or tOut
            const float prim = (float)*p1, aux = (float)*p2, sday = a.start_day[b];
            // fp32 hardware transcendentals: the observation block is float32 (observation_space dtype)
            const float sat = 610.78f * __builtin_amdgcn_exp2f(1.44269504f * 17.2694f * aux * __builtin_amdgcn_rcpf(aux + 238.3f));
            const float rh = fminf(fmaxf(100.0f * prim * __builtin_amdgcn_rcpf(sat), 0.0f), 100.0f);
            const float ppm = kPpm * (aux + 273.15f) * prim;
            // v_sin_f32 / v_cos_f32 take revolutions: sin(2*pi*x)   (tomato_env.py:126-128)
            const int c = j - 18;
            const double rev = (c <= 2) ? ((double)sday + (double)ts * a.doy_inc) * (1.0 / 365.0)
                                        : (double)ts * a.hod_inc * (1.0 / 24.0);
            const float fr = (float)(rev - floor(rev));
            const float clk = (c == 1 || c == 3) ? __builtin_amdgcn_sinf(fr) : __builtin_amdgcn_cosf(fr);
            float v = prim;
            v = (j == 0 || j == 16) ? ppm : v;
            v = (j == 2 || j == 15) ? rh : v;
            v = (j == 18) ? (float)k : v;
            v = (j > 18) ? clk : v;
            // column of feature j in the configured module order (TomatoEnv._get_obs concatenates the modules in
            // the order of the yml list, tomato_env.py:193-198)
            const int mo = j < 4 ? a.moff[0] : j < 7 ? a.moff[1] : j < 13 ? a.moff[2] : j < 18 ? a.moff[3] : a.moff[4];
            const int jl = j < 4 ? j : j < 7 ? j - 4 : j < 13 ? j - 7 : j < 18 ? j - 13 : j - 18;
            if (r < nrows && mo >= 0) span[r * dim + mo + jl] = v;
        }
        // raw forecast rows, no unit conversion (:175-182): element q of the block = weather[base+1 + q/5][q%5]
        const int nf = a.moff[5] >= 0 ? 5 * a.Np : 0;
        for (int q = tid; q < nf; q += 256) {
            const int i = q / 5, c = q - i * 5;              // division by a constant: mul + shift
            float v[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {                 // eight independent gathers in flight per lane
                int row = base_r[r] + 1 + i;
                row = row >= a.weather_rows ? a.weather_rows - 1 : (row < 0 ? 0 : row);
                v[r] = (float)a.weather[(size_t)row * a.nd + c];
            }
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
                if (r < nrows) span[r * dim + a.moff[5] + q] = v[r];
        }
        __syncthreads();
        float* out = a.obs + (size_t)rb * dim;
        if (!a.mask && nrows == ROWS) {                      // full span: 16-byte stores (rb*dim*4 is a multiple of 32)
            const int n4 = (ROWS * dim) >> 2;                // ROWS multiple of 8 -> exact
            float4* out4 = reinterpret_cast<float4*>(out);
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f* src4 = reinterpret_cast<const v4f*>(span4);
            v4f* dst4 = reinterpret_cast<v4f*>(out4);
            for (int e = tid; e < n4; e += 256) __builtin_nontemporal_store(src4[e], &dst4[e]);      // 69 MB per call, read by nobody on the device before the next step kernel has run
        } else {
            for (int r = 0; r < nrows; ++r) {
                if (a.mask && !a.mask[rb + r]) continue;
                for (int jj = tid; jj < dim; jj += 256) {
                    const size_t e = (size_t)r * dim + jj;
                    if (a.mask && a.term_obs) a.term_obs[(size_t)rb * dim + e] = out[e];     // SB3 terminal_observation
                    out[e] = span[e];
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// rule-based controller (baseline.py:68-227), one thread per env, fp64 throughout
// ---------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void rule_based_kernel(glgym_rule_cfg c, int B, int ld, const T* __restrict__ x,
                                                         const T* __restrict__ weather, int weather_rows,
                                                         const int* __restrict__ w_off, const int* __restrict__ timestep,
                                                         const float* __restrict__ start_day, const double* hour,
                                                         const double* doy_in, double doy_inc, double hod_inc,
                                                         T* __restrict__ control, int nd)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int ts = timestep[b];
    int row = w_off[b] + ts;
    row = row >= weather_rows ? weather_rows - 1 : (row < 0 ? 0 : row);
    const T* w = weather + (size_t)row * nd;
    const double iGlob = (double)w[0], tOut = (double)w[1], dli = (double)w[7], isDay = (double)w[8],
                 isDaySmooth = (double)w[9];
    const double co2Air = (double)x[b], tAir = (double)x[(size_t)2 * ld + b], vpAir = (double)x[(size_t)15 * ld + b];
    const double hod = hour ? hour[b] : fmod((double)ts * hod_inc, 24.0);
    const double doy = doy_in ? doy_in[b] : (double)start_day[b] + (double)ts * doy_inc;
    // proportional band: lo + (hi-lo) / (1 + exp(-2/pBand * ln(100) * (v - setPt - pBand/2)))   (:226-227)
    auto pband = [](double v, double sp, double band, double lo, double hi) {
        return lo + (hi - lo) * (1.0 / (1.0 + exp(-2.0 / band * 4.605170185988092 * (v - sp - band * 0.5))));
    };
    const double tod = c.lamps_on <= c.lamps_off ? (double)(c.lamps_on < hod && hod < c.lamps_off)
                                                 : (double)(c.lamps_on < hod || hod < c.lamps_off);              // :76-77
    const double doy_ok = c.lamps_day_start <= c.lamps_day_stop
                              ? (double)(c.lamps_day_start < doy && doy < c.lamps_day_stop)
                              : (double)(c.lamps_day_start < doy || doy < c.lamps_day_stop);                   // :85-86
    const double below_dli = (double)(dli < c.lamp_rad_sum_limit);
    const double lamp_no_cons = (double)(iGlob < c.lamps_off_sun) * below_dli * tod * doy_ok;                   // :98
    const double sw_on = fmax(0.0, fmin(1.0, hod - c.lamps_on + 1.0));                                          // :107
    const double sw_off = fmax(0.0, fmin(1.0, c.lamps_off - hod + 1.0));                                        // :113
    const double both = c.lamps_on == c.lamps_off ? 0.0
                        : (c.lamps_on < c.lamps_off ? fmin(sw_on, sw_off) : fmax(sw_on, sw_off));               // :119-120
    const double smooth_lamp = both * below_dli * doy_ok;                                                       // :128
    const double day_inside = fmax(smooth_lamp, isDay);                                                         // :133
    const double heat_sp = day_inside * c.temp_setpoint_day + (1.0 - day_inside) * c.temp_setpoint_night +
                           c.heat_correction * lamp_no_cons;                                                    // :136
    const double heat_max = heat_sp + c.heat_deadzone;
    const double co2_sp = day_inside * c.co2_day;
    const double co2_ppm = 1e6 * 8.3144598 * (tAir + 273.15) * (1e-6 * co2Air) / (101325.0 * 44.01e-3);         // :145
    const double rh_in = 100.0 * vpAir / (610.78 * exp(17.2694 * tAir / (tAir + 238.3)));                       // :151
    const double vent_heat = pband(tAir, heat_max, c.vent_heat_Pband, 0, 1);
    const double vent_rh = pband(rh_in, c.rh_max + 0.0 * c.mech_dehumid_Pband, c.vent_rh_Pband, 0, 1);
    const double vent_cold = pband(tAir, heat_sp - c.t_vent_off, c.vent_cold_Pband, 1, 0);
    const double th_sp = isDay * c.thScrSpDay + (1.0 - isDay) * c.thScrSpNight;
    const double th_cold = pband(tOut, th_sp, c.thScrPband, 0, 1);
    const double th_heat = pband(tAir, heat_sp + c.thScrDeadZone, -c.thScrPband, 1, 0);
    const double th_rh = fmax(pband(rh_in, c.rhMax + c.thScrRh, c.thScrRhPband, 1, 0), 1.0 - vent_cold);
    const double lamp_on = lamp_no_cons * pband(tAir, heat_max + c.lampExtraHeat, -0.5, 0, 1) *
                           (isDaySmooth + (1.0 - isDaySmooth)) *
                           fmax(pband(rh_in, c.rhMax + c.blScrExtraRh, -0.5, 0, 1), 1.0 - vent_cold);           // :189-191
    control[b] = (T)pband(tAir, heat_sp, c.tHeatBand, 0, 1);
    control[(size_t)1 * ld + b] = (T)pband(co2_ppm, co2_sp, c.co2Band, 0, 1);
    control[(size_t)2 * ld + b] = (T)fmin(th_cold, fmax(th_heat, th_rh));
    control[(size_t)3 * ld + b] = (T)fmin(vent_cold, fmax(vent_heat, vent_rh));
    control[(size_t)4 * ld + b] = (T)lamp_on;
    control[(size_t)5 * ld + b] = (T)(c.useBlScr * (1.0 - isDaySmooth) * lamp_on);
}

// counter-based generator shared by the reset (episode start draw) and crop-noise kernels
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned* out)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}


// ---------------------------------------------------------------------------------------------------
// masked reset: init_state (utils.py:13-46)
// ---------------------------------------------------------------------------------------------------
template <class T>
__global__ void reset_kernel(int B, int ld, const unsigned char* mask, T* x, T* u, int* timestep, const T* weather,
                             int weather_rows, int* w_off, const int* start_rows, const float* start_days, int n_starts,
                             float* start_day, int* episode, unsigned long long seed, int nd)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B || (mask && !mask[b])) return;
    if (start_rows && n_starts > 0) {            // draw this episode's start (tomato_env.py:236-244)
        const int ep = episode ? episode[b] : 0;
        unsigned rnd[4];
        philox4x32_10((unsigned)b, (unsigned)ep, 0x5eedu, 0u, (unsigned)seed, (unsigned)(seed >> 32), rnd);
        const int j = (int)(rnd[0] % (unsigned)n_starts);
        w_off[b] = start_rows[j];
        if (start_day && start_days) start_day[b] = start_days[j];
        if (episode) episode[b] = ep + 1;
    }
    int r = w_off[b];
    r = r < 0 ? 0 : (r >= weather_rows ? weather_rows - 1 : r);
    const double co2Out = (double)weather[(size_t)r * nd + 3], tSoOut = (double)weather[(size_t)r * nd + 6];
    const double tAir = 16.5;
    double xi[NX];
    xi[0] = xi[1] = co2Out;
    for (int i = 2; i <= 10; ++i) xi[i] = tAir;
    xi[4] = tAir + 4;
    xi[11] = 0.25 * (3.0 * tAir + tSoOut);
    xi[12] = 0.25 * (2.0 * tAir + 2 * tSoOut);
    xi[13] = 0.25 * (tAir + 3 * tSoOut);
    xi[14] = tSoOut;
    xi[15] = xi[16] = 90.0 / 100.0 * (610.78 * exp(17.2694 * tAir / (tAir + 238.3)));
    xi[17] = xi[18] = xi[19] = xi[20] = tAir;
    xi[21] = xi[4];
    xi[22] = 0.0; xi[23] = 9.5283e4; xi[24] = 2.5107e5; xi[25] = 5.5338e4; xi[26] = 3.0978e3; xi[27] = 0.0;
    for (int i = 0; i < NX; ++i) x[(size_t)i * ld + b] = T(xi[i]);
    for (int j = 0; j < NU; ++j) u[(size_t)j * ld + b] = T(0);
    timestep[b] = 0;
}

// ---------------------------------------------------------------------------------------------------
// crop-parameter noise (noise.py:3-23) with a counter-based generator: Philox4x32-10
// ---------------------------------------------------------------------------------------------------
template <class T>
__global__ void crop_noise_kernel(T* crop_p, int B, int ld, const float* p0, float scale, unsigned long long seed,
                                  unsigned long long draw)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float pn[NCROP + 2];
    for (int blk = 0; blk < 9; ++blk) {
        unsigned r[4];
        philox4x32_10((unsigned)b, (unsigned)draw, (unsigned)(draw >> 32), (unsigned)blk, (unsigned)seed,
                      (unsigned)(seed >> 32), r);
        for (int q = 0; q < 4; ++q) {
            const int i = blk * 4 + q;
            if (i < NCROP) {
                const float un = ((float)(r[q] >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0,1), 24 bits
                const float noise = (un - 0.5f) * scale;
                pn[i] = p0[i] + noise * p0[i];
            }
        }
    }
    pn[144 - CROP0] = pn[141 - CROP0] / pn[142 - CROP0];            // cLeafMax = laiMax / sla (noise.py:22)
    for (int i = 0; i < NCROP; ++i) crop_p[(size_t)i * ld + b] = T(pn[i]);
}


// ---------------------------------------------------------------------------------------------------
// VecNormalize on the device (HBM-bound: one read pass for the moments, one read + write pass to normalise)
// ---------------------------------------------------------------------------------------------------
// pass 1: per-feature sum and sum of squares over the batch.  Thread t owns columns t, t+256, ... (consecutive lanes ->
// consecutive addresses of a row), walks a strip of rows in registers (fp64), then one atomic pair per column.  Eight row
// loads are issued before the first is consumed: with two in flight the pass ran at 1.7 TB/s (41.6 us for 69 MB), the
// latency-bandwidth product of the chip wants ~60 KB in flight per CU.  The atomics go to one of VN_REP replicas of the
// accumulator block (merged by vecnorm_merge_kernel): 1 024 blocks x 526 fp64 atomics onto the 33 cache lines of a single
// block serialise in the L2 -- that, not the read, was most of the pass.
constexpr int VN_REP = 16;
__global__ __launch_bounds__(1024) void vecnorm_moments_kernel(const float* __restrict__ obs, int B, int dim,
                                                               double* __restrict__ acc_all)
{
    double* acc = acc_all + (size_t)(blockIdx.x % VN_REP) * 2 * dim;
    const int rows_per_block = (B + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(B, r0 + rows_per_block);
    for (int c = threadIdx.x; c < dim; c += blockDim.x) {       // blockDim >= dim in practice: one column per thread
        double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
        int r = r0;
        for (; r + 7 < r1; r += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = obs[(size_t)(r + j) * dim + c];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const double a = (double)v[j], b = (double)v[j + 1];
                s0 += a; q0 += a * a; s1 += b; q1 += b * b;
            }
        }
        for (; r < r1; ++r) { const double v = (double)obs[(size_t)r * dim + c]; s0 += v; q0 += v * v; }
        if (r1 > r0) { atomicAdd(acc + c, s0 + s1); atomicAdd(acc + dim + c, q0 + q1); }
    }
}

// discounted returns + their batch moments (wave reduction, one atomic pair per wave)
template <class T>
__global__ __launch_bounds__(256) void vecnorm_returns_kernel(const T* __restrict__ reward, double* __restrict__ returns,
                                                              int B, double gamma, double* __restrict__ acc2)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    double v = 0.0;
    if (b < B) { v = returns[b] * gamma + (double)reward[b]; returns[b] = v; }
    double s = v, q = v * v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, WAVE); q += __shfl_xor(q, o, WAVE); }
    __shared__ double part[2][4];
    const int wv = threadIdx.x / WAVE;
    if ((threadIdx.x & (WAVE - 1)) == 0) { part[0][wv] = s; part[1][wv] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {                                   // one atomic pair per 256-env block
        atomicAdd(acc2, part[0][0] + part[0][1] + part[0][2] + part[0][3]);
        atomicAdd(acc2 + 1, part[1][0] + part[1][1] + part[1][2] + part[1][3]);
    }
}

// RunningMeanStd.update_from_moments for every feature (and for the scalar return statistics), then the batch count, then
// per-column 1/sqrt(var + eps) (fp64, so that pass 2 is two fp64 ops per element) -- ONE block: the count every column
// reads is updated after a barrier, and scale[c] may overwrite acc[c], which only column c's own thread has read.
__global__ __launch_bounds__(1024) void vecnorm_merge_kernel(double* mean, double* var, double* count, const double* acc, int dim,
                                                            int B, double* ret_stats, const double* acc2, int do_obs,
                                                            int do_ret, double eps, double* scale)
{
    const double n = (double)B;
    const double cnt = *count;
    for (int c = threadIdx.x; c < dim; c += blockDim.x) {
        if (do_obs) {
            double sum = 0.0, sq = 0.0;
            for (int r = 0; r < VN_REP; ++r) { sum += acc[(size_t)r * 2 * dim + c]; sq += acc[(size_t)r * 2 * dim + dim + c]; }
            const double bm = sum / n, bv = fmax(sq / n - bm * bm, 0.0);
            const double tot = cnt + n, delta = bm - mean[c];
            const double m2 = var[c] * cnt + bv * n + delta * delta * cnt * n / tot;
            mean[c] += delta * n / tot;
            var[c] = m2 / tot;
        }
        if (scale) scale[c] = 1.0 / sqrt(var[c] + eps);
    }
    if (do_ret && threadIdx.x == 0) {
        const double bm = acc2[0] / n, bv = fmax(acc2[1] / n - bm * bm, 0.0);
        const double rc = ret_stats[2], tot = rc + n, delta = bm - ret_stats[0];
        const double m2 = ret_stats[1] * rc + bv * n + delta * delta * rc * n / tot;
        ret_stats[0] += delta * n / tot;
        ret_stats[1] = m2 / tot;
        ret_stats[2] = tot;
    }
    __syncthreads();
    if (do_obs && threadIdx.x == 0) *count = cnt + n;
}

// pass 2: clip((x - mean) * scale).  The subtraction stays in fp64 like SB3 (float32 obs - float64 mean): for
// near-constant features a ~1e-8 difference is divided by sqrt(eps).  Lane i handles element i (coalesced).
__global__ __launch_bounds__(1024) void vecnorm_apply_kernel(const float* __restrict__ obs, float* __restrict__ out,
                                                            int B, int dim, const double* __restrict__ mean,
                                                            const double* __restrict__ scale, float clip)
{
    const int rows_per_block = (B + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(B, r0 + rows_per_block);
    for (int c = threadIdx.x; c < dim; c += blockDim.x) {
        const double mu = mean[c], sc = scale[c];
        for (int r = r0; r < r1; ++r) {
            const size_t i = (size_t)r * dim + c;
            const float v = (float)(((double)obs[i] - mu) * sc);
            out[i] = fminf(fmaxf(v, -clip), clip);
        }
    }
}

template <class T>
__global__ void vecnorm_reward_kernel(const T* __restrict__ reward, float* __restrict__ out, double* __restrict__ returns,
                                      const unsigned char* __restrict__ done, int B, const double* ret_stats, double eps,
                                      float clip, int norm_reward)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float r = (float)reward[b];
    if (norm_reward) r = fminf(fmaxf((float)((double)reward[b] / sqrt(ret_stats[1] + eps)), -clip), clip);
    out[b] = r;
    if (done && done[b]) returns[b] = 0.0;
}


// ---------------------------------------------------------------------------------------------------
// Weather pipeline on the device (gl_gym/environments/utils.py:48-125): raw 300-s samples -> unit
// conversions -> daily light sum / daylight flags -> PCHIP resample to the env's grid.  fp64 throughout (it runs once per
// season table, not per step); the result is written in the handle's dtype.  Mirrors gl_gym_amd/utils.py
// (weather_from_raw), itself checked bit for bit against the reference's loader.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double w_sat_vp(double t) { return 610.78 * exp(17.2694 * t / (t + 238.3)); }

// columns 0..6 of the raw-grid table W[10][n] (SoA)
__global__ void weather_convert_kernel(int n, const double* __restrict__ time, const double* __restrict__ i_glob,
                                       const double* __restrict__ t_out, const double* __restrict__ rh,
                                       const double* __restrict__ wind, const double* __restrict__ t_sky, double co2_ppm,
                                       double* __restrict__ W)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double R = 8.3144598, C2K = 273.15, M_CO2 = 44.01e-3, M_H2O = 18.01528e-3, P_ATM = 101325.0;
    const double t = t_out[k];
    const double rho_v = (rh[k] / 100.0) * w_sat_vp(t) * M_H2O / (R * (t + C2K));              // rh2vaporDens
    const double rho_sat = (100.0 / 100.0) * w_sat_vp(t) * M_H2O / (R * (t + C2K));
    W[0 * (size_t)n + k] = i_glob[k];
    W[1 * (size_t)n + k] = t;
    W[2 * (size_t)n + k] = w_sat_vp(t) * (rho_v / rho_sat);                                    // vaporDens2pres
    W[3 * (size_t)n + k] = (P_ATM * 1e-6 * co2_ppm * M_CO2 / (R * (t + C2K))) * 1e6;           // co2ppm2dens * 1e6
    W[4 * (size_t)n + k] = wind[k];
    W[5 * (size_t)n + k] = t_sky[k];
    const double year = 3600.0 * 24.0 * 365.0;
    W[6 * (size_t)n + k] = 10.0 + 5.0 * sin(2.0 * 3.14159265358979323846 * (time[k] + 0.625 * year) / year);   // soilTempNl
}

// columns 7..9: the two sequential passes of the loader (dailLightSum: utils.py:216-250; computeisDay: :177-214, whose
// later tests see the ramps written by earlier ones).  One lane: O(n) work, n ~ 3e4.
__global__ void weather_scan_kernel(int n, const double* __restrict__ time, double* __restrict__ W)
{
    if (blockIdx.x != 0 || threadIdx.x != 0 || n < 2) return;
    const double* rad = W;
    double* dli = W + 7 * (size_t)n;
    double* is_day = W + 8 * (size_t)n;
    double* smooth = W + 9 * (size_t)n;
    const double c = 86400.0, interval = time[1] - time[0];
    auto next_jump = [&](int from) {
        for (int k = from < 0 ? 0 : from; k + 1 < n; ++k)
            if (floor(time[k + 1] / c) - floor(time[k] / c) == 1.0) return k;
        return -1;
    };
    int before = 0, j = next_jump(0), after = j >= 0 ? j + 1 : n, i = 0;
    while (i < n) {
        const int seg_end = after < n ? after : n;
        double sum = 0.0;
        for (int k = before; k <= after && k < n; ++k) sum += rad[k];
        for (int k = i; k < seg_end; ++k) dli[k] = sum * interval * 1e-6;
        i = seg_end;
        if (i >= n) break;
        before = after;
        j = next_jump(before + 2);
        after = j >= 0 ? j : n;
    }
    double dt_mean = (time[n - 1] - time[0]) / (double)(n - 1);
    for (int k = 0; k < n; ++k) { is_day[k] = rad[k] > 0.0 ? 1.0 : 0.0; smooth[k] = is_day[k]; }
    const int n_tr = (int)(3600.0 / dt_mean), half = n_tr / 2;
    if (n_tr < 2) return;
    const double step = 1.0 / (double)(n_tr - 1);
    bool in_sunset = false;
    for (int k = n_tr; k < n - n_tr; ++k) {
        const double cur = is_day[k], nxt = is_day[k + 1];
        if (cur == 0.0) {
            in_sunset = false;
            if (nxt == 1.0)
                for (int q = 0; q < 2 * half && q < n_tr; ++q) {
                    const double r = (q == n_tr - 1) ? 1.0 : (double)q * step;
                    is_day[k - half + q] = r;
                    smooth[k - half + q] = 1.0 / (1.0 + exp(-10.0 * (r - 0.5)));
                }
        } else if (cur == 1.0 && nxt == 0.0 && !in_sunset) {
            for (int q = 0; q < 2 * half && q < n_tr; ++q) {
                const double r = (q == n_tr - 1) ? 1.0 : (double)q * step;
                is_day[k - half + q] = 1.0 - r;
                smooth[k - half + q] = 1.0 - 1.0 / (1.0 + exp(-10.0 * (r - 0.5)));
            }
            in_sunset = true;
        }
    }
}

// PCHIP slopes (Fritsch-Carlson as scipy.interpolate.PchipInterpolator): D[c][k]
__device__ __forceinline__ double w_sign(double v) { return v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : 0.0); }
__device__ __forceinline__ double pchip_edge(double h0, double h1, double m0, double m1)
{
    double d = ((2.0 * h0 + h1) * m0 - h0 * m1) / (h0 + h1);
    if (w_sign(d) != w_sign(m0)) d = 0.0;
    else if (w_sign(m0) != w_sign(m1) && fabs(d) > 3.0 * fabs(m0)) d = 3.0 * m0;
    return d;
}
__global__ void pchip_slopes_kernel(int n, int n_col, const double* __restrict__ x, const double* __restrict__ W,
                                    double* __restrict__ D)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x, col = blockIdx.y;
    if (k >= n || col >= n_col) return;
    const double* y = W + (size_t)col * n;
    auto h = [&](int i) { return x[i + 1] - x[i]; };
    auto m = [&](int i) { return (y[i + 1] - y[i]) / h(i); };
    double d;
    if (n == 2) d = m(0);
    else if (k == 0) d = pchip_edge(h(0), h(1), m(0), m(1));
    else if (k == n - 1) d = pchip_edge(h(n - 2), h(n - 3), m(n - 2), m(n - 3));
    else {
        const double m0 = m(k - 1), m1 = m(k), h0 = h(k - 1), h1 = h(k);
        if (w_sign(m0) != w_sign(m1) || m0 == 0.0 || m1 == 0.0) d = 0.0;
        else {
            const double w1 = 2.0 * h1 + h0, w2 = h1 + 2.0 * h0;
            d = 1.0 / ((w1 / m0 + w2 / m1) / (w1 + w2));
        }
    }
    D[(size_t)col * n + k] = d;
}

// evaluation on linspace(x[0], x[n-1], n_out) in the power-series order of scipy's PPoly; iGlob < 1e-10 -> 0
template <class T>
__global__ void pchip_eval_kernel(int n, int n_col, int n_out, int nd, const double* __restrict__ x,
                                  const double* __restrict__ W, const double* __restrict__ D, T* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, col = blockIdx.y;
    if (i >= n_out || col >= n_col) return;
    const double start = x[0], stop = x[n - 1];
    const double stepo = n_out > 1 ? (stop - start) / (double)(n_out - 1) : 0.0;
    const double t = (i == n_out - 1 && n_out > 1) ? stop : (double)i * stepo + start;
    int lo = 0, hi = n - 1;                         // last k with x[k] <= t, clipped to n - 2
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (x[mid] <= t) lo = mid; else hi = mid; }
    const int k = lo > n - 2 ? n - 2 : lo;
    const double* y = W + (size_t)col * n;
    const double* d = D + (size_t)col * n;
    const double dx = x[k + 1] - x[k], slope = (y[k + 1] - y[k]) / dx;
    const double tt = (d[k] + d[k + 1] - 2.0 * slope) / dx;
    const double c0 = tt / dx, c1 = (slope - d[k]) / dx - tt, c2 = d[k], c3 = y[k];
    const double sdx = t - x[k];
    double res = 0.0, z = 1.0;
    res += c3 * z; z *= sdx;
    res += c2 * z; z *= sdx;
    res += c1 * z; z *= sdx;
    res += c0 * z;
    if (col == 0 && res < 1e-10) res = 0.0;
    out[(size_t)i * nd + col] = T(res);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------------
struct glgym_handle_s {
    int device = 0;
    int dtype = GLGYM_F32;
    int n_sub = 256;
    double dt = 900.0;
    double p[NP];
    ModelConst<float> mf;
    ModelConst<double> md;
    glgym_reward_cfg rcfg;
    RewardConst<float> rf;
    RewardConst<double> rd;
    double max_profit = 0, min_profit = 0, fixed_costs = 0;
    float* p0_crop_dev = nullptr;       // shared p[128..161] as f32 (noise kernel input)
    int* fail_dev = nullptr;            // glgym_evalF: number of rows whose integration failed
    int nd = ND;                        // weather / disturbance row stride: 10, or up to 16 (ODE_pipe reads columns 10, 12)
    int variant = GLGYM_ODE;            // GLGYM_ODE | GLGYM_ODE_PIPE
    int scheme = GLGYM_SCHEME_RK4;      // GLGYM_SCHEME_RK4 | GLGYM_SCHEME_RK2 | GLGYM_SCHEME_RK3
    int verify_mode = GLGYM_VERIFY_AUTO;   // glgym_set_verify
    int use_specialised = 1;            // GLGYM_GENERIC=1 in the environment forces the generic kernels (A/B tests)
    int window = 0;                     // glgym_set_window: 0 = the scheme's own
    int layout = GLGYM_LAYOUT_AUTO;     // glgym_set_layout (fp32); initial value from GLGYM_LAYOUT at glgym_create
    int occupancy = 0;                  // glgym_set_occupancy (one-lane fp32 kernel): 0 = by batch size; initial value from GLGYM_OCC at glgym_create
    int ladder_parallel = 1;            // glgym_set_ladder_parallel: verified glgym_evalF calls may run two rungs of the ladder side by side
    int n_simd = 1024;                  // SIMDs of the device (4 per CU)
    float du = 0.1f, u_min[NU] = {0, 0, 0, 0, 0, 0}, u_max[NU] = {1, 1, 1, 1, 1, 1};   // glgym_set_control_limits
    int obs_modules[6] = {0, 1, 2, 3, 4, 5};   // observation modules in output order (glgym_set_obs_modules)
    int n_obs_modules = 6;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // scratch for the host-pointer entry points
    double* scratch = nullptr;
    size_t scratch_elems = 0;
};

static void default_reward(glgym_reward_cfg& c)
{
    c.elec_price = 0.3; c.heating_price = 0.09; c.co2_price = 0.3; c.fruit_price = 1.6; c.dmfm = 0.065;
    c.fixed_greenhouse_cost = 15.0; c.fixed_co2_cost = 0.015; c.fixed_lamp_cost = 0.07; c.fixed_screen_cost = 2.0;
    c.pen_lamp = 0.1;
    c.co2_min = 300; c.co2_max = 1600; c.temp_min = 15; c.temp_max = 34; c.rh_min = 50; c.rh_max = 85;
}

static int refresh(glgym_handle h)
{
    std::memset(&h->mf, 0, sizeof h->mf);
    std::memset(&h->md, 0, sizeof h->md);
    make_model_const<float>(h->p, h->mf);
    make_model_const<double>(h->p, h->md);
    make_reward_const<float>(h->p, h->dt, h->rcfg, h->rf, &h->max_profit, &h->min_profit, &h->fixed_costs);
    make_reward_const<double>(h->p, h->dt, h->rcfg, h->rd, nullptr, nullptr, nullptr);
    float pc[NCROP];
    for (int i = 0; i < NCROP; ++i) pc[i] = (float)h->p[CROP0 + i];
    HIPCHK(hipMemcpy(h->p0_crop_dev, pc, sizeof pc, hipMemcpyHostToDevice));
    return GLGYM_OK;
}

// The device-pointer entry points launch on the CALLER's stream; HIP wants the calling thread's current device to be the stream's.
// One process per GPU (the layout this library is built for) never notices; a process that holds handles on several devices does:
// bind the handle's device for the duration of the call and put the caller's back (a no-op when it already is current).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    // a null handle binds nothing (the entry point then returns GLGYM_EINVAL without having created a context on any device)
    explicit DeviceGuard(const glgym_handle_s* h)
    {
        if (h && hipGetDevice(&prev) == hipSuccess && prev != h->device) switched = hipSetDevice(h->device) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

extern "C" {

const char* glgym_version(void) { return "glgym 0.5 (gfx950; ABI 5; stability-controlled, step-doubling-verified sub-steppers in delta form: five-stage fourth-order 2N scheme / RK4 / RK3 / midpoint)"; }
int glgym_abi_version(void) { return GLGYM_ABI_VERSION; }
const char* glgym_last_error(void) { return g_err.c_str(); }

int glgym_destroy(glgym_handle h);

int glgym_create(int nx, int nu, int nd, int np, double dt, const double* p, int dtype, int n_sub, int device,
                 glgym_handle* out)
{
    if (!out || !p || nx != NX || nu != NU || nd < ND || nd > 16 || np != NP || !(dt > 0) || n_sub < 1 ||
        (dtype != GLGYM_F32 && dtype != GLGYM_F64)) {
        g_err = "glgym_create: expected nx=28 nu=6 nd=10..16 np=208, dt>0, n_sub>=1, dtype in {F32,F64}";
        return GLGYM_EINVAL;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev) {
        g_err = "glgym_create: no usable HIP device (this library has no CPU fallback)";
        return GLGYM_ENODEV;
    }
    HIPCHK(hipSetDevice(device));
    glgym_handle h = new (std::nothrow) glgym_handle_s();
    if (!h) return GLGYM_ENOMEM;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            h->n_simd = 4 * prop.multiProcessorCount;
    }
    h->device = device; h->dtype = dtype; h->n_sub = n_sub; h->dt = dt; h->nd = nd;
    std::memcpy(h->p, p, sizeof h->p);
    default_reward(h->rcfg);
    if (const char* e = std::getenv("GLGYM_GENERIC")) h->use_specialised = (e[0] == '1') ? 0 : 1;
    if (const char* e = std::getenv("GLGYM_VERIFY"))          // default of glgym_set_verify: auto | always | never (A/B tests)
        h->verify_mode = (e[0] == 'n') ? GLGYM_VERIFY_NEVER : (e[0] == 'a' && e[1] == 'l') ? GLGYM_VERIFY_ALWAYS : GLGYM_VERIFY_AUTO;
    // initial values of glgym_set_layout / glgym_set_occupancy, read ONCE here (A/B tools set them before creating their handle); the
    // launch path reads handle state only
    if (const char* e = std::getenv("GLGYM_LAYOUT")) h->layout = (e[0] == 'q') ? GLGYM_LAYOUT_QUAD : (e[0] == 'o') ? GLGYM_LAYOUT_ONE : GLGYM_LAYOUT_AUTO;
    if (const char* e = std::getenv("GLGYM_OCC")) h->occupancy = (std::atoi(e) == 2) ? 2 : (std::atoi(e) == 1) ? 1 : 0;
    // from here on a failure must release what was acquired: run the steps through one exit point
    int rc = [&]() -> int {
        HIPCHK(hipMalloc(&h->p0_crop_dev, NCROP * sizeof(float)));
        HIPCHK(hipMalloc(&h->fail_dev, sizeof(int)));
        HIPCHK(hipEventCreate(&h->ev0));
        HIPCHK(hipEventCreate(&h->ev1));
        return refresh(h);
    }();
    if (rc != GLGYM_OK) {
        const std::string keep = g_err;
        (void)glgym_destroy(h);
        g_err = keep;
        return rc;
    }
    *out = h;
    return GLGYM_OK;
}

int glgym_destroy(glgym_handle h)
{
    if (!h) return GLGYM_EINVAL;
    (void)hipSetDevice(h->device);
    if (h->p0_crop_dev) (void)hipFree(h->p0_crop_dev);
    if (h->fail_dev) (void)hipFree(h->fail_dev);
    if (h->scratch) (void)hipFree(h->scratch);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    delete h;
    return GLGYM_OK;
}

int glgym_set_params(glgym_handle h, const double* p)
{
    if (!h || !p) return GLGYM_EINVAL;
    std::memcpy(h->p, p, sizeof h->p);
    return refresh(h);
}

int glgym_set_params_keep_reward_scale(glgym_handle h, const double* p)
{
    if (!h || !p) return GLGYM_EINVAL;
    const float minf = h->rf.minProfit, invf = h->rf.invRange;
    const double mind = h->rd.minProfit, invd = h->rd.invRange, maxp = h->max_profit, minp = h->min_profit;
    std::memcpy(h->p, p, sizeof h->p);
    const int rc = refresh(h);
    h->rf.minProfit = minf; h->rf.invRange = invf;
    h->rd.minProfit = mind; h->rd.invRange = invd;
    h->max_profit = maxp; h->min_profit = minp;
    return rc;
}

int glgym_set_model_variant(glgym_handle h, int variant)
{
    if (!h || (variant != GLGYM_ODE && variant != GLGYM_ODE_PIPE) || (variant == GLGYM_ODE_PIPE && h->nd < 14)) {
        g_err = "glgym_set_model_variant: GLGYM_ODE or GLGYM_ODE_PIPE (the latter needs a handle created with nd >= 14)";
        return GLGYM_EINVAL;
    }
    h->variant = variant;
    return GLGYM_OK;
}

int glgym_set_scheme(glgym_handle h, int scheme)
{
    if (!h || (scheme != GLGYM_SCHEME_RK4 && scheme != GLGYM_SCHEME_RK2 && scheme != GLGYM_SCHEME_RK3 && scheme != GLGYM_SCHEME_LS5)) {
        g_err = "glgym_set_scheme: GLGYM_SCHEME_RK4, GLGYM_SCHEME_RK2, GLGYM_SCHEME_RK3 or GLGYM_SCHEME_LS5";
        return GLGYM_EINVAL;
    }
    h->scheme = scheme;
    return GLGYM_OK;
}

int glgym_set_window(glgym_handle h, int window)
{
    if (!h || window < 0 || window > 8) { g_err = "glgym_set_window: 0 (the scheme's own) or 1..8 nominal sub-steps per window"; return GLGYM_EINVAL; }
    h->window = window;
    return GLGYM_OK;
}

int glgym_set_layout(glgym_handle h, int layout)
{
    if (!h || (layout != GLGYM_LAYOUT_AUTO && layout != GLGYM_LAYOUT_ONE && layout != GLGYM_LAYOUT_QUAD)) {
        g_err = "glgym_set_layout: GLGYM_LAYOUT_AUTO, GLGYM_LAYOUT_ONE or GLGYM_LAYOUT_QUAD";
        return GLGYM_EINVAL;
    }
    h->layout = layout;
    return GLGYM_OK;
}

int glgym_set_occupancy(glgym_handle h, int waves_per_simd)
{
    if (!h || waves_per_simd < 0 || waves_per_simd > 2) { g_err = "glgym_set_occupancy: 0 (by batch size), 1 or 2 waves per SIMD"; return GLGYM_EINVAL; }
    h->occupancy = waves_per_simd;
    return GLGYM_OK;
}

int glgym_set_ladder_parallel(glgym_handle h, int on)
{
    if (!h || (on != 0 && on != 1)) { g_err = "glgym_set_ladder_parallel: 0 or 1"; return GLGYM_EINVAL; }
    h->ladder_parallel = on;
    return GLGYM_OK;
}

int glgym_set_control_limits(glgym_handle h, const double* u_min, const double* u_max, double delta_u_max)
{
    if (!h || !u_min || !u_max || !(delta_u_max >= 0)) { g_err = "glgym_set_control_limits: bad arguments"; return GLGYM_EINVAL; }
    for (int j = 0; j < NU; ++j)
        if (!(u_min[j] <= u_max[j])) { g_err = "glgym_set_control_limits: u_min > u_max"; return GLGYM_EINVAL; }
    for (int j = 0; j < NU; ++j) { h->u_min[j] = (float)u_min[j]; h->u_max[j] = (float)u_max[j]; }    // float32 arrays,
    h->du = (float)delta_u_max;                                                                       // base_env.py:72-74
    return GLGYM_OK;
}

int glgym_set_verify(glgym_handle h, int mode)
{
    if (!h || (mode != GLGYM_VERIFY_AUTO && mode != GLGYM_VERIFY_ALWAYS && mode != GLGYM_VERIFY_NEVER)) {
        g_err = "glgym_set_verify: GLGYM_VERIFY_AUTO, GLGYM_VERIFY_ALWAYS or GLGYM_VERIFY_NEVER";
        return GLGYM_EINVAL;
    }
    h->verify_mode = mode;
    return GLGYM_OK;
}

int glgym_set_n_sub(glgym_handle h, int n_sub)
{
    if (!h || n_sub < 1) return GLGYM_EINVAL;
    h->n_sub = n_sub;
    return GLGYM_OK;
}

int glgym_set_reward(glgym_handle h, const glgym_reward_cfg* cfg)
{
    if (!h || !cfg) return GLGYM_EINVAL;
    h->rcfg = *cfg;
    return refresh(h);
}

int glgym_get_reward_scale(glgym_handle h, double* max_profit, double* min_profit, double* fixed_costs)
{
    if (!h) return GLGYM_EINVAL;
    if (max_profit) *max_profit = h->max_profit;
    if (min_profit) *min_profit = h->min_profit;
    if (fixed_costs) *fixed_costs = h->fixed_costs;
    return GLGYM_OK;
}

}  // extern "C"

// ---- host-pointer entry points --------------------------------------------------------------------
static int ensure_scratch(glgym_handle h, size_t elems)
{
    if (h->scratch_elems >= elems) return GLGYM_OK;
    if (h->scratch) (void)hipFree(h->scratch);
    h->scratch = nullptr; h->scratch_elems = 0;
    HIPCHK(hipMalloc(&h->scratch, elems * sizeof(double)));
    h->scratch_elems = elems;
    return GLGYM_OK;
}

template <class T, int SCH>
static void launch_evalf_sch(glgym_handle h, const ModelConst<T>& m, const double* p_used, const double* dx, const double* du,
                             const double* dd, const double* dcrop, int B, double* dout, dim3 grid, dim3 block)
{
    const int verify = h->verify_mode != GLGYM_VERIFY_NEVER;
    // verified calls on batches that leave lanes free run the ladder two rungs at a time, two quads per row (evalf_kernel_quad<PAIR>):
    // up to one wavefront per SIMD of the device (8 lanes per row); beyond that the sequential ladder does the same work on fewer lanes
    const bool pair = verify && h->ladder_parallel && !dcrop && (size_t)8 * B <= (size_t)WAVE * h->n_simd &&
                      (sizeof(T) == 8 || h->layout != GLGYM_LAYOUT_ONE);
    if constexpr (sizeof(T) == 8) {      // fp64: four lanes per row (no one-lane fp64 integrator exists any more)
        const dim3 qgrid((4 * B + WAVE - 1) / WAVE), pgrid((8 * B + WAVE - 1) / WAVE);
        const int pipe = h->variant == GLGYM_ODE_PIPE ? 1 : 0;
        if (pair) hipLaunchKernelGGL((evalf_kernel_quad<T, SCH, true, false, true>), pgrid, block, 0, (hipStream_t)0, dx, du, dd, dcrop, B, T(h->dt), h->n_sub,
                                     T(p_used[39]), T(p_used[162]), m, dout, h->nd, h->fail_dev, verify, pipe, h->window);
        else if (dcrop) hipLaunchKernelGGL((evalf_kernel_quad<T, SCH, true, true>), qgrid, block, 0, (hipStream_t)0, dx, du, dd, dcrop, B, T(h->dt), h->n_sub,
                                      T(p_used[39]), T(p_used[162]), m, dout, h->nd, h->fail_dev, verify, pipe, h->window);
        else hipLaunchKernelGGL((evalf_kernel_quad<T, SCH, true, false>), qgrid, block, 0, (hipStream_t)0, dx, du, dd, dcrop, B, T(h->dt), h->n_sub,
                                T(p_used[39]), T(p_used[162]), m, dout, h->nd, h->fail_dev, verify, pipe, h->window);
    } else {
        const dim3 pgrid((8 * B + WAVE - 1) / WAVE);
        if (pair) hipLaunchKernelGGL((evalf_kernel_quad<T, SCH, false, false, true>), pgrid, block, 0, (hipStream_t)0, dx, du, dd, dcrop, B, T(h->dt), h->n_sub,
                                     T(p_used[39]), T(p_used[162]), m, dout, h->nd, h->fail_dev, verify, 0, h->window);
        // fp32 rows without their own parameter block take the four-lanes-per-row kernel where glgym_step does (up to one round of
        // quad wavefronts: 16 384 rows on MI355X; glgym_set_layout overrides)
        else if (!dcrop && (h->layout == GLGYM_LAYOUT_QUAD || (h->layout == GLGYM_LAYOUT_AUTO && B <= 4 * h->n_simd * 4)))
            hipLaunchKernelGGL((evalf_kernel_quad<T, SCH, false, false, false>), dim3((4 * B + WAVE - 1) / WAVE), block, 0, (hipStream_t)0, dx, du, dd, dcrop, B,
                               T(h->dt), h->n_sub, T(p_used[39]), T(p_used[162]), m, dout, h->nd, h->fail_dev, verify, 0, h->window);
        else if (dcrop) hipLaunchKernelGGL((evalf_kernel<T, true, false, SCH>), grid, block, 0, (hipStream_t)0, dx, du, dd, dcrop, B, T(h->dt), h->n_sub,
                                      T(p_used[39]), T(p_used[162]), m, dout, 0, h->nd, h->fail_dev, verify, h->window);
        else hipLaunchKernelGGL((evalf_kernel<T, false, false, SCH>), grid, block, 0, (hipStream_t)0, dx, du, dd, dcrop, B, T(h->dt), h->n_sub,
                                T(p_used[39]), T(p_used[162]), m, dout, 0, h->nd, h->fail_dev, verify, h->window);
    }
}

template <class T>
static int run_evalf(glgym_handle h, const ModelConst<T>& m, const double* p_used, const double* dx, const double* du,
                     const double* dd, const double* dcrop, int B, double* dout, int rhs_only)
{
    const dim3 grid((B + WAVE - 1) / WAVE), block(WAVE);
    const bool pipe = h->variant == GLGYM_ODE_PIPE;
    if (rhs_only) {                      // the right-hand side at one state per row (test hook): one lane per row in either dtype
        if (pipe) hipLaunchKernelGGL((rhs_kernel<T, false, true>), grid, block, 0, (hipStream_t)0, dx, du, dd, dcrop, B, T(p_used[39]), T(p_used[162]), m, dout, h->nd);
        else if (dcrop) hipLaunchKernelGGL((rhs_kernel<T, true, false>), grid, block, 0, (hipStream_t)0, dx, du, dd, dcrop, B, T(p_used[39]), T(p_used[162]), m, dout, h->nd);
        else hipLaunchKernelGGL((rhs_kernel<T, false, false>), grid, block, 0, (hipStream_t)0, dx, du, dd, dcrop, B, T(p_used[39]), T(p_used[162]), m, dout, h->nd);
        HIPCHK(hipGetLastError());
        return GLGYM_OK;
    }
    if (pipe) {
        if (dcrop || h->scheme != GLGYM_SCHEME_RK4) {
            g_err = "glgym_evalF: GLGYM_ODE_PIPE supports neither per-row parameter blocks nor schemes other than GLGYM_SCHEME_RK4";
            return GLGYM_EINVAL;
        }
        if constexpr (sizeof(T) == 8) {      // (the fp64 kernels select the variant at run time)
            launch_evalf_sch<T, GLGYM_SCHEME_RK4>(h, m, p_used, dx, du, dd, dcrop, B, dout, grid, block);
        } else {
            const int verify = h->verify_mode != GLGYM_VERIFY_NEVER;
            hipLaunchKernelGGL((evalf_kernel<T, false, true>), grid, block, 0, (hipStream_t)0, dx, du, dd, dcrop, B, T(h->dt), h->n_sub,
                               T(p_used[39]), T(p_used[162]), m, dout, 0, h->nd, h->fail_dev, verify, h->window);
        }
    } else if (h->scheme == GLGYM_SCHEME_RK2) {
        launch_evalf_sch<T, GLGYM_SCHEME_RK2>(h, m, p_used, dx, du, dd, dcrop, B, dout, grid, block);
    } else if (h->scheme == GLGYM_SCHEME_RK3) {
        launch_evalf_sch<T, GLGYM_SCHEME_RK3>(h, m, p_used, dx, du, dd, dcrop, B, dout, grid, block);
    } else if (h->scheme == GLGYM_SCHEME_LS5) {
        launch_evalf_sch<T, GLGYM_SCHEME_LS5>(h, m, p_used, dx, du, dd, dcrop, B, dout, grid, block);
    } else {
        launch_evalf_sch<T, GLGYM_SCHEME_RK4>(h, m, p_used, dx, du, dd, dcrop, B, dout, grid, block);
    }
    HIPCHK(hipGetLastError());
    return GLGYM_OK;
}

static int evalf_impl(glgym_handle h, const double* x, const double* u, const double* d, const double* p, int p_rows,
                      int B, double* out, int rhs_only)
{
    if (!h || !x || !u || !d || !out || B < 1 || (p && p_rows != 1 && p_rows != B)) {
        g_err = "glgym_evalF: bad arguments";
        return GLGYM_EINVAL;
    }
    HIPCHK(hipSetDevice(h->device));
    const double* p_used = p ? p : h->p;
    // per-row parameter blocks may differ only inside the crop block p[128..161] (what noise.py perturbs);
    // anything else is handled one row at a time.
    bool per_row = p && p_rows == B && B > 1;
    if (per_row) {
        for (int b = 1; b < B && per_row; ++b)
            for (int i = 0; i < NP; ++i)
                if ((i < CROP0 || i >= CROP0 + NCROP) && p[(size_t)b * NP + i] != p[i]) {
                    for (int r = 0; r < B; ++r) {
                        const int rc = evalf_impl(h, x + (size_t)r * NX, u + (size_t)r * NU, d + (size_t)r * h->nd,
                                                  p + (size_t)r * NP, 1, 1, out + (size_t)r * NX, rhs_only);
                        if (rc != GLGYM_OK) return rc;
                    }
                    return GLGYM_OK;
                }
    }
    const size_t n_in = (size_t)B * (NX + NU + h->nd + (per_row ? NCROP : 0));
    int rc = ensure_scratch(h, n_in + (size_t)B * NX);
    if (rc != GLGYM_OK) return rc;
    double* dx = h->scratch; double* du = dx + (size_t)B * NX; double* dd = du + (size_t)B * NU;
    double* dcrop = per_row ? dd + (size_t)B * h->nd : nullptr;
    double* dout = h->scratch + n_in;
    HIPCHK(hipMemcpy(dx, x, (size_t)B * NX * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(du, u, (size_t)B * NU * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dd, d, (size_t)B * h->nd * sizeof(double), hipMemcpyHostToDevice));
    if (per_row) {
        std::vector<double> crop((size_t)B * NCROP);
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < NCROP; ++i) crop[(size_t)b * NCROP + i] = p[(size_t)b * NP + CROP0 + i];
        HIPCHK(hipMemcpy(dcrop, crop.data(), crop.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemset(h->fail_dev, 0, sizeof(int)));
    HIPCHK(hipMemset(dout, 0xFF, (size_t)B * NX * sizeof(double)));      // a row no lane writes reads back as NaN, not as the previous call's
    if (h->dtype == GLGYM_F32) {
        ModelConst<float> m = h->mf;
        if (p) make_model_const<float>(p_used, m);
        rc = run_evalf<float>(h, m, p_used, dx, du, dd, dcrop, B, dout, rhs_only);
    } else {
        ModelConst<double> m = h->md;
        if (p) make_model_const<double>(p_used, m);
        rc = run_evalf<double>(h, m, p_used, dx, du, dd, dcrop, B, dout, rhs_only);
    }
    if (rc != GLGYM_OK) return rc;
    HIPCHK(hipMemcpy(out, dout, (size_t)B * NX * sizeof(double), hipMemcpyDeviceToHost));
    int n_failed = 0;
    HIPCHK(hipMemcpy(&n_failed, h->fail_dev, sizeof(int), hipMemcpyDeviceToHost));
    if (n_failed > 0) {
        g_err = "glgym_evalF: the integration failed for " + std::to_string(n_failed) + " of " + std::to_string(B) +
                " rows (no two consecutive attempts of the n_sub, 2x, 4x, 8x ladder agreed); their rows of x_next are NaN";
        return GLGYM_EODE;
    }
    // Every row of a call that reports no failure has been written by exactly one lane group; the buffer was NaN-filled before the
    // launch, so a row that still reads NaN here was written by NO lane -- the signature of the lane-mask miscompile of round 5
    // (tools/README.md "pair ladder"; designed around, not root-caused).  Never hand such a row back with GLGYM_OK.
    if (!rhs_only) {
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < NX; ++i)
                if (std::isnan(out[(size_t)b * NX + i])) {
                    g_err = "glgym_evalF: row " + std::to_string(b) + " of " + std::to_string(B) + " came back unwritten (state " + std::to_string(i) +
                            " is NaN although no integration was reported as failed): internal error of the kernel's row-writing lane selection";
                    return GLGYM_EHIP;
                }
    }
    return GLGYM_OK;
}

extern "C" {

int glgym_evalF(glgym_handle h, const double* x, const double* u, const double* d, const double* p, int p_rows, int B,
                double* x_next)
{
    return evalf_impl(h, x, u, d, p, p_rows, B, x_next, 0);
}

int glgym_rhs(glgym_handle h, const double* x, const double* u, const double* d, int B, double* dx)
{
    return evalf_impl(h, x, u, d, nullptr, 1, B, dx, 1);
}

}  // extern "C"

// ---- device-pointer hot path ----------------------------------------------------------------------
// one lane per environment (fp32 only since round 4)
template <int SCH>
static void launch_step_sch(const glgym_step_args* a, const StepArgsT<float>& k, const ModelConst<float>& m, const RewardConst<float>& rw,
                            dim3 grid, dim3 block, hipStream_t st, bool def, bool occ2)
{
    using T = float;
    if (occ2) {
        hipLaunchKernelGGL((step_kernel<T, false, true, false, SCH, 2>), grid, block, 0, st, k, m, rw);
        return;
    }
    if (a->crop_p) {
        if (def) hipLaunchKernelGGL((step_kernel<T, true, true, false, SCH>), grid, block, 0, st, k, m, rw);
        else hipLaunchKernelGGL((step_kernel<T, true, false, false, SCH>), grid, block, 0, st, k, m, rw);
    } else {
        if (def) hipLaunchKernelGGL((step_kernel<T, false, true, false, SCH>), grid, block, 0, st, k, m, rw);
        else hipLaunchKernelGGL((step_kernel<T, false, false, false, SCH>), grid, block, 0, st, k, m, rw);
    }
}

// four lanes per environment.  fp64: every scheme, per-env crop blocks, the handle's parameters as a kernel argument; fp32 (small
// batches): the shared-crop kernels of every scheme, with the default block compiled in where the handle holds it.
// (No fp64 build with the default block compiled in: measured without the scheduler flag that used to break it -- csrc/Makefile --
// it buys 0.7 % over the LDS-staged block, 1.379e6 against 1.369e6 env-steps/s at config 2, for six more 500-register kernels.)
// pair: verified steps on batches that leave lanes free run the ladder two rungs at a time on two quads per environment (PAIR)
template <class T, int SCH>
static void launch_quad_sch(const glgym_step_args* a, const StepArgsT<T>& k, const ModelConst<T>& m, const RewardConst<T>& rw,
                            dim3 qgrid, dim3 block, hipStream_t st, bool def, bool pair)
{
    const dim3 pgrid((8 * (size_t)a->B + WAVE - 1) / WAVE);
    if constexpr (sizeof(T) == 8) {          // ODE_pipe compiled in, selected by k.pipe
        if (pair) hipLaunchKernelGGL((step_kernel_quad<T, false, SCH, true, false, true>), pgrid, block, 0, st, k, m, rw);
        else if (a->crop_p) hipLaunchKernelGGL((step_kernel_quad<T, false, SCH, true, true>), qgrid, block, 0, st, k, m, rw);
        else hipLaunchKernelGGL((step_kernel_quad<T, false, SCH, true, false>), qgrid, block, 0, st, k, m, rw);
    } else {
        if (pair && def) hipLaunchKernelGGL((step_kernel_quad<T, true, SCH, false, false, true>), pgrid, block, 0, st, k, m, rw);
        else if (pair) hipLaunchKernelGGL((step_kernel_quad<T, false, SCH, false, false, true>), pgrid, block, 0, st, k, m, rw);
        else if (def) hipLaunchKernelGGL((step_kernel_quad<T, true, SCH, false, false>), qgrid, block, 0, st, k, m, rw);
        else hipLaunchKernelGGL((step_kernel_quad<T, false, SCH, false, false>), qgrid, block, 0, st, k, m, rw);
    }
}

template <class T>
static int launch_step(glgym_handle h, const glgym_step_args* a, const ModelConst<T>& m, const RewardConst<T>& rw,
                       hipStream_t st)
{
    StepArgsT<T> k;
    k.B = a->B; k.ld = a->ld;
    k.x = (T*)a->x; k.u = (T*)a->u; k.action = a->action; k.control = (const T*)a->control;
    k.weather = (const T*)a->weather; k.weather_rows = a->weather_rows;
    k.w_off = a->w_off; k.timestep = a->timestep; k.crop_p = (const T*)a->crop_p; k.N = a->N;
    k.reward = (T*)a->reward; k.info = (T*)a->info; k.done = a->done; k.metrics = a->metrics; k.step_flags = a->step_flags;
    k.dt = T(h->dt); k.n_sub = h->n_sub; k.gasR = T(h->p[39]); k.tCanMin = T(h->p[162]); k.nd = h->nd;
    k.du = h->du;
    for (int j = 0; j < NU; ++j) { k.u_min[j] = h->u_min[j]; k.u_max[j] = h->u_max[j]; }
    // AUTO: verified wherever the control can jump -- raw controls (step_raw_control, the rule-based controller), or an action
    // path whose delta_u_max is wider than the reference's 0.1 (TomatoEnv.yml; base_env.py:74)
    k.pipe = h->variant == GLGYM_ODE_PIPE ? 1 : 0;
    k.window = h->window;
    k.verify = h->verify_mode == GLGYM_VERIFY_ALWAYS || (h->verify_mode == GLGYM_VERIFY_AUTO && (!a->action || h->du > 0.1001f));
    const dim3 grid((a->B + WAVE - 1) / WAVE), block(WAVE), qgrid((4 * a->B + WAVE - 1) / WAVE);
    const bool pipe = h->variant == GLGYM_ODE_PIPE;
    if (pipe && (a->crop_p || h->scheme != GLGYM_SCHEME_RK4)) {
        g_err = "glgym_step: GLGYM_ODE_PIPE supports neither per-env crop parameters nor schemes other than GLGYM_SCHEME_RK4";
        return GLGYM_EINVAL;
    }
    // Layout.  fp64 (the parity configuration): four lanes per environment, always -- coefficient blocks in LDS, no mailbox, no scratch
    // to speak of; it scales with the batch in rounds of 16 384 environments (2.86 ms per round at n_sub 240).  fp32: four lanes per
    // environment while the batch leaves SIMDs idle (B <= 16 384; shared crop parameters, default ODE), one lane per environment
    // beyond.  glgym_set_layout(h, one | quad) overrides for fp32 -- handle state since round 5; the environment variable GLGYM_LAYOUT is
    // only its INITIAL value, read once at glgym_create (changing os.environ afterwards has no effect on an existing handle).
    const bool def = h->use_specialised && std::memcmp(&m, &DefaultConst<T>::value, sizeof m) == 0;
    // verified steps (raw controls) on batches of at most one wavefront per SIMD at EIGHT lanes per environment: the ladder two rungs at a
    // time (glgym_set_ladder_parallel, as glgym_evalF since round 5): two thirds of the latency, identical results and step_flags
    const bool pair = k.verify && h->ladder_parallel && !a->crop_p && (size_t)8 * a->B <= (size_t)WAVE * h->n_simd;
    if constexpr (sizeof(T) == 8) {
        if (h->scheme == GLGYM_SCHEME_RK2) launch_quad_sch<T, GLGYM_SCHEME_RK2>(a, k, m, rw, qgrid, block, st, def, pair);
        else if (h->scheme == GLGYM_SCHEME_RK3) launch_quad_sch<T, GLGYM_SCHEME_RK3>(a, k, m, rw, qgrid, block, st, def, pair);
        else if (h->scheme == GLGYM_SCHEME_LS5) launch_quad_sch<T, GLGYM_SCHEME_LS5>(a, k, m, rw, qgrid, block, st, def, pair);
        else launch_quad_sch<T, GLGYM_SCHEME_RK4>(a, k, m, rw, qgrid, block, st, def, pair);
        HIPCHK(hipGetLastError());
        return GLGYM_OK;
    } else {
        const int layout_env = h->layout;                        // handle state (glgym_set_layout): 0 auto, 1 one, 2 quad
        const bool quad_ok = !pipe && !a->crop_p;
        const int b_small = 4 * h->n_simd * 4;                   // 16 384 on MI355X: one quad-kernel round
        if (quad_ok && (layout_env == 2 || (layout_env == 0 && a->B <= b_small))) {
            if (h->scheme == GLGYM_SCHEME_RK2) launch_quad_sch<T, GLGYM_SCHEME_RK2>(a, k, m, rw, qgrid, block, st, def, pair);
            else if (h->scheme == GLGYM_SCHEME_RK3) launch_quad_sch<T, GLGYM_SCHEME_RK3>(a, k, m, rw, qgrid, block, st, def, pair);
            else if (h->scheme == GLGYM_SCHEME_LS5) launch_quad_sch<T, GLGYM_SCHEME_LS5>(a, k, m, rw, qgrid, block, st, def, pair);
            else launch_quad_sch<T, GLGYM_SCHEME_RK4>(a, k, m, rw, qgrid, block, st, def, pair);
            HIPCHK(hipGetLastError());
            return GLGYM_OK;
        }
        if (pipe) {
            hipLaunchKernelGGL((step_kernel<T, false, false, true>), grid, block, 0, st, k, m, rw);
            HIPCHK(hipGetLastError());
            return GLGYM_OK;
        }
        // The two-waves-per-SIMD build (256 registers; the windows' state in LDS since round 5: profiles/r05_occupancy2.txt) is what batches
        // of at least two wavefronts per SIMD take (131 072 environments on MI355X: 1.08x there, 1.11x from 524 288); glgym_set_occupancy
        // forces either build.  Default parameters and shared crop blocks only (the variants it is instantiated for).
        const bool occ2 = def && !a->crop_p && (h->occupancy == 2 || (h->occupancy == 0 && (size_t)a->B >= (size_t)2 * WAVE * h->n_simd));
        if (h->scheme == GLGYM_SCHEME_RK2) launch_step_sch<GLGYM_SCHEME_RK2>(a, k, m, rw, grid, block, st, def, occ2);
        else if (h->scheme == GLGYM_SCHEME_RK3) launch_step_sch<GLGYM_SCHEME_RK3>(a, k, m, rw, grid, block, st, def, occ2);
        else if (h->scheme == GLGYM_SCHEME_LS5) launch_step_sch<GLGYM_SCHEME_LS5>(a, k, m, rw, grid, block, st, def, occ2);
        else launch_step_sch<GLGYM_SCHEME_RK4>(a, k, m, rw, grid, block, st, def, occ2);
        HIPCHK(hipGetLastError());
        return GLGYM_OK;
    }
}

extern "C" int glgym_step(glgym_handle h, const glgym_step_args* a, void* stream)
{
    if (!h || !a) { g_err = "glgym_step: null handle / arguments"; return GLGYM_EINVAL; }
    if (a->struct_size != (int32_t)sizeof(glgym_step_args)) {     // checked before any pointer member is read
        g_err = "glgym_step: glgym_step_args.struct_size is " + std::to_string(a->struct_size) + ", this library expects " +
                std::to_string(sizeof(glgym_step_args)) + " (ABI " + std::to_string(GLGYM_ABI_VERSION) + "): rebuild against include/glgym.h";
        return GLGYM_EINVAL;
    }
    if (a->B < 1 || a->ld < a->B || !a->x || !a->u || !a->weather || !a->w_off || !a->timestep ||
        !a->reward || !a->done || (!a->action) == (!a->control) || a->weather_rows < 1) {
        g_err = "glgym_step: bad arguments (exactly one of action/control, ld >= B, non-null state/outputs)";
        return GLGYM_EINVAL;
    }
    DeviceGuard dev_guard(h);
    hipStream_t st = (hipStream_t)stream;
    return h->dtype == GLGYM_F32 ? launch_step<float>(h, a, h->mf, h->rf, st)
                                 : launch_step<double>(h, a, h->md, h->rd, st);
}

static const int OBS_MODULE_SIZE[6] = {4, 3, 6, 5, 5, 0};      // observations.py:64,84,102,123,143; forecast = 5 * Np (:168)

template <class T> static int launch_obs(glgym_handle h, const glgym_obs_args* a, hipStream_t st)
{
    ObsArgsT<T> k;
    k.B = a->B; k.ld = a->ld; k.x = (const T*)a->x; k.u = (const T*)a->u; k.weather = (const T*)a->weather;
    k.weather_rows = a->weather_rows; k.w_off = a->w_off; k.timestep = a->timestep; k.start_day = a->start_day;
    k.Np = a->Np; k.obs = a->obs; k.mask = a->mask; k.term_obs = a->term_obs; k.doy_inc = std::fmod(h->dt / 86400.0, 365.0); k.hod_inc = h->dt / 3600.0; k.nd = h->nd;
    int blocks = (a->B + OBS_ROWS - 1) / OBS_ROWS;   // OBS_ROWS env rows per block-iteration
    static const int cap = [] { const char* e = std::getenv("GLGYM_OBS_BLOCKS"); return e ? std::atoi(e) : 4096; }();
    if (blocks > cap) blocks = cap;              // grid-stride beyond that
    for (int m = 0; m < 6; ++m) k.moff[m] = -1;
    k.dim = 0;
    for (int i = 0; i < h->n_obs_modules; ++i) {
        const int m = h->obs_modules[i];
        k.moff[m] = k.dim;
        k.dim += m == GLGYM_OBS_FORECAST ? 5 * a->Np : OBS_MODULE_SIZE[m];
    }
    const size_t lds = (size_t)OBS_ROWS * k.dim * sizeof(float);
    hipLaunchKernelGGL((obs_kernel<T>), dim3(blocks), dim3(256), lds, st, k);
    HIPCHK(hipGetLastError());
    return GLGYM_OK;
}



extern "C" int glgym_weather(glgym_handle h, const glgym_weather_args* a, void* stream)
{
    DeviceGuard dev_guard(h);
    if (!h || !a || a->n_raw < 3 || a->n_out < 1 || a->nd < ND || !a->time || !a->i_glob || !a->t_out || !a->rh || !a->wind ||
        !a->t_sky || !a->out || !a->workspace) {
        g_err = "glgym_weather: bad arguments (>= 3 raw samples, nd >= 10, non-null device pointers)";
        return GLGYM_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const int n = a->n_raw;
    double* W = a->workspace;                       // [10][n] converted columns on the raw grid
    double* D = a->workspace + (size_t)10 * n;      // [10][n] PCHIP slopes
    hipLaunchKernelGGL(weather_convert_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, a->time, a->i_glob, a->t_out,
                       a->rh, a->wind, a->t_sky, a->co2_ppm, W);
    hipLaunchKernelGGL(weather_scan_kernel, dim3(1), dim3(64), 0, st, n, a->time, W);
    hipLaunchKernelGGL(pchip_slopes_kernel, dim3((n + 255) / 256, 10), dim3(256), 0, st, n, 10, a->time, W, D);
    if (a->nd > ND) {                               // extra columns (ODE_pipe rows) are the caller's: start from zero
        HIPCHK(hipMemsetAsync(a->out, 0, (size_t)a->n_out * a->nd * (h->dtype == GLGYM_F32 ? 4 : 8), st));
    }
    const dim3 grid((a->n_out + 255) / 256, 10);
    if (h->dtype == GLGYM_F32)
        hipLaunchKernelGGL((pchip_eval_kernel<float>), grid, dim3(256), 0, st, n, 10, a->n_out, a->nd, a->time, W, D,
                           (float*)a->out);
    else
        hipLaunchKernelGGL((pchip_eval_kernel<double>), grid, dim3(256), 0, st, n, 10, a->n_out, a->nd, a->time, W, D,
                           (double*)a->out);
    HIPCHK(hipGetLastError());
    return GLGYM_OK;
}

extern "C" int glgym_vecnorm(glgym_handle h, const glgym_vecnorm_args* a, void* stream)
{
    DeviceGuard dev_guard(h);
    if (!h || !a || a->B < 1 || a->dim < 1 || !a->obs || !a->obs_out || !a->obs_mean || !a->obs_var || !a->obs_count ||
        !a->workspace || (a->reward && (!a->reward_out || !a->ret_stats || !a->returns))) {
        g_err = "glgym_vecnorm: bad arguments";
        return GLGYM_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const int B = a->B, dim = a->dim;
    double* acc = a->workspace;
    double* acc2 = a->workspace + (size_t)VN_REP * 2 * dim;
    HIPCHK(hipMemsetAsync(a->workspace, 0, ((size_t)VN_REP * 2 * dim + 2) * sizeof(double), st));
    const int do_obs = a->training && a->norm_obs, do_ret = a->training && a->reward != nullptr;
    int col_threads = (dim + WAVE - 1) / WAVE * WAVE;       // one thread per observation column
    if (col_threads > 1024) col_threads = 1024;
    if (do_obs) {
        int blocks = (B + 63) / 64;                 // 64-row strips: <= 1024 blocks x 2*dim fp64 atomics
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(vecnorm_moments_kernel, dim3(blocks), dim3(col_threads), 0, st, a->obs, B, dim, acc);
    }
    if (do_ret) {
        if (h->dtype == GLGYM_F32)
            hipLaunchKernelGGL((vecnorm_returns_kernel<float>), dim3((B + 255) / 256), dim3(256), 0, st,
                               (const float*)a->reward, a->returns, B, a->gamma, acc2);
        else
            hipLaunchKernelGGL((vecnorm_returns_kernel<double>), dim3((B + 255) / 256), dim3(256), 0, st,
                               (const double*)a->reward, a->returns, B, a->gamma, acc2);
    }
    double* scale = acc;                           // the moment accumulators are dead once their column is merged
    if (do_obs || do_ret || a->norm_obs)
        hipLaunchKernelGGL(vecnorm_merge_kernel, dim3(1), dim3(1024), 0, st, a->obs_mean, a->obs_var, a->obs_count, acc, dim, B,
                           a->ret_stats, acc2, do_obs, do_ret, a->epsilon, a->norm_obs ? scale : nullptr);
    const size_t total = (size_t)B * dim;
    if (a->norm_obs) {
        int blocks = (B + 15) / 16;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(vecnorm_apply_kernel, dim3(blocks), dim3(col_threads), 0, st, a->obs, a->obs_out, B, dim,
                           a->obs_mean, scale, a->clip_obs);
    } else if (a->obs_out != a->obs) {
        HIPCHK(hipMemcpyAsync(a->obs_out, a->obs, total * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    if (a->reward) {
        if (h->dtype == GLGYM_F32)
            hipLaunchKernelGGL((vecnorm_reward_kernel<float>), dim3((B + 255) / 256), dim3(256), 0, st,
                               (const float*)a->reward, a->reward_out, a->returns, a->done, B, a->ret_stats, a->epsilon,
                               a->clip_reward, a->norm_reward);
        else
            hipLaunchKernelGGL((vecnorm_reward_kernel<double>), dim3((B + 255) / 256), dim3(256), 0, st,
                               (const double*)a->reward, a->reward_out, a->returns, a->done, B, a->ret_stats, a->epsilon,
                               a->clip_reward, a->norm_reward);
    }
    HIPCHK(hipGetLastError());
    return GLGYM_OK;
}

extern "C" {

int glgym_obs(glgym_handle h, const glgym_obs_args* a, void* stream)
{
    DeviceGuard dev_guard(h);
    if (!h || !a || a->B < 1 || a->ld < a->B || !a->x || !a->u || !a->weather || !a->w_off || !a->timestep ||
        !a->start_day || !a->obs || a->Np < 0 || a->Np > OBS_MAX_NP) {
        g_err = "glgym_obs: bad arguments (null pointer, ld < B, or Np outside 0..128)";
        return GLGYM_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    return h->dtype == GLGYM_F32 ? launch_obs<float>(h, a, st) : launch_obs<double>(h, a, st);
}

int glgym_set_obs_modules(glgym_handle h, const int32_t* modules, int n)
{
    if (!h || !modules || n < 1 || n > 6) { g_err = "glgym_set_obs_modules: 1..6 module ids expected"; return GLGYM_EINVAL; }
    int seen = 0;
    for (int i = 0; i < n; ++i) {
        if (modules[i] < 0 || modules[i] > 5 || (seen >> modules[i] & 1)) {
            g_err = "glgym_set_obs_modules: ids must be distinct GLGYM_OBS_* values";
            return GLGYM_EINVAL;
        }
        seen |= 1 << modules[i];
    }
    for (int i = 0; i < n; ++i) h->obs_modules[i] = modules[i];
    h->n_obs_modules = n;
    return GLGYM_OK;
}

int glgym_obs_dim(glgym_handle h, int Np)
{
    if (!h || Np < 0 || Np > OBS_MAX_NP) return GLGYM_EINVAL;
    int dim = 0;
    for (int i = 0; i < h->n_obs_modules; ++i)
        dim += h->obs_modules[i] == GLGYM_OBS_FORECAST ? 5 * Np : OBS_MODULE_SIZE[h->obs_modules[i]];
    return dim;
}

int glgym_reset(glgym_handle h, const glgym_reset_args* a, void* stream)
{
    DeviceGuard dev_guard(h);
    if (!h || !a || a->B < 1 || a->ld < a->B || !a->x || !a->u || !a->timestep || !a->weather || !a->w_off) {
        g_err = "glgym_reset: bad arguments";
        return GLGYM_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((a->B + 255) / 256), block(256);
    if (h->dtype == GLGYM_F32)
        hipLaunchKernelGGL((reset_kernel<float>), grid, block, 0, st, a->B, a->ld, a->mask, (float*)a->x, (float*)a->u,
                           a->timestep, (const float*)a->weather, a->weather_rows, a->w_off, a->start_rows, a->start_days,
                           a->n_starts, a->start_day, a->episode, (unsigned long long)a->seed, h->nd);
    else
        hipLaunchKernelGGL((reset_kernel<double>), grid, block, 0, st, a->B, a->ld, a->mask, (double*)a->x,
                           (double*)a->u, a->timestep, (const double*)a->weather, a->weather_rows, a->w_off,
                           a->start_rows, a->start_days, a->n_starts, a->start_day, a->episode,
                           (unsigned long long)a->seed, h->nd);
    HIPCHK(hipGetLastError());
    return GLGYM_OK;
}

int glgym_crop_noise(glgym_handle h, void* crop_p, int B, int ld, double scale, uint64_t seed, uint64_t draw_index,
                     void* stream)
{
    DeviceGuard dev_guard(h);
    if (!h || !crop_p || B < 1 || ld < B) return GLGYM_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((B + 255) / 256), block(256);
    if (h->dtype == GLGYM_F32)
        hipLaunchKernelGGL((crop_noise_kernel<float>), grid, block, 0, st, (float*)crop_p, B, ld, h->p0_crop_dev,
                           (float)scale, (unsigned long long)seed, (unsigned long long)draw_index);
    else
        hipLaunchKernelGGL((crop_noise_kernel<double>), grid, block, 0, st, (double*)crop_p, B, ld, h->p0_crop_dev,
                           (float)scale, (unsigned long long)seed, (unsigned long long)draw_index);
    HIPCHK(hipGetLastError());
    return GLGYM_OK;
}

int glgym_rule_based(glgym_handle h, const glgym_rule_cfg* cfg, const glgym_rule_args* a, void* stream)
{
    DeviceGuard dev_guard(h);
    if (!h || !cfg || !a || a->B < 1 || a->ld < a->B || !a->x || !a->weather || !a->w_off || !a->timestep || !a->control ||
        a->weather_rows < 1 || (!a->start_day && !a->doy) || cfg->vent_heat_Pband == 0 || cfg->vent_rh_Pband == 0 ||
        cfg->vent_cold_Pband == 0 || cfg->thScrPband == 0 || cfg->thScrRhPband == 0 || cfg->tHeatBand == 0 ||
        cfg->co2Band == 0) {
        g_err = "glgym_rule_based: bad arguments (null pointer, ld < B, or a zero proportional band)";
        return GLGYM_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((a->B + 255) / 256), block(256);
    const double doy_inc = std::fmod(h->dt / 86400.0, 365.0), hod_inc = h->dt / 3600.0;
    if (h->dtype == GLGYM_F32)
        hipLaunchKernelGGL((rule_based_kernel<float>), grid, block, 0, st, *cfg, a->B, a->ld, (const float*)a->x,
                           (const float*)a->weather, a->weather_rows, a->w_off, a->timestep, a->start_day, a->hour, a->doy,
                           doy_inc, hod_inc, (float*)a->control, h->nd);
    else
        hipLaunchKernelGGL((rule_based_kernel<double>), grid, block, 0, st, *cfg, a->B, a->ld, (const double*)a->x,
                           (const double*)a->weather, a->weather_rows, a->w_off, a->timestep, a->start_day, a->hour,
                           a->doy, doy_inc, hod_inc, (double*)a->control, h->nd);
    HIPCHK(hipGetLastError());
    return GLGYM_OK;
}

int glgym_timer_start(glgym_handle h, void* stream)
{
    DeviceGuard dev_guard(h);
    if (!h) return GLGYM_EINVAL;
    HIPCHK(hipEventRecord(h->ev0, (hipStream_t)stream));
    return GLGYM_OK;
}

int glgym_timer_stop(glgym_handle h, void* stream, float* elapsed_ms)
{
    DeviceGuard dev_guard(h);
    if (!h || !elapsed_ms) return GLGYM_EINVAL;
    HIPCHK(hipEventRecord(h->ev1, (hipStream_t)stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    HIPCHK(hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
    return GLGYM_OK;
}

}  // extern "C"
