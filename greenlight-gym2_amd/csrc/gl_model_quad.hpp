// gl_model_quad.hpp -- the north-star layout: SEVERAL LANES PER ENVIRONMENT (four), for batches that leave SIMDs idle.
//
// gl_model.hpp integrates one environment per lane: 28 states, ~370 vector instructions per stage, and a lone wavefront is
// issued one of them only every ~5 cycles -- 1 ms per env-step whether the batch holds 8 environments or 65 536.  Below 16 384
// environments most of the chip's 1 024 SIMDs hold no wave at all.  Here a QUAD of lanes integrates one environment (16 per
// wavefront); the stage's instruction stream shrinks to what one lane of the quad has to do:
//   lane r owns one PAIR of radiating surfaces -- the pairs gl_model.hpp already packs into v_pk_* registers -- r = 0: (tCan,
//   tPipe), 1: (tFlr, tLamp), 2: (tThScr, tBlScr), 3: (tCovIn, tCovE) -- and a quarter of the slow / constant-rate states; the six
//   air-side states (co2Air, co2Top, tAir, tTop, vpAir, vpTop) are carried redundantly by all four lanes.  Every surface is one
//   row of the same algebra with per-lane coefficients (LaneK):
//       net_i = src_i + sum_j C_ij (q_j - q_i) + C_i,sky (q_sky - q_i)          long wave, q = (T + 273.15)^4
//               + cA_i |dA_i|^nA_i dA_i          exchange with its air node A (air | top | outside),  dA = T_A - T_i
//               + L wet_i hecA_i gate(vp_A - satVp(T_i))                        condensation on the wet surfaces
//               - cB_i |dB_i|^(1/3) dB_i         second exchange (screens -> top compartment),        dB = T_i - tTop
//   (the conduction inside the cover pair is not a row term: the integrator treats it exactly -- rk_delta_quad, COVEXP of gl_model.hpp)
//               - L mvCanAir                     transpiration (canopy)
//   Lanes talk through DPP quad_perm only (a full crossbar inside four lanes, no LDS): 8 moves gather the eight q's per stage,
//   4 x 2 DPP adds reduce the four sums the air / top balances need; per window ~20 more broadcast the inputs of tier 2b and
//   of the rate bound.
// Same scheme as rk_delta / rk4_delta_guarded of gl_model.hpp, decision for decision (windows, tier 2b at the predicted
// midpoint, exact harvest sub-flow, wet surfaces as differences to their air node, rate bound -> sub-steps per window, movement
// limiter, embedded error estimate, branch invariant, closing evaluation, step-doubling ladder): the CPU checker's restatement of
// that scheme is the reference for both layouts (tests/).  Every member of the scheme family (RK4 / three-stage / midpoint, all
// with the exponential cover conduction), both ODE variants, shared or per-environment crop constants, interlights on or off: in
// fp64 this layout is the only integrator on the device (round 4); in fp32 it serves the small batches.
// Measured (tools/lanes_stage_proto.hip, profiles/r04_lanes_stage_proto.txt): the bare RK4 chain runs 1.47x (fp32) / 1.57x (fp64)
// the env-steps per second of the one-lane layout for B <= 4 096 and 0.73x at B = 65 536 -- hence the dispatch by batch size.  As
// PRODUCT kernels the gain in fp32 is 1.15x at B = 8 ... 1.18x at 16 384 (profiles/r04_small_batch_rate_fp32.txt): a fifth of this
// kernel is window-level work every lane repeats (tier 2b, rate bound, estimate: 1.9 us per window); in fp64 there
// is no one-lane kernel to compare with any more (the last one took 7.8 ms per 24 576 environments against this layout's 5.7).
// Eight lanes per environment were measured too: 0.94x of this layout in fp32, 1.16x in fp64.
#pragma once
#include "gl_model.hpp"

#if defined(__HIPCC__)
namespace glm {

// ---- pairs: float -> one v_pk_* register pair, double -> two registers ------------------------------------------------------
struct gq_d2 { double x, y; };
__device__ __forceinline__ gq_d2 operator+(gq_d2 a, gq_d2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ gq_d2 operator-(gq_d2 a, gq_d2 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ gq_d2 operator*(gq_d2 a, gq_d2 b) { return {a.x * b.x, a.y * b.y}; }
typedef float gq_f2 __attribute__((ext_vector_type(2)));
template <class T> struct GqPair;
template <> struct GqPair<float> { typedef gq_f2 type; };
template <> struct GqPair<double> { typedef gq_d2 type; };
template <class T> using P2 = typename GqPair<T>::type;
template <class T> __device__ __forceinline__ P2<T> gq_mk(T a, T b) { P2<T> r; r.x = a; r.y = b; return r; }
template <class T> __device__ __forceinline__ P2<T> gq_sp(T a) { return gq_mk<T>(a, a); }

// ---- DPP inside a quad --------------------------------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ int gq_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ float gq_dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ double gq_dpp(double v)
{
    // two 32-bit DPP moves on the halves, kept apart: hipcc 7.2 otherwise fuses them into a 64-bit DPP move, which gfx950
    // implements for row_newbcast only -- quad_perm came back with garbage (profiles/r03_lanes_stage_proto.txt)
    unsigned lo = (unsigned)__builtin_bit_cast(unsigned long long, v), hi = (unsigned)(__builtin_bit_cast(unsigned long long, v) >> 32);
    asm volatile("" : "+v"(lo));
    asm volatile("" : "+v"(hi));
    unsigned rl = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, 0xf, 0xf, true);
    asm volatile("" : "+v"(rl));
    unsigned rh = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, 0xf, 0xf, true);
    asm volatile("" : "+v"(rh));
    return __builtin_bit_cast(double, ((unsigned long long)rh << 32) | rl);
}
template <int S, class T> __device__ __forceinline__ T gq_bcast(T v) { return gq_dpp<S * 0x55>(v); }       // lane S of the quad
template <class T> __device__ __forceinline__ T gq_sum(T v) { v += gq_dpp<0xB1>(v); v += gq_dpp<0x4E>(v); return v; }
template <class T> __device__ __forceinline__ T gq_max(T v)
{
    v = Math<T>::max(v, gq_dpp<0xB1>(v)); v = Math<T>::max(v, gq_dpp<0x4E>(v)); return v;
}
__device__ __forceinline__ int gq_or(int v) { v |= gq_dpp<0xB1>(v); v |= gq_dpp<0x4E>(v); return v; }

// ---- who owns what -------------------------------------------------------------------------------------------------------------
// pair of lane `role` (component c); its four "other" states; the six shared ones.  In the integrator's coordinates slots 5, 7, 20
// are the differences tTop - tCovIn, tAir - tThScr, tAir - tBlScr (gl_model.hpp rhs_fast<WETDIFF>).
__host__ __device__ constexpr int gq_pair_ix(int role, int c) { return role == 0 ? (c ? 9 : 4) : role == 1 ? (c ? 17 : 8) : role == 2 ? (c ? 20 : 7) : (c ? 6 : 5); }
__host__ __device__ constexpr int gq_other_ix(int role, int j)
{
    return role == 0 ? (j == 0 ? 21 : j == 1 ? 26 : j == 2 ? 10 : 11) : role == 1 ? (j == 0 ? 12 : j == 1 ? 13 : j == 2 ? 14 : 19)
           : role == 2 ? (j == 0 ? 22 : j == 1 ? 23 : j == 2 ? 24 : 25) : (j == 0 ? 18 : 27);   // lane 3: slots 2, 3 unused
}
__host__ __device__ constexpr int gq_sh_ix(int i) { return i < 4 ? i : 11 + i; }                 // 0 1 2 3 15 16

template <class T> struct QVec { P2<T> p; T sh[6]; T o[4]; };

// ---- per-lane coefficients: functions of the lane's role, the env-step's StepCoef and the window's SlowCoef ---------------------
template <class T> struct LaneK {
    P2<T> cA, nA, cA2, nA2, sgA;   // exchange with node A: c |sg dA + eps|^n dA; the "2" set applies where dA < 0 (the floor's two regimes)
    P2<T> cB;                      // second exchange (to the top compartment): the two screens
    P2<T> src, iCap, wetC, mAir;
    T mTopX, trKx;                 // x components of the "exchanges with the top compartment" mask and of the transpiration coefficient (y: always 0)
    P2<T> firX[4], firY[4];        // C[own x|y][lane s .x] and C[own x|y][lane s .y]
    P2<T> cSky;
    P2<T> iC;                      // interlight long-wave coefficients of the pair (zero power in the reference; geometry may be present)
    T ro[4];                       // rates of the lane's constant-rate states (tier 2b)
    // ETD coefficients of the pair's y component for the sub-step in use (rk_delta_quad; part of this record so that the fp64
    // build keeps them in LDS with the rest of it): cover lane a = 2 cCovCond / capCov, every other lane a = 0 = the classical scheme.
    // RK4 / three-stage / midpoint: ec; the five-stage 2N scheme: lc (same storage)
    union { EtdCoef<T> ec; LsCoef<T> lc; };
};

template <class T>
__device__ __forceinline__ void gq_make_lane(int role, const StepCoef<T>& s, const ModelConst<T>& m, const SlowCoef<T>& q, LaneK<T>& K)
{
    const T z = T(0), one = T(1), third = T(1.0 / 3.0), L64 = T(6.4e-9);
    auto pk = [&](T a0, T a1, T a2, T a3) { return role == 0 ? a0 : role == 1 ? a1 : role == 2 ? a2 : a3; };
    // rows of the symmetric long-wave matrix over (Can, Pipe | Flr, Lamp | ThScr, BlScr | CovIn, CovE) (FirBlock::run), by source lane
    //                      own x = Can            Flr             ThScr           CovIn         own y = Pipe         Lamp            BlScr           CovE
    K.firX[0] = gq_mk<T>(pk(z, q.kCanFlr, q.kCanThScr, q.kCanCovIn), pk(q.kPipeCan, q.kLampCan, q.kCanBlScr, z));                 // source Can
    K.firY[0] = gq_mk<T>(pk(q.kPipeCan, m.fPipeFlr, q.kPipeThScr, q.kPipeCovIn), pk(z, q.kLampPipe, q.kPipeBlScr, z));            // source Pipe
    K.firX[1] = gq_mk<T>(pk(q.kCanFlr, z, q.kFlrThScr, q.kFlrCovIn), pk(m.fPipeFlr, q.kLampFlr, q.kFlrBlScr, z));                 // source Flr
    K.firY[1] = gq_mk<T>(pk(q.kLampCan, q.kLampFlr, s.cLampThScr, s.cLampCovIn), pk(q.kLampPipe, z, s.cLampBlScr, z));            // source Lamp
    K.firX[2] = gq_mk<T>(pk(q.kCanThScr, q.kFlrThScr, z, s.cThScrCovIn), pk(q.kPipeThScr, s.cLampThScr, s.cBlScrThScr, z));       // source ThScr
    K.firY[2] = gq_mk<T>(pk(q.kCanBlScr, q.kFlrBlScr, s.cBlScrThScr, s.cBlScrCovIn), pk(q.kPipeBlScr, s.cLampBlScr, z, z));       // source BlScr
    K.firX[3] = gq_mk<T>(pk(q.kCanCovIn, q.kFlrCovIn, s.cThScrCovIn, z), pk(q.kPipeCovIn, s.cLampCovIn, s.cBlScrCovIn, z));       // source CovIn
    K.firY[3] = gq_mk<T>(z, z);                                                                                                   // source CovE: sky only
    K.cSky = gq_mk<T>(pk(q.kCanSky, q.kFlrSky, s.cThScrSky, z), pk(q.kPipeSky, s.cLampSky, s.cBlScrSky, m.fCovESky));
    K.cA = gq_mk<T>(pk(q.hCanAirK, T(1.3), s.hTh, m.cTopCov), pk(m.cPipeAir, m.cLampAir, s.hBl, s.covOutK));
    K.cA2 = gq_mk<T>(pk(q.hCanAirK, T(1.7), s.hTh, m.cTopCov), K.cA.y);
    K.nA = gq_mk<T>(pk(z, T(0.25), third, third), pk(T(0.32), z, third, z));
    K.nA2 = gq_mk<T>(pk(z, third, third, third), K.nA.y);
    K.sgA = gq_mk<T>(one, pk(-one, one, one, one));      // the product's orientation of |dT + 1e-10| (the floor: by the sign of dA)
    K.cB = gq_mk<T>(pk(z, z, s.hTh, z), pk(z, z, s.hBl, z));
    K.src = gq_mk<T>(pk(q.swCan + q.rGroPipeCan, q.swFlr - q.hFlrSo1, z, z), pk(s.hBoilPipe, s.lampNet, z, s.sunCovE));
    K.iCap = gq_mk<T>(pk(q.iCapCan, m.iCapFlr, m.iCapThScr, m.iCapCov), pk(m.iCapPipe, m.iCapLamp, m.iCapBlScr, m.iCapCov));
    K.wetC = gq_mk<T>(pk(z, z, L64, L64), pk(z, z, L64, z));
    K.mAir = gq_mk<T>(pk(one, one, one, z), pk(one, one, one, z));
    K.mTopX = pk(z, z, z, one);
    K.trKx = pk(q.mvCanK, z, z, z);
    K.iC = gq_mk<T>(pk(q.iCan, q.iFlr, q.iThScr, q.iCovIn), pk(q.iPipe, q.iLamp, q.iBlScr, z));
    K.ro[0] = pk(z, q.dSo3, q.dBuf, z); K.ro[1] = pk(z, q.dSo4, q.dLeaf, z);
    K.ro[2] = pk(q.dSo1, q.dSo5, q.dStem, z); K.ro[3] = pk(q.dSo2, q.dGro, q.dFruit, z);
}

// ---- the same record built in two parts (the register build, fp32): the selects by role cost 365 instructions per window when the
// whole record is rebuilt there (measured: 0.75 of the 2.6 us a window costs at B = 8).  Only the canopy-dependent part changes
// from window to window, and all of its long-wave entries are  c x aCan,  c x gap  or  c  (slow_coef): with the three coefficient
// sets (f0, fa, fg) selected ONCE per env-step an entry is  fa aCan + (fg gap + f0)  -- two packed FMAs, exact (one of the three
// coefficients is non-zero, so each entry is the one rounded product slow_coef forms) -- and the rest of the record is written once.
template <class T> struct LaneStep {
    P2<T> f0[8], fa[8], fg[8];     // entries 0..3 firX[0..3], 4..6 firY[0..2], 7 cSky
    T cAx, iCapx;                  // the step-level x components lane 0 overrides per window (canopy: q.hCanAirK, q.iCapCan)
};
template <class T>
__device__ __forceinline__ void gq_make_lane_step(int role, const StepCoef<T>& s, const ModelConst<T>& m, LaneK<T>& K, LaneStep<T>& LS)
{
    const T z = T(0), one = T(1), third = T(1.0 / 3.0), L64 = T(6.4e-9);
    auto pk = [&](T a0, T a1, T a2, T a3) { return role == 0 ? a0 : role == 1 ? a1 : role == 2 ? a2 : a3; };
    // kind of an entry: 'C' constant, 'A' times aCan, 'G' times gap.  X(...) / Y(...) fill the x / y component of entry e from the four
    // roles' (kind, coefficient) pairs
    struct E { int kind; T c; };
    auto C = [&](T c) { return E{0, c}; };
    auto A = [&](T c) { return E{1, c}; };
    auto G = [&](T c) { return E{2, c}; };
    auto put = [&](int e, bool ycomp, E e0, E e1, E e2, E e3) {
        const int kind = role == 0 ? e0.kind : role == 1 ? e1.kind : role == 2 ? e2.kind : e3.kind;
        const T c = pk(e0.c, e1.c, e2.c, e3.c);
        const T v0 = kind == 0 ? c : z, va = kind == 1 ? c : z, vg = kind == 2 ? c : z;
        if (ycomp) { LS.f0[e].y = v0; LS.fa[e].y = va; LS.fg[e].y = vg; }
        else { LS.f0[e].x = v0; LS.fa[e].x = va; LS.fg[e].x = vg; }
    };
    // rows of the symmetric long-wave matrix (gq_make_lane), by source lane;  q.kCanX = c aCan,  q.kPipeX / q.kFlrX / q.kLamp{Flr,Pipe} = c gap
    put(0, false, C(z), A(m.fCanFlr_a), A(s.cCanThScr), A(s.cCanCovIn));            put(0, true, A(m.fPipeCan_a), A(m.fLampCan_a), A(s.cCanBlScr), C(z));      // source Can
    put(4, false, A(m.fPipeCan_a), C(m.fPipeFlr), G(s.cPipeThScr), G(s.cPipeCovIn)); put(4, true, C(z), G(m.fLampPipe_g), G(s.cPipeBlScr), C(z));                // source Pipe
    put(1, false, A(m.fCanFlr_a), C(z), G(s.cFlrThScr), G(s.cFlrCovIn));            put(1, true, C(m.fPipeFlr), G(m.fLampFlr_g), G(s.cFlrBlScr), C(z));        // source Flr
    put(5, false, A(m.fLampCan_a), G(m.fLampFlr_g), C(s.cLampThScr), C(s.cLampCovIn)); put(5, true, G(m.fLampPipe_g), C(z), C(s.cLampBlScr), C(z));            // source Lamp
    put(2, false, A(s.cCanThScr), G(s.cFlrThScr), C(z), C(s.cThScrCovIn));          put(2, true, G(s.cPipeThScr), C(s.cLampThScr), C(s.cBlScrThScr), C(z));     // source ThScr
    put(6, false, A(s.cCanBlScr), G(s.cFlrBlScr), C(s.cBlScrThScr), C(s.cBlScrCovIn)); put(6, true, G(s.cPipeBlScr), C(s.cLampBlScr), C(z), C(z));             // source BlScr
    put(3, false, A(s.cCanCovIn), G(s.cFlrCovIn), C(s.cThScrCovIn), C(z));          put(3, true, G(s.cPipeCovIn), C(s.cLampCovIn), C(s.cBlScrCovIn), C(z));     // source CovIn
    put(7, false, A(s.cCanSky), G(s.cFlrSky), C(s.cThScrSky), C(z));                put(7, true, G(s.cPipeSky), C(s.cLampSky), C(s.cBlScrSky), C(m.fCovESky));  // sky
    K.firY[3] = gq_mk<T>(z, z);                                                     // source CovE: sky only
    // the entries no window changes
    LS.cAx = pk(z, T(1.3), s.hTh, m.cTopCov);
    K.cA = gq_mk<T>(LS.cAx, pk(m.cPipeAir, m.cLampAir, s.hBl, s.covOutK));
    K.cA2 = gq_mk<T>(pk(z, T(1.7), s.hTh, m.cTopCov), K.cA.y);
    K.nA = gq_mk<T>(pk(z, T(0.25), third, third), pk(T(0.32), z, third, z));
    K.nA2 = gq_mk<T>(pk(z, third, third, third), K.nA.y);
    K.sgA = gq_mk<T>(one, pk(-one, one, one, one));
    K.cB = gq_mk<T>(pk(z, z, s.hTh, z), pk(z, z, s.hBl, z));
    LS.iCapx = pk(z, m.iCapFlr, m.iCapThScr, m.iCapCov);
    K.iCap = gq_mk<T>(LS.iCapx, pk(m.iCapPipe, m.iCapLamp, m.iCapBlScr, m.iCapCov));
    K.wetC = gq_mk<T>(pk(z, z, L64, L64), pk(z, z, L64, z));
    K.mAir = gq_mk<T>(pk(one, one, one, z), pk(one, one, one, z));
    K.mTopX = pk(z, z, z, one);
    K.src = gq_mk<T>(z, pk(s.hBoilPipe, s.lampNet, z, s.sunCovE));
    K.trKx = z;
    K.iC = gq_mk<T>(z, z);
}
template <class T>
__device__ __forceinline__ void gq_make_lane_win(int role, const StepCoef<T>& s, const ModelConst<T>& m, const SlowCoef<T>& q, LaneK<T>& K,
                                                 const LaneStep<T>& LS)
{
    const T z = T(0);
    auto pk = [&](T a0, T a1, T a2, T a3) { return role == 0 ? a0 : role == 1 ? a1 : role == 2 ? a2 : a3; };
    const bool lane0 = role == 0;
    const P2<T> aC = gq_sp<T>(q.aCan), gp = gq_sp<T>(q.gap);
    auto row = [&](int e) { return LS.fa[e] * aC + (LS.fg[e] * gp + LS.f0[e]); };
    K.firX[0] = row(0); K.firX[1] = row(1); K.firX[2] = row(2); K.firX[3] = row(3);
    K.firY[0] = row(4); K.firY[1] = row(5); K.firY[2] = row(6);
    K.cSky = row(7);
    K.cA = gq_mk<T>(lane0 ? q.hCanAirK : LS.cAx, K.cA.y);
    K.cA2 = gq_mk<T>(lane0 ? q.hCanAirK : K.cA2.x, K.cA2.y);
    K.iCap = gq_mk<T>(lane0 ? q.iCapCan : LS.iCapx, K.iCap.y);
    K.src = gq_mk<T>(pk(q.swCan + q.rGroPipeCan, q.swFlr - q.hFlrSo1, z, z), K.src.y);
    K.trKx = lane0 ? q.mvCanK : z;
    if (m.intLampActive) K.iC = gq_mk<T>(pk(q.iCan, q.iFlr, q.iThScr, q.iCovIn), pk(q.iPipe, q.iLamp, q.iBlScr, z));
    K.ro[0] = pk(z, q.dSo3, q.dBuf, z); K.ro[1] = pk(z, q.dSo4, q.dLeaf, z);
    K.ro[2] = pk(q.dSo1, q.dSo5, q.dStem, z); K.ro[3] = pk(q.dSo2, q.dGro, q.dFruit, z);
}

// what the rate bound / branch invariant need from a stage, per lane (gl_model.hpp rhs_fast<RATES>)
template <class T> struct QRates { P2<T> hecA, hecB, sv, rr, g, Tsurf; T fScrAbs, fRoofAbs, tTopK; };

// ---- one stage.  y: the lane's states in the integrator's coordinates; k: derivatives in the same coordinates ------------------
// PIPE: the reference's ODE_pipe (ode.hpp:126-263) is compiled in and selected at RUN TIME by s.pipeOde -- lane 0's pipe follows the
// measured temperature while tracking (s.pipeTrack), the grow pipe (lane 1, fourth "other") stands still (rhs_fast<PIPE>).  Run time,
// not another instantiation: the fp64 builds of this layout sit at the register limit, and every additional variant is one more
// binary hipcc 7.2 can get wrong (a separate ODE_pipe build computed a wrong second soil layer); this way ODE_pipe runs the very
// kernel the default variant's rounding-level parity tests cover.
template <class T, bool RATES, bool PIPE = false>
__device__ __forceinline__ void gq_stage(int role, const QVec<T>& y, const LaneK<T>& K, const StepCoef<T>& s, const ModelConst<T>& m,
                                         const SlowCoef<T>& q, QVec<T>& k, QRates<T>* R)
{
    using M = Math<T>;
    const T one = T(1), eps = T(1e-10), c2k = Kelvin<T>::c2k(), third = T(1.0 / 3.0);
    const T co2Air = y.sh[0], co2Top = y.sh[1], tAir = y.sh[2], tTop = y.sh[3], vpAir = y.sh[4], vpTop = y.sh[5];
    const bool cov = role == 3, scr = role == 2, lane0 = role == 0;
    // physical temperatures of the pair: lanes 2 / 3 carry differences to their air node (rhs_fast<WETDIFF>)
    // (lane 3, round 4: y.p.y = w = tCovIn - tCovE, rhs_fast<COVEXP>)
    const T TpX = scr ? tAir - y.p.x : cov ? tTop - y.p.x : y.p.x;
    const P2<T> Tp = gq_mk<T>(TpX, scr ? tAir - y.p.y : cov ? TpX - y.p.y : y.p.y);
    // ---- long wave: gather the eight q's, 4 source lanes x 2 packed terms
    P2<T> qp;                                                        // Q4<T>::of on the pair (fp32: the offset form, gl_model.hpp Q4)
    if constexpr (sizeof(T) == 4) qp = Tp * (gq_sp<T>(Q4<float>::C3) + Tp * (gq_sp<T>(Q4<float>::C2) + Tp * (gq_sp<T>(Q4<float>::C1) + Tp)));
    else { const P2<T> kk = Tp + gq_sp<T>(c2k), k2 = kk * kk; qp = k2 * k2; }
    P2<T> fir = K.cSky * (gq_sp<T>(s.qSky) - qp);
    {
        const T q0x = gq_bcast<0>(qp.x), q0y = gq_bcast<0>(qp.y), q1x = gq_bcast<1>(qp.x), q1y = gq_bcast<1>(qp.y);
        const T q2x = gq_bcast<2>(qp.x), q2y = gq_bcast<2>(qp.y), q3x = gq_bcast<3>(qp.x);
        fir = fir + K.firX[0] * (gq_sp<T>(q0x) - qp) + K.firY[0] * (gq_sp<T>(q0y) - qp);
        fir = fir + K.firX[1] * (gq_sp<T>(q1x) - qp) + K.firY[1] * (gq_sp<T>(q1y) - qp);
        fir = fir + K.firX[2] * (gq_sp<T>(q2x) - qp) + K.firY[2] * (gq_sp<T>(q2y) - qp);
        fir = fir + K.firX[3] * (gq_sp<T>(q3x) - qp);
    }
    // ---- interlights: geometry only (aux_states.hpp:261 hard-wires their power to zero); every term is exactly 0 with the default block
    T dInt = T(0), hIntAir = T(0);
    if (m.intLampActive) {
        const T tInt = gq_bcast<3>(y.o[0]);                          // x18 lives on lane 3
        const T qInt = Q4<T>::of(tInt);
        const P2<T> iTo = K.iC * (gq_sp<T>(qInt) - qp);              // into the pair's surfaces
        fir = fir + iTo;
        hIntAir = m.cIntLampAir * (tInt - tAir);
        dInt = m.iCapIntLamp * (-hIntAir - q.iSky * (qInt - s.qSky) - gq_sum(iTo.x + iTo.y));
    }
    // ---- exchange with node A (dA = T_A - T_i: exactly the carried difference on the wet lanes)
    const P2<T> dA = gq_mk<T>(scr || cov ? y.p.x : tAir - Tp.x, scr ? y.p.y : (cov ? s.tOut : tAir) - Tp.y);
    const bool negx = dA.x < T(0), negy = dA.y < T(0);
    const P2<T> nA = gq_mk<T>(negx ? K.nA2.x : K.nA.x, negy ? K.nA2.y : K.nA.y), cA = gq_mk<T>(negx ? K.cA2.x : K.cA.x, negy ? K.cA2.y : K.cA.y);
    const T sgx = role == 1 ? (negx ? -one : one) : K.sgA.x;
    // fp64 (a power is 57 instructions there, 3 in fp32): lane 0's x law is linear -- the canopy, exponent 0, its power slot would
    // evaluate x^0 = 1 -- so that slot computes the air side's |tAir - tTop|^0.66 for the whole quad instead (one power per stage less).
    // (Round 6 tried the rest of that idea and measured nothing: of the quad's 32 transcendental slots 15 are needed -- two y laws are linear,
    // only the screen lane has a second exchange, four surfaces need a saturation pressure and three a gate -- so the two cube roots went to
    // the spare y power slots of lanes 1 / 3 and the blackout screen's saturation pressure + gate to the dry floor lane, nine 64-bit DPP
    // broadcasts in all.  Static vector instructions of the sub-step loop 4 436 -> 4 424, 2.024 -> 2.016 us per stage at B = 8: a 64-bit
    // quad broadcast (two fenced 32-bit DPP moves + the selects around it) costs what a 20-instruction cube root does.  Reverted.)
    constexpr bool SHARE_POW = sizeof(T) == 8;
    const T powArgX = (SHARE_POW && lane0) ? M::abs(tAir - tTop + eps) : M::abs(sgx * dA.x + eps);
    const T powExpX = (SHARE_POW && lane0) ? T(0.66) : nA.x;
    const T powX = M::powa(powArgX, powExpX);
    const P2<T> hecA = cA * gq_mk<T>((SHARE_POW && lane0) ? one : powX, M::powa(M::abs(K.sgA.y * dA.y + eps), nA.y));
    const P2<T> fluxA = hecA * dA;                                   // into the surface
    // ---- second exchange: screens -> top compartment
    const P2<T> dB = Tp - gq_sp<T>(tTop);
    const P2<T> hecB = K.cB * gq_mk<T>(M::cbrta(M::abs(dB.x + eps)), M::cbrta(M::abs(dB.y + eps)));
    const P2<T> fluxB = hecB * dB;
    // ---- saturation pressure, condensation gate, transpiration
    const P2<T> rr = gq_mk<T>(M::rcpn(Tp.x + T(238.3)), M::rcpn(Tp.y + T(238.3)));
    const P2<T> sv = gq_sp<T>(T(610.78)) * gq_mk<T>(M::expk(T(17.2694), Tp.x * rr.x), M::expk(T(17.2694), Tp.y * rr.y));
    const P2<T> dv = gq_mk<T>(cov ? vpTop : vpAir, vpAir) - sv;
    const P2<T> g = dv * gq_mk<T>(M::rcp(one + M::expk(T(-0.1), dv.x)), M::rcp(one + M::expk(T(-0.1), dv.y)));
    const P2<T> mv = K.wetC * hecA * g;                              // vapour condensing on the surface
    const T vpd = sv.x - vpAir;                                      // lane 0: x = canopy
    const T co2Dev = m.etaMgPpm * co2Air - T(200);
    const T rfCo2 = M::min(T(1.5), one + s.cEvap3 * (co2Dev * co2Dev));
    const T rfVp = M::min(T(5.8), one + s.cEvap4 * (vpd * vpd));
    const T mvCan = vpd * K.trKx * M::rcpn(m.rB + s.rSK * rfCo2 * rfVp);
    // ---- the pair's balances
    const T L = m.latent;
    const P2<T> net = K.src + fir + fluxA + gq_sp<T>(L) * mv - fluxB + gq_mk<T>(-(L * mvCan), T(0));
    const P2<T> dTp = K.iCap * net;
    // ---- sums the air / top balances need
    const P2<T> fa = fluxA * K.mAir, ma = mv * K.mAir;
    const T ftx = fluxA.x * K.mTopX, mtx = mv.x * K.mTopX;       // (the y components of that mask are 0 on every lane)
    const T sHeatAir = gq_sum(-(fa.x + fa.y));
    const T sHeatTop = gq_sum((fluxB.x + fluxB.y) - ftx);
    const T sVapAir = gq_sum(mvCan - (ma.x + ma.y));
    const T sVapTop = gq_sum(-mtx);
    const T tCan = gq_bcast<0>(Tp.x);
    // ---- air side (identical in the four lanes): ventilation, screen air flux, air streams (rhs_fast)
    const T dTOut = tAir - s.tOut;
    const T buoy = m.gHVent * dTOut * M::rcpn(tAir + s.tOutK2);
    const T fVentRoof = s.ventK * M::sqrt0(M::abs(buoy + s.windTerm)) + s.ventElse + s.leakTop;
    const T tAirK = tAir + c2k, tTopK = tTop + c2k;
    const T iAirK = M::rcpn(tAirK), iTopK = M::rcpn(tTopK);
    const T rhoMean = T(0.5) * m.kRho * (iAirK + iTopK);
    const T dRho = M::abs(m.kRho * (tTop - tAir) * iAirK * iTopK);
    const T pw66 = SHARE_POW ? gq_bcast<0>(powX) : M::powa(M::abs(tAir - tTop + eps), T(0.66));
    const T iRhoMean = M::rcpn(rhoMean);
    const T fTh = s.kTh * pw66 + s.oneMinusUTh * iRhoMean * M::sqrtn(m.gHalf * rhoMean * s.oneMinusUTh * dRho + eps);
    const T fBl = s.kBl * pw66 + s.oneMinusUBl * iRhoMean * M::sqrtn(m.gHalf * rhoMean * s.oneMinusUBl * dRho + eps);
    const T fScrAbs = M::abs(M::min(fTh, fBl)), fRoofAbs = M::abs(fVentRoof), fSideAbs = M::abs(s.fVentSide);
    T vAirOverT, vTopOverT;
    if (sizeof(T) == 8) { vAirOverT = vpAir * M::rcpn(tAir + Kelvin<T>::c2kF32()); vTopOverT = vpTop * M::rcpn(tTop + Kelvin<T>::c2kF32()); }
    else { vAirOverT = vpAir * iAirK; vTopOverT = vpTop * iTopK; }
    const T kMv = T(0.002165);
    const T hAirTop = m.rhoCp * fScrAbs * (tAir - tTop), hTopOut = m.rhoCp * fRoofAbs * (tTop - s.tOut);
    const T mvAirTop = kMv * fScrAbs * (vAirOverT - vTopOverT), mvTopOut = kMv * fRoofAbs * (vTopOverT - s.vpOutOverT);
    const T mcAirTop = fScrAbs * (co2Air - co2Top), mcTopOut = fRoofAbs * (co2Top - s.co2Out);
    const T mvAirOut = kMv * fSideAbs * (vAirOverT - s.vpOutOverT), mcAirOut = fSideAbs * (co2Air - s.co2Out);
    const T hAirOut = s.hAirOutK * dTOut;
    k.sh[0] = m.iCapCo2Air * (s.mcExtAir - q.mcAirCan - mcAirTop - mcAirOut);
    k.sh[1] = m.iCapCo2Top * (mcAirTop - mcTopOut);
    k.sh[2] = m.iCapAir * (sHeatAir + q.swAir - hAirOut - hAirTop + q.hGroPipeAir + hIntAir);
    k.sh[3] = m.iCapTop * (sHeatTop + hAirTop - hTopOut);
    k.sh[4] = m.kCapVpAir * tAirK * (sVapAir - mvAirTop - mvAirOut);
    k.sh[5] = m.kCapVpTop * tTopK * (sVapTop + mvAirTop - mvTopOut);
    // the pair in the integrator's coordinates: d(tAir - T)/dt on the screens, d(tTop - tCovIn)/dt on the cover
    // (cover lane: the classical part of slot 5, d(tTop - sigma / 2)/dt, and N_w = nIn - nOut: rhs_fast<COVEXP>)
    k.p = gq_mk<T>(scr ? k.sh[2] - dTp.x : cov ? k.sh[3] - T(0.5) * (dTp.x + dTp.y) : dTp.x, scr ? k.sh[2] - dTp.y : cov ? dTp.x - dTp.y : dTp.y);
    const T perDay = T(1.0 / 86400.0);
    k.o[0] = lane0 ? perDay * (tCan - y.o[0]) : cov ? dInt : K.ro[0];  // lane 0: tCan24, tCanSum; lane 3: tIntLamp, time
    k.o[1] = lane0 ? perDay * tCan : cov ? perDay : K.ro[1];
    k.o[2] = K.ro[2]; k.o[3] = K.ro[3];
    if (PIPE) {
        k.p = gq_mk<T>(k.p.x, (lane0 && s.pipeTrack != T(0)) ? s.tPipeSet - Tp.y : k.p.y);      // ode.hpp:184-189 (pipeTrack is 0 outside ODE_pipe)
        k.o[3] = (role == 1 && s.pipeOde != T(0)) ? T(0) : k.o[3];                               // ode.hpp:240
    }
    if (RATES) { R->hecA = hecA; R->hecB = hecB; R->sv = sv; R->rr = rr; R->g = g; R->Tsurf = Tp; R->fScrAbs = fScrAbs; R->fRoofAbs = fRoofAbs; R->tTopK = tTopK; }
}

// ---- rate bound + branch-invariant bits from the first stage of a window (gl_model.hpp rhs_fast<RATES>, term for term) ------------
template <class T>
__device__ __forceinline__ T gq_rate_bound(int role, const QVec<T>& y, const QVec<T>& k, const QRates<T>& R, const LaneK<T>& K,
                                           const StepCoef<T>& s, const ModelConst<T>& m, T h_nominal, int* side)
{
    using M = Math<T>;
    const T f43 = T(4.0 / 3.0), kMv = T(0.002165), L = m.latent, LK = L * T(6.4e-9), kDs = T(1.1 * 17.2694 * 238.3);
    const bool cov = role == 3, scr = role == 2;
    // values the top-compartment rows need from the cover lane (3) and the screen lane (2)
    const T hTopCovAbs = M::abs(gq_bcast<3>(R.hecA.x)), hecThTop = gq_bcast<2>(R.hecB.x), hecBlTop = gq_bcast<2>(R.hecB.y);
    const T fAir = R.fScrAbs + R.fRoofAbs;
    const T r1 = m.iCapCo2Top * fAir;
    const T r3 = m.iCapTop * (m.rhoCp * (R.fRoofAbs + T(5.0 / 3.0) * R.fScrAbs) + f43 * (hTopCovAbs + hecThTop + hecBlTop));
    const T r16 = m.kCapVpTop * (kMv * fAir + R.tTopK * T(6.4e-9 * 1.1) * hTopCovAbs);
    const T gam = m.iCapCov * m.cCovCond;                   // the conduction is integrated exactly: it leaves both cover rows
    const T rOther = M::max(M::max(r1, r3), M::max(r16, s.rateCovE - T(2) * gam));
    // the lane's own wet surfaces: x (and y on the screen lane).  Lane 3: cover; lane 2: thermal, blackout screen
    auto wet_smooth = [&](T hec, T sv, T r) { return LK * hec * (kDs * sv * r * r); };
    int sbits = 0;
    const bool want_far = GL_WAVE_ANY(*side != 0);          // in: was the window just taken capped? (rhs_fast<RATES>)
    T rows = T(0);
    auto surface = [&](bool on, int j, T iCap, T hcoef, T hecAbs, T g, T tSurf, T dT, T ddT, T base) {
        // harm gate and side bits: sc_policy.hpp sc_wet_surface; (second pass) the pinned rate: gl_model.hpp sc_pinned_rate()
        const bool harm = sc_wet_surface<T>(on, j, iCap, hcoef, hecAbs, g, tSurf, dT, ddT, LK, h_nominal, want_far, sbits);
        T row = iCap * (base + f43 * hecAbs);
        if (GL_WAVE_ANY(harm)) row = sc_pinned_rate_inl<T>(harm, iCap, hcoef, hecAbs, LK * g, dT, ddT, row, T(SC_LOOK) * h_nominal);
        rows = M::max(rows, on ? row : T(0));
    };
    // x component: cover (lane 3) | thermal screen (lane 2)
    {
        const T hecAbs = M::abs(R.hecA.x);
        const T base = cov ? wet_smooth(hecAbs, R.sv.x, R.rr.x) + s.firCovIn
                           : f43 * R.hecB.x + wet_smooth(hecAbs, R.sv.x, R.rr.x) + s.firTh;
        // cover: the true d(tTop - tCovIn)/dt = k.p.x - N_w / 2 + gam w (rk_delta's movement limiter has the same expression)
        const T ddT = cov ? k.p.x - T(0.5) * k.p.y + gam * y.p.y : k.p.x;
        surface(cov || scr, cov ? 0 : 1, cov ? m.iCapCov : m.iCapThScr, cov ? m.cTopCov : s.hTh, hecAbs, R.g.x, R.Tsurf.x, y.p.x, ddT, base);
    }
    // y component: blackout screen (lane 2)
    {
        const T hecAbs = M::abs(R.hecA.y);
        const T base = f43 * R.hecB.y + wet_smooth(hecAbs, R.sv.y, R.rr.y) + s.firBl;
        surface(scr, 2, m.iCapBlScr, s.hBl, hecAbs, R.g.y, R.Tsurf.y, y.p.y, k.p.y, base);
    }
    *side = gq_or(sbits);
    return M::max(rOther, gq_max(rows));
}

// 1 / tolerance of the error estimate / movement limiter for the lane's pair (sc_fast / sc_itol of gl_model.hpp: the fast states are
// co2Top tTop | z5 tCovE z7 | vpAir vpTop | tLamp | z20; the lamp is exempt from the movement limiter)
template <class T> struct QTol { P2<T> est, mov; };
template <class T> __device__ __forceinline__ QTol<T> gq_tol(int role)
{
    const T z = T(0), t8 = T(1.0 / 0.125), tl = T(1.0 / 0.5);
    QTol<T> t;
    t.est = gq_mk<T>(role >= 2 ? t8 : z, role >= 2 ? t8 : role == 1 ? tl : z);
    t.mov = gq_mk<T>(role >= 2 ? t8 : z, role >= 2 ? t8 : z);
    return t;
}
// shared fast states: co2Top (sh 1), tTop (sh 3), vpAir (sh 4), vpTop (sh 5)
template <class T> __device__ __forceinline__ T gq_fast_max(const QVec<T>& a, P2<T> tolP)
{
    using M = Math<T>;
    T w = M::max(M::max(M::abs(a.sh[1]) * T(1.0 / 12.5), M::abs(a.sh[3]) * T(1.0 / 0.125)),
                 M::max(M::abs(a.sh[4]) * T(1.0 / 12.5), M::abs(a.sh[5]) * T(1.0 / 12.5)));
    return M::max(w, M::max(M::abs(a.p.x) * tolP.x, M::abs(a.p.y) * tolP.y));
}

// ---- the sub-stepper: rk_delta<T, false, 4, WIN> of gl_model.hpp over the quad ---------------------------------------------------
// LDSQ = true (the fp64 build): the env-step's coefficient block `s` (one copy per quad) and the lane's `LaneK` record live in LDS;
// a compiler fence in front of every stage, whose operands are their addresses, makes hipcc re-read them there instead of
// hoisting ~115 doubles into registers -- in fp64 that is the difference between 512 registers + scratch (whose spill code hipcc
// 7.2 gets wrong on this kernel too) and a kernel that fits.
// ORDER: 4 (RK4), 3 (three-stage scheme), 2 (midpoint rule) -- rk_delta's three members of the exponential family.
// LDSC: the crop constants `cr` are a per-quad LDS record as well (per-env crop blocks); otherwise they are part of `m` and their
// address must NOT reach the fence (it would force a private copy of the whole kernel argument).
// (LDSQ builds hold the parameter block `m` in LDS too: its address joins the fence)
#define GQ_FENCE() do { if (LDSQ && LDSC) asm volatile("" : : "v"(&s), "v"(&K), "v"(&m), "v"(&cr) : "memory"); \
                        else if (LDSQ) asm volatile("" : : "v"(&s), "v"(&K), "v"(&m) : "memory"); } while (0)
// HVU: harvest_flow's wave-uniform early exits (gl_model.hpp; false for the fp32 kernels that hold the handle's parameters in SGPRs)
template <class T, int ORDER, int WIN, bool LDSQ, bool PIPE, bool LDSC = false, bool HVU = true>
__device__ __forceinline__ void rk_delta_quad(int role, const QVec<T>& z0, const StepCoef<T>& s, LaneK<T>& K, const ModelConst<T>& m,
                                              const CropConst<T>& cr, T dt, int n_sub, QVec<T>& del, ScStat<T>& st, int win_rt = 0)
{
    static_assert(ORDER == 5 || ORDER == 4 || ORDER == 3 || ORDER == 2, "ORDER");
    using M = Math<T>;
    const int WINR = win_rt > 0 ? win_rt : WIN;               // run-time window (glgym_set_window), rk_delta
    // the nominal windows; each window gets its own length from its rate bound (sc_policy.hpp: the decisions below are rk_delta's, by
    // construction -- both layouts call the same functions)
    const ScGrid<T> grid = sc_grid<T>(dt, n_sub, WINR);
    const T hw_nom = grid.hw_nom, hnom_nom = grid.hnom_nom;
    T hw = hw_nom, hnom = hnom_nom;
    T t_now = T(0), t_harv = T(0.5) * hw_nom;
    int n_left = 0;
    const T S = ScScheme<T, ORDER>::S(), est_fac = ScScheme<T, ORDER>::est_fac();
    T ls_Nprev = T(0), ls_hprev = T(0);       // ORDER 5, cover lane: N_w at the start of the previous sub-step and its length (0: none yet)
    const bool lane0 = role == 0, crop = role == 2, cov = role == 3;
    const T gam = m.iCapCov * m.cCovCond, cw = cov ? T(0.5) : T(0);
    QVec<T> y, xs, k, acc;
    // increments over the previous window / at the start of this one of the SEVEN entries tier 2b reads from a lane (pair x, shared
    // co2Air and tAir, the four "others"): ym of slow_coef
    T dprev[7], dwin[7];
    auto slow7 = [](const QVec<T>& v, T* o7) { o7[0] = v.p.x; o7[1] = v.sh[0]; o7[2] = v.sh[2]; for (int j = 0; j < 4; ++j) o7[3 + j] = v.o[j]; };
    P2<T> estP = gq_sp<T>(T(0));          // comparison stage of the error estimate: the pair and the six shared states
    T estS[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    auto zero = [](QVec<T>& v) { v.p = gq_sp<T>(T(0)); for (int i = 0; i < 6; ++i) v.sh[i] = T(0); for (int j = 0; j < 4; ++j) v.o[j] = T(0); };
    zero(del);
    for (int j = 0; j < 7; ++j) { dprev[j] = T(0); dwin[j] = T(0); }
    int n_steps = 0, flags = 0, side_prev = 0;
    bool capped_prev = false;
    T t_cap = T(0);               // time spent in windows taken at the refinement cap
    T h_last = T(-1);             // length of the last sub-step taken = the one K.ec holds the coefficients of
    // the pair's y component by the ETD formulas of rk_delta: the cover lane with a = 2 gam, every other lane with a = 0, for which the
    // coefficients are those of classical RK4 -- one instruction stream for the four lanes
    // harvest: cLeaf = x23, cFruit = x25 live on lane 2 (o[1], o[3])
    auto harvest = [&](T hh) {
        const T a = harvest_flow<T, HVU>(z0.o[1] + del.o[1], cr.cLeafMax, hh), b = harvest_flow<T, HVU>(z0.o[3] + del.o[3], cr.cFruitMax, hh);
        del.o[1] += crop ? a : T(0); del.o[3] += crop ? b : T(0);
    };
    QVec<T> winc;                 // ORDER 5: the window's increments (see the sub-step loop)
    T w_now = T(0);               // ORDER 5: the pair's y component at the start of the sub-step about to be taken
    auto stage_in = [&]() {       // ORDER 5: xs = y + winc (y = z0 + del of the window's start)
        xs.p = y.p + winc.p;
        for (int i = 0; i < 6; ++i) xs.sh[i] = y.sh[i] + winc.sh[i];
        xs.o[0] = y.o[0] + winc.o[0]; xs.o[1] = y.o[1] + winc.o[1]; xs.o[2] = y.o[2]; xs.o[3] = y.o[3];
    };
    auto state_now = [&]() { y.p = z0.p + del.p; for (int i = 0; i < 6; ++i) y.sh[i] = z0.sh[i] + del.sh[i]; for (int j = 0; j < 4; ++j) y.o[j] = z0.o[j] + del.o[j]; };
    // the cover lane's derivatives of (tTop - tCovIn, tCovE) from the integrator's (classical part of slot 5, N_w): rk_delta's movement limiter
    auto true_rates = [&](const QVec<T>& kk, const QVec<T>& yy) {
        QVec<T> r = kk;
        const T c = T(0.5) * kk.p.y - gam * yy.p.y;
        r.p = gq_mk<T>(cov ? kk.p.x - c : kk.p.x, cov ? (kk.sh[3] - kk.p.x) - c : kk.p.y);
        return r;
    };
    SlowCoef<T> q;
    LaneStep<T> LS;
    if (!LDSQ) gq_make_lane_step<T>(role, s, m, K, LS);
    harvest(T(0.5) * hw_nom);                                  // leading half of the exact harvest flow (rk_delta)
    for (int it = 0;; ++it) {
        const T t_left = dt - t_now;
        const bool closing = it > 0 && n_left <= 1;
        flags |= (t_cap > T(SC_CAP_S)) ? SC_FLAG_CAP : 0;
        if (flags & SC_FLAG_CAP) break;
        // ---- window start: tier 2b at the predicted window midpoint; every lane evaluates it from the gathered inputs
        state_now();
        {
            T now7[7], mid[7];
            slow7(y, now7);
            for (int j = 0; j < 7; ++j) mid[j] = now7[j] + T(0.5) * dprev[j];
            slow7(del, dwin);
            T ym[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) ym[i] = T(0);
            ym[0] = mid[1]; ym[2] = mid[2];
            ym[4] = gq_bcast<0>(mid[0]); ym[21] = gq_bcast<0>(mid[3]); ym[26] = gq_bcast<0>(mid[4]); ym[10] = gq_bcast<0>(mid[5]);
            ym[11] = gq_bcast<0>(mid[6]);
            ym[8] = gq_bcast<1>(mid[0]); ym[12] = gq_bcast<1>(mid[3]); ym[13] = gq_bcast<1>(mid[4]); ym[14] = gq_bcast<1>(mid[5]);
            ym[19] = gq_bcast<1>(mid[6]);
            ym[22] = gq_bcast<2>(mid[3]); ym[23] = gq_bcast<2>(mid[4]); ym[24] = gq_bcast<2>(mid[5]); ym[25] = gq_bcast<2>(mid[6]);
            slow_coef<T>(ym, s, m, cr, q);
            if (LDSQ) gq_make_lane<T>(role, s, m, q, K);
            else gq_make_lane_win<T>(role, s, m, q, K, LS);
        }
        // ---- first stage of the window's first sub-step with the rate bound; branch invariant; error estimate of the last sub-step
        QRates<T> R;
        GQ_FENCE(); gq_stage<T, true, PIPE>(role, y, K, s, m, q, k, &R);
        int side = capped_prev ? 1 : 0;
        T lam = gq_rate_bound<T>(role, y, k, R, K, s, m, hnom_nom, &side);
        if (PIPE) lam = (s.pipeTrack != T(0)) ? M::max(lam, T(1)) : lam;      // dxdt(9) = tPipeSet - x9: rate 1 1/s (rhs_fast<RATES, PIPE>)
        flags |= sc_branch_flag(side_prev, side, capped_prev);
        side_prev = side;
        if (it > 0) {
            QVec<T> dif;
            dif.p = estP - k.p;
            for (int i = 0; i < 6; ++i) dif.sh[i] = estS[i] - k.sh[i];
            const T worst = gq_max(gq_fast_max(dif, gq_mk<T>(gq_tol<T>(role).est.x, (ORDER == 4 || ORDER == 3) ? gq_tol<T>(role).est.y * K.ec.w3 : gq_tol<T>(role).est.y)));     // (the ETD component's estimate carries f3)
            flags |= sc_estimate_flag<T>(grid, worst, h_last, est_fac, t_now);
        }
        if (closing) break;
        // this window's length from its rate bound; its sub-steps from stability, the movement limiter (the quad's largest movement
        // rate) and the refinement cap (sc_policy.hpp)
        sc_window_length<T>(grid, S, lam, t_left, it == 0, hw, hnom, n_left);
        const ScPlan<T> plan = sc_plan<T, ORDER>(grid, S, lam, hw, hnom, gq_max(gq_fast_max(true_rates(k, y), gq_tol<T>(role).mov)));
        const bool capped = plan.capped;
        t_cap += capped ? hw : T(0);
        capped_prev = capped;
        T n_rem = plan.n_rem;
        T h = plan.h;
        // ORDER 5: a limiter-bound window re-partitions its remainder sub-step by sub-step; uniform inside the quad
        const bool adaptive = plan.adaptive;
        T t_rem = hw;
        if (h != h_last) {
            if (ORDER == 5) ls_coefs<T>(cov ? T(2) * gam : T(0), h, K.lc); else etd_coefs<T>(cov ? T(2) * gam : T(0), h, K.ec);
        }
        h_last = h;
        // one sub-step from (y, k = f(y)): the classical scheme on the pair's x component, the shared states and lane 0's / lane 3's first
        // two "others" (tCan24, tCanSum | tIntLamp, time), its ETD sibling on the pair's y component (classical coefficients off the
        // cover lane), the cover lane's x assembled from tTop, sigma and w (rk_delta), constant rate for the rest
        auto sub_step = [&]() {
            const T h2 = T(0.5) * h, h6 = h * T(1.0 / 6.0);
            const T w0 = (ORDER == 5) ? w_now : y.p.y, n1 = k.p.y;
            const bool full01 = lane0 || role == 3;
            auto fill = [&](T c, T dW) {
                xs.p = gq_mk<T>(y.p.x + c * k.p.x - cw * dW, w0 + dW);
                for (int i = 0; i < 6; ++i) xs.sh[i] = y.sh[i] + c * k.sh[i];
                xs.o[0] = y.o[0] + c * k.o[0]; xs.o[1] = y.o[1] + c * k.o[1]; xs.o[2] = y.o[2]; xs.o[3] = y.o[3];
            };
            if (ORDER == 5) {
                // the five-stage 2N scheme (gl_model.hpp rk_delta ORDER 5): acc holds dy / h; the stage input is z0 + del.  The pair's y
                // component by the exponential form with per-lane coefficients -- a = 2 gam on the cover lane; a = 0 elsewhere, for
                // which E = 1, dphi = h dc and the formulas ARE the plain 2N scheme: one instruction stream for the four lanes
                const T F0 = n1 - (cov ? T(2) * gam : T(0)) * w0;
                // linear predictor of the cover lane's forcing (rk_delta): slope from the previous sub-step's start value; none elsewhere
                // (a = 0: the plain 2N scheme needs none)
                const T slope = (cov && h <= T(2.0001) * ls_hprev) ? (n1 - ls_Nprev) * M::rcp(ls_hprev) : T(0);
                ls_Nprev = n1; ls_hprev = h;
                const T slope_ia = slope * M::rcp(T(2) * gam);
                T vv = T(0), dv = T(0);
#pragma unroll
                for (int stg = 0; stg < 5; ++stg) {
                    if (stg > 0) { stage_in(); GQ_FENCE(); gq_stage<T, false, PIPE>(role, xs, K, s, m, q, k, nullptr); }
                    const T Ai = T(Ls5<T>::A(stg)), Bi = T(Ls5<T>::B(stg)), Bh = Bi * h;
                    acc.p.x = (stg == 0) ? k.p.x : Ai * acc.p.x + k.p.x;
                    for (int i = 0; i < 6; ++i) { acc.sh[i] = (stg == 0) ? k.sh[i] : Ai * acc.sh[i] + k.sh[i]; winc.sh[i] += Bh * acc.sh[i]; }
                    acc.o[0] = (stg == 0) ? k.o[0] : Ai * acc.o[0] + k.o[0]; acc.o[1] = (stg == 0) ? k.o[1] : Ai * acc.o[1] + k.o[1];
                    dv = (stg == 0) ? T(0) : Ai * dv + h * ((k.p.y - n1) - slope * (T(Ls5<T>::c(stg)) * h));
                    const T vnext = K.lc.E[stg] * (vv + Bi * dv);
                    dv = K.lc.E[stg] * dv;
                    const T dW = K.lc.dphi[stg] * F0 + (h * T(Ls5<T>::c(stg + 1) - Ls5<T>::c(stg)) - K.lc.dphi[stg]) * slope_ia + (vnext - vv);
                    vv = vnext;
                    winc.p = gq_mk<T>(winc.p.x + (Bh * acc.p.x - cw * dW), winc.p.y + dW);
                    winc.o[0] += full01 ? Bh * acc.o[0] : T(0);
                    winc.o[1] += full01 ? Bh * acc.o[1] : T(0);
                }
                del.o[0] += full01 ? T(0) : h * k.o[0];          // constant-rate "others" of lanes 1 / 2: straight into del
                del.o[1] += full01 ? T(0) : h * k.o[1];
            } else if (ORDER == 4) {
                T dWa, accW;
                auto accum = [&]() {
                    acc.p.x += T(2) * k.p.x;
                    for (int i = 0; i < 6; ++i) acc.sh[i] += T(2) * k.sh[i];
                    acc.o[0] += T(2) * k.o[0]; acc.o[1] += T(2) * k.o[1];
                    accW += K.ec.f2d * k.p.y;
                };
                acc = k;
                dWa = K.ec.e2m1 * w0 + K.ec.q * n1; accW = K.ec.f1 * n1;
                fill(h2, dWa);
                GQ_FENCE(); gq_stage<T, false, PIPE>(role, xs, K, s, m, q, k, nullptr);
                accum();
                fill(h2, K.ec.e2m1 * w0 + K.ec.q * k.p.y);
                GQ_FENCE(); gq_stage<T, false, PIPE>(role, xs, K, s, m, q, k, nullptr);
                accum();
                fill(h, K.ec.e2m1 * w0 + K.ec.e2 * dWa + K.ec.q * (T(2) * k.p.y - n1));
                GQ_FENCE(); gq_stage<T, false, PIPE>(role, xs, K, s, m, q, k, nullptr);
                const T dW = K.ec.em1 * w0 + accW + K.ec.f3 * k.p.y;
                del.p = gq_mk<T>(del.p.x + h6 * (acc.p.x + k.p.x) - cw * dW, del.p.y + dW);
                for (int i = 0; i < 6; ++i) del.sh[i] += h6 * (acc.sh[i] + k.sh[i]);
                del.o[0] += full01 ? h6 * (acc.o[0] + k.o[0]) : h * k.o[0];
                del.o[1] += full01 ? h6 * (acc.o[1] + k.o[1]) : h * k.o[1];
            } else if (ORDER == 3) {
                // k2 = f(y + h/2 k1), k3 = f(y + h (2 k2 - k1)), y+ = y + h/6 (k1 + 4 k2 + k3);  w: ETD3RK (rk_delta)
                acc = k;
                T accW = K.ec.f1 * n1;
                fill(h2, K.ec.e2m1 * w0 + K.ec.q * n1);
                GQ_FENCE(); gq_stage<T, false, PIPE>(role, xs, K, s, m, q, k, nullptr);
                {
                    const T dWb = K.ec.em1 * w0 + K.ec.hp1 * (T(2) * k.p.y - n1);
                    xs.p = gq_mk<T>(y.p.x + h * (T(2) * k.p.x - acc.p.x) - cw * dWb, w0 + dWb);
                    for (int i = 0; i < 6; ++i) xs.sh[i] = y.sh[i] + h * (T(2) * k.sh[i] - acc.sh[i]);
                    xs.o[0] = y.o[0] + h * (T(2) * k.o[0] - acc.o[0]); xs.o[1] = y.o[1] + h * (T(2) * k.o[1] - acc.o[1]);
                    xs.o[2] = y.o[2]; xs.o[3] = y.o[3];
                    acc.p.x += T(4) * k.p.x;
                    for (int i = 0; i < 6; ++i) acc.sh[i] += T(4) * k.sh[i];
                    acc.o[0] += T(4) * k.o[0]; acc.o[1] += T(4) * k.o[1];
                    accW += T(2) * K.ec.f2d * k.p.y;
                }
                GQ_FENCE(); gq_stage<T, false, PIPE>(role, xs, K, s, m, q, k, nullptr);
                const T dW = K.ec.em1 * w0 + accW + K.ec.f3 * k.p.y;
                del.p = gq_mk<T>(del.p.x + h6 * (acc.p.x + k.p.x) - cw * dW, del.p.y + dW);
                for (int i = 0; i < 6; ++i) del.sh[i] += h6 * (acc.sh[i] + k.sh[i]);
                del.o[0] += full01 ? h6 * (acc.o[0] + k.o[0]) : h * k.o[0];
                del.o[1] += full01 ? h6 * (acc.o[1] + k.o[1]) : h * k.o[1];
            } else {
                // midpoint rule: k2 = f(y + h/2 k1), y+ = y + h k2;  w: ETD2RK.  The estimate's comparison stage is 2 k2 - k1
                estP = gq_mk<T>(-k.p.x, -k.p.y);
                for (int i = 0; i < 6; ++i) estS[i] = -k.sh[i];
                fill(h2, K.ec.e2m1 * w0 + K.ec.q * n1);
                GQ_FENCE(); gq_stage<T, false, PIPE>(role, xs, K, s, m, q, k, nullptr);
                const T dW = K.ec.em1 * w0 + K.ec.hp1 * k.p.y;
                del.p = gq_mk<T>(del.p.x + h * k.p.x - cw * dW, del.p.y + dW);
                for (int i = 0; i < 6; ++i) del.sh[i] += h * k.sh[i];
                del.o[0] += h * k.o[0]; del.o[1] += h * k.o[1];
                estP = gq_mk<T>(estP.x + T(2) * k.p.x, estP.y + T(2) * k.p.y);
                for (int i = 0; i < 6; ++i) estS[i] += T(2) * k.sh[i];
            }
            del.o[2] += h * k.o[2]; del.o[3] += h * k.o[3];
            ++n_steps;
        };
        // ORDER 5: the window's increments of the pair, the shared states and (lanes 0 / 3) the first two "others" are accumulated apart
        // and reach del at the window's end; the stage input is y + winc (rk_delta: rounding)
        if (ORDER == 5) { winc.p = gq_sp<T>(T(0)); for (int i = 0; i < 6; ++i) winc.sh[i] = T(0); winc.o[0] = T(0); winc.o[1] = T(0); }
        w_now = y.p.y;
        sub_step();
        t_rem -= h;
        for (n_rem -= T(1); n_rem >= T(0.5); n_rem -= T(1)) {
            if (ORDER == 5) {
                stage_in();
                w_now = xs.p.y;
                GQ_FENCE(); gq_stage<T, false, PIPE>(role, xs, K, s, m, q, k, nullptr);
            } else {
                state_now();
                GQ_FENCE(); gq_stage<T, false, PIPE>(role, y, K, s, m, q, k, nullptr);
            }
            if (ORDER == 5 && GL_WAVE_ANY(adaptive)) {
                // the limiter again with this sub-step's first stage; the rest of the window re-partitioned (rk_delta, decision for decision)
                sc_replan<T>(grid, plan, adaptive, t_rem, gq_max(gq_fast_max(true_rates(k, xs), gq_tol<T>(role).mov)), h, n_rem);
                if (h != h_last) ls_coefs<T>(cov ? T(2) * gam : T(0), h, K.lc);
                h_last = h;
            }
            sub_step();
            t_rem -= h;
        }
        if (ORDER != 2) {                                  // the last stage of the window's last sub-step (midpoint: set in sub_step)
            estP = k.p;
            for (int i = 0; i < 6; ++i) estS[i] = k.sh[i];
        }
        // ---- window end
        if (ORDER == 5) {
            del.p = del.p + winc.p;
            for (int i = 0; i < 6; ++i) del.sh[i] += winc.sh[i];
            del.o[0] += winc.o[0]; del.o[1] += winc.o[1];
        }
        {
            T end7[7];
            slow7(del, end7);
            for (int j = 0; j < 7; ++j) dprev[j] = end7[j] - dwin[j];
        }
        t_now = (n_left <= 1) ? dt : t_now + hw;
        harvest(sc_harvest_advance<T>(dt, t_now, hw, t_harv));
    }
    if (role == 3) del.o[1] = dt * T(1.0 / 86400.0);          // x27 = time [days]
    st.n_steps = n_steps;
    st.flags = flags;
}

// physical increments of the lane's fast states (for the agreement test of the guard) and finiteness
template <class T> __device__ __forceinline__ void gq_phys_pair(int role, const QVec<T>& del, P2<T>& out)
{
    const T dx = role == 2 ? del.sh[2] - del.p.x : role == 3 ? del.sh[3] - del.p.x : del.p.x;     // tThScr | tCovIn from the differences
    out = gq_mk<T>(dx, role == 2 ? del.sh[2] - del.p.y : role == 3 ? dx - del.p.y : del.p.y);     // tBlScr | tCovE = tCovIn - w
}

// ---- the guard: rk4_delta_guarded of gl_model.hpp over the quad (same ladder, same acceptance rules) ---------------------------------
template <class T, int ORDER, int WIN, bool LDSQ, bool PIPE, bool LDSC = false, bool HVU = true>
__device__ __forceinline__ int rk4_delta_guarded_quad(int role, const QVec<T>& z0, const StepCoef<T>& s, LaneK<T>& K, const ModelConst<T>& m,
                                                      const CropConst<T>& cr, T dt, int n_sub, QVec<T>& del, bool* failed, int* extra_steps,
                                                      bool verify, int* first_flags, int win_rt = 0)
{
    const int WINR = win_rt > 0 ? win_rt : WIN;
    const QTol<T> tol = gq_tol<T>(role);
    ScLadder L = sc_ladder_start(n_sub);
    QVec<T> prev;
    prev.p = gq_sp<T>(T(0));
    for (int i = 0; i < 6; ++i) prev.sh[i] = T(0);
    for (int attempt = 0; attempt < SC_ATTEMPTS; ++attempt) {
        if (L.done) break;                                 // uniform inside the quad: every decision below is
        ScStat<T> st;
        rk_delta_quad<T, ORDER, WIN, LDSQ, PIPE, LDSC, HVU>(role, z0, s, K, m, cr, dt, L.n, del, st, win_rt);
        T chk = (del.p.x + del.p.y) * T(0);
        for (int i = 0; i < 6; ++i) chk += del.sh[i] * T(0);
        for (int j = 0; j < 4; ++j) chk += del.o[j] * T(0);
        const bool finite = gq_or((chk == T(0)) ? 0 : 1) == 0;
        QVec<T> now;
        gq_phys_pair<T>(role, del, now.p);
        for (int i = 0; i < 6; ++i) now.sh[i] = del.sh[i];
        QVec<T> dif;
        dif.p = now.p - prev.p;
        for (int i = 0; i < 6; ++i) dif.sh[i] = now.sh[i] - prev.sh[i];
        const T worst = gq_max(gq_fast_max(dif, tol.est));
        sc_ladder_judge<T>(L, attempt, st.flags, st.n_steps, WINR, finite, worst, verify, first_flags);      // the acceptance rules: sc_policy.hpp
        prev = now;
    }
    *failed = !L.ok;
    if (extra_steps) *extra_steps = sc_ladder_extra_steps(L, n_sub, WINR);
    return L.extra;
}

// The verified ladder with TWO rungs at a time (round 5; glgym_evalF at small batches, where lanes are free and the call is a latency
// chain): two quads per row, `half` 0 / 1, integrate the same row with n_sub and 2 n_sub sub-steps side by side (attempts 0 and 1),
// exchange their results across the quads (ds_bpermute) and both replay rk4_delta_guarded_quad's decisions on them in its order; the
// rare later attempts (4 n_sub, 8 n_sub) run on lane group 0 alone.  The accepted attempt, `failed` and the returned state are those
// of the sequential ladder bit for bit (each attempt is a pure function of (z0, n)); what differs is the elapsed time -- the 2 n
// attempt's instead of n + 2 n.  Verified mode only
// (no attempt is accepted on its own; unverified integrations accept a clean first attempt and have nothing to run beside it).
// *mine: this quad holds the accepted attempt in `del` (the quad that writes the row).
// Round 6 (glgym_step(control = ...) at small batches runs it too): *extra / *extra_steps / *first_flags as rk4_delta_guarded_quad reports
// them -- extra attempts, sub-steps beyond the nominal count over all attempts run, the GLGYM_SF_* word of the first attempt and of the
// acceptance (sc_policy.hpp sc_ladder_judge in verified mode, replayed with integer selects).
// BOOK = false (glgym_evalF, which reports none of it): the bookkeeping is compiled out -- three more values live across the integrator
// cost the fp32 evalF kernel (380 registers, the hot loop's operands partly in AGPRs) 6 % of its sub-step loop.
template <class T, int ORDER, int WIN, bool LDSQ, bool PIPE, bool LDSC = false, bool BOOK = false, bool HVU = true>
__device__ __forceinline__ void rk4_delta_guarded_quad_pair(int role, int half, const QVec<T>& z0, const StepCoef<T>& s, LaneK<T>& K,
                                                            const ModelConst<T>& m, const CropConst<T>& cr, T dt, int n_sub, QVec<T>& del,
                                                            int* failed, int* mine, int win_rt = 0, int* extra = nullptr,
                                                            int* extra_steps = nullptr, int* first_flags = nullptr)
{
    static_assert(SC_ATTEMPTS == 4, "two rounds of two attempts");
    const int WINR = win_rt > 0 ? win_rt : WIN;
    const QTol<T> tol = gq_tol<T>(role);
    // The ladder's state is carried as INTEGERS in vector registers (0 / 1), updated by selects, with an opaque barrier per round: as
    // `bool`s assigned under `break`s the compiler turned them into lane masks in spilled scalar registers whose loop-carried merge read
    // a slot no path had written (hipcc 7.2, the fp32 RK4 / three-stage builds: the row-0 lane of the accepted quad sometimes did not
    // write, depending on what earlier kernels had left in that register -- tools/README.md "pair ladder").
    int done = 0, ok = 0, have_prev = 0, winner = 0;
    int n_extra = 0, total = 0, fflags = 0;                // what the sequential ladder reports (step_flags): same selects, same barriers
    QVec<T> prev;
    prev.p = gq_sp<T>(T(0));
    for (int i = 0; i < 6; ++i) prev.sh[i] = T(0);
    // round 0: attempts 0 and 1 side by side; rounds 1, 2 (rare): attempt round + 1 on lane group 0 alone, the other group waits --
    // integrating attempt 3 speculatively beside attempt 2 would cost a row that needs three attempts 2 + 8 = 10 n where the sequential
    // ladder takes 1 + 2 + 4 = 7 n; this way it is 2 + 4 = 6 n (four attempts: 14 n against 15 n)
    for (int round = 0; round < SC_ATTEMPTS - 1; ++round) {
        if (done != 0) break;                              // uniform over the two quads of a row
        const int att_mine = (round == 0) ? half : round + 1;
        const int n = n_sub << att_mine;
        ScStat<T> st;
        st.flags = 0; st.n_steps = 0;
        if (round == 0 || half == 0) rk_delta_quad<T, ORDER, WIN, LDSQ, PIPE, LDSC, HVU>(role, z0, s, K, m, cr, dt, n, del, st, win_rt);
        const int n_nom = ((n + WINR - 1) / WINR) * WINR;
        T chk = (del.p.x + del.p.y) * T(0);
        for (int i = 0; i < 6; ++i) chk += del.sh[i] * T(0);
        for (int j = 0; j < 4; ++j) chk += del.o[j] * T(0);
        const int nonfinite = gq_or((chk == T(0)) ? 0 : 1);
        // bit 0: complete; bit 1: no flag; bit 2: not heavy; bits 3-6: the attempt's SC_FLAG_* word; bits 8...: its sub-steps
        int code = (((nonfinite == 0) && !(st.flags & (SC_FLAG_CAP | SC_FLAG_NONFINITE))) ? 1 : 0) | ((st.flags == 0) ? 2 : 0) |
                   ((st.n_steps < SC_HEAVY * n_nom) ? 4 : 0) | (BOOK ? (((st.flags & 15) << 3) | (st.n_steps << 8)) : 0);
        asm volatile("" : "+v"(code));
        QVec<T> now, oth;
        gq_phys_pair<T>(role, del, now.p);
        for (int i = 0; i < 6; ++i) now.sh[i] = del.sh[i];
        const int ocode = __shfl_xor(code, 4);
        oth.p = gq_mk<T>(__shfl_xor(now.p.x, 4), __shfl_xor(now.p.y, 4));
        for (int i = 0; i < 6; ++i) oth.sh[i] = __shfl_xor(now.sh[i], 4);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {                   // the attempts of this round in the sequential ladder's order; no early exit:
            const int act = (done ^ 1) & ((round == 0 || hh == 0) ? 1 : 0);      // a decided row keeps its state through the selects below
            const int holder = (round == 0) ? hh : 0, att = (round == 0) ? hh : round + 1;
            const int own = (holder == half) ? 1 : 0;
            const int c = own ? code : ocode;
            QVec<T> cur, dif;
            cur.p = gq_mk<T>(own ? now.p.x : oth.p.x, own ? now.p.y : oth.p.y);
            for (int i = 0; i < 6; ++i) cur.sh[i] = own ? now.sh[i] : oth.sh[i];
            dif.p = cur.p - prev.p;
            for (int i = 0; i < 6; ++i) dif.sh[i] = cur.sh[i] - prev.sh[i];
            const T worst = gq_max(gq_fast_max(dif, tol.est));
            const int complete = c & 1, last = (att == SC_ATTEMPTS - 1) ? 1 : 0;
            const int agree = complete & have_prev & ((worst <= T(SC_AGREE)) ? 1 : 0);
            const int ok_n = agree | (last & complete & ((c >> 1) & 1));
            if constexpr (BOOK) {      // sc_ladder_judge's bookkeeping (verified mode: no attempt is accepted as merely clean)
                total = act ? total + (c >> 8) : total;
                fflags = (act && att == 0) ? (((c >> 3) & 15) | (((c >> 2) & 1) ? 0 : 16)) : fflags;
                fflags = (act && ok_n) ? (fflags | (agree ? ((((c >> 1) & 1) == 0) ? 32 : 0) : 64)) : fflags;
                n_extra = (act && !(ok_n | last)) ? n_extra + 1 : n_extra;
            }
            ok = act ? ok_n : ok;
            winner = act ? holder : winner;
            have_prev = act ? complete : have_prev;
            prev.p = gq_mk<T>(act ? cur.p.x : prev.p.x, act ? cur.p.y : prev.p.y);
            for (int i = 0; i < 6; ++i) prev.sh[i] = act ? cur.sh[i] : prev.sh[i];
            done = act ? (ok_n | last) : done;
            asm volatile("" : "+v"(done), "+v"(ok), "+v"(winner), "+v"(have_prev));
            if constexpr (BOOK) asm volatile("" : "+v"(n_extra), "+v"(total), "+v"(fflags));
        }
    }
    *failed = ok ^ 1;
    *mine = (half == winner) ? 1 : 0;
    if (extra) *extra = n_extra;
    if (extra_steps) { const int ex = total - ((n_sub + WINR - 1) / WINR) * WINR; *extra_steps = ex > 0 ? ex : 0; }
    if (first_flags) *first_flags = fflags;
}

}  // namespace glm
#endif
