// TESTS ONLY.  Host (g++) instantiation of greenlight-gym2_amd/csrc/gl_model.hpp so that the exact arithmetic the
// gfx950 kernels run (tier-1/2/3 split, delta-form RK4, fp32 reformulations) can be unit-tested against the
// oracle on a machine without a GPU.  The product library (libglgym.so) never links or loads this file.
#include "gl_model.hpp"

using namespace glm;

template <class T, bool PIPE = false>
static void run(const double* x, const double* u, const double* d, const double* p, int per_env_crop, double dt,
                int n_sub, double* out, int rhs_only)
{
    ModelConst<T> m;
    make_model_const<T>(p, m);
    T x0[NX], uu[NU], dd[7];
    for (int i = 0; i < NX; ++i) x0[i] = T(x[i]);
    for (int i = 0; i < NU; ++i) uu[i] = T(u[i]);
    for (int i = 0; i < 7; ++i) dd[i] = T(d[i]);
    CropConst<T> crLocal;
    if (per_env_crop) {
        T pc[NCROP];
        for (int i = 0; i < NCROP; ++i) pc[i] = T(p[CROP0 + i]);
        make_crop_const<T, T>(pc, T(p[39]), T(p[162]), crLocal);
    }
    const CropConst<T>& cr = per_env_crop ? crLocal : m.crop;
    StepCoef<T> s;
    precompute(uu, dd, m, cr, s);
    if (PIPE) {                                   // same two lines as step_kernel / evalf_kernel (d has 14 entries)
        const T tPipe = T(d[10]), swOff = T(d[12]);
        s.pipeTrack = ((tPipe < T(1)) || (swOff > T(0))) ? T(0) : T(1);
        s.tPipeSet = tPipe;
    }
    if (rhs_only) {
        T k[NX];
        rhs<T, true, PIPE>(x0, s, m, cr, k);
        for (int i = 0; i < NX; ++i) out[i] = (double)k[i];
        return;
    }
    T del[NX];
    rk4_delta<T, PIPE>(x0, s, m, cr, T(dt), n_sub, del);
    for (int i = 0; i < NX; ++i) out[i] = (double)x0[i] + (double)del[i];
}

// other integrator settings of rk_delta<T, PIPE, ORDER, WIN>: (order, window) in {(4,1), (4,2), (4,3), (4,4), (2,1), (2,2), (2,4), (3,1), (3,3), (3,4), (5,1), (5,2)}
template <class T, int ORDER, int WIN>
static void run_scheme(const double* x, const double* u, const double* d, const double* p, double dt, int n_sub, double* out,
                       double* stats)
{
    ModelConst<T> m;
    make_model_const<T>(p, m);
    T x0[NX], uu[NU], dd[7], del[NX];
    for (int i = 0; i < NX; ++i) x0[i] = T(x[i]);
    for (int i = 0; i < NU; ++i) uu[i] = T(u[i]);
    for (int i = 0; i < 7; ++i) dd[i] = T(d[i]);
    StepCoef<T> s;
    precompute(uu, dd, m, m.crop, s);
    ScStat<T> st;
    rk_delta<T, false, ORDER, WIN>(x0, s, m, m.crop, T(dt), n_sub, del, st);
    for (int i = 0; i < NX; ++i) out[i] = (double)x0[i] + (double)del[i];
    if (stats) { stats[0] = st.n_steps; stats[1] = st.flags; }
}
// the guarded step map exactly as step_kernel / evalf_kernel call it: returns retries, *failed, extra sub-steps
template <class T, int ORDER, int WIN>
static int run_guarded(const double* x, const double* u, const double* d, const double* p, double dt, int n_sub, double* out,
                       double* stats, int verify)
{
    ModelConst<T> m;
    make_model_const<T>(p, m);
    T x0[NX], uu[NU], dd[7], del[NX];
    for (int i = 0; i < NX; ++i) x0[i] = T(x[i]);
    for (int i = 0; i < NU; ++i) uu[i] = T(u[i]);
    for (int i = 0; i < 7; ++i) dd[i] = T(d[i]);
    StepCoef<T> s;
    precompute(uu, dd, m, m.crop, s);
    bool failed;
    int extra = 0;
    const int r = rk4_delta_guarded<T, false, ORDER, WIN>(x0, s, m, m.crop, T(dt), n_sub, del, &failed, &extra, verify != 0);
    for (int i = 0; i < NX; ++i) out[i] = (double)x0[i] + (double)del[i];
    if (stats) { stats[0] = extra; stats[1] = failed ? 1 : 0; }
    return r;
}
// rate bound of rhs_fast<RATES> at one state
template <class T> static double run_rate(const double* x, const double* u, const double* d, const double* p)
{
    ModelConst<T> m;
    make_model_const<T>(p, m);
    T x0[NX], uu[NU], dd[7], k[NX], lam = T(1e6);   // huge nominal sub-step: every wet surface counts as harmful (the pinned analysis always runs)
    for (int i = 0; i < NX; ++i) x0[i] = T(x[i]);
    for (int i = 0; i < NU; ++i) uu[i] = T(u[i]);
    for (int i = 0; i < 7; ++i) dd[i] = T(d[i]);
    StepCoef<T> s;
    precompute(uu, dd, m, m.crop, s);
    SlowCoef<T> q;
    slow_coef(x0, s, m, m.crop, q);
    rhs_fast<T, false, false, true>(x0, q, s, m, m.crop, k, &lam);
    return (double)lam;
}
extern "C" {
double hostmath_harvest_flow(double c, double cmax, double t, int f32)
{
    return f32 ? (double)harvest_flow<float>((float)c, (float)cmax, (float)t) : harvest_flow<double>(c, cmax, t);
}
void hostmath_rhs(const double* x, const double* u, const double* d, const double* p, int f32, int per_env_crop,
                  double* dx)
{
    if (f32) run<float>(x, u, d, p, per_env_crop, 0, 0, dx, 1);
    else run<double>(x, u, d, p, per_env_crop, 0, 0, dx, 1);
}
void hostmath_rhs_pipe(const double* x, const double* u, const double* d14, const double* p, int f32, double* dx)
{
    if (f32) run<float, true>(x, u, d14, p, 0, 0, 0, dx, 1);
    else run<double, true>(x, u, d14, p, 0, 0, 0, dx, 1);
}
void hostmath_step_pipe(const double* x, const double* u, const double* d14, const double* p, int f32, double dt,
                        int n_sub, double* x_next)
{
    if (f32) run<float, true>(x, u, d14, p, 0, dt, n_sub, x_next, 0);
    else run<double, true>(x, u, d14, p, 0, dt, n_sub, x_next, 0);
}
int hostmath_step_scheme(const double* x, const double* u, const double* d, const double* p, int f32, double dt,
                         int n_sub, int order, int win, double* x_next, double* stats)
{
#define GL_CASE(O, W)                                                                     \
    if (order == O && win == W) {                                                         \
        if (f32) run_scheme<float, O, W>(x, u, d, p, dt, n_sub, x_next, stats);           \
        else run_scheme<double, O, W>(x, u, d, p, dt, n_sub, x_next, stats);              \
        return 0;                                                                         \
    }
    GL_CASE(4, 1) GL_CASE(4, 2) GL_CASE(4, 3) GL_CASE(4, 4) GL_CASE(2, 1) GL_CASE(2, 2) GL_CASE(2, 4) GL_CASE(3, 1) GL_CASE(3, 3) GL_CASE(3, 4) GL_CASE(5, 1) GL_CASE(5, 2)
#undef GL_CASE
    return -1;
}
int hostmath_step_guarded2(const double* x, const double* u, const double* d, const double* p, int f32, double dt,
                           int n_sub, int order, int win, int verify, double* x_next, double* stats)
{
#define GL_CASE(O, W)                                                                            \
    if (order == O && win == W)                                                                  \
        return f32 ? run_guarded<float, O, W>(x, u, d, p, dt, n_sub, x_next, stats, verify)      \
                   : run_guarded<double, O, W>(x, u, d, p, dt, n_sub, x_next, stats, verify);
    GL_CASE(4, 1) GL_CASE(4, 2) GL_CASE(4, 3) GL_CASE(4, 4) GL_CASE(2, 4) GL_CASE(3, 3) GL_CASE(3, 4) GL_CASE(5, 1) GL_CASE(5, 2)
#undef GL_CASE
    return -1;
}
double hostmath_rate_bound(const double* x, const double* u, const double* d, const double* p, int f32)
{
    return f32 ? run_rate<float>(x, u, d, p) : run_rate<double>(x, u, d, p);
}
void hostmath_step(const double* x, const double* u, const double* d, const double* p, int f32, int per_env_crop,
                   double dt, int n_sub, double* x_next)
{
    if (f32) run<float>(x, u, d, p, per_env_crop, dt, n_sub, x_next, 0);
    else run<double>(x, u, d, p, per_env_crop, dt, n_sub, x_next, 0);
}
}
