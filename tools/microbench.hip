// Issue-rate microbenchmarks for gfx950 (each launch runs several ms after a same-length warm-up)
// Issue-rate microbenchmarks for gfx950 (what limits a VALU/transcendental-bound, one-lane-per-env kernel).
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o tools/microbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)

constexpr int ITER = 50000;      // x8 unrolled body     // >= 5 ms per launch: shorter kernels measure the DVFS clock ramp, not the issue rate

// per-wave placement / timing record (round 2): where did the dispatcher put each 64-thread block, and how many shader
// cycles did its instruction stream take?  HW_ID (hwreg 4): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12]
// se_id[15:13] ...; XCC_ID (hwreg 20): xcc_id[3:0].
struct WaveRec { unsigned hw_id, xcc_id; unsigned long long t0, t1; };
__device__ WaveRec* g_rec = nullptr;

template <int KIND> __global__ __launch_bounds__(64) void k(float* out, const float* in, int n)
{
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
    __shared__ float lds[256];
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ f4 lds4[KIND == 18 || KIND == 19 ? 8 * 64 : 1];    // KIND 18 / 19: per-lane coefficient tiles, float4 SoA (8 KB per wave)
    float a0 = in[threadIdx.x], a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float c = in[64], d = in[65];
    float vc = c + 0.f * threadIdx.x, vd = d + 0.f * threadIdx.x;     // the same two constants in VGPRs (KIND 14..17)
    asm volatile("" : "+v"(vc), "+v"(vd));
    lds[threadIdx.x] = a0; lds[threadIdx.x + 64] = a1; lds[threadIdx.x + 128] = a2; lds[threadIdx.x + 192] = a3;
    if (KIND == 18 || KIND == 19)
        for (int b = 0; b < 8; ++b) { f4 v = {c, c + 1e-7f * b, c, c - 1e-7f * b}; lds4[b * 64 + threadIdx.x] = v; }
    __syncthreads();
    f4 cur0 = {c, c, c, c}, cur1 = cur0;                              // KIND 18: the coefficients in use, read one block ahead
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pc = {c, c}, pd = {d, d};
    // KIND 10 / 11: the v_fma_f32 body with only the lower 32 / 16 lanes of the wave active (does a half-empty EXEC mask
    // shorten the 4-cycle pass of a wave64 instruction?)
    const bool active = KIND == 10 ? threadIdx.x < 32 : (KIND == 11 ? threadIdx.x < 16 : true);
    if (active)
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int rep = 0; rep < 8; ++rep) {      // 64+ instructions per loop trip: the taken branch costs a lone wave ~40 cycles
        if (KIND == 0 || KIND == 10 || KIND == 11) {          // 8 independent v_fma_f32
            a0 = fmaf(a0, c, d); a1 = fmaf(a1, c, d); a2 = fmaf(a2, c, d); a3 = fmaf(a3, c, d);
            a4 = fmaf(a4, c, d); a5 = fmaf(a5, c, d); a6 = fmaf(a6, c, d); a7 = fmaf(a7, c, d);
        } else if (KIND == 1) {   // 4 independent v_pk_fma_f32 (8 fmas)
            p0 = __builtin_elementwise_fma(p0, pc, pd); p1 = __builtin_elementwise_fma(p1, pc, pd);
            p2 = __builtin_elementwise_fma(p2, pc, pd); p3 = __builtin_elementwise_fma(p3, pc, pd);
        } else if (KIND == 2) {   // 8 independent v_exp_f32
            a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
            a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
        } else if (KIND == 3) {   // 4 fma + 4 exp interleaved
            a0 = fmaf(a0, c, d); a1 = __builtin_amdgcn_exp2f(a1); a2 = fmaf(a2, c, d); a3 = __builtin_amdgcn_exp2f(a3);
            a4 = fmaf(a4, c, d); a5 = __builtin_amdgcn_exp2f(a5); a6 = fmaf(a6, c, d); a7 = __builtin_amdgcn_exp2f(a7);
        } else if (KIND == 4) {   // 8 x (v_readlane + v_fma) : SGPR-spill pattern
            a0 = fmaf(a0, __builtin_amdgcn_readlane(a7, 3), d); a1 = fmaf(a1, __builtin_amdgcn_readlane(a0, 5), d);
            a2 = fmaf(a2, __builtin_amdgcn_readlane(a1, 7), d); a3 = fmaf(a3, __builtin_amdgcn_readlane(a2, 9), d);
            a4 = fmaf(a4, __builtin_amdgcn_readlane(a3, 11), d); a5 = fmaf(a5, __builtin_amdgcn_readlane(a4, 13), d);
            a6 = fmaf(a6, __builtin_amdgcn_readlane(a5, 15), d); a7 = fmaf(a7, __builtin_amdgcn_readlane(a6, 17), d);
        } else if (KIND == 5) {   // 8 x (ds_read broadcast + v_fma): constants staged in LDS
            const volatile float* l = lds;
            a0 = fmaf(a0, l[(i + 0) & 255], d); a1 = fmaf(a1, l[(i + 1) & 255], d); a2 = fmaf(a2, l[(i + 2) & 255], d); a3 = fmaf(a3, l[(i + 3) & 255], d);
            a4 = fmaf(a4, l[(i + 4) & 255], d); a5 = fmaf(a5, l[(i + 5) & 255], d); a6 = fmaf(a6, l[(i + 6) & 255], d); a7 = fmaf(a7, l[(i + 7) & 255], d);
        } else if (KIND == 6) {   // 8 x (per-lane ds_read + v_fma): coefficients staged in LDS
            const volatile float* l = lds;
            const int t = threadIdx.x;
            a0 = fmaf(a0, l[t], d); a1 = fmaf(a1, l[t + 64], d); a2 = fmaf(a2, l[t + 128], d); a3 = fmaf(a3, l[t + 192], d);
            a4 = fmaf(a4, l[t], d); a5 = fmaf(a5, l[t + 64], d); a6 = fmaf(a6, l[t + 128], d); a7 = fmaf(a7, l[t + 192], d);
        } else if (KIND == 18) {  // 8 x v_fma with PER-LANE coefficients from LDS: two ds_read_b128 per eight, issued one block ahead
            const f4 nx0 = lds4[((2 * rep) & 7) * 64 + threadIdx.x], nx1 = lds4[((2 * rep + 1) & 7) * 64 + threadIdx.x];
            a0 = fmaf(a0, cur0.x, d); a1 = fmaf(a1, cur0.y, d); a2 = fmaf(a2, cur0.z, d); a3 = fmaf(a3, cur0.w, d);
            a4 = fmaf(a4, cur1.x, d); a5 = fmaf(a5, cur1.y, d); a6 = fmaf(a6, cur1.z, d); a7 = fmaf(a7, cur1.w, d);
            cur0 = nx0; cur1 = nx1;
        } else if (KIND == 19) {  // the same without the look-ahead: each block of eight waits for its own two ds_read_b128
            const f4 nx0 = lds4[((2 * rep) & 7) * 64 + threadIdx.x], nx1 = lds4[((2 * rep + 1) & 7) * 64 + threadIdx.x];
            a0 = fmaf(a0, nx0.x, d); a1 = fmaf(a1, nx0.y, d); a2 = fmaf(a2, nx0.z, d); a3 = fmaf(a3, nx0.w, d);
            a4 = fmaf(a4, nx1.x, d); a5 = fmaf(a5, nx1.y, d); a6 = fmaf(a6, nx1.z, d); a7 = fmaf(a7, nx1.w, d);
        } else if (KIND == 7) {   // dependent fma chain (latency)
            a0 = fmaf(a0, c, d); a0 = fmaf(a0, c, d); a0 = fmaf(a0, c, d); a0 = fmaf(a0, c, d);
            a0 = fmaf(a0, c, d); a0 = fmaf(a0, c, d); a0 = fmaf(a0, c, d); a0 = fmaf(a0, c, d);
        } else if (KIND == 8) {   // 8 independent v_rcp_f32
            a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3);
            a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7);
        } else if (KIND == 9) {   // 8 x v_mul_f32 with an SGPR operand (uniform constant resident in SGPR)
            a0 *= c; a1 *= c; a2 *= c; a3 *= c; a4 *= c; a5 *= c; a6 *= c; a7 *= c;
        } else if (KIND == 14) {  // encodings: 8 x v_fmac_f32_e32 (VOP2, 4 bytes; VGPR operands only)
            asm volatile("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n"
                         "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(vc), "v"(vd));
        } else if (KIND == 15) {  // 8 x v_add_f32_e32 (VOP2, 4 bytes)
            asm volatile("v_add_f32_e32 %0, %8, %0\n v_add_f32_e32 %1, %8, %1\n v_add_f32_e32 %2, %8, %2\n v_add_f32_e32 %3, %8, %3\n"
                         "v_add_f32_e32 %4, %8, %4\n v_add_f32_e32 %5, %8, %5\n v_add_f32_e32 %6, %8, %6\n v_add_f32_e32 %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(vd));
        } else if (KIND == 16) {  // 8 x v_add_f32_e64 (VOP3, 8 bytes; same operation)
            asm volatile("v_add_f32_e64 %0, %8, %0\n v_add_f32_e64 %1, %8, %1\n v_add_f32_e64 %2, %8, %2\n v_add_f32_e64 %3, %8, %3\n"
                         "v_add_f32_e64 %4, %8, %4\n v_add_f32_e64 %5, %8, %5\n v_add_f32_e64 %6, %8, %6\n v_add_f32_e64 %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(vd));
        } else if (KIND == 17) {  // 8 x v_add_f32_e32 with a 32-bit literal (VOP2 + literal, 8 bytes)
            asm volatile("v_add_f32_e32 %0, 0x3a83126f, %0\n v_add_f32_e32 %1, 0x3a83126f, %1\n v_add_f32_e32 %2, 0x3a83126f, %2\n"
                         "v_add_f32_e32 %3, 0x3a83126f, %3\n v_add_f32_e32 %4, 0x3a83126f, %4\n v_add_f32_e32 %5, 0x3a83126f, %5\n"
                         "v_add_f32_e32 %6, 0x3a83126f, %6\n v_add_f32_e32 %7, 0x3a83126f, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 12) {  // 8 x (cross-lane move from the neighbouring lane + v_fma): what splitting one env over two lanes pays
            a0 = fmaf(a0, c, __shfl_xor(a7, 1)); a1 = fmaf(a1, c, __shfl_xor(a0, 1)); a2 = fmaf(a2, c, __shfl_xor(a1, 1));
            a3 = fmaf(a3, c, __shfl_xor(a2, 1)); a4 = fmaf(a4, c, __shfl_xor(a3, 1)); a5 = fmaf(a5, c, __shfl_xor(a4, 1));
            a6 = fmaf(a6, c, __shfl_xor(a5, 1)); a7 = fmaf(a7, c, __shfl_xor(a6, 1));
        } else if (KIND == 13) {  // the same exchange as a DPP move (quad_perm [1,0,3,2]: lane ^ 1), the cheapest cross-lane path
            auto nb = [](float v) {
                return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
            };
            a0 = fmaf(a0, c, nb(a7)); a1 = fmaf(a1, c, nb(a0)); a2 = fmaf(a2, c, nb(a1)); a3 = fmaf(a3, c, nb(a2));
            a4 = fmaf(a4, c, nb(a3)); a5 = fmaf(a5, c, nb(a4)); a6 = fmaf(a6, c, nb(a5)); a7 = fmaf(a7, c, nb(a6));
        }
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + cur0.x + cur1.w + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (g_rec && threadIdx.x == 0) {
        WaveRec r;
        r.hw_id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
        r.xcc_id = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
        r.t0 = t_begin; r.t1 = __builtin_amdgcn_s_memtime();
        g_rec[blockIdx.x] = r;
    }
}

// Role split over two waves of one workgroup (upper / lower compartment): per stage each wave publishes 8 values per lane in
// LDS, both meet at a barrier, each reads the partner's 8 values.  Time per exchange round, with `work` dependent FMAs between
// rounds standing in for the half stage each wave would compute.
__global__ __launch_bounds__(128) void exch(float* out, const float* in, int rounds, int work)
{
    __shared__ float box[2][8][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = in[lane] + j;
    const float c = in[64], d = in[65];
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < work; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], c, d);
#pragma unroll
        for (int j = 0; j < 8; ++j) box[w][j][lane] = v[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += box[1 - w][j][lane];
        __syncthreads();
    }
    float acc = 0;
    for (int j = 0; j < 8; ++j) acc += v[j];
    out[blockIdx.x * 128 + threadIdx.x] = acc;
}

int run_exchange(float* out, float* in)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int rounds = 20000;
    for (int wps : {2, 4}) {
        const int blocks = 256 * 4 * wps / 2;
        float t[2];
        for (int k = 0; k < 2; ++k) {
            const int work = k == 0 ? 0 : 20;                      // 160 dependent FMAs per wave per round ~ half a stage
            hipLaunchKernelGGL(exch, dim3(blocks), dim3(128), 0, 0, out, in, rounds, work);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(exch, dim3(blocks), dim3(128), 0, 0, out, in, rounds, work);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&t[k], e0, e1));
        }
        printf("two-wave LDS exchange (8 values/lane each way + 2 barriers) waves/SIMD=%d: %.1f ns = %.0f cycles @2.27GHz per round bare; "
               "with 160 FMAs of work per wave per round %.1f ns = %.0f cycles (the work alone: 160 x issue cost)\n", wps,
               t[0] * 1e6 / rounds, t[0] * 1e6 / rounds * 2.27, t[1] * 1e6 / rounds, t[1] * 1e6 / rounds * 2.27);
    }
    return 0;
}

template <int KIND> int run(const char* name, float* out, float* in, int ops_per_iter)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    int n_cu = 256;
    static WaveRec* rec_dev = nullptr;
    static std::vector<WaveRec> rec_host;
    if (!rec_dev) {
        CHK(hipMalloc(&rec_dev, sizeof(WaveRec) * 256 * 4 * 8));
        CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_rec), &rec_dev, sizeof rec_dev));
        rec_host.resize(256 * 4 * 8);
    }
    for (int wps : {1, 2, 4, 8}) {            // waves per SIMD
        const int blocks = n_cu * 4 * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, in, ITER);     // warm-up at full length
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, in, ITER);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        const double wave_instr = (double)ITER * 8 * ops_per_iter;          // per wave
        const double ns_per_instr_per_simd = ms * 1e6 / (wave_instr * wps);  // time per wave-instruction on one SIMD
        // placement: how many waves did each (xcc, se, cu, simd) actually get, and shader cycles per own instruction
        CHK(hipMemcpy(rec_host.data(), rec_dev, sizeof(WaveRec) * blocks, hipMemcpyDeviceToHost));
        std::vector<int> per_simd(8 * 8 * 16 * 4 * 2, 0);
        double cyc = 0; unsigned long long tmin = ~0ull, tmax = 0;
        for (int b = 0; b < blocks; ++b) {
            const WaveRec& r = rec_host[b];
            const int simd = (r.hw_id >> 4) & 3, cu = (r.hw_id >> 8) & 15, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7;
            per_simd[(((((r.xcc_id & 7) * 8 + se) * 2 + sh) * 16 + cu) * 4) + simd]++;
            cyc += (double)(r.t1 - r.t0);
            if (r.t0 < tmin) tmin = r.t0;
            if (r.t1 > tmax) tmax = r.t1;
        }
        int used = 0, mx = 0, mn = 1 << 30; int hist[17] = {0};
        for (int v : per_simd) if (v) { ++used; if (v > mx) mx = v; if (v < mn) mn = v; hist[v > 16 ? 16 : v]++; }
        const double cyc_per_instr = cyc / blocks / wave_instr;          // shader (s_memtime = 100 MHz? see below) ticks
        printf("%-34s waves/SIMD=%d  %8.3f ms  -> %.3f ns per wave-instr per SIMD (%.2f cycles @2.27GHz) | placement: %d SIMDs used, "
               "waves per used SIMD min %d max %d (hist 1:%d 2:%d 3:%d 4:%d 5:%d 6:%d 7:%d 8:%d) | s_memtime ticks per own wave-instr "
               "%.3f, span %.3f ms at 100 MHz\n", name, wps, ms, ns_per_instr_per_simd, ns_per_instr_per_simd * 2.27, used, mn, mx,
               hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], hist[8], cyc_per_instr, (double)(tmax - tmin) / 1e5);
    }
    return 0;
}

int main()
{
    float *out, *in;
    CHK(hipMalloc(&out, 256 * 4 * 16 * 64 * sizeof(float)));
    CHK(hipMalloc(&in, 128 * sizeof(float)));
    std::vector<float> h(128, 0.999f); h[64] = 0.9999f; h[65] = 1e-3f;
    CHK(hipMemcpy(in, h.data(), 128 * sizeof(float), hipMemcpyHostToDevice));
    run<0>("v_fma_f32 x8", out, in, 8);
    run<1>("v_pk_fma_f32 x4 (=8 fma)", out, in, 4);
    run<2>("v_exp_f32 x8", out, in, 8);
    run<8>("v_rcp_f32 x8", out, in, 8);
    run<3>("4 fma + 4 exp", out, in, 8);
    run<9>("v_mul_f32 sgpr-operand x8", out, in, 8);
    run<4>("(v_readlane + v_fma) x8", out, in, 16);
    run<5>("(ds_read bcast + v_fma) x8", out, in, 16);
    run<6>("(ds_read per-lane + v_fma) x8", out, in, 16);
    run<18>("8 v_fma + 2 ds_read_b128 ahead (per 8 fma)", out, in, 8);
    run<19>("8 v_fma + 2 ds_read_b128 in place", out, in, 8);
    run<7>("dependent v_fma chain x8", out, in, 8);
    run<12>("(ds_bpermute lane^1 + v_fma) x8", out, in, 16);
    run<13>("(DPP quad_perm lane^1 + v_fma) x8", out, in, 16);
    run_exchange(out, in);
    run<14>("v_fmac_f32_e32 x8 (4-byte encoding)", out, in, 8);
    run<15>("v_add_f32_e32 x8 (4-byte encoding)", out, in, 8);
    run<16>("v_add_f32_e64 x8 (8-byte encoding)", out, in, 8);
    run<17>("v_add_f32_e32 + literal x8 (8 bytes)", out, in, 8);
    if (getenv("MICROBENCH_ALL")) {
        run<10>("v_fma_f32 x8, 32 of 64 lanes active", out, in, 8);
        run<11>("v_fma_f32 x8, 16 of 64 lanes active", out, in, 8);
    }
    return 0;
}
