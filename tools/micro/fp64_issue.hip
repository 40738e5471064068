// Issue rate and dependent latency of the vector ALU's fp64 / fp32 operations on gfx950, one wavefront per SIMD (the regime of the
// four-lanes-per-environment fp64 kernel at B <= 4096: 256 of 1024 SIMDs, one wave each).   Build + run: tools/micro/run_fp64_issue.sh
//   K independent chains of N dependent operations each; cycles per instruction = s_memtime delta / (K * N).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

template <typename T> struct Op_fma { static __device__ T f(T a, T b, T c) { return __builtin_fma(a, b, c); } };
template <> struct Op_fma<float> { static __device__ float f(float a, float b, float c) { return __builtin_fmaf(a, b, c); } };
template <typename T> struct Op_mul { static __device__ T f(T a, T b, T) { return a * b; } };
template <typename T> struct Op_add { static __device__ T f(T a, T, T c) { return a + c; } };
template <typename T> struct Op_rcp {};
template <> struct Op_rcp<double> { static __device__ double f(double a, double, double) { return __builtin_amdgcn_rcp(a); } };
template <> struct Op_rcp<float> { static __device__ float f(float a, float, float) { return __builtin_amdgcn_rcpf(a); } };
template <typename T> struct Op_max { static __device__ T f(T a, T b, T) { return a > b ? a : b; } };
template <typename T> struct Op_dpp {};
template <> struct Op_dpp<double> {
    static __device__ double f(double a, double, double) {
        unsigned long long u = __builtin_bit_cast(unsigned long long, a);
        int lo = __builtin_amdgcn_mov_dpp((int)u, 0xB1, 0xF, 0xF, true), hi = __builtin_amdgcn_mov_dpp((int)(u >> 32), 0xB1, 0xF, 0xF, true);
        return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    }
};
template <> struct Op_dpp<float> {
    static __device__ float f(float a, float, float) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true)); }
};

// inline-asm forms: the compiler cannot pair these into packed instructions
template <typename T> struct Op_fma_asm {};
template <> struct Op_fma_asm<float> { static __device__ float f(float a, float b, float c) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); return a; } };
template <> struct Op_fma_asm<double> { static __device__ double f(double a, double b, double c) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); return a; } };
template <typename T> struct Op_mul_asm {};
template <> struct Op_mul_asm<float> { static __device__ float f(float a, float b, float) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b)); return a; } };
template <typename T> struct Op_exp_asm {};
template <> struct Op_exp_asm<float> { static __device__ float f(float a, float, float) { asm volatile("v_exp_f32 %0, %0" : "+v"(a)); return a; } };
template <typename T> struct Op_cnd_asm {};
template <> struct Op_cnd_asm<float> { static __device__ float f(float a, float b, float) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b)); return a; } };

template <typename T, template <typename> class OP, int K>
__global__ void __launch_bounds__(64) chain(T* out, long long* cyc, T b, T c, int n) {
    T a[K];
#pragma unroll
    for (int k = 0; k < K; ++k) a[k] = (T)(threadIdx.x + k) * (T)1e-3 + (T)1;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int k = 0; k < K; ++k) a[k] = OP<T>::f(a[k], b, c);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    T s = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) s += a[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename T, template <typename> class OP, int K> void run(const char* name, int blocks, T b, T c) {
    T* out; long long* cyc; const int n = 2000;
    hipMalloc(&out, sizeof(T) * 64 * blocks); hipMalloc(&cyc, 8 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    chain<T, OP, K><<<blocks, 64>>>(out, cyc, b, c, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chain<T, OP, K><<<blocks, 64>>>(out, cyc, b, c, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    double ops = (double)n * 16 * K;
    printf("%-10s %-4s K=%d  blocks %5d  %.3f ms  %.2f ns/op-per-wave  counter ticks/op %.3f\n", name, sizeof(T) == 8 ? "f64" : "f32", K, blocks, ms, ms * 1e6 / ops,
           (double)h[0] / ops);
    hipFree(out); hipFree(cyc);
}

#define ALLK(T, OP, name, b, c) run<T, OP, 1>(name, blocks, b, c); run<T, OP, 2>(name, blocks, b, c); run<T, OP, 4>(name, blocks, b, c); run<T, OP, 8>(name, blocks, b, c);

int main(int argc, char** argv) {
    int blocks = argc > 1 ? atoi(argv[1]) : 256;
    printf("== %d workgroups of one wavefront ==\n", blocks);
    ALLK(double, Op_fma, "fma", 0.999999, 1e-7)
    ALLK(double, Op_mul, "mul", 0.999999, 0.0)
    ALLK(double, Op_add, "add", 0.0, 1e-7)
    ALLK(double, Op_max, "max", 0.5, 0.0)
    ALLK(double, Op_rcp, "rcp", 0.0, 0.0)
    ALLK(double, Op_dpp, "dpp-mov", 0.0, 0.0)
    ALLK(double, Op_fma_asm, "fma-asm", 0.999999, 1e-7)
    ALLK(float, Op_fma_asm, "fma-asm", 0.999999f, 1e-7f)
    ALLK(float, Op_mul_asm, "mul-asm", 0.999999f, 0.0f)
    ALLK(float, Op_exp_asm, "exp-asm", 0.0f, 0.0f)
    ALLK(float, Op_cnd_asm, "cndmask", 0.5f, 0.0f)
    ALLK(float, Op_fma, "fma", 0.999999f, 1e-7f)
    ALLK(float, Op_mul, "mul", 0.999999f, 0.0f)
    ALLK(float, Op_rcp, "rcp", 0.0f, 0.0f)
    ALLK(float, Op_dpp, "dpp-mov", 0.0f, 0.0f)
    return 0;
}
