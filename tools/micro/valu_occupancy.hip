// Does a SIMD of gfx950 issue vector-ALU instructions faster from two resident wavefronts than from one?  (The ruler of the bench's
// "roofline" block prices every vector instruction at 4 cycles per SIMD.)  256 workgroups (one per CU) of 256 / 512 / 1024 threads =
// exactly 1 / 2 / 4 wavefronts on every SIMD; each wavefront runs K independent chains of n x 16 inline-asm operations (the compiler
// cannot pack or fold them).  Reported: wall time, cycles per instruction PER SIMD = time x clock / (waves per SIMD x K x n x 16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define OPS(X) X(fma32, float, "v_fma_f32 %0, %0, %1, %2") X(mul32, float, "v_mul_f32 %0, %0, %1") X(pkfma32, double, "v_pk_fma_f32 %0, %0, %1, %2") \
    X(fma64, double, "v_fma_f64 %0, %0, %1, %2") X(exp32, float, "v_exp_f32 %0, %0") X(rcp32, float, "v_rcp_f32 %0, %0") \
    X(mov32, float, "v_mov_b32 %0, %1") X(dpp32, float, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")

#define DEF(name, T, ASM)                                                                                              \
    template <int K> __global__ void __launch_bounds__(1024) k_##name(T* out, T b, T c, int n) {                       \
        T a[K];                                                                                                        \
        _Pragma("unroll") for (int k = 0; k < K; ++k) a[k] = (T)(threadIdx.x + k) * (T)1e-3 + (T)1;                    \
        for (int i = 0; i < n; ++i) {                                                                                  \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                           \
                _Pragma("unroll") for (int k = 0; k < K; ++k) asm volatile(ASM : "+v"(a[k]) : "v"(b), "v"(c));         \
            }                                                                                                          \
        }                                                                                                              \
        T s = 0;                                                                                                       \
        _Pragma("unroll") for (int k = 0; k < K; ++k) s += a[k];                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                \
    }
OPS(DEF)

template <typename T, typename F> void run(const char* name, F kern, int K, int grid, int block, double ghz) {
    T* out; const int n = 4000;
    (void)hipMalloc(&out, sizeof(T) * (size_t)grid * block);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kern<<<grid, block>>>(out, (T)0.999999, (T)1e-7, 10);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        kern<<<grid, block>>>(out, (T)0.999999, (T)1e-7, n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double waves_per_simd = (double)grid * block / 64.0 / 1024.0;
    double instr = (double)n * 16 * K;
    printf("%-8s K=%d  grid %4d x %4d (%.0f waves/SIMD)  %.3f ms  %.2f cycles per instruction per SIMD at %.2f GHz  (%.2f per wave)\n", name, K, grid, block, waves_per_simd,
           best, best * 1e-3 * ghz * 1e9 / (instr * waves_per_simd), ghz, best * 1e-3 * ghz * 1e9 / instr);
    (void)hipFree(out);
}

int main(int argc, char** argv) {
    double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    const int blocks[] = {256, 512, 1024};
#define RUN(name, T, ASM)                                                                                    \
    for (int block : blocks) { run<T>(#name, k_##name<1>, 1, 256, block, ghz); run<T>(#name, k_##name<8>, 8, 256, block, ghz); }
    OPS(RUN)
    return 0;
}
