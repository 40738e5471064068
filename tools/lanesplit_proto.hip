// Prototype of the north-star's "split one environment over lanes" idea on its BEST-CASE block: the long-wave (FIR)
// exchange network of the GreenLight RHS -- 28 pair terms c_ij (T_i^4 - T_j^4) between 8 surfaces + sky -- which is the most
// regular part of a stage (everything else is less regular: DESIGN.md section 5).  Measured per environment-evaluation:
//   (a) one lane per environment, the block as the product has it (register pairs, v_pk_*), 65 536 envs = 1 wave per SIMD;
//   (b) two lanes per environment: each lane owns 4 surfaces, computes their T^4, fetches the partner's four with DPP
//       (quad_perm lane^1), evaluates 14 of the 28 pair terms, and the 8 partial sums are combined with 8 more DPP moves
//       -- 131 072 lanes = 2 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/lanesplit_proto.hip -o tools/lanesplit_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float nb(float v)       // value of lane ^ 1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}

// 28 pairs over nodes 0..7 (+ node 8 = sky): every pair i < j of the 8 surfaces is 28 pairs; keep it generic
__global__ __launch_bounds__(64) void fir_one_lane(float* out, const float* in, int iters)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    float T[8], c[28];
    for (int i = 0; i < 8; ++i) T[i] = in[i] + 1e-3f * (b & 255);
    for (int i = 0; i < 28; ++i) c[i] = in[8 + i];
    for (int it = 0; it < iters; ++it) {
        float q[8], net[8];
#pragma unroll
        for (int i = 0; i < 8; i += 2) {          // packed T^4
            f2 k = {T[i] + 273.15f, T[i + 1] + 273.15f}; f2 k2 = k * k; f2 k4 = k2 * k2; q[i] = k4.x; q[i + 1] = k4.y;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) net[i] = 0.f;
        int p = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = i + 1; j < 8; ++j) { const float r = c[p++] * (q[i] - q[j]); net[i] -= r; net[j] += r; }
#pragma unroll
        for (int i = 0; i < 8; ++i) T[i] += 1e-12f * net[i];
    }
    float acc = 0; for (int i = 0; i < 8; ++i) acc += T[i];
    out[b] = acc;
}

__global__ __launch_bounds__(64) void fir_two_lanes(float* out, const float* in, int iters)
{
    const int lane = blockIdx.x * 64 + threadIdx.x, b = lane >> 1, par = lane & 1;
    // lane parity 0 owns surfaces 0..3, parity 1 owns 4..7.  Pair (i, j): both in own set -> the owner does it (6 + 6);
    // cross pairs (16) are split 8 / 8: each lane takes (own i, partner's (i + d) & 3) for d = 0, 1 -- with the odd lane's
    // surfaces stored rotated by one the two halves are disjoint, and the index pattern is compile-time in both lanes.
    float T[4], cown[6], ccross[8];
    for (int i = 0; i < 4; ++i) T[i] = in[4 * par + i] + 1e-3f * (b & 255);
    for (int i = 0; i < 6; ++i) cown[i] = in[8 + 6 * par + i];
    for (int i = 0; i < 8; ++i) ccross[i] = in[20 + i];
    for (int it = 0; it < iters; ++it) {
        float q[4], qo[4], net[4], neto[4];
        {
            f2 k = {T[0] + 273.15f, T[1] + 273.15f}; f2 k2 = k * k; f2 k4 = k2 * k2; q[0] = k4.x; q[1] = k4.y;
            f2 l = {T[2] + 273.15f, T[3] + 273.15f}; f2 l2 = l * l; f2 l4 = l2 * l2; q[2] = l4.x; q[3] = l4.y;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { qo[i] = nb(q[i]); net[i] = 0.f; neto[i] = 0.f; }     // partner's T^4 (4 DPP moves)
        int p = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = i + 1; j < 4; ++j) { const float r = cown[p++] * (q[i] - q[j]); net[i] -= r; net[j] += r; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int j = (i + d) & 3;
                const float r = ccross[2 * i + d] * (q[i] - qo[j]); net[i] -= r; neto[j] += r;
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) net[i] += nb(neto[i]);                                    // partner's share (4 DPP + 4 adds)
#pragma unroll
        for (int i = 0; i < 4; ++i) T[i] += 1e-12f * net[i];
    }
    float acc = 0; for (int i = 0; i < 4; ++i) acc += T[i];
    out[lane] = acc;
}

int main()
{
    float *out, *in;
    CHK(hipMalloc(&out, 131072 * 4 * sizeof(float)));
    CHK(hipMalloc(&in, 64 * sizeof(float)));
    std::vector<float> h(64);
    for (int i = 0; i < 64; ++i) h[i] = i < 8 ? 10.f + i : 1e-9f * (1 + i);
    CHK(hipMemcpy(in, h.data(), 64 * sizeof(float), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 200000, envs = 65536;
    float ms[2];
    for (int k = 0; k < 2; ++k) {
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipEventRecord(e0));
            if (k == 0) hipLaunchKernelGGL(fir_one_lane, dim3(envs / 64), dim3(64), 0, 0, out, in, iters);
            else hipLaunchKernelGGL(fir_two_lanes, dim3(2 * envs / 64), dim3(64), 0, 0, out, in, iters);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&ms[k], e0, e1));
        }
    }
    printf("FIR block, 65 536 environments x %d evaluations:\n", iters);
    printf("  (a) one lane per env   (1 wave/SIMD):  %8.2f ms  = %.1f ns per evaluation per SIMD-wave\n", ms[0], ms[0] * 1e6 / iters);
    printf("  (b) two lanes per env  (2 waves/SIMD): %8.2f ms  = %.1f ns per evaluation\n", ms[1], ms[1] * 1e6 / iters);
    printf("  speed-up of the split: %.2fx\n", ms[0] / ms[1]);
    return 0;
}
