// mixed_peak.hip -- can the f32 matrix pipe and the f32 vector pipe run at the same time?
// Workgroups of 256 threads, four per CU; even workgroups issue v_mfma_f32_32x32x2_f32 chains, odd ones packed
// v_pk_fma_f32 chains (registers only).  Modes: MFMA only, VALU only, both.
// Build: hipcc --offload-arch=gfx950 -O3 -o mixed_peak mixed_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>   // 1 = mfma waves work, 2 = valu waves work, 3 = both
__global__ __launch_bounds__(256) void k_mixed(float* out, int iters, float a0, float b0) {
    const bool is_mfma = ((blockIdx.x >> 8) & 1) == 0;   // blocks go round-robin to 8 XCDs x 32 CUs: roles alternate per 256 blocks so that every CU hosts both
    float s = 0.f;
    if (is_mfma) {
        if (MODE & 1) {
            f32x16 acc[8];
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
            float a = a0 + threadIdx.x * 1e-6f, b = b0;
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) s += acc[i][r];
        }
    } else {
        if (MODE & 2) {
            // per iteration: 8 MFMA (8 x 64 cycles = 512 cycles on the other waves) ~ 128 pk_fma (4 cycles each)
            f32x2 acc[32];
#pragma unroll
            for (int i = 0; i < 32; i++) acc[i] = f32x2{0.f, 0.f};
            f32x2 a = {a0 + threadIdx.x * 1e-6f, a0}, b = {b0, b0 * 0.5f};
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int i = 0; i < 32; i++) acc[i] = __builtin_elementwise_fma(a, b, acc[i]);
            }
#pragma unroll
            for (int i = 0; i < 32; i++) s += acc[i][0] + acc[i][1];
        }
    }
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int ncu, float* out) {
    const int iters = 8192;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(ncu * 4), block(256);     // 4 workgroups per CU: on average every SIMD holds two MFMA and two VALU waves
    hipLaunchKernelGGL(k_mixed<MODE>, grid, block, 0, 0, out, 64, 1.0f, 1e-3f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_mixed<MODE>, grid, block, 0, 0, out, iters, 1.0f, 1e-3f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double mf = (MODE & 1) ? (double)(grid.x / 2) * 4 * iters * 8.0 * 4096.0 : 0.0;
    double vf = (MODE & 2) ? (double)(grid.x / 2) * 4 * 64 * iters * 4.0 * 32.0 * 4.0 : 0.0;   // pk_fma = 4 flop per lane
    printf("%-12s %8.3f ms   matrix %7.1f TFLOP/s   vector %7.1f TFLOP/s   total %7.1f\n", name, ms, mf / ms / 1e9, vf / ms / 1e9,
           (mf + vf) / ms / 1e9);
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    int ncu = prop.multiProcessorCount;
    float* out;
    (void)hipMalloc(&out, 1 << 24);
    run<1>("MFMA only", ncu, out);
    run<2>("VALU only", ncu, out);
    run<3>("both", ncu, out);
    run<3>("both (rep)", ncu, out);
    (void)hipFree(out);
    return 0;
}
