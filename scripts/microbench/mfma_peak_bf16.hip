// mfma_peak_bf16.hip -- measured ceiling of v_mfma_f32_32x32x16_bf16 on this GPU (operands in registers, no memory
// traffic in the loop): every wave issues NACC independent accumulator chains back to back, on RANDOM operand bits and,
// for comparison, on zeros (MI355X lowers its clock under an MFMA-dense load on non-trivial data).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak_bf16 mfma_peak_bf16.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(const bf16x8* __restrict__ ops, float* out, int iters) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { a[u] = ops[(threadIdx.x * 8 + u) & 4095]; b[u] = ops[(threadIdx.x * 8 + 4 + u) & 4095]; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[(u + i) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += acc[i][r];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// the same flop count per iteration with v_mfma_f32_16x16x32_bf16 (4 x NACC accumulators of 4 registers)
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma16(const bf16x8* __restrict__ ops, float* out, int iters) {
    f32x4 acc[NACC * 2];
#pragma unroll
    for (int i = 0; i < NACC * 2; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[i][r] = 0.f;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { a[u] = ops[(threadIdx.x * 8 + u) & 4095]; b[u] = ops[(threadIdx.x * 8 + 4 + u) & 4095]; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < NACC * 2; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b[(u + i) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC * 2; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) s += acc[i][r];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
static void run16(const char* name, int wg_per_cu, int ncu, const bf16x8* ops, float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(ncu * wg_per_cu), block(256);
    hipLaunchKernelGGL(k_mfma16<NACC>, grid, block, 0, 0, ops, out, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma16<NACC>, grid, block, 0, 0, ops, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid.x * 4 * iters * 4.0 * NACC * 2 * 16384.0;
    printf("%-40s grid %5d  %8.3f ms  %7.1f TFLOP/s\n", name, grid.x, ms, flops / ms / 1e9);
}

template <int NACC>
static void run(const char* name, int wg_per_cu, int ncu, const bf16x8* ops, float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(ncu * wg_per_cu), block(256);
    hipLaunchKernelGGL(k_mfma<NACC>, grid, block, 0, 0, ops, out, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<NACC>, grid, block, 0, 0, ops, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid.x * 4 /*waves*/ * iters * 4.0 * NACC * 32768.0;
    printf("%-40s grid %5d  %8.3f ms  %7.1f TFLOP/s\n", name, grid.x, ms, flops / ms / 1e9);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int ncu = prop.multiProcessorCount;
    printf("%s: %d CUs, clock %d MHz\n", prop.gcnArchName, ncu, prop.clockRate / 1000);
    float* out;
    hipMalloc(&out, 1 << 24);
    std::vector<unsigned short> h(4096 * 8);
    srand(7);
    for (auto& v : h) {                       // random bf16 in about [-2, 2): random sign and mantissa, exponent 120..127
        unsigned m = rand() & 0x7f, e = 120 + (rand() & 7), s = rand() & 1;
        v = (unsigned short)((s << 15) | (e << 7) | m);
    }
    bf16x8 *rnd, *zero;
    hipMalloc(&rnd, h.size() * 2); hipMalloc(&zero, h.size() * 2);
    hipMemcpy(rnd, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemset(zero, 0, h.size() * 2);
    for (int rep = 0; rep < 3; rep++) {
        run<4>("random, 4 acc/wave, 1 wave/SIMD", 1, ncu, rnd, out, 16384);
        run<4>("random, 4 acc/wave, 2 waves/SIMD", 2, ncu, rnd, out, 16384);
        run<2>("random, 2 acc/wave, 4 waves/SIMD", 4, ncu, rnd, out, 16384);
        run<4>("zeros,  4 acc/wave, 2 waves/SIMD", 2, ncu, zero, out, 16384);
        run16<4>("random, 16x16x32, 8 acc/wave, 2 waves/SIMD", 2, ncu, rnd, out, 16384);
        run16<2>("random, 16x16x32, 4 acc/wave, 4 waves/SIMD", 4, ncu, rnd, out, 16384);
    }
    return 0;
}
