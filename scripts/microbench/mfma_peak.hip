// mfma_peak.hip -- measured ceiling of v_mfma_f32_32x32x2_f32 on this GPU (no memory traffic):
// every wave issues NACC independent accumulator chains back to back.  Prints TFLOP/s for 1, 2 and 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    // a0 < 0: RANDOM operand bits (a hash of the lane, values in about [-2, 2)) -- the part lowers its clock under an
    // MFMA-dense load on non-trivial data; otherwise the near-constant operands of the first version of this benchmark
    float av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        if (a0 < 0.f) {
            unsigned h = (threadIdx.x * 2654435761u) ^ (u * 0x9e3779b9u) ^ (blockIdx.x * 40503u);
            h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
            av[u] = __uint_as_float(0x3f000000u | (h & 0x80ffffffu)) ;          // +-[0.5, 2)
            h *= 0x297a2d39u; h ^= h >> 15;
            bv[u] = __uint_as_float(0x3f000000u | (h & 0x80ffffffu));
        } else { av[u] = a0 + threadIdx.x * 1e-6f; bv[u] = b0; }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[(u + i) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += acc[i][r];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
static void run(const char* name, int wg_per_cu, int ncu, float* out, float a0 = 1.0f) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(ncu * wg_per_cu), block(256);
    hipLaunchKernelGGL(k_mfma<NACC>, grid, block, 0, 0, out, 64, a0, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<NACC>, grid, block, 0, 0, out, iters, a0, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid.x * 4 /*waves*/ * iters * 4.0 * NACC * 4096.0;
    printf("%-28s grid %5d  %8.3f ms  %7.1f TFLOP/s\n", name, grid.x, ms, flops / ms / 1e9);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    int ncu = prop.multiProcessorCount;
    printf("%s: %d CUs, clock %d MHz\n", prop.gcnArchName, ncu, prop.clockRate / 1000);
    float* out;
    hipMalloc(&out, 1 << 24);
    run<4>("4 acc/wave, 1 wave/SIMD", 1, ncu, out);
    run<4>("4 acc/wave, 2 waves/SIMD", 2, ncu, out);
    run<4>("4 acc/wave, 4 waves/SIMD", 4, ncu, out);
    run<10>("10 acc/wave, 1 wave/SIMD", 1, ncu, out);
    run<10>("10 acc/wave, 2 waves/SIMD", 2, ncu, out);
    run<2>("2 acc/wave, 2 waves/SIMD", 2, ncu, out);
    run<1>("1 acc/wave, 4 waves/SIMD", 4, ncu, out);
    // long run: sustained clocks
    for (int rep = 0; rep < 3; rep++) run<10>("10 acc/wave, 2 waves (rep)", 8, ncu, out);
    for (int rep = 0; rep < 3; rep++) {
        run<4>("RANDOM, 4 acc/wave, 2 waves/SIMD", 2, ncu, out, -1.f);
        run<2>("RANDOM, 2 acc/wave, 2 waves/SIMD", 2, ncu, out, -1.f);
        run<10>("RANDOM, 10 acc/wave, 2 waves (long)", 8, ncu, out, -1.f);
    }
    hipFree(out);
    return 0;
}
