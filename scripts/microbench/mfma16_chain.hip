// mfma16_chain.hip -- is v_mfma_f32_16x16x4_f32 the same k-ascending float32 fmaf chain as v_mfma_f32_32x32x2_f32?
// (a 16x16x4 tile walks 4 k per 8-pass instruction, i.e. a 4x shorter dependent chain for the
// attack's small launches -- usable only if every output is still acc = fmaf(a_k, b_k, acc) for k ascending, the
// order the oracle and the big kernels use.)  Prints, for random operands, how many of the 256 / 1024 outputs equal the
// CPU chain bit for bit, and which alternative summation orders match if that one does not.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o mfma16_chain mfma16_chain.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int K = 64;

__global__ void k16(const float* A /*[16][K]*/, const float* B /*[K][16]*/, float* D /*[16][16]*/) {
    const int l = threadIdx.x, ij = l & 15, kq = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[ij * K + k0 + kq], B[(k0 + kq) * 16 + ij], acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(4 * kq + r) * 16 + ij] = acc[r];
}
__global__ void k32(const float* A /*[32][K]*/, const float* B /*[K][32]*/, float* D /*[32][32]*/) {
    const int l = threadIdx.x, ij = l & 31, kh = l >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[ij * K + k0 + kh], B[(k0 + kh) * 32 + ij], acc, 0, 0, 0);
    for (int r = 0; r < 16; r++) D[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + ij] = acc[r];
}

static float frand() { return (float)((rand() % 2000001) - 1000000) * 1.7e-6f * (1.0f + (rand() % 7)); }

int main() {
    srand(12345);
    for (int n : {16, 32}) {
        float *hA = (float*)malloc(4 * n * K), *hB = (float*)malloc(4 * K * n), *hD = (float*)malloc(4 * n * n);
        for (int i = 0; i < n * K; i++) hA[i] = frand();
        for (int i = 0; i < K * n; i++) hB[i] = frand();
        float *dA, *dB, *dD;
        hipMalloc(&dA, 4 * n * K); hipMalloc(&dB, 4 * K * n); hipMalloc(&dD, 4 * n * n);
        hipMemcpy(dA, hA, 4 * n * K, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 4 * K * n, hipMemcpyHostToDevice);
        if (n == 16) hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        else hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(hD, dD, 4 * n * n, hipMemcpyDeviceToHost);
        int chain = 0, pair = 0, dot4 = 0, rev = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                float c = 0.f, p = 0.f, d4 = 0.f, rv = 0.f;
                for (int k = 0; k < K; k++) c = fmaf(hA[i * K + k], hB[k * n + j], c);          // the oracle's order
                for (int k = 0; k < K; k += 4) {                                                 // alternatives per instruction
                    const float* a = hA + i * K + k;
                    double e = 0;
                    for (int q = 0; q < 4; q++) e += (double)a[q] * (double)hB[(k + q) * n + j];
                    d4 = (float)((double)d4 + e);                                                // exact 4-term dot, one rounding
                    float s01 = fmaf(a[1], hB[(k + 1) * n + j], a[0] * hB[k * n + j]);
                    float s23 = fmaf(a[3], hB[(k + 3) * n + j], a[2] * hB[(k + 2) * n + j]);
                    p = p + (s01 + s23);                                                         // pairwise tree
                    for (int q = 3; q >= 0; q--) rv = fmaf(a[q], hB[(k + q) * n + j], rv);       // descending inside the instruction
                }
                float g = hD[i * n + j];
                chain += memcmp(&g, &c, 4) == 0; pair += memcmp(&g, &p, 4) == 0;
                dot4 += memcmp(&g, &d4, 4) == 0; rev += memcmp(&g, &rv, 4) == 0;
            }
        printf("v_mfma_f32_%dx%dx%d_f32, K = %d: %d of %d outputs == k-ascending fmaf chain; == pairwise tree %d; == exact dot4 %d; == descending chain %d\n",
               n, n, n == 16 ? 4 : 2, K, chain, n * n, pair, dot4, rev);
    }
    return 0;
}
