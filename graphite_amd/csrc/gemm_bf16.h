// gemm_bf16.h -- C[M][N] = act(A[M][K] * W + bias [+ res]) on v_mfma_f32_32x32x16_bf16, for GTSRBNet's fc5 / fc6 and
// BagNet's stride-1 1x1 convs in the bf16 throughput mode (GR_PREC_BF16; not the parity path).
//   A: f32 row-major activations, rounded to bf16 (nearest even) while a BM x BK stage goes global -> registers -> LDS
//      (row stride BK*2+16 bytes = an odd number of 16-byte slots: conflict-free ds_read_b128 fragments).
//   W: bf16, packed at load in fragment order [K/16][N/32][2][32][8] (conv_halo_bf16.h), staged the same way.
//   Two LDS buffers, one barrier per K step of BK, register ring one stage deep beyond the LDS buffers; bias + ReLU
//   epilogue, f32 store.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "conv_halo_bf16.h"

struct GemmBfParams {
    const void* a;          // [M][K] f32, or bf16 (IN_BF)
    const uint16_t* w;      // packed bf16
    const float* bias;      // [N]
    void* c;                // [M][N] f32, or bf16 (OUT_BF)
    int M, N, K, relu;
    const float* res;       // [M][N] f32 residual added before the ReLU, or null (BagNet's bottleneck blocks)
};

template <int WM, int WN, int TM, int TN, int BK, bool IN_BF, bool OUT_BF>
__global__ __launch_bounds__(WM* WN * 64, 2) void k_gemm_bf16(GemmBfParams p) {
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int ASTR = BK * 2 + 16;                     // bytes per A row in LDS
    constexpr int A_ST = BM * ASTR;
    constexpr int KBS = BK / 16;
    constexpr int B_ST = KBS * (BN / 32) * 1024;
    constexpr int EPS = IN_BF ? 8 : 4;                    // A elements per 16-byte load
    constexpr int AV = BM * (BK / EPS) / NT;              // 16-byte A loads per thread and stage
    constexpr int BV = B_ST / 16 / NT;                    // 16-byte loads per thread and stage
    static_assert(BM * (BK / EPS) % NT == 0 && (B_ST / 16) % NT == 0, "stage loads must divide evenly");
    static_assert((ASTR / 16) % 2 == 1, "A row stride must be an odd number of 16-byte slots");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_g[];
    unsigned char* asm_ = smem_g;                         // 2 x A_ST
    unsigned char* bsm = smem_g + 2 * A_ST;               // 2 x B_ST

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int lp = lane & 31, lh = lane >> 5;
    const int nbn = p.N / BN;
    const int m0 = (blockIdx.x / nbn) * BM, n0 = (blockIdx.x % nbn) * BN;   // consecutive workgroups share the A rows
    const int NS = p.K / BK;

    u32x4 areg[2][AV];
    u32x4 breg[2][BV];
    const u32x4* wg = reinterpret_cast<const u32x4*>(p.w);
    auto gload = [&](u32x4 (&ar)[AV], u32x4 (&br)[BV], int s) {
        if (s >= NS) s = NS - 1;                          // clamped, never conditional (conv_halo.h)
#pragma unroll
        for (int v = 0; v < AV; v++) {
            const int f = tid + v * NT;
            int row = m0 + f / (BK / EPS);
            if (row >= p.M) row = p.M - 1;
            ar[v] = reinterpret_cast<const u32x4*>(p.a)[((size_t)row * p.K + s * BK) / EPS + f % (BK / EPS)];
        }
#pragma unroll
        for (int v = 0; v < BV; v++) {
            const int idx = tid + v * NT;
            const int c = idx / (BN / 32 * 64), r = idx % (BN / 32 * 64);
            br[v] = wg[((size_t)(s * KBS + c) * (p.N / 32) + n0 / 32) * 64 + r];
        }
    };
    auto lstore = [&](const u32x4 (&ar)[AV], const u32x4 (&br)[BV], int buf) {
#pragma unroll
        for (int v = 0; v < AV; v++) {
            const int f = tid + v * NT;
            unsigned char* d = asm_ + buf * A_ST + (f / (BK / EPS)) * ASTR + (f % (BK / EPS)) * (EPS * 2);
            if constexpr (IN_BF) {
                *reinterpret_cast<u32x4*>(d) = ar[v];
            } else {
                const f32x4v fv = __builtin_bit_cast(f32x4v, ar[v]);    // whole vector: an element-wise bit_cast miscompiles (ROCm 7.2)
                bf16x4 h;
                h[0] = (__bf16)fv[0]; h[1] = (__bf16)fv[1]; h[2] = (__bf16)fv[2]; h[3] = (__bf16)fv[3];
                *reinterpret_cast<bf16x4*>(d) = h;
            }
        }
#pragma unroll
        for (int v = 0; v < BV; v++) *reinterpret_cast<u32x4*>(bsm + buf * B_ST + (tid + v * NT) * 16) = br[v];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int a_lane = (wm * TM * 32 + lp) * ASTR + lh * 16;
    const int b_lane = (wn * TN) * 1024 + lane * 16;
    auto compute = [&](int buf) {
        const unsigned char* ab = asm_ + buf * A_ST + a_lane;
        const unsigned char* bb = bsm + buf * B_ST + b_lane;
#pragma unroll
        for (int c = 0; c < KBS; c++) {
            bf16x8 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) av[i] = *reinterpret_cast<const bf16x8*>(ab + i * 32 * ASTR + c * 32);
#pragma unroll
            for (int j = 0; j < TN; j++) bv[j] = *reinterpret_cast<const bf16x8*>(bb + (c * (BN / 32) + j) * 1024);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };

    gload(areg[0], breg[0], 0);
    gload(areg[1], breg[1], 1);
    lstore(areg[0], breg[0], 0);
    gload(areg[0], breg[0], 2);
    __syncthreads();
    // iteration s: stage s from LDS buffer s&1; stage s+1 leaves register set (s+1)&1 for the other buffer, and that
    // set takes the loads of stage s+3.  Two iterations per trip keep the register-set indices compile-time.
    for (int s = 0; s < NS; s += 2) {
        compute(0);
        lstore(areg[1], breg[1], 1);
        gload(areg[1], breg[1], s + 3);
        __syncthreads();
        if (s + 1 < NS) {
            compute(1);
            lstore(areg[0], breg[0], 0);
            gload(areg[0], breg[0], s + 4);
        }
        __syncthreads();
    }

    // Epilogue through a wave-private LDS patch (conv_halo_bf16.h): an accumulator tile has its 32 columns on the lanes and
    // its rows in the registers, so stored directly every value is its own 2- or 4-byte access (16 per tile and lane, and a
    // store costs the wave about as much as three MFMAs: a quarter of fc5's time, 40 % of fc6's).  The loop's last barrier
    // has retired every read of the A / B buffers; each wave transposes its tile in its own 32 x 144-byte patch (LDS is in
    // order within a wave) and leaves with 16-byte stores: 8 bf16 (or 4 f32) columns of one row per lane.
    {
        constexpr int EROW = 144;
        static_assert(NT / 64 * 32 * EROW <= 2 * A_ST + 2 * B_ST, "epilogue patches must fit the kernel's LDS");
        unsigned char* patch = smem_g + wave * (32 * EROW);
        constexpr int CPV = OUT_BF ? 8 : 4, GPR = 32 / CPV, RPP = 64 / GPR, PASSES = 32 / RPP;
        const int cg = lane % GPR, prow = lane / GPR;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int nb = n0 + (wn * TN + j) * 32;
            const float bias = p.bias ? p.bias[nb + lp] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int mb = m0 + (wm * TM + i) * 32;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    float v = acc[i][j][r] + bias;
                    if (p.relu && !p.res) v = v > 0.f ? v : 0.f;
                    *reinterpret_cast<float*>(patch + ((r & 3) + 8 * (r >> 2) + 4 * lh) * EROW + lp * 4) = v;
                }
                asm volatile("" ::: "memory");
                f32x4v pv[PASSES][OUT_BF ? 2 : 1];
#pragma unroll
                for (int it = 0; it < PASSES; it++) {
                    const unsigned char* src = patch + (prow + RPP * it) * EROW + cg * (CPV * 4);
                    pv[it][0] = *reinterpret_cast<const f32x4v*>(src);
                    if constexpr (OUT_BF) pv[it][1] = *reinterpret_cast<const f32x4v*>(src + 16);
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int it = 0; it < PASSES; it++) {
                    const int m = mb + prow + RPP * it;
                    if (m >= p.M) continue;
                    const size_t off = (size_t)m * p.N + nb + cg * CPV;
                    if (p.res) {                                     // (acc + bias) + res, then the ReLU: conv_igemm.h's order
#pragma unroll
                        for (int q = 0; q < (OUT_BF ? 2 : 1); q++) {
                            const f32x4v rv = *reinterpret_cast<const f32x4v*>(p.res + off + 4 * q);
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                float v = pv[it][q][e] + rv[e];
                                pv[it][q][e] = (p.relu && !(v > 0.f)) ? 0.f : v;
                            }
                        }
                    }
                    if constexpr (OUT_BF) {
                        bf16x8 h;
#pragma unroll
                        for (int e = 0; e < 4; e++) { h[e] = (__bf16)pv[it][0][e]; h[4 + e] = (__bf16)pv[it][1][e]; }
                        *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.c) + off) = h;
                    } else {
                        *reinterpret_cast<f32x4v*>(reinterpret_cast<float*>(p.c) + off) = pv[it][0];
                    }
                }
            }
        }
    }
}

template <int WM, int WN, int TM, int TN, int BK, bool IN_BF = false, bool OUT_BF = false>
static inline hipError_t launch_gemm_bf16(const GemmBfParams& p, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr size_t smem = 2 * (size_t)BM * (BK * 2 + 16) + 2 * (size_t)(BK / 16) * (BN / 32) * 1024;
    static_assert(smem <= 160 * 1024, "LDS");
    if (p.M <= 0 || p.N % BN || p.K % BK || p.K / BK < 2) return hipErrorInvalidValue;
    auto kern = k_gemm_bf16<WM, WN, TM, TN, BK, IN_BF, OUT_BF>;
    static std::atomic<bool> attr_set[64];                   // the attribute is per device
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev_]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)smem);
        if (e != hipSuccess) return e;
        attr_set[dev_] = true;
    }
    const long long grid = (long long)((p.M + BM - 1) / BM) * (p.N / BN);
    if (grid > 0x7fffffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WM * WN * 64), smem, st, p);
    return hipGetLastError();
}
