// conv21_bf16.h -- GTSRBNet's conv2_1 (5x5, 3 -> 32 channels, on the NHWC4 output of conv1_1) on
// v_mfma_f32_32x32x16_bf16, for the bf16 throughput mode (NOT the parity path).
//
// With 3 input channels a tap holds almost no k, so a whole KERNEL ROW is one "tap" (as conv_halo.h does for this layer):
// k = (ky ; kx*4 + c) with the 5 x 4 = 20 values of a row padded to 32 = two k-blocks of 16 (weights zero in the pad,
// and in channel 3).  The 8 k-values a lane feeds one MFMA are two neighbouring pixels of the staged tile:
// pixel x + 4*cb + 2*lh of row y + ky, 16 contiguous bytes of the bf16 NHWC4 tile.  Those 16 bytes are only 8-byte
// aligned for odd x, so the tile is kept twice in LDS, the second copy shifted by one pixel: a lane reads the copy of
// its x parity and every ds_read_b128 is 16-byte aligned (copy 1 starts 8 slots further in the bank row, which also
// keeps the even and the odd lanes of a 16-lane group on different banks).
// All 10 k-blocks of weights (10 KB) sit in LDS for the whole workgroup; one barrier in total.
// Epilogue: + bias, ReLU, f32 NHWC store (what conv2_2 reads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "conv_halo_bf16.h"

struct Conv21BfParams {
    const float* in;        // [B,32,32,4] f32 (channel 3 is zero)
    const uint16_t* wgt;    // bf16 [10 k-blocks][2][32][8]
    const float* bias;      // [32]
    void* out;              // [B,32,32,32] f32, or bf16 (OUT_BF)
    int B;
};

template <int TH, bool OUT_BF>   // tile = TH rows x 32 columns of one image; 4 wavefronts x TM = TH/4 m-tiles (one row each)
__global__ __launch_bounds__(256) void k_conv21_bf16(Conv21BfParams p) {
    constexpr int NT = 256, TW = 32, H = 32, W = 32;
    constexpr int THH = TH + 4;
    constexpr int TWP = TW + 4 + 4;                       // 2 + 32 + 2 pixels, + 4 that only zero weights ever touch
    constexpr int ROWB = TWP * 8;                         // 320 bytes = 20 slots
    constexpr int TM = TH / 4;
    static_assert(TH % 4 == 0, "one m-tile per tile row, TH/4 rows per wavefront");
    constexpr int C1 = ((THH * ROWB + 255) / 256) * 256 + 128;   // byte offset of copy 1: 8 slots into a 256-byte bank row
    constexpr int WOFF = (C1 + THH * ROWB + 15) / 16 * 16;       // weights

    __shared__ __attribute__((aligned(256))) unsigned char sm[WOFF + 10 * 1024];
    unsigned char* wsm = sm + WOFF;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int lp = lane & 31, lh = lane >> 5;
    constexpr int TPI = H / TH;
    int bid;                                              // XCD-aware: all tiles of an image on one XCD
    {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int g = (j / TPI) * 8 + xcd;
        if (g >= p.B) return;
        bid = g * TPI + j % TPI;
    }
    const int y0 = (bid % TPI) * TH, b = bid / TPI;

    // weights: 640 16-byte vectors
    {
        const u32x4* wg = reinterpret_cast<const u32x4*>(p.wgt);
        u32x4 w0 = wg[tid], w1 = wg[256 + tid], w2 = wg[tid < 128 ? 512 + tid : 0];
        *reinterpret_cast<u32x4*>(wsm + tid * 16) = w0;
        *reinterpret_cast<u32x4*>(wsm + (256 + tid) * 16) = w1;
        if (tid < 128) *reinterpret_cast<u32x4*>(wsm + (512 + tid) * 16) = w2;
    }
    // tile: pixel (hy, hx) of the haloed tile -> copy 0 at hx*8, copy 1 at (hx-1)*8 (its pixel 0 is never read)
    {
        constexpr int NPX = THH * TWP;
        constexpr int IT = (NPX + NT - 1) / NT;
        f32x4v v[IT];                                      // native vectors: HIP's float4 struct arrays cost registers / scratch
#pragma unroll
        for (int u = 0; u < IT; u++) {
            const int i = tid + u * NT;
            const int hy = i / TWP, hx = i % TWP;
            const int gy = y0 + hy - 2, gx = hx - 2;
            v[u] = f32x4v{0.f, 0.f, 0.f, 0.f};
            if (i < NPX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                v[u] = *reinterpret_cast<const f32x4v*>(p.in + (((size_t)b * H + gy) * W + gx) * 4);
        }
#pragma unroll
        for (int u = 0; u < IT; u++) {
            const int i = tid + u * NT;
            if (i < NPX) {
                const int hy = i / TWP, hx = i % TWP;
                bf16x4 h;
                h[0] = (__bf16)v[u][0]; h[1] = (__bf16)v[u][1]; h[2] = (__bf16)v[u][2]; h[3] = (__bf16)v[u][3];
                *reinterpret_cast<bf16x4*>(sm + hy * ROWB + hx * 8) = h;
                if (hx > 0) *reinterpret_cast<bf16x4*>(sm + C1 + hy * ROWB + (hx - 1) * 8) = h;
            }
        }
    }
    __syncthreads();

    // lane (x = lp, k half lh): copy of x's parity, even pixel base
    const unsigned char* abase = sm + (lp & 1) * C1 + ((lp & ~1) + 2 * lh) * 8;
    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 5; ky++)
#pragma unroll
        for (int cb = 0; cb < 2; cb++) {
            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(wsm + (ky * 2 + cb) * 1024 + lane * 16);
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int y = wave * TM + i;              // tile row of this m-tile
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(abase + (y + ky) * ROWB + cb * 32);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
            }
        }

    // epilogue through a wave-private LDS patch (conv_halo_bf16.h): 16-byte stores instead of one 2- / 4-byte store per value
    __syncthreads();                                      // every wave is done with the tile and the weights: LDS is free
    const float bias = p.bias[lp];
    {
        constexpr int EROW = 144;
        static_assert(4 * 32 * EROW <= WOFF + 10 * 1024, "epilogue patches must fit the kernel's LDS");
        unsigned char* patch = sm + wave * (32 * EROW);
        constexpr int CPV = OUT_BF ? 8 : 4, GPR = 32 / CPV, RPP = 64 / GPR, PASSES = 32 / RPP;
        const int cg = lane % GPR, prow = lane / GPR;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int gy = y0 + wave * TM + i;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                float v = acc[i][r] + bias;
                v = v > 0.f ? v : 0.f;
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
                const int gx = prow + RPP * it;
                const size_t off = (((size_t)b * H + gy) * W + gx) * 32 + cg * CPV;
                if constexpr (OUT_BF) {
                    bf16x8 h;
#pragma unroll
                    for (int e = 0; e < 4; e++) { h[e] = (__bf16)pv[it][0][e]; h[4 + e] = (__bf16)pv[it][1][e]; }
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.out) + off) = h;
                } else {
                    *reinterpret_cast<f32x4v*>(reinterpret_cast<float*>(p.out) + off) = pv[it][0];
                }
            }
        }
    }
}

template <int TH, bool OUT_BF>
static inline hipError_t launch_conv21_bf16(const Conv21BfParams& p, hipStream_t st) {
    const long long groups = ((long long)p.B + 7) / 8 * 8;
    const long long nblk = groups * (32 / TH);
    if (p.B <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_conv21_bf16<TH, OUT_BF>), dim3((unsigned)nblk), dim3(256), 0, st, p);
    return hipGetLastError();
}
