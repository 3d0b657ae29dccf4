// fc_small.h -- linear layers of SMALL launches (the attack's 100 .. 1000-image evaluations) on v_mfma_f32_16x16x4_f32.
//
// A 100-row linear layer cannot fill the chip whatever the tile; what it pays for is the LENGTH of the dependent MFMA
// chain behind every output.  v_mfma_f32_32x32x2_f32 walks 2 k per 16-pass instruction (32 cycles per k),
// v_mfma_f32_16x16x4_f32 4 k per 8-pass instruction (8 cycles per k): the same exact-f32 arithmetic with a chain a quarter
// as long.  It is still acc = fmaf(a_k, b_k, acc) for k ascending -- inside the instruction and across instructions
// (scripts/microbench/mfma16_chain.hip: 256 of 256 outputs equal the CPU chain bit for bit) -- so the logits stay
// bit-identical to the big kernels' and the oracle's.
//
// One wavefront = one 16 x 16 output tile, no LDS, no barrier.  The instruction wants lane (ij = lane % 16, kq = lane / 16)
// to hold A[m0 + ij][k + kq] and B[k + kq][n0 + ij]: fetched like that, every instruction costs two 4-byte gathers over
// 16 + 4 cache lines and the texture path, not the MFMA chain, sets the pace (measured: 220 us for the three layers of a
// 100-image call, against 148 us on 32x32x2 tiles).  Instead a lane loads 16 BYTES of its activation row and 16 bytes of
// its weight row -- the weights are read from the TRANSPOSED copy [Cout][K] that the input-gradient pass keeps anyway
// (gr_model::gb) -- i.e. k = 16 t + 4 kq + e in register e, and two rounds of v_permlane32_swap / v_permlane16_swap
// transpose that 4 x 4 block (register e, lane group kq) across the wavefront, so that register s holds k = 16 t + 4 s + kq:
// what instruction s of the stage consumes.  Two 16-byte loads and 8 lane swaps per 4 instructions, ST stages in flight.
// Epilogue: + bias, ReLU, plain stores.
#pragma once
#include "conv_igemm.h"

typedef float fcs_f32x4 __attribute__((ext_vector_type(4)));

typedef unsigned fcs_u32x2 __attribute__((ext_vector_type(2)));

// in: register e of lane group q (= lane / 16) holds X[e][q]; out: register s of lane group q holds X[q][s]
__device__ __forceinline__ void fcs_transpose4(const float4& v, float (&t)[4]) {
    // v_permlane32_swap a, b: lanes 32..63 of a <-> lanes 0..31 of b;  v_permlane16_swap a, b: rows 1, 3 of a <-> rows 0, 2 of b
    fcs_u32x2 p02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v.x), __float_as_uint(v.z), false, false);
    fcs_u32x2 p13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v.y), __float_as_uint(v.w), false, false);
    fcs_u32x2 q01 = __builtin_amdgcn_permlane16_swap(p02[0], p13[0], false, false);
    fcs_u32x2 q23 = __builtin_amdgcn_permlane16_swap(p02[1], p13[1], false, false);
    t[0] = __uint_as_float(q01[0]); t[1] = __uint_as_float(q01[1]);
    t[2] = __uint_as_float(q23[0]); t[3] = __uint_as_float(q23[1]);
}

// wT: the layer's weights as [Cout (rows >= 16 * gridDim.x)][ldT], k contiguous
// TM x TN tiles of 16 x 16 per wavefront (rows m0 + 16 i, columns n0 + 16 j): TM + TN loads for TM * TN chains
template <int WAVES, int ST, int K, int TM, int TN>
__global__ __launch_bounds__(WAVES * 64) void k_fc_small(ConvParams p, const float* __restrict__ wT, int ldT) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ij = lane & 15, kq = lane >> 4;
    const int m0 = ((int)blockIdx.y * WAVES + wave) * (16 * TM);
    const int n0 = (int)blockIdx.x * (16 * TN);
    if (m0 >= p.M) return;                                      // wave-uniform; the kernel has no barrier
    const float* ap[TM];
    const float* bp[TN];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int row = m0 + 16 * i + ij < p.M ? m0 + 16 * i + ij : p.M - 1;   // rows past M repeat the last one and are not stored
        ap[i] = p.in + (size_t)row * K + 4 * kq;
    }
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int col = n0 + 16 * j + ij < p.NP ? n0 + 16 * j + ij : p.NP - 1; // wT has NP >= Cout rows
        bp[j] = wT + (size_t)col * ldT + 4 * kq;
    }
    fcs_f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = fcs_f32x4{0.f, 0.f, 0.f, 0.f};
    float4 a[ST][TM], b[ST][TN];
    auto load = [&](int slot, int t) {
#pragma unroll
        for (int i = 0; i < TM; i++) a[slot][i] = *reinterpret_cast<const float4*>(ap[i] + 16 * t);
#pragma unroll
        for (int j = 0; j < TN; j++) b[slot][j] = *reinterpret_cast<const float4*>(bp[j] + 16 * t);
    };
#pragma unroll
    for (int s = 0; s < ST - 1; s++) load(s, s);
    // K is a template argument and the walk is unrolled end to end: in straight-line code the compiler's s_waitcnt counts
    // are exact (ST - 1 stages stay in flight); around a loop's back edge it drains the whole ring once per trip
    constexpr int NT = K / 16;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        if (t + ST - 1 < NT) load((t + ST - 1) % ST, t + ST - 1);
        __builtin_amdgcn_sched_barrier(0);          // keep the stage's loads ahead of its MFMAs: the ring is the latency cover
        float ta[TM][4], tb[TN][4];
#pragma unroll
        for (int i = 0; i < TM; i++) fcs_transpose4(a[t % ST][i], ta[i]);
#pragma unroll
        for (int j = 0; j < TN; j++) fcs_transpose4(b[t % ST][j], tb[j]);
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[i][u], tb[j][u], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int co = n0 + 16 * j + ij;
        if (co < p.Cout) {
            const float bias = p.bias[co];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int m = m0 + 16 * i + 4 * kq + r;
                    if (m < p.M) {
                        float v = acc[i][j][r] + bias;
                        if (p.relu) v = v > 0.f ? v : 0.f;
                        p.out_raw[(size_t)m * p.Cout + co] = v;
                    }
                }
        }
    }
}

// the caller has checked: 1x1 layer on 1x1 maps (in = [M][K], 16-byte aligned), wT has NP >= Cout rows of ldT >= K floats
// (ldT % 4 == 0), out_raw only
template <int WAVES, int ST, int K, int TM = 1, int TN = 1>
static inline hipError_t launch_fc_small(const ConvParams& p, const float* wT, int ldT, hipStream_t st) {
    static_assert(K % 16 == 0 && K / 16 >= ST, "whole stages");
    dim3 grid((p.Cout + 16 * TN - 1) / (16 * TN), (p.M + 16 * TM * WAVES - 1) / (16 * TM * WAVES));
    hipLaunchKernelGGL((k_fc_small<WAVES, ST, K, TM, TN>), grid, dim3(WAVES * 64), 0, st, p, wT, ldT);
    return hipGetLastError();
}
