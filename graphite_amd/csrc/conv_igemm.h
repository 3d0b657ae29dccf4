// conv_igemm.h -- fp32 implicit-GEMM convolution / linear layer on the gfx950 matrix cores.
//
//   out[m, co] = epilogue( sum_k A[m, k] * Wp[k, co] ),   m = output pixel, k = (ky, kx, ci)
//   (or k = (ci/32, ky, kx, ci%32) for the layers the oracle walks in channel chunks, ConvParams::kchunk)
//
// * MFMA: v_mfma_f32_32x32x2_f32 (exact f32 in, f32 accumulate).  Each output element is ONE
//   k-ascending fmaf chain starting from 0 -- no split-K, no multiple accumulators per output --
//   which is what oracle/graphite_oracle.c::conv_nhwc computes, so logits compare bit for bit.
// * Activations NHWC, weights pre-packed [K][NP] (NP = Cout padded to the N tile, zero filled).
// * Workgroup = WM x WN wavefronts (64 lanes each); each wavefront owns TM x TN tiles of 32x32.
//   K is walked in chunks of BK through double-buffered LDS: global loads of chunk i+1 are in
//   flight while chunk i feeds the matrix cores; one barrier per chunk.
// * A-tile rows are padded to BK+1 floats: lane (p, h) reads A[p][2kk+h], a stride-(BK+1) access
//   that is bank-conflict free for the 32 lanes of each half-wave.
// * POOL: the M index enumerates 2x2 pooling windows (m = 4*window + 2*dy + dx) so that the four
//   pixels of a window sit in four consecutive accumulator registers of one lane; bias + ReLU +
//   max-pool happen in registers and only the pooled tensor is written.
// * Epilogue (runtime-uniform options): + bias, + residual, ReLU, raw store, and a second
//   "activated" store relu(v*scale+shift) (folded eval BatchNorm of the NEXT layer, WideResNet).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const float* in;      // [B,H,W,Cin] NHWC
    const float* wgt;     // [K][NP]
    const float* bias;    // [NP] or null
    const float* res;     // [M,Cout] residual or null
    const float* scale;   // [NP] folded BN of the consumer, or null
    const float* shift;
    float* out_raw;       // [M,Cout] (POOL: [M/4,Cout]) or null
    float* out_act;       // [M,Cout] or null
    int H, W, Cin, Ho, Wo, Cout, NP;
    int KH, KW, stride, pad;
    int M;                // B*Ho*Wo
    int relu;
    const float* gate;    // [M,Cout] or null: the output is zeroed where gate <= 0 (ReLU backward of the producer's activation)
    int kchunk;           // 0: k = (ky,kx,ci).  BK: k = (ci/BK, ky, kx, ci%BK) -- the order conv_halo_cc.h walks
    int vec_epi;          // 1: Cout % 4 == 0 and every output / residual / gate pointer is 16-byte aligned: tiles leave through
                          //    a wave-private LDS patch as 16-byte stores (set by the launcher)
    int raster;           // 1: 1-D grid, XCD-aware tile order (below); 0: grid (M blocks, N blocks)
    int mblocks, nblocks;
};

template <int BK, int WM, int WN, int TM, int TN, bool POOL>
__global__ __launch_bounds__(WM* WN * 64) void k_conv_igemm(ConvParams p) {
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int LDA = BK + 1;
    constexpr int SEGS = BK / 4;                       // 16-byte segments per A row
    constexpr int A_SEGS = BM * SEGS;
    constexpr int A_IT = (A_SEGS + NT - 1) / NT;
    constexpr int B_VEC = BK * BN / 4;
    constexpr int B_IT = (B_VEC + NT - 1) / NT;
    static_assert(BK % 4 == 0 && BK % 2 == 0, "BK");

    extern __shared__ float smem[];
    float* As = smem;                                  // [2][BM][LDA]
    float* Bs = smem + 2 * BM * LDA;                   // [2][BK][BN]

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int lp = lane & 31, lh = lane >> 5;
    // Tile order.  Workgroup i runs on XCD i % 8 (round-robin dispatch), and each XCD has its own 4 MB L2.  With the plain
    // 2-D grid the N blocks of one M block are thousands of workgroups apart, so its A rows (BM x K floats, 1 MB for fc5)
    // come from HBM once per N block.  Raster order: XCD x takes the M blocks m = x (mod 8) and walks ALL N blocks of an
    // M block back to back -- A is fetched from HBM once and re-read from that XCD's L2; the weights are shared by every
    // workgroup anyway.
    int mb, nb;
    if (p.raster) {
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        nb = j % p.nblocks;
        mb = (j / p.nblocks) * 8 + xcd;
        if (mb >= p.mblocks) return;             // padding workgroups of the last group of 8 M blocks (uniform exit, before any barrier)
    } else { mb = blockIdx.x; nb = blockIdx.y; }
    const int m0 = mb * BM;
    const int n0 = nb * BN;
    const int K = p.KH * p.KW * p.Cin;
    const int nchunk = K / BK;

    // ---- per-thread A staging state: which rows/segments this thread copies ----
    // UNI: every segment a thread copies has the same position inside the chunk (NT % SEGS == 0),
    // so one running (ky,kx,c) decode serves all of them.
    constexpr bool UNI = (NT % SEGS == 0);
    constexpr int DEC = UNI ? 1 : A_IT;
    long long a_base[A_IT];      // element offset of pixel (b, iy0, ix0), may point outside the image
    int a_yx[A_IT];              // iy0 << 16 | ix0 & 0xffff ; invalid rows get iy0 = -16384
    int a_ky[DEC], a_kx[DEC], a_c[DEC];                // running decode of k = chunk*BK + seg*4
#pragma unroll
    for (int i = 0; i < A_IT; i++) {
        int sid = tid + i * NT;
        int row = sid / SEGS, seg = sid % SEGS;
        int m = m0 + row;
        bool ok = (sid < A_SEGS) && (m < p.M);
        int b, oy, ox;
        if (POOL) {
            int P = m >> 2, q = m & 3;
            int Hp = p.Ho >> 1, Wp = p.Wo >> 1;
            b = P / (Hp * Wp);
            int r = P % (Hp * Wp);
            oy = 2 * (r / Wp) + (q >> 1);
            ox = 2 * (r % Wp) + (q & 1);
        } else {
            b = m / (p.Ho * p.Wo);
            int r = m % (p.Ho * p.Wo);
            oy = r / p.Wo;
            ox = r % p.Wo;
        }
        int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        a_base[i] = (((long long)b * p.H + iy0) * p.W + ix0) * p.Cin;
        a_yx[i] = ((ok ? iy0 : -16384) << 16) | (ix0 & 0xffff);
        if (i < DEC) {
            int k = seg * 4;
            int tap = k / p.Cin;
            a_c[i] = k % p.Cin;
            a_ky[i] = tap / p.KW;
            a_kx[i] = tap % p.KW;
        }
    }

    float4 a_reg[A_IT], b_reg[B_IT];

    auto load_chunk = [&](int kc) {
#pragma unroll
        for (int i = 0; i < A_IT; i++) {
            const int d = UNI ? 0 : i;
            int iy = (a_yx[i] >> 16) + a_ky[d], ix = (int)(short)(a_yx[i] & 0xffff) + a_kx[d];
            bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) v = *reinterpret_cast<const float4*>(p.in + a_base[i] + ((long long)a_ky[d] * p.W + a_kx[d]) * p.Cin + a_c[d]);
            a_reg[i] = v;
        }
#pragma unroll
        for (int d = 0; d < DEC; d++) {                 // advance the decode by one chunk
            if (p.kchunk) {                             // next tap of this channel chunk, then the next channel chunk
                if (++a_kx[d] == p.KW) {
                    a_kx[d] = 0;
                    if (++a_ky[d] == p.KH) { a_ky[d] = 0; a_c[d] += BK; }
                }
            } else {
                a_c[d] += BK;
                while (a_c[d] >= p.Cin) {
                    a_c[d] -= p.Cin;
                    if (++a_kx[d] == p.KW) { a_kx[d] = 0; ++a_ky[d]; }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; i++) {
            int vid = tid + i * NT;
            int k = vid / (BN / 4), j4 = vid % (BN / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (vid < B_VEC) v = *reinterpret_cast<const float4*>(p.wgt + (size_t)(kc * BK + k) * p.NP + n0 + j4 * 4);
            b_reg[i] = v;
        }
    };
    auto store_chunk = [&](int buf) {
        float* a = As + buf * BM * LDA;
        float* b = Bs + buf * BK * BN;
#pragma unroll
        for (int i = 0; i < A_IT; i++) {
            if (tid + i * NT < A_SEGS) {
                int sid = tid + i * NT;
                float* d = a + (sid / SEGS) * LDA + (sid % SEGS) * 4;
                d[0] = a_reg[i].x; d[1] = a_reg[i].y; d[2] = a_reg[i].z; d[3] = a_reg[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; i++) {
            int vid = tid + i * NT;
            if (vid < B_VEC) *reinterpret_cast<float4*>(b + vid * 4) = b_reg[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    for (int kc = 0; kc < nchunk; kc++) {
        const int buf = kc & 1;
        if (kc + 1 < nchunk) load_chunk(kc + 1);
        const float* a = As + buf * BM * LDA + (wm * TM * 32 + lp) * LDA + lh;
        const float* b = Bs + buf * BK * BN + lh * BN + wn * TN * 32 + lp;
        // The fragments of k-pair kk+1 are read BEFORE the MFMAs of k-pair kk are issued (two register sets), and a
        // scheduling barrier keeps the compiler from sinking the reads back down to their use: left alone it reads a
        // k-pair's fragments, waits for the LDS (lgkmcnt(0)) and only then issues that pair's MFMAs -- a full LDS latency
        // exposed per 4 MFMAs with two waves per SIMD to cover it (108 TFLOP/s on the plain GEMM shapes).
        float av[2][TM], bv[2][TN];
#pragma unroll
        for (int i = 0; i < TM; i++) av[0][i] = a[i * 32 * LDA];
#pragma unroll
        for (int j = 0; j < TN; j++) bv[0][j] = b[j * 32];
#pragma unroll
        for (int kk = 0; kk < BK / 2; kk++) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; i++) av[nxt][i] = a[i * 32 * LDA + 2 * (kk + 1)];
#pragma unroll
                for (int j = 0; j < TN; j++) bv[nxt][j] = b[2 * (kk + 1) * BN + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kc + 1 < nchunk) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds column co = .. + lp, rows (r&3) + 8*(r>>2) + 4*lh of each tile ----
    if (!POOL && p.vec_epi) {
        // A lane's 16 values of a tile are 16 different ROWS of one column: stored directly they leave as 4-byte scalars,
        // 128 contiguous bytes per row and instruction (16 TFLOP/s on BagNet's K = 64 -> N = 256 expansions, which are
        // all epilogue).  Each wave transposes its 32x32 tile through a private LDS patch (the A / B buffers are free
        // after the last barrier; 36-float rows: conflict-free writes, 16-byte aligned reads; LDS is in order within a
        // wave, so no barrier) and every lane then owns 4 consecutive columns of 4 rows: float4 loads of the residual /
        // gate, float4 stores.  The arithmetic per element is unchanged: (acc + bias) + res, ReLU, gate, fma(scale, shift).
        float* patch = smem + wave * (32 * 36);
        const int prow = lane >> 3, pc4 = (lane & 7) * 4;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int cb = n0 + (wn * TN + j) * 32;
            const float bias = (p.bias && cb + lp < p.Cout) ? p.bias[cb + lp] : 0.f;
            const int co = cb + pc4;
            const bool cok = co < p.Cout;
            float4 sc4 = make_float4(0.f, 0.f, 0.f, 0.f), sh4 = sc4;
            if (p.out_act && cok) {
                sc4 = *reinterpret_cast<const float4*>(p.scale + co);
                sh4 = *reinterpret_cast<const float4*>(p.shift + co);
            }
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int mt = m0 + (wm * TM + i) * 32;
#pragma unroll
                for (int r = 0; r < 16; r++) patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + lp] = acc[i][j][r] + bias;
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const int row = prow + 8 * it;
                    float4 v = *reinterpret_cast<const float4*>(patch + row * 36 + pc4);
                    const int m = mt + row;
                    if (cok && m < p.M) {
                        const size_t o = (size_t)m * p.Cout + co;
                        if (p.res) {
                            const float4 q = *reinterpret_cast<const float4*>(p.res + o);
                            v.x = v.x + q.x; v.y = v.y + q.y; v.z = v.z + q.z; v.w = v.w + q.w;
                        }
                        if (p.relu) {
                            v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                            v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                        }
                        if (p.gate) {
                            const float4 g = *reinterpret_cast<const float4*>(p.gate + o);
                            if (!(g.x > 0.f)) v.x = 0.f;
                            if (!(g.y > 0.f)) v.y = 0.f;
                            if (!(g.z > 0.f)) v.z = 0.f;
                            if (!(g.w > 0.f)) v.w = 0.f;
                        }
                        if (p.out_raw) *reinterpret_cast<float4*>(p.out_raw + o) = v;
                        if (p.out_act) {
                            float4 w;
                            w.x = fmaf(v.x, sc4.x, sh4.x); w.y = fmaf(v.y, sc4.y, sh4.y);
                            w.z = fmaf(v.z, sc4.z, sh4.z); w.w = fmaf(v.w, sc4.w, sh4.w);
                            w.x = w.x > 0.f ? w.x : 0.f; w.y = w.y > 0.f ? w.y : 0.f;
                            w.z = w.z > 0.f ? w.z : 0.f; w.w = w.w > 0.f ? w.w : 0.f;
                            *reinterpret_cast<float4*>(p.out_act + o) = w;
                        }
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int co = n0 + (wn * TN + j) * 32 + lp;
        const bool cok = co < p.Cout;
        const float bias = (p.bias && cok) ? p.bias[co] : 0.f;
        const float sc = (p.scale && cok) ? p.scale[co] : 0.f;
        const float sh = (p.shift && cok) ? p.shift[co] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int mt = m0 + (wm * TM + i) * 32;
            if (POOL) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    float v0 = acc[i][j][4 * g + 0] + bias, v1 = acc[i][j][4 * g + 1] + bias;
                    float v2 = acc[i][j][4 * g + 2] + bias, v3 = acc[i][j][4 * g + 3] + bias;
                    if (p.relu) {
                        v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f;
                        v2 = v2 > 0.f ? v2 : 0.f; v3 = v3 > 0.f ? v3 : 0.f;
                    }
                    float mx = v0 > v1 ? v0 : v1;
                    mx = mx > v2 ? mx : v2;
                    mx = mx > v3 ? mx : v3;
                    int m = mt + 8 * g + 4 * lh;               // first pixel of the window
                    if (cok && m < p.M) p.out_raw[(size_t)(m >> 2) * p.Cout + co] = mx;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    int m = mt + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (cok && m < p.M) {
                        float v = acc[i][j][r] + bias;
                        if (p.res) v = v + p.res[(size_t)m * p.Cout + co];
                        if (p.relu) v = v > 0.f ? v : 0.f;
                        if (p.gate && !(p.gate[(size_t)m * p.Cout + co] > 0.f)) v = 0.f;
                        if (p.out_raw) p.out_raw[(size_t)m * p.Cout + co] = v;
                        if (p.out_act) {
                            float w = fmaf(v, sc, sh);
                            p.out_act[(size_t)m * p.Cout + co] = w > 0.f ? w : 0.f;
                        }
                    }
                }
            }
        }
    }
}

template <int BK, int WM, int WN, int TM, int TN, bool POOL>
static inline hipError_t launch_conv_igemm(const ConvParams& p, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr size_t smem_ab = sizeof(float) * (2 * BM * (BK + 1) + 2 * BK * BN);
    constexpr size_t smem_ep = sizeof(float) * (WM * WN * 32 * 36);          // the epilogue's per-wave patches reuse the buffers
    constexpr size_t smem = smem_ab > smem_ep ? smem_ab : smem_ep;
    auto kern = k_conv_igemm<BK, WM, WN, TM, TN, POOL>;
    // the attribute is per device: one flag per device this process launches on (atomic: two host threads with two contexts
    // may meet here; setting the attribute twice is harmless)
    static std::atomic<bool> attr_set[64];
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev_]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)smem);
        if (e != hipSuccess) return e;
        attr_set[dev_] = true;
    }
    ConvParams q = p;
    q.mblocks = (p.M + BM - 1) / BM;
    q.nblocks = p.NP / BN;
    static int raster = -1;                      // GRAPHITE_IGEMM_RASTER=0: the plain 2-D grid (A/B runs)
    if (raster < 0) { const char* e = getenv("GRAPHITE_IGEMM_RASTER"); raster = (e && e[0] == '0') ? 0 : 1; }
    auto a16 = [](const void* ptr) { return ((uintptr_t)ptr & 15) == 0; };
    static int vec = -1;                         // GRAPHITE_IGEMM_VEC=0: the scalar epilogue (A/B runs)
    if (vec < 0) { const char* e = getenv("GRAPHITE_IGEMM_VEC"); vec = (e && e[0] == '0') ? 0 : 1; }
    q.vec_epi = (vec && !POOL && p.Cout % 4 == 0 && a16(p.out_raw) && a16(p.out_act) && a16(p.res) && a16(p.gate) &&
                 a16(p.scale) && a16(p.shift)) ? 1 : 0;
    const long long padded = (long long)((q.mblocks + 7) / 8) * 8 * q.nblocks;
    q.raster = (raster && q.nblocks > 1 && padded <= 0x7fffffffLL) ? 1 : 0;
    dim3 grid = q.raster ? dim3((unsigned)padded) : dim3(q.mblocks, q.nblocks);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, st, q);
    return hipGetLastError();
}
