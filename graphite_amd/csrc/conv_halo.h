// conv_halo.h -- fp32 MFMA convolution for stride-1 "same" convs (GTSRBNet's six 5x5 layers).
//
// Same arithmetic as conv_igemm.h (one k-ascending v_mfma_f32_32x32x2_f32 chain per output, k =
// (ky,kx,ci); bit-identical results), different data movement:
//   * A (activations): the workgroup stages the haloed input tile ONCE into LDS -- (TH+4)x(TW+4)
//     pixels x Cin for NIMG images -- instead of re-staging an im2col chunk for each of the 25 taps.
//     Lane (pixel, k-half) reads A[pixel + tap][ci] straight from the tile; the per-pixel stride
//     PS = Cin+1 keeps the 32 lanes of a half-wave on distinct banks.
//   * B (weights): streamed from global memory (L1/L2 resident: <= 1.6 MB per layer, shared by every
//     workgroup) directly into registers, KU k-pairs ahead of use; no LDS, no barrier.
//   => the main loop has NO workgroup barrier: the four wavefronts run decoupled and the matrix
//      pipe only waits at tile boundaries, where the CU's other resident workgroups cover.
//   * Epilogue as conv_igemm.h: + bias, ReLU, optional 2x2 max-pool in registers (pool windows are
//     mapped to 4 consecutive accumulator registers), NHWC store.
// conv2_1 (Cin = 3, NHWC4 in HBM) uses KW = 1: one "tap" is a whole kernel row.  PS = 3 / CIN = 16: the tile keeps
// the three real channels only, so a kernel row's 5 pixels x 3 channels are 15 contiguous floats, + 1 zero-weight k
// (K = 80; a stride of 3 floats is conflict-free for 32 consecutive pixels).  PS = 4 / CIN = 20: the pad channel
// is kept (K = 100), the earlier form.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_igemm.h"

struct HaloParams {
    const float* in;      // [B,H,W,CPH] NHWC
    const float* wgt;     // [K][NP]
    const float* bias;    // [NP]
    float* out;           // [B,H,W,Cout] or pooled [B,H/2,W/2,Cout]
    int B, H, W, Cout;
    int relu;
    int dbg;              // tuning aid (GRAPHITE_HALO_DBG): 1 = skip the tile staging, 2 = skip the MFMA loop, 4 = skip the epilogue (wrong results)
};

// CIN: k extent of one tap; CPH: physical channels per pixel; KH,KW: taps; PS: LDS pixel stride (floats);
// TH,TW: output tile; NIMG: images per workgroup; WM x WN wavefronts, TM x TN 32x32 tiles each;
// KU: k-pairs of B prefetched per stage; NP: padded Cout (row length of wgt).
#ifndef HALO_UB
#define HALO_UB 0
#endif
template <int CIN, int CPH, int KH, int KW, int PS, int TH, int TW, int NIMG, int WM, int WN, int TM, int TN, int KU, int NP,
          bool POOL, int PIPE>
__global__ __launch_bounds__(WM* WN * 64) void k_conv_halo(HaloParams p) {
    constexpr int NT = WM * WN * 64;
    constexpr int HALO = 2;
    constexpr int THH = TH + 2 * HALO, TWH = TW + 2 * HALO;
    // LDS row stride.  A half-wave reads one m-tile = XR pixels of each of 32/XR rows (pixels of a row sit on
    // consecutive banks because PS = 1 mod 32); padding the row so that RS = XR (mod 32) puts the rows
    // side by side on the 32 banks: conflict-free A reads (also for the 2x2-window order of POOL).
    constexpr int XR = TW < 32 ? TW : 32;
    // (skipped for the one case -- 16-wide, un-pooled conv3_1 -- where the pad would cost a resident workgroup)
    constexpr int RPAD = (PS % 32 == 1 && XR < 32 && (POOL || TW <= 8)) ? ((XR - (TWH * PS) % 32) + 32) % 32 : 0;
    constexpr int RS = TWH * PS + RPAD;
    constexpr int IMGS = THH * RS;                // LDS image stride
    constexpr int MT = NIMG * TH * TW / 32;       // m-tiles per workgroup
    constexpr int KK_TAP = CIN / 2;               // k-pairs per tap
    constexpr int SUBS = KK_TAP / KU;
    constexpr int NS = KH * KW * SUBS;            // operand stages
    static_assert(MT == WM * TM, "m tiling");
    static_assert(NP == WN * TN * 32, "n tiling");
    static_assert(KK_TAP % KU == 0, "KU");
    static_assert(CPH % 4 == 0, "CPH");
    static_assert(PS != 3 || (CPH == 4 && KW == 1 && CIN == 16 && KH * 3 == 15), "packed three-channel rows: conv2_1 only");

    extern __shared__ float smem[];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int lp = lane & 31, lh = lane >> 5;
    const int tiles_x = p.W / TW, tiles_y = p.H / TH;
    // Workgroups are dealt round-robin to the 8 XCDs (one L2 each).  Decode blockIdx so that all tiles of one image
    // group run on the SAME XCD, back to back: the halo rows two neighbouring tiles share are then read from HBM once.
    int bid;
    {
        const int tpi = tiles_x * tiles_y, groups = (p.B + NIMG - 1) / NIMG;
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int g = (j / tpi) * 8 + xcd;
        if (g >= groups) return;                  // grid is rounded up to a multiple of 8 groups; whole workgroup leaves
        bid = g * tpi + j % tpi;
    }
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const int b0 = bid * NIMG;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- stage the haloed input tile (zero outside the image / batch) ----
    // Loads are issued in batches of up to 8 per thread before any LDS write, so their latencies overlap.
    {
        constexpr int SEG = CPH / 4;
        constexpr int TOT = NIMG * THH * TWH * SEG;
        constexpr int IT = (TOT + NT - 1) / NT;
        constexpr int UB = HALO_UB > 0 ? HALO_UB : (IT <= 16 ? IT : (IT + 1) / 2);   // loads in flight per thread before the first LDS write
        if (!(p.dbg & 1))
#pragma unroll
        for (int it0 = 0; it0 < IT; it0 += UB) {
            float4 v[UB];
            int dst[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int i = tid + (it0 + u) * NT;
                int seg = i % SEG;
                int px = i / SEG;
                int hx = px % TWH; px /= TWH;
                int hy = px % THH;
                int im = px / THH;
                int gy = y0 + hy - HALO, gx = x0 + hx - HALO, gb = b0 + im;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                dst[u] = (it0 + u < IT && i < TOT) ? im * IMGS + hy * RS + hx * PS + seg * 4 : -1;
                if (dst[u] >= 0 && gb < p.B && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
                    v[u] = *reinterpret_cast<const float4*>(p.in + (((size_t)gb * p.H + gy) * p.W + gx) * CPH + seg * 4);
            }
#pragma unroll
            for (int u = 0; u < UB; u++) {
                if (dst[u] >= 0) {
                    float* d = smem + dst[u];
                    d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z;
                    if (PS != 3) d[3] = v[u].w;          // PS == 3: the pad channel of an NHWC4 pixel is not kept
                }
            }
        }
    }

    if (PS == 3 && tid == 0) smem[NIMG * IMGS] = 0.f;   // the 16th k of the last pixel's kernel row reads one float past the tile (x 0 weight)

    // ---- per-lane A offsets: which pixel of the tile each of my TM m-tiles' row `lp` is ----
    int a_off[TM];
    int o_img[TM], o_y[TM], o_x[TM];             // kept for the epilogue
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int mt = wm * TM + i;
        int im, y, x;
        if (POOL) {
            constexpr int WPR = TW / 2, WPI = (TH / 2) * WPR;
            int w = mt * 8 + (lp >> 2), q = lp & 3;
            im = w / WPI;
            int r = w % WPI;
            y = 2 * (r / WPR) + (q >> 1);
            x = 2 * (r % WPR) + (q & 1);
        } else {
            int l = mt * 32 + lp;
            im = l / (TH * TW);
            int r = l % (TH * TW);
            y = r / TW;
            x = r % TW;
        }
        a_off[i] = im * IMGS + y * RS + x * PS + lh;
        o_img[i] = im; o_y[i] = y; o_x[i] = x;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // ---- operand streams ----
    //   B: lane reads W[k = 2kk + lh][n0 + j*32 + lp] from global (linear in k across taps), one stage
    //      (KU k-pairs) ahead, into one of two alternating register sets.  The loads are UNCONDITIONAL: the
    //      stream runs one stage past the end of the weights (make_layer pads the buffer with zero rows).  A
    //      conditional prefetch makes the compiler's vmcnt bookkeeping assume the shortest path, and the MFMAs of
    //      a stage then wait for the prefetch issued during that very stage.
    //   A: lane reads tile[pixel + tap][ci = 2kk + lh] from LDS.
    //      PIPE 0: just in time, B of the next stage loaded as one burst before the stage's MFMAs;
    //      PIPE = AD (+16): A fetched AD k-pairs ahead through a register ring, B loads of the next stage interleaved
    //              with this stage's MFMAs; without +16 sched_group_barrier pins the order
    //              (TM A reads, then (TM MFMAs, 1 B load) x TN per k-pair), with +16 the compiler schedules.
    const float* bptr = p.wgt + (size_t)lh * NP + (wn * TN) * 32 + lp;
    float b0r[KU][TN], b1r[KU][TN];
    int nky = 0, nkx = 0, nsub = 0;               // (ky,kx,sub) of the next stage whose A base is needed
    auto next_a_base = [&]() {
        const float* a_tap = smem + nky * RS + nkx * PS + nsub * (2 * KU);
        if (++nsub == SUBS) { nsub = 0; if (++nkx == KW) { nkx = 0; if (++nky == KH) nky = 0; } }
        return a_tap;
    };
    if (p.dbg & 2) { __syncthreads(); } else
    if (PIPE == 0) {
        auto load_b = [&](float (&br)[KU][TN]) {
#pragma unroll
            for (int kk = 0; kk < KU; kk++)
#pragma unroll
                for (int j = 0; j < TN; j++) br[kk][j] = bptr[(size_t)(2 * kk) * NP + j * 32];
            bptr += (size_t)2 * KU * NP;
        };
        auto compute = [&](float (&br)[KU][TN]) {
            const float* a_tap = next_a_base();
#pragma unroll
            for (int kk = 0; kk < KU; kk++) {
                float av[TM];
#pragma unroll
                for (int i = 0; i < TM; i++) av[i] = a_tap[a_off[i] + 2 * kk];
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], br[kk][j], acc[i][j], 0, 0, 0);
            }
        };
        load_b(b0r);
        __syncthreads();                          // the only barrier: tile staged
        for (int s = 0; s + 1 < NS; s += 2) {
            load_b(b1r);
            compute(b0r);
            load_b(b0r);
            compute(b1r);
        }
        if (NS & 1) compute(b0r);
    } else {
        // A ring: the values of k-pair kk+AD are fetched while k-pair kk feeds the matrix cores
        constexpr int AD = (PIPE & 15) ? (PIPE & 15) : 1;
        constexpr bool PIN = (PIPE & 16) == 0;
        static_assert(AD >= 1 && KU % AD == 0, "A prefetch distance must divide KU");
        float aring[AD][TM];
        auto stage = [&](float (&bc)[KU][TN], float (&bn)[KU][TN], const float* a_tap, const float* a_nxt) {
#pragma unroll
            for (int kk = 0; kk < KU; kk++) {
                float av[TM];
#pragma unroll
                for (int i = 0; i < TM; i++) av[i] = aring[kk % AD][i];
#pragma unroll
                for (int i = 0; i < TM; i++)
                    aring[kk % AD][i] = kk + AD < KU ? a_tap[a_off[i] + 2 * (kk + AD)] : a_nxt[a_off[i] + 2 * (kk + AD - KU)];
#pragma unroll
                for (int j = 0; j < TN; j++) bn[kk][j] = bptr[(size_t)(2 * kk) * NP + j * 32];
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bc[kk][j], acc[i][j], 0, 0, 0);
                if (PIN) {
                    __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
#pragma unroll
                    for (int j = 0; j < TN; j++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
            }
            bptr += (size_t)2 * KU * NP;
        };
#pragma unroll
        for (int kk = 0; kk < KU; kk++)
#pragma unroll
            for (int j = 0; j < TN; j++) b0r[kk][j] = bptr[(size_t)(2 * kk) * NP + j * 32];
        bptr += (size_t)2 * KU * NP;
        __syncthreads();                          // the only barrier: tile staged
        const float* a_cur = next_a_base();
#pragma unroll
        for (int d = 0; d < AD; d++)
#pragma unroll
            for (int i = 0; i < TM; i++) aring[d][i] = a_cur[a_off[i] + 2 * d];
        for (int s = 0; s + 1 < NS; s += 2) {
            const float* a1 = next_a_base();
            stage(b0r, b1r, a_cur, a1);
            a_cur = next_a_base();                // past the last stage: wraps to tap 0 (values unused)
            stage(b1r, b0r, a1, a_cur);
        }
        if (NS & 1) stage(b0r, b1r, a_cur, smem);
    }

    // ---- epilogue ----
    if (p.dbg & 4) return;
    if (!POOL) __syncthreads();                   // the LDS patches below overlay the input tile other waves may still read
    // Row r of an accumulator tile belongs to tile row i_r = (r&3) + 8*(r>>2) + 4*lh, i.e. NOT this lane's
    // own `lp` pixel: recompute the pixel of each row from the m-tile mapping.
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int co = (wn * TN + j) * 32 + lp;
        const bool cok = co < p.Cout;
        const float bias = cok ? p.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int mt = wm * TM + i;
            if (POOL) {
                constexpr int WPR = TW / 2, WPI = (TH / 2) * WPR;
                const int Hp = p.H >> 1, Wp = p.W >> 1;
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
                    int w = mt * 8 + 2 * g + lh;                 // window index inside the workgroup tile
                    int im = w / WPI, r = w % WPI;
                    int py = (y0 >> 1) + r / WPR, px = (x0 >> 1) + r % WPR;
                    int gb = b0 + im;
                    if (cok && gb < p.B) p.out[(((size_t)gb * Hp + py) * Wp + px) * p.Cout + co] = mx;
                }
            } else {
                // 32 pixels x 32 channels of f32: through a wave-private LDS patch so that the tile leaves as 16-byte stores
                // (4 consecutive channels of one pixel per lane) instead of 16 scalar stores per lane -- for conv2_1, whose
                // tiles are only 50 MFMAs deep, the scalar stores were half the kernel's time.  Values are unchanged.
                constexpr int EROW = 144;                // bytes per patch row: 32 x 4 + 16 (an odd number of 16-byte slots)
                unsigned char* patch = reinterpret_cast<unsigned char*>(smem) + wave * (32 * EROW);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    float v = acc[i][j][r] + bias;
                    if (p.relu) v = v > 0.f ? v : 0.f;
                    *reinterpret_cast<float*>(patch + ((r & 3) + 8 * (r >> 2) + 4 * lh) * EROW + lp * 4) = v;
                }
                asm volatile("" ::: "memory");           // wave-private patch, in-order LDS: only the compiler must not reorder
                typedef float f32x4h __attribute__((ext_vector_type(4)));
                const int c4 = lane & 7, prow = lane >> 3;
                f32x4h pv[4];
#pragma unroll
                for (int it = 0; it < 4; it++) pv[it] = *reinterpret_cast<const f32x4h*>(patch + (prow + 8 * it) * EROW + c4 * 16);
                asm volatile("" ::: "memory");
                const int cob = (wn * TN + j) * 32 + c4 * 4;
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const int l = mt * 32 + prow + 8 * it;
                    const int im = l / (TH * TW), q = l % (TH * TW);
                    const int gy = y0 + q / TW, gx = x0 + q % TW, gb = b0 + im;
                    if (gb < p.B && cob < p.Cout)
                        *reinterpret_cast<f32x4h*>(p.out + (((size_t)gb * p.H + gy) * p.W + gx) * p.Cout + cob) = pv[it];
                }
            }
        }
    }
    (void)o_img; (void)o_y; (void)o_x;
}

template <int CIN, int CPH, int KH, int KW, int PS, int TH, int TW, int NIMG, int WM, int WN, int TM, int TN, int KU, int NP,
          bool POOL, int PIPE = 0>
static inline hipError_t launch_conv_halo(const HaloParams& p, hipStream_t st) {
    constexpr size_t tile_b = sizeof(float) * ((size_t)NIMG * (TH + 4) * ((TW + 4) * PS + 32) + 32);
    constexpr size_t patch_b = POOL ? 0 : (size_t)WM * WN * 32 * 144;     // epilogue patches overlay the tile
    constexpr size_t smem = tile_b > patch_b ? tile_b : patch_b;
    auto kern = k_conv_halo<CIN, CPH, KH, KW, PS, TH, TW, NIMG, WM, WN, TM, TN, KU, NP, POOL, PIPE>;
    // the attribute is per device: one flag per device this process launches on
    static std::atomic<bool> attr_set[64];
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev_]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)smem);
        if (e != hipSuccess) return e;
        attr_set[dev_] = true;
    }
    int groups = (p.B + NIMG - 1) / NIMG;
    groups = (groups + 7) / 8 * 8;                // see the blockIdx decode in the kernel
    dim3 grid((unsigned)(groups * (p.H / TH) * (p.W / TW)));
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, st, p);
    return hipGetLastError();
}
