// conv_halo_bf16.h -- bf16 MFMA convolution for GTSRBNet's stride-1 5x5 convs (the "throughput mode" of the
// north star: bf16 operands, f32 accumulation).  NOT the parity path: the fp32 kernels of conv_halo.h are.
//
// Same data movement idea as conv_halo.h -- the haloed input tile is staged in LDS once per workgroup -- with the
// operands shaped for v_mfma_f32_32x32x16_bf16 (8 k-values per lane, one ds_read_b128 each):
//   * A: the tile is converted f32 -> bf16 (round to nearest even) while it is staged, or copied when the producing
//     layer already stored bf16 (IN_BF / OUT_BF: the values are the same, only the HBM traffic halves); pixel stride
//     CIN*2+16 bytes (an odd number of 16-byte slots) and a row stride chosen so that the 16-lane groups of
//     ds_read_b128 hit 16 distinct slots for every pixel order used here (linear 16- and 8-wide tiles, 2x2 pool
//     windows).  The staging registers are native ext_vector_type vectors: arrays of HIP's uint4 / float4 structs
//     ended up in scratch memory or cost 60 registers here.
//   * B: weights are packed at load as [k-block of 16][32-wide n tile][k half][cout][8 k] bf16, i.e. in the exact
//     order the lanes read them, and streamed global -> registers -> LDS in stages of KBS k-blocks: two LDS buffers,
//     one workgroup barrier per stage, and a register ring of PF+1 stages -- a stage is only 0.2-0.5 us of MFMA work,
//     well under the L2 latency, so the loads of stage s+PF+2 are issued while stage s is computed (an MFMA here is 16x
//     faster than the fp32 one, so the weights can no longer come straight from L2 into every wave's registers).
//     The stream reads stage indices clamped to the last stage instead of branching (conv_halo.h: a conditional
//     prefetch costs a full vmcnt drain).
//   * k order: (tap = ky*5+kx, ci); accumulation order inside an MFMA is the hardware's.
//   * Epilogue as conv_halo.h: + bias, ReLU, optional 2x2 max-pool in registers, f32 (or bf16) NHWC store.
//   * __launch_bounds__ asks for the waves per SIMD that the LDS footprint admits (at most 4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "conv_igemm.h"

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int... I, class F>
__device__ __forceinline__ void bf_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void bf_static_for(F&& f) {
    bf_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct HaloBfParams {
    const void* in;         // [B,H,W,CIN] NHWC, f32 (IN_BF = false) or bf16 (IN_BF = true)
    const uint16_t* wgt;    // bf16, [K/16][COUT/32][2][32][8]
    const float* bias;      // [COUT]
    void* out;              // [B,H,W,COUT] or pooled [B,H/2,W/2,COUT], f32 or bf16 (OUT_BF)
    int B, H, W, relu;
    int dbg;                // tuning aid (GRAPHITE_BF_DBG, scripts/bf16_phases.sh): 1 = skip the tile staging, 2 = skip the MFMA loop, 4 = skip the epilogue (wrong results)
};

// geometry shared by the kernel and its launcher
template <int CIN, int COUT, int TH, int TW, int NIMG, int KBS, bool POOL>
struct HaloBfGeom {
    static constexpr int THH = TH + 4, TWH = TW + 4;
    static constexpr int PSB = CIN * 2 + 16;                       // bytes per pixel
    static constexpr int RWANT = (POOL || TW == 8) ? 8 : 0;        // row stride in 16-byte slots, mod 16
    static constexpr int RS0 = TWH * PSB;
    static constexpr int RSB = RS0 + ((RWANT - (RS0 / 16) % 16 + 16) % 16) * 16;
    static constexpr int IMGB = THH * RSB;
    static constexpr int A_BYTES = NIMG * IMGB;
    static constexpr int NTILE = COUT / 32;
    static constexpr int STAGE_B = KBS * NTILE * 1024;
    static constexpr int LDS_BYTES = A_BYTES + 2 * STAGE_B;
    // wavefronts per SIMD that LDS admits (HIP's second __launch_bounds__ argument) -> the register budget the
    // compiler is asked to respect: left alone it spends 170-200 registers on some instantiations and halves the
    // resident waves.  Capped at 4 (128 registers).
    static constexpr int occ(int nt) {
        const int w = (160 * 1024 / LDS_BYTES) * (nt / 64) / 4;
        return w < 1 ? 1 : w > 4 ? 4 : w;
    }
};

template <int CIN, int COUT, int TH, int TW, int NIMG, int WM, int WN, int TM, int TN, int KBS, bool POOL, int PF, bool IN_BF, bool OUT_BF, int PD>
__global__ __launch_bounds__(WM* WN * 64, (HaloBfGeom<CIN, COUT, TH, TW, NIMG, KBS, POOL>::occ(WM * WN * 64)))
void k_conv_halo_bf16(HaloBfParams p) {
    using G = HaloBfGeom<CIN, COUT, TH, TW, NIMG, KBS, POOL>;
    constexpr int NT = WM * WN * 64;
    constexpr int THH = G::THH, TWH = G::TWH, PSB = G::PSB, RSB = G::RSB, IMGB = G::IMGB;
    constexpr int NTILE = G::NTILE, STAGE_B = G::STAGE_B;
    constexpr int KB_TAP = CIN / 16;                    // k-blocks per tap
    constexpr int NKB = 25 * KB_TAP;
    constexpr int NS = NKB / KBS;                        // stages
    constexpr int SV = STAGE_B / 16;                     // 16-byte vectors per stage
    constexpr int VPT = (SV + NT - 1) / NT;
    constexpr int MT = NIMG * TH * TW / 32;
    static_assert(MT == WM * TM, "m tiling");
    static_assert(NTILE == WN * TN, "n tiling");
    static_assert(CIN % 16 == 0 && COUT % 32 == 0, "channels");
    static_assert(NKB % KBS == 0 && (KBS % KB_TAP == 0 || KB_TAP % KBS == 0), "stage size");
    static_assert((PSB / 16) % 2 == 1, "pixel stride must be an odd number of 16-byte slots");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_bf[];
    unsigned char* asm_ = smem_bf;
    unsigned char* bsm = smem_bf + G::A_BYTES;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int lp = lane & 31, lh = lane >> 5;
    const int tiles_x = p.W / TW, tiles_y = p.H / TH;
    int bid;                                             // XCD-aware tile order, as conv_halo.h
    {
        const int tpi = tiles_x * tiles_y, groups = (p.B + NIMG - 1) / NIMG;
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int g = (j / tpi) * 8 + xcd;
        if (g >= groups) return;
        bid = g * tpi + j % tpi;
    }
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const int b0 = bid * NIMG;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- weight stage 0 on its way while the tile is staged ----
    const u32x4* wg = reinterpret_cast<const u32x4*>(p.wgt);
    u32x4 breg[PF + 1][VPT];                            // slot s % (PF+1) holds stage s
    auto gload = [&](u32x4 (&br)[VPT], int s) {
        const u32x4* src = wg + (size_t)(s < NS ? s : NS - 1) * SV;
#pragma unroll
        for (int v = 0; v < VPT; v++) {
            const int idx = tid + v * NT;
            br[v] = src[(SV % NT == 0 || idx < SV) ? idx : SV - 1];
        }
    };
    auto lstore = [&](const u32x4 (&br)[VPT], int buf) {
#pragma unroll
        for (int v = 0; v < VPT; v++) {
            const int idx = tid + v * NT;
            if (SV % NT == 0 || idx < SV) *reinterpret_cast<u32x4*>(bsm + buf * STAGE_B + idx * 16) = br[v];
        }
    };
#pragma unroll
    for (int d = 0; d <= PF; d++) gload(breg[d], d);

    // ---- stage the haloed input tile (zero outside the image / batch): f32 -> bf16 on the way, or a plain copy when
    // the producer already stored bf16 (same values: it rounded the same f32 numbers) ----
    // Thread = (16-byte segment of a pixel's channels, pixel); the pixel walks the tile in steps of NT/SEG.
    {
        constexpr int EPS = IN_BF ? 8 : 4;               // channels per 16-byte global segment
        constexpr int SEG = CIN / EPS;
        static_assert(NT % SEG == 0, "segments per pixel must divide the workgroup");
        constexpr int PSTEP = NT / SEG;                   // pixels per pass
        constexpr int NPX = NIMG * THH * TWH;
        constexpr int IT = (NPX + PSTEP - 1) / PSTEP;
        constexpr int UB0 = IT <= 26 ? IT : (IT + 1) / 2 <= 26 ? (IT + 1) / 2 : 16;   // loads in flight per thread
        constexpr int UB = (IN_BF && UB0 > 4) ? 4 : UB0;   // bf16 input: a single fully unrolled batch (UB == IT) makes hipcc spill (228-308 B of scratch per lane, 3x the kernel time)
        static_assert(PSTEP / TWH + 1 <= THH, "tile too small for the incremental walk");
        const int seg = tid % SEG;
        int px = tid / SEG;
        int hx = px % TWH, hy = (px / TWH) % THH, im = px / (TWH * THH);
        const u32x4* src = reinterpret_cast<const u32x4*>(p.in) + seg;
        if (!(p.dbg & 1))
#pragma unroll
        for (int it0 = 0; it0 < IT; it0 += UB) {
            u32x4 v[UB];
            int dst[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int gy = y0 + hy - 2, gx = x0 + hx - 2, gb = b0 + im;
                v[u] = u32x4{0u, 0u, 0u, 0u};
                dst[u] = (it0 + u < IT && px < NPX) ? im * IMGB + hy * RSB + hx * PSB + seg * (EPS * 2) : -1;
                if (dst[u] >= 0 && gb < p.B && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
                    v[u] = src[(((size_t)gb * p.H + gy) * p.W + gx) * SEG];
                px += PSTEP;
                hx += PSTEP % TWH;
                hy += PSTEP / TWH;
                if (hx >= TWH) { hx -= TWH; hy++; }
                if (hy >= THH) { hy -= THH; im++; }
            }
#pragma unroll
            for (int u = 0; u < UB; u++) {
                if (dst[u] >= 0) {
                    if constexpr (IN_BF) {
                        *reinterpret_cast<u32x4*>(asm_ + dst[u]) = v[u];
                    } else {
                        const f32x4v f = __builtin_bit_cast(f32x4v, v[u]);   // whole vector: an element-wise bit_cast miscompiles (ROCm 7.2)
                        bf16x4 h;
                        h[0] = (__bf16)f[0]; h[1] = (__bf16)f[1]; h[2] = (__bf16)f[2]; h[3] = (__bf16)f[3];
                        *reinterpret_cast<bf16x4*>(asm_ + dst[u]) = h;
                    }
                }
            }
        }
    }

    // ---- per-lane A byte offsets: which pixel of the tile row `lp` of each of my m-tiles is ----
    int a_off[TM];
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
        a_off[i] = im * IMGB + y * RSB + x * PSB + lh * 16;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int b_lane = (wn * TN) * 1024 + lane * 16;
    // Fragments of k-block g (flattened over the stages): A from the resident tile, B from LDS buffer (g / KBS) & 1.
    auto read_a = [&](bf16x8 (&av)[TM], int g) {
        const int s = g / KBS, kbl = g % KBS;
        int tap, c;
        if (KBS % KB_TAP == 0) { tap = s * (KBS / KB_TAP) + kbl / KB_TAP; c = kbl % KB_TAP; }
        else { tap = s / (KB_TAP / KBS); c = (s % (KB_TAP / KBS)) * KBS + kbl; }
        const int ky = tap / 5, kx = tap - 5 * ky;
        const unsigned char* at = asm_ + ky * RSB + kx * PSB + c * 32;
#pragma unroll
        for (int i = 0; i < TM; i++) av[i] = *reinterpret_cast<const bf16x8*>(at + a_off[i]);
    };
    auto read_b = [&](bf16x8 (&bv)[TN], int g) {
        const int s = g / KBS, kbl = g % KBS;
        const unsigned char* bb = bsm + (s & 1) * STAGE_B + b_lane;
#pragma unroll
        for (int j = 0; j < TN; j++) bv[j] = *reinterpret_cast<const bf16x8*>(bb + (kbl * NTILE + j) * 1024);
    };

    lstore(breg[0], 0);
    gload(breg[0], PF + 1);
    __syncthreads();                                     // tile + weight stage 0 staged
    // The k loop, flattened over the stages and unrolled at compile time.  Fragments are read PD k-blocks ahead of the
    // MFMAs that use them (PD + 1 register sets): left to itself the compiler reads a k-block's fragments, waits for all of
    // them and only then issues its MFMAs.  A fragments come from the resident tile and are read ahead across a stage
    // boundary too; B fragments of a new stage can only be read after its barrier.
    // Stage boundary after the last k-block of stage s: stage s+1 leaves its registers for the other LDS buffer (last read
    // in stage s-1, before the previous barrier) and those registers take the loads of stage s+PF+2.
    static_assert(PD >= 0 && PD <= KBS, "read-ahead distance");
    constexpr int NSET = PD + 1;
    bf16x8 fa[NSET][TM], fb[NSET][TN];
    if constexpr (PD == 0) {                             // no read-ahead: the compiler's own schedule (conv4_2: one workgroup
        if (!(p.dbg & 2))                                // of 112 KB per CU, measured 4 % faster this way)
        bf_static_for<NS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int slot = (s + 1) % (PF + 1);
#pragma unroll
            for (int kbl = 0; kbl < KBS; kbl++) {
                read_a(fa[0], s * KBS + kbl);
                read_b(fb[0], s * KBS + kbl);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
            }
            lstore(breg[slot], (s + 1) & 1);
            gload(breg[slot], s + PF + 2);
            __syncthreads();
        });
    } else if (!(p.dbg & 2)) {
        bf_static_for<PD>([&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if constexpr (d < NKB) { read_a(fa[d % NSET], d); read_b(fb[d % NSET], d); }   // PD <= KBS: all in stage 0
        });
        bf_static_for<NKB>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int cur = g % NSET;
            constexpr int s = g / KBS;
            constexpr bool last_of_stage = (g % KBS) == KBS - 1;
            if constexpr (g + PD < NKB) {
                read_a(fa[(g + PD) % NSET], g + PD);
                if constexpr ((g + PD) / KBS == s) read_b(fb[(g + PD) % NSET], g + PD);
            }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
            if constexpr (last_of_stage) {
                constexpr int slot = (s + 1) % (PF + 1);
                lstore(breg[slot], (s + 1) & 1);
                gload(breg[slot], s + PF + 2);
                __syncthreads();
                // B fragments of the new stage that the read-ahead had to leave out
                bf_static_for<PD>([&](auto dc) {
                    constexpr int g2 = g + 1 + decltype(dc)::value;
                    if constexpr (g2 < NKB && g2 / KBS == s + 1 && g2 <= g + PD) read_b(fb[g2 % NSET], g2);
                });
            }
            __builtin_amdgcn_sched_barrier(0);           // keep the read-ahead where it is written, not next to its use
        });
    }

    // ---- epilogue: + bias, ReLU, optional 2x2 max-pool, NHWC store ----
    // An accumulator tile has its 32 output channels on the lanes and its pixels in the registers: stored directly that is
    // one 2- or 4-byte access per value (16 per tile and lane), and a store costs the wave about as much as three MFMAs.
    // Each tile therefore goes through a wave-private LDS patch (the loop's last barrier has retired every read of the
    // input tile and of the weight buffers) and leaves as 16-byte stores: 8 bf16 (or 4 f32) channels of one pixel per lane.
    if (!(p.dbg & 4)) {
        constexpr int EROW = 144;                        // bytes per patch row: 32 channels x 4 B + 16 (9 slots, odd)
        static_assert(NT / 64 * 32 * EROW <= G::LDS_BYTES, "epilogue patches must fit the kernel's LDS");
        unsigned char* patch = smem_bf + wave * (32 * EROW);
        constexpr int CPV = OUT_BF ? 8 : 4;              // channels per 16-byte store
        constexpr int GPR = 32 / CPV;                    // lane groups per patch row
        constexpr int RPP = 64 / GPR;                    // patch rows one pass of the wave covers
        const int cg = lane % GPR, prow = lane / GPR;
        constexpr int Hp_div = POOL ? 2 : 1;
        const int Ho = p.H / Hp_div, Wo = p.W / Hp_div;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int co = (wn * TN + j) * 32 + lp;
            const float bias = p.bias[co];
            const int cob = (wn * TN + j) * 32 + cg * CPV;
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int mt = wm * TM + i;
                constexpr int ROWS = POOL ? 8 : 32;      // patch rows in use: pool windows or pixels of the tile
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
                        *reinterpret_cast<float*>(patch + (2 * g + lh) * EROW + lp * 4) = mx;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        float v = acc[i][j][r] + bias;
                        if (p.relu) v = v > 0.f ? v : 0.f;
                        *reinterpret_cast<float*>(patch + ((r & 3) + 8 * (r >> 2) + 4 * lh) * EROW + lp * 4) = v;
                    }
                }
                // the patch is private to this wavefront and its LDS operations execute in order: only the compiler has to
                // be kept from moving the reads above the writes (and the next tile's writes above these reads)
                asm volatile("" ::: "memory");
                constexpr int PASSES = (ROWS + RPP - 1) / RPP;
                f32x4v pv[PASSES][OUT_BF ? 2 : 1];
#pragma unroll
                for (int it = 0; it < PASSES; it++) {
                    const int row = prow + RPP * it;
                    const unsigned char* src = patch + (row < ROWS ? row : 0) * EROW + cg * (CPV * 4);
                    pv[it][0] = *reinterpret_cast<const f32x4v*>(src);
                    if constexpr (OUT_BF) pv[it][1] = *reinterpret_cast<const f32x4v*>(src + 16);
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int it = 0; it < PASSES; it++) {
                    const int row = prow + RPP * it;
                    if (row >= ROWS) continue;
                    int gb, gy, gx;
                    if (POOL) {
                        constexpr int WPR = TW / 2, WPI = (TH / 2) * WPR;
                        const int w = mt * 8 + row;
                        const int im = w / WPI, r = w % WPI;
                        gy = (y0 >> 1) + r / WPR; gx = (x0 >> 1) + r % WPR; gb = b0 + im;
                    } else {
                        const int l = mt * 32 + row;
                        const int im = l / (TH * TW), q = l % (TH * TW);
                        gy = y0 + q / TW; gx = x0 + q % TW; gb = b0 + im;
                    }
                    if (gb >= p.B) continue;
                    const size_t off = (((size_t)gb * Ho + gy) * Wo + gx) * COUT + cob;
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
}

template <int CIN, int COUT, int TH, int TW, int NIMG, int WM, int WN, int TM, int TN, int KBS, bool POOL, int PF = 3, bool IN_BF = false,
          bool OUT_BF = false, int PD = 2>
static inline hipError_t launch_conv_halo_bf16(const HaloBfParams& p, hipStream_t st) {
    using G = HaloBfGeom<CIN, COUT, TH, TW, NIMG, KBS, POOL>;
    constexpr size_t smem = G::LDS_BYTES;
    static_assert(smem <= 160 * 1024, "LDS");
    if (p.H % TH || p.W % TW) return hipErrorInvalidValue;
    auto kern = k_conv_halo_bf16<CIN, COUT, TH, TW, NIMG, WM, WN, TM, TN, KBS, POOL, PF, IN_BF, OUT_BF, PD>;
    static std::atomic<bool> attr_set[64];                  // the attribute is per device
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev_]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)smem);
        if (e != hipSuccess) return e;
        attr_set[dev_] = true;
    }
    int groups = (p.B + NIMG - 1) / NIMG;
    groups = (groups + 7) / 8 * 8;
    dim3 grid((unsigned)(groups * (p.H / TH) * (p.W / TW)));
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, st, p);
    return hipGetLastError();
}
