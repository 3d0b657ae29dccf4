// conv_halo_cc_bf16.h -- bf16 MFMA convolution for WideResNet's 3x3 convs, stride 1 and 2 (Cin, Cout multiples of 32 / 160):
// the bf16 "throughput mode" twin of conv_halo_cc.h (NOT the parity path).
//
// As conv_halo_cc.h the contraction is walked in channel chunks of CC = 32, k = (chunk, ky, kx, ci mod 32), because a
// tile with all Cin channels does not fit LDS; operands are shaped for v_mfma_f32_32x32x16_bf16 as in conv_halo_bf16.h:
//   * A: per chunk the (TH+2)x(TW+2) x 32-channel slice of the tile, f32 -> bf16 (nearest even) on its way into LDS,
//     pixel stride 80 bytes (5 16-byte slots), row stride chosen for conflict-free ds_read_b128.  The slice of chunk
//     c+1 is loaded into registers while chunk c is computed, so a chunk boundary costs one extra barrier, no latency.
//   * B: bf16 weights in fragment order [k-block][Cout/32][2][32][8]; one stage = one tap (2 k-blocks x 5 n-tiles =
//     10 KB), two LDS buffers, one barrier per stage, register ring three stages deep (9 taps per chunk keep the ring
//     slot of a tap compile-time).  Stage indices are clamped, never branched on.
//   * Epilogue as conv_halo_cc.h (+ residual, raw store, relu(v*scale+shift) store = folded BN of the consumer), but
//     transposed through LDS so that every global access is 16 bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "conv_halo_bf16.h"

struct HaloCcBfParams {
    const void* in;         // [B,H,W,Cin] NHWC, f32 or bf16 (IN_BF)
    const uint16_t* wgt;    // bf16, [K/16][Cout/32][2][32][8], k = (chunk, tap, ci%32)
    const float* res;       // [B,H/stride,W/stride,Cout] or null
    const float* scale;     // [Cout] folded BN of the consumer, or null
    const float* shift;
    float* out_raw;         // [B,H,W,Cout] or null
    void* out_act;          // [B,H/stride,W/stride,Cout] or null; f32 or bf16 (ACT_BF)
    int B, H, W, Cin, Cout; // H, W: input size
};

// STRIDE 1: the slice is one (TH+2) x (TW+2) image per input image.  STRIDE 2 (pad 1): output (y, x) reads input
// (2y+ky-1, 2x+kx-1), so the (2TH+1) x (2TW+1) input region is stored as four PARITY PLANES of (TH+1) x (TW+1) pixels
// (row parity, column parity of the region index): tap (ky, kx) then reads plane (ky&1, kx&1) at (y + ky/2, x + kx/2),
// a stride-1 walk inside the plane with the same conflict-free ds_read_b128 pattern as the stride-1 kernel.
template <int TH, int TW, int NIMG, int STRIDE>
struct HaloCcBfGeom {
    static constexpr int CC = 32;
    static constexpr int THH = STRIDE == 1 ? TH + 2 : 2 * TH + 1;     // input region walked by the staging loop
    static constexpr int TWH = STRIDE == 1 ? TW + 2 : 2 * TW + 1;
    static constexpr int PH = STRIDE == 1 ? TH + 2 : TH + 1;          // rows / columns of one LDS plane
    static constexpr int PW = STRIDE == 1 ? TW + 2 : TW + 1;
    static constexpr int PSB = CC * 2 + 16;
    static constexpr int RWANT = TW == 8 ? 8 : 0;
    static constexpr int RS0 = PW * PSB;
    static constexpr int RSB = RS0 + ((RWANT - (RS0 / 16) % 16 + 16) % 16) * 16;
    static constexpr int PLB = PH * RSB;                              // bytes of one plane
    static constexpr int IMGB = (STRIDE == 1 ? 1 : 4) * PLB;
    static constexpr int A_BYTES = NIMG * IMGB;
};

// WM wavefronts x TM 32-pixel m-tiles cover the NIMG x TH x TW pixels; every wavefront holds TN n-tiles (N = TN*32).
template <int TH, int TW, int NIMG, int WM, int TM, int TN, int OCC, int STRIDE, bool IN_BF, bool ACT_BF>
__global__ __launch_bounds__(WM * 64, OCC) void k_conv_halo_cc_bf16(HaloCcBfParams p) {
    using G = HaloCcBfGeom<TH, TW, NIMG, STRIDE>;
    constexpr int CC = 32, NT = WM * 64;
    constexpr int THH = G::THH, TWH = G::TWH, PSB = G::PSB, RSB = G::RSB, IMGB = G::IMGB, PLB = G::PLB;
    constexpr int KBS = CC / 16;                         // k-blocks per stage (one tap)
    constexpr int STAGE_B = KBS * TN * 1024;
    constexpr int SV = STAGE_B / 16, TV = TN * 64;       // vectors per stage / per k-block
    constexpr int VPT = (SV + NT - 1) / NT;
    constexpr int MT = NIMG * TH * TW / 32;
    constexpr int EPS = IN_BF ? 8 : 4;                   // channels per 16-byte global segment
    constexpr int SEG = CC / EPS, NPX = NIMG * THH * TWH, PSTEP = NT / SEG;
    constexpr int AIT = (NPX + PSTEP - 1) / PSTEP;       // 16-byte loads per thread and chunk slice
    static_assert(MT == WM * TM, "m tiling");
    static_assert(PSTEP / TWH + 1 <= THH, "tile too small for the incremental walk");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_cb[];
    unsigned char* asm_ = smem_cb;
    unsigned char* bsm = smem_cb + G::A_BYTES;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int lp = lane & 31, lh = lane >> 5;
    const int Ho = p.H / STRIDE, Wo = p.W / STRIDE;     // p.H, p.W: input; tiles are output tiles
    const int tiles_x = Wo / TW, tiles_y = Ho / TH;
    int bid, ntile;                                      // XCD-aware order, as conv_halo_cc.h
    {
        const int tpi = tiles_x * tiles_y, groups = (p.B + NIMG - 1) / NIMG, nts = p.Cout / (TN * 32);
        const int xcd = blockIdx.x & 7;
        int j = blockIdx.x >> 3;
        ntile = j % nts; j /= nts;
        const int g = (j / tpi) * 8 + xcd;
        if (g >= groups) return;
        bid = g * tpi + j % tpi;
    }
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const int b0 = bid * NIMG;
    const int y0 = ty * TH, x0 = tx * TW;
    const int n0 = ntile * (TN * 32);
    const int nchunk = p.Cin / CC;
    const int NS = nchunk * 9;

    // ---- B ring ----
    const u32x4* wg = reinterpret_cast<const u32x4*>(p.wgt) + (size_t)(n0 / 32) * 64;
    const size_t kb_stride = (size_t)(p.Cout / 32) * 64;  // vectors per k-block row of the packed weights
    u32x4 breg[3][VPT];
    auto gload = [&](u32x4 (&br)[VPT], int s) {
        if (s >= NS) s = NS - 1;
#pragma unroll
        for (int v = 0; v < VPT; v++) {
            int idx = tid + v * NT;
            if (SV % NT != 0 && idx >= SV) idx = SV - 1;
            br[v] = wg[(size_t)(s * KBS + idx / TV) * kb_stride + idx % TV];
        }
    };
    auto lstore = [&](const u32x4 (&br)[VPT], int buf) {
#pragma unroll
        for (int v = 0; v < VPT; v++) {
            const int idx = tid + v * NT;
            if (SV % NT == 0 || idx < SV) *reinterpret_cast<u32x4*>(bsm + buf * STAGE_B + idx * 16) = br[v];
        }
    };

    // ---- A slice: thread = (4-channel segment, pixel walking the tile in steps of NT/SEG) ----
    u32x4 areg[AIT];
    const int seg = tid % SEG, px0 = tid / SEG;
    auto aload = [&](int ch) {
        if (ch >= nchunk) ch = nchunk - 1;
        const u32x4* src = reinterpret_cast<const u32x4*>(p.in) + (ch * CC) / EPS + seg;
        const int cvec = p.Cin / EPS;                     // 16-byte vectors per pixel
        int px = px0, hx = px0 % TWH, hy = (px0 / TWH) % THH, im = px0 / (TWH * THH);
#pragma unroll
        for (int u = 0; u < AIT; u++) {
            const int gy = STRIDE * y0 + hy - 1, gx = STRIDE * x0 + hx - 1, gb = b0 + im;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (px < NPX && gb < p.B && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
                v = src[(((size_t)gb * p.H + gy) * p.W + gx) * cvec];
            areg[u] = v;
            px += PSTEP; hx += PSTEP % TWH; hy += PSTEP / TWH;
            if (hx >= TWH) { hx -= TWH; hy++; }
            if (hy >= THH) { hy -= THH; im++; }
        }
    };
    auto astore = [&]() {
        int px = px0, hx = px0 % TWH, hy = (px0 / TWH) % THH, im = px0 / (TWH * THH);
#pragma unroll
        for (int u = 0; u < AIT; u++) {
            if (px < NPX) {
                const int off = STRIDE == 1 ? hy * RSB + hx * PSB
                                            : ((hy & 1) * 2 + (hx & 1)) * PLB + (hy >> 1) * RSB + (hx >> 1) * PSB;
                unsigned char* d = asm_ + im * IMGB + off + seg * (EPS * 2);
                if constexpr (IN_BF) {
                    *reinterpret_cast<u32x4*>(d) = areg[u];
                } else {
                    const f32x4v f = __builtin_bit_cast(f32x4v, areg[u]);    // whole vector (conv_halo_bf16.h)
                    bf16x4 h;
                    h[0] = (__bf16)f[0]; h[1] = (__bf16)f[1]; h[2] = (__bf16)f[2]; h[3] = (__bf16)f[3];
                    *reinterpret_cast<bf16x4*>(d) = h;
                }
            }
            px += PSTEP; hx += PSTEP % TWH; hy += PSTEP / TWH;
            if (hx >= TWH) { hx -= TWH; hy++; }
            if (hy >= THH) { hy -= THH; im++; }
        }
    };

    int a_off[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int l = (wave * TM + i) * 32 + lp;
        const int im = l / (TH * TW), r = l % (TH * TW);
        a_off[i] = im * IMGB + (r / TW) * RSB + (r % TW) * PSB + lh * 16;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    aload(0);
    gload(breg[0], 0);
    gload(breg[1], 1);
    gload(breg[2], 2);
    astore();
    aload(1);
    lstore(breg[0], 0);
    gload(breg[0], 3);
    __syncthreads();

    for (int ch = 0; ch < nchunk; ch++) {
        bf_static_for<9>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int slot = (t + 1) % 3;             // ring slot of stage s+1 (9 stages per chunk: s % 3 == t % 3)
            const int s = ch * 9 + t;
            constexpr int ky = t / 3, kx = t % 3;
            constexpr int toff = STRIDE == 1 ? ky * RSB + kx * PSB
                                             : ((ky & 1) * 2 + (kx & 1)) * PLB + (ky >> 1) * RSB + (kx >> 1) * PSB;
            const unsigned char* at = asm_ + toff;
            const unsigned char* bb = bsm + (s & 1) * STAGE_B + lane * 16;
#pragma unroll
            for (int c = 0; c < KBS; c++) {
                bf16x8 av[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; i++) av[i] = *reinterpret_cast<const bf16x8*>(at + a_off[i] + c * 32);
#pragma unroll
                for (int j = 0; j < TN; j++) bv[j] = *reinterpret_cast<const bf16x8*>(bb + (c * TN + j) * 1024);
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
            lstore(breg[slot], (s + 1) & 1);
            gload(breg[slot], s + 4);
            __syncthreads();
        });
        // every wave is past its last read of this chunk's slice: the next one (already in registers) takes its place
        astore();
        aload(ch + 2);
        __syncthreads();
    }

    // ---- epilogue: + residual, raw store, relu(v*scale+shift) store ----
    // An accumulator tile has its 32 output channels on the lanes and its 32 pixels in the registers, so storing it
    // directly takes one 4-byte access per value: 3 x 80 scalar loads / stores per lane here, as much time as ~60 of the
    // 45-180 weight stages.  Each 32x32 tile therefore goes through a wave-private 4.5 KB LDS patch (the loop's last
    // barrier has retired every read of the slice and the weight buffers) and comes back with 4 consecutive channels
    // of one pixel per lane: 16-byte accesses, a quarter of the instructions.
    {
        constexpr int EROW = 144;                        // bytes per pixel row of the patch: 9 slots, odd
        static_assert(NT / 64 * 32 * EROW <= G::A_BYTES + 2 * STAGE_B, "epilogue patches must fit the kernel's LDS");
        unsigned char* patch = smem_cb + wave * (32 * EROW);
        const int c4 = lane & 7, prow = lane >> 3;       // read side: channels 4*c4 .. +3 of pixel rows prow + 8*it
        typedef float f32x4e __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int co4 = n0 + j * 32 + c4 * 4;
            f32x4e sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (p.out_act) {
                sc = *reinterpret_cast<const f32x4e*>(p.scale + co4);
                sh = *reinterpret_cast<const f32x4e*>(p.shift + co4);
            }
#pragma unroll
            for (int i = 0; i < TM; i++) {
#pragma unroll
                for (int r = 0; r < 16; r++)
                    *reinterpret_cast<float*>(patch + ((r & 3) + 8 * (r >> 2) + 4 * lh) * EROW + lp * 4) = acc[i][j][r];
                // the patch is private to this wavefront and its LDS operations execute in order: only the compiler has to
                // be kept from moving the reads above the writes (and the next tile's writes above these reads)
                asm volatile("" ::: "memory");
                f32x4e pv[4];
#pragma unroll
                for (int it = 0; it < 4; it++) pv[it] = *reinterpret_cast<const f32x4e*>(patch + (prow + 8 * it) * EROW + c4 * 16);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const int rowi = prow + 8 * it;
                    f32x4e v = pv[it];
                    const int l = (wave * TM + i) * 32 + rowi;
                    const int im = l / (TH * TW), q = l % (TH * TW);
                    const int gy = y0 + q / TW, gx = x0 + q % TW, gb = b0 + im;
                    if (gb >= p.B) continue;
                    const size_t off = (((size_t)gb * Ho + gy) * Wo + gx) * p.Cout + co4;
                    if (p.res) v = v + *reinterpret_cast<const f32x4e*>(p.res + off);
                    if (p.out_raw) *reinterpret_cast<f32x4e*>(p.out_raw + off) = v;
                    if (p.out_act) {
                        f32x4e w;
#pragma unroll
                        for (int e = 0; e < 4; e++) { const float t = fmaf(v[e], sc[e], sh[e]); w[e] = t > 0.f ? t : 0.f; }
                        if constexpr (ACT_BF) {
                            bf16x4 h;
                            h[0] = (__bf16)w[0]; h[1] = (__bf16)w[1]; h[2] = (__bf16)w[2]; h[3] = (__bf16)w[3];
                            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.out_act) + off) = h;
                        } else {
                            *reinterpret_cast<f32x4e*>(reinterpret_cast<float*>(p.out_act) + off) = w;
                        }
                    }
                }
            }
        }
    }
}

template <int TH, int TW, int NIMG, int WM, int TM, int TN, int OCC, int STRIDE = 1, bool IN_BF = false, bool ACT_BF = false>
static inline hipError_t launch_conv_halo_cc_bf16(const HaloCcBfParams& p, hipStream_t st) {
    using G = HaloCcBfGeom<TH, TW, NIMG, STRIDE>;
    constexpr size_t smem = G::A_BYTES + 2 * (size_t)(2 * TN * 1024);
    static_assert(smem <= 160 * 1024, "LDS");
    if (p.H % (TH * STRIDE) || p.W % (TW * STRIDE) || p.Cin % 32 || p.Cout % (TN * 32) || (p.out_act && !(p.scale && p.shift))) return hipErrorInvalidValue;
    auto kern = k_conv_halo_cc_bf16<TH, TW, NIMG, WM, TM, TN, OCC, STRIDE, IN_BF, ACT_BF>;
    static std::atomic<bool> attr_set[64];                  // the attribute is per device
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev_]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)smem);
        if (e != hipSuccess) return e;
        attr_set[dev_] = true;
    }
    long long groups = (p.B + NIMG - 1) / NIMG;
    groups = (groups + 7) / 8 * 8;
    const long long nblk = groups * (p.H / STRIDE / TH) * (p.W / STRIDE / TW) * (p.Cout / (TN * 32));
    if (nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(WM * 64), smem, st, p);
    return hipGetLastError();
}
