// conv_halo_cc.h -- fp32 MFMA convolution for WideResNet's stride-1 3x3 convs (Cin = 160 / 320 / 640).
//
// conv_halo.h's data movement (haloed input tile in LDS, weights streamed global -> registers one stage
// ahead, no barrier inside the MFMA loop) does not fit these layers as is: a tile with all Cin channels
// is larger than LDS.  So the contraction is walked in CHANNEL CHUNKS of CC = 32:
//
//     k = (chunk, ky, kx, ci mod 32)         weights packed in that order by nets.hip::make_layer
//
// and per chunk the workgroup stages the (TH+2)x(TW+2) x 32-channel slice of its tile (two barriers per
// chunk = per 9*16 = 144 k-pairs, against one per 16 k-pairs in conv_igemm.h).  The CU's second resident
// workgroup covers the staging.  The B stream runs across chunk boundaries without draining.
// Every output is still ONE fmaf chain in that k order; oracle/graphite_oracle.c::conv_nhwc uses the same
// order for these layers (stride 1, 3x3, Cin % 32 == 0), and conv_igemm.h can walk it too (ConvParams::kchunk)
// for the N tiles this file does not instantiate, so the two kernels stay bit-identical.
// Epilogue as conv_igemm.h: + bias, + residual, ReLU, raw store, relu(v*scale+shift) store.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_igemm.h"

struct HaloCcParams {
    const float* in;      // [B,H,W,Cin] NHWC
    const float* wgt;     // [K][NP], k = (chunk, ky, kx, ci%CC)
    const float* bias;    // [NP] or null
    const float* res;     // [B,H,W,Cout] or null
    const float* scale;   // [NP] folded BN of the consumer, or null
    const float* shift;
    float* out_raw;       // [B,H,W,Cout] or null
    float* out_act;       // [B,H,W,Cout] or null
    int B, H, W, Cin, Cout, NP;
    int relu;
};

// KS: kernel size (odd); CC: channels per chunk; TH,TW: output tile; NIMG: images per workgroup;
// WM x WN wavefronts, TM x TN 32x32 tiles each; RP: pad LDS rows.
// B prefetch: two alternating register sets of KU k-pairs, the set for stage s+1 refilled while stage s computes.
// MODEV & 3 = 0: A read from LDS just in time; 1: A read one k-pair ahead; 2: as 1 with the instruction interleave
// pinned by sched_group_barrier.  MODEV & 4: 16-byte B loads / stores through a column remap.
template <int KS, int CC, int TH, int TW, int NIMG, int WM, int WN, int TM, int TN, int KU, bool RP, int MODEV>
__global__ __launch_bounds__(WM* WN * 64, 2) void k_conv_halo_cc(HaloCcParams p) {
    constexpr int MODE = MODEV & 3;
    constexpr bool VEC = (MODEV & 4) != 0;
    constexpr int NT = WM * WN * 64;
    constexpr int HALO = KS / 2;
    constexpr int PS = CC + 1;                    // pixel stride: a half-wave's 32 pixels hit 32 banks
    constexpr int THH = TH + 2 * HALO, TWH = TW + 2 * HALO;
    constexpr int XR = TW < 32 ? TW : 32;
    constexpr int RPAD = (RP && XR < 32) ? ((XR - (TWH * PS) % 32) + 32) % 32 : 0;
    constexpr int RS = TWH * PS + RPAD;           // LDS row stride
    constexpr int IMGS = THH * RS;                // LDS image stride
    constexpr int MT = NIMG * TH * TW / 32;
    constexpr int KK_TAP = CC / 2;
    constexpr int SUBS = KK_TAP / KU;
    constexpr int NS = KS * KS * SUBS;            // operand stages per chunk
    static_assert(MT == WM * TM, "m tiling");
    static_assert(KK_TAP % KU == 0, "KU");
    static_assert(NS % 2 == 0, "stages per chunk must be even (two alternating B register sets)");
    static_assert(CC % 4 == 0, "CC");

    extern __shared__ float smem[];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int lp = lane & 31, lh = lane >> 5;
    const int tiles_x = p.W / TW, tiles_y = p.H / TH;
    // 1-D grid, decoded XCD-aware (workgroups are dealt round-robin to the 8 XCDs, one L2 each): the N tiles of one
    // spatial tile, then the spatial tiles of one image group, run back to back on the SAME XCD, so the input slice
    // they share (whole tile / halo rows) comes from that XCD's L2.
    int bid, ntile;
    {
        const int tpi = tiles_x * tiles_y, groups = (p.B + NIMG - 1) / NIMG, nts = p.NP / (WN * TN * 32);
        const int xcd = blockIdx.x & 7;
        int j = blockIdx.x >> 3;
        ntile = j % nts; j /= nts;
        const int g = (j / tpi) * 8 + xcd;
        if (g >= groups) return;                  // grid is rounded up to a multiple of 8 groups; whole workgroup leaves
        bid = g * tpi + j % tpi;
    }
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const int b0 = bid * NIMG;
    const int y0 = ty * TH, x0 = tx * TW;
    const int n0 = ntile * (WN * TN * 32);

    // ---- stage channels [c0, c0+CC) of the haloed tile (zero outside the image / batch) ----
    auto stage_tile = [&](int c0) {
        constexpr int SEG = CC / 4;
        constexpr int TOT = NIMG * THH * TWH * SEG;
        constexpr int IT = (TOT + NT - 1) / NT;
        constexpr int UB = 6;
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
                    v[u] = *reinterpret_cast<const float4*>(p.in + (((size_t)gb * p.H + gy) * p.W + gx) * p.Cin + c0 + seg * 4);
            }
#pragma unroll
            for (int u = 0; u < UB; u++) {
                if (dst[u] >= 0) {
                    float* d = smem + dst[u];
                    d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                }
            }
        }
    };

    // ---- per-lane A offsets: pixel `lp` of each of my TM m-tiles ----
    int a_off[TM];
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int l = (wm * TM + i) * 32 + lp;
        const int im = l / (TH * TW), r = l % (TH * TW);
        a_off[i] = im * IMGS + (r / TW) * RS + (r % TW) * PS + lh;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // B: lane reads W[k = 2kk + lh][n0 + j*32 + lp], linear in k over the whole (chunk, tap, ci) walk.
    // The loads are UNCONDITIONAL (the stream runs one stage past the end of the weights, which make_layer pads
    // with zero rows): a conditional prefetch makes the compiler's vmcnt bookkeeping assume the shortest path, and
    // the MFMAs of a stage then wait for the prefetch issued during that very stage.
    // Column mapping of the wavefront's TN n-tiles: tile j, lane lp <-> column j*32 + lp.  With VEC (MODEV & 4) tiles
    // are taken in groups of four whose lane `lp` owns the four CONSECUTIVE columns g*128 + 4*lp .. +3 (one 16-byte
    // load per k-pair, 16-byte stores in the epilogue); measured slower than dword loads (profiles/r01_conv_tune2.md).
    constexpr int TN4 = VEC ? TN / 4 * 4 : 0;
    constexpr int NLD = TN4 / 4 + (TN - TN4);     // B loads per k-pair
    const float* bptr = p.wgt + (size_t)lh * p.NP + n0 + (wn * TN) * 32;
    const size_t bstep = (size_t)2 * p.NP;
    float b0r[KU][TN], b1r[KU][TN];
    auto load_b_pair = [&](float (&b)[TN]) {
#pragma unroll
        for (int g = 0; g < TN4 / 4; g++) {
            const float4 q = *reinterpret_cast<const float4*>(bptr + g * 128 + 4 * lp);
            b[4 * g + 0] = q.x; b[4 * g + 1] = q.y; b[4 * g + 2] = q.z; b[4 * g + 3] = q.w;
        }
#pragma unroll
        for (int j = TN4; j < TN; j++) b[j] = bptr[j * 32 + lp];
        bptr += bstep;
    };
    int nky = 0, nkx = 0, nsub = 0;               // tap / sub-stage of the NEXT stage to start
    float anext[TM];                              // A values of that stage's first k-pair (MODE >= 1)
    auto next_a_base = [&]() {
        const float* a_tap = smem + nky * RS + nkx * PS + nsub * (2 * KU);
        if (++nsub == SUBS) { nsub = 0; if (++nkx == KS) { nkx = 0; if (++nky == KS) nky = 0; } }
        return a_tap;
    };
    // one stage: MFMAs on (bc) while (bn) is refilled for the stage after
    auto stage = [&](float (&bc)[KU][TN], float (&bn)[KU][TN], const float* a_tap, const float* a_nxt) {
#pragma unroll
        for (int kk = 0; kk < KU; kk++) {
            float av[TM];
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < TM; i++) av[i] = a_tap[a_off[i] + 2 * kk];
            } else {
#pragma unroll
                for (int i = 0; i < TM; i++) av[i] = anext[i];
#pragma unroll
                for (int i = 0; i < TM; i++) anext[i] = kk + 1 < KU ? a_tap[a_off[i] + 2 * (kk + 1)] : a_nxt[a_off[i]];
            }
            load_b_pair(bn[kk]);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bc[kk][j], acc[i][j], 0, 0, 0);
            if (MODE == 2) {                       // pin the interleave: A reads, then (TM MFMAs, 1 B load) x NLD
                __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
#pragma unroll
                for (int l = 0; l < NLD; l++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
        }
    };

    const int nchunk = p.Cin / CC;
#pragma unroll
    for (int kk = 0; kk < KU; kk++) load_b_pair(b0r[kk]);
    for (int ch = 0; ch < nchunk; ch++) {
        if (ch) __syncthreads();                  // every wave is done reading the previous slice
        stage_tile(ch * CC);
        __syncthreads();
        nky = nkx = nsub = 0;
        const float* a_cur = next_a_base();
        if (MODE >= 1) {
#pragma unroll
            for (int i = 0; i < TM; i++) anext[i] = a_cur[a_off[i]];
        }
        for (int s = 0; s < NS; s += 2) {
            const float* a1 = next_a_base();
            stage(b0r, b1r, a_cur, a1);
            a_cur = next_a_base();                // after the chunk's last stage: wraps to tap 0 (value unused)
            stage(b1r, b0r, a1, a_cur);
        }
    }

    // ---- epilogue: lane holds rows (r&3) + 8*(r>>2) + 4*lh of each tile; columns per the mapping above ----
    auto finish = [&](float v, float res, float bias) {
        v = v + bias;
        v = v + res;
        return p.relu ? (v > 0.f ? v : 0.f) : v;
    };
    auto activate = [&](float v, float sc, float sh) {
        float w = fmaf(v, sc, sh);
        return w > 0.f ? w : 0.f;
    };
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto out_row = [&](int mt, int r, size_t* row) {     // NHWC offset of accumulator row r of m-tile mt; false = outside the batch
        const int l = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int im = l / (TH * TW), q = l % (TH * TW);
        const int gy = y0 + q / TW, gx = x0 + q % TW, gb = b0 + im;
        *row = (((size_t)gb * p.H + gy) * p.W + gx) * p.Cout;
        return gb < p.B;
    };
#pragma unroll
    for (int g = 0; g < TN4 / 4; g++) {
        const int co = n0 + (wn * TN) * 32 + g * 128 + 4 * lp;
        if (co >= p.Cout) continue;                               // Cout % 4 == 0 (checked at launch)
        const float4 bi = p.bias ? *reinterpret_cast<const float4*>(p.bias + co) : z4;
        const float4 sc = p.out_act ? *reinterpret_cast<const float4*>(p.scale + co) : z4;
        const float4 sh = p.out_act ? *reinterpret_cast<const float4*>(p.shift + co) : z4;
#pragma unroll
        for (int i = 0; i < TM; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                size_t row;
                if (!out_row(wm * TM + i, r, &row)) continue;
                const float4 rs = p.res ? *reinterpret_cast<const float4*>(p.res + row + co) : z4;
                float4 v;
                v.x = finish(acc[i][4 * g + 0][r], rs.x, bi.x);
                v.y = finish(acc[i][4 * g + 1][r], rs.y, bi.y);
                v.z = finish(acc[i][4 * g + 2][r], rs.z, bi.z);
                v.w = finish(acc[i][4 * g + 3][r], rs.w, bi.w);
                if (p.out_raw) *reinterpret_cast<float4*>(p.out_raw + row + co) = v;
                if (p.out_act) {
                    float4 w;
                    w.x = activate(v.x, sc.x, sh.x); w.y = activate(v.y, sc.y, sh.y);
                    w.z = activate(v.z, sc.z, sh.z); w.w = activate(v.w, sc.w, sh.w);
                    *reinterpret_cast<float4*>(p.out_act + row + co) = w;
                }
            }
        }
    }
    // Scalar-column tiles (all of them unless VEC): an accumulator tile has its 32 channels on the lanes and its 32
    // pixels in the registers, so a direct store is one 4-byte access per value -- 3 x 16 x TM x TN scalar loads /
    // stores per lane.  Each tile goes through a wave-private LDS patch instead (the barrier below retires every read
    // of the slice) and comes back with 4 consecutive channels of one pixel per lane: 16-byte accesses, a quarter of
    // the instructions; the arithmetic per value is unchanged (finish / activate), so the results are bit-identical.
    if (TN4 < TN) {
        constexpr int EROW = 36;                          // floats per patch row: 9 16-byte slots, odd
        static_assert(WM * WN * 32 * EROW <= NIMG * IMGS, "epilogue patches must fit the slice's LDS");
        __syncthreads();
        float* patch = smem + wave * (32 * EROW);
        const int c4 = lane & 7, prow = lane >> 3;
#pragma unroll
        for (int j = TN4; j < TN; j++) {
            const int co4 = n0 + (wn * TN + j) * 32 + c4 * 4;
            const bool cok = co4 < p.Cout;                // Cout % 4 == 0 (checked at launch)
            const float4 bi = (p.bias && cok) ? *reinterpret_cast<const float4*>(p.bias + co4) : z4;
            const float4 sc = (p.out_act && cok) ? *reinterpret_cast<const float4*>(p.scale + co4) : z4;
            const float4 sh = (p.out_act && cok) ? *reinterpret_cast<const float4*>(p.shift + co4) : z4;
#pragma unroll
            for (int i = 0; i < TM; i++) {
#pragma unroll
                for (int r = 0; r < 16; r++) patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * EROW + lp] = acc[i][j][r];
                asm volatile("" ::: "memory");            // wave-private, LDS operations of a wave execute in order
                float4 pv[4];
#pragma unroll
                for (int it = 0; it < 4; it++) pv[it] = *reinterpret_cast<const float4*>(patch + (prow + 8 * it) * EROW + c4 * 4);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const int l = (wm * TM + i) * 32 + prow + 8 * it;
                    const int im = l / (TH * TW), q = l % (TH * TW);
                    const int gy = y0 + q / TW, gx = x0 + q % TW, gb = b0 + im;
                    if (gb >= p.B || !cok) continue;
                    const size_t row = (((size_t)gb * p.H + gy) * p.W + gx) * p.Cout;
                    const float4 rs = p.res ? *reinterpret_cast<const float4*>(p.res + row + co4) : z4;
                    float4 v;
                    v.x = finish(pv[it].x, rs.x, bi.x); v.y = finish(pv[it].y, rs.y, bi.y);
                    v.z = finish(pv[it].z, rs.z, bi.z); v.w = finish(pv[it].w, rs.w, bi.w);
                    if (p.out_raw) *reinterpret_cast<float4*>(p.out_raw + row + co4) = v;
                    if (p.out_act) {
                        float4 w;
                        w.x = activate(v.x, sc.x, sh.x); w.y = activate(v.y, sc.y, sh.y);
                        w.z = activate(v.z, sc.z, sh.z); w.w = activate(v.w, sc.w, sh.w);
                        *reinterpret_cast<float4*>(p.out_act + row + co4) = w;
                    }
                }
            }
        }
    }
}

template <int KS, int CC, int TH, int TW, int NIMG, int WM, int WN, int TM, int TN, int KU, bool RP, int MODE = 0>
static inline hipError_t launch_conv_halo_cc(const HaloCcParams& p, hipStream_t st) {
    constexpr int TWH = TW + 2 * (KS / 2), XR = TW < 32 ? TW : 32;
    constexpr int RPAD = (RP && XR < 32) ? ((XR - (TWH * (CC + 1)) % 32) + 32) % 32 : 0;
    constexpr size_t smem = sizeof(float) * ((size_t)NIMG * (TH + 2 * (KS / 2)) * (TWH * (CC + 1) + RPAD) + 32);
    auto kern = k_conv_halo_cc<KS, CC, TH, TW, NIMG, WM, WN, TM, TN, KU, RP, MODE>;
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
    if (p.H % TH || p.W % TW || p.Cin % CC || p.NP % (WN * TN * 32) || p.Cout % 4 || (p.out_act && !(p.scale && p.shift)))
        return hipErrorInvalidValue;
    long long groups = (p.B + NIMG - 1) / NIMG;
    groups = (groups + 7) / 8 * 8;                // see the blockIdx decode in the kernel
    const long long nblk = groups * (p.H / TH) * (p.W / TW) * (p.NP / (WN * TN * 32));
    if (nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    dim3 grid((unsigned)nblk);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, st, p);
    return hipGetLastError();
}
