// pg.hip -- the PatchGuard robust classifier's pixel-side kernels (reference
// patchguard/patchguard_predict.py:21-89) on gfx950:
//
//   k_pg_pre      :42-44 + :30-32   z = (x + 0.5) * 255 -> F.interpolate(z, D, 'bicubic') / 255 -> Normalize(mean, std)
//                 REFERENCE CODE (torch's upsample_bicubic2d: A = -0.75, half-pixel centres, border taps clamped);
//                 fused with BagNet's first layer (conv1, 1x1 3 -> 64, no bias / BN / ReLU) when it feeds the classifier,
//                 so the 192 x 192 x 3 image never exists in HBM.
//   k_pg_masking  :48   masking_defense(local_feature, thres, window_shape) -- THIRD-PARTY code (inspire-group/PatchGuard,
//                 utils/defense_utils.py), restated from PatchGuard's Algorithm 1; "parity unpinned" (oracle: or_masking_defense)
//   k_pg_mean     :59,83  aggregation='mean' of PatchGuardNoDefense
//
// All of it is bandwidth / latency work: no MFMA.  Every float operation below is written so that it rounds exactly as
// oracle/graphite_oracle.c does (the library is built with -ffp-contract=off; sums are sequential and in the same order).
#include "common.h"

__device__ __forceinline__ float pg_cc1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float pg_cc2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// taps of one destination coordinate (ATen UpSample.h: area_pixel_compute_source_index(cubic), guard_index_and_lambda,
// get_cubic_upsample_coefficients, upsample_get_value_bounded)
__device__ __forceinline__ void pg_taps(int d, int S, float scale, int (&idx)[4], float (&w)[4]) {
    const float A = -0.75f;
    float real = scale * ((float)d + 0.5f) - 0.5f;
    float fl = floorf(real);
    int in = (int)fl;
    float lam = real - fl;
    lam = lam < 0.f ? 0.f : (lam > 1.f ? 1.f : lam);
    w[0] = pg_cc2(lam + 1.f, A);
    w[1] = pg_cc1(lam, A);
    float l2 = 1.f - lam;
    w[2] = pg_cc1(l2, A);
    w[3] = pg_cc2(l2 + 1.f, A);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int q = in - 1 + j;
        idx[j] = q < 0 ? 0 : (q > S - 1 ? S - 1 : q);
    }
}

struct PgNorm { float mean[3], std[3]; };

// One thread per destination pixel, 256 pixels of one image per workgroup.
// IN4:   x is [N,S,S,4] NHWC4 (the fused warp kernel's output) instead of [N,3,S,S] CHW.
// CONV1: out = [N,D,D,64] = conv1(normalised pixel) (w1 = [64][3] OIHW 1x1); otherwise out = [N,3,D,D] CHW normalised.
template <bool IN4, bool CONV1>
__global__ __launch_bounds__(256) void k_pg_pre(const float* __restrict__ x, int N, int S, int D, float scale, PgNorm nm,
                                                const float* __restrict__ w1, float* __restrict__ out) {
    __shared__ float w1s[64 * 3];
    __shared__ float px[256 * 3];
    const int tid = threadIdx.x;
    const int per_img = (D * D + 255) / 256;
    const int n = blockIdx.x / per_img;
    const int p0 = (blockIdx.x % per_img) * 256;
    const int p = p0 + tid;
    if (CONV1 && tid < 192) w1s[tid] = w1[tid];
    float v3[3] = {0.f, 0.f, 0.f};
    if (p < D * D) {
        const int dy = p / D, dx = p % D;
        int iy[4], ix[4];
        float wy[4], wx[4];
        pg_taps(dy, S, scale, iy, wy);
        pg_taps(dx, S, scale, ix, wx);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float h = 0.f;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float s = IN4 ? x[(((size_t)n * S + iy[j]) * S + ix[i]) * 4 + c]
                                        : x[(((size_t)n * 3 + c) * S + iy[j]) * S + ix[i]];
                    const float v = (s + 0.5f) * 255.f;                       // :42
                    h = i == 0 ? v * wx[0] : h + v * wx[i];
                }
                acc = j == 0 ? h * wy[0] : acc + h * wy[j];
            }
            const float z = __fdiv_rn(acc, 255.f);                            // :43
            v3[c] = __fdiv_rn(z - nm.mean[c], nm.std[c]);                     // Normalize
        }
    }
    if (!CONV1) {
        if (p < D * D) {
#pragma unroll
            for (int c = 0; c < 3; c++) out[((size_t)n * 3 + c) * D * D + p] = v3[c];
        }
        return;
    }
    px[tid * 3 + 0] = v3[0]; px[tid * 3 + 1] = v3[1]; px[tid * 3 + 2] = v3[2];
    __syncthreads();
    // 256 pixels x 64 channels leave as 4096 16-byte stores, 16 consecutive lanes per pixel (256 contiguous bytes)
    float4* o4 = reinterpret_cast<float4*>(out + ((size_t)n * D * D + p0) * 64);
#pragma unroll 4
    for (int i = 0; i < 16; i++) {
        const int v = tid + i * 256;
        const int pp = v >> 4, c4 = (v & 15) * 4;
        if (p0 + pp < D * D) {
            const float a0 = px[pp * 3], a1 = px[pp * 3 + 1], a2 = px[pp * 3 + 2];
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float* wk = w1s + (c4 + k) * 3;
                float acc = 0.f;
                acc = fmaf(a0, wk[0], acc);
                acc = fmaf(a1, wk[1], acc);
                acc = fmaf(a2, wk[2], acc);
                r[k] = acc;
            }
            o4[v] = make_float4(r[0], r[1], r[2], r[3]);
        }
    }
}

static const PgNorm PG_NORM = {{0.4914f, 0.4822f, 0.4465f}, {0.2023f, 0.1994f, 0.2010f}};   // patchguard_predict.py:31

int pg_preprocess(gr_ctx* ctx, const float* x, int N, int S, int D, bool in_nhwc4, const float* conv1_w, float* out,
                  hipStream_t st) {
    if (N <= 0) return GR_OK;
    if (S < 1 || D < 1 || S > 4096 || D > 4096) return gr_fail(ctx, GR_EINVAL, "pg_preprocess: bad sizes %d -> %d", S, D);
    const long long blocks = (long long)N * ((D * D + 255) / 256);
    if (blocks > 0x7fffffffLL) return gr_fail(ctx, GR_EINVAL, "pg_preprocess: too many images");
    const float scale = (float)S / (float)D;
    // algorithmic bytes: the 32x32x3 input once, the output once
    ProfScope ps_(ctx, conv1_w ? "k_pg_pre<conv1>" : "k_pg_pre", st, conv1_w ? 2.0 * N * D * D * 64 * 3 : 0.0,
                  4.0 * N * (3.0 * S * S + (conv1_w ? 64.0 : 3.0) * D * D));
    dim3 g((unsigned)blocks), b(256);
    if (conv1_w) {
        if (in_nhwc4) hipLaunchKernelGGL((k_pg_pre<true, true>), g, b, 0, st, x, N, S, D, scale, PG_NORM, conv1_w, out);
        else hipLaunchKernelGGL((k_pg_pre<false, true>), g, b, 0, st, x, N, S, D, scale, PG_NORM, conv1_w, out);
    } else {
        if (in_nhwc4) hipLaunchKernelGGL((k_pg_pre<true, false>), g, b, 0, st, x, N, S, D, scale, PG_NORM, conv1_w, out);
        else hipLaunchKernelGGL((k_pg_pre<false, false>), g, b, 0, st, x, N, S, D, scale, PG_NORM, conv1_w, out);
    }
    GR_HIP(ctx, hipGetLastError());
    return GR_OK;
}

// ---- masking_defense: one workgroup per image ------------------------------------------------------------------
// u = max(local, 0) in LDS; per class c: total_c (sequential row-major sum), win_c = max over window positions of the
// window's sequential row-major sum; masked_c = win_c / total_c > thres ? total_c - win_c : total_c; pred = LAST maximum.
// Window sums are >= 0, so their float order is their int order: one LDS atomicMax per (window, class).
// STAGED: the clipped map of the image sits in LDS (bagnet17: 19 KB); maps too large for that are clipped on the fly from
// global memory (L2), same sums in the same order.
template <bool STAGED>
__global__ __launch_bounds__(256) void k_pg_masking(const float* __restrict__ local, int h, int w, int ncls, int wx, int wy,
                                                    float thres, float* __restrict__ scores, long long* __restrict__ preds) {
    extern __shared__ float smem_pg[];           // [STAGED: h*w*ncls] + totals[ncls] + winmax[ncls] (int bits)
    const int tid = threadIdx.x, n = blockIdx.x, npos = h * w;
    const float* src = local + (size_t)n * npos * ncls;
    float* tot = smem_pg + (STAGED ? npos * ncls : 0);
    int* wmax = reinterpret_cast<int*>(tot + ncls);
    auto u = [&](int i) { if (STAGED) return smem_pg[i]; const float v = src[i]; return v > 0.f ? v : 0.f; };
    if (STAGED)
        for (int i = tid; i < npos * ncls; i += 256) { float v = src[i]; smem_pg[i] = v > 0.f ? v : 0.f; }
    if (tid < ncls) wmax[tid] = 0;               // bits of +0.0f; every window sum is >= 0
    __syncthreads();
    for (int c = tid; c < ncls; c += 256) {
        float s = 0.f;
        for (int p = 0; p < npos; p++) s += u(p * ncls + c);
        tot[c] = s;
    }
    const int nwx = h - wx + 1, nwy = w - wy + 1;
    for (int i = tid; i < nwx * nwy * ncls; i += 256) {
        const int c = i % ncls, wi = i / ncls;
        const int x0 = wi / nwy, y0 = wi % nwy;
        float s = 0.f;
        for (int a = 0; a < wx; a++)
            for (int b = 0; b < wy; b++) s += u(((x0 + a) * w + y0 + b) * ncls + c);
        atomicMax(&wmax[c], __float_as_int(s));
    }
    __syncthreads();
    if (tid == 0) {                              // sequential on purpose: the last-maximum rule
        int best = 0;
        float bv = 0.f;
        for (int c = 0; c < ncls; c++) {
            const float total = tot[c], win = __int_as_float(wmax[c]);
            const float m = (__fdiv_rn(win, total) > thres) ? total - win : total;     // 0/0 = nan: not masked
            if (scores) scores[(size_t)n * ncls + c] = m;
            if (c == 0 || m >= bv) { bv = m; best = c; }
        }
        if (preds) preds[n] = best;
    }
}

int pg_masking(gr_ctx* ctx, const float* local, int N, int h, int w, int ncls, int wx, int wy, float thres, float* scores,
               int64_t* preds, hipStream_t st) {
    if (N <= 0) return GR_OK;
    if (h <= 0 || w <= 0 || ncls <= 0 || ncls > 4096 || wx <= 0 || wy <= 0 || wx > h || wy > w || (long long)h * w * ncls > 0x7fffffffLL)
        return gr_fail(ctx, GR_EINVAL, "masking_defense: bad shape %dx%dx%d, window %dx%d", h, w, ncls, wx, wy);
    const size_t staged = sizeof(float) * ((size_t)h * w * ncls + 2 * ncls);
    ProfScope ps_(ctx, "k_pg_masking", st, 0.0, 4.0 * N * ((double)h * w * ncls + ncls));
    if (staged <= 64 * 1024)
        hipLaunchKernelGGL(k_pg_masking<true>, dim3(N), dim3(256), staged, st, local, h, w, ncls, wx, wy, thres, scores, (long long*)preds);
    else
        hipLaunchKernelGGL(k_pg_masking<false>, dim3(N), dim3(256), sizeof(float) * 2 * ncls, st, local, h, w, ncls, wx, wy, thres, scores,
                           (long long*)preds);
    GR_HIP(ctx, hipGetLastError());
    return GR_OK;
}

// aggregation='mean': scores[n][c] = (sequential row-major sum over the positions) * (1 / (h*w))
__global__ void k_pg_mean(const float* __restrict__ local, int N, int npos, int ncls, float* __restrict__ scores) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * ncls) return;
    int n = idx / ncls, c = idx % ncls;
    const float* s = local + (size_t)n * npos * ncls + c;
    float sum = 0.f;
    for (int p = 0; p < npos; p++) sum += s[(size_t)p * ncls];
    scores[idx] = sum * (1.0f / (float)npos);
}

int pg_mean(gr_ctx* ctx, const float* local, int N, int h, int w, int ncls, float* scores, hipStream_t st) {
    if (N <= 0) return GR_OK;
    ProfScope ps_(ctx, "k_pg_mean", st, 0.0, 4.0 * N * ((double)h * w * ncls + ncls));
    hipLaunchKernelGGL(k_pg_mean, dim3((N * ncls + 255) / 256), dim3(256), 0, st, local, N, h * w, ncls, scores);
    GR_HIP(ctx, hipGetLastError());
    return GR_OK;
}
