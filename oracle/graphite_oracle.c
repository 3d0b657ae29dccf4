/*
 * graphite_oracle.c -- CPU restatement of GRAPHITE's transform-robustness scoring path.
 * TEST INFRASTRUCTURE ONLY (see graphite_oracle.h).  Build: oracle/Makefile
 *   gcc -O2 -ffp-contract=off -mavx2 -mfma -fopenmp -shared -fPIC
 * -ffp-contract=off matters: every a*b+c below is two roundings unless written fmaf().
 *
 * Reference lines followed (paths relative to /root/reference):
 *   transforms.py:143-159   add_noise
 *   transforms.py:50-77     apply_transformation (blur, composite, gamma LUT)
 *   transforms.py:18-29     convert2Network (resize 32x32, clamp, -0.5)
 *   perspective_transform.py:47  cv2.warpPerspective(img, M, (crop_x,crop_y), BORDER_REPLICATE)
 *   GTSRB/GTSRBNet.py:35-52 forward ; cifar/wideresnet.py:25-34, 86-94 forward
 *   utils.py:51-52          argmax == target count
 *   patchguard/patchguard_predict.py:42-44, 30-32   bicubic x6 + Normalize (torch semantics; checked against torch itself)
 *   patchguard/patchguard_predict.py:24, 48         BagNet-17 and masking_defense: THIRD-PARTY code (inspire-group/PatchGuard),
 *                           absent from the reference tree, restated from the publications -- parity unpinned
 * OpenCV 4.2 algorithms (resize / GaussianBlur / warpPerspective+remap / LUT) are restated
 * from the published implementation; marked [OpenCV-recalled, unpinned] where they are used.
 */
#include "graphite_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline int clipi(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; } /* cv::clip */

/* ------------------------------------------------------------------------------------------
 * cv::resize INTER_LINEAR, float  [OpenCV-recalled, unpinned]
 *   scale = 1/(dsize/ssize) in double; f = (float)((d+0.5)*scale-0.5); s = floor(f); f -= s
 *   horizontal: s<0 -> (s,f)=(0,0); s>=sw-1 -> (s,f)=(sw-1,0); taps S[s]*(1-f) + S[s+1]*f
 *   vertical:   rows clip(s), clip(s+1); weights (1-f, f) from the UNclamped f
 *   horizontal pass first (per source row), then vertical; equal sizes -> copy.
 *   EXACTLY 2x down in both directions: cv::resize switches INTER_LINEAR to the INTER_AREA fast path
 *   (resize.cpp: "in case of scale_x && scale_y is equal to 2 INTER_AREA (fast) also is equal to INTER_LINEAR");
 *   for 3-channel float images that is the scalar ResizeAreaFast loop: sum = ((S00 + S01) + S10) + S11, D = sum * 0.25f
 *   (the same taps as the bilinear path, whose weights are all 0.5, in a different summation order: <= 1 ulp apart).
 * ------------------------------------------------------------------------------------------ */
typedef struct { int i0, i1; float w0, w1; } tap_t;

static void resize_taps(int ssize, int dsize, int horizontal, tap_t* t) {
    double inv_scale = (double)dsize / (double)ssize;
    double scale = 1.0 / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (horizontal) {
            if (s < 0) { f = 0.f; s = 0; }
            if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
            t[d].i0 = s;
            t[d].i1 = s + 1 < ssize ? s + 1 : ssize - 1;
        } else {
            t[d].i0 = clipi(s, 0, ssize);
            t[d].i1 = clipi(s + 1, 0, ssize);
        }
        t[d].w0 = 1.f - f;
        t[d].w1 = f;
    }
}

static void resize_plane(const float* S, int sh, int sw, float* D, int dh, int dw, const tap_t* tx, const tap_t* ty) {
    if (sh == dh && sw == dw) { memcpy(D, S, sizeof(float) * (size_t)sh * sw); return; }
    if (sh == 2 * dh && sw == 2 * dw) {                     /* INTER_AREA fast path, scale 2 x 2 */
        for (int dy = 0; dy < dh; dy++) {
            const float* r0 = S + (size_t)(2 * dy) * sw;
            const float* r1 = r0 + sw;
            for (int dx = 0; dx < dw; dx++) {
                float sum = 0.f;
                sum += r0[2 * dx] + r0[2 * dx + 1] + r1[2 * dx] + r1[2 * dx + 1];
                D[(size_t)dy * dw + dx] = sum * 0.25f;
            }
        }
        return;
    }
    for (int dy = 0; dy < dh; dy++) {
        const float* r0 = S + (size_t)ty[dy].i0 * sw;
        const float* r1 = S + (size_t)ty[dy].i1 * sw;
        float b0 = ty[dy].w0, b1 = ty[dy].w1;
        for (int dx = 0; dx < dw; dx++) {
            float a0 = tx[dx].w0, a1 = tx[dx].w1;
            float h0 = r0[tx[dx].i0] * a0 + r0[tx[dx].i1] * a1;
            float h1 = r1[tx[dx].i0] * a0 + r1[tx[dx].i1] * a1;
            D[(size_t)dy * dw + dx] = h0 * b0 + h1 * b1;
        }
    }
}

int or_resize_linear(const float* src, int C, int sh, int sw, float* dst, int dh, int dw) {
    if (!src || !dst || C <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0) return -1;
    tap_t* tx = (tap_t*)malloc(sizeof(tap_t) * dw);
    tap_t* ty = (tap_t*)malloc(sizeof(tap_t) * dh);
    resize_taps(sw, dw, 1, tx);
    resize_taps(sh, dh, 0, ty);
    for (int c = 0; c < C; c++)
        resize_plane(src + (size_t)c * sh * sw, sh, sw, dst + (size_t)c * dh * dw, dh, dw, tx, ty);
    free(tx); free(ty);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * add_noise  (transforms.py:143-159)
 * ------------------------------------------------------------------------------------------ */
int or_add_noise(const float* img, int H, int W, const float* mask, const float* theta, int nh, int nw,
                 float lbd, int clip, float* comb, float* pert, float* mask_lg) {
    if (!img || !mask || !theta) return -1;
    size_t n = (size_t)3 * H * W;
    float* th = (float*)malloc(sizeof(float) * n);
    float* ml = (float*)malloc(sizeof(float) * n);
    or_resize_linear(theta, 3, nh, nw, th, H, W);          /* :148 */
    or_resize_linear(mask, 3, nh, nw, ml, H, W);           /* :151 */
    for (size_t i = 0; i < n; i++) {
        float c = img[i] + lbd * th[i];                    /* :149 */
        float m = ml[i] > 0.5f ? 1.f : 0.f;                /* :152 */
        if (clip) c = c < 0.f ? 0.f : (c > 1.f ? 1.f : c); /* :155 */
        if (comb) comb[i] = c;
        if (pert) pert[i] = m > 0.5f ? c : 0.f;            /* :157 */
        if (mask_lg) mask_lg[i] = m;
    }
    free(th); free(ml);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * cv::GaussianBlur(src,(k,k),0) float, BORDER_REFLECT_101  [OpenCV-recalled, unpinned]
 *   sigma<=0 and k<=7 -> fixed kernels; separable: row pass then column pass.
 *   k<=5 rows: centre*k0 + (l1+r1)*k1 [+ (l2+r2)*k2]; k=7 rows: left-to-right accumulate;
 *   columns: centre*k0 then += kj*(S[+j]+S[-j]).
 * ------------------------------------------------------------------------------------------ */
static const float GK3[2] = {0.5f, 0.25f};
static const float GK5[3] = {0.375f, 0.25f, 0.0625f};
static const float GK7[7] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};

static inline int refl101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; }
    return i;
}

static void blur_plane(const float* S, int H, int W, int k, float* D, float* tmp) {
    int r = k / 2;
    for (int y = 0; y < H; y++) {
        const float* s = S + (size_t)y * W;
        float* t = tmp + (size_t)y * W;
        for (int x = 0; x < W; x++) {
            float v;
            if (k == 3) {
                v = s[x] * GK3[0] + (s[refl101(x - 1, W)] + s[refl101(x + 1, W)]) * GK3[1];
            } else if (k == 5) {
                v = s[x] * GK5[0] + (s[refl101(x - 1, W)] + s[refl101(x + 1, W)]) * GK5[1]
                    + (s[refl101(x - 2, W)] + s[refl101(x + 2, W)]) * GK5[2];
            } else {
                v = GK7[0] * s[refl101(x - 3, W)];
                for (int j = 1; j < 7; j++) v += GK7[j] * s[refl101(x - 3 + j, W)];
            }
            t[x] = v;
        }
    }
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            float v;
            if (k == 3) {
                v = tmp[(size_t)y * W + x] * GK3[0]
                    + (tmp[(size_t)refl101(y - 1, H) * W + x] + tmp[(size_t)refl101(y + 1, H) * W + x]) * GK3[1];
            } else {
                const float* kk = (k == 5) ? GK5 : (GK7 + 3);
                v = kk[0] * tmp[(size_t)y * W + x];
                for (int j = 1; j <= r; j++)
                    v += kk[j] * (tmp[(size_t)refl101(y + j, H) * W + x] + tmp[(size_t)refl101(y - j, H) * W + x]);
            }
            D[(size_t)y * W + x] = v;
        }
    }
}

int or_composite(const float* img, const float* pert, const float* mask_lg, int H, int W, int blur, float* att) {
    if (!img || !pert || !mask_lg || !att) return -1;
    if (!(blur == 0 || blur == 3 || blur == 5 || blur == 7)) return -2;
    size_t n = (size_t)3 * H * W, pl = (size_t)H * W;
    const float* p = pert;
    float* pb = NULL;
    if (blur) {                                             /* transforms.py:55-56 */
        pb = (float*)malloc(sizeof(float) * n);
        float* tmp = (float*)malloc(sizeof(float) * pl);
        for (int c = 0; c < 3; c++) blur_plane(pert + c * pl, H, W, blur, pb + c * pl, tmp);
        free(tmp);
        p = pb;
    }
    for (size_t i = 0; i < n; i++) {                        /* :60-62 */
        float v = mask_lg[i] > 0.5f ? p[i] : img[i];
        att[i] = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
    }
    free(pb);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * cv::warpPerspective INTER_LINEAR BORDER_REPLICATE, float  [OpenCV-recalled, unpinned]
 *   M inverted in double by the closed-form 3x3 adjugate (cv::invert, n==3);
 *   per dst pixel (x = xb + x1 with xb the start of its 64-wide block):
 *     W = (M6*xb + M7*y + M8) + M6*x1;  W = 32*(W ? 1/W : 0)
 *     X = cvRound(clamp((M0*xb+M1*y+M2 + M0*x1)*W)), same for Y   (round-half-even)
 *     sx = X>>5, sy = Y>>5, weights from (X&31)/32, (Y&31)/32 as float products
 *     D = v00*w0 + v01*w1 + v10*w2 + v11*w3, taps clamped to the image (replicate).
 * ------------------------------------------------------------------------------------------ */
static void invert3x3(const double* S, double* t) {
    double det = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6])
                 + S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (det == 0.0) { memset(t, 0, 9 * sizeof(double)); return; }
    double d = 1.0 / det;
    t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
    t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
    t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
    t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
    t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
    t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
    t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
    t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
    t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
}

static inline int cv_round_sat(double v) {
    if (v < (double)INT_MIN) v = (double)INT_MIN;
    if (v > (double)INT_MAX) v = (double)INT_MAX;
    return (int)nearbyint(v);   /* default rounding mode: nearest-even, as cvRound */
}
static inline int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

/* block width of WarpPerspectiveInvoker: BLOCK_SZ=32 */
static int warp_block_w(int width, int height) {
    int bh0 = 16 < height ? 16 : height;
    int bw0 = 1024 / bh0; if (bw0 > width) bw0 = width;
    return bw0;
}

/* one warped pixel, all 3 channels */
static inline void warp_pixel(const float* att, int H, int W, const double* Mi, int bw0, int x, int y, float* o3) {
    int xb = (x / bw0) * bw0, x1 = x - xb;
    double X0 = Mi[0] * xb + Mi[1] * y + Mi[2];
    double Y0 = Mi[3] * xb + Mi[4] * y + Mi[5];
    double W0 = Mi[6] * xb + Mi[7] * y + Mi[8];
    double Wd = W0 + Mi[6] * x1;
    Wd = 32.0 * (Wd != 0.0 ? 1.0 / Wd : 0.0);
    int X = cv_round_sat((X0 + Mi[0] * x1) * Wd);
    int Y = cv_round_sat((Y0 + Mi[3] * x1) * Wd);
    int sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
    float fx = (float)(X & 31) * (1.f / 32.f), fy = (float)(Y & 31) * (1.f / 32.f);
    float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
    int sx0 = clipi(sx, 0, W), sx1 = clipi(sx + 1, 0, W), sy0 = clipi(sy, 0, H), sy1 = clipi(sy + 1, 0, H);
    size_t pl = (size_t)H * W;
    for (int c = 0; c < 3; c++) {
        const float* p = att + c * pl;
        o3[c] = p[(size_t)sy0 * W + sx0] * w0 + p[(size_t)sy0 * W + sx1] * w1 + p[(size_t)sy1 * W + sx0] * w2
                + p[(size_t)sy1 * W + sx1] * w3;
    }
}

/* Margin probe (tests/pixel_margin.py): or_set_warp_nudge(n) moves every warped value n ulps (n = +-1) before
 * the uint8 truncation below -- the one discontinuous step of the pixel path -- to measure how far a last-ulp difference
 * between this restatement and real OpenCV could move pixels, labels and hit counts.  0 (the default) = the oracle proper. */
static int g_warp_nudge = 0;
void or_set_warp_nudge(int n) { g_warp_nudge = n; }
int or_contracted_build(void) {
#ifdef OR_FMA_CONTRACTED
    return 1;
#else
    return 0;
#endif
}

/* transforms.py:69-75: u8 = trunc(v*255); lut; (u8/255.0 as double) -> float; clip */
static inline float gamma_stage(float v, const uint8_t* lut) {
    for (int n = g_warp_nudge; n > 0; n--) v = nextafterf(v, INFINITY);
    for (int n = g_warp_nudge; n < 0; n++) v = nextafterf(v, -INFINITY);
    if (v < 0.f) v = 0.f;                /* a nudge below 0 would wrap in the byte cast; OpenCV inputs are clipped to [0,1] */
    float s = v * 255.0f;
    int u = (int)s;                      /* astype(uint8) of a value in [0,256) */
    u &= 255;
    float r = (float)((double)lut[u] / 255.0);
    return r < 0.f ? 0.f : (r > 1.f ? 1.f : r);
}

/* convert2Network's output size.  net_size == 32: 32 x 32 (transforms.py:19-23).  Otherwise (transforms.py:35-40) the
 * short side becomes net_size and the long side int(round(net_size / short * long)): Python float arithmetic, round()
 * = half to even = nearbyint in the default rounding mode. */
void or_net_size(int crop_x, int crop_y, int net_size, int* nx, int* ny) {
    if (net_size == 32) { *nx = *ny = 32; return; }
    if (crop_y > crop_x) { *nx = net_size; *ny = (int)nearbyint((double)net_size / (double)crop_x * (double)crop_y); }
    else { *ny = net_size; *nx = (int)nearbyint((double)net_size / (double)crop_y * (double)crop_x); }
}

/* warp + gamma + resize to ny x nx + clamp (+ -0.5) */
static int warp_one_sized(const float* att, int H, int W, const or_xform* xf, int normalize, int nx, int ny, float* out,
                          float* warped) {
    if (!att || !xf || !out || xf->crop_x <= 0 || xf->crop_y <= 0) return -1;
    int cw = xf->crop_x, chh = xf->crop_y;
    double Mi[9];
    invert3x3(xf->M, Mi);
    int bw0 = warp_block_w(cw, chh);
    size_t cpl = (size_t)cw * chh;
    float* g = (float*)malloc(sizeof(float) * 3 * cpl);   /* gamma'd warp, planar */
    for (int y = 0; y < chh; y++)
        for (int x = 0; x < cw; x++) {
            float o3[3];
            warp_pixel(att, H, W, Mi, bw0, x, y, o3);
            for (int c = 0; c < 3; c++) {
                if (warped) warped[c * cpl + (size_t)y * cw + x] = o3[c];
                g[c * cpl + (size_t)y * cw + x] = gamma_stage(o3[c], xf->lut);
            }
        }
    or_resize_linear(g, 3, chh, cw, out, ny, nx);          /* transforms.py:23 / :42 */
    for (int i = 0; i < 3 * nx * ny; i++) {
        float v = out[i];
        v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);            /* :25 / :44 */
        out[i] = normalize ? (v - 0.5f) : v;                /* :28 (net_size 32, GTSRB only) */
    }
    free(g);
    return 0;
}

int or_warp_one(const float* att, int H, int W, const or_xform* xf, int normalize, int net, float* out,
                float* warped) {
    return warp_one_sized(att, H, W, xf, normalize, net, net, out, warped);
}

/* The net_size != 32 branch (transforms.py:31-45): out = [3, ny, nx] with (nx, ny) = or_net_size(crop), clamp only. */
int or_warp_one_var(const float* att, int H, int W, const or_xform* xf, int net_size, float* out) {
    int nx, ny;
    if (!xf || net_size == 32) return -1;
    or_net_size(xf->crop_x, xf->crop_y, net_size, &nx, &ny);
    return warp_one_sized(att, H, W, xf, 0, nx, ny, out, NULL);
}

int or_warp_batch(const float* img, int H, int W, const float* masks, const float* thetas, int C, int nh, int nw,
                  const or_xform* xf, int T, float lbd, int normalize, int net, float* out, int nthreads) {
    if (!img || !masks || !thetas || !xf || !out || C < 0 || T < 0) return -1;
    size_t n = (size_t)3 * H * W, small = (size_t)3 * nh * nw, on = (size_t)3 * net * net;
    int used[8] = {0};
    for (int t = 0; t < T; t++) {
        int b = xf[t].blur;
        if (!(b == 0 || b == 3 || b == 5 || b == 7)) return -2;
        used[b] = 1;
    }
    if (nthreads < 1) nthreads = 1;
    for (int c = 0; c < C; c++) {
        float* pert = (float*)malloc(sizeof(float) * n);
        float* ml = (float*)malloc(sizeof(float) * n);
        float* att[8] = {0};
        or_add_noise(img, H, W, masks + c * small, thetas + c * small, nh, nw, lbd, 1, NULL, pert, ml);
        for (int b = 0; b < 8; b++)
            if (used[b]) {
                att[b] = (float*)malloc(sizeof(float) * n);
                or_composite(img, pert, ml, H, W, b, att[b]);
            }
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
        for (int t = 0; t < T; t++)
            or_warp_one(att[xf[t].blur], H, W, &xf[t], normalize, net, out + ((size_t)c * T + t) * on, NULL);
        for (int b = 0; b < 8; b++) free(att[b]);
        free(pert); free(ml);
    }
    return 0;
}

/* get_transformed_images with net_size != 32: image (c, t) at out + (c*T + t) * out_stride as [3, ny_t, nx_t] */
int or_warp_batch_var(const float* img, int H, int W, const float* masks, const float* thetas, int C, int nh, int nw,
                      const or_xform* xf, int T, float lbd, int net_size, float* out, long long out_stride, int nthreads) {
    if (!img || !masks || !thetas || !xf || !out || C < 0 || T < 0 || net_size == 32) return -1;
    size_t n = (size_t)3 * H * W, small = (size_t)3 * nh * nw;
    int used[8] = {0};
    for (int t = 0; t < T; t++) {
        int b = xf[t].blur, nx, ny;
        if (!(b == 0 || b == 3 || b == 5 || b == 7)) return -2;
        used[b] = 1;
        or_net_size(xf[t].crop_x, xf[t].crop_y, net_size, &nx, &ny);
        if ((long long)3 * nx * ny > out_stride) return -3;
    }
    if (nthreads < 1) nthreads = 1;
    for (int c = 0; c < C; c++) {
        float* pert = (float*)malloc(sizeof(float) * n);
        float* ml = (float*)malloc(sizeof(float) * n);
        float* att[8] = {0};
        or_add_noise(img, H, W, masks + c * small, thetas + c * small, nh, nw, lbd, 1, NULL, pert, ml);
        for (int b = 0; b < 8; b++)
            if (used[b]) {
                att[b] = (float*)malloc(sizeof(float) * n);
                or_composite(img, pert, ml, H, W, b, att[b]);
            }
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
        for (int t = 0; t < T; t++)
            or_warp_one_var(att[xf[t].blur], H, W, &xf[t], net_size, out + ((size_t)c * T + t) * (size_t)out_stride);
        for (int b = 0; b < 8; b++) free(att[b]);
        free(pert); free(ml);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Classifier layers.  Activations NHWC; weights repacked [k=(ky,kx,ci)][cout].
 * Every output is ONE k-ascending fmaf chain from 0, then + bias: the summation order the
 * reference leaves to cuDNN/MKL is fixed here (and in the HIP kernels, whose f32 MFMA is a
 * k-ordered fmaf chain) so that logits can be compared bit for bit.  Against the reference's
 * own PyTorch forward the pinned bound is 1e-4 relative (tests/golden/nets.npz).
 * ------------------------------------------------------------------------------------------ */
static float* repack_oihw(const float* w, int Co, int Ci, int KH, int KW) {
    float* p = (float*)malloc(sizeof(float) * (size_t)Co * Ci * KH * KW);
    for (int o = 0; o < Co; o++)
        for (int i = 0; i < Ci; i++)
            for (int y = 0; y < KH; y++)
                for (int x = 0; x < KW; x++)
                    p[((size_t)(y * KW + x) * Ci + i) * Co + o] = w[(((size_t)o * Ci + i) * KH + y) * KW + x];
    return p;
}

/* out[(oy,ox),co] = chain_k in[(oy*s+ky-pad, ox*s+kx-pad), ci] * wp[k][co]   (zero padding taps skipped:
 * fmaf(0,w,acc)==acc, so skipping is bit-identical).
 * Order of the chain: k = (ky,kx,ci) ascending -- except for stride-1 3x3 convs whose Ci is a multiple of 32
 * (WideResNet's block convs), which are walked in channel chunks of 32, k = (ci/32, ky, kx, ci%32): the order
 * of the HIP kernel that keeps a 32-channel slice of the input tile in LDS (csrc/conv_halo_cc.h).  The reference
 * leaves the order to cuDNN; either order is within 2e-6 relative of its PyTorch forward (tests/golden/nets.npz). */
static void conv_nhwc(const float* in, int H, int W, int Ci, const float* wp, int KH, int KW, int stride, int pad,
                      float* out, int Ho, int Wo, int Co) {
    float acc[4096];
    const int cc = (KH == 3 && KW == 3 && stride == 1 && Ci % 32 == 0) ? 32 : Ci;
    for (int oy = 0; oy < Ho; oy++)
        for (int ox = 0; ox < Wo; ox++) {
            for (int o = 0; o < Co; o++) acc[o] = 0.f;
            for (int c0 = 0; c0 < Ci; c0 += cc)
                for (int ky = 0; ky < KH; ky++) {
                    int iy = oy * stride + ky - pad;
                    if (iy < 0 || iy >= H) continue;
                    for (int kx = 0; kx < KW; kx++) {
                        int ix = ox * stride + kx - pad;
                        if (ix < 0 || ix >= W) continue;
                        const float* a = in + ((size_t)iy * W + ix) * Ci;
                        const float* wk = wp + (size_t)(ky * KW + kx) * Ci * Co;
                        for (int i = c0; i < c0 + cc; i++) {
                            float av = a[i];
                            const float* wr = wk + (size_t)i * Co;
                            for (int o = 0; o < Co; o++) acc[o] = fmaf(av, wr[o], acc[o]);
                        }
                    }
                }
            float* d = out + ((size_t)oy * Wo + ox) * Co;
            for (int o = 0; o < Co; o++) d[o] = acc[o];
        }
}

static void bias_relu(float* x, size_t npix, int C, const float* b, int relu) {
    for (size_t p = 0; p < npix; p++)
        for (int c = 0; c < C; c++) {
            float v = x[p * C + c] + (b ? b[c] : 0.f);
            x[p * C + c] = (relu && !(v > 0.f)) ? 0.f : v;
        }
}

static void maxpool2(const float* in, int H, int W, int C, float* out) {
    for (int y = 0; y < H / 2; y++)
        for (int x = 0; x < W / 2; x++)
            for (int c = 0; c < C; c++) {
                float a = in[((size_t)(2 * y) * W + 2 * x) * C + c], b = in[((size_t)(2 * y) * W + 2 * x + 1) * C + c];
                float d = in[((size_t)(2 * y + 1) * W + 2 * x) * C + c], e = in[((size_t)(2 * y + 1) * W + 2 * x + 1) * C + c];
                float m = a > b ? a : b; m = m > d ? m : d; m = m > e ? m : e;
                out[((size_t)y * (W / 2) + x) * C + c] = m;
            }
}

int or_gtsrb_forward(const float* const* t, const float* x, int N, float* logits, int nthreads) {
    if (!t || !x || !logits || N < 0) return -1;
    /* repack once */
    float* w21 = repack_oihw(t[2], 32, 3, 5, 5);
    float* w22 = repack_oihw(t[4], 32, 32, 5, 5);
    float* w31 = repack_oihw(t[6], 64, 32, 5, 5);
    float* w32 = repack_oihw(t[8], 64, 64, 5, 5);
    float* w41 = repack_oihw(t[10], 128, 64, 5, 5);
    float* w42 = repack_oihw(t[12], 128, 128, 5, 5);
    /* fc5: reference flattens NCHW (k = c*16 + y*4 + x, GTSRBNet.py:47); activations here are NHWC,
     * so the columns are permuted to k' = (y*4+x)*128 + c and the chain runs over k' ascending. */
    float* w5 = (float*)malloc(sizeof(float) * 2048 * 1024);
    for (int o = 0; o < 1024; o++)
        for (int c = 0; c < 128; c++)
            for (int s = 0; s < 16; s++) w5[((size_t)s * 128 + c) * 1024 + o] = t[14][(size_t)o * 2048 + c * 16 + s];
    float* w6 = repack_oihw(t[16], 1024, 1024, 1, 1);
    float* w7 = repack_oihw(t[18], 43, 1024, 1, 1);
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        float* a0 = (float*)malloc(sizeof(float) * 32 * 32 * 3);
        float* a1 = (float*)malloc(sizeof(float) * 32 * 32 * 32);
        float* a2 = (float*)malloc(sizeof(float) * 32 * 32 * 32);
#pragma omp for schedule(dynamic, 1)
        for (int n = 0; n < N; n++) {
            const float* xi = x + (size_t)n * 3 * 1024;
            /* conv1_1 1x1 3->3 + relu (GTSRBNet.py:36) */
            for (int p = 0; p < 1024; p++)
                for (int o = 0; o < 3; o++) {
                    float acc = 0.f;
                    for (int i = 0; i < 3; i++) acc = fmaf(xi[i * 1024 + p], t[0][o * 3 + i], acc);
                    acc += t[1][o];
                    a0[p * 3 + o] = acc > 0.f ? acc : 0.f;
                }
            conv_nhwc(a0, 32, 32, 3, w21, 5, 5, 1, 2, a1, 32, 32, 32); bias_relu(a1, 1024, 32, t[3], 1);   /* :38 */
            conv_nhwc(a1, 32, 32, 32, w22, 5, 5, 1, 2, a2, 32, 32, 32); bias_relu(a2, 1024, 32, t[5], 1);  /* :39 */
            maxpool2(a2, 32, 32, 32, a1);
            conv_nhwc(a1, 16, 16, 32, w31, 5, 5, 1, 2, a2, 16, 16, 64); bias_relu(a2, 256, 64, t[7], 1);   /* :41 */
            conv_nhwc(a2, 16, 16, 64, w32, 5, 5, 1, 2, a1, 16, 16, 64); bias_relu(a1, 256, 64, t[9], 1);   /* :42 */
            maxpool2(a1, 16, 16, 64, a2);
            conv_nhwc(a2, 8, 8, 64, w41, 5, 5, 1, 2, a1, 8, 8, 128); bias_relu(a1, 64, 128, t[11], 1);     /* :44 */
            conv_nhwc(a1, 8, 8, 128, w42, 5, 5, 1, 2, a2, 8, 8, 128); bias_relu(a2, 64, 128, t[13], 1);    /* :45 */
            maxpool2(a2, 8, 8, 128, a1);                              /* a1 = [4,4,128] NHWC = 2048 */
            conv_nhwc(a1, 1, 1, 2048, w5, 1, 1, 1, 0, a2, 1, 1, 1024); bias_relu(a2, 1, 1024, t[15], 1);   /* :49 */
            conv_nhwc(a2, 1, 1, 1024, w6, 1, 1, 1, 0, a1, 1, 1, 1024); bias_relu(a1, 1, 1024, t[17], 1);   /* :50 */
            conv_nhwc(a1, 1, 1, 1024, w7, 1, 1, 1, 0, a2, 1, 1, 43); bias_relu(a2, 1, 43, t[19], 0);       /* :51 */
            memcpy(logits + (size_t)n * 43, a2, sizeof(float) * 43);
        }
        free(a0); free(a1); free(a2);
    }
    free(w21); free(w22); free(w31); free(w32); free(w41); free(w42); free(w5); free(w6); free(w7);
    return 0;
}

/* ---- WideResNet ---------------------------------------------------------------------------- */
typedef struct { float* scale; float* shift; int C; } bn_t;

/* eval BatchNorm folded as PyTorch's CPU kernel does: alpha = w/sqrt(var+eps), beta = b - mean*alpha */
static bn_t bn_fold(const float* w, const float* b, const float* m, const float* v, int C) {
    bn_t r; r.C = C;
    r.scale = (float*)malloc(sizeof(float) * C);
    r.shift = (float*)malloc(sizeof(float) * C);
    for (int c = 0; c < C; c++) {
        float invstd = 1.0f / sqrtf(v[c] + 1e-5f);
        r.scale[c] = w[c] * invstd;
        r.shift[c] = b[c] - m[c] * r.scale[c];
    }
    return r;
}
/* mask (optional): 1 where the unit is active (v > 0), NHWC order -- the ReLU DECISIONS of this forward pass */
static void bn_relu_m(const float* in, size_t npix, const bn_t* bn, float* out, uint8_t* mask) {
    for (size_t p = 0; p < npix; p++)
        for (int c = 0; c < bn->C; c++) {
            float v = fmaf(in[p * bn->C + c], bn->scale[c], bn->shift[c]);
            out[p * bn->C + c] = v > 0.f ? v : 0.f;
            if (mask) mask[p * bn->C + c] = v > 0.f ? 1 : 0;
        }
}
static void bn_relu(const float* in, size_t npix, const bn_t* bn, float* out) { bn_relu_m(in, npix, bn, out, NULL); }

typedef struct {
    int cin, cout, stride;
    bn_t bn1, bn2;
    float *w1, *w2, *wsc;
} wrn_block_t;

/* bytes of ReLU decisions per image that or_wrn_forward_masks writes: per block relu(bn1(x)) [H,H,cin] then relu(bn2(conv1))
 * [Ho,Ho,cout], then the final relu(bn1) [8,8,ch3]; each NHWC, one byte per unit */
long long or_wrn_mask_bytes(int depth, int widen) {
    if ((depth - 4) % 6 != 0) return -1;
    int n = (depth - 4) / 6, ch[4] = {16, 16 * widen, 32 * widen, 64 * widen}, H = 32;
    long long tot = 0;
    for (int g = 0; g < 3; g++)
        for (int i = 0; i < n; i++) {
            int cin = i == 0 ? ch[g] : ch[g + 1], stride = (i == 0 && g > 0) ? 2 : 1, Ho = H / stride;
            tot += (long long)H * H * cin + (long long)Ho * Ho * ch[g + 1];
            H = Ho;
        }
    return tot + (long long)H * H * ch[3];
}

static int wrn_forward_impl(const float* const* t, int nt, int depth, int widen, int ncls, const float* x, int N,
                            float* logits, int nthreads, uint8_t* masks);
int or_wrn_forward(const float* const* t, int nt, int depth, int widen, int ncls, const float* x, int N,
                   float* logits, int nthreads) {
    return wrn_forward_impl(t, nt, depth, widen, ncls, x, N, logits, nthreads, NULL);
}
/* the same forward, also writing every ReLU decision (or_wrn_mask_bytes per image): the HIP library's forward is this
 * forward bit for bit, so these are ITS decisions -- what a gradient reference has to share to be compared tightly */
int or_wrn_forward_masks(const float* const* t, int nt, int depth, int widen, int ncls, const float* x, int N,
                         float* logits, int nthreads, uint8_t* masks) {
    return wrn_forward_impl(t, nt, depth, widen, ncls, x, N, logits, nthreads, masks);
}
static int wrn_forward_impl(const float* const* t, int nt, int depth, int widen, int ncls, const float* x, int N,
                            float* logits, int nthreads, uint8_t* masks) {
    if (!t || !x || !logits || (depth - 4) % 6 != 0) return -1;
    const long long mbytes = or_wrn_mask_bytes(depth, widen);
    int n = (depth - 4) / 6;
    int ch[4] = {16, 16 * widen, 32 * widen, 64 * widen};
    if (ch[3] > 1024) return -2;
    int nb = 3 * n, ti = 0;
    float* wconv1 = repack_oihw(t[ti++], 16, 3, 3, 3);
    wrn_block_t* blk = (wrn_block_t*)calloc(nb, sizeof(wrn_block_t));
    for (int g = 0; g < 3; g++)
        for (int i = 0; i < n; i++) {
            wrn_block_t* b = &blk[g * n + i];
            b->cin = i == 0 ? ch[g] : ch[g + 1];
            b->cout = ch[g + 1];
            b->stride = (i == 0 && g > 0) ? 2 : 1;
            b->bn1 = bn_fold(t[ti], t[ti + 1], t[ti + 2], t[ti + 3], b->cin); ti += 4;
            b->w1 = repack_oihw(t[ti++], b->cout, b->cin, 3, 3);
            b->bn2 = bn_fold(t[ti], t[ti + 1], t[ti + 2], t[ti + 3], b->cout); ti += 4;
            b->w2 = repack_oihw(t[ti++], b->cout, b->cout, 3, 3);
            b->wsc = (b->cin != b->cout) ? repack_oihw(t[ti++], b->cout, b->cin, 1, 1) : NULL;
        }
    bn_t bnf = bn_fold(t[ti], t[ti + 1], t[ti + 2], t[ti + 3], ch[3]); ti += 4;
    const float* fcw = t[ti++];
    const float* fcb = t[ti++];
    if (ti != nt) return -3;
    float* wfc = repack_oihw(fcw, ncls, ch[3], 1, 1);
    size_t maxact = (size_t)32 * 32 * ch[1];
    if ((size_t)16 * 16 * ch[2] > maxact) maxact = (size_t)16 * 16 * ch[2];
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        float* xa = (float*)malloc(sizeof(float) * maxact);   /* block input x */
        float* o = (float*)malloc(sizeof(float) * maxact);    /* relu(bn1(x)) */
        float* u = (float*)malloc(sizeof(float) * maxact);    /* conv1 out -> relu(bn2) */
        float* v = (float*)malloc(sizeof(float) * maxact);    /* conv2 out */
        float* s = (float*)malloc(sizeof(float) * maxact);    /* shortcut */
#pragma omp for schedule(dynamic, 1)
        for (int im = 0; im < N; im++) {
            const float* xi = x + (size_t)im * 3 * 1024;
            for (int p = 0; p < 1024; p++)
                for (int c = 0; c < 3; c++) o[p * 3 + c] = xi[c * 1024 + p];
            int H = 32;
            uint8_t* mk = masks ? masks + (size_t)im * mbytes : NULL;
            conv_nhwc(o, 32, 32, 3, wconv1, 3, 3, 1, 1, xa, 32, 32, 16);                  /* wideresnet.py:87 */
            for (int bi = 0; bi < nb; bi++) {
                wrn_block_t* b = &blk[bi];
                int Ho = H / b->stride;
                bn_relu_m(xa, (size_t)H * H, &b->bn1, o, mk);                               /* :27/:29 */
                if (mk) mk += (size_t)H * H * b->cin;
                conv_nhwc(o, H, H, b->cin, b->w1, 3, 3, b->stride, 1, u, Ho, Ho, b->cout);  /* :30 */
                bn_relu_m(u, (size_t)Ho * Ho, &b->bn2, u, mk);
                if (mk) mk += (size_t)Ho * Ho * b->cout;
                conv_nhwc(u, Ho, Ho, b->cout, b->w2, 3, 3, 1, 1, v, Ho, Ho, b->cout);       /* :33 */
                const float* sc = xa;
                if (b->wsc) { conv_nhwc(o, H, H, b->cin, b->wsc, 1, 1, b->stride, 0, s, Ho, Ho, b->cout); sc = s; }
                size_t ne = (size_t)Ho * Ho * b->cout;
                for (size_t e = 0; e < ne; e++) xa[e] = sc[e] + v[e];                       /* :34 */
                H = Ho;
            }
            bn_relu_m(xa, (size_t)H * H, &bnf, o, mk);                                      /* :91 */
            int C = ch[3];
            for (int c = 0; c < C; c++) {                                                   /* :92 avg_pool2d(8) */
                float sum = 0.f;
                for (int p = 0; p < H * H; p++) sum += o[(size_t)p * C + c];
                u[c] = sum * (1.0f / (float)(H * H));
            }
            conv_nhwc(u, 1, 1, C, wfc, 1, 1, 1, 0, v, 1, 1, ncls);                          /* :94 */
            for (int c = 0; c < ncls; c++) logits[(size_t)im * ncls + c] = v[c] + fcb[c];
        }
        free(xa); free(o); free(u); free(v); free(s);
    }
    free(wconv1); free(wfc);
    for (int i = 0; i < nb; i++) {
        free(blk[i].bn1.scale); free(blk[i].bn1.shift); free(blk[i].bn2.scale); free(blk[i].bn2.shift);
        free(blk[i].w1); free(blk[i].w2); free(blk[i].wsc);
    }
    free(bnf.scale); free(bnf.shift); free(blk);
    return 0;
}

/* utils.py:51-52 with torch.max first-maximum tie rule (GTSRBNet.py:66) */
int or_argmax_hits(const float* logits, int N, int ncls, int T, int target, int64_t* preds, int32_t* hits) {
    if (!logits || N < 0 || ncls <= 0 || T <= 0) return -1;
    if (hits) for (int c = 0; c < (N + T - 1) / T; c++) hits[c] = 0;
    for (int n = 0; n < N; n++) {
        const float* l = logits + (size_t)n * ncls;
        int best = 0;
        for (int c = 1; c < ncls; c++) if (l[c] > l[best]) best = c;
        if (preds) preds[n] = best;
        if (hits && best == target) hits[n / T]++;
    }
    return 0;
}

/* ==========================================================================================
 * PatchGuard robust classifier (reference patchguard/patchguard_predict.py:21-53).
 *
 *   z = (x + 0.5) * 255                                  :42     reference code
 *   z = F.interpolate(z, 192, mode='bicubic') / 255      :43     reference code (torch's bicubic, A = -0.75, half-pixel
 *                                                                centres, border taps clamped)
 *   z = Normalize(mean, std)(z)                          :30-32,44  reference code
 *   local = bagnet17(aggregation='none')(z)              :24,44  THIRD PARTY (inspire-group/PatchGuard @ ae23629, nets/bagnet.py
 *                                                                = Brendel & Bethge's BagNet with the fc applied per position);
 *                                                                source and weights absent: restated from the published
 *                                                                architecture, "parity unpinned"
 *   pred = masking_defense(local[i], thres=0.0, window_shape=[6,6])   :48  THIRD PARTY (utils/defense_utils.py): restated
 *                                                                from PatchGuard (Xiang et al., USENIX Security 2021)
 *                                                                Algorithm 1 "robust masking", "parity unpinned"
 * ========================================================================================== */

/* torch upsample_bicubic2d (ATen UpSample.h): cubic_convolution1/2 and get_cubic_upsample_coefficients, A = -0.75 */
static inline float pg_cc1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
static inline float pg_cc2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
typedef struct { int i[4]; float w[4]; } pg_tap_t;
static void pg_taps(int ssize, int dsize, pg_tap_t* t) {
    const float A = -0.75f;
    const float scale = (float)ssize / (float)dsize;          /* area_pixel_compute_scale, align_corners=False, no scale_factor */
    for (int d = 0; d < dsize; d++) {
        float real = scale * ((float)d + 0.5f) - 0.5f;        /* area_pixel_compute_source_index(cubic=true): not clamped at 0 */
        float fl = floorf(real);
        int in = (int)fl;
        float lam = real - fl;
        lam = lam < 0.f ? 0.f : (lam > 1.f ? 1.f : lam);      /* guard_index_and_lambda */
        t[d].w[0] = pg_cc2(lam + 1.f, A);
        t[d].w[1] = pg_cc1(lam, A);
        float l2 = 1.f - lam;
        t[d].w[2] = pg_cc1(l2, A);
        t[d].w[3] = pg_cc2(l2 + 1.f, A);
        for (int j = 0; j < 4; j++) {
            int q = in - 1 + j;
            t[d].i[j] = q < 0 ? 0 : (q > ssize - 1 ? ssize - 1 : q);          /* upsample_get_value_bounded */
        }
    }
}

static const float PG_MEAN[3] = {0.4914f, 0.4822f, 0.4465f};   /* patchguard_predict.py:31 */
static const float PG_STD[3] = {0.2023f, 0.1994f, 0.2010f};

/* one output value: rows combined left to right, then the four row results top to bottom (cubic_interp1d order) */
static inline float pg_sample(const float* plane, int S, const pg_tap_t* ty, const pg_tap_t* tx) {
    float acc = 0.f;
    for (int j = 0; j < 4; j++) {
        const float* r = plane + (size_t)ty->i[j] * S;
        float h = 0.f;
        for (int i = 0; i < 4; i++) {
            float v = (r[tx->i[i]] + 0.5f) * 255.f;           /* :42 */
            h = i == 0 ? v * tx->w[0] : h + v * tx->w[i];
        }
        acc = j == 0 ? h * ty->w[0] : acc + h * ty->w[j];
    }
    return acc;
}

/* x [N,3,S,S] in [-0.5,0.5] -> out [N,3,D,D]: lines 42-44 + the Normalize of :30-32 */
int or_pg_preprocess(const float* x, int N, int S, int D, float* out, int nthreads) {
    if (!x || !out || N < 0 || S <= 0 || D <= 0) return -1;
    pg_tap_t* t = (pg_tap_t*)malloc(sizeof(pg_tap_t) * D);
    pg_taps(S, D, t);
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int nc = 0; nc < N * 3; nc++) {
        const int c = nc % 3;
        const float* plane = x + (size_t)nc * S * S;
        float* o = out + (size_t)nc * D * D;
        for (int dy = 0; dy < D; dy++)
            for (int dx = 0; dx < D; dx++) {
                float v = pg_sample(plane, S, &t[dy], &t[dx]) / 255.f;        /* :43 */
                o[(size_t)dy * D + dx] = (v - PG_MEAN[c]) / PG_STD[c];        /* Normalize: sub_(mean).div_(std) */
            }
    }
    free(t);
    return 0;
}

/* ---- BagNet (nets/bagnet.py of the PatchGuard repository = bagnets/pytorchnet.py, restated) -----------------------
 * conv1 1x1 3->64, conv2 3x3 64->64 (no padding), bn1, relu; four stages of Bottleneck blocks (planes 64/128/256/512,
 * expansion 4): conv1 1x1 -> bn -> relu -> conv2 kxk stride s NO padding -> bn -> relu -> conv3 1x1 -> bn; the
 * residual (1x1 stride-s conv + bn "downsample" in the first block of a stage, identity otherwise) is cropped at the
 * bottom/right to the output's size ([:, :, :-diff, :-diff]) before the add; relu.  k = 3 in block i of stage g iff
 * kernel3[g] > i; the stage stride sits on block 0's conv2.  bagnet17: layers [3,4,6,3], strides [2,2,2,1], kernel3
 * [1,1,1,0]: receptive field 17, stride 8 (patchguard_predict.py:33-34).  aggregation='none': the fc is applied at every
 * position -> local logits [h, w, ncls].
 * Eval BatchNorm is FOLDED into the preceding conv here and in the HIP library (w' = w * scale[co] in float32, bias' =
 * shift[co]): one fmaf chain per output, then + bias', + residual, ReLU -- so that the two compare bit for bit; against
 * a plain PyTorch forward of the same architecture the difference is rounding (tests: 1e-4 relative). */
typedef struct { float *w1, *b1, *w2, *b2, *w3, *b3, *wd, *bd; int cin, planes, k, stride; } bag_block_t;

static float* fold_conv_bn(const float* w, int Co, int Ci, int KH, int KW, const float* bw, const float* bb, const float* bm,
                           const float* bv, float** bias_out) {
    bn_t bn = bn_fold(bw, bb, bm, bv, Co);
    float* scaled = (float*)malloc(sizeof(float) * (size_t)Co * Ci * KH * KW);
    for (int o = 0; o < Co; o++)
        for (size_t e = 0; e < (size_t)Ci * KH * KW; e++) scaled[(size_t)o * Ci * KH * KW + e] = w[(size_t)o * Ci * KH * KW + e] * bn.scale[o];
    float* p = repack_oihw(scaled, Co, Ci, KH, KW);
    free(scaled); free(bn.scale);
    *bias_out = bn.shift;
    return p;
}

/* spatial size of the local-logit map for a D x D input */
int or_bagnet_out_size(int D, const int* layers, const int* strides, const int* kernel3) {
    int H = D - 2;                                            /* conv2, 3x3 no padding */
    for (int g = 0; g < 4; g++)
        for (int i = 0; i < layers[g]; i++) {
            int k = kernel3[g] > i ? 3 : 1, s = i == 0 ? strides[g] : 1;
            H = (H - k) / s + 1;
        }
    return H;
}

int or_bagnet_forward(const float* const* t, int nt, const int* layers, const int* strides, const int* kernel3, int ncls,
                      const float* z, int N, int D, float* local, int nthreads) {
    if (!t || !layers || !strides || !kernel3 || !z || !local || N < 0 || D < 8 || ncls < 1 || ncls > 1024) return -1;
    int ti = 0;
    const float* wc1 = t[ti++];                               /* conv1.weight [64,3,1,1] */
    float *bc2, *wc2 = fold_conv_bn(t[ti], 64, 64, 3, 3, t[ti + 1], t[ti + 2], t[ti + 3], t[ti + 4], &bc2); ti += 5;
    int nb = layers[0] + layers[1] + layers[2] + layers[3];
    bag_block_t* blk = (bag_block_t*)calloc(nb, sizeof(bag_block_t));
    int inpl = 64, bi = 0;
    for (int g = 0; g < 4; g++)
        for (int i = 0; i < layers[g]; i++, bi++) {
            bag_block_t* b = &blk[bi];
            b->planes = 64 << g; b->cin = inpl; b->k = kernel3[g] > i ? 3 : 1; b->stride = i == 0 ? strides[g] : 1;
            b->w1 = fold_conv_bn(t[ti], b->planes, b->cin, 1, 1, t[ti + 1], t[ti + 2], t[ti + 3], t[ti + 4], &b->b1); ti += 5;
            b->w2 = fold_conv_bn(t[ti], b->planes, b->planes, b->k, b->k, t[ti + 1], t[ti + 2], t[ti + 3], t[ti + 4], &b->b2); ti += 5;
            b->w3 = fold_conv_bn(t[ti], 4 * b->planes, b->planes, 1, 1, t[ti + 1], t[ti + 2], t[ti + 3], t[ti + 4], &b->b3); ti += 5;
            if (i == 0) {                                     /* stride != 1 or inplanes != planes * 4: always true for block 0 */
                b->wd = fold_conv_bn(t[ti], 4 * b->planes, b->cin, 1, 1, t[ti + 1], t[ti + 2], t[ti + 3], t[ti + 4], &b->bd); ti += 5;
            }
            inpl = 4 * b->planes;
        }
    if (ti + 2 != nt) { free(blk); return -3; }
    float* wfc = repack_oihw(t[ti], ncls, inpl, 1, 1);
    const float* bfc = t[ti + 1];
    size_t big = (size_t)D * D * 64;
    { int H = D - 2;
      for (int q = 0; q < nb; q++) { size_t a = (size_t)H * H * 4 * blk[q].planes; if (a > big) big = a;
                                     H = (H - blk[q].k) / blk[q].stride + 1; } }
    if (nthreads < 1) nthreads = 1;
    const int Hl = or_bagnet_out_size(D, layers, strides, kernel3);
#pragma omp parallel num_threads(nthreads)
    {
        float* X = (float*)malloc(sizeof(float) * big);
        float* T1 = (float*)malloc(sizeof(float) * big);
        float* T2 = (float*)malloc(sizeof(float) * big);
        float* R = (float*)malloc(sizeof(float) * big);
#pragma omp for schedule(dynamic, 1)
        for (int n = 0; n < N; n++) {
            const float* zi = z + (size_t)n * 3 * D * D;
            for (int p = 0; p < D * D; p++)                   /* conv1: 1x1, 3 -> 64, no bias, no BN, no ReLU */
                for (int o = 0; o < 64; o++) {
                    float acc = 0.f;
                    for (int c = 0; c < 3; c++) acc = fmaf(zi[(size_t)c * D * D + p], wc1[o * 3 + c], acc);
                    T1[(size_t)p * 64 + o] = acc;
                }
            int H = D - 2;
            conv_nhwc(T1, D, D, 64, wc2, 3, 3, 1, 0, X, H, H, 64); bias_relu(X, (size_t)H * H, 64, bc2, 1);
            for (int q = 0; q < nb; q++) {
                bag_block_t* b = &blk[q];
                int Ho = (H - b->k) / b->stride + 1, Co = 4 * b->planes;
                conv_nhwc(X, H, H, b->cin, b->w1, 1, 1, 1, 0, T1, H, H, b->planes); bias_relu(T1, (size_t)H * H, b->planes, b->b1, 1);
                conv_nhwc(T1, H, H, b->planes, b->w2, b->k, b->k, b->stride, 0, T2, Ho, Ho, b->planes);
                bias_relu(T2, (size_t)Ho * Ho, b->planes, b->b2, 1);
                const float* res; int rstride;               /* residual rows are rstride pixels apart (cropped view) */
                if (b->wd) {                                  /* only the Ho x Ho corner that survives the crop is computed */
                    conv_nhwc(X, H, H, b->cin, b->wd, 1, 1, b->stride, 0, R, Ho, Ho, Co); bias_relu(R, (size_t)Ho * Ho, Co, b->bd, 0);
                    res = R; rstride = Ho;
                } else { res = X; rstride = H; }
                conv_nhwc(T2, Ho, Ho, b->planes, b->w3, 1, 1, 1, 0, T1, Ho, Ho, Co);
                for (int y = 0; y < Ho; y++)
                    for (int x2 = 0; x2 < Ho; x2++)
                        for (int o = 0; o < Co; o++) {
                            float v = T1[((size_t)y * Ho + x2) * Co + o] + b->b3[o];
                            v = v + res[((size_t)y * rstride + x2) * Co + o];
                            T1[((size_t)y * Ho + x2) * Co + o] = v > 0.f ? v : 0.f;
                        }
                float* tmp = X; X = T1; T1 = tmp;
                H = Ho;
            }
            conv_nhwc(X, H, H, inpl, wfc, 1, 1, 1, 0, T1, H, H, ncls);      /* x.permute(0,2,3,1); fc(x) */
            float* lo = local + (size_t)n * Hl * Hl * ncls;
            for (size_t p = 0; p < (size_t)H * H; p++)
                for (int c = 0; c < ncls; c++) lo[p * ncls + c] = T1[p * ncls + c] + bfc[c];
        }
        free(X); free(T1); free(T2); free(R);
    }
    free(wc2); free(bc2); free(wfc);
    for (int q = 0; q < nb; q++) {
        free(blk[q].w1); free(blk[q].b1); free(blk[q].w2); free(blk[q].b2); free(blk[q].w3); free(blk[q].b3);
        free(blk[q].wd); free(blk[q].bd);
    }
    free(blk);
    return 0;
}

/* ---- masking_defense(local_feature, thres, window_shape)  [THIRD PARTY, restated from the paper's Algorithm 1, unpinned] ----
 *   u = clip(local, 0, inf)                                        (clipping = -1 -> [0, inf))
 *   for every class c: total_c = sum u[:,:,c]; win_c = max over all window positions of sum u[x:x+wx, y:y+wy, c]
 *                      masked_c = total_c - win_c   if win_c / total_c > thres   (0/0 = nan -> not masked), else total_c
 *   prediction = LAST index of the maximum of masked (np.argsort(kind='stable')[-1])
 * Sums are float32, row-major (x outer, y inner), one chain per (window, class) -- the order the HIP kernel uses.
 * scores [N,ncls] = masked class evidence; preds [N]. */
int or_masking_defense(const float* local, int N, int h, int w, int ncls, int wx, int wy, float thres, float* scores,
                       int64_t* preds) {
    if (!local || N < 0 || h <= 0 || w <= 0 || ncls <= 0 || wx <= 0 || wy <= 0 || wx > h || wy > w) return -1;
    for (int n = 0; n < N; n++) {
        const float* u = local + (size_t)n * h * w * ncls;
        int best = 0; float bv = 0.f;
        for (int c = 0; c < ncls; c++) {
            float total = 0.f;
            for (int p = 0; p < h * w; p++) { float v = u[(size_t)p * ncls + c]; total += v > 0.f ? v : 0.f; }
            float win = -INFINITY;
            for (int x = 0; x + wx <= h; x++)
                for (int y = 0; y + wy <= w; y++) {
                    float s = 0.f;
                    for (int i = 0; i < wx; i++)
                        for (int j = 0; j < wy; j++) { float v = u[((size_t)(x + i) * w + y + j) * ncls + c]; s += v > 0.f ? v : 0.f; }
                    if (s > win) win = s;
                }
            float m = (win / total > thres) ? total - win : total;
            if (scores) scores[(size_t)n * ncls + c] = m;
            if (c == 0 || m >= bv) { bv = m; best = c; }
        }
        if (preds) preds[n] = best;
    }
    return 0;
}

/* PatchGuardNoDefense (patchguard_predict.py:56-89): aggregation='mean' over the positions, then np.argmax (first maximum).
 * scores [N,ncls] = sequential row-major float32 sum * (1 / (h*w)). */
int or_pg_mean_logits(const float* local, int N, int h, int w, int ncls, float* scores) {
    if (!local || !scores || N < 0 || h <= 0 || w <= 0 || ncls <= 0) return -1;
    for (int n = 0; n < N; n++)
        for (int c = 0; c < ncls; c++) {
            float s = 0.f;
            for (int p = 0; p < h * w; p++) s += local[((size_t)n * h * w + p) * ncls + c];
            scores[(size_t)n * ncls + c] = s * (1.0f / (float)(h * w));
        }
    return 0;
}
